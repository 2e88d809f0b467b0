#!/usr/bin/env python3
"""k-mers/s indexed (k=31 DNA CountIndex, 150 bp synthetic FASTQ) on N MI355X GPUs.

One step = one pass of the hot path over one batch: FASTQ bytes resident in HBM ->
k-mer extraction -> canonical strand -> (N>1: KeyToRank routing + RCCL all-to-all) ->
count-index build. N=1 workload = BASELINE.json configs[1]: 10 M reads / 1.2e9 k-mers.
N>1 is weak scaling: every rank gets its own 10 M reads of a genome N times as long.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the
launch stream inside the timed region) and `cpu_baseline` (the oracle's restatement of
the reference MPI path on the host cores, bounded sample)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 10.625          # whole-job contract figure, HBM bytes read per k-mer: 315/120 + 8
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md)
# Algorithmic HBM bytes per k-mer of each kernel of the fused build: what the kernel
# must read + write once, for config 2 (k=31, 315-byte records -> 2.625 input bytes per k-mer, u = distinct /
# total = 1/12). The roofline entry prices the dominant kernel with its own figure.
KERNEL_ALG_BYTES = {
    "fastq_scan_tiles": 2.625 + 2.625 * (2 + 1) / 8,       # raw bytes in; 2-bit stream + EOL bitmap out
    "fastq_list": 2.625 / 8 + 0.25,                        # EOL bitmap in; one 2-byte entry per 8 windows out
    "fastq_hist": 0.25 + 2.625 * 2 / 8,                    # entry list + packed stream in
    "fastq_scatter": 0.25 + 2.625 * 2 / 8 + 8.0,           # entries + stream in; 8-byte key out
    "scatter_fine": 16.0,                                  # key in, key out
    "bucket_reduce": 8.0 + 12.0 / 12,                      # key in; (key, count) of each distinct key out
    "bucket_compact": 2 * 12.0 / 12,
    # N > 1 and insert-from-keys paths
    "fastq_rank_hist": 0.25 + 2.625 * 2 / 8 + 1.0,         # + one rank-bucket byte per window out
    "fastq_rank_scatter": 1.25 + 2.625 * 2 / 8 + 8.0,
    "hist_fine": 8.0,
    # combine-first exchange (units are index entries, not k-mers)
    "split_count": 9.0, "split_scatter": 25.0, "bucket_merge": 24.0,
    "scatter_coarse": 16.0,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (config 2: 10 M)")
    ap.add_argument("--genome", type=int, default=100_000_000, help="genome bases per GPU")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--cpu-sample-reads", type=int, default=500_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extract-only / insert-only / query rates (outside the timed steps)")
    ap.add_argument("--dist-mode", default="auto", choices=["auto", "combine", "raw"],
                    help="N > 1: 'combine' reduces the rank's own reads first and exchanges (k-mer, count) pairs; 'raw' routes "
                         "every k-mer occurrence as the reference does (kmi_extract_route_dev + insert); 'auto' combines when the "
                         "rank's own reads cover the genome at least 2.5 times (the pairs are then a fraction of the occurrences)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (routing + all_to_all_single + insert) even with one rank: exercises the RCCL calls on one GPU")
    ap.add_argument("--chunks", type=int, default=4, help="N > 1: chunks per step (exchange of one overlaps parsing of the next)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 flow with ranks sharing one GPU (exchange staged through the host)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import kmerind_amd as K
    from kmerind_amd import _lib as L
    from kmerind_amd import dist as kdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        if args.force_dist and "RANK" not in os.environ:      # one-rank rehearsal started without torch.distributed.run
            for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29541")):
                os.environ.setdefault(key, val)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    k, read_len = args.k, 150
    kmers_per_read = read_len - k + 1
    n_reads = args.reads
    genome_len = args.genome * world
    seed = 2 if not multi else 3

    stream = torch.cuda.current_stream(dev)
    ctx = K.Context(device=local_rank, rank=rank, nranks=world, stream=stream.cuda_stream)
    cfg = K.make_config(k, "DNA", strand="canonical", dist_hash="murmur", store_hash="murmur")

    host = K.synth_fastq(seed, genome_len, n_reads, read_len, first_read=rank * n_reads)
    nbytes = int(host.nbytes)
    d_bytes = torch.from_numpy(host).to(dev)
    idx = K.CountIndex(ctx, cfg)
    n_kmers = n_reads * kmers_per_read

    # expected k-mer coverage of the genome by ONE rank's reads: what a local reduction can take out before the exchange
    local_cov = n_kmers / float(genome_len)
    dist_mode = args.dist_mode if args.dist_mode != "auto" else ("combine" if local_cov >= 2.5 else "raw")
    combine = multi and dist_mode == "combine"
    nch = 1
    if combine:
        # N > 1, combine-first (kmerind_amd.dist.DistributedCountIndex): local count index of the rank's reads (the one-rank
        # pipeline), split by KeyToRank, all_to_all_single of (k-mer, count) pairs + per-bucket counts, merge.
        didx = kdist.DistributedCountIndex(ctx, cfg, stage_through_host=(args.backend != "nccl"), device=dev)
        idx.close()
        idx = didx.index
    elif multi:
        # N > 1: the batch goes through in NCH record-aligned chunks. Chunk c is parsed and grouped by destination rank
        # on the device (kmi_extract_route_dev: read_file + the bucketing half of imxx::distribute, fused) while the
        # all-to-all of chunk c-1 is still travelling over xGMI on RCCL's stream; every rank then inserts what it
        # received in one go. Send buffers are double-buffered, the receive buffer takes the chunks back to back.
        # this RCCL build corrupts a peer message above 2^27 eight-byte elements (1 GiB; measured with a self send,
        # tools/a2a_debug.py), so the chunk count also keeps every per-peer message below that with 10 % headroom
        MSG_MAX = 1 << 27
        need = -(-int(n_kmers * 1.1) // (world * MSG_MAX))
        nch = max(1, min(max(args.chunks, need), n_reads))
        rec_bytes = nbytes // n_reads                      # synthetic records have one size (315 bytes)
        # chunk starts stay 16-byte aligned (the byte kernels load 16 bytes per lane): whole multiples of 16 records
        bounds = [((n_reads * c // nch) // 16 * 16) * rec_bytes for c in range(nch)] + [nbytes]
        chunk_kmers = max(((bounds[c + 1] - bounds[c]) // rec_bytes) * kmers_per_read for c in range(nch))
        d_send = [torch.empty((chunk_kmers + 64, 1), dtype=torch.int64, device=dev) for _ in range(2)]
        recv_cap = int(n_kmers * 1.25) + 4096              # murmur % p is balanced to a fraction of a percent
        d_recv = torch.empty((recv_cap, 1), dtype=torch.int64, device=dev)
        counts = np.zeros(world, dtype=np.uint64)
        nt, ns = C.c_uint64(), C.c_uint64()
        cdev = dev if args.backend == "nccl" else torch.device("cpu")

    def step():
        idx.clear()
        if not multi:
            idx.build_device(d_bytes.data_ptr(), nbytes)
            return
        if combine:
            didx.build_device(d_bytes.data_ptr(), nbytes, dev)
            return
        pos, works = 0, []
        for c in range(nch):
            if c >= 2:
                works[c - 2].wait()                        # the send buffer about to be rewritten has left
            send = d_send[c & 1]
            ctx.check(L.lib.kmi_extract_route_dev(ctx.h, C.byref(cfg), C.c_void_p(d_bytes.data_ptr() + bounds[c]),
                                                  bounds[c + 1] - bounds[c], world, C.c_void_p(send.data_ptr()), send.shape[0],
                                                  C.byref(nt), C.byref(ns), counts.ctypes.data_as(C.c_void_p)))
            sc = [int(x) for x in counts]
            rc = kdist.exchange_counts(sc, device=cdev)
            n_in = int(sum(rc))
            if pos + n_in > recv_cap:
                raise SystemExit("receive buffer too small: %d > %d" % (pos + n_in, recv_cap))
            if args.backend == "nccl" and max(max(sc), max(rc)) > MSG_MAX:
                raise SystemExit("a peer message of %d keys exceeds what this RCCL build moves correctly; raise --chunks" % max(max(sc), max(rc)))
            if args.backend == "nccl":
                works.append(dist.all_to_all_single(d_recv[pos:pos + n_in], send[: nt.value], output_split_sizes=rc,
                                                    input_split_sizes=sc, async_op=True))
            else:                                          # rehearsal: ranks share one GPU, payload staged through the host
                r_host = torch.empty((n_in, 1), dtype=torch.int64)
                dist.all_to_all_single(r_host, send[: nt.value].cpu(), output_split_sizes=rc, input_split_sizes=sc)
                d_recv[pos:pos + n_in].copy_(r_host)

                class _Done:
                    def wait(self):
                        return True
                works.append(_Done())
            pos += n_in
        for w in works[-2:]:
            w.wait()
        idx.insert_device(d_recv.data_ptr(), pos, transformed=True)      # routed keys are already canonical

    def sync():
        torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    ctx.profile(True)
    ctx.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = ctx.profile_get()
    ctx.profile(False)

    local_distinct = idx.local_size()
    if multi:
        cdev = dev if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        distinct = kdist.global_size(local_distinct, device=cdev)
    else:
        distinct = local_distinct

    if rank == 0:
        total_kmers = n_kmers * world
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_kmers * args.steps / elapsed
        prof = [p for p in prof if p["launches"] > 0]
        dom = max(prof, key=lambda p: p["total_ms"]) if prof else None
        roofline = None
        if dom:
            avg_ms = dom["total_ms"] / dom["launches"]
            b_kernel = KERNEL_ALG_BYTES.get(dom["name"], B_ALG)
            per_launch = n_kmers * args.steps / dom["launches"]      # k-mers one launch processes (N > 1 runs in chunks)
            achieved = per_launch * b_kernel / (avg_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": dom["name"], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": measured_traffic(dom["name"], n_reads),
                        "avg_kernel_ms": round(avg_ms, 4),
                        "alg_bytes_per_kmer": b_kernel,
                        "alg_bytes_per_launch": per_launch * b_kernel,
                        "pipeline_frac": round(n_kmers * B_ALG / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "kernels_ms_per_step": {p["name"]: round(p["total_ms"] / args.steps, 4) for p in
                                                sorted(prof, key=lambda p: -p["total_ms"])},
                        # what this box's HBM delivers to plain streaming kernels (outside the timed steps), next to the 8 TB/s figure
                        "measured_stream_gbs": measured_stream(torch, dev)}
        out = {"metric": "kmers_per_sec_indexed", "value": value, "unit": "k-mers/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"workload": "k=%d DNA CountIndex (canonical), %d synthetic %d-bp reads per GPU as 315-byte "
                                      "FASTQ records, genome %d bp, seed %d" % (k, n_reads, read_len, genome_len, seed),
                          "kmers_per_step": total_kmers, "distinct_kmers": distinct,
                          "exchange": "none (1 rank)" if not multi else
                          "%s all_to_all_single (counts + payload), %d chunks per step, overlapped with parsing" %
                          ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)", nch) if not combine else
                          "%s all_to_all_single of locally reduced (k-mer, count) pairs + per-bucket counts" %
                          ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)")},
               "roofline": roofline}
        if not multi and not args.no_extra:
            out["extra"] = extra_rates(ctx, cfg, idx, d_bytes, nbytes, n_kmers, dev, torch)
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, args, k, n_reads)
        print(json.dumps(out), flush=True)

    if combine:
        didx.close()
    else:
        idx.close()
    ctx.close()
    if multi:
        dist.destroy_process_group()


def extra_rates(ctx, cfg, idx, d_bytes, nbytes, n_kmers, dev, torch):
    """The other sections the reference's benchmark times (BenchmarkKmerIndex.cpp:526-581), outside the timed steps, two
    runs each, HBM-resident operands: read (extract only), insert (from an extracted tuple array), count and find of
    10 M query k-mers (5 M present, 5 M random 62-bit values)."""
    import kmerind_amd as K
    from kmerind_amd import _lib as L

    def timed(fn, reps=2):
        fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / reps

    d_keys = torch.empty((n_kmers + 64, 1), dtype=torch.int64, device=dev)
    nt, ns = C.c_uint64(), C.c_uint64()

    def extract():
        ctx.check(L.lib.kmi_extract_dev(ctx.h, C.byref(cfg), C.c_void_p(d_bytes.data_ptr()), nbytes, 0, C.c_void_p(d_keys.data_ptr()),
                                        None, n_kmers, C.byref(nt), C.byref(ns)))
    t_extract = timed(extract)
    idx2 = K.CountIndex(ctx, cfg)

    def insert():
        idx2.clear()
        idx2.insert_device(d_keys.data_ptr(), nt.value)
    t_insert = timed(insert)
    idx2.close()
    nq = min(10_000_000, n_kmers)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    q = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    q[: nq // 2] = d_keys[torch.randint(0, n_kmers, (nq // 2,), device=dev, generator=g)]
    q[nq // 2:] = torch.randint(0, 1 << 62, (nq - nq // 2, 1), device=dev, generator=g, dtype=torch.int64)
    ok = torch.empty((nq, 1), dtype=torch.int64, device=dev)
    ov = torch.empty((nq,), dtype=torch.int64, device=dev)
    n_out = C.c_uint64()

    def count():
        ctx.check(L.lib.kmi_index_count_dev(idx.h, C.c_void_p(q.data_ptr()), nq, C.c_void_p(ok.data_ptr()), C.c_void_p(ov.data_ptr()),
                                            C.byref(n_out)))
    t_count = timed(count)
    n_distinct_q = n_out.value

    def find():
        ctx.check(L.lib.kmi_index_find_dev(idx.h, C.c_void_p(q.data_ptr()), nq, C.c_void_p(ok.data_ptr()), C.c_void_p(ov.data_ptr()),
                                           C.byref(n_out)))
    t_find = timed(find)
    return {"extract_only_kmers_per_s": n_kmers / t_extract, "insert_only_kmers_per_s": n_kmers / t_insert,
            "count_queries_per_s": nq / t_count, "find_queries_per_s": nq / t_find, "queries": nq,
            "distinct_query_keys": n_distinct_q, "found": n_out.value,
            "note": "outside the timed steps; 2 runs each after one warm-up; operands resident in HBM"}


def measured_stream(torch, dev):
    """read-only (sum) and copy (read + write) rates of 2 GiB int64 tensors, best of 5, HIP-event timed"""
    try:
        n = 1 << 28
        x = torch.ones(n, dtype=torch.int64, device=dev)
        y = torch.empty_like(x)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = {"read": 0.0, "copy": 0.0}
        for _ in range(5):
            ev[0].record(); x.sum(); ev[1].record(); ev[1].synchronize()
            best["read"] = max(best["read"], n * 8 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9)
            ev[0].record(); y.copy_(x); ev[1].record(); ev[1].synchronize()
            best["copy"] = max(best["copy"], 2 * n * 8 / (ev[0].elapsed_time(ev[1]) * 1e-3) / 1e9)
        del x, y
        return {k: round(v, 1) for k, v in best.items()}
    except Exception as e:   # never let the side measurement break the bench line
        return {"error": str(e)[:80]}


def measured_traffic(kernel, n_reads):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    profiles/*_hbm_traffic.json, collected on config 2); None for other workloads or kernels."""
    import glob
    if n_reads != 10_000_000:
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            rec = json.load(f)["kernels"].get(kernel)
        return rec["hbm_bytes"] if rec else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(host, args, k, n_reads):
    """oracle = CPU restatement of the reference path (thread ranks + in-memory all-to-all),
    timed on the host cores over the first `cpu_sample_reads` reads of the same FASTQ."""
    from tests import oracle as orc
    sample = min(args.cpu_sample_reads, n_reads)
    # the GPU box gives one GPU a 16-core CPU share even though more logical CPUs are visible
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    rec = 315
    sec, nk, nd = orc.bench_count_index(host[: sample * rec], k, orc.CANONICAL, cores)
    return {"value": nk / sec, "unit": "k-mers/s", "cores": cores, "kind": "port",
            "sample": "first %d reads (%d k-mers) of the same FASTQ; %d thread-ranks, murmur KeyToRank + in-memory "
                      "all-to-all + chained hash map; %.2f s" % (sample, nk, cores, sec)}


if __name__ == "__main__":
    main()
