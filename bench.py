#!/usr/bin/env python3
"""k-mers/s indexed (k=31 DNA CountIndex, 150 bp synthetic FASTQ) on N MI355X GPUs.

One step = one pass of the hot path over one batch: FASTQ bytes resident in HBM ->
k-mer extraction -> canonical strand -> (N>1: KeyToRank routing + RCCL all-to-all) ->
count-index build. N=1 workload = BASELINE.json configs[1] (config 2 of SURVEY.md 8d): 10 M reads
of a 100 Mbp genome, seed 2, 1.2e9 k-mers. N>1 is the weak-scaling family of configs[2] (config 3:
100 M reads of a 1 Gbp genome over 8 GPUs, seed 3): every rank gets 12.5 M reads, the genome is
N x 125 Mbp, so N=8 IS config 3 and the genome coverage (12x) is the same at every N.

`python bench.py --gpus N` without RANK/WORLD_SIZE in the environment starts the N ranks itself (fresh
child processes, one per GPU, before anything touches the GPU) and relays rank 0's line; under
torch.distributed.run it reads RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* as usual.

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
    # fused build through super-k-mers (kmi_superkmer.h): a record is 16 bytes for about 9.3 k-mers (0.108 records per k-mer,
    # 1.73 B per k-mer), an item 4 bytes per record, a run entry 4 bytes per read
    "sk_minimizer": 2.625 * 2 / 8 + 0.03 + 0.43 + 0.03,    # packed stream + run list in; items + per-run item index out (general front end)
    # one-pass front end (kmi_front.h): input bytes in; per run a 48-byte packed row + a 4-byte item index, per record a 4-byte item out
    "sk_front": 2.625 + 48.0 / 120 + 0.43 + 0.03,
    "sk_scatter": 48.0 / 120 + 0.43 + 0.03 + 1.73,         # rows + items + run index in; records out
    "sk_fine_count": 1.73,                                 # records in
    "sk_scatter_fine": 2 * 1.73,                           # records in, records out
    "sk_reduce": 1.73 + 12.0 / 12,                         # records in; (key, count) of each distinct key out
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
    ap.add_argument("--reads", type=int, default=None, help="reads per GPU (default: 10 M at N=1 = config 2; 12.5 M at N>1 = config 3 / 8)")
    ap.add_argument("--genome", type=int, default=None, help="genome bases per GPU (default: 100 Mbp at N=1; 125 Mbp at N>1)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--cpu-sample-reads", type=int, default=500_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extract-only / insert-only / query rates (outside the timed steps)")
    ap.add_argument("--dist-mode", default="auto", choices=["auto", "superkmer", "combine", "raw"],
                    help="N > 1: 'superkmer' exchanges 16-byte super-k-mer records (about nine k-mers each) and keeps a k-mer on the "
                         "owner of its minimizer's bucket; 'combine' reduces the rank's own reads first and exchanges (k-mer, count) "
                         "pairs; 'raw' routes every k-mer occurrence as the reference does (kmi_extract_route_dev + insert); 'auto': "
                         "super-k-mers over 2 / 4 / 8 ranks, else combine when the rank's own reads cover the genome at least 2.5 "
                         "times (the pairs are then a fraction of the occurrences), else raw")
    ap.add_argument("--transport", default="kmi", choices=["kmi", "torch"],
                    help="N > 1 with the super-k-mer exchange: 'kmi' = the product's own RCCL layer (kmi_comm_* + kmi_index_build_dist_dev: "
                         "chunked, the exchange of a chunk on its own stream beside the next chunk's front end: what Index::build_partition "
                         "runs); 'torch' = kmerind_amd.dist over torch.distributed")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (routing + all_to_all_single + insert) even with one rank: exercises the RCCL calls on one GPU")
    ap.add_argument("--chunks", type=int, default=4, help="N > 1: chunks per step (exchange of one overlaps parsing of the next)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal of the N>1 flow with ranks sharing one GPU (exchange staged through the host)")
    args = ap.parse_args()
    multi_default = args.gpus > 1 or args.force_dist
    if args.reads is None:
        args.reads = 12_500_000 if args.gpus > 1 else 10_000_000
    if args.genome is None:
        args.genome = 125_000_000 if args.gpus > 1 else 100_000_000
    del multi_default

    if args.gpus > 1 and "RANK" not in os.environ:
        # the driver's form: no launcher around us. Start the ranks as fresh children BEFORE this process touches the GPU
        # (it never does) and relay rank 0's line.
        sys.exit(launch_ranks(args.gpus))

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
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
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
    workload_tag = ""
    if world == 1 and (n_reads, genome_len) == (10_000_000, 100_000_000):
        workload_tag = " [BASELINE configs[1] = SURVEY config 2]"
    elif world > 1 and (n_reads, args.genome) == (12_500_000, 125_000_000):
        workload_tag = " [weak-scaling family of config 3: %d M reads, %d Mbp genome in all%s]" % (
            n_reads * world // 1_000_000, genome_len // 1_000_000, "; N=8 is config 3 verbatim" if world != 8 else " = config 3 verbatim")
    peer_counts, verified = None, False

    if args.force_dist:
        os.environ["KMI_FORCE_DIST"] = "1"                 # the library's collectives run their exchange with one rank too
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
    dist_mode = args.dist_mode if args.dist_mode != "auto" else \
        ("superkmer" if world in (2, 4, 8) else ("combine" if local_cov >= 2.5 else "raw"))
    combine = multi and dist_mode in ("combine", "superkmer")     # both run through kmerind_amd.dist.DistributedCountIndex
    nch = 1
    kmi_comm = None
    transport_note = None
    use_kmi = multi and dist_mode == "superkmer" and args.transport == "kmi" and args.backend == "nccl"
    if use_kmi:
        # the product's own exchange: an ncclUniqueId from rank 0 goes round once (torch.distributed is only the messenger
        # here, and the barrier / MAX of the timing contract), everything of the step is kmi_index_build_dist_dev
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            raw = (C.c_char * 128)()
            ctx.check(L.lib.kmi_comm_unique_id(raw))
            uid = torch.frombuffer(bytearray(bytes(raw)), dtype=torch.uint8).clone()
        if world > 1:
            uid_dev = uid.to(dev)
            dist.broadcast(uid_dev, src=0)
            uid = uid_dev.cpu()
        kmi_comm = C.c_void_p()
        st = L.lib.kmi_comm_create(ctx.h, (C.c_char * 128).from_buffer_copy(bytes(uid.numpy().tobytes())), C.byref(kmi_comm))
        # every rank must take the same transport: if the library's communicator could not be created on ANY rank, all of them say so
        # loudly and take the torch.distributed transport (the line's config.transport then reads "torch (...)": never silently)
        ok = torch.tensor([1 if st == 0 else 0], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            sys.stderr.write("bench.py rank %d: kmi_comm_create failed on a rank (status here: %d, %s); all ranks fall back to "
                             "--transport torch\n" % (rank, st, (L.lib.kmi_last_error(ctx.h) or b"").decode() if st else "ok"))
            if st == 0:
                L.lib.kmi_comm_destroy(kmi_comm)
            kmi_comm, use_kmi, transport_note = None, False, "torch (kmi_comm_create failed on a rank: see stderr)"
        else:
            nch = int(os.environ.get("KMI_DIST_CHUNKS", "4"))
            combine = False
    if combine:
        # N > 1, combine-first (kmerind_amd.dist.DistributedCountIndex): local count index of the rank's reads (the one-rank
        # pipeline), split by KeyToRank, all_to_all_single of (k-mer, count) pairs + per-bucket counts, merge.
        didx = kdist.DistributedCountIndex(ctx, cfg, stage_through_host=(args.backend != "nccl"), device=dev)
        idx.close()
        idx = didx.index
        sk_bounds = None
        if dist_mode == "superkmer":
            # the step goes through in record-aligned chunks: the records of chunk c travel over xGMI while chunk c + 1 is cut
            # into super-k-mers (chunk starts on multiples of 16 records = 16-byte aligned addresses)
            nch = max(1, min(args.chunks, n_reads // 16))
            rec_bytes = nbytes // n_reads
            sk_bounds = [((n_reads * c // nch) // 16 * 16) * rec_bytes for c in range(nch)] + [nbytes]
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
        nonlocal peer_counts, verified
        idx.clear()
        if not multi:
            idx.build_device(d_bytes.data_ptr(), nbytes)
            return
        if use_kmi:
            ctx.check(L.lib.kmi_index_build_dist_dev(idx.h, kmi_comm, C.c_void_p(d_bytes.data_ptr()), nbytes, 0))
            verified = True
            return
        if combine:
            didx.build_device(d_bytes.data_ptr(), nbytes, dev, mode=dist_mode, bounds=sk_bounds)     # (its first exchange carries checksums)
            verified = True
            return
        pos, works = 0, []
        peer_counts = [0] * world
        for c in range(nch):
            if c >= 2:
                works[c - 2].wait()                        # the send buffer about to be rewritten has left
            send = d_send[c & 1]
            ctx.check(L.lib.kmi_extract_route_dev(ctx.h, C.byref(cfg), C.c_void_p(d_bytes.data_ptr() + bounds[c]),
                                                  bounds[c + 1] - bounds[c], world, C.c_void_p(send.data_ptr()), send.shape[0],
                                                  C.byref(nt), C.byref(ns), counts.ctypes.data_as(C.c_void_p)))
            sc = [int(x) for x in counts]
            peer_counts = [a + b for a, b in zip(peer_counts, sc)]
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
            if not verified:                               # first exchange of the process: checksums travel with the payload
                works[-1].wait()
                kdist.verify_exchange(send[: nt.value], sc, d_recv[pos:pos + n_in], rc, stage_through_host=(args.backend != "nccl"))
                verified = True
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

    # peer balance of the exchange (max / mean message of this rank's last step; worst rank reported)
    peer_ratio = None
    if multi:
        sc_last = (didx.last_send_counts if combine else peer_counts) or [0]
        mean = float(sum(sc_last)) / max(1, len(sc_last))
        t = torch.tensor([max(sc_last) / mean if mean > 0 else 1.0], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        peer_ratio = round(float(t.item()), 4)
        group_ranks = dist.get_world_size()
        backend_name = dist.get_backend()

    if rank == 0:
        total_kmers = n_kmers * world
        ms_per_step = elapsed * 1e3 / args.steps
        value = total_kmers * args.steps / elapsed
        prof = [p for p in prof if p["launches"] > 0]
        dom = max(prof, key=lambda p: p["total_ms"]) if prof else None
        # roofline: the WHOLE step against the contract figure (B_alg = 10.625 B per k-mer, input bytes once + the
        # materialised tuple once) per GPU; the dominant kernel priced with its OWN algorithmic bytes sits under "dominant_kernel"
        step_gbs = n_kmers * B_ALG / (ms_per_step * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(None, n_reads) if not multi else (None, None)
        roofline = {"bound": "hbm", "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(step_gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "scope": "whole step per GPU: k-mers x 10.625 B / step time",
                    "alg_bytes_per_kmer": B_ALG, "alg_bytes_per_step": n_kmers * B_ALG}
        if dom:
            avg_ms = dom["total_ms"] / dom["launches"]
            b_kernel = KERNEL_ALG_BYTES.get(dom["name"], B_ALG)
            per_launch = n_kmers * args.steps / dom["launches"]      # k-mers one launch processes (N > 1 runs in chunks)
            achieved = per_launch * b_kernel / (avg_ms * 1e-3) / 1e9
            ktraffic, ksrc = measured_traffic(dom["name"], n_reads) if not multi else (None, None)
            roofline["dominant_kernel"] = {"kernel": dom["name"], "avg_kernel_ms": round(avg_ms, 4), "alg_bytes_per_kmer": b_kernel,
                                           "alg_bytes_per_launch": per_launch * b_kernel, "achieved": round(achieved, 1),
                                           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": ktraffic, "traffic_source": ksrc,
                                           "timing": "HIP events on the launch stream inside the timed region"}
        roofline["kernels_ms_per_step"] = {p["name"]: round(p["total_ms"] / args.steps, 4) for p in
                                           sorted(prof, key=lambda p: -p["total_ms"])}
        # what this box's HBM delivers to plain streaming kernels (outside the timed steps), next to the 8 TB/s figure
        roofline["measured_stream_gbs"] = measured_stream(torch, dev)
        out = {"metric": "kmers_per_sec_indexed", "value": value, "unit": "k-mers/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
               "scaling": "weak" if world > 1 else None, "vs_baseline": None, "dtype": "u64", "data": "synthetic",
               "config": {"input_residency": "hbm (the timed step starts from FASTQ bytes resident in device memory; the host-resident rate is extra.host_resident_kmers_per_s)",
                          "workload": "k=%d DNA CountIndex (canonical), %d synthetic %d-bp reads per GPU as 315-byte "
                                      "FASTQ records, genome %d bp, seed %d%s" % (k, n_reads, read_len, genome_len, seed, workload_tag),
                          "kmers_per_step": total_kmers, "distinct_kmers": distinct,
                          "index_form": "sparse (the entries of a bucket sit in its output slots; ensure_dense on the first export / insert / erase)",
                          "exchange": "none (1 rank)" if not multi else
                          ("kmi_comm (the library's RCCL layer): grouped ncclSend / ncclRecv of 16-byte super-k-mer records (owner = the "
                           "minimizer bucket's rank), %d record-aligned chunks per step, the exchange of a chunk on its own stream beside "
                           "the next chunk's front end (kmi_index_build_dist_dev)" % nch) if use_kmi else
                          "%s all_to_all_single (counts + payload), %d chunks per step, overlapped with parsing" %
                          ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)", nch) if not combine else
                          ("%%s all_to_all_single of 16-byte super-k-mer records (owner = the minimizer bucket's rank), %d chunks per step, "
                           "overlapped with the next chunk's front end" % nch if dist_mode == "superkmer"
                           else "%s all_to_all_single of locally reduced (k-mer, count) pairs + per-bucket counts") %
                          ("RCCL" if args.backend == "nccl" else "gloo (rehearsal)")},
               "roofline": roofline}
        if multi:
            out["config"].update({"dist_mode": dist_mode, "transport": "kmi" if use_kmi else (transport_note or "torch"), "backend": backend_name, "rccl_ranks": group_ranks if args.backend == "nccl" else 0,
                                  "group_ranks": group_ranks, "peer_bucket_max_over_mean": peer_ratio,
                                  "exchange_checksum": "verified on the first exchange" if verified else "not run"})
        if not multi and not args.no_extra:
            out["extra"] = extra_rates(ctx, cfg, idx, d_bytes, nbytes, n_kmers, dev, torch, host)
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, args, k, n_reads)
        print(json.dumps(out), flush=True)

    if kmi_comm is not None:
        L.lib.kmi_comm_destroy(kmi_comm)
    if combine:
        didx.close()
    else:
        idx.close()
    ctx.close()
    if multi:
        dist.destroy_process_group()


def launch_ranks(n):
    """`python bench.py --gpus N` as the driver runs it: N child ranks of this same script (one per GPU, RCCL over xGMI between
    them), rank 0's stdout relayed, non-zero exit if any rank fails. The parent never initialises the GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading
    chunks = []
    rd = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    deadline = time.time() + 1800
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for r, pr in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = pr.poll()
        failed = any(rc not in (None, 0) for rc in rcs)
        if failed or time.time() > deadline:                    # a dead rank leaves the others waiting in a collective
            for r, pr in enumerate(procs):
                if rcs[r] is None:
                    pr.kill()
                    rcs[r] = pr.wait()
            break
        time.sleep(0.2)
    rd.join(timeout=10)
    out0 = "".join(chunks)
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed: %s\n" % bad)
        return 1
    return 0


def extra_rates(ctx, cfg, idx, d_bytes, nbytes, n_kmers, dev, torch, host=None):
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
    # the same build from bytes resident in HOST memory (SURVEY 8d defines the wall from there): pinned host buffer ->
    # H2D copy on the build's stream -> build. PCIe-bound; reported next to `value`, never as `value`.
    host_rate, host_note = None, "not measured"
    if host is not None:
        try:
            h_pinned = torch.from_numpy(host).pin_memory()
            idx3 = K.CountIndex(ctx, cfg)

            def host_build_serial():
                idx3.clear()
                d_bytes.copy_(h_pinned, non_blocking=True)
                idx3.build_device(d_bytes.data_ptr(), nbytes)

            def host_build():   # kmi_index_build_host: the copy in chunks on its own stream, the front end's byte ranges behind every chunk
                idx3.clear()
                ctx.check(L.lib.kmi_index_build_host(idx3.h, C.c_void_p(h_pinned.data_ptr()), nbytes, 0))

            def copy_only():
                d_bytes.copy_(h_pinned, non_blocking=True)
            t_serial = timed(host_build_serial)
            t_copy = timed(copy_only)
            t_host = timed(host_build)
            assert idx3.local_size() == idx.local_size()
            idx3.close()
            del h_pinned
            host_rate = n_kmers / t_host
            host_note = ("pinned host buffer through kmi_index_build_host (chunked H2D copy, the front end's ranges behind every chunk): %.1f ms; "
                         "the copy alone %.1f ms (%.1f GB/s); one copy, then the build: %.1f ms" % (t_host * 1e3, t_copy * 1e3, nbytes / t_copy / 1e9, t_serial * 1e3))
        except Exception as e:   # pinning 3 GB can fail on a small box: the side measurement must not break the bench line
            host_note = "failed: " + str(e)[:80]
    # what the context carries from one build to the next (workspace blocks; sk_reduce's pass structure and duplication hints):
    # (a) the FIRST build of a fresh context, allocations included; (b) a build of the warm context with the hints forgotten
    cold = {}
    try:
        def fresh_context_build():
            """first build of a new context beside the warm one; returns (ms, ms inside hipMalloc / hipFree, GB that reached hipMalloc, cached blocks reused)"""
            ctx2 = K.Context(device=ctx.device, stream=ctx.stream)
            idx4 = K.CountIndex(ctx2, cfg)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            idx4.build_device(d_bytes.data_ptr(), nbytes)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) * 1e3
            v = C.c_uint64()
            c = []
            for w in (1, 2, 4):
                ctx2.check(L.lib.kmi_ctx_debug_counter(ctx2.h, w, C.byref(v)))
                c.append(v.value)
            idx4.close(); ctx2.close()
            return ms, c[0] / 1e3, c[1] / 1e9, c[2]
        # (a1) every block of its workspace comes from hipMalloc: what that costs is the node's business (1.4 ms for 30 GB on an idle
        # box, hundreds of ms on a loaded one) and is reported beside the total; (a2) the next context finds the blocks the first
        # one left in the library's process-wide cache (kmi_release_cached_memory): no hipMalloc in its first build
        ms, in_malloc, gb, _ = fresh_context_build()
        cold["first_build_of_a_fresh_context_ms"] = round(ms, 2)
        cold["of_which_inside_hipmalloc_ms"] = round(in_malloc, 2)
        cold["hipmalloc_gb"] = round(gb, 2)
        ms, in_malloc, gb, reused = fresh_context_build()
        cold["first_build_of_a_recycled_context_ms"] = round(ms, 2)
        cold["recycled_context_blocks_from_cache"] = int(reused)
        L.lib.kmi_release_cached_memory(-1, None)

        def no_hints():
            idx.clear()
            ctx.check(L.lib.kmi_ctx_reset_hints(ctx.h))
            idx.build_device(d_bytes.data_ptr(), nbytes)
        cold["build_without_carried_hints_ms"] = round(timed(no_hints) * 1e3, 3)
        idx.clear(); idx.build_device(d_bytes.data_ptr(), nbytes)      # (leaves the index as the timed steps left it)
    except Exception as e:
        cold["error"] = str(e)[:80]
    return {"cold": cold, "host_resident_kmers_per_s": host_rate, "host_resident_note": host_note,
            "extract_only_kmers_per_s": n_kmers / t_extract, "insert_only_kmers_per_s": n_kmers / t_insert,
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
    """(HBM bytes, source file) from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE per launch,
    profiles/*_hbm_traffic.json, collected on config 2 by tools/profile_round.sh): of one launch of `kernel`, or of the whole
    step (all kernels) for kernel=None. These are NOT counters of this run; (None, None) for other workloads."""
    import glob
    if n_reads != 10_000_000:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None, None
    src = "profiles/" + os.path.basename(files[-1])
    try:
        with open(files[-1]) as f:
            kernels = json.load(f)["kernels"]
        if kernel is None:
            return sum(r["hbm_bytes"] for r in kernels.values()), src
        rec = kernels.get(kernel)
        return (rec["hbm_bytes"], src) if rec else (None, None)
    except (OSError, ValueError, KeyError):
        return None, None


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
