"""bench.py prints ONE JSON line with the driver's contract fields (a small workload; the numbers are not judged here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--reads", "200000",
                          "--genome", "2000000", "--cpu-sample-reads", "20000"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "kmers_per_sec_indexed" and d["unit"] == "k-mers/s" and d["n_gpus"] == 1
    assert d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] is None   # one GPU: no scaling claim
    assert d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["kmers_per_step"] == 200000 * 120
    assert d["value"] > 0 and abs(d["value"] - d["config"]["kmers_per_step"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the headline fraction is the WHOLE step against the contract figure (10.625 B per k-mer), the dominant kernel sits below it
    assert abs(r["achieved"] - d["config"]["kmers_per_step"] * 10.625 / (d["ms_per_step"] * 1e-3) / 1e9) / r["achieved"] < 1e-2
    dk = r["dominant_kernel"]
    assert dk["kernel"] in r["kernels_ms_per_step"] and dk["avg_kernel_ms"] > 0 and dk["avg_kernel_ms"] <= d["ms_per_step"]
    assert d["extra"]["host_resident_kmers_per_s"] is None or 0 < d["extra"]["host_resident_kmers_per_s"] < d["value"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1
    # the small genome is covered 15x: every distinct canonical 31-mer of 2 Mbp (about 2e6) is in the index
    assert 1_900_000 < d["config"]["distinct_kmers"] <= 2_000_000


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's form): the script starts two ranks itself and relays
    rank 0's single line. Rehearsal on one GPU: gloo group, both ranks share the card, payload staged through the host."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for mode in ("superkmer", "combine", "raw"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                              "--reads", "100000", "--genome", "1000000", "--dist-mode", mode], capture_output=True, text=True, timeout=900,
                             cwd=ROOT, env=env)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["group_ranks"] == 2 and d["config"]["dist_mode"] == mode
        assert d["config"]["kmers_per_step"] == 2 * 100000 * 120 and d["config"]["exchange_checksum"].startswith("verified")
        assert 1.0 <= d["config"]["peer_bucket_max_over_mean"] < 1.1
        # 2 Mbp genome covered 12x by both ranks' reads together
        assert 1_900_000 < d["config"]["distinct_kmers"] <= 2_000_000


@pytest.mark.parametrize("transport", ["kmi", "torch"])
def test_bench_one_rank_rccl_rehearsal_of_the_super_kmer_exchange(transport):
    """`bench.py --force-dist --dist-mode superkmer`: the N > 1 flow with one rank over RCCL (self exchange): records produced in
    chunks, asynchronous exchange, consume -- the calls the 8-GPU run makes. transport kmi (the default): the library's own RCCL
    layer and kmi_index_build_dist_dev (what Index::build_partition runs); torch: kmerind_amd.dist over torch.distributed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--dist-mode", "superkmer", "--transport", transport,
                          "--steps", "2", "--warmup", "1", "--reads", "200000", "--genome", "2000000", "--no-cpu-baseline", "--no-extra"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["config"]["dist_mode"] == "superkmer" and d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1
    assert d["config"]["transport"] == transport
    assert d["config"]["exchange_checksum"].startswith("verified") and 1_900_000 < d["config"]["distinct_kmers"] <= 2_000_000
    assert "sk_recv_scatter" in d["roofline"]["kernels_ms_per_step"]
