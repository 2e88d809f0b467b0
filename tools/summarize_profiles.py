"""gpurun_out/<tag>/ (written by tools/profile_round.sh) -> profiles/<tag>_{kernel_stats.csv, bench.json,
bench_under_rocprof.json, hbm_traffic.json}.   usage: python tools/summarize_profiles.py r01_g"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")

# short names as bench.py's ProfScope uses them
SHORT = [("fastq_scan_tiles_kernel", "fastq_scan_tiles"), ("fastq_list_kernel", "fastq_list"), ("fastq_hist_list_kernel", "fastq_hist"),
         ("fastq_scatter_list_kernel", "fastq_scatter"), ("scatter_fine_lines_kernel", "scatter_fine"), ("bucket_reduce_kernel", "bucket_reduce"),
         ("bucket_compact_kernel", "bucket_compact"), ("fine_offsets_kernel", "fine_offsets"),
         ("sk_front_kernel", "sk_front"), ("sk_scatter_rows_kernel", "sk_scatter"),
         ("sk_scatter_fine_slack_lines_kernel", "sk_scatter_fine"), ("sk_scatter_fine_slack_kernel", "sk_scatter_fine"), ("sk_minimizer_kernel", "sk_minimizer"), ("sk_scatter_kernel", "sk_scatter_general"), ("sk_fine_count_kernel", "sk_fine_count"),
         ("scatter_fine_kernel", "sk_scatter_fine"), ("sk_reduce_kernel", "sk_reduce")]


def find(pattern):
    f = glob.glob(os.path.join(src, pattern), recursive=True)
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


shutil.copy(find("stats/**/*kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
for name in ("bench.json", "bench_under_rocprof.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, tag + "_" + name))


def counter_sums(sub, counter):
    sums, launches = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(find(sub + "/**/*counter_collection.csv"))):
        if r["Counter_Name"] == counter:
            sums[r["Kernel_Name"]] += float(r["Counter_Value"])
            launches[r["Kernel_Name"]] += 1
    return {k: sums[k] / launches[k] for k in sums}


fetch, write = counter_sums("fetch", "FETCH_SIZE"), counter_sums("write", "WRITE_SIZE")
kernels = {}
for needle, short in SHORT:
    full = [k for k in fetch if needle in k]
    if not full:
        continue
    k = max(full, key=lambda n: fetch[n])
    rd, wr = fetch[k] * 1024 * 2, write.get(k, 0.0) * 1024
    kernels[short] = {"kernel": k.split("(")[0], "FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k, 0.0),
                      "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only), one bench step of config 2 "
                   "(1.2e9 k-mers, 1 launch of each kernel). FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE is doubled on gfx950 per "
                   "MI355X_MICROARCH.md (HBM section). Bytes per launch.",
           "command": "bash tools/profile_round.sh " + tag, "kernels": kernels},
          open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
