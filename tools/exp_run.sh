set -o pipefail
bash tools/profile_round.sh r03_m > gpurun_out/r03_m_profile.log 2>&1; echo "profile rc=$?"
tail -3 gpurun_out/r03_m_profile.log
cat gpurun_out/r03_m/bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernels_ms_per_step']); print(d['extra']['cold'], d['extra']['host_resident_note']); print(d['cpu_baseline'])"
