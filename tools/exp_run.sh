set -o pipefail
mkdir -p gpurun_out/r03_b
timeout -k 10 300 python -m pytest tests/test_gpu_index.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r03_b/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_b/rc.txt
tail -3 gpurun_out/r03_b/pytest.log
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r03_b/bench_new.log 2>&1; echo "bench_new rc=$?" >> gpurun_out/r03_b/rc.txt
cat gpurun_out/r03_b/rc.txt
grep '^{' gpurun_out/r03_b/bench_new.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('NEW', d['ms_per_step'], d['roofline']['kernels_ms_per_step'])"
bash tools/pmc_kernels.sh r03_b_pmc 2>&1 | tail -16
