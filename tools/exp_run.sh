set -o pipefail
mkdir -p gpurun_out/r03_i
timeout -k 10 600 python -m pytest tests/test_gpu_comm.py tests/test_gpu_bench_contract.py tests/test_gpu_dist.py tests/test_gpu_kmer_ops.py tests/test_gpu_position_index.py -m gpu -x -q > gpurun_out/r03_i/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_i/rc.txt
tail -15 gpurun_out/r03_i/pytest.log
for t in kmi torch; do
timeout -k 10 200 python bench.py --force-dist --dist-mode superkmer --transport $t --steps 8 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/r03_i/bench_fd_$t.log 2>&1; echo "bench $t rc=$?" >> gpurun_out/r03_i/rc.txt
grep '^{' gpurun_out/r03_i/bench_fd_$t.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['config'].get('distinct_kmers'), d['roofline']['kernels_ms_per_step'])"
done
cat gpurun_out/r03_i/rc.txt
