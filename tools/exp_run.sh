set -o pipefail
mkdir -p gpurun_out/r03_j
timeout -k 10 600 python -m pytest tests/test_gpu_facade.py tests/test_gpu_comm.py tests/test_gpu_kmer_ops.py tests/test_abi_and_host.py -x -q > gpurun_out/r03_j/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_j/rc.txt
tail -25 gpurun_out/r03_j/pytest.log
cat gpurun_out/r03_j/rc.txt
