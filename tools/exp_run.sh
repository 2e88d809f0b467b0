set -o pipefail
mkdir -p gpurun_out/r03_k
timeout -k 10 600 python -m pytest tests/test_gpu_fasta.py tests/test_gpu_index.py tests/test_abi_and_host.py tests/test_gpu_facade.py -x -q > gpurun_out/r03_k/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_k/rc.txt
tail -25 gpurun_out/r03_k/pytest.log
cat gpurun_out/r03_k/rc.txt
