set -o pipefail
mkdir -p gpurun_out/r03_g
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_g/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r03_g/rc.txt
tail -5 gpurun_out/r03_g/pytest.log
cat gpurun_out/r03_g/rc.txt
