set -x
mkdir -p gpurun_out/r02
timeout -k 10 300 python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/r02/pytest_fused.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r02/pytest_fused.log
tail -30 gpurun_out/r02/pytest_fused.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/time_small_runs.py > gpurun_out/r02/small_runs.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/small_runs.txt
