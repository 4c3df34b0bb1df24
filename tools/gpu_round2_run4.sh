set -x
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_random.py tests/test_gpu_fused.py -x -q > gpurun_out/r02/pytest4.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r02/pytest4.log
tail -15 gpurun_out/r02/pytest4.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/run_build_only.py > gpurun_out/r02/build_kernels.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/build_kernels.txt
timeout -k 10 300 python tools/time_small_runs.py --rows 600,2400,10000 > gpurun_out/r02/small_runs3.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/small_runs3.txt
