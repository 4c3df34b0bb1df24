set -x
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest3.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r02/pytest3.log
tail -15 gpurun_out/r02/pytest3.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python tools/time_small_runs.py --rows 600,2400,10000,30000 > gpurun_out/r02/small_runs2.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/small_runs2.txt
timeout -k 10 300 python tools/time_restarts.py > gpurun_out/r02/restart_schedules2.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/restart_schedules2.txt
