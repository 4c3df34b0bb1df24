set -x
mkdir -p gpurun_out/r02
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02/pytest1.log
tail -5 gpurun_out/r02/pytest1.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench_default.json 2> gpurun_out/r02/bench_default.log; echo "rc=$?"
tail -3 gpurun_out/r02/bench_default.log
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --total-rows 200000 --no-cpu-baseline --steps 10 > gpurun_out/r02/bench_spawn2_gloo.json 2> gpurun_out/r02/bench_spawn2_gloo.log; echo "rc=$?"
tail -3 gpurun_out/r02/bench_spawn2_gloo.log
timeout -k 10 300 python bench.py --mode restarts --restarts 10 --no-cpu-baseline > gpurun_out/r02/bench_restarts10.json 2> gpurun_out/r02/bench_restarts10.log; echo "rc=$?"
tail -3 gpurun_out/r02/bench_restarts10.log
timeout -k 10 300 python tools/time_restarts.py > gpurun_out/r02/restart_schedules.txt 2>&1; echo "rc=$?"
cat gpurun_out/r02/restart_schedules.txt
