set -x
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest12.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r02/pytest12.log
tail -8 gpurun_out/r02/pytest12.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/r02/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r02/smoke.log
timeout -k 10 400 python bench.py --rows 1250000 --no-cpu-baseline > gpurun_out/r02/bench_1250k_per_gpu.json 2> gpurun_out/r02/bench_1250k_per_gpu.log; echo "rc=$?"
tail -3 gpurun_out/r02/bench_1250k_per_gpu.log; cat gpurun_out/r02/bench_1250k_per_gpu.json
timeout -k 10 300 python bench.py --total-rows 125000 --force-dist --no-cpu-baseline > gpurun_out/r02/bench_125k_one_rank_rccl.json 2> gpurun_out/r02/bench_125k_one_rank_rccl.log; echo "rc=$?"
tail -2 gpurun_out/r02/bench_125k_one_rank_rccl.log; cat gpurun_out/r02/bench_125k_one_rank_rccl.json
