set -x
mkdir -p gpurun_out/r02
MXM_LIB=$PWD/mixemt_amd/lib/tune/fused_stamps.so timeout -k 10 200 python tools/time_small_runs.py --rows 600,2400,10000 --stamps > gpurun_out/r02/fused_stamps.txt 2>&1; echo "rc=$?"
grep -v "amdgpu.ids\|kernels" gpurun_out/r02/fused_stamps.txt
timeout -k 10 200 python bench.py --restarts 10 --no-cpu-baseline > gpurun_out/r02/bench_1m_10restarts_loop.json 2> gpurun_out/r02/bench_1m_10restarts_loop.log; echo "rc=$?"
tail -2 gpurun_out/r02/bench_1m_10restarts_loop.log; cat gpurun_out/r02/bench_1m_10restarts_loop.json
