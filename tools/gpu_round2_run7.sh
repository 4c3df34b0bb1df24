set -x
mkdir -p gpurun_out/r02
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest7.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r02/pytest7.log
tail -12 gpurun_out/r02/pytest7.log
if [ $rc -ne 0 ]; then exit $rc; fi
echo "== counter tree (default)" > gpurun_out/r02/barrier_variants.txt
timeout -k 10 200 python tools/time_small_runs.py --rows 600,2400 >> gpurun_out/r02/barrier_variants.txt 2>&1
echo "== one flat counter (-D FUSED_BARRIER=1)" >> gpurun_out/r02/barrier_variants.txt
MXM_LIB=$PWD/mixemt_amd/lib/tune/barrier_flat.so timeout -k 10 200 python tools/time_small_runs.py --rows 600,2400 >> gpurun_out/r02/barrier_variants.txt 2>&1
grep -v "amdgpu.ids\|kernels" gpurun_out/r02/barrier_variants.txt
timeout -k 10 300 python tools/run_pipeline.py --reads 1000000 > gpurun_out/r02/pipeline_1m.txt 2>&1; echo "rc=$?"
tail -30 gpurun_out/r02/pipeline_1m.txt
timeout -k 10 200 python bench.py --restarts 10 --no-cpu-baseline > gpurun_out/r02/bench_1m_10restarts_loop.json 2> gpurun_out/r02/bench_1m_10restarts_loop.log; echo "rc=$?"
tail -2 gpurun_out/r02/bench_1m_10restarts_loop.log; cat gpurun_out/r02/bench_1m_10restarts_loop.json
