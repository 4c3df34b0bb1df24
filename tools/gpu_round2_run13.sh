set -x
mkdir -p gpurun_out/r02
timeout -k 10 400 python -m pytest tests/test_gpu_fused.py tests/test_gpu_em.py -x -q > gpurun_out/r02/pytest13.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r02/pytest13.log
tail -6 gpurun_out/r02/pytest13.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/time_small_runs.py --rows 300,600,768,769,2400 > gpurun_out/r02/small_runs5.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r02/small_runs5.txt
MXM_LIB=$PWD/mixemt_amd/lib/tune/fused_stamps.so timeout -k 10 200 python tools/time_small_runs.py --rows 600 --stamps > gpurun_out/r02/fused_stamps_resident.txt 2>&1; echo "rc=$?"
grep -v "amdgpu.ids\|kernels" gpurun_out/r02/fused_stamps_resident.txt
