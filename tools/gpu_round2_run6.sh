set -x
mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_build.py tests/test_gpu_random.py -x -q -k "build or lut or kernel or random_tables" > gpurun_out/r02/pytest6.log 2>&1; rc=$?; echo "pytest rc=$rc" >> gpurun_out/r02/pytest6.log
tail -8 gpurun_out/r02/pytest6.log
if [ $rc -ne 0 ]; then exit $rc; fi
echo "== A default (CPL 4, unroll 8, pipelined)" > gpurun_out/r02/lut_variants.txt
timeout -k 10 200 python tools/run_build_only.py 1000000 lut lut+sort lut+sort+P >> gpurun_out/r02/lut_variants.txt 2>&1
for v in B C D E F; do
  echo "== $v" >> gpurun_out/r02/lut_variants.txt
  MXM_LIB=$PWD/mixemt_amd/lib/tune/lut_$v.so timeout -k 10 200 python tools/run_build_only.py 1000000 lut lut+sort lut+sort+P >> gpurun_out/r02/lut_variants.txt 2>&1
done
grep -v "amdgpu.ids\|^one MI355X" gpurun_out/r02/lut_variants.txt
