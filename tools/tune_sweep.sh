#!/bin/bash
# Kernel-shape sweep on the GPU box: every mixemt_amd/lib/tune/*.so (same ABI, different
# -DMXM_V* shapes) through bench.py.  usage: tools/tune_sweep.sh "A D I" [restarts] [rows]
libs="$1"; restarts="${2:-1}"; rows="${3:-1000000}"
mkdir -p gpurun_out/tune
for v in $libs; do
  MXM_LIB=$PWD/mixemt_amd/lib/tune/$v.so timeout 600 python bench.py --rows $rows --restarts $restarts --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/tune/${v}_B${restarts}.json 2> gpurun_out/tune/${v}_B${restarts}.log
  python - "$v" "$restarts" <<'PY'
import json, sys
v, b = sys.argv[1], sys.argv[2]
try:
    d = json.load(open("gpurun_out/tune/%s_B%s.json" % (v, b)))
    print("%-3s B=%s  ms/step %.3f  kernel_ms %.3f  frac %.3f  restart-iters/s %.1f  ok=%s" % (v, b, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["em_iters_per_s"], d["sanity_ok"]))
except Exception as e:
    print(v, b, "FAILED", e)
PY
done
