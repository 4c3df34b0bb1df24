#!/bin/bash
# What the driver runs at round end, in one gpurun call from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/gpu_regress.sh'
# GPU parity suite (incl. the 10^6-row tests), smoke(), the default bench line.  Each step under its own
# timeout; a failed step stops the script (no further GPU work after a failure).
set -x
mkdir -p gpurun_out/regress
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/regress/pytest_gpu.log 2>&1; rc=$?
tail -8 gpurun_out/regress/pytest_gpu.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/regress/smoke.log 2>&1; rc=$?
tail -2 gpurun_out/regress/smoke.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/regress/bench.json 2> gpurun_out/regress/bench.log; rc=$?
tail -4 gpurun_out/regress/bench.log; cat gpurun_out/regress/bench.json
exit $rc
