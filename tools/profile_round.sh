#!/bin/bash
# Re-create the judged profile files of a round on the GPU box (run through gpurun from the
# repo root):  bash tools/profile_round.sh r01
# Writes under gpurun_out/<round>/; copy what should be kept into profiles/<round>/.
set -u
round=${1:-r01}
repo=$PWD
out=$repo/gpurun_out/$round
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
python3 $repo/bench.py > $out/bench_1m.json 2> $out/bench_1m.log
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt -o kt -- python3 $repo/bench.py --no-cpu-baseline > $out/bench_1m_under_rocprof.json 2> $out/kt.log
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_write.log
# where the waves' time goes (one SQ pass: 8 slots) + the effective clock (GRBM)
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out/pmc_sq -o sq -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_sq.log
python3 $repo/bench.py --restarts 10 --no-cpu-baseline > $out/bench_1m_10restarts.json 2> /dev/null
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt10 -o kt -- python3 $repo/bench.py --restarts 10 --no-cpu-baseline > /dev/null 2> $out/kt10.log
python3 $repo/bench.py --storage f32 --no-cpu-baseline > $out/bench_1m_f32_storage_variant.json 2> /dev/null
python3 $repo/bench.py --rows 100000 --no-cpu-baseline > $out/bench_100k.json 2> /dev/null
python3 $repo/tools/pmc_summary.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv > $out/pmc_traffic_1m.json
find $out -name "*.csv" | head -40
