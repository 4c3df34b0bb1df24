#!/bin/bash
# The quad kernel's counter passes alone (steps "12q" of tools/profile_round.sh): bash tools/profile_quads_pmc.sh r05
set -e
round=${1:-r05}
repo=$(cd "$(dirname "$0")/.." && pwd)
out=$repo/gpurun_out/$round
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch_quads -o f -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads > $out/pmc_calibration_quads.json 2> $out/pmc_fetch_quads.log
echo "[profile_quads_pmc] fetch pass done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write_quads -o w -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads > /dev/null 2> $out/pmc_write_quads.log
echo "[profile_quads_pmc] write pass done"
python3 $repo/tools/pmc_summary.py $out/pmc_fetch_quads/f_counter_collection.csv $out/pmc_write_quads/w_counter_collection.csv 1000000 5408 coded $out/pmc_calibration_quads.json > $out/pmc_traffic_quads_1m.json
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch_quads_records -o f -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads --records > $out/pmc_calibration_quads_records.json 2> $out/pmc_fetch_quads_records.log
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write_quads_records -o w -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads --records > /dev/null 2> $out/pmc_write_quads_records.log
python3 $repo/tools/pmc_summary.py $out/pmc_fetch_quads_records/f_counter_collection.csv $out/pmc_write_quads_records/w_counter_collection.csv 1000000 5408 coded $out/pmc_calibration_quads_records.json > $out/pmc_traffic_quads_records_1m.json
cat $out/pmc_calibration_quads.json $out/pmc_calibration_quads_records.json
grep -A9 "em_iter_quad_coded_kernel" $out/pmc_traffic_quads_1m.json $out/pmc_traffic_quads_records_1m.json
