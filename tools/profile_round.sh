#!/bin/bash
# Re-create the judged profile files of a round on the GPU box (run through gpurun from the
# repo root):  bash tools/profile_round.sh r04 [a|b|all]   (a gpurun call is capped at 20 minutes: two calls, a then b)
# Writes under gpurun_out/<round>/; copy what should be kept into profiles/<round>/.
# rocprofv3 gets the program itself after `--` (python3 <script>), never a shell or env hop, and
# counter passes (--pmc) are separate runs with --kernel-trace only.
set -u
round=${1:-r06}
part=${2:-all}        # a: headline + counters + row dictionaries; b: pipelines, other configurations, small runs, build; c (round 6): paired-end rows, config 5; all
repo=$PWD
out=$repo/gpurun_out/$round
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
if [ "$part" = a ] || [ "$part" = all ]; then
# --- the headline line and its kernel trace -------------------------------------------------------
python3 $repo/bench.py > $out/bench_1m.json 2> $out/bench_1m.log
echo "[profile_round] step 1 done"
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt -o kt -- python3 $repo/bench.py --no-cpu-baseline > $out/bench_1m_under_rocprof.json 2> $out/kt.log
echo "[profile_round] step 2 done"
# --- HBM traffic of the same command: FETCH_SIZE and WRITE_SIZE in passes of their own -------------
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_fetch.log
echo "[profile_round] step 3 done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_write.log
echo "[profile_round] step 4 done"
python3 $repo/tools/pmc_summary.py $out/pmc_fetch/f_counter_collection.csv $out/pmc_write/w_counter_collection.csv > $out/pmc_traffic_1m.json
echo "[profile_round] step 5 done"
# where the waves' time goes (one SQ pass: 8 slots) + the effective clock (GRBM)
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out/pmc_sq -o sq -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_sq.log
echo "[profile_round] step 6 done"
python3 $repo/tools/sq_summary.py $out/pmc_sq/sq_counter_collection.csv > $out/pmc_sq_summary.txt 2>&1
echo "[profile_round] step 7 done"
# --- the same iteration over row dictionaries (opt-in lossless storage) -------------------
python3 $repo/bench.py --storage coded > $out/bench_1m_coded.json 2> $out/bench_1m_coded.log
echo "[profile_round] step 8 done"
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt_coded -o kt -- python3 $repo/bench.py --storage coded --no-cpu-baseline > $out/bench_1m_coded_under_rocprof.json 2> $out/kt_coded.log
echo "[profile_round] step 9 done"
# counter traffic of the records kernel, CALIBRATED: a bare reader of exactly the same records runs in the same FETCH_SIZE pass
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch_coded -o f -- python3 $repo/tools/pmc_calibrate_coded.py > $out/pmc_calibration_coded.json 2> $out/pmc_fetch_coded.log
echo "[profile_round] step 10 done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write_coded -o w -- python3 $repo/tools/pmc_calibrate_coded.py > /dev/null 2> $out/pmc_write_coded.log
echo "[profile_round] step 11 done"
python3 $repo/tools/pmc_summary.py $out/pmc_fetch_coded/f_counter_collection.csv $out/pmc_write_coded/w_counter_collection.csv 1000000 5408 coded $out/pmc_calibration_coded.json > $out/pmc_traffic_coded_1m.json
echo "[profile_round] step 12 done"
# ... and of the kernel a plan of this size runs with the quad dictionary beside the records (em_iter_quad_coded_kernel)
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch_quads -o f -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads > $out/pmc_calibration_quads.json 2> $out/pmc_fetch_quads.log
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write_quads -o w -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads > /dev/null 2> $out/pmc_write_quads.log
python3 $repo/tools/pmc_summary.py $out/pmc_fetch_quads/f_counter_collection.csv $out/pmc_write_quads/w_counter_collection.csv 1000000 5408 coded $out/pmc_calibration_quads.json > $out/pmc_traffic_quads_1m.json
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch_quads_records -o f -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads --records > $out/pmc_calibration_quads_records.json 2> $out/pmc_fetch_quads_records.log
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write_quads_records -o w -- python3 $repo/tools/pmc_calibrate_coded.py 1000000 --quads --records > /dev/null 2> $out/pmc_write_quads_records.log
python3 $repo/tools/pmc_summary.py $out/pmc_fetch_quads_records/f_counter_collection.csv $out/pmc_write_quads_records/w_counter_collection.csv 1000000 5408 coded $out/pmc_calibration_quads_records.json > $out/pmc_traffic_quads_records_1m.json
echo "[profile_round] step 12q done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out/pmc_sq_coded -o sq -- python3 $repo/bench.py --storage coded --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2> $out/pmc_sq_coded.log
echo "[profile_round] step 13 done"
python3 $repo/tools/sq_summary.py $out/pmc_sq_coded/sq_counter_collection.csv > $out/coded_pmc_sq_summary.txt 2>&1
echo "[profile_round] step 14 done"
python3 $repo/tools/time_coded.py > $out/coded_storage_1m.txt 2>&1
python3 $repo/tools/time_coded_parts.py 10000 125000 1000000 > $out/coded_parts.txt 2>&1
python3 $repo/tools/stream_ceiling.py 5624000000 --records > $out/records_read_ceiling.txt 2>&1
echo "[profile_round] step 15 done"
python3 $repo/bench.py --storage coded --restarts 10 --no-cpu-baseline > $out/bench_1m_coded_10restarts.json 2> /dev/null
echo "[profile_round] step 16 done"
python3 $repo/bench.py --storage coded --total-rows 125000 --force-dist --no-cpu-baseline > $out/bench_125k_coded_one_rank_rccl.json 2> /dev/null
echo "[profile_round] step 17 done"
python3 $repo/bench.py --total-rows 125000 --force-dist --no-cpu-baseline > $out/bench_125k_one_rank_rccl.json 2> /dev/null
echo "[profile_round] step 18 done"
python3 $repo/bench.py --rows 1250000 --no-cpu-baseline > $out/bench_1250k_per_gpu.json 2> /dev/null
echo "[profile_round] step 19 done"
python3 $repo/bench.py --records > $out/bench_1m_records.json 2> /dev/null
echo "[profile_round] step 20 done"
python3 $repo/bench.py --records --total-rows 125000 --force-dist --no-cpu-baseline > $out/bench_125k_records_one_rank_rccl.json 2> /dev/null
echo "[profile_round] step 21 done"
# --- round 6: three restarts per pass over the records (configs 3 and 5) -------------------------------
python3 $repo/tools/time_quads_batched.py 1000000 > $out/quads_batched_1m.txt 2>&1
python3 $repo/bench.py --storage coded --restarts 10 --steps 48 --no-cpu-baseline > $out/bench_1m_coded_10restarts_48steps.json 2> /dev/null
python3 $repo/tools/ab_quad_ranking.py 1000000 > $out/quads_step_1m.txt 2>&1
echo "[profile_round] step 21b done"
fi
if [ "$part" = b ] || [ "$part" = all ]; then
python3 $repo/tools/run_pipeline.py --reads 1000000 > $out/pipeline_1m_records.txt 2>&1
echo "[profile_round] step 22 done"
python3 $repo/tools/run_pipeline.py --reads 10000000 > $out/pipeline_10m_records_one_gpu.txt 2>&1
echo "[profile_round] step 23 done"
python3 $repo/bench.py --records --total-rows 10000000 > $out/bench_10m_records_one_gpu.json 2> $out/bench_10m_records_one_gpu.log
echo "[profile_round] step 24 done"
python3 $repo/tools/run_pipeline.py --reads 1000000 --dense --storage f64 > $out/pipeline_1m.txt 2>&1
echo "[profile_round] step 25 done"
python3 $repo/tools/run_pipeline.py --reads 1000000 --dense --storage coded > $out/pipeline_1m_coded.txt 2>&1
echo "[profile_round] step 26 done"
# --- other configurations of the same script --------------------------------------------------------
python3 $repo/bench.py --restarts 10 --no-cpu-baseline > $out/bench_1m_10restarts.json 2> /dev/null
echo "[profile_round] step 27 done"
python3 $repo/bench.py --total-rows 100000 --no-cpu-baseline > $out/bench_100k.json 2> /dev/null
echo "[profile_round] step 28 done"
python3 $repo/bench.py --mode restarts --restarts 10 --no-cpu-baseline > $out/bench_restarts10_run.json 2> /dev/null
echo "[profile_round] step 29 done"
python3 $repo/bench.py --mode restarts --restarts 16 --no-cpu-baseline > $out/bench_restarts16_run.json 2> /dev/null
echo "[profile_round] step 30 done"
# --- the one-launch loop (cache-resident matrices) and the matrix build kernels ------------------------
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt_small -o kt -- python3 $repo/tools/time_small_runs.py --rows 600,2400,10000 > $out/small_runs_under_rocprof.txt 2> $out/kt_small.log
echo "[profile_round] step 31 done"
python3 $repo/tools/time_small_runs.py > $out/small_runs.txt 2>&1
echo "[profile_round] step 32 done"
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt_build -o kt -- python3 $repo/tools/run_build_only.py 1000000 sparse records bytes lut lut+sort linearize > $out/build_under_rocprof.txt 2> $out/kt_build.log
echo "[profile_round] step 33 done"
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch_build -o f -- python3 $repo/tools/run_build_only.py 1000000 sparse bytes lut+sort > /dev/null 2> $out/pmc_fetch_build.log
echo "[profile_round] step 34 done"
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write_build -o w -- python3 $repo/tools/run_build_only.py 1000000 sparse bytes lut+sort > /dev/null 2> $out/pmc_write_build.log
echo "[profile_round] step 35 done"
rocprofv3 --output-format csv --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $out/pmc_l2_build -o l2 -- python3 $repo/tools/run_build_only.py 1000000 bytes lut+sort > /dev/null 2> $out/pmc_l2_build.log
echo "[profile_round] step 36 done"
python3 $repo/tools/pmc_summary.py $out/pmc_fetch_build/f_counter_collection.csv $out/pmc_write_build/w_counter_collection.csv > $out/pmc_traffic_build_1m.json
echo "[profile_round] step 37 done"
python3 $repo/tools/run_build_only.py > $out/build_kernels.txt 2>&1
echo "[profile_round] step 38 done"
python3 $repo/tools/time_build_variants.py 1000000 > $out/build_kernel_alone.txt 2>&1
echo "[profile_round] step 39 done"
python3 $repo/tools/time_restarts.py > $out/restart_schedules.txt 2>&1
echo "[profile_round] step 40 done"
python3 $repo/tools/time_dropin_build.py > $out/dropin_build.txt 2>&1
echo "[profile_round] step 41 done"
# --- round 5: the front end (host threads of the box), the pipeline from alignments, the row-pass experiments ----------
python3 $repo/tools/time_frontend.py > $out/frontend_1m.txt 2>&1
echo "[profile_round] step 42 done"
python3 $repo/tools/run_pipeline.py --reads 1000000 --alignments > $out/pipeline_1m_alignments.txt 2>&1
echo "[profile_round] step 43 done"
# (the experiment's own library: built here when the snapshot did not bring it)
[ -f $repo/tools/experiments/_build/libcoded3.so ] || { mkdir -p $repo/tools/experiments/_build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I $repo/include -I $repo/mixemt_amd/csrc $repo/tools/experiments/coded3_experiment.hip -o $repo/tools/experiments/_build/libcoded3.so; }
python3 $repo/tools/experiments/time_coded3.py 1000000 > $out/row_pass_experiments.txt 2>&1
echo "[profile_round] step 44 done"
fi
if [ "$part" = c ] || [ "$part" = all ]; then
# --- round 6: paired-end fragments (synth-pe-v1) and 250-bp reads through the default route; config 5 on one rank ------
MXM_PIPELINE_TIMING=1 python3 $repo/tools/run_pipeline.py --reads 1000000 --pairs > $out/pipeline_1m_pe.txt 2>&1
echo "[profile_round] step 45 done"
MXM_PIPELINE_TIMING=1 python3 $repo/tools/run_pipeline.py --reads 1000000 --read-len 250 > $out/pipeline_1m_250bp.txt 2>&1
echo "[profile_round] step 46 done"
rocprofv3 --output-format csv --kernel-trace --stats -d $out/kt_pe -o kt -- python3 $repo/tools/run_pipeline.py --reads 1000000 --pairs > $out/pipeline_1m_pe_under_rocprof.txt 2> $out/kt_pe.log
echo "[profile_round] step 47 done"
python3 $repo/tools/run_build_only.py --pairs 1000000 sparse records > $out/build_kernels_pe.txt 2>&1
python3 $repo/tools/run_build_only.py --pairs --no-long 1000000 sparse records >> $out/build_kernels_pe.txt 2>&1
python3 $repo/tools/run_build_only.py --read-len 250 1000000 sparse records > $out/build_kernels_250bp.txt 2>&1
python3 $repo/tools/time_build_variants.py --pairs 1000000 > $out/build_kernel_alone_pe.txt 2>&1
echo "[profile_round] step 48 done"
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out/pmc_sq_build_pe -o sq -- python3 $repo/tools/time_build_variants.py --pairs 400000 > /dev/null 2> $out/pmc_sq_build_pe.log
python3 $repo/tools/sq_summary.py $out/pmc_sq_build_pe/sq_counter_collection.csv > $out/build_pe_pmc_sq_summary.txt 2>&1
echo "[profile_round] step 49 done"
python3 $repo/tools/time_config5_one_rank.py > $out/config5_one_rank.txt 2>&1
echo "[profile_round] step 50 done"
python3 $repo/tools/stress_parity.py --seed 606 --budget 420 > $out/stress_parity_606.txt 2>&1
echo "[profile_round] step 51 done"
# --- the quad dictionary's two encoders (a workgroup / a wave per row), their counters, and the loop's idle gaps --------------
python3 $repo/tools/ab_quad_encoder.py > $out/ab_quad_encoder_1m.txt 2>&1
python3 $repo/tools/ab_quad_encoder.py --pairs > $out/ab_quad_encoder_pe.txt 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $out/pmc_sq_qenc -o sq -- python3 $repo/tools/ab_quad_encoder.py > /dev/null 2> $out/pmc_sq_qenc.log
python3 $repo/tools/sq_summary.py $out/pmc_sq_qenc/sq_counter_collection.csv > $out/quad_encoder_pmc_sq_summary.txt 2>&1
rocprofv3 --output-format csv --kernel-trace -d $out/kt_loop -o kt -- python3 $repo/tools/run_pipeline.py --reads 1000000 > $out/pipeline_1m_records_under_rocprof.txt 2> $out/kt_loop.log
python3 $repo/tools/loop_gaps.py $(find $out/kt_loop -name "kt_kernel_trace.csv" | head -1) > $out/loop_gaps_1m.txt 2>&1
echo "[profile_round] step 52 done"
fi
find $out -name "*.csv" | head -60
