#!/bin/bash
# Copy the summaries of a tools/profile_round.sh run from gpurun_out/<round>/ (scratch) into
# profiles/<round>/ (tracked).  usage: bash tools/collect_profiles.sh r02
set -u
round=${1:-r06}
src=gpurun_out/$round
dst=profiles/$round
mkdir -p $dst
for f in bench_1m.json bench_1m_under_rocprof.json bench_1m_10restarts.json bench_100k.json \
         bench_restarts10_run.json bench_restarts16_run.json pmc_traffic_1m.json pmc_traffic_build_1m.json \
         pmc_sq_summary.txt small_runs.txt small_runs_under_rocprof.txt build_kernels.txt build_under_rocprof.txt \
         restart_schedules.txt dropin_build.txt build_kernel_alone.txt \
         bench_1m_coded.json bench_1m_coded_under_rocprof.json bench_1m_coded_10restarts.json \
         bench_125k_coded_one_rank_rccl.json bench_125k_one_rank_rccl.json bench_1250k_per_gpu.json \
         pmc_traffic_coded_1m.json coded_pmc_sq_summary.txt coded_storage_1m.txt pipeline_1m.txt pipeline_1m_coded.txt \
         bench_1m_records.json bench_10m_records_one_gpu.json bench_10m_records_one_gpu.log \
         bench_125k_records_one_rank_rccl.json pipeline_1m_records.txt pipeline_10m_records_one_gpu.txt \
         pmc_calibration_coded.json pmc_calibration_quads.json pmc_traffic_quads_1m.json pmc_calibration_quads_records.json pmc_traffic_quads_records_1m.json quad_bare_reader_1m.txt quad_batched_1m.txt quad_mfma_sum_1m.txt coded_parts.txt records_read_ceiling.txt \
         frontend_1m.txt pipeline_1m_alignments.txt row_pass_experiments.txt \
         bam_reader_1m.txt pipeline_1m_bam.txt quad_1m.txt quads_product_1m.txt quad_build_1m.txt step_sequence.txt \
         alloc_big.txt records_build_alignments.txt stress_parity_707.txt stress_parity_808.txt stress_parity_909.txt stress_parity_1111.txt \
         bench_125k_records_one_rank_rccl.json bench_125k_records_one_rank_oneshot.json exchange_tests.txt \
         quads_batched_1m.txt bench_1m_coded_10restarts_48steps.json quads_step_1m.txt pipeline_1m_pe.txt pipeline_1m_250bp.txt \
         pipeline_1m_pe_under_rocprof.txt build_kernels_pe.txt build_kernels_250bp.txt build_kernel_alone_pe.txt \
         build_pe_pmc_sq_summary.txt config5_one_rank.txt stress_parity_606.txt \
         ab_quad_encoder_1m.txt ab_quad_encoder_pe.txt quad_encoder_pmc_sq_summary.txt loop_gaps_1m.txt pipeline_1m_records_under_rocprof.txt; do
  [ -f $src/$f ] && cp $src/$f $dst/$f
done
[ -f $src/bench_1m.log ] && cp $src/bench_1m.log $dst/bench_1m.log
cp $src/kt/kt_kernel_stats.csv $dst/bench_1m_kernel_stats.csv 2>/dev/null
cp $src/kt_small/kt_kernel_stats.csv $dst/small_runs_kernel_stats.csv 2>/dev/null
cp $src/kt_coded/kt_kernel_stats.csv $dst/coded_kernel_stats.csv 2>/dev/null
cp $src/pmc_fetch_coded/f_counter_collection.csv $dst/pmc_fetch_size_coded.csv 2>/dev/null
cp $src/pmc_fetch_quads/f_counter_collection.csv $dst/pmc_fetch_size_quads.csv 2>/dev/null
cp $src/pmc_sq_coded/sq_counter_collection.csv $dst/pmc_sq_coded.csv 2>/dev/null
[ -f $src/bench_1m_coded.log ] && cp $src/bench_1m_coded.log $dst/bench_1m_coded.log
cp $src/kt_build/kt_kernel_stats.csv $dst/build_kernel_stats.csv 2>/dev/null
cp $src/kt_pe/kt_kernel_stats.csv $dst/pipeline_1m_pe_kernel_stats.csv 2>/dev/null
cp $src/pmc_fetch/f_counter_collection.csv $dst/pmc_fetch_size.csv 2>/dev/null
cp $src/pmc_write/w_counter_collection.csv $dst/pmc_write_size.csv 2>/dev/null
cp $src/pmc_sq/sq_counter_collection.csv $dst/pmc_sq.csv 2>/dev/null
cp $src/pmc_fetch_build/f_counter_collection.csv $dst/pmc_fetch_size_build.csv 2>/dev/null
cp $src/pmc_write_build/w_counter_collection.csv $dst/pmc_write_size_build.csv 2>/dev/null
cp $src/pmc_l2_build/l2_counter_collection.csv $dst/pmc_l2_build.csv 2>/dev/null
ls -la $dst
