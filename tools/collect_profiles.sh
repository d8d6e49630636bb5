#!/bin/bash
# Copies the judged summaries of a round from gpurun_out/$1 (scratch, merged back from the GPU box) into profiles/$1 (tracked).
R=${1:-r06}; S=gpurun_out/$R; D=profiles/$R
mkdir -p $D
for f in bench.json bench_c5.json bench_c2.json kernel_stats.csv kernel_stats_C4.csv kernel_stats_C5.csv f3_pnp_kernel_stats.csv lm_c4_priors_kernel_stats.csv \
         pmc_summary.json pmc_mfma_summary.json pmc_valu_summary.json pmc_valu_C4.csv pmc_valu_C5.csv pmc_f64_C4.csv pmc_f64_C5.csv pmc_lds_C4.csv pmc_lds_C5.csv pmc_l2_C4.csv pmc_l2_C5.csv pmc_mfma_C4.csv pmc_mfma_C5.csv \
         phase_times.txt lm_time_c4.txt lm_time_c5.txt lm_time_c4_free_ratio.txt lm_time_c4_priors.txt chol_trace_c4.txt schur_trace_c4.txt setup_time.txt \
         factored_groups_ab.txt filter_time.txt hbm_calib.txt mfma_f64_rate.txt tile_factor.txt xcd_handoff.txt pytest_gpu.txt dag_stress_summary.txt \
         bench_hook_c4_n2.json bench_hook_c4_n4.json bench_hook_c4_n8.json bench_hook_c5_n8.json bench_native_mock_c4_n2.json bench_native_mock_c4_n8.json sharded_per_rank_device_time.txt lm_time_per_frame_intrinsics.txt chol_poll_split_final.txt; do
  [ -f $S/$f ] && cp $S/$f $D/$f
done
for c in C4 C5; do
  [ -f $S/pmc_fetch_$c/f_counter_collection.csv ] && cp $S/pmc_fetch_$c/f_counter_collection.csv $D/pmc_fetch_${c}_counter_collection.csv
  [ -f $S/pmc_write_$c/w_counter_collection.csv ] && cp $S/pmc_write_$c/w_counter_collection.csv $D/pmc_write_${c}_counter_collection.csv
done
ls $D | wc -l
