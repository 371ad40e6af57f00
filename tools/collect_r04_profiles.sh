#!/bin/bash
# copies what tools/profile_r04.sh left under gpurun_out/r04prof_<tag>/ into profiles/ under the tracked round-4 names
set -u
for t in ${TAGS:-f32 c4 c5 f16 f32s c4f33}; do
  O=gpurun_out/r04prof_$t
  for f in kernel_stats.csv bench_under_rocprof.json pmc_mfma_lds.csv pmc_traffic.json; do
    [ -s $O/$f ] && cp $O/$f profiles/r04_${t}_$f
  done
done
[ -s gpurun_out/r04prof_f32/pmc_traffic.json ] && cp gpurun_out/r04prof_f32/pmc_traffic.json profiles/r04_pmc_traffic_9x9_f32.json
[ -s gpurun_out/r04prof_c4/pmc_traffic.json ] && cp gpurun_out/r04prof_c4/pmc_traffic.json profiles/r04_pmc_traffic_19x19_f32.json
[ -s gpurun_out/r04prof_c5/pmc_traffic.json ] && cp gpurun_out/r04prof_c5/pmc_traffic.json profiles/r04_pmc_traffic_19x19_f16.json
[ -s gpurun_out/r04prof_f16/pmc_traffic.json ] && cp gpurun_out/r04prof_f16/pmc_traffic.json profiles/r04_pmc_traffic_9x9_f16.json
for f in bench_f32 c4_bench c4_bench_f33 c5_bench bench_f16 bench_2rank_single_device bench_8rank_single_device generation; do
  [ -s gpurun_out/r04lines/$f.json ] && tail -1 gpurun_out/r04lines/$f.json > profiles/r04_$f.json
done
[ -s gpurun_out/r04lines/gpu_tests.log ] && cp gpurun_out/r04lines/gpu_tests.log profiles/r04_gpu_tests.log
git status --short profiles | head -40
