#!/bin/bash
# copies what tools/profile_r03_all.sh left under gpurun_out/ into profiles/ under the tracked round-3 names
set -u
for t in f32 c4 c5 f16 f32s; do
  O=gpurun_out/r03prof_$t
  for f in kernel_stats.csv bench_under_rocprof.json pmc_mfma_lds.csv pmc_traffic.json; do
    [ -s $O/$f ] && cp $O/$f profiles/r03_${t}_$f
  done
done
cp gpurun_out/r03prof_f32/pmc_traffic.json profiles/r03_pmc_traffic_9x9_f32.json
cp gpurun_out/r03prof_f16/pmc_traffic.json profiles/r03_pmc_traffic_9x9_f16.json
cp gpurun_out/r03prof_f32s/pmc_traffic.json profiles/r03_pmc_traffic_9x9_f32s.json
cp gpurun_out/r03prof_c4/pmc_traffic.json profiles/r03_pmc_traffic_19x19_f32.json
cp gpurun_out/r03prof_c5/pmc_traffic.json profiles/r03_pmc_traffic_19x19_f16.json
for f in bench_f32 c4_bench c5_bench bench_f16 bench_2rank_single_device bench_generation; do
  [ -s gpurun_out/r03lines/$f.json ] && tail -1 gpurun_out/r03lines/$f.json > profiles/r03_$f.json
done
[ -s gpurun_out/r03e/gpu_tests.log ] && cp gpurun_out/r03e/gpu_tests.log profiles/r03_gpu_tests.log
git status --short profiles | head -40
