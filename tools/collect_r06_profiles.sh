#!/bin/bash
# copies what tools/profile_r06.sh left under gpurun_out/r06prof_<tag>/ into profiles/ under the tracked round-6 names
set -u
for t in ${TAGS:-f32 c4 c5}; do
  O=gpurun_out/r06prof_$t
  for f in kernel_stats.csv bench_under_rocprof.json pmc_mfma_lds.csv pmc_traffic.json; do
    [ -s $O/$f ] && cp $O/$f profiles/r06_${t}_$f
  done
done
[ -s gpurun_out/r06prof_f32/pmc_traffic.json ] && cp gpurun_out/r06prof_f32/pmc_traffic.json profiles/r06_pmc_traffic_9x9_f32.json
[ -s gpurun_out/r06prof_c4/pmc_traffic.json ] && cp gpurun_out/r06prof_c4/pmc_traffic.json profiles/r06_pmc_traffic_19x19_f32.json
[ -s gpurun_out/r06prof_c5/pmc_traffic.json ] && cp gpurun_out/r06prof_c5/pmc_traffic.json profiles/r06_pmc_traffic_19x19_f16.json
for f in bench_default bench_2rank_selflaunched bench_8rank_selflaunched; do
  [ -s gpurun_out/r06lines/$f.json ] && tail -1 gpurun_out/r06lines/$f.json > profiles/r06_$f.json
done
[ -s gpurun_out/r06lines/gpu_tests.log ] && cp gpurun_out/r06lines/gpu_tests.log profiles/r06_gpu_tests.log
git status --short profiles | head -40
