#!/bin/bash
# copies what tools/profile_r05.sh left under gpurun_out/r05prof_<tag>/ into profiles/ under the tracked round-5 names
set -u
for t in ${TAGS:-f32 c4 c5}; do
  O=gpurun_out/r05prof_$t
  for f in kernel_stats.csv bench_under_rocprof.json pmc_mfma_lds.csv pmc_traffic.json; do
    [ -s $O/$f ] && cp $O/$f profiles/r05_${t}_$f
  done
done
[ -s gpurun_out/r05prof_f32/pmc_traffic.json ] && cp gpurun_out/r05prof_f32/pmc_traffic.json profiles/r05_pmc_traffic_9x9_f32.json
[ -s gpurun_out/r05prof_c4/pmc_traffic.json ] && cp gpurun_out/r05prof_c4/pmc_traffic.json profiles/r05_pmc_traffic_19x19_f32.json
[ -s gpurun_out/r05prof_c5/pmc_traffic.json ] && cp gpurun_out/r05prof_c5/pmc_traffic.json profiles/r05_pmc_traffic_19x19_f16.json
for f in bench_default bench_2rank_selflaunched bench_8rank_selflaunched; do
  [ -s gpurun_out/r05lines/$f.json ] && tail -1 gpurun_out/r05lines/$f.json > profiles/r05_$f.json
done
[ -s gpurun_out/r05lines/gpu_tests.log ] && cp gpurun_out/r05lines/gpu_tests.log profiles/r05_gpu_tests.log
git status --short profiles | head -40
