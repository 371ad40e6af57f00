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
git status --short profiles | head -40
