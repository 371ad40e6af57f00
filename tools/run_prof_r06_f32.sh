cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
STEPS=20 bash tools/profile_r06.sh f32 1024 9 > gpurun_out/r06prof_f32.log 2>&1
STEPS=20 bash tools/profile_r06.sh f32w5 1024 9 --winograd 3 > gpurun_out/r06prof_f32w5.log 2>&1
for w in 1 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06prof_1chain_w$w -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt-precision --no-config-legs --generation 0 --no-live-traffic --no-sustained --tower-streams 1 --winograd $w > gpurun_out/r06prof_1chain_w$w.json 2>/dev/null
  find gpurun_out/r06prof_1chain_w$w -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} gpurun_out/r06_1chain_w${w}_kernel_stats.csv
  rm -rf gpurun_out/r06prof_1chain_w$w
done
head -8 gpurun_out/r06_1chain_w1_kernel_stats.csv; head -8 gpurun_out/r06_1chain_w3_kernel_stats.csv
cat gpurun_out/r06prof_f32/pmc_mfma_lds.csv; cat gpurun_out/r06prof_f32w5/pmc_mfma_lds.csv
