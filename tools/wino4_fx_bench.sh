#!/bin/bash
# the C4 shard step with and without the fix-up launches (AGZ_WINO4_FX=4: results WRONG, the load is the same shape): its share of
# a step as the product runs it (two chains, power limit).  Needs gpurun_ab/libagz_FX.so (tools/build_fx_lib.sh).
cp alphago.jl_amd/libagz.so /tmp/keep.so
cp gpurun_ab/libagz_FX.so alphago.jl_amd/libagz.so
for x in 0 4 0 4; do
  echo -n "FX=$x "
  AGZ_WINO4_FX=$x python bench.py --board 19 --tower 20 --readouts 800 --games 256 --steps 40 --warmup 5 --no-cpu-baseline --no-alt-precision 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['power']['sclk_mhz']['mean'])"
done
cp /tmp/keep.so alphago.jl_amd/libagz.so
