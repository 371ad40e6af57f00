#!/bin/bash
# same-box A/B of two builds of libagz.so on a bench.py command: tools/ab_bench.sh <libA> <libB> [bench args...]  (alternating, 3 rounds)
A=$1; B=$2; shift 2
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
for rep in 1 2 3; do
  for lib in $A $B; do
    cp $lib alphago.jl_amd/libagz.so
    echo -n "$(basename $lib) "
    python bench.py --no-cpu-baseline --no-alt-precision "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],2), round(d['ms_per_step'],2), round(d['power']['sclk_mhz']['mean']), round(d['power']['socket_power_w']['mean']))"
  done
done
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
