#!/bin/bash
# same-box alternating A/B of the one-pass (k_wino_gemm4, --winograd 1) and the five-pass 64 x 128 (k_wino5_gemm, --winograd 3)
# F(3x3,3x3) tower on the headline workload: tools/ab_wino5.sh [rounds] [extra bench args...]
R=${1:-3}; shift
for rep in $(seq 1 $R); do
  for w in 1 3; do
    echo -n "winograd=$w "
    python bench.py --winograd $w --steps 40 --warmup 5 --no-cpu-baseline --no-alt-precision --no-config-legs --generation 0 --no-live-traffic "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('pos/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'layer ms', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],4), 'MHz', round(d['power']['sclk_mhz']['mean']), 'W', round(d['power']['socket_power_w']['mean']))"
  done
done
