#!/bin/bash
# layer chains x kernel on the headline workload, same box: one-pass (--winograd 1) and five-pass (--winograd 3), 1..4 chains, twice
for rep in 1 2; do
for c in 1 2 3 4; do for w in 1 3; do
  echo -n "chains=$c winograd=$w "
  python bench.py --winograd $w --tower-streams $c --steps 40 --warmup 5 --no-cpu-baseline --no-alt-precision --no-config-legs --generation 0 --no-live-traffic --no-sustained 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('pos/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],3), 'MHz', round(d['power']['sclk_mhz']['mean']), 'W', round(d['power']['socket_power_w']['mean']))"
done; done; done
