#!/bin/bash
# headline step time against the number of tower layer chains (agz_net_set_tower_streams), alternating on one box
for rep in 1 2; do
  for n in 1 2 3 4; do
    echo -n "chains=$n "
    python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-alt-precision --generation 0 --no-sustained --tower-streams $n "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value'],1), round(d['ms_per_step'],2), round(d['power']['sclk_mhz']['mean']), round(d['power']['socket_power_w']['mean']))"
  done
done
