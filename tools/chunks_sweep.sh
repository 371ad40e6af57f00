#!/bin/bash
# headline step time against the number of tower ranges per forward (AGZ_TOWER_CHUNKS; 2 streams): does a range's V stay in the
# Infinity Cache between consecutive layers when a stream runs its ranges one after the other?  GPU box, repo root.
for rep in 1 2; do
for c in 0 4 8 16 32 64; do
  AGZ_TOWER_CHUNKS=$c python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-alt-precision --no-config-legs --generation 0 --no-live-traffic --no-sustained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('chunks $c', 'ms/step %.2f' % d['ms_per_step'], 'value %.1f' % d['value'], 'layer ms %.3f' % d['roofline']['avg_launch_ms'], 'W %.0f MHz %.0f' % (d['power']['socket_power_w']['mean'], d['power']['sclk_mhz']['mean']))"
done; done
