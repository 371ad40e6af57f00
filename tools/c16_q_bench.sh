#!/bin/bash
# BASELINE configs[4] shard (19x19, tower 20, 1600 readouts, 512 games, fp16 tower): the product against AGZ_C16_Q=1 / 2
# (2 x 2 wave arrangement for both half-in / half-out forms / for the no-residual form only), alternating on one box.
for rep in 1 2; do for q in 0 1 2; do
  AGZ_C16_Q=$q python bench.py --board 19 --tower 20 --readouts 1600 --games 512 --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --no-sustained 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('Q=$q', 'ms/step %.2f' % d['ms_per_step'], 'value %.2f' % d['value'], 'layer ms %.4f' % d['roofline']['avg_launch_ms'], 'W %.0f MHz %.0f' % (d['power']['socket_power_w']['mean'], d['power']['sclk_mhz']['mean']))"
done; done
