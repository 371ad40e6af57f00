#!/bin/bash
# Energy decomposition of the fp16 tower layer (BASELINE configs[4]'s shard: 19x19, 4096 positions) with VALID operands:
# every variant's MFMAs multiply real weights and real activations (HISTORY.md 12: round 4's variants multiplied zeros or
# constants).  Cumulative: product -> no result stores (AGZ_C16_POLICY=16) -> + no weight re-loads (MEAS 32) -> + no slab DMA
# (96) -> + no LDS operand reads (224) -> + no epilogue arithmetic (480 = MFMAs on real data, nothing else).
# ms / MHz / W per variant (tools/energy_table.py).  GPU box, repo root; writes gpurun_out/r05_c16_meas.txt
O=gpurun_out/r05_c16_meas.txt
python tools/energy_table.py --precision f16 --board 19 --batch 4096 --env AGZ_C16_POLICY --variants 0 8 16 --seconds 3 2>&1 | grep "^#" > $O
python tools/energy_table.py --precision f16 --board 19 --batch 4096 --env AGZ_C16_MEAS --variants 0 32 96 224 480 0 --extra-env AGZ_C16_POLICY=16 --seconds 3 2>&1 | grep "^#" | sed 's/^# /# (stores dropped) /' >> $O
cat $O
