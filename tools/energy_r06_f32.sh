#!/bin/bash
# ms / MHz / W / mJ per launch of the 9x9 f32 tower layer (conv2 form, 8192 positions, one chain, back to back): the one-pass
# kernel, its timing variants with VALID operands (1 = K loop only, 2 = no phase 2, 3 = phase 2 without its stores, 4 = no DMA
# after the prologue: operands re-read from the three resident stages, 13 = every stage's DMA from two L2-resident images,
# 11 = y only, 12 = next V only / no residual) and the five-pass kernel.  Needs gpurun_ab/libagz_T.so
# (ALSO="agz_wino agz_wino5" tools/build_timing_lib.sh).
python tools/energy_table.py --precision f32 --board 9 --batch 8192 --seconds 4 --env AGZ_WINO_X --variants 0 1 2 3 4 13 11 12 --swap
python tools/energy_table.py --precision f32 --board 9 --batch 8192 --seconds 4 --env AGZ_WINO_X --variants 0 --winograd 3 --swap
