# fp16 tower conv (agz_conv16.hip) timing variants; needs a build with make EXTRA=-DAGZ_TIMING_EXPERIMENTS.
# AGZ_C16_DEBUG = bit mask of what is compiled out (wrong results): 1 epilogue, 2 weight loads, 4 LDS operand reads,
# 8 slab DMA, 16 MFMA, 32 result stores (instantiated: 1 3 5 9 15 32); AGZ_C16_RB=4: 128-row tiles, two workgroups per CU
for d in ${DS:-0 1 3 5 9 15 32}; do echo -n "DEBUG=$d RB=${AGZ_C16_RB:-7} "; AGZ_C16_DEBUG=$d python tools/nn_micro.py --batches ${B:-8192} --board ${N:-9} --precision f16 --iters 5 2>&1 | grep forward_ms | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('conv_ms', round(d['conv_ms_same_layer_loop'],3), 'forward_ms', round(d['forward_ms'],2))"; done
