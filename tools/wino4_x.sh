#!/bin/bash
# timing variants of the F(4x4,3x3) GEMM (agz_wino4.hip, AGZ_WINO4_X; results are WRONG for X != 0): needs gpurun_ab/libagz_T.so
# = libagz.so with agz_wino4.o built with -DAGZ_TIMING_EXPERIMENTS.  One steady-state conv2-form layer (GEMM + fix-up) at B positions.
# X: 0 = product, 1 = K loops + folds only, 2 = no phase 2, 3 = phase 2 without its stores, 4 = no DMA after the prologue, 5 = no MFMA, 6 = no folds
B=${B:-2048}
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
cp gpurun_ab/libagz_T.so alphago.jl_amd/libagz.so
for x in ${XS:-0 1 2 3 4 5 6 0}; do
  echo -n "X=$x "
  AGZ_WINO4_X=$x python tools/nn_micro.py --board 19 --tower 4 --batches $B --algos 1 --iters 5 2>&1 | grep forward_ms | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('conv_ms', round(d['conv_ms_same_layer_loop'],3))"
done
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
