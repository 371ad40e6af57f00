#!/bin/bash
# same-box comparison of timing builds of the F(4x4,3x3) GEMM: gpurun_ab/libagz_T*.so, X = 0 (and the listed XS)
B=${B:-2048}
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
for rep in 1 2; do
for lib in ${LIBS:-libagz_T.so libagz_T_sb.so libagz_T_la7.so libagz_T_la3.so libagz_T_sb7.so}; do
  cp gpurun_ab/$lib alphago.jl_amd/libagz.so
  for x in ${XS:-0}; do
    echo -n "$lib X=$x "
    AGZ_WINO4_X=$x python tools/nn_micro.py --board 19 --tower 4 --batches $B --algos 1 --iters 5 2>&1 | grep forward_ms | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('conv_ms', round(d['conv_ms_same_layer_loop'],3))"
  done
done; done
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
