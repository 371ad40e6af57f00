#!/bin/bash
# timing variants of the F(4x4,3x3) fix-up transform (k_wino4_in<true>; AGZ_WINO4_FX: 1 no stores, 2 no loads, 3 streaming stores (round-4 form), 4 no kernel; splitting the
# channel passes over 4x the workgroups measured +-0 and was removed): needs gpurun_ab/libagz_FX.so = libagz.so with agz_wino4.o built with
# -DAGZ_FIXUP_EXPERIMENTS.  One steady-state conv2-form layer (GEMM + fix-up) at B positions; results WRONG for 1, 2, 4.
B=${B:-2048}
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
cp gpurun_ab/libagz_FX.so alphago.jl_amd/libagz.so
for x in ${XS:-0 4 1 2 3 0 4}; do
  echo -n "FX=$x "
  AGZ_WINO4_FX=$x python tools/nn_micro.py --board 19 --tower 4 --batches $B --algos 1 --iters ${ITERS:-5} 2>&1 | grep forward_ms | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('conv_ms', round(d['conv_ms_same_layer_loop'],4), 'forward_ms', round(d['forward_ms'],3))"
done
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
