#!/bin/bash
# same-box phase traces of two timing builds (gpurun_ab/libagz_At.so, libagz_Bt.so; see tools/trace_gemm4.py)
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
mkdir -p gpurun_out
for i in 1 2; do for v in At Bt; do
  cp gpurun_ab/libagz_$v.so alphago.jl_amd/libagz.so
  AGZ_WINO_TRACE=gpurun_out/g4_trace_$v$i.bin python tools/nn_micro.py --batches 8192 --algos 1 --iters 2 > /dev/null 2>&1
  echo "== $v $i"; python tools/trace_gemm4.py gpurun_out/g4_trace_$v$i.bin
done; done
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
