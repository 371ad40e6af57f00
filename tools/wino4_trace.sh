#!/bin/bash
# wall-clock trace of one F(4x4,3x3) layer launch (needs gpurun_ab/libagz_T.so); output: $1 (default gpurun_out/wino4_trace.bin)
OUT=${1:-gpurun_out/wino4_trace.bin}
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
cp gpurun_ab/libagz_T.so alphago.jl_amd/libagz.so
AGZ_WINO4_TRACE=$OUT python tools/nn_micro.py --board 19 --tower 4 --batches ${B:-2048} --algos 1 --iters 2 > /dev/null 2>&1
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
python tools/trace_wino4.py $OUT
