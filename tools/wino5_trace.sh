#!/bin/bash
# wall-clock traces of one steady-state layer launch of the five-pass (k_wino5_gemm) and the one-pass (k_wino_gemm4) F(3x3,3x3)
# kernels on the same box (needs gpurun_ab/libagz_T5.so: ALSO="agz_wino agz_wino5" OUT=libagz_T5.so tools/build_timing_lib.sh)
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
cp gpurun_ab/libagz_T5.so alphago.jl_amd/libagz.so
AGZ_WINO5_TRACE=gpurun_out/wino5_trace.bin python tools/nn_micro.py --board 9 --tower 4 --batches ${B:-8192} --algos 3 --iters 2 2>/dev/null
AGZ_WINO_TRACE=gpurun_out/gemm4_trace.bin python tools/nn_micro.py --board 9 --tower 4 --batches ${B:-8192} --algos 1 --iters 2 2>/dev/null
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
echo "--- k_wino5_gemm"; python tools/trace_wino5.py gpurun_out/wino5_trace.bin
echo "--- k_wino_gemm4"; python tools/trace_gemm4.py gpurun_out/gemm4_trace.bin
