#!/bin/bash
# same-box comparison of several builds: tools/ab_multi.sh <rounds> <tag> <tag> ...   (gpurun_ab/libagz_<tag>.so)
R=$1; shift
cp alphago.jl_amd/libagz.so /tmp/libagz_keep.so
for i in $(seq $R); do for v in "$@"; do
  cp gpurun_ab/libagz_$v.so alphago.jl_amd/libagz.so
  echo -n "$v "; python tools/nn_micro.py --batches 8192 --algos 1 --iters 10 2>&1 | grep forward_ms | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('conv_ms', round(d['conv_ms_same_layer_loop'],4), 'fwd', round(d['forward_ms'],3))"
done; done
cp /tmp/libagz_keep.so alphago.jl_amd/libagz.so
