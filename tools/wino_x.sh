#!/bin/bash
# timing experiments of the Winograd GEMM (needs a libagz.so built with EXTRA=-DAGZ_TIMING_EXPERIMENTS)
for x in ${XS:-0 11 12 1 2 3 4 5 6}; do
  echo -n "X=$x "
  AGZ_WINO_X=$x python tools/nn_micro.py --batches 8192 --algos 1 --iters 5 ${PREC:+--precision $PREC} 2>&1 | grep forward_ms | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('conv_ms', round(d['conv_ms_same_layer_loop'],3))"
done
