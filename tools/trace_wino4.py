#!/usr/bin/env python3
"""Per-workgroup wall-clock trace of one k_wino4_gemm layer launch (conv2 form: residual in, y and the next V out).  A timing
build writes it:  AGZ_WINO4_TRACE=<file> python tools/nn_micro.py --board 19 --tower 4 --batches 2048 --algos 1 --iters 2
with gpurun_ab/libagz_T.so in place of libagz.so (tools/build_timing_lib.sh).  Phase durations per workgroup (100 MHz clock)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(8192, 20).astype(np.int64)
idx = np.nonzero(a[:, 1] > 0)[0]
t = a[idx] / 100.0          # us
names = [("prologue (first stage published)", 1, 2)] + [("pass %d: K loop + fold" % p, 2 + p, 3 + p) for p in range(6)]
for hh in range(2):
    b = 9 + 4 * hh
    names += [("half %d: residual in" % hh, b - 1 if hh == 0 else 12, b), ("half %d: image = ReLU(res + value)" % hh, b, b + 1),
              ("half %d: y" % hh, b + 1, b + 2), ("half %d: next V" % hh, b + 2, b + 3)]
for n, p, q in names:
    d = t[:, q] - t[:, p]
    print("%-40s mean %7.2f us  (p10 %6.2f p90 %6.2f)" % (n, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
tot = t[:, 16] - t[:, 1]
print("workgroup total %.2f us (K loops + folds %.2f, epilogue %.2f); kernel span %.1f us; %d workgroups"
      % (tot.mean(), (t[:, 8] - t[:, 2]).mean(), (t[:, 16] - t[:, 8]).mean(), t[:, 16].max() - t[:, 1].min(), len(idx)))
