#!/usr/bin/env python3
"""Per-workgroup wall-clock trace of one k_wino5_gemm layer launch (conv2 form: residual in, y and the next V out).  A timing
build writes it:  AGZ_WINO5_TRACE=<file> python tools/nn_micro.py --board 9 --tower 4 --batches 8192 --algos 3 --iters 2
with gpurun_ab/libagz_T5.so in place of libagz.so (ALSO=agz_wino5 OUT=libagz_T5.so tools/build_timing_lib.sh).
Phase durations per workgroup (100 MHz clock)."""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(4096, 24).astype(np.int64)
idx = np.nonzero(a[:, 1] > 0)[0]
t = a[idx] / 100.0          # us
names = [("prologue (first super-stage published)", 1, 2)]
for p in range(5):
    names += [("pass %d: K loop" % p, 2 if p == 0 else 2 + 2 * p, 3 + 2 * p), ("pass %d: fold" % p, 3 + 2 * p, 4 + 2 * p)]
for hh in range(2):
    b = 13 + 4 * hh
    names += [("half %d: residual in (+ outputs out of the AGPRs)" % hh, 12 if hh == 0 else 16, b), ("half %d: image = ReLU(res + value)" % hh, b, b + 1),
              ("half %d: y" % hh, b + 1, b + 2), ("half %d: next V" % hh, b + 2, b + 3)]
for n, p, q in names:
    d = t[:, q] - t[:, p]
    print("%-52s mean %7.2f us  (p10 %6.2f p90 %6.2f)" % (n, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
tot = t[:, 20] - t[:, 1]
kl = sum((t[:, 3 + 2 * p] - t[:, 2 + 2 * p]).mean() for p in range(5))
fo = sum((t[:, 4 + 2 * p] - t[:, 3 + 2 * p]).mean() for p in range(5))
print("workgroup total %.2f us (prologue %.2f, K loops %.2f, folds %.2f, epilogue %.2f); kernel span %.1f us; %d workgroups"
      % (tot.mean(), (t[:, 2] - t[:, 1]).mean(), kl, fo, (t[:, 20] - t[:, 12]).mean(), t[:, 20].max() - t[:, 1].min(), len(idx)))
