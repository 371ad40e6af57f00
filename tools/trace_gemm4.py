#!/usr/bin/env python3
"""Per-workgroup wall-clock trace of one k_wino_gemm4 layer launch (a -DAGZ_TIMING_EXPERIMENTS build writes it:
AGZ_WINO_TRACE=<file> python tools/nn_micro.py --batches 8192 --algos 1 --iters 2): phase durations per workgroup,
and the idle gap on a CU between one workgroup's end and the next one's start."""
import collections
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(16384, 8).astype(np.int64)
idx = np.nonzero(a[:, 1] > 0)[0]
t = a[idx] / 100.0          # us
names = ["start -> K loop done", "phase 1: residual issue + inverse transform", "phase 1: wait for the residual, barrier",
         "phase 1: image = ReLU(residual + value)", "phase 1b (y)", "phase 2 (next V)"]
for n, (p, q) in zip(names, [(1, 2), (2, 6), (6, 7), (7, 3), (3, 4), (4, 5)]):
    d = t[:, q] - t[:, p]
    print("%-46s mean %7.2f us  (p10 %6.2f p90 %6.2f)" % (n, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
print("workgroup total %.2f us; kernel span %.1f us; %d workgroups" % ((t[:, 5] - t[:, 1]).mean(), t[:, 5].max() - t[:, 1].min(), len(idx)))
hw = a[:, 0] & 0xffffffff
key = ((a[:, 0] >> 32) & 0xf) << 16 | ((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 4 | ((hw >> 8) & 0xf)
per = collections.defaultdict(list)
for b in idx:
    per[int(key[b])].append(b)
gaps = []
for bs in per.values():
    bs = sorted(bs, key=lambda b: a[b, 1])
    gaps += [(a[y_, 1] - a[x_, 5]) / 100.0 for x_, y_ in zip(bs[:-1], bs[1:])]
gaps = np.array(gaps)
print("CUs %d, workgroups per CU %d..%d; gap between a workgroup's end and the next one's start on its CU: mean %.2f us (p10 %.2f p90 %.2f)"
      % (len(per), min(map(len, per.values())), max(map(len, per.values())), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90)))
