#!/usr/bin/env python3
"""Set-up time of a k_wino_gemm4 workgroup from a trace written by a build with -DAGZ_TIMING_EXPERIMENTS
-DAGZ_X_PROLOGUE_STAMP (AGZ_WINO_TRACE=<file>): is the set-up hidden behind the first stages' flight?"""
import sys

import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(16384, 8).astype(np.int64)
idx = np.nonzero(a[:, 1] > 0)[0]
t = a[idx] / 100.0
for n, (p, q) in (("start -> set-up done (2 stages' DMA issued, point table, offsets)", (1, 6)),
                  ("set-up done -> first stage landed + barrier", (6, 7)), ("first barrier -> K loop done", (7, 2))):
    d = t[:, q] - t[:, p]
    print("%-66s mean %7.2f us (p10 %.2f p90 %.2f)" % (n, d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
