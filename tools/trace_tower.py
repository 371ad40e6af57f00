#!/usr/bin/env python3
"""Reads the per-workgroup accounting a timing build of k_wino_tower leaves (AGZ_TOWER_TRACE=<file>, third forward)."""
import sys

import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = raw[:256]
items = a[:, 1]
us = lambda x: x / 100.0
print("items per workgroup %d..%d" % (items.min(), items.max()))
for name, col in (("dependency wait + barrier", 2), ("body", 3), ("drain stores + barrier + flag", 4)):
    per = us(a[:, col]) / np.maximum(items, 1)
    print("%-32s per item: mean %7.2f us (p10 %6.2f p90 %6.2f max %6.2f)" % (name, per.mean(), np.percentile(per, 10), np.percentile(per, 90), per.max()))
span = us(a[:, 6] - a[:, 5])
print("workgroup lifetime: mean %.1f us, min %.1f, max %.1f; kernel span %.1f us" % (span.mean(), span.min(), span.max(), us(a[:, 6].max() - a[:, 5].min())))
xcd = a[:, 0] & 0xff
print("per-XCD lifetime:", [round(float(span[xcd == x].mean()), 1) for x in range(8)])

if len(raw) >= 256 + 32 * 384:
    it = raw[256:256 + 32 * 384].reshape(8, 4, 384, 8)
    t0 = a[:, 5].min()
    print("first quad of every XCD: body time per item (us), and how far apart its four workgroups start an item:")
    for x in range(8):
        n = int(min(items[(xcd == x) & ((a[:, 0] >> 8) == k)][0] for k in range(4)))
        r = it[x][:, :n]
        d = us(r[0, :, 1] - r[0, :, 0])
        lay = r[0, :, 2]
        st = us(r[:, :, 0])
        skew = st.max(0) - st.min(0)
        print("  XCD %d: %3d items, even layers %.1f, odd layers %.1f; by fifths of the run: %s; start skew within the quad: mean %.2f us, by fifths %s, max %.2f"
              % (x, n, d[lay % 2 == 0].mean(), d[lay % 2 == 1].mean(), [round(float(q.mean()), 1) for q in np.array_split(d, 5)],
                 skew.mean(), [round(float(q.mean()), 2) for q in np.array_split(skew, 5)], skew.max()))
