#!/usr/bin/env python3
"""Idle time between the kernels of a self-play step, from a rocprofv3 --kernel-trace CSV of bench.py:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-precision
   python tools/step_gaps.py gpurun_out/kt
A step starts at a k_pre dispatch.  Prints the span, the kernel time and the idle time per step, and the gaps by
(previous kernel -> next kernel)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("agz::", "")
    return n[:44]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + root)
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_pre")]
    if len(starts) < 4:
        sys.exit("fewer than 4 steps in the trace")
    starts = starts[2:]                       # skip the first steps (warm-up, lazy initialisation)
    spans, busy = [], []
    gaps = collections.defaultdict(list)
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b + 1]                   # up to and including the next step's k_pre (for the last gap)
        spans.append(seg[-1][0] - seg[0][0])
        busy.append(sum(e - s for s, e, _ in seg[:-1]))
        for (s0, e0, n0), (s1, e1, n1) in zip(seg[:-1], seg[1:]):
            gaps[(n0, n1)].append(s1 - e0)
    n = len(spans)
    span = sum(spans) / n / 1e6
    kern = sum(busy) / n / 1e6
    print("%d steps: span %.3f ms, kernels %.3f ms, idle %.3f ms (%.2f %%), %d dispatches per step"
          % (n, span, kern, span - kern, 100 * (span - kern) / span, (starts[1] - starts[0])))
    print("gaps per step by (previous -> next), microseconds:")
    for (a, b), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        print("  %-46s -> %-46s  x%-3d mean %7.2f  per step %8.2f" % (a, b, len(v) // n, sum(v) / len(v) / 1e3, sum(v) / n / 1e3))


if __name__ == "__main__":
    main()
