#!/usr/bin/env python3
"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/nn_micro.py --batches B --algos 1.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python tools/nn_micro.py --batches 8192 --algos 1 --iters 2
  rocprofv3 --pmc WRITE_SIZE ...                                  -d gpurun_out/pmc_WRITE_SIZE ...
  python tools/pmc_traffic.py 8192 9 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > profiles/pmc_traffic.json

Units and calibration (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your
own access pattern"): the counters are KiB per dispatch.  k_wino_gemm's WRITE_SIZE equals the
algorithmic output bytes exactly (B*N^2*256*4 B), k_wino_in's FETCH_SIZE is 1.085x its algorithmic
input (8 B/lane loads) -- so both are taken at face value (x1.0); the halving the guide reports for
16 B/lane streaming reads is NOT observed for the gemm's global_load_lds_dwordx4 stream here
(halved, V alone would read 0.94 GB; the counter says 2.36 GB against 1.89 GB V + 0.34 GB residual).
"""
import collections
import csv
import glob
import json
import sys

B, N = int(sys.argv[1]), int(sys.argv[2])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[3:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = B * N * N
per_kernel = {}
total = 0.0
for k, cs in agg.items():
    if "wino" not in k:
        continue
    per_kernel[k] = {c: 1024.0 * sum(v) / len(v) for c, v in cs.items()}
    total += sum(per_kernel[k].values())
print(json.dumps({
    "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/nn_micro.py --batches {B}, gfx950; KiB counters x1.0 (calibrated on known byte counts, see tools/pmc_traffic.py)",
    "rows_per_launch": rows, "bytes_per_launch": total, "bytes_per_row": total / rows,
    "algorithmic_bytes_per_row": 2.5 * 256 * 4,
    "per_kernel_bytes_per_launch": per_kernel,
}, indent=1))
