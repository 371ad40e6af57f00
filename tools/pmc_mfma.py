#!/usr/bin/env python3
"""MFMA-pipe utilisation, effective clock and LDS bank conflicts per kernel from rocprofv3 --pmc passes:

  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d OUT1 -- python tools/nn_micro.py ...
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d OUT2 -- python tools/nn_micro.py ...
  python tools/pmc_mfma.py OUT1 OUT2 > profiles/<name>.csv

mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs) -- GRBM_GUI_ACTIVE is summed
over the 8 XCDs by rocprofv3; eff_clock = GRBM_GUI_ACTIVE / 8 / duration."""
import collections
import csv
import glob
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace(", ", ";")       # keep the CSV one-field-per-comma
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
print("kernel,avg_ns," + ",".join(cols) + ",eff_clock_GHz,mfma_pipe_util,lds_conflict_frac")
for k in sorted(agg):
    if not any(t in k for t in ("wino", "conv3x3")):
        continue
    a = {c: (sum(v) / len(v) if v else 0.0) for c, v in ((c, agg[k].get(c, [])) for c in cols)}
    ns = sum(dur[k]) / len(dur[k])
    clk = a["GRBM_GUI_ACTIVE"] / 8 / ns if a["GRBM_GUI_ACTIVE"] else 0.0
    util = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * a["GRBM_GUI_ACTIVE"] / 8) if a["GRBM_GUI_ACTIVE"] else 0.0
    conf = a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"] if a["SQ_LDS_IDX_ACTIVE"] else 0.0
    print(f"{k},{ns:.0f}," + ",".join(f"{a[c]:.0f}" for c in cols) + f",{clk:.3f},{util:.3f},{conf:.3f}")
