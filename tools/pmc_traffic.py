#!/usr/bin/env python3
"""profiles/*pmc_traffic*.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the bench.py command
itself (round 3, tools/profile_r03.sh; round 2 used tools/nn_micro.py --batches B --algos 1: same kernels).
Usage: pmc_traffic.py <positions per launch> <board N> <FETCH dir> <WRITE dir>

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python tools/nn_micro.py --batches 8192 --algos 1 --iters 2
  rocprofv3 --pmc WRITE_SIZE ...                                  -d gpurun_out/pmc_WRITE_SIZE ...
  python tools/pmc_traffic.py 8192 9 gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > profiles/pmc_traffic.json

Units and calibration (MI355X_MICROARCH.md, HBM section: "FETCH_SIZE reports exactly 1/2 of the bytes
of a wide coalesced streaming read ... double it", "other access widths and WRITE_SIZE are
uncalibrated: calibrate on a known byte count in your own access pattern").  The counters are KiB per
dispatch.  Known byte counts here: k_wino_gemm writes exactly B*N^2*256*4 B and WRITE_SIZE reports
that number (x1.0); k_wino_in must read its B*N^2*256*4 B input at least once.  Its FETCH_SIZE was
1.085x that with the first version of the kernel and is 0.53x with the LDS-staged one -- the same
bytes now arrive as 128-B requests tallied at 64 B -- so a kernel whose FETCH_SIZE is below its
compulsory input is corrected x2, as the guide prescribes; both raw and corrected values are kept.
The gemm's global_load_lds_dwordx4 stream reads 1.12x (V + residual + the L2 misses of U): x1.0.
"""
import collections
import csv
import glob
import json
import sys


def summarise(B, N, dirs):
    """the pmc_traffic object from the counter_collection.csv files under `dirs` (one FETCH_SIZE pass, one WRITE_SIZE pass)"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = B * N * N
    per_kernel, counts = {}, {}
    total = 0.0
    layers = 0
    for k, cs in agg.items():
        if "wino" not in k and "conv3x3_f16" not in k:
            continue
        # bytes summed over every dispatch of the run; a tower layer = one GEMM dispatch (k_wino_in runs once per forward)
        per_kernel[k] = {c: 1024.0 * sum(v) for c, v in cs.items()}
        counts[k] = max(len(v) for v in cs.values())
        stem = k.rstrip().endswith(", 8>")          # the stem's 8-stage GEMM and its feature-plane transform: listed, not counted
        if ("gemm" in k or "conv3x3_f16" in k) and not stem:
            layers += counts[k]
        # MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane,
        # global_load and buffer_load ... lds alike) -- double it before comparing with a byte count".  Every read of these
        # kernels is of that kind (k_wino_in: 8 B/lane on 64-B runs was calibrated x2 in round 1 on its compulsory input).
        if "FETCH_SIZE" in per_kernel[k]:
            per_kernel[k]["FETCH_SIZE_raw"] = per_kernel[k]["FETCH_SIZE"]
            per_kernel[k]["FETCH_SIZE"] *= 2.0
        if not stem:
            total += sum(v for c, v in per_kernel[k].items() if c in ("FETCH_SIZE", "WRITE_SIZE"))
    # the search kernels of a step ("search / feature / legal kernels reported as HBM GB/s"): bytes per dispatch,
    # FETCH_SIZE as counted (their loads are 1-8 B per lane, not the 16 B/lane streaming reads the x2 rule is calibrated on)
    search = {}
    for k, cs in agg.items():
        short = k.replace("agz::", "")
        if short in ("k_pre", "k_expand", "k_scan", "k_leaf_features", "k_post"):
            n = max(len(v) for v in cs.values())
            search[short] = {"dispatches": n, **{c + "_bytes_per_dispatch": 1024.0 * sum(v) / len(v) for c, v in cs.items()}}
    have = {c for cs in per_kernel.values() for c in cs}
    if not layers or not {"FETCH_SIZE", "WRITE_SIZE"} <= have:
        raise RuntimeError(f"no tower-layer dispatches with both counters under {dirs} (have {sorted(have)}, {layers} layers)")
    return {
        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on the bench.py command, <= {B} positions per launch, {N}x{N}, gfx950; KiB counters; "
                  "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads (raw kept); bytes of ALL "
                  "tower-layer Winograd kernels of the run / number of tower-layer GEMM dispatches (the stem's 8-stage GEMM is listed, not counted)",
        "rows_per_launch": rows, "tower_layer_dispatches": layers, "bytes_per_launch": total / layers, "bytes_per_row": total / layers / rows,
        "algorithmic_bytes_per_row": 2.5 * 256 * 4, "board": N,
        "per_kernel_total_bytes": per_kernel, "per_kernel_dispatches": counts, "search_kernels": search,
    }


if __name__ == "__main__":
    print(json.dumps(summarise(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]), indent=1))
