#!/bin/bash
# ad-hoc PMC comparison of the tower-layer GEMM kernels on tools/nn_micro.py (one counter group per pass);
# usage: tools/pmc_x.sh "<counters pass 1>" "<counters pass 2>" ...   (run on the GPU box from the repo root)
export TMPDIR=/tmp
O=gpurun_out/r03pmcx
rm -rf $O; mkdir -p $O
i=0
for C in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python tools/nn_micro.py --batches 8192 --algos ${ALGOS:-1 2} --iters 1 > $O/p$i.log 2>&1
done
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r03pmcx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","")
        if "gemm" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k in sorted(agg):
    print(k, "avg_us %.1f"%(sum(dur[k])/len(dur[k])/1e3), "dispatches", len(dur[k])//max(1,len(agg[k])))
    for c,v in sorted(agg[k].items()): print("   %-28s %.4g"%(c, sum(v)/len(v)))
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
