export TMPDIR=/tmp
for o in 0 1; do for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES"; do
  D=gpurun_out/pmct_${o}_$(echo $C | cut -d' ' -f1)
  AGZ_TOWER_ORDER=$o rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python tools/nn_micro.py --batches 8192 --algos 1 --iters 3 --tower-persistent 1 > $D.log 2>&1
done; done
python - <<'PY'
import csv,glob,collections
for o in (0,1):
    agg=collections.defaultdict(list); dur=[]
    for f in glob.glob("gpurun_out/pmct_%d_*/**/*counter_collection.csv"%o, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_wino_tower" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]].append(float(r["Counter_Value"])); dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    ms=sum(dur)/len(dur)/1e6
    print("order",o,"tower %.2f ms"%ms, {k:"%.4g"%(sum(v)/len(v)) for k,v in agg.items()})
    f=sum(agg["FETCH_SIZE"])/len(agg["FETCH_SIZE"]); w=sum(agg["WRITE_SIZE"])/len(agg["WRITE_SIZE"])
    print("   per layer: fetch (KiB counter x2 for 16 B/lane reads) %.2f GB, write %.2f GB; clock %.3f GHz, mfma busy %.3f"%(2*f*1024/20/1e9, w*1024/20/1e9, sum(agg["GRBM_GUI_ACTIVE"])/len(agg["GRBM_GUI_ACTIVE"])/8/(ms*1e6), (sum(agg["SQ_VALU_MFMA_BUSY_CYCLES"])/len(agg["SQ_VALU_MFMA_BUSY_CYCLES"]))/(4*256*sum(agg["GRBM_GUI_ACTIVE"])/len(agg["GRBM_GUI_ACTIVE"])/8)))
PY
find gpurun_out -path "*pmct_*" -name "*.csv" -delete; find gpurun_out -name "*.db" -delete
