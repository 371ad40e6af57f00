#!/bin/bash
# fp16 tower layer at BASELINE configs[4]'s shard: cache policy of the trickled result stores (AGZ_C16_POLICY: 0 plain, 1 sc1 = write-through,
# 2 nt, 3 sc0 sc1).  ms / MHz / W per variant (tools/energy_table.py), then one PMC
# pass per counter group per variant: LDS conflicts, MFMA busy + clock, TCP->TCC requests, TCC hits / EA traffic.
# Run on the GPU box from the repo root; writes gpurun_out/r05_c16_policy.*
export TMPDIR=/tmp
O=gpurun_out/r05_c16_policy
rm -rf $O; mkdir -p $O
python tools/energy_table.py --precision f16 --board 19 --batch 4096 --env AGZ_C16_POLICY --variants 0 1 3 2 0 1 --seconds 3 > $O.energy.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -o "TCP_TCC_[A-Z_]*\|TCC_[A-Z_]*_sum\|TCP_[A-Z_]*_sum" | sort -u | tr '\n' ' ' > $O.counters_avail.txt
G1="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
G2="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
G3="TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_HIT_sum"
G4="TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum"
for zb in 0 1; do
  i=0
  for C in "$G1" "$G2" "$G3" "$G4"; do
    i=$((i+1))
    AGZ_C16_POLICY=$zb rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/zb$zb/p$i -- python tools/nn_micro.py --board 19 --tower 3 --batches 4096 --precision f16 --iters 2 > $O/zb$zb.p$i.log 2>&1
  done
done
python - <<'PY' > gpurun_out/r05_c16_policy.pmc.txt
import csv,glob,collections
for zb in (0,1):
    agg=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r05_c16_policy/zb{zb}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("void ","")
            if "conv3x3_f16" not in k: continue
            k=k[:60]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"])); dur[k].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k in sorted(agg):
        print(f"POLICY={zb}", k, "avg_us %.1f"%(sum(dur[k])/len(dur[k])/1e3))
        for c,v in sorted(agg[k].items()): print("   %-28s %.5g"%(c, sum(v)/len(v)))
PY
find $O -name '*.csv' -delete; find $O -name '*.db' -delete
cat $O.energy.txt | tail -12; cat gpurun_out/r05_c16_policy.pmc.txt
