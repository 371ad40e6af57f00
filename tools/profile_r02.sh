#!/bin/bash
# Round-2 profile artefacts (run on the GPU box from the repo root; outputs under gpurun_out/r02prof/).
#   1. rocprofv3 --kernel-trace --stats of the default bench.py command          -> kernel_stats.csv + bench line
#   2. PMC passes (separate runs, --kernel-trace only) on tools/nn_micro.py B=8192: FETCH_SIZE, WRITE_SIZE,
#      MFMA busy / clock, LDS conflicts
# usage: tools/profile_r02.sh [f32|f16]
set -u
P=${1:-f32}
O=gpurun_out/r02prof_$P
mkdir -p $O
export TMPDIR=/tmp
EXTRA=""
[ "$P" != f32 ] && EXTRA="--precision $P"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline $EXTRA > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
find $O/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$O/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python tools/nn_micro.py --batches 8192 --algos 1 --iters 2 $EXTRA > $D.log 2>&1
done
python tools/pmc_traffic.py 8192 9 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json 2> $O/pmc_traffic.err
python tools/pmc_mfma.py "$O/pmc_GRBM_GUI_ACTIVE_SQ_BUSY_CYCLES_SQ_VALU_M" $O/pmc_SQ_LDS_BANK_CONFLICT_SQ_LDS_IDX_ACTIVE > $O/pmc_mfma_lds.csv 2> $O/pmc_mfma.err
# keep the merged directory small: the raw per-dispatch CSVs are large
find $O -name '*kernel_trace.csv' -delete
find $O -name '*counter_collection.csv' -size +4M -delete
head -c 1500 $O/bench_under_rocprof.json; echo; head -8 $O/kernel_stats.csv; cat $O/pmc_mfma_lds.csv; head -c 1500 $O/pmc_traffic.json
