#!/bin/bash
# Round-6 profile artefacts, all from the bench.py command ITSELF (round 2's PMC passes ran tools/nn_micro.py).
# Run on the GPU box from the repo root; outputs under gpurun_out/r06prof_<tag>/; copy what is to be judged into profiles/.
#   1. rocprofv3 --kernel-trace --stats of `python bench.py <bench args>`            -> kernel_stats.csv + the bench line of that run
#   2. four PMC passes (separate runs, --kernel-trace only) of the same command with few steps:
#      FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
#   3. tools/pmc_traffic.py / tools/pmc_mfma.py summaries
# usage: tools/profile_r03.sh <tag> <games> <board> [bench args...]      e.g.  tools/profile_r03.sh c4 256 19 --board 19 --tower 20 --readouts 800 --games 256
set -u
TAG=$1; GAMES=$2; BOARD=$3; shift 3
O=gpurun_out/r06prof_$TAG
mkdir -p $O
export TMPDIR=/tmp
STEPS=${STEPS:-40}
PSTEPS=${PSTEPS:-3}
COMMON="--no-cpu-baseline --no-alt-precision --no-config-legs --generation 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps $STEPS --warmup 5 $COMMON "$@" > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
find $O/stats -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  D=$O/pmc_$(echo $C | tr ' ' '_' | cut -c1-40)
  # (counter passes with ONE tower chain: with two, the halves of a layer are two overlapping dispatches and per-dispatch
  # counters / per-layer byte counts stop meaning "a layer"; the bytes and the MFMA cycles of a layer do not depend on it)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python bench.py --steps $PSTEPS --warmup 1 $COMMON --tower-streams 1 "$@" > $D.log 2>&1
done
python tools/pmc_traffic.py $((GAMES * 8)) $BOARD $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_traffic.json 2> $O/pmc_traffic.err
python tools/pmc_mfma.py "$O/pmc_GRBM_GUI_ACTIVE_SQ_BUSY_CYCLES_SQ_VALU_M" $O/pmc_SQ_LDS_BANK_CONFLICT_SQ_LDS_IDX_ACTIVE > $O/pmc_mfma_lds.csv 2> $O/pmc_mfma.err
# keep the merged directory small: the raw per-dispatch CSVs are large
find $O -name '*kernel_trace.csv' -delete
find $O -name '*counter_collection.csv' -delete
find $O -name '*.db' -delete
head -c 1200 $O/bench_under_rocprof.json; echo; head -12 $O/kernel_stats.csv; cat $O/pmc_mfma_lds.csv; head -c 1200 $O/pmc_traffic.json; cat $O/pmc_traffic.err | tail -3
