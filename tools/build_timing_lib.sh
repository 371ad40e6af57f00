#!/bin/bash
# gpurun_ab/libagz_T.so = the product objects with the sources named in ALSO (default: agz_wino4) rebuilt with
# -DAGZ_TIMING_EXPERIMENTS.  Since round 6 the timing variants and wall-clock stamps of the two Winograd GEMM files live in
# frozen copies under tools/experiments/ (agz_wino_variants.hip, agz_wino4_variants.hip: the product files hold product code
# only); a source named in ALSO is taken from there when such a copy exists.
#   ALSO="agz_wino agz_wino5" OUT=libagz_T5.so tools/build_timing_lib.sh
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/alphago.jl_amd/csrc"
make -s
T=/tmp/agz_timing_build; mkdir -p $T
OBJS=""
for f in agz_nn agz_wino agz_wino4 agz_wino5 agz_conv16 agz_engine agz_capi agz_comm agz_train; do
  if [[ " ${ALSO:-agz_wino4} " == *" $f "* ]]; then
    SRC=$f.hip
    [ -f "$ROOT/tools/experiments/${f}_variants.hip" ] && SRC="$ROOT/tools/experiments/${f}_variants.hip"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -DAGZ_TIMING_EXPERIMENTS ${EXTRA:-} -c $SRC -o $T/$f.o
    OBJS="$OBJS $T/$f.o"
  else OBJS="$OBJS $f.o"; fi
done
mkdir -p ../../gpurun_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/${OUT:-libagz_T.so} $OBJS -ldl
ls -la ../../gpurun_ab/${OUT:-libagz_T.so}
