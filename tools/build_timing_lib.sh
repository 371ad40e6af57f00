#!/bin/bash
# gpurun_ab/libagz_T.so = the product objects with agz_wino4.o (and, with ALSO=..., other sources) rebuilt with -DAGZ_TIMING_EXPERIMENTS
set -e
cd "$(dirname "$0")/../alphago.jl_amd/csrc"
make -s
T=/tmp/agz_timing_build; mkdir -p $T
OBJS=""
for f in agz_nn agz_wino agz_wino4 agz_wino5 agz_conv16 agz_engine agz_capi agz_comm agz_train; do
  if [[ " agz_wino4 ${ALSO:-} " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGZ_TIMING_EXPERIMENTS ${EXTRA:-} -c $f.hip -o $T/$f.o
    OBJS="$OBJS $T/$f.o"
  else OBJS="$OBJS $f.o"; fi
done
mkdir -p ../../gpurun_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_ab/${OUT:-libagz_T.so} $OBJS -ldl
ls -la ../../gpurun_ab/${OUT:-libagz_T.so}
