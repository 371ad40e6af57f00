#!/bin/bash
# gpurun_ab/libagz_FX.so = the product objects with agz_wino4.o rebuilt with -DAGZ_FIXUP_EXPERIMENTS (tools/wino4_fx.sh)
set -e
C="$(cd "$(dirname "$0")/../alphago.jl_amd/csrc" && pwd)"
make -s -C "$C"
T=/tmp/agz_fx; mkdir -p $T "$C/../../gpurun_ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DAGZ_FIXUP_EXPERIMENTS -c "$C/agz_wino4.hip" -o $T/agz_wino4.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$C/../../gpurun_ab/libagz_FX.so" "$C"/agz_nn.o "$C"/agz_wino.o $T/agz_wino4.o "$C"/agz_conv16.o "$C"/agz_engine.o "$C"/agz_capi.o "$C"/agz_comm.o "$C"/agz_train.o -ldl
ls -la "$C/../../gpurun_ab/libagz_FX.so"
