#!/bin/bash
# Builds build/variants/lib<name>.so: libsvo_hip.so with ONE translation unit recompiled with extra
# -D defines (A/B timing of kernel variants on the GPU box: SVO_HIP_LIB=build/variants/lib<name>.so).
# usage: scripts/build_variant.sh <name> <unit, e.g. pose_optimizer_wave> [-DFOO=1 ...]
set -e
cd "$(dirname "$0")/.."
name=$1; unit=$2; shift 2
mkdir -p build/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Irpg_svo_amd/csrc "$@" \
    -c rpg_svo_amd/csrc/$unit.hip -o build/variants/${unit}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/obj/*.o | grep -v "/$unit.hip.o") build/variants/${unit}_$name.o \
    -lhipsolver -o build/variants/lib$name.so
ls -la build/variants/lib$name.so
