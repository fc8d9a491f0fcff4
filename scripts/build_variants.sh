#!/bin/bash
# Builds the debug variants of the sparse-align kernel that scripts/ablate.sh times
# (build/variants/lib{A,B,C,D}.so): fixed iteration count, optionally without the current-image
# loads and/or without the solve/update step.  Run after `python -c "import __graft_entry__ as g; g.build()"`.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/variants
OBJ=build/obj
for v in "A:-DSIA_DBG_FIXED_ITERS" "B:-DSIA_DBG_FIXED_ITERS -DSIA_DBG_NOLOAD" "C:-DSIA_DBG_FIXED_ITERS -DSIA_DBG_NOSOLVE" \
         "D:-DSIA_DBG_FIXED_ITERS -DSIA_DBG_NOLOAD -DSIA_DBG_NOSOLVE"; do
  n=${v%%:*}; d=${v#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Iinclude -Irpg_svo_amd/csrc $d \
      -c rpg_svo_amd/csrc/sparse_align.hip -o build/variants/sa_$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls $OBJ/*.o | grep -v sparse_align) build/variants/sa_$n.o \
      -lhipsolver -o build/variants/lib$n.so
done
ls -la build/variants/*.so
