#!/bin/bash
# build_patched_variant.sh NAME 'sed-expression' [file ...]: a whole-library A/B variant built from a COPY of csrc/ with one sed
# expression applied to the named files (default: every file) -> build/variants/libsvo_hip_NAME.so.  No switch enters the tree.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); name=$1; expr=$2; shift 2
T=$(mktemp -d /tmp/variant_XXXX); mkdir -p $T/csrc $T/obj $R/build/variants
cp $R/rpg_svo_amd/csrc/* $T/csrc/
if [ $# -eq 0 ]; then set -- $(cd $T/csrc && ls); fi
for f in "$@"; do sed -i -E "$expr" $T/csrc/$f; done
if diff -rq $R/rpg_svo_amd/csrc $T/csrc > /dev/null; then echo "variant $name: the expression changed nothing" >&2; exit 1; fi
ls $T/csrc/*.hip | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -c -I$R/include -I$T/csrc {} -o $T/obj/\$(basename {}).o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $T/obj/*.o -lhipsolver -o $R/build/variants/libsvo_hip_$name.so
rm -rf $T; echo $R/build/variants/libsvo_hip_$name.so
