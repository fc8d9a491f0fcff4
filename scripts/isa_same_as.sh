#!/bin/bash
# Is the device code of the default build the same as at <commit>?  Cross-compiles every translation unit of
# rpg_svo_amd/csrc at both states for gfx950 and compares the assembly (comments and the per-source __hip_cuid_ symbol
# left out).  For the hours without a GPU: a change that is meant to leave the kernels alone (opt-in variants behind
# #ifdef, host-compile guards) is checked against the last commit the GPU tests ran on.
# usage: scripts/isa_same_as.sh <commit> [extra hipcc flags]
set -u
C=${1:?commit}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/isa_same_XXXX)
git -C "$R" archive "$C" rpg_svo_amd/csrc include | tar -x -C "$W"
FL="--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -S --cuda-device-only $*"
rc=0
for src in "$R"/rpg_svo_amd/csrc/*.hip; do
  f=$(basename "$src" .hip)
  (
    [ -f "$W/rpg_svo_amd/csrc/$f.hip" ] || { echo "$f NEW (not in $C)"; exit 0; }
    hipcc $FL -I"$W/include" -I"$W/rpg_svo_amd/csrc" "$W/rpg_svo_amd/csrc/$f.hip" -o "$W/old_$f.s" 2>/dev/null
    hipcc $FL -I"$R/include" -I"$R/rpg_svo_amd/csrc" "$src" -o "$W/new_$f.s" 2>/dev/null
    a=$(grep -v '^\s*;\|hip_cuid' "$W/old_$f.s" | md5sum | cut -c1-12)
    b=$(grep -v '^\s*;\|hip_cuid' "$W/new_$f.s" | md5sum | cut -c1-12)
    [ "$a" == "$b" ] && echo "$f same" || echo "$f DIFFERENT"
  ) &
done 2>/dev/null
wait
rm -rf "$W"
