#!/bin/bash
# The HOST side of every translation unit of rpg_svo_amd/csrc -- argument checks, workspace carving, launch wrappers --
# against <commit>: x86 assembly of `hipcc --cuda-host-only`, with what depends on file names and line numbers (debug
# directives, the per-source __hip_cuid_ / __hip_fatbin_<hash> symbols, .file / .ident) left out; lambda and anonymous-namespace numbering is NOT normalised:
# a unit that only moved code between files can therefore read DIFFERENT here; `diff` the two .s files it leaves behind to see).
# usage: scripts/host_same_as.sh <commit>
set -u
C=${1:?commit}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/host_same_XXXX)
git -C "$R" archive "$C" rpg_svo_amd/csrc include | tar -x -C "$W"
FL="--offload-arch=gfx950 -O2 -std=c++17 -S --cuda-host-only $*"
norm() { grep -v '^\s*#\|^\s*\.file\|^\s*\.ident\|^\s*\.loc\|hip_cuid\|^\s*\.section\s*\.debug\|^\.L[A-Za-z_]*[0-9]*:$' "$1" | sed 's/\.L[A-Za-z_]*[0-9]\+/.L/g; s/__hip_\(gpubin_handle\|fatbin\)_[0-9a-f]*/__hip_\1_H/g'; }
for src in "$R"/rpg_svo_amd/csrc/*.hip; do
  f=$(basename "$src" .hip)
  (
    [ -f "$W/rpg_svo_amd/csrc/$f.hip" ] || { echo "$f NEW (not in $C)"; exit 0; }
    hipcc $FL -I"$W/include" -I"$W/rpg_svo_amd/csrc" "$W/rpg_svo_amd/csrc/$f.hip" -o "$W/old_$f.s" 2>/dev/null
    hipcc $FL -I"$R/include" -I"$R/rpg_svo_amd/csrc" "$src" -o "$W/new_$f.s" 2>/dev/null
    a=$(norm "$W/old_$f.s" | md5sum | cut -c1-12)
    b=$(norm "$W/new_$f.s" | md5sum | cut -c1-12)
    [ "$a" == "$b" ] && echo "$f same" || echo "$f DIFFERENT ($(diff <(norm "$W/old_$f.s") <(norm "$W/new_$f.s") | grep -c '^[<>]') lines; $W/old_$f.s $W/new_$f.s)"
  ) &
done 2>/dev/null
wait
