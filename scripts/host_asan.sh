#!/bin/bash
# The HOST side of the drop-in -- the reference's control plane + rpg_svo_amd/host (device layer, drop-in bodies, the map
# mirror's bookkeeping, arenas) -- compiled with AddressSanitizer and run on the mock device (tests/dropin, flavour hipmock):
# five configurations x 120 frames, then SVO_HIP_MAP_MIRROR=verify over 230 frames with keyframe removals.  No GPU involved.
# (Once in a dozen runs the mapper-thread configuration ends in `AddressSanitizer CHECK failed ... real___cxa_throw`: the
# preloaded runtime has no __cxa_throw to forward to when the thread shim interrupts a waiting mapper thread with an
# exception -- a limitation of preloading the runtime into an uninstrumented interpreter, not a finding.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -s -j8 -C "$R/tests/dropin" B=_build_asan \
  CXXFLAGS="-O1 -g -std=c++11 -fPIC -w -fno-math-errno -ffp-contract=off -pthread -DSVO_TRACE -fsanitize=address -fno-omit-frame-pointer" \
  _build_asan/libsvo_pipeline_hipmock.so
cat > /tmp/svo_host_asan.py <<'PY'
import sys, os
R = sys.argv[1]
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "dropin")]
import pypipeline as pp
pp.BUILD = os.path.join(pp.HERE, "_build_asan")
from test_dropin_pipeline import _sequence
n, cfgs = (230, (dict(max_n_kfs=4), dict(max_n_kfs=3, defer_mapper=1))) if os.environ.get("SVO_HIP_MAP_MIRROR") == "verify" else \
    (120, (dict(), dict(defer_mapper=1), dict(max_n_kfs=4), dict(pool_slots=7, defer_mapper=1), dict(mapper_thread=1)))
cam, imgs, T = _sequence(n)
for cfg in cfgs:
    st = {}
    r = pp.run_sequence("hipmock", cam, imgs, T, stats_out=st, **cfg)
    print(cfg, "frames", len(r), "keyframes", sum(x["is_keyframe"] for x in r), "mirror calls / fallbacks", st["map_mirror"]["calls"], st["map_mirror"]["fallbacks"])
print("done")
PY
export LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
python /tmp/svo_host_asan.py "$R" 2>&1 | grep -v "INFO\|^$"
SVO_HIP_MAP_MIRROR=verify python /tmp/svo_host_asan.py "$R" 2>&1 | grep -v "INFO\|^$"
