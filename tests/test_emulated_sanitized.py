"""The kernels' memory accesses, checked without a GPU: the host-emulated C-ABI library (tests/emu_build.py) built with
AddressSanitizer and driven by the emulated parity tests in a child process (the sanitizer's runtime has to be loaded
before the interpreter: scripts/emu_sanitize.sh).  The tests hand the kernels numpy buffers of exactly the sizes the
C ABI asks for, and LDS arrays are plain static arrays with red zones around them: a load or store of K0 / K1 / K2 / K3 /
K5 / the map mirror that leaves its buffer -- on the device a fault at best, silently wrong data at worst -- stops the child
with the source line.  The subset here keeps the CPU suite short; the script without arguments runs all of them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# one test per kernel chain, on the default build (the script without arguments runs all emulated tests on both builds:
# profiles/r04w_emulated_sanitizers.txt)
SUBSET = ["tests/test_sparse_align_emulated.py::test_emulated_sparse_align_edge_cases[default]",
          "tests/test_track_emulated.py::test_emulated_find_match_direct[default-pinhole]",
          "tests/test_track_emulated.py::test_emulated_update_seeds[default-pinhole-0-1]",
          "tests/test_optimizers_emulated.py::test_emulated_pose_optimize[default-pinhole-wave]",
          "tests/test_map_mirror_emulated.py::test_emulated_batches_and_patches[default]",
          "tests/test_fast_emulated.py::test_emulated_fast_detect_empty_and_full_occupancy",
          "-q", "-x", "-p", "no:cacheprovider"]


def _runtime_or_skip(kind):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu_build import sanitizer_runtime
    rt = sanitizer_runtime(kind)
    if rt is None:
        pytest.skip(f"no {kind} sanitizer runtime next to ROCm's clang++")
    return rt


@pytest.fixture(scope="module")
def sweeps(tmp_path_factory, oracle, schedule_runs):
    """both sanitizer sweeps as child processes, started together (they are independent and take about as long; the
    checker they both use is built before they start) -- and, through `schedule_runs`, next to the two reversed-schedule
    runs of the last test: four children on the suite's idle cores instead of one after the other"""
    procs = {}
    for kind in ("address", "thread"):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_build import sanitizer_runtime
        if sanitizer_runtime(kind) is None:
            continue
        log = str(tmp_path_factory.mktemp(kind) / "report")
        out = open(log + ".out", "w")
        procs[kind] = (subprocess.Popen([os.path.join(ROOT, "scripts", "emu_sanitize.sh"), kind, *SUBSET], stdout=out,
                                        stderr=subprocess.STDOUT, cwd=ROOT, env=dict(os.environ, SVO_TSAN_LOG=log)), log)
    done = {}

    def result(kind):
        if kind not in procs:
            pytest.skip(f"no {kind} sanitizer runtime next to ROCm's clang++")
        if kind not in done:
            p, log = procs[kind]
            try:
                rc = p.wait(timeout=1500)
            except subprocess.TimeoutExpired:
                p.kill()
                raise
            done[kind] = (rc, open(log + ".out").read(), log)
        return done[kind]
    yield result
    for p, _ in procs.values():
        if p.poll() is None:
            p.kill()


def test_emulated_kernels_stay_inside_their_buffers(sweeps):
    rc, out, _ = sweeps("address")
    assert "ERROR: AddressSanitizer" not in out, out[-4000:]
    assert rc == 0 and " passed" in out, out[-4000:]


def test_the_sanitizer_sees_a_kernel_leave_its_buffer():
    """(the check of the checker: a pyramid store 4 KiB too small is reported at the kernel's store)"""
    rt = _runtime_or_skip("address")
    code = (
        "import sys, ctypes as C, numpy as np\n"
        "sys.path.insert(0, 'tests')\n"
        "from emu_build import build_emulated\n"
        "from rpg_svo_amd import capi\n"
        "emu = build_emulated(())\n"
        "w, h = 64, 48\n"
        "layout = capi.pyr_layout(w, h, 3)\n"
        "store = np.zeros(capi.pyr_store_bytes(layout, 1) - 4096, np.uint8)\n"
        "img = np.zeros((1, h, w), np.uint8)\n"
        "p = lambda a: C.c_void_p(a.ctypes.data)\n"
        "emu.svo_hip_pyramid_build_tiled(C.byref(layout), p(store), 0, 1, p(img), C.c_longlong(h * w), w, capi.HALFSAMPLE_AUTO, 0, None)\n"
        "print('survived')\n")
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0", SVO_EMU_SANITIZE="address")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode != 0 and "survived" not in r.stdout
    assert "heap-buffer-overflow" in r.stderr and "pyramid.hip" in r.stderr, r.stderr[-3000:]


def _races(log_prefix):
    import glob
    text = "".join(open(f).read() for f in glob.glob(log_prefix + "*"))
    return text.count("WARNING: ThreadSanitizer: data race"), text


def test_emulated_kernels_order_their_exchanges(sweeps):
    """ThreadSanitizer over K1, the matcher's and the depth filter's chains and the map mirror: every work-item is a
    thread of its own to the sanitizer and the only ordering between work-items is what the kernels ask for (workgroup
    barriers, wave-wide collectives and hand-overs, lane-group hand-overs: tests/host/hip_emu.h).  An LDS or global-memory
    exchange between work-items without one of these in between is a data race report, and there must be none."""
    rc, out, log = sweeps("thread")
    n, text = _races(log)
    assert n == 0, text[:6000]
    assert " passed" in out and " failed" not in out, out[-4000:]


@pytest.mark.parametrize("mode,racy", [(0, False), (1, True), (2, True), (3, False)],
                         ids=["workgroup-barrier", "no-barrier", "hand-over-narrower-than-the-exchange", "hand-over-inside-the-wave"])
def test_the_race_detector_sees_a_missing_barrier(tmp_path, mode, racy):
    """(the check of the checker: tests/host/emu_race_probe.cpp)"""
    rt = _runtime_or_skip("thread")
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, 'tests')\n"
        "from emu_build import build_race_probe\n"
        "lib = build_race_probe()\n"
        "a = np.arange(128, dtype=np.int32); o = np.zeros(128, np.int32)\n"
        f"assert lib.probe_neighbour_sum({mode}, a.ctypes.data, o.ctypes.data) == 0\n"
        "print('exact', int((o == a + np.roll(a, -1)).sum()))\n")
    log = str(tmp_path / "tsan")
    env = dict(os.environ, LD_PRELOAD=rt, TSAN_OPTIONS=f"report_signal_unsafe=0:log_path={log}", SVO_EMU_SANITIZE="thread", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    n, text = _races(log)
    assert "exact" in r.stdout, r.stderr[-3000:]
    if racy:
        assert n >= 1 and "emu_race_probe.cpp:13" in text and "emu_race_probe.cpp:17" in text, text[:4000]   # the store and the load
    else:
        assert n == 0, text[:4000]


@pytest.mark.parametrize("mode,between,racy", [(4, "1", False), (5, "0", False), (5, "1", True)],
                         ids=["disjoint-workgroups", "ordered-workgroups-hide-it", "one-workgroup-reads-what-another-writes"])
def test_the_race_detector_between_workgroups(tmp_path, mode, between, racy):
    """SVO_EMU_TSAN_BETWEEN_WORKGROUPS=1: the workgroups of a launch are not ordered (as on the device) and LDS is exempt; a
    workgroup that reads global memory another workgroup of the same launch writes is a report (tests/host/emu_race_probe.cpp)."""
    rt = _runtime_or_skip("thread")
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, 'tests')\n"
        "from emu_build import build_race_probe\n"
        "lib = build_race_probe()\n"
        "a = np.arange(128, dtype=np.int32); o = np.zeros(256, np.int32)\n"
        f"assert lib.probe_neighbour_sum({mode}, a.ctypes.data, o.ctypes.data) == 0\n"
        "print('exact', int((o[128:] == np.roll(a, -1) + 1).sum()))\n")
    log = str(tmp_path / "tsan")
    env = dict(os.environ, LD_PRELOAD=rt, TSAN_OPTIONS=f"report_signal_unsafe=0:log_path={log}", SVO_EMU_SANITIZE="thread", OMP_NUM_THREADS="1",
               SVO_EMU_TSAN_BETWEEN_WORKGROUPS=between)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    n, text = _races(log)
    assert "exact 128" in r.stdout, r.stderr[-3000:]
    assert (n >= 1 and "two_workgroups" in text) if racy else n == 0, text[:4000]


@pytest.fixture(scope="module")
def schedule_runs(tmp_path_factory, oracle):
    """both reversed-schedule runs as child processes, started together (independent, about equally long)"""
    tests = [t for t in SUBSET if t.startswith("tests/")] + ["tests/test_entries_emulated.py::test_emulated_align_batch_and_its_phased_form[default]",
                                                             "tests/test_fast_emulated.py::test_emulated_fast_detect_bit_exact"]
    procs = {}
    for schedule in ("reverse", "waves"):
        log = str(tmp_path_factory.mktemp("schedule_" + schedule) / "out")
        out = open(log, "w")
        procs[schedule] = (subprocess.Popen([sys.executable, "-m", "pytest", *tests, "-q", "-x", "-p", "no:cacheprovider"], stdout=out,
                                            stderr=subprocess.STDOUT, cwd=ROOT, env=dict(os.environ, SVO_EMU_SCHEDULE=schedule)), log)

    def result(schedule):
        p, log = procs[schedule]
        try:
            rc = p.wait(timeout=1500)
        except subprocess.TimeoutExpired:
            p.kill()
            raise
        return rc, open(log).read()
    yield result
    for p, _ in procs.values():
        if p.poll() is None:
            p.kill()


@pytest.mark.parametrize("schedule", ["reverse", "waves"])
def test_results_do_not_depend_on_the_schedule(schedule, schedule_runs):
    """SVO_EMU_SCHEDULE: the emulator lets the work-items and workgroups (reverse), or the waves (waves), take their turns in the
    opposite order.  The parity tests -- bit-exact ones among them: the phased alignment with its atomically filled queues,
    FAST's per-cell maxima, the scan's dynamic group fetch -- must not notice (children: the mode is read once per process)."""
    rc, out = schedule_runs(schedule)
    assert rc == 0 and " passed" in out, out[-4000:]
