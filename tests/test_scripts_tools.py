"""The two small source tools of round 5: scripts/unifdef.py (how losing compile-time experiments were deleted) and
scripts/design_table.py (DESIGN.md's measured table from a bench details file)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_unifdef_resolves_one_symbol_and_keeps_the_rest():
    from unifdef import unifdef
    src = "\n".join([
        "a", "#ifdef KEEP_ME", "k1", "#else", "k2", "#endif",
        "#ifdef GONE", "g1", "#ifdef KEEP_ME", "nested", "#endif", "#else", "g2", "#endif",
        "#ifndef GONE", "n1", "#endif", "#if defined(GONE)", "d1", "#else", "d2", "#endif", "z"])
    off = unifdef(src, "GONE", False).split("\n")
    assert off == ["a", "#ifdef KEEP_ME", "k1", "#else", "k2", "#endif", "g2", "n1", "d2", "z"]
    on = unifdef(src, "GONE", True).split("\n")
    assert on == ["a", "#ifdef KEEP_ME", "k1", "#else", "k2", "#endif", "g1", "#ifdef KEEP_ME", "nested", "#endif", "d1", "z"]


def test_no_experiment_switch_is_left_in_the_kernels():
    """csrc/ keeps three switches: the reference-width build the bench times, the phase counters, the CPU test seams."""
    import glob
    import re
    allowed = {"SIA_F64_PARTIALS", "SIA_PROFILE", "SCAN_PROFILE", "SIA_PACKED", "SVO_HOST_MATH_TEST", "SIA_VCC_SELECT", "ALIGN_PHASE_MIN_M_VALUE", "ALIGN_WAVE_MAX_M_VALUE", "ALIGN_BLOCK_VALUE", "SCAN_CHUNK_VALUE",
               "__HIP_DEVICE_COMPILE__", "SVO_HIP_EMU"}  # (SVO_HIP_EMU: set by tests/host/hip_emu.h next to SVO_HOST_MATH_TEST)
    found = set()
    for f in glob.glob(os.path.join(ROOT, "rpg_svo_amd", "csrc", "*")):
        for m in re.finditer(r"^\s*#\s*(?:ifdef|ifndef|if|elif)\s+(.*)$", open(f).read(), re.M):
            found |= set(re.findall(r"[A-Z_][A-Z0-9_]{3,}", m.group(1)))
    assert found <= allowed, sorted(found - allowed)


def test_design_table_from_a_committed_details_file(tmp_path):
    # (the details file DESIGN.md's table names as its source)
    import re
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    src = os.path.join(ROOT, re.search(r"<!-- measured:begin -->\s*Source: `([^`]+)`", design).group(1))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "design_table.py"), src], capture_output=True, text=True, check=True).stdout
    d = json.load(open(src))
    assert f"{d['value'] / 1e6:.2f} M frames/s" in out and "epi_scan_kernel" in out and "SIA_F64_PARTIALS" in out
    # DESIGN.md carries exactly this table
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert out.strip() in design


def test_flag_screen_tabulates_a_kernel_per_variant():
    """scripts/flag_screen.py (static screening of compiler flags, no GPU): one small translation unit, the default build
    and one variant -- a row per kernel and build with registers, occupancy and an instruction mix."""
    import shutil
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        import pytest
        pytest.skip("no hipcc")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "flag_screen.py"), "--units=point_optimizer",
                          "ilp=-mllvm,-amdgpu-sched-strategy=max-ilp"], capture_output=True, text=True, check=True, timeout=300).stdout
    lines = out.splitlines()
    assert any(l.startswith("point_optimizer:") and "point_opt_kernel" in l for l in lines), out
    rows = [l.split() for l in lines if l.strip().startswith(("default", "ilp"))]
    assert {r[0] for r in rows} == {"default", "ilp"}
    for r in rows:
        d = dict(zip(r[1::2], r[2::2]))
        assert int(d["vgpr"]) > 0 and int(d["occ"]) >= 1 and int(d["valu"]) > 50 and int(d["code"]) > 500, r


def test_env_knobs_runs_its_children_on_the_mock_device(tmp_path):
    """scripts/env_knobs.py (the single-stream drop-in under runtime settings, one child per setting) end to end on the mock
    device: the default setting and the deferred-mapper follow-up, same trajectory, a summary file."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "dropin"))
    import pypipeline as pp
    if not pp.available("hipmock"):
        import pytest
        pytest.skip("tests/dropin/_build/libsvo_pipeline_hipmock.so absent")
    rel = os.path.relpath(str(tmp_path), ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "env_knobs.py"), "frames=16", "flavour=hipmock", "only=NONE", "reps=1",
                        "out=" + rel], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.load(open(os.path.join(str(tmp_path), "env_knobs.json")))
    assert list(d["settings"]) == ["default"] and d["settings"]["default"][0]["tot_time_median_us"] > 0
    assert d["follow_up"]["default, deferred mapper"][0]["pose_checksum"] == d["settings"]["default"][0]["pose_checksum"]
    assert all(d["same_trajectory_as_default"].values())
