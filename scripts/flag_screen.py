#!/usr/bin/env python3
"""Static screening of compiler-flag variants of the gfx950 library (no GPU): every translation unit is compiled to
assembly with the default flags of rpg_svo_amd/build.py plus a variant's extra flags, and per kernel the register
budget, spills, occupancy, code length and instruction mix are tabulated against the default build.
  python scripts/flag_screen.py [--units matcher,depth_filter] name=flag,flag ...
e.g.  python scripts/flag_screen.py ilp=-mllvm,-amdgpu-sched-strategy=max-ilp"""
import glob, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rpg_svo_amd import build as b  # noqa: E402

OUT = os.path.join(ROOT, "build", "flag_screen")


def compile_asm(unit, name, extra):
    src = os.path.join(b.CSRC, unit + ".hip")
    d = os.path.join(OUT, name)
    os.makedirs(d, exist_ok=True)
    asm = os.path.join(d, unit + ".s")
    cmd = [b.HIPCC, *b.FLAGS, "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"), "-I" + b.CSRC, *extra, src, "-o", asm]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        return unit, name, None, r.stderr[-400:]
    return unit, name, asm, ""


def parse(asm):
    """kernel -> dict of the figures the assembler prints after each kernel, plus an instruction mix"""
    out, cur, mix = {}, None, None
    for line in open(asm):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur, mix = m.group(1), {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "waitcnt": 0, "scratch": 0}
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith(".amdhsa_kernel") or s.startswith(".section") and ".rodata" in s:
            pass
        t = s.split()[0] if s and not s.startswith((";", ".")) else ""
        if t.startswith("v_"):
            mix["valu"] += 1
        elif t == "s_waitcnt":
            mix["waitcnt"] += 1
        elif t.startswith("s_"):
            mix["salu"] += 1
        elif t.startswith(("global_", "flat_", "buffer_")):
            mix["vmem"] += 1
        elif t.startswith("scratch_"):
            mix["scratch"] += 1
        elif t.startswith("ds_"):
            mix["lds"] += 1
        m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|codeLenInByte|NumSgprs|LDSByteSize)\s*[:=]\s*(\d+)", s)
        if m:
            out.setdefault(cur, dict(mix))[m.group(1)] = int(m.group(2))
            if m.group(1) == "codeLenInByte":
                out[cur].update(mix)
    return {k: v for k, v in out.items() if "Occupancy" in v}


def short(k):
    m = re.search(r"\d+([a-z_]+_kernel)", k)
    return (m.group(1) if m else k)[:28] + ("<" + k[-12:] + ">" if "ILi" in k else "")


def main():
    args = [a for a in sys.argv[1:]]
    units = [os.path.basename(s)[:-4] for s in sorted(glob.glob(os.path.join(b.CSRC, "*.hip")))]
    variants = [("default", [])]
    for a in args:
        if a.startswith("--units"):
            units = a.split("=", 1)[1].split(",")
        else:
            n, f = a.split("=", 1)
            variants.append((n, f.split(",")))
    jobs = [(u, n, f) for u in units for n, f in variants]
    with ThreadPoolExecutor(max_workers=16) as ex:
        res = list(ex.map(lambda j: compile_asm(*j), jobs))
    tab = {}
    for unit, name, asm, err in res:
        if asm is None:
            print(f"!! {unit} [{name}] failed: {err}")
            continue
        for k, v in parse(asm).items():
            tab.setdefault((unit, k), {})[name] = v
    for (unit, k), byv in sorted(tab.items()):
        d = byv.get("default")
        if d is None:
            continue
        print(f"{unit}:{short(k)}")
        for n, _ in variants:
            v = byv.get(n)
            if v is None:
                continue
            print(f"   {n:14s} vgpr {v.get('NumVgprs',0):3d} agpr {v.get('NumAgprs',0):3d} scratch {v.get('ScratchSize',0):4d} occ {v.get('Occupancy',0)} "
                  f"code {v.get('codeLenInByte',0):6d} valu {v['valu']:5d} salu {v['salu']:5d} vmem {v['vmem']:4d} lds {v['lds']:4d} waitcnt {v['waitcnt']:4d}")


if __name__ == "__main__":
    main()
