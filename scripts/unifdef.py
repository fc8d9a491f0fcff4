#!/usr/bin/env python3
"""Resolve the #if / #ifdef / #ifndef blocks of ONE symbol with a known truth value in a source file, in place
(what `unifdef` does; not in this image).  usage: scripts/unifdef.py FILE SYMBOL 0|1
Used in round 5 to delete the compile-time experiments that lost their A/B (or won it and became the only code)."""
import re, sys


def unifdef(text, sym, defined):
    out, stack = [], []
    for line in text.split("\n"):
        st = line.strip()
        m = re.match(r"#\s*(ifdef|ifndef|if)\s+(.*)$", st)
        if m:
            kind, cond = m.group(1), m.group(2).split("//")[0].strip()
            if cond in (sym, f"defined({sym})", f"!defined({sym})"):
                val = defined
                if kind == "ifndef" or cond == f"!defined({sym})":
                    val = not defined
                stack.append([True, val])
                continue
            stack.append([False, None])
        elif re.match(r"#\s*else\b", st) and stack and stack[-1][0]:
            stack[-1][1] = not stack[-1][1]
            continue
        elif re.match(r"#\s*endif\b", st) and stack:
            if stack.pop()[0]:
                continue
        if all(k for o, k in stack if o):
            out.append(line)
    return "\n".join(out)


if __name__ == "__main__":
    f, sym, val = sys.argv[1], sys.argv[2], sys.argv[3] == "1"
    s = open(f).read()
    open(f, "w").write(unifdef(s, sym, val))
