#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing, split at the last MFMA (main loop | epilogue).
Usage: isa_count.py <file.s> <substring of the mangled kernel name> [top]"""
import collections
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    s = open(path).read()
    names = [m.group(1) for m in re.finditer(r'^(\w+):\s*; @\1', s, re.M) if pat in m.group(1)]
    for name in names:
        a = s.index("\n" + name + ":")
        b = s.index("s_endpgm", a)
        lines = [ln.strip() for ln in s[a:b].split("\n")[1:] if ln.strip() and not ln.strip().startswith((".", ";", "//"))]
        lines = [ln for ln in lines if not ln.endswith(":")]
        mf = [i for i, ln in enumerate(lines) if "v_mfma" in ln]
        cut = mf[-1] + 1 if mf else 0
        print(f"== {name}: {len(lines)} instructions, {len(mf)} MFMA, epilogue = {len(lines) - cut}")
        c = collections.Counter(ln.split()[0] for ln in lines[cut:])
        print("  ".join(f"{k}:{v}" for k, v in c.most_common(top)))


if __name__ == "__main__":
    main()
