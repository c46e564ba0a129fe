#!/usr/bin/env python3
"""Restore HM-16.14's inter checks in a COPY of the reference's TEncCu.cpp.

The reference comments the P/B-slice candidates of TEncCu::xCompressCU out with two `/** ... **/` blocks
(TEncCu.cpp:605-630, 650-796) and replaces the AMP_ENC_SPEEDUP recursion by a plain call (:926-940), so any non-I slice
aborts (TEncCu.cpp:1055).  Config 4 (encoder_lowdelay_P_main.cfg) is therefore defined by vanilla HM: this script turns
the four delimiter lines into ordinary comments and un-comments the `#if AMP_ENC_SPEEDUP ... #else ... #endif` recursion.
usage: restore_inter.py <TEncCu.cpp in> <TEncCu.cpp out>     (carries no HM source text)
"""
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    text = open(src).read()
    lines = text.split("\n")
    closers = [i for i, ln in enumerate(lines) if re.fullmatch(r"\s*\*\*/\s*", ln)]
    if len(closers) != 2:
        raise SystemExit(f"expected two commented-out inter blocks, found {len(closers)} closing delimiters")
    for c in closers:  # the opener of a block is the nearest '/**' line above its '**/' (other '/**' lines are doc comments)
        o = next((i for i in range(c - 1, -1, -1) if re.fullmatch(r"\s*/\*\*\s*", lines[i])), None)
        if o is None or any("*/" in lines[i] for i in range(o + 1, c)):
            raise SystemExit("inter block delimiters not found where expected")
        lines[o] = "// (inter block restored)"
        lines[c] = "// (end of restored inter block)"
    # the AMP_ENC_SPEEDUP recursion: lines '//#if AMP_ENC_SPEEDUP' .. '//#endif' lose their leading '//'
    try:
        a = next(i for i, ln in enumerate(lines) if re.fullmatch(r"\s*//#if AMP_ENC_SPEEDUP\s*", ln))
        b = next(i for i in range(a, len(lines)) if re.fullmatch(r"\s*//#endif\s*", lines[i]))
    except StopIteration:
        raise SystemExit("AMP_ENC_SPEEDUP recursion block not found")
    for i in range(a, b + 1):
        lines[i] = re.sub(r"^(\s*)//", r"\1", lines[i], count=1)
    open(dst, "w").write("\n".join(lines))
    print(f"restored inter checks: {src} -> {dst} (lines {a + 1}-{b + 1} un-commented)")


if __name__ == "__main__":
    main()
