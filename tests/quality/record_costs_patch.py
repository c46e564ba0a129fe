#!/usr/bin/env python3
"""TEST/QUALITY INFRASTRUCTURE.  Writes a copy of HM-16.14's TEncCu.cpp in which xCompressCU reports, for every CU node it
evaluates both ways, the RD cost of the best non-split mode and of the four-way split -- the two numbers its own
xCheckBestMode compares (TEncCu.cpp:1036) -- to a recorder in the harness (oracle/ref_rdo_harness.cpp, FHREF_RECORD_COSTS).
Used by `make -C oracle costs` for the label generator (tests/quality/make_labels.py --costs): cost-sensitive training needs to know
what a wrong decision costs, not only which decision was right.  The script carries no HM source text.

usage: record_costs_patch.py <TEncCu.cpp of the HM tree> <output file>
"""
import re
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    s = open(src).read()
    s, n = re.subn(r'(#include <algorithm>\n)', r'\1extern "C" void fhref_record_split( unsigned ctu, unsigned zorder, unsigned depth, double costNoSplit, double costSplit );\n', s, count=1)
    if n != 1:
        raise SystemExit("anchor not found: includes")
    s, n = re.subn(r'^(\s*)(xCheckBestMode\( rpcBestCU, rpcTempCU, uiDepth [^\n]*// RD compare current larger prediction\n)',
                   r'\1fhref_record_split( rpcBestCU->getCtuRsAddr(), rpcBestCU->getZorderIdxInCtu(), uiDepth, rpcBestCU->getTotalCost(), rpcTempCU->getTotalCost() );\n\1\2',
                   s, count=1, flags=re.M)
    if n != 1:
        raise SystemExit("anchor not found: split comparison")
    open(dst, "w").write(s)
    print("patched TEncCu.cpp ->", dst)


if __name__ == "__main__":
    main()
