#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a hipcc -S listing (static counts; the caller supplies the path).

usage: isa_blocks.py listing.s kernel-substring [first_label]
Prints, per basic block in program order: label, line, counts of MFMA / other VALU / LDS / VMEM / SALU / waitcnt, and the
block's terminator.  s_barrier and s_setprio are shown as their own rows so that the phases can be read off.
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(tuple(": ;")) or (l.startswith("_Z") and key in l and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    cur = {"label": "entry", "line": start + 1, "mfma": 0, "valu": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "other": 0}
    rows = []

    def flush(term=""):
        nonlocal cur
        cur["term"] = term
        rows.append(cur)

    for i in range(start + 1, end + 1):
        l = lines[i].split(";")[0].strip()
        if not l or l.startswith("."):
            if l.startswith(".LBB") and l.endswith(":"):
                flush("fall")
                cur = {"label": l[:-1], "line": i + 1, "mfma": 0, "valu": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "other": 0}
            continue
        op = l.split()[0]
        if op in ("s_barrier",) or op == "s_setprio":
            flush(l)
            cur = {"label": "  |", "line": i + 1, "mfma": 0, "valu": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "other": 0}
            continue
        if op.startswith("s_cbranch") or op == "s_branch":
            flush(l)
            cur = {"label": "  |", "line": i + 1, "mfma": 0, "valu": 0, "lds": 0, "vmem": 0, "salu": 0, "wait": 0, "other": 0}
            continue
        cur[classify(op)] += 1
    flush("end")
    print(f"{'label':12s} {'line':>6s} {'mfma':>5s} {'valu':>5s} {'lds':>5s} {'vmem':>5s} {'salu':>5s} {'wait':>5s}  terminator")
    for r in rows:
        if r["mfma"] + r["valu"] + r["lds"] + r["vmem"] + r["salu"] + r["wait"] == 0 and not r["term"].startswith(("s_barrier", "s_setprio")):
            continue
        print(f"{r['label']:12s} {r['line']:6d} {r['mfma']:5d} {r['valu']:5d} {r['lds']:5d} {r['vmem']:5d} {r['salu']:5d} {r['wait']:5d}  {r['term']}")


if __name__ == "__main__":
    main()
