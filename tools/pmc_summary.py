#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs (one pass per directory) per fhevc kernel -> JSON.
usage: tools/pmc_summary.py <dir with pass*/...> <out.json>"""
import collections
import csv
import glob
import json
import sys

src, out = sys.argv[1], sys.argv[2]
res = {}
for p in sorted(glob.glob(src + "/pass*/*/*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = collections.defaultdict(set)
    for r in csv.DictReader(open(p)):
        if "fhevc" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].split("(anonymous namespace)::")[-1].split("(")[0] + "@grid" + r["Grid_Size"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        seen[k].add(r["Dispatch_Id"])
    for k in agg:
        for c, v in agg[k].items():
            res.setdefault(k, {})[c] = v / len(seen[k])
        res[k]["dispatches_averaged"] = len(seen[k])
for k, d in res.items():
    if "FETCH_SIZE" in d:
        # MI355X_MICROARCH.md: on gfx950 FETCH_SIZE (KB) reports half of the bytes of a wide coalesced streaming read
        d["hbm_read_bytes_corrected"] = 2.0 * d["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True))
