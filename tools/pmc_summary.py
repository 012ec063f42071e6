#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean counter value per dispatch.
usage: pmc_summary.py <dir> [kernel-substring ...]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
subs = sys.argv[2:] or ["k_encode", "k_hist", "k_tile_planes", "k_tree", "k_layout", "k_nb_scan"]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        key = next((s for s in subs if s in name), None)
        if key is None:
            continue
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in subs:
    if k not in acc:
        continue
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s n=%3d mean=%16.1f" % (c, len(v), sum(v) / len(v)))
