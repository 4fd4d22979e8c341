#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs per kernel: mean of each counter over dispatches.
usage: tools/pmc_summary.py <dir> [<dir> ...]"""
import csv, sys, glob, collections, re
for d in sys.argv[1:]:
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            name = re.sub(r"void smhip::sm_kernel<smhip::(\w+)(<.*)?>\(.*", r"\1", row["Kernel_Name"])
            m = re.search(r"SPlan<(\d+)", row["Kernel_Name"])
            if m: name += "<" + m.group(1) + ">"
            agg[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", d)
    for name, cs in sorted(agg.items()):
        n = max(len(v) for v in cs.values())
        print(f"{name:24s} n={n:3d} " + " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
