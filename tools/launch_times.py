#!/usr/bin/env python3
"""Durations of the individual launches of one kernel (in launch order) from a rocprofv3 --kernel-trace CSV.
usage: tools/launch_times.py <dir with *kernel_trace.csv> <substring of the kernel name> [last N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if sys.argv[2] in r["Kernel_Name"]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
print(sys.argv[2], "us:", [round(x) for x in d[-n:]])
