#!/usr/bin/env python3
"""print the top kernels of a rocprofv3 --kernel-trace --stats CSV.  usage: summarize_kernel_stats.py <kernel_stats.csv> [n]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms:", round(tot / 1e6, 2), " launches:", sum(int(r["Calls"]) for r in rows))
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:84]
    print("%-84s %6s %8.2fms %8.1fus %5.1f%%" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                               100 * float(r["TotalDurationNs"]) / tot))
