#!/usr/bin/env python3
"""Top kernels by total time of a rocprofv3 --kernel-trace --stats output directory:  python tools/kernel_top.py <dir> [n]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6:.1f} ms in {sum(int(r['Calls']) for r in rows)} launches")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{r['Name'][:72]:72s} {int(r['Calls']):5d} x {float(r['AverageNs']) / 1e3:9.1f} us = {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
