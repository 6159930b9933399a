#!/usr/bin/env python3
"""Print (kernel, calls, average us) of a rocprofv3 *kernel_stats.csv, optionally filtered by substrings: kstats.py file.csv [substr ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r.get("Name") or r.get("Kernel_Name") or ""
    if len(sys.argv) > 2 and not any(s in n for s in sys.argv[2:]):
        continue
    print(f"{float(r['AverageNs']) / 1e3:9.1f} us  x{int(r['Calls']):5d}  {float(r['Percentage']):5.1f} %  {n[:90]}")
