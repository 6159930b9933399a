#!/usr/bin/env python3
"""Print name, calls, average duration (us) of the kernels matching argv[2] from a rocprofv3 kernel_stats.csv."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if len(sys.argv) < 3 or sys.argv[2] in r["Name"]:
        print(f'{r["Name"][:40]:40s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  total {float(r["TotalDurationNs"]) / 1e6:8.3f} ms')
