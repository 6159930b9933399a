#!/usr/bin/env python3
"""Per-kernel breakdown of one steady-state training step from a rocprofv3 kernel trace of bench.py."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_project_scatter')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a, b = idx[k], idx[k + 1]
step = rows[a:b]
t0, t1 = int(step[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
agg = collections.defaultdict(lambda: [0, 0])
for r in step:
    agg[r['Kernel_Name'][:100]][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[r['Kernel_Name'][:100]][1] += 1
print(f"step wall {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels, busy {sum(v[0] for v in agg.values()) / 1e6:.3f} ms")
for n, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{d / 1e6:8.3f} ms {c:4d}  {n}")
