#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/geo_bench.py: loss kernel vs its read-only twin, cold and warm."""
import csv, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
reps = int(sys.argv[2])
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
probe = [dur(r) for r in rows if 'k_probe_read' in r['Kernel_Name']]
loss = [dur(r) for r in rows if 'k_icp_loss' in r['Kernel_Name']]
med = statistics.median
print(f"bytes per launch: 52 B x 8 x 64 x 2048 = 54.5 MB")
print(f"k_probe_read cold behind dirty lines (1 GB of unrelated read-modify-write before each launch): median {med(probe[:reps]):.2f} us  -> {54.526 / med(probe[:reps]):.2f} TB/s")
print(f"k_probe_read warm  (back to back, operands in the 256 MB infinity cache): median {med(probe[reps:2 * reps]):.2f} us  -> {54.526 / med(probe[reps:2 * reps]):.2f} TB/s")
print(f"k_icp_loss in the pipeline loop (after projection/normals/search): median {med(loss[:reps]):.2f} us  -> {54.526 / med(loss[:reps]):.2f} TB/s")
print(f"k_icp_loss cold behind dirty lines: median {med(loss[reps:2 * reps]):.2f} us  -> {54.526 / med(loss[reps:2 * reps]):.2f} TB/s")
if len(probe) >= 3 * reps and len(loss) >= 3 * reps:
    pc, lc = med(probe[2 * reps:3 * reps]), med(loss[2 * reps:3 * reps])
    print(f"k_probe_read cold behind clean lines (the same flush followed by 1 GiB of unrelated reads): median {pc:.2f} us  -> {54.526 / pc:.2f} TB/s")
    print(f"k_icp_loss cold behind clean lines: median {lc:.2f} us  -> {54.526 / lc:.2f} TB/s  = {54.526 / lc / 8:.3f} of 8 TB/s")
