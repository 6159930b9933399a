#!/usr/bin/env python3
"""HBM bytes per launch of the Winograd kernel from two rocprofv3 --pmc passes over `tools/bin/conv_harness wino`
(FETCH_SIZE and WRITE_SIZE in separate runs; gfx950: read bytes = FETCH_SIZE KB x 1024 x 2, MI355X_MICROARCH.md):
   python tools/exp/conv_hbm.py <fetch counter_collection.csv> <write counter_collection.csv> > conv_hbm_pmc.json"""
import collections, csv, json, sys


def per_grid(path, name):
    v = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "k_wino_conv" in r["Kernel_Name"] and int(r["Grid_Size"]) > 100000:
            v[int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return {g: sum(x) / len(x) for g, x in v.items()}


fetch, write = per_grid(sys.argv[1], "FETCH_SIZE"), per_grid(sys.argv[2], "WRITE_SIZE")
rows = {}
for g in sorted(fetch):
    if g in write:
        rows[str(g)] = {"read_bytes_corrected_x2": int(fetch[g] * 1024 * 2), "write_bytes": int(write[g] * 1024),
                        "hbm_bytes_per_launch": int(fetch[g] * 1024 * 2 + write[g] * 1024)}
mean = sum(r["hbm_bytes_per_launch"] for r in rows.values()) / max(1, len(rows))
print(json.dumps({"workload": "tools/bin/conv_harness wino 2: the four stride-1 layer shapes at batch 8, forward and input gradient; rocprofv3 "
                              "--kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; rows keyed by grid size (threads)",
                  "correction": "gfx950: read bytes = FETCH_SIZE x 1024 x 2 (64 B counted per 128-B request); WRITE_SIZE x 1024",
                  "k_wino_conv": {"by_grid": rows, "hbm_bytes_per_launch_mean": int(mean),
                                  "algorithmic_note": "x + y of a layer = 2 x 67 MB (layer1) .. 2 x 8.4 MB (layer4) + the transformed weights"}},
                 indent=1))
