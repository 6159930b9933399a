#!/usr/bin/env python3
"""Copy the summaries of a tools/prof_round.sh bundle (gpurun_out/profiles_<round>/) into profiles/ (tracked):
   python tools/collect_profiles.py r01"""
import csv, json, os, shutil, statistics, subprocess, sys

R = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(ROOT, "gpurun_out", f"profiles_{R}"), os.path.join(ROOT, "profiles")
OWN = ["k_project_scatter", "k_project_resolve", "k_normals", "k_nn_tiles", "k_nn_window", "k_nn_scan16", "k_nn_hard16", "k_nn_hard(", "k_nn_pass_b",
       "k_icp_loss", "k_icp_reduce", "k_probe_read"]


def last_json(path):
    for line in reversed(open(path).read().strip().splitlines()):      # (libraries may print after the line: RCCL's version banner did)
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError(path + ": no JSON line")


NEW = os.path.exists(os.path.join(src, "step_breakdown_f32.txt"))          # bundle layout since round 3 (tools/prof_round.sh)
if NEW:
    for a in ("bench.json", "bench_bf16.json", "bench_graph.json", "bench_profiled_f32.json", "bench_profiled_bf16.json"):
        if os.path.exists(os.path.join(src, a)):
            json.dump(last_json(os.path.join(src, a)), open(os.path.join(dst, f"{R}_{a}"), "w"), indent=1)
    for name in ("bench_kernel_stats_f32.csv", "bench_kernel_stats_bf16.csv", "step_breakdown_f32.txt", "step_breakdown_bf16.txt",
                 "geometry_kernel_stats.csv", "loss_calibration.txt", "loss_warm.txt", "loss_cold.txt", "conv_harness.txt", "convh_harness.txt",
                 "conv_layers_float32.txt", "conv_layers_bfloat16.txt", "conv_pmc.txt", "convh_pmc.txt", "scatter_probe.txt",
                 "step_breakdown_64x720_b1_f32.txt", "step_breakdown_64x720_b1_bf16.txt", "step_breakdown_64x720_b8_f32.txt",
                 "shipped_step_b1_ab.txt", "feed_ranks.json", "wino_lab.txt", "wino_lab_pmc.txt", "nn_lab_final.txt", "nn_counts.txt"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{R}_{name}"))
    merged = {"launches": {}}
    for m in ("float32", "bfloat16"):
        f = os.path.join(src, f"conv_hbm_pmc_{m}.json")
        if os.path.exists(f) and os.path.getsize(f):
            d = json.load(open(f))
            merged.setdefault("workload", []).append(d["workload"])
            merged["correction"] = d["correction"]
            merged["launches"].update(d["launches"])
    json.dump(merged, open(os.path.join(dst, f"{R}_conv_hbm_pmc.json"), "w"), indent=1)
else:
    for a, b in [("bench.json", f"{R}_bench.json"), ("bench_profiled.json", f"{R}_bench_profiled.json")]:
        json.dump(last_json(os.path.join(src, a)), open(os.path.join(dst, b), "w"), indent=1)
    shutil.copy(os.path.join(src, "bench_trace", "bench_kernel_stats.csv"), os.path.join(dst, f"{R}_bench_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "geo_trace", "geo_kernel_stats.csv"), os.path.join(dst, f"{R}_geometry_kernel_stats.csv"))
    for name in ("loss_warm.txt", "ring_bench.txt", "conv_harness.txt", "conv_pmc.txt", "conv_hbm_pmc.json", "scatter_probe.txt", "step_breakdown.txt", "miopen_layers.txt"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{R}_{name}"))


def counter(path, name):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        for k in OWN:
            if k in r["Kernel_Name"]:
                vals.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: statistics.mean(v) for k, v in vals.items()}


def geometry_pmc(tag, order):
    fetch = counter(os.path.join(src, "geo_fetch" + tag, "geo_counter_collection.csv"), "FETCH_SIZE")
    write = counter(os.path.join(src, "geo_write" + tag, "geo_counter_collection.csv"), "WRITE_SIZE")
    pmc = {"workload": f"tools/geo_bench.py 5 0.4 {order}: bench batch B=8, 64x2048, points in {order} order, residual motion 0.4 m; rocprofv3 "
                       "--kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes",
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read (MI355X_MICROARCH.md, HBM section) -> read "
                         "bytes = FETCH_SIZE*1024*2; WRITE_SIZE*1024 uncorrected; exact for streaming kernels (k_icp_loss), an upper bound for "
                         "gather-heavy kernels",
           "kernels": {}}
    for k in OWN:
        if k in fetch and k in write:
            rd, wr = fetch[k] * 1024 * 2, write[k] * 1024
            pmc["kernels"][k] = {"FETCH_SIZE_KB_raw": fetch[k], "WRITE_SIZE_KB_raw": write[k], "read_bytes_corrected_x2": rd, "write_bytes": wr,
                                 "hbm_bytes_per_launch": rd + wr}
    if "k_project_scatter" in pmc["kernels"] and "k_project_resolve" in pmc["kernels"]:
        total = pmc["kernels"]["k_project_scatter"]["hbm_bytes_per_launch"] + pmc["kernels"]["k_project_resolve"]["hbm_bytes_per_launch"]
        pmc["dl_project"] = {"hbm_bytes_per_launch_vote_plus_resolve": total,
                             "note": "against the algorithmic 12 B/point + 36 B/pixel of the same batch (bench.py kernels: dl_project, ~102.6 MB)"}
    return pmc


if os.path.exists(os.path.join(src, "geo_fetch_shuffled", "geo_counter_collection.csv")):
    json.dump(geometry_pmc("_shuffled", "shuffled"), open(os.path.join(dst, f"{R}_geometry_pmc_shuffled.json"), "w"), indent=1)
pmc = geometry_pmc("", "raster")
json.dump(pmc, open(os.path.join(dst, f"{R}_geometry_pmc.json"), "w"), indent=1)
if not NEW:
    trace = os.path.join(src, "bench_trace", "bench_kernel_trace.csv")
    if os.path.exists(trace):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "step_breakdown.py"), trace, "20", "40"], capture_output=True, text=True)
        open(os.path.join(dst, f"{R}_step_breakdown.txt"), "w").write(out.stdout)
    elif os.path.exists(os.path.join(src, "step_breakdown.txt")):      # computed on the GPU box (the raw trace is too large to travel)
        shutil.copy(os.path.join(src, "step_breakdown.txt"), os.path.join(dst, f"{R}_step_breakdown.txt"))
    geo = os.path.join(src, "geo_trace", "geo_kernel_trace.csv")
    if os.path.exists(geo):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loss_calibration.py"), geo, "30"], capture_output=True, text=True)
        open(os.path.join(dst, f"{R}_loss_calibration.txt"), "w").write(out.stdout)
print("k_icp_loss PMC bytes/launch:", pmc["kernels"].get("k_icp_loss", {}).get("hbm_bytes_per_launch"))
