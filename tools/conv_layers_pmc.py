#!/usr/bin/env python3
"""HBM traffic per convolution launch, by profile-row name: maps the dispatches of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE)
over tools/conv_layers.py back to the rows of its order file (the script launches every (kernel, pass, layer shape) once, in a fixed
order; each op is run twice -- warm-up + measured -- and the MEASURED dispatches are taken).
   python tools/conv_layers_pmc.py order.json fetch_counter_collection.csv write_counter_collection.csv > rNN_conv_hbm_pmc.json
gfx950: read bytes = FETCH_SIZE [KB] x 1024 x 2 (64 B counted per 128-B request of a wide coalesced read, MI355X_MICROARCH.md);
WRITE_SIZE [KB] x 1024 uncorrected."""
import csv, json, sys

FAMILIES = ("k_wino_conv", "k_wino_wgrad<", "k_wino_wgrad(", "k_wgrad_f32", "k_conv_f32", "k_convh", "k_wgradh<", "k_stem_wgrad(")


def family_of(row):
    return row.split(" ")[0]


def dispatches(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = []
    for r in rows:
        n = r["Kernel_Name"]
        for f in FAMILIES:
            if f in n:
                out.append((f.rstrip("(<"), float(r["Counter_Value"])))
                break
    return out


def traffic_rows(order, fetch_csv, write_csv):
    """{profile-row name: {...bytes per launch...}} for the rows of `order` (the dict tools/conv_layers.py writes)."""
    fetch, write = dispatches(fetch_csv, "FETCH_SIZE"), dispatches(write_csv, "WRITE_SIZE")
    res = {}
    pf = pw = 0
    for o in order["order"]:
        fam, n = family_of(o["row"]), o["launches"]
        vals = []
        for seq, pos in ((fetch, "f"), (write, "w")):
            p = pf if pos == "f" else pw
            # two runs of the op (warm-up, measured): skip the first n dispatches of the family, take the next n
            got, skipped, taken = [], 0, 0
            while p < len(seq) and taken < n:
                if seq[p][0] == fam:
                    if skipped < n:
                        skipped += 1
                    else:
                        got.append(seq[p][1]); taken += 1
                p += 1
            if pos == "f":
                pf = p
            else:
                pw = p
            vals.append(sum(got))
        rd, wr = vals[0] * 1024 * 2, vals[1] * 1024
        prev = res.get(o["row"])
        if prev is not None:
            # the same kernel on the same shape in two passes (e.g. Winograd forward and input gradient): launch-weighted mean
            rd, wr, n = rd + prev["read_bytes_corrected_x2"], wr + prev["write_bytes"], n + prev["kernel_launches"]
            op, comp = prev["op"] + " + " + o["op"], prev["compulsory_bytes_per_op"] + int(o["compulsory_bytes"])
        else:
            op, comp = o["op"], int(o["compulsory_bytes"])
        res[o["row"]] = {"op": op, "kernel_launches": n, "read_bytes_corrected_x2": int(rd), "write_bytes": int(wr),
                         "hbm_bytes_per_launch": int((rd + wr) / max(n, 1)), "hbm_bytes_per_op": int(rd + wr),
                         "compulsory_bytes_per_op": comp}
    return res


CORRECTION = ("gfx950: read bytes = FETCH_SIZE x 1024 x 2 (64 B counted per 128-B request); WRITE_SIZE x 1024; Infinity-Cache hits are "
              "counted by these counters (fabric-side requests), so this is traffic beyond the L2, not only HBM")

if __name__ == "__main__":
    order = json.load(open(sys.argv[1]))
    print(json.dumps({"workload": order["workload"] + "; rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/conv_layers.py",
                      "correction": CORRECTION, "launches": traffic_rows(order, sys.argv[2], sys.argv[3])}, indent=1))
