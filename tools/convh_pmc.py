#!/usr/bin/env python3
"""Matrix-core utilisation and wave-state breakdown of the half-precision convolution kernels from a rocprofv3 --pmc pass over
tools/bin/convh_harness time (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if ("k_convh" in n or "k_wgradh<" in n) and int(r["Grid_Size"]) >= 16384:
        agg[(n[:70], r["Grid_Size"], r.get("LDS_Block_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid, lds), v in sorted(agg.items()):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if "GRBM_GUI_ACTIVE" not in m or "SQ_WAVE_CYCLES" not in m:
        continue
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4)
    w = m["SQ_WAVE_CYCLES"]
    print(f"{k} grid {grid}: kernel {cycles:.0f} cycles, MFMA busy {busy / cycles:.3f} of the kernel; waves: issuing {m['SQ_ACTIVE_INST_ANY'] / w:.2f}, "
          f"issue-stalled {m['SQ_WAIT_INST_ANY'] / w:.2f}, waiting {m['SQ_WAIT_ANY'] / w:.2f}; LDS conflict / LDS cycles "
          f"{m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1):.2f}; LDS index-active cycles per CU / kernel cycles "
          f"{m['SQ_LDS_IDX_ACTIVE'] / 256 / cycles:.2f}")
