#!/usr/bin/env python3
"""Matrix-core utilisation of the Winograd kernel from a rocprofv3 --pmc pass (tools/prof_round.sh):
MFMA busy cycles per SIMD / kernel cycles, effective clock, wait breakdown of the waves."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
KERNEL = sys.argv[2] if len(sys.argv) > 2 else "k_wino_conv"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if (KERNEL + "<" in r["Kernel_Name"] or KERNEL + "(" in r["Kernel_Name"]) and int(r["Grid_Size"]) > 100000:
        agg[(r["Kernel_Name"][:28], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0                     # summed over the 8 XCDs
    busy = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4)         # per SIMD
    w = m["SQ_WAVE_CYCLES"]
    print(f"{k} grid {grid}: kernel {cycles:.0f} cycles, MFMA busy {busy:.0f} cycles/SIMD = {busy / cycles:.3f} of the kernel; "
          f"waves: issuing {m['SQ_ACTIVE_INST_ANY'] / w:.2f}, issue-stalled {m['SQ_WAIT_INST_ANY'] / w:.2f}, waiting {m['SQ_WAIT_ANY'] / w:.2f}; "
          f"LDS conflict cycles / LDS cycles {m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1.0):.2f}")
