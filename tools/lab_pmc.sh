#!/bin/bash
# rocprofv3 counter passes over tools/bin/wino_lab (or $1): MFMA busy, wave states, LDS conflicts per kernel / grid
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
BIN=${1:-tools/bin/wino_lab}
mkdir -p gpurun_out
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS"
rm -rf /tmp/lab_pmc; rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/lab_pmc -o lab -- $BIN time 2 > /dev/null 2>&1
PMC2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"
rm -rf /tmp/lab_pmc2; rocprofv3 --kernel-trace --pmc $PMC2 --output-format csv -d /tmp/lab_pmc2 -o lab -- $BIN time 2 > /dev/null 2>&1
python3 - <<'PY' | tee gpurun_out/lab_pmc.txt
import csv, glob, collections
def load(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f: return agg
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'k_wino' not in k or int(r['Grid_Size']) < 60000: continue
        agg[(k.split('(')[0][:40], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg
a, b = load('/tmp/lab_pmc'), load('/tmp/lab_pmc2')
for key in a:
    m = {c: sum(x) / len(x) for c, x in a[key].items()}
    n = {c: sum(x) / len(x) for c, x in b.get(key, {}).items()}
    cyc = m.get('GRBM_GUI_ACTIVE', 0) / 8.0
    w = max(m.get('SQ_WAVE_CYCLES', 1), 1)
    busy = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / 1024
    print(f"{key[0]:42s} grid {key[1]:>7s}: {cyc:9.0f} cycles  MFMA busy {busy / max(cyc, 1):.3f}  issuing {m.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f} stalled {m.get('SQ_WAIT_INST_ANY', 0) / w:.2f} "
          f"wait-LDS {m.get('SQ_WAIT_INST_LDS', 0) / w:.2f}  LDS conflict/active {m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.2f} (LDS active {m.get('SQ_LDS_IDX_ACTIVE', 0) / 256 / max(cyc, 1):.2f} of cycles/CU)")
    if n:
        mf = max(n.get('SQ_INSTS_MFMA', 1), 1)
        print(f"{'':42s}   per MFMA: VALU {n.get('SQ_INSTS_VALU', 0) / mf:.2f}  LDS {n.get('SQ_INSTS_LDS', 0) / mf:.2f}  SALU {n.get('SQ_INSTS_SALU', 0) / mf:.2f}  VMEM {n.get('SQ_INSTS_VMEM', 0) / mf:.3f}")
PY
