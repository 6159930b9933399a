#!/bin/bash
# tools/bin/nn_lab (or $1) on the GPU box: check, times per regime, per-kernel times and counters (separate rocprofv3 passes) -> gpurun_out/$2.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
BIN=${1:-tools/bin/nn_lab}
OUT=gpurun_out/${2:-nn_lab}.txt
mkdir -p gpurun_out
{
  echo "== $BIN check"; timeout 300 $BIN check
  echo "== $BIN time 20"; timeout 300 $BIN time 20
  for regime in 0 3; do
    rm -rf /tmp/nnk; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nnk -o nn -- $BIN time 10 $regime > /dev/null 2>&1
    echo "== kernels, regime $regime"; python3 tools/kstats.py $(find /tmp/nnk -name '*kernel_stats.csv' | head -1) k_nn k_fill
    PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU"
    rm -rf /tmp/nnp; rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/nnp -o nn -- $BIN time 2 $regime > /dev/null 2>&1
    PMC2="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"
    rm -rf /tmp/nnq; rocprofv3 --kernel-trace --pmc $PMC2 --output-format csv -d /tmp/nnq -o nn -- $BIN time 2 $regime > /dev/null 2>&1
    echo "== counters, regime $regime"
    python3 - <<'PY'
import csv, glob, collections
def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            if 'k_nn' not in k: continue
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items()}
a, b = load('/tmp/nnp'), load('/tmp/nnq')
for k in a:
    m, n = a[k], b.get(k, {})
    w = max(m.get('SQ_WAVE_CYCLES', 1), 1)
    print(f"{k[:28]:28s} issuing {m.get('SQ_ACTIVE_INST_ANY', 0) / w:.2f}  stalled {m.get('SQ_WAIT_INST_ANY', 0) / w:.2f}  VALU-active {m.get('SQ_ACTIVE_INST_VALU', 0) / w:.2f}  VMEM-active {m.get('SQ_ACTIVE_INST_VMEM', 0) / w:.2f}"
          f"  insts VALU {m.get('SQ_INSTS_VALU', 0):.3g} VMEM {m.get('SQ_INSTS_VMEM', 0):.3g} SALU {m.get('SQ_INSTS_SALU', 0):.3g} LDS {n.get('SQ_INSTS_LDS', 0):.3g}"
          f"  waves {n.get('SQ_WAVES', 0):.0f}  cycles {n.get('GRBM_GUI_ACTIVE', 0) / 8:.0f}  L1 accesses {n.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.3g} -> L2 reads {n.get('TCP_TCC_READ_REQ_sum', 0):.3g}"
          f"  L2 hit {n.get('TCC_HIT_sum', 0):.3g} miss {n.get('TCC_MISS_sum', 0):.3g}")
PY
  done
} > $OUT 2>&1
tail -40 $OUT
