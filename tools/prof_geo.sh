#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pg; mkdir -p gpurun_out/pg
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pg -o geo -- python tools/geo_bench.py ${REPS:-20} ${SHIFT:-0.0} > gpurun_out/pg.log 2>&1
tail -2 gpurun_out/pg.log
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/pg/geo_kernel_stats.csv')))
for r in rows:
    n=r['Name']
    if 'k_' in n[:12] or n.startswith('void k_') or 'reduce_kernel' in n or 'copy' in n.lower()[:120]:
        print(f"{n[:60]:60s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:9.2f} min_us {float(r['MinNs'])/1e3:9.2f}")
PY
