#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pe; mkdir -p gpurun_out/pe
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pe -o e -- python $1 > gpurun_out/pe.log 2>&1
tail -2 gpurun_out/pe.log
python - <<PY
import csv, statistics
rows=list(csv.DictReader(open('gpurun_out/pe/e_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for pat in ("$2".split(",")):
    sel=[r for r in rows if pat in r['Kernel_Name']]
    for i in range(0,len(sel),${3:-10}):
        ch=sel[i:i+${3:-10}]
        d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in ch]
        print(pat, len(ch), "median_us %.2f min %.2f"%(statistics.median(d), min(d)))
PY
