#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pe; mkdir -p gpurun_out/pe
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pe -o e -- python $1 > gpurun_out/pe.log 2>&1
tail -3 gpurun_out/pe.log
python - <<'PY'
import csv, statistics
rows=list(csv.DictReader(open('gpurun_out/pe/e_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'k_icp_loss' in r['Kernel_Name'] or 'k_var' in r['Kernel_Name']]
# print in chunks of 30 launches
for i in range(0,len(sel),30):
    ch=sel[i:i+30]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in ch]
    print(ch[0]['Kernel_Name'][:50], len(ch), "median_us %.2f min %.2f"%(statistics.median(d), min(d)))
PY
