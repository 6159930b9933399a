cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/conv_pmc -o conv -- tools/bin/conv_harness wino 2 > /dev/null 2>&1
python tools/exp/conv_pmc.py $(find /tmp/conv_pmc -name "*counter_collection.csv" | head -1)
PMC2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
rocprofv3 --kernel-trace --pmc $PMC2 --output-format csv -d /tmp/conv_pmc2 -o conv -- tools/bin/conv_harness wino 2 > /dev/null 2>&1
python - <<'PY'
import csv,glob,collections
f=glob.glob('/tmp/conv_pmc2/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'k_wino_conv<' in r['Kernel_Name'] and int(r['Grid_Size'])>100000:
        agg[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
for g,v in agg.items():
    print(g,{k: round(sum(x)/len(x)) for k,x in v.items()})
PY
