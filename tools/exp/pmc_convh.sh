cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rm -rf /tmp/cp; rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/cp -o conv -- tools/bin/convh_harness time 2 > /dev/null 2>&1
python tools/exp/convh_pmc.py $(find /tmp/cp -name "*counter_collection.csv" | head -1) | grep "GeomConv<3, 1, 1>\|GeomDgrad<3, 1, 1\|k_wgradh<false, 128, 64\|k_wgradh<false, 64, 32, 2, 1, 1"
