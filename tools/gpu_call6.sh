#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_convergence.py -m gpu -q --timeout 500 > $O/pytest_conv.txt 2>&1; echo "pytest rc $?" >> $O/pytest_conv.txt
cp gpurun_out/convergence.json $O/ 2>/dev/null
grep "measured\]" $O/pytest_conv.txt; tail -3 $O/pytest_conv.txt
ROUND=r05 bash tools/prof_round.sh > $O/prof_round.txt 2>&1
tail -5 $O/prof_round.txt
