#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=15 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
cp gpurun_out/parity_measured.json gpurun_out/convergence.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
grep -A22 "slowest" $O/pytest.txt | head -30; tail -5 $O/pytest.txt; tail -3 $O/smoke.txt
