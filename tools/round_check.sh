#!/bin/bash
# One gpurun call at the end of a round: the whole `-m gpu` suite (its parity table and the convergence report are copied next to the log),
# the smoke test, then the profile bundle of tools/prof_round.sh.   gpurun --timeout 1200 -- 'bash tools/round_check.sh'
# then:  python tools/collect_profiles.py r05; cp gpurun_out/final/parity_measured.json profiles/r05_parity_measured.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 --durations=6 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
cp gpurun_out/parity_measured.json gpurun_out/convergence.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt
ROUND=r05 bash tools/prof_round.sh > $O/prof_round.txt 2>&1
tail -4 $O/pytest.txt; tail -2 $O/smoke.txt; ls gpurun_out/profiles_r05 | wc -l
