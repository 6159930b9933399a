#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c7; mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_zz_convergence.py -m gpu -q --timeout 600 > $O/pytest_conv.txt 2>&1; echo "pytest rc $?" >> $O/pytest_conv.txt
cp gpurun_out/convergence.json $O/ 2>/dev/null
grep "measured\]" $O/pytest_conv.txt; tail -3 $O/pytest_conv.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --feed-steps 0 --long-steps 0 --autocast-steps 0 --variant-steps 0 --shipped-steps 0 --kernel-reps 2 --no-live-pmc > $O/bench_stdout.txt 2> $O/bench_stderr.txt; wc -l $O/bench_stdout.txt; head -c 200 $O/bench_stdout.txt
