#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c8; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=8 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
cp gpurun_out/parity_measured.json gpurun_out/convergence.json $O/ 2>/dev/null
python tools/shipped_step.py 1 eager 100 2>/dev/null | grep shipped_step > $O/b1.txt
python tools/shipped_step.py 1 graph 100 2>/dev/null | grep shipped_step >> $O/b1.txt
python bench.py --no-cpu-baseline --feed-steps 0 --long-steps 100 --autocast-steps 0 --variant-steps 0 --shipped-steps 0 --ddp-steps 0 --no-live-pmc > $O/bench_short.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr8 -o t -- python tools/shipped_step.py 1 eager 30 > /dev/null 2>&1
grep -h "k_wino_weights_batch\|k_wino_split_sum" $(find /tmp/tr8 -name "*kernel_stats.csv" | head -1) > $O/weights_kernel_stats.txt
tail -4 $O/pytest.txt; cat $O/b1.txt; head -c 300 $O/bench_short.json; echo; cat $O/weights_kernel_stats.txt
