#!/bin/bash
# round 5, second GPU call: whole GPU suite, batch-1 traces (split-K A/B), longer convergence run, feed of 8 ranks on this host
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c2; mkdir -p $O
(tools/bin/conv_harness wino 5 2>&1 | tail -12) > $O/conv_harness_wino.txt 2>&1
for sp in 1 0; do for amp in "" bfloat16; do DL_WINO_SPLIT=$sp python tools/shipped_step.py 1 eager 100 $amp 2>/dev/null | grep shipped_step; done; done > $O/b1_ab.txt
python tools/shipped_step.py 1 graph 100 2>/dev/null | grep shipped_step >> $O/b1_ab.txt
python tools/shipped_step.py 8 eager 40 2>/dev/null | grep shipped_step >> $O/b1_ab.txt
DL_WINO_SPLIT=0 python tools/shipped_step.py 8 eager 40 2>/dev/null | grep shipped_step >> $O/b1_ab.txt
for amp in f32 bf16; do
  a=""; [ $amp = bf16 ] && a=bfloat16
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$amp -o t -- python tools/shipped_step.py 1 eager 30 $a > /dev/null 2>&1
  python tools/step_breakdown.py $(find /tmp/tr_$amp -name "*kernel_trace.csv" | head -1) 20 60 > $O/step_breakdown_b1_$amp.txt 2>&1
done
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
cp gpurun_out/parity_measured.json $O/ 2>/dev/null
timeout 600 python tools/convergence.py --epochs 300 --out $O/convergence.json > $O/convergence.txt 2>&1
timeout 300 python tools/feed_ranks.py --out $O/feed_ranks.json > $O/feed_ranks.txt 2>&1
cat $O/b1_ab.txt; tail -c 2500 $O/pytest.txt; grep -v Index $O/convergence.txt | tail -5; tail -2 $O/feed_ranks.txt
