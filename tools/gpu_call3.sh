#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c3; mkdir -p $O
# (1) why is the fp32 training run slow?  a short run under the kernel trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_conv -o t -- python tools/convergence.py --epochs 8 --precisions float32 --out $O/conv_short.json > $O/conv_short.txt 2>&1
python tools/step_breakdown.py $(find /tmp/tr_conv -name "*kernel_trace.csv" | head -1) 100 40 > $O/conv_step_breakdown.txt 2>&1
grep -c "MODULE path" $O/conv_short.txt >> $O/conv_step_breakdown.txt
# (2) eight ranks on one GPU, traced and bounded
DELORA_BENCH_TRACE=1 DELORA_BENCH_SHARE_GPU=1 DELORA_BENCH_BACKEND=gloo OMP_NUM_THREADS=1 timeout -s KILL 240 python bench.py --gpus 8 --steps 3 --warmup 1 --batch 1 --width 256 --rotate 2 --kernel-reps 2 --no-live-pmc --no-profile > $O/bench8.json 2> $O/bench8.err; echo "rc $?" >> $O/bench8.err
# (3) the tests that changed
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "packed_feed or product_loop or run_training_cli or bench_final_loss or graphed or replays_as_a_hip_graph or projection or preprocessing or wino or trunk" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
# (4) batch-1 step again (split rule changed)
for amp in "" bfloat16; do python tools/shipped_step.py 1 eager 100 $amp 2>/dev/null | grep shipped_step; done > $O/b1.txt
tail -3 $O/bench8.err; head -30 $O/conv_step_breakdown.txt; tail -c 1500 $O/pytest.txt; cat $O/b1.txt
