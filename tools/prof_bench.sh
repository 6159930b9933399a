#!/bin/bash
# rocprofv3 kernel trace of the default bench command (run on the GPU box through gpurun); summaries land in gpurun_out/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o bench -- python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline ${EXTRA} > gpurun_out/bench_prof.log 2>&1
tail -2 gpurun_out/bench_prof.log
find gpurun_out/prof -name "*stats*" | head
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
head -40 "$f"
