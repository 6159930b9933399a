#!/bin/bash
# kernel-level breakdown of the timed steps (rocprofv3 kernel trace): usage  MODE="--amp bfloat16" bash tools/exp/pb_ab.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-f32}
rm -rf /tmp/pb_$tag; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$tag -o bench -- python bench.py $MODE --no-cpu-baseline --no-profile --no-live-pmc --long-steps 0 --feed-steps 0 --autocast-steps 0 --disk-pairs 0 --variant-steps 0 --kernel-reps 2 > /dev/null 2>&1
python tools/step_breakdown.py $(find /tmp/pb_$tag -name "*kernel_trace.csv" | head -1) 20 200 > gpurun_out/sb_$tag.txt 2>&1
head -${LINES_OUT:-40} gpurun_out/sb_$tag.txt
