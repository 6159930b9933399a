#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun): kernel trace of the bench command, kernel trace of the
# geometry micro-benchmark, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) for the geometry kernels.
# Counters are collected in their own runs, with --kernel-trace only (never together with sys/hip/hsa tracing).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${ROUND:-r01}
out=gpurun_out/profiles_$R; rm -rf $out; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -o bench -- python bench.py --no-cpu-baseline > $out/bench_profiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/geo_trace -o geo -- python tools/geo_bench.py 30 0.4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/geo_fetch -o geo -- python tools/geo_bench.py 5 0.4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/geo_write -o geo -- python tools/geo_bench.py 5 0.4 > /dev/null 2>&1
python tools/loss_warm.py 50 > $out/loss_warm.txt 2>/dev/null
PYTHONPATH=. python tools/ring_bench.py 20 > $out/ring_bench.txt 2>/dev/null
find $out -name "*.csv" | head -30
# keep only summaries (the raw traces are large)
find $out -name "*kernel_trace.csv" -size +20M -delete
du -sh $out
