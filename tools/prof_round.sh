#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun): kernel trace of the bench command, kernel trace of the
# geometry micro-benchmark, and the two PMC passes (FETCH_SIZE, WRITE_SIZE) for the geometry kernels.
# Counters are collected in their own runs, with --kernel-trace only (never together with sys/hip/hsa tracing).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${ROUND:-r01}
out=gpurun_out/profiles_$R; rm -rf $out; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_trace -o bench -- python bench.py --no-cpu-baseline > $out/bench_profiled.json 2>/dev/null
python tools/step_breakdown.py $(find $out/bench_trace -name "*kernel_trace.csv" | head -1) 20 45 > $out/step_breakdown.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/geo_trace -o geo -- python tools/geo_bench.py 30 0.4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/geo_fetch -o geo -- python tools/geo_bench.py 5 0.4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/geo_write -o geo -- python tools/geo_bench.py 5 0.4 > /dev/null 2>&1
python tools/loss_warm.py 50 > $out/loss_warm.txt 2>/dev/null
PYTHONPATH=. python tools/ring_bench.py 20 > $out/ring_bench.txt 2>/dev/null
# convolution kernels: host-checked correctness, per-layer times, sustained MFMA peak, Winograd; matrix-core counters
(tools/bin/conv_harness peak; tools/bin/conv_harness all 10; tools/bin/conv_harness wino 10) > $out/conv_harness.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/conv_pmc -o conv -- tools/bin/conv_harness wino 2 > /dev/null 2>&1
python tools/exp/conv_pmc.py $(find $out/conv_pmc -name "*counter_collection.csv" | head -1) > $out/conv_pmc.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/conv_fetch -o conv -- tools/bin/conv_harness wino 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/conv_write -o conv -- tools/bin/conv_harness wino 2 > /dev/null 2>&1
python tools/exp/conv_hbm.py $(find $out/conv_fetch -name "*counter_collection.csv" | head -1) $(find $out/conv_write -name "*counter_collection.csv" | head -1) > $out/conv_hbm_pmc.json 2>$out/conv_hbm.err
(tools/bin/scatter_probe 20; echo "-- points in raster order"; tools/bin/scatter_probe 20 1) > $out/scatter_probe.txt 2>&1
python tools/miopen_layers.py 10 2>/dev/null | grep miopen > $out/miopen_layers.txt
find $out -name "*.csv" | head -30
# keep only summaries (the raw traces are large)
find $out -name "*kernel_trace.csv" -size +20M -delete
du -sh $out
