#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun, ROUND=r03): bench lines, kernel traces of the fp32 and the autocast
# step with per-step breakdowns, the geometry micro-benchmark with its PMC passes, and the convolution kernels: per-layer
# launch table + FETCH_SIZE / WRITE_SIZE per layer (tools/conv_layers.py), matrix-core counters of the Winograd and the
# half-precision kernels.  Counters are collected in their own runs, with --kernel-trace only.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=${ROUND:-r06}
out=gpurun_out/profiles_$R; rm -rf $out; mkdir -p $out
T=/tmp/prof_$R; rm -rf $T; mkdir -p $T
python bench.py > $out/bench.json 2> $out/bench.err; tail -1 $out/bench.json | cut -c1-300
python bench.py --amp bfloat16 --no-cpu-baseline --feed-steps 0 --long-steps 0 > $out/bench_bf16.json 2>/dev/null
python bench.py --graph --no-cpu-baseline --feed-steps 0 --long-steps 0 --autocast-steps 0 --kernel-reps 2 > $out/bench_graph.json 2>/dev/null
for m in f32 bf16; do
  fl=""; [ $m = bf16 ] && fl="--amp bfloat16"
  rocprofv3 --kernel-trace --stats --output-format csv -d $T/bench_$m -o bench -- python bench.py $fl --no-cpu-baseline --no-profile --no-live-pmc --long-steps 0 --feed-steps 0 --autocast-steps 0 --kernel-reps 2 > $out/bench_profiled_$m.json 2>/dev/null
  python tools/step_breakdown.py $(find $T/bench_$m -name "*kernel_trace.csv" | head -1) 20 200 > $out/step_breakdown_$m.txt 2>&1
  cp $(find $T/bench_$m -name "*kernel_stats.csv" | head -1) $out/bench_kernel_stats_$m.csv
done
# the reference's default operating point (64x720, stored lists): one steady-state step at batch 1 (fp32, bf16) and batch 8, eager
for v in "1 f32" "1 bf16" "8 f32"; do
  set -- $v; a=""; [ $2 = bf16 ] && a=bfloat16
  rocprofv3 --kernel-trace --output-format csv -d $T/ship_$1_$2 -o t -- python tools/shipped_step.py $1 eager 30 $a > $out/shipped_step_b$1_$2.txt 2>/dev/null
  python tools/step_breakdown.py $(find $T/ship_$1_$2 -name "*kernel_trace.csv" | head -1) 20 60 > $out/step_breakdown_64x720_b$1_$2.txt 2>&1
done
(for sp in 1 0; do DL_WINO_SPLIT=$sp python tools/shipped_step.py 1 eager 100 2>/dev/null | grep shipped_step; done; python tools/shipped_step.py 1 graph 100 2>/dev/null | grep shipped_step) > $out/shipped_step_b1_ab.txt
timeout 300 python tools/feed_ranks.py --out $out/feed_ranks.json > /dev/null 2>&1
# geometry kernels alone + their HBM counters
rocprofv3 --kernel-trace --stats --output-format csv -d $T/geo_trace -o geo -- python tools/geo_bench.py 30 0.4 > /dev/null 2>&1
cp $(find $T/geo_trace -name "*kernel_stats.csv" | head -1) $out/geometry_kernel_stats.csv
python tools/loss_calibration.py $(find $T/geo_trace -name "*kernel_trace.csv" | head -1) 30 > $out/loss_calibration.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/geo_fetch -o geo -- python tools/geo_bench.py 5 0.4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/geo_write -o geo -- python tools/geo_bench.py 5 0.4 > /dev/null 2>&1
mkdir -p $out/geo_fetch $out/geo_write $out/geo_fetch_shuffled $out/geo_write_shuffled
cp $(find $T/geo_fetch -name "*counter_collection.csv" | head -1) $out/geo_fetch/geo_counter_collection.csv
cp $(find $T/geo_write -name "*counter_collection.csv" | head -1) $out/geo_write/geo_counter_collection.csv
# (the same counters with the points of every scan randomly permuted: the input order of rounds 1-4)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/geo_fetch_s -o geo -- python tools/geo_bench.py 5 0.4 shuffled > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/geo_write_s -o geo -- python tools/geo_bench.py 5 0.4 shuffled > /dev/null 2>&1
cp $(find $T/geo_fetch_s -name "*counter_collection.csv" | head -1) $out/geo_fetch_shuffled/geo_counter_collection.csv
cp $(find $T/geo_write_s -name "*counter_collection.csv" | head -1) $out/geo_write_shuffled/geo_counter_collection.csv
python tools/loss_warm.py 50 > $out/loss_warm.txt 2>/dev/null
python tools/loss_cold.py 2>/dev/null | grep '^B=' > $out/loss_cold.txt
# convolution kernels: stand-alone harnesses (host-checked correctness + per-layer times)
(tools/bin/conv_harness all 10; tools/bin/conv_harness wino 10; tools/bin/wino_wgrad check; tools/bin/wino_wgrad time 10) > $out/conv_harness.txt 2>&1
(tools/bin/convh_harness check; tools/bin/convh_harness time 10; tools/bin/convh_harness check f16 | tail -3; tools/bin/hw_probe | tail -3) > $out/convh_harness.txt 2>&1
# per-layer launch table and HBM traffic of every convolution of a step (fp32 and bf16)
for m in float32 bfloat16; do
  python tools/conv_layers.py $m $out/conv_layers_$m.json > $out/conv_layers_$m.txt 2>/dev/null
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $T/cl_fetch_$m -o c -- python tools/conv_layers.py $m > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $T/cl_write_$m -o c -- python tools/conv_layers.py $m > /dev/null 2>&1
  python tools/conv_layers_pmc.py $out/conv_layers_$m.json $(find $T/cl_fetch_$m -name "*counter_collection.csv" | head -1) $(find $T/cl_write_$m -name "*counter_collection.csv" | head -1) > $out/conv_hbm_pmc_$m.json 2> $out/conv_hbm_$m.err
done
# matrix-core counters: fp32 Winograd kernels (conv_harness wino + wino_wgrad), half-precision kernels (convh_harness time)
PMC="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $T/conv_pmc -o conv -- tools/bin/conv_harness wino 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $T/ww_pmc -o conv -- tools/bin/wino_wgrad time 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $T/convh_pmc -o conv -- tools/bin/convh_harness time 2 > /dev/null 2>&1
(python tools/conv_pmc.py $(find $T/conv_pmc -name "*counter_collection.csv" | head -1); python tools/conv_pmc.py $(find $T/ww_pmc -name "*counter_collection.csv" | head -1) k_wino_wgrad) > $out/conv_pmc.txt 2>&1
python tools/convh_pmc.py $(find $T/convh_pmc -name "*counter_collection.csv" | head -1) > $out/convh_pmc.txt 2>&1
(tools/bin/scatter_probe 20; echo "-- points in raster order"; tools/bin/scatter_probe 20 1) > $out/scatter_probe.txt 2>&1
# round 6: the Winograd kernels alone (check, per-layer times, phase stamps, counters) and what issues in the shadow of an fp32 MFMA
(tools/bin/wino_lab check; tools/bin/wino_lab time 30; tools/bin/wino_lab phases) > $out/wino_lab.txt 2>&1
tools/lab_pmc.sh > /dev/null 2>&1; cp gpurun_out/lab_pmc.txt $out/wino_lab_pmc.txt
(timeout 120 tools/bin/mfma_overlap 512; timeout 60 tools/bin/mfma_dma; timeout 60 tools/bin/pk_probe) > $out/mfma_shadow.txt 2>&1
# round 6: the exact search alone (check against the exhaustive kernel, times per regime, per-kernel times and counters), list sizes on the bench batch
bash tools/nn_lab_run.sh tools/bin/nn_lab nn_lab_bundle > /dev/null 2>&1; cp gpurun_out/nn_lab_bundle.txt $out/nn_lab_final.txt
python tools/nn_counts.py 2>/dev/null | grep '^shift' > $out/nn_counts.txt
du -sh $out; ls $out
