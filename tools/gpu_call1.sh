#!/bin/bash
# round 5, first GPU call: new tests, projection probe, first look at the default operating point and the convergence run
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/c1
(tools/bin/scatter_probe 20; echo "-- points in raster order"; tools/bin/scatter_probe 20 1) > gpurun_out/c1/scatter_probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/c1/pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/c1/pytest.txt
cp gpurun_out/parity_measured.json gpurun_out/c1/ 2>/dev/null
timeout 400 python tools/convergence.py --epochs 60 --out gpurun_out/c1/convergence.json > gpurun_out/c1/convergence.txt 2>&1
timeout 600 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
tail -c 1500 gpurun_out/c1/pytest.txt; tail -5 gpurun_out/c1/convergence.txt; cat gpurun_out/c1/scatter_probe.txt
