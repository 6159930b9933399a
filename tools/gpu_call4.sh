#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/c4; mkdir -p $O
python tools/convergence.py --epochs 30 --precisions float32 --out $O/conv_ram.json 2>&1 | grep -v "^Index\|Dataset will" > $O/conv_ram.txt
python tools/convergence.py --epochs 30 --precisions float32 --on-disk --out $O/conv_disk.json 2>&1 | grep -v "^Index\|Dataset will" > $O/conv_disk.txt
python tools/convergence.py --epochs 30 --precisions float32 --on-disk --workers 2 --out $O/conv_disk2.json 2>&1 | grep -v "^Index\|Dataset will" > $O/conv_disk2.txt
timeout 600 python -m pytest tests -m gpu -q --timeout 500 -k "eight_ranks or two_ranks_on_one_gpu" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
grep "hip_graph auto\|packed feed\|float32 steps" $O/conv_*.txt; tail -4 $O/pytest.txt; tail -c 600 $O/bench.json
