#!/bin/bash
# run every tools/bin/lab_* variant (ablation / experiment builds of wino_lab): check + phases of the product build, times of all
mkdir -p gpurun_out
out=gpurun_out/${1:-lab}.txt
{
  if [ "$2" != "nocheck" ]; then echo "== wino_lab check"; timeout 120 tools/bin/wino_lab check; fi
  echo "== wino_lab time"; timeout 120 tools/bin/wino_lab time 20
  if [ "$2" == "phases" ]; then echo "== wino_lab phases"; timeout 120 tools/bin/wino_lab phases; fi
  for b in tools/bin/lab_*; do echo "== $b"; timeout 120 $b time 20; done
} > $out 2>&1
tail -3 $out
