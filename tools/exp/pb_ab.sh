cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 1 0; do
rm -rf /tmp/pb$v; DELORA_WGRAD_BATCHED=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb$v -o bench -- python bench.py --no-cpu-baseline --no-profile --no-live-pmc --long-steps 0 --feed-steps 0 --autocast-steps 0 --disk-pairs 0 --variant-steps 0 --kernel-reps 2 > /dev/null 2>&1
python tools/step_breakdown.py $(find /tmp/pb$v -name "*kernel_trace.csv" | head -1) 20 200 > gpurun_out/sb_f32_batched$v.txt 2>&1
done
head -24 gpurun_out/sb_f32_batched1.txt
