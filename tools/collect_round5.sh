#!/bin/bash
# Round-5 profile set, ONE gpurun call (one box): the round-4 set again with the final library + the round's new artefacts.
# -> gpurun_out/round5_*  (copied to profiles/ by hand after a look)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
tools/collect_profiles.sh round5 conv3x3_mx
tools/gan_kt.sh; cp gpurun_out/gan_kernel_stats.csv gpurun_out/round5_config4_kernel_stats.csv
python tools/gan_graph_times.py > gpurun_out/round5_config4_graph_times.txt 2>/dev/null
export VQK_BENCH_SELF_LAUNCH=1 VQK_FORCE_DIST=1
A="--gpus 1 --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline"
python bench.py $A 2>/dev/null | tail -1 > gpurun_out/round5_bench_dist1.json
VQK_OVERLAP_ALLREDUCE=1 python bench.py $A 2>/dev/null | tail -1 > gpurun_out/round5_bench_dist1_overlap.json
python bench.py $A --quantizer ema --codebook 1024 2>/dev/null | tail -1 > gpurun_out/round5_bench_dist1_ema.json
python bench.py $A --gan --batch 16 2>/dev/null | tail -1 > gpurun_out/round5_bench_dist1_gan.json
unset VQK_BENCH_SELF_LAUNCH VQK_FORCE_DIST
tools/trace_step.sh round5
