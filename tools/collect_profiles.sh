#!/bin/bash
# Round profile collection on the GPU box: bench line, kernel-trace stats of the same command, PMC traffic passes.
# usage: tools/collect_profiles.sh <tag> [kernel substring]   -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,fetch.txt,write.txt}
tag=${1:-round3}
kern=${2:-conv3x3_mx}          # dominant kernel (substring of the symbol) for the traffic passes
R=/root/repo
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > $R/gpurun_out/${tag}_kt.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kt_$tag/p_results.db $R/gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pf_$tag/p_results.db $kern > $R/gpurun_out/${tag}_fetch.txt
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pw_$tag/p_results.db $kern > $R/gpurun_out/${tag}_write.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d /tmp/ps_$tag -o p -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/ps_$tag/p_results.db conv3x3_ > $R/gpurun_out/${tag}_sq.txt
python $R/tools/timeline.py /tmp/kt_$tag/p_results.db > $R/gpurun_out/${tag}_step_timeline.txt 2>&1
python $R/tools/per_shape.py > $R/gpurun_out/${tag}_per_shape.txt 2>&1
python $R/tools/make_traffic_json.py $R/gpurun_out/${tag}_fetch.txt $R/gpurun_out/${tag}_write.txt $R/gpurun_out/${tag}_traffic.json "conv3x3_mx_kernel<bf16>" > /dev/null 2>&1
