R=${GRAFT_REPO_ROOT:-/root/repo}
tag=round5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -o p -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration > $R/gpurun_out/${tag}_kt.log 2>&1
python $R/tools/rocpd_stats.py /tmp/kt_$tag/p_results.db $R/gpurun_out/${tag}_kernel_stats.csv
python $R/tools/timeline.py /tmp/kt_$tag/p_results.db > $R/gpurun_out/${tag}_step_timeline.txt 2>&1
A="--steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 --no-calibration"
rocprofv3 --pmc FETCH_SIZE -d /tmp/pf_$tag -o p -- python $R/bench.py $A > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pf_$tag/p_results.db conv3x3_mx > $R/gpurun_out/${tag}_fetch.txt
rocprofv3 --pmc WRITE_SIZE -d /tmp/pw_$tag -o p -- python $R/bench.py $A > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/pw_$tag/p_results.db conv3x3_mx > $R/gpurun_out/${tag}_write.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d /tmp/ps_$tag -o p -- python $R/bench.py $A > /dev/null 2>&1
python $R/tools/pmc_summary.py /tmp/ps_$tag/p_results.db conv3x3_ > $R/gpurun_out/${tag}_sq.txt
python $R/tools/make_traffic_json.py $R/gpurun_out/${tag}_fetch.txt $R/gpurun_out/${tag}_write.txt $R/gpurun_out/${tag}_traffic.json "conv3x3_mx_kernel<bf16>"
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --no-other-configs --traffic off --sustain-s 0 2>/dev/null | tail -1 > $R/gpurun_out/${tag}_recollect_box.json
