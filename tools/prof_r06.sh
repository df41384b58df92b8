#!/bin/bash
# Round-6 profile bundle: bench.py under rocprofv3 (kernel stats + separate PMC passes, tools/prof_round.sh), kernel statistics of the
# secondary paths, the MPC CLOSED LOOP (300 warm steps: VERDICT r5 asked for a profiler row behind the in-kernel stamps) and the
# stall / LDS counters of the Gram kernel (tools/prof_stalls.sh).  Every rocprofv3 call is wrapped in `timeout`.  Afterwards, here:
#   KP_ROUND=r06 python3 tools/pmc_summarize.py; cp gpurun_out/r06_* profiles/
export KP_ROUND=r06
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_round.sh > $R/gpurun_out/prof_r06_round.log 2>&1
for p in "mpc tools/prof_mpc.py" "mpc_closed_loop tools/mpc_closed_loop_probe.py" "lasso_grid tools/lasso_grid_probe.py 64" "sweep tools/sweep_profile.py 1024" \
         "widths tools/gram_shapes_probe.py" "wide tools/wide_probe.py" "rankdef tools/arm_rankdef_latency.py" ${KP_PROF_EXTRA}; do
  set -- $p; name=$1; shift
  KP_PROF_LINES=3 bash $R/tools/prof_one.sh $name $R/"$@" > $R/gpurun_out/prof_r06_$name.log 2>&1
  cp $R/gpurun_out/prof_${name}_kernel_stats.csv $R/gpurun_out/r06_${name}_kernel_stats.csv 2>/dev/null
done
bash $R/tools/prof_stalls.sh > $R/gpurun_out/prof_r06_stalls.log 2>&1
cp $R/gpurun_out/prof_stalls/summary.json $R/gpurun_out/r06_gram_stall_counters.json 2>/dev/null
bash $R/tools/pmc_one.sh wide "kp_tn_gemm" $R/tools/wide_probe.py > $R/gpurun_out/r06_wide_pmc.txt 2>&1
ls -la $R/gpurun_out/r06_* 2>/dev/null | head -30
