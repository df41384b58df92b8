#!/bin/bash
# Round-5 profile bundle: bench.py under rocprofv3 (kernel stats + separate PMC passes, tools/prof_round.sh), then kernel statistics of the
# secondary paths and of the round's new kernels (wide dictionaries, rank-revealing solve, econ lift on the matrix pipe) + PMC of the
# new prelift kernel and its consumer.  Every rocprofv3 call is wrapped in `timeout`.
export KP_ROUND=r05
R=$GRAFT_REPO_ROOT
bash $R/tools/prof_round.sh > $R/gpurun_out/prof_r05_round.log 2>&1
cd $R && python3 tools/pmc_summarize.py > gpurun_out/prof_r05_summarize.log 2>&1
for p in "mpc tools/prof_mpc.py" "lasso_grid tools/lasso_grid_probe.py 64" "sweep tools/sweep_profile.py 1024" "widths tools/gram_shapes_probe.py" \
         "wide tools/wide_probe.py" "rankdef tools/arm_rankdef_latency.py" "lasso_ill tools/lasso_illcond_probe.py 2 1 bilinear"; do
  set -- $p; name=$1; shift
  KP_PROF_LINES=3 bash $R/tools/prof_one.sh $name $R/"$@" > $R/gpurun_out/prof_r05_$name.log 2>&1
  cp $R/gpurun_out/prof_${name}_kernel_stats.csv $R/gpurun_out/r05_${name}_kernel_stats.csv 2>/dev/null
done
bash $R/tools/pmc_one.sh prelift "kp_gram3" $R/tools/prelift_time.py > $R/gpurun_out/r05_prelift_pmc.txt 2>&1
bash $R/tools/pmc_one.sh wide "kp_tn_gemm" $R/tools/wide_probe.py > $R/gpurun_out/r05_wide_pmc.txt 2>&1
ls -la $R/gpurun_out/r05_* $R/profiles/r05_* 2>/dev/null | head -30
tail -3 $R/gpurun_out/prof_r05_summarize.log
