#!/bin/bash
# rocprofv3 kernel stats of the lasso grid (BASELINE configs[3] shape); summary copied to profiles/ by hand
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_lasso -- python3 $R/tools/lasso_probe.py ${1:-64} > $R/gpurun_out/prof_lasso.log 2>&1
find $R/gpurun_out/prof_lasso -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/lasso_kernel_stats.csv
