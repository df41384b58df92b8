#!/bin/bash
# rocprofv3 kernel stats of the batched random-system sweep (BASELINE configs[4] shape)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sweep -- python3 $R/tools/sweep_profile.py ${1:-1024} > $R/gpurun_out/prof_sweep.log 2>&1
find $R/gpurun_out/prof_sweep -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/sweep_kernel_stats.csv
