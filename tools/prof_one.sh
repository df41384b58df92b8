#!/bin/bash
# rocprofv3 kernel statistics of ONE probe script: tools/prof_one.sh <name> <script> [args]; summary -> gpurun_out/prof_<name>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; name=$1; shift
O=$R/gpurun_out/prof_$name; rm -rf $O; mkdir -p $O
timeout ${KP_PROF_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 "$@" > $O.log 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $R/gpurun_out/prof_${name}_kernel_stats.csv && head -${KP_PROF_LINES:-25} $f | cut -c1-220
tail -5 $O.log
