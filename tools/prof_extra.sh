#!/bin/bash
# rocprofv3 kernel statistics of the secondary paths (MPC step, lasso grid, batched random-system sweep)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_extra
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/mpc -- python3 $R/tools/prof_mpc.py > $O/mpc.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lasso -- python3 $R/tools/lasso_probe.py 64 > $O/lasso.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sweep -- python3 $R/tools/sweep_profile.py 1024 > $O/sweep.log 2>&1
# the other fit shapes of SURVEY 8(d) and the fourier / gaussian dictionaries (bench.bench_width_points)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/widths -- python3 $R/tools/gram_shapes_probe.py > $O/widths.log 2>&1
# the 64-value lasso grid of bench.py (configs[3])
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lasso_grid -- python3 $R/tools/lasso_grid_probe.py 64 > $O/lasso_grid.log 2>&1
for d in mpc lasso sweep widths lasso_grid; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/${d}_kernel_stats.csv; done
ls -la $O/*.csv
