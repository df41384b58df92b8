#!/bin/bash
# rocprofv3 kernel statistics of the other fit shapes (bench.bench_width_points) after the prelift kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_extra
rm -rf $O/widths; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/widths -- python3 $R/tools/gram_shapes_probe.py > $O/widths.log 2>&1
f=$(find $O/widths -name "*kernel_stats.csv" | head -1); cp $f $O/widths_kernel_stats.csv
rm -rf $O/lasso_ill; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lasso_ill -- python3 $R/tools/lasso_illcond_probe.py 2 1 bilinear > $O/lasso_ill.log 2>&1
f=$(find $O/lasso_ill -name "*kernel_stats.csv" | head -1); cp $f $O/lasso_ill_kernel_stats.csv
head -12 $O/widths_kernel_stats.csv | cut -c1-150
head -8 $O/lasso_ill_kernel_stats.csv | cut -c1-150
