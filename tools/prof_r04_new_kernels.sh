#!/bin/bash
# PMC counters of the round-4 kernels that are not bound by the matrix pipe: the lasso homotopy (kp_lasso_path_kernel, arm data W = 92) and
# the prelift kernels (W = 136 with pcs, gaussian-20), one counter set per run, each under its own time limit.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r04_new
rm -rf $O; mkdir -p $O
for prog in "lasso tools/lasso_illcond_probe.py 2 1 bilinear" "prelift tools/prelift_time.py" "prelift_ext tools/prelift_ext_time.py"; do
  set -- $prog; name=$1; shift
  for pass in "valu SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES" "lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY" "busy SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
    set -- $pass; pn=$1; shift
    timeout 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/${name}_$pn -- python3 $R/$(echo $prog | cut -d' ' -f2-) > $O/${name}_$pn.log 2>&1
    echo "$name $pn rc $?"
  done
done
