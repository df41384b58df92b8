#!/bin/bash
# The PMC passes of the sweep profile alone (each under its own time limit: a counter pass once sat for 40 minutes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04_sweep
mkdir -p $O
S="python3 $R/tools/sweep_profile.py 1024"
for pass in "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY" "fetch FETCH_SIZE" "write WRITE_SIZE" "valu SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  set -- $pass; name=$1; shift
  rm -rf $O/pmc_$name
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$name -- $S > $O/$name.log 2>&1
  echo "$name rc $?"
done
