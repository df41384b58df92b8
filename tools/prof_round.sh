#!/bin/bash
# rocprofv3 passes for the round's profile bundle: kernel trace/stats, then PMC counters in
# separate runs (never combined with trace domains other than --kernel-trace).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_${KP_ROUND:-r03}
rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --no-cpu-baseline --no-mpc --no-extras --steps 50 --warmup 5"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O.trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O.fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O.write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > $O.mfma.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -- $B > $O.lds.log 2>&1
ls -R $O | head -40
timeout 600 rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|FETCH_SIZE|WRITE_SIZE" | head -30
