#!/bin/bash
# Round-4 profile bundle (run on the GPU box through gpurun): rocprofv3 kernel statistics and PMC counters, every counter set in
# its own run with --kernel-trace only.  Condensed afterwards by tools/pmc_summarize.py (KP_ROUND=r04) and tools/sweep_pmc_r04.py.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export KP_ROUND=r04
bash $R/tools/prof_round.sh > $R/gpurun_out/prof_r04_round.log 2>&1
O=$R/gpurun_out/prof_r04_sweep
rm -rf $O; mkdir -p $O
S="python3 $R/tools/sweep_profile.py 1024"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $S > $O/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_valu -- $S > $O/valu.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -- $S > $O/lds.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $S > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $S > $O/write.log 2>&1
bash $R/tools/prof_extra.sh > $R/gpurun_out/prof_r04_extra.log 2>&1
ls $R/gpurun_out/prof_r04 $O $R/gpurun_out/prof_extra | head -40
