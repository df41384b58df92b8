#!/bin/bash
# rocprofv3 counter passes of ONE probe script (each counter set its own run, --kernel-trace only): tools/pmc_one.sh <name> <kernel substring> <script> [args]
# prints the per-launch mean of every counter for the kernels whose name contains the substring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; name=$1; kern=$2; shift; shift
O=$R/gpurun_out/pmc_$name; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout ${KP_PROF_TIMEOUT:-300} rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 "$@" > $O/p$i.log 2>&1
done
python3 - $O "$kern" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]: acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    print(k)
    for c,v in sorted(cs.items()): print("   %-32s %.4g  (n=%d)"%(c,sum(v)/len(v),len(v)))
PY
