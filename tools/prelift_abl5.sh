#!/bin/bash
# timing-only ablations of kp_gram3_prelift_mfma_kernel (KP_PM_ABL bits: 1 no MFMA loop, 2 no lift, 4 no component stores, 8 no other entries)
# NOTE: the library must be built with -DKP_ABLATIONS (make CXXFLAGS+=-DKP_ABLATIONS): the shipped build ignores KP_PM_ABL.
for a in ${KP_ABL_LIST:-0 1 2 4 8 3 15 12}; do KP_PM_ABL=$a tools/prof_one.sh prelift$a $GRAFT_REPO_ROOT/tools/prelift_time.py > /dev/null 2>&1; python3 - $a <<'PY'
import csv,sys,os
f=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","prof_prelift%s_kernel_stats.csv"%sys.argv[1])
for r in csv.DictReader(open(f)):
    if "prelift_mfma" in r["Name"] or "kp_gram3_kernel" in r["Name"]: print("ABL", sys.argv[1], r["Name"][:40], r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3), "min %.1f"%(float(r["MinNs"])/1e3))
PY
done
