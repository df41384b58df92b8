python -m pytest tests/test_gpu_sweep.py tests/test_gpu_multi.py -m gpu -x -q -W ignore::DeprecationWarning 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-mpc --no-one-caller --steps 20 > gpurun_out/r4_b_cols.json 2>gpurun_out/r4_b_cols.err
python - <<PY
import json
d=json.loads(open("gpurun_out/r4_b_cols.json").read().strip().splitlines()[-1])
print(d["rand_sweep"]["seconds"], d["value"])
for k in d["kernels"]:
    if "rand sweep" in k["point"]: print(k["point"], round(k["ms"],4), round(k["frac"],3), round(k["dense_equivalent_frac"],3), round(k.get("hbm_frac"),3))
PY
