timeout 300 python tools/lasso_grid_probe.py 64 2>&1 | grep "grid of\|active" | tail -4
KP_LASSO_POWER_IT=1 timeout 300 python tools/lasso_grid_probe.py 64 2>&1 | grep "grid of\|active" | tail -2
timeout 300 python tools/lasso_probe.py 8 2>&1 | grep batch
timeout 900 python -m pytest tests/test_gpu_lasso.py -x -q -m gpu 2>&1 | tail -3
