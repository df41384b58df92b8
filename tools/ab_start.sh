for fc in 1 2 4 6 10; do
  echo "== first check $fc"; KP_LASSO_FIRST_CHECK=$fc timeout 300 python tools/lasso_grid_probe.py 64 2>&1 | grep "grid of" | tail -2
done
for ce in 5 20; do
  echo "== first check 2 every $ce"; KP_LASSO_CHECK=$ce KP_LASSO_FIRST_CHECK=2 timeout 300 python tools/lasso_grid_probe.py 64 2>&1 | grep "grid of" | tail -2
done
KP_LASSO_FIRST_CHECK=2 KP_LASSO_TRACE=1 timeout 300 python tools/lasso_grid_probe.py 64 2>&1 | grep -v "^grid\|nnz" | tail -40
