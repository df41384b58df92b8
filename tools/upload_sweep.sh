cd $GRAFT_REPO_ROOT
nproc
timeout 600 python -m pytest tests/test_gpu_fit.py -x -q -k "refilled or async_fit or config2" 2>&1 | tail -5
for T in 1 2 4 8; do echo "== threads $T"; KP_COPY_THREADS=$T timeout 120 python tools/upload_probe.py 2>&1 | tail -6; done
echo "== chunk 128"; KP_COPY_CHUNK_KB=128 timeout 120 python tools/upload_probe.py 2>&1 | tail -6
echo "== chunk 2048"; KP_COPY_CHUNK_KB=2048 timeout 120 python tools/upload_probe.py 2>&1 | tail -6
