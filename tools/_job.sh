cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=800 -k "peel_cubes" --durations=25 > gpurun_out/xy.log 2>&1; echo "xy rc=$?"; grep -E "s call|passed|failed" gpurun_out/xy.log | cut -c1-200 | head -30
