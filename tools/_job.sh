cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout=800 -k "healpix_maps" --durations=8 -rs > gpurun_out/xy.log 2>&1; echo "xy rc=$?"; grep -E "s call|passed|failed|SKIP|AssertionError: \(|Error" gpurun_out/xy.log | cut -c1-300 | head -30
