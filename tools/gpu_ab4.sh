#!/bin/bash
# variants x lanes sweep
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in $VARIANTS; do
  [ "$v" = "default" ] && v=""
  for l in $LANES; do
    echo "=== lib '$v' lanes $l"
    LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu$v.so timeout -k 10 200 python tools/perf_explore.py --steps 3 --events 4096 --lanes $l > gpurun_out/ab${v}_l$l.log 2>&1; rc=$?
    tail -n 2 gpurun_out/ab${v}_l$l.log | cut -c1-600
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit $rc; fi
  done
done
