#!/bin/bash
# lanes sweep for one tuning build
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for l in $LANES; do
  echo "=== lib '$V' lanes $l"
  LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu$V.so timeout -k 10 200 python tools/perf_explore.py --steps 3 --events 4096 --lanes $l > gpurun_out/ab_l$l.log 2>&1; rc=$?
  tail -n 1 gpurun_out/ab_l$l.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit $rc; fi
done
