#!/bin/bash
# A/B of tuning builds only (no tests)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in $VARIANTS; do
  [ "$v" = "default" ] && v=""
  echo "=== variant '$v'"
  LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu$v.so timeout -k 10 200 python tools/perf_explore.py --steps 4 --events 4096 > gpurun_out/ab$v.log 2>&1; rc=$?
  tail -n 2 gpurun_out/ab$v.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT: stopping"; exit $rc; fi
done
