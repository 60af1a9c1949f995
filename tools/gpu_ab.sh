#!/bin/bash
# A/B of tuning builds on the headline workload + parity suite on the default build
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log; tail -n 4 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi; return 0; }
step tests 1000 python -m pytest tests -m gpu -q --timeout=600 -x
for v in "" $VARIANTS; do
  LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu$v.so step ab$v 200 python tools/perf_explore.py --steps 4 --events 4096
done
