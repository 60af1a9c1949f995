#!/bin/bash
# first contact with the GPU: smoke, parity tests, a short bench.  Each step under its own timeout; a step that
# times out (124/137) stops the sequence.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
step() { # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi
  return 0
}
rocminfo | grep -m3 -E "gfx|Compute Unit" > gpurun_out/rocminfo.txt 2>&1
step smoke 300 python __graft_entry__.py smoke
step tests 900 python -m pytest tests -m gpu -q -x --timeout=600
step bench_short 400 python bench.py --steps 2 --warmup 1 --events-per-step 512 --no-cpu-baseline
