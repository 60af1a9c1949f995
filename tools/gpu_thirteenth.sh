#!/bin/bash
# GPU batch 13: par%save_all_photons, then the whole parity suite and the default bench
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log; tail -n 40 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi; return 0; }
step tests_allph 400 python -m pytest tests -m gpu -q --timeout=300 -k "save_all_photons or velocity_fields"
step tests 1000 python -m pytest tests -m gpu -q --timeout=600 -x
step bench_default 600 python bench.py
