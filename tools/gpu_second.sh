#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log; tail -n 12 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi; return 0; }
step tests 900 python -m pytest tests -m gpu -q --timeout=600
step age_profile 400 python tools/perf_explore.py --steps 160 --events 8192 --every 8
step lanes65k 200 python tools/perf_explore.py --steps 6 --events 8192 --lanes 65536
step lanes131k 200 python tools/perf_explore.py --steps 6 --events 8192 --lanes 131072
step lanes524k 200 python tools/perf_explore.py --steps 6 --events 8192 --lanes 524288
