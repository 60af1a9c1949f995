#!/bin/bash
# PC-sampling attempt (beta): host_trap, time unit; bounded workload
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 180 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv -d gpurun_out/pcsamp -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --events-per-step 1024 > gpurun_out/pcsamp.log 2>&1
echo rc=$?
tail -n 15 gpurun_out/pcsamp.log
find gpurun_out/pcsamp -type f | head; du -sh gpurun_out/pcsamp
