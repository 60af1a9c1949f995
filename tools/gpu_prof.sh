#!/bin/bash
# section-clock profile of the transport kernel (tuning build -DLART_PROFILE)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu_prof.so timeout -k 10 300 python tools/perf_explore.py --steps 4 --events 4096 > gpurun_out/section_prof.log 2>&1
echo rc=$?; tail -n 8 gpurun_out/section_prof.log
