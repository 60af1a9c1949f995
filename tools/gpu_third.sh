#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log; tail -n 8 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi; return 0; }
step tests 900 python -m pytest tests -m gpu -q --timeout=600
for w in _w2 _w4; do
  LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu$w.so step occ$w 200 python tools/perf_explore.py --steps 6 --events 8192 --lanes 524288
done
step counters_list 120 rocprofv3 -L
step prof_stats 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --steps 3 --warmup 1 --events-per-step 2048 --no-cpu-baseline
step prof_pmc_sq 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof_pmc_sq -- python3 bench.py --steps 2 --warmup 1 --events-per-step 2048 --no-cpu-baseline
step prof_pmc_fetch 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --events-per-step 2048 --no-cpu-baseline
step prof_pmc_write 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_pmc_write -- python3 bench.py --steps 2 --warmup 1 --events-per-step 2048 --no-cpu-baseline
find gpurun_out -name "*.csv" | head -30
