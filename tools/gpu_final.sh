#!/bin/bash
# final GPU batch of the round: whole parity suite, other-config rates, default bench, kernel-trace + separate PMC passes -> r01_v6
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log; tail -n 14 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi; return 0; }
step tests 1000 python -m pytest tests -m gpu -q --timeout=600 -x
step configs 900 python tools/perf_configs.py
step bench_default 600 python bench.py
CMD="python3 bench.py --no-cpu-baseline"
step prof_stats 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof10_stats -- $CMD
step prof_pmc_sq 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof10_pmc_sq -- $CMD
step prof_pmc_fetch 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof10_pmc_fetch -- $CMD
step prof_pmc_write 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof10_pmc_write -- $CMD
step pmc_summary 60 python3 tools/pmc_summary.py gpurun_out/r01_v6_pmc_summary.json 4096 "$CMD" gpurun_out/prof10_pmc_sq gpurun_out/prof10_pmc_fetch gpurun_out/prof10_pmc_write
find gpurun_out/prof10_stats -name "*stats*.csv" | head
