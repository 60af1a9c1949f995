#!/bin/bash
# One parameterised GPU batch script (run through gpurun from the repo root):  tools/gpu_job.sh STEP [STEP ...]
# Every step runs under its own `timeout -k 10`, logs to gpurun_out/<step>.log and a step that times out (124 / 137) ends
# the batch: no further GPU step is started after a kill.
#
#   smoke                      __graft_entry__.smoke()
#   avail                      rocprofv3 --list-avail (counter names of this GPU) -> gpurun_out/avail.log
#   tests[:EXPR]               pytest -m gpu (optionally -k EXPR)
#   slow:EXPR                  pytest -m "gpu and slow" -k EXPR with a heartbeat line per minute (long silent set-ups: the 74e6-clump placement)
#   pytest:FILES               pytest -m gpu on the given files (commas for spaces)
#   bench[:ARGS]               python bench.py ARGS            (ARGS with commas for spaces: bench:--steps,2,--no-cpu-baseline)
#   configs                    tools/perf_configs.py (events/s of the other BASELINE configs)
#   perf:C1,C2[:ARGS]          tools/perf_kernel.py --config C (steady-state launches of one kernel variant + its section-8(d) roofline)
#   prof:TAG[:ARGS]            rocprofv3 of `python3 bench.py --no-cpu-baseline --completion-photons 0 ARGS`: kernel-trace --stats,
#                              then SEPARATE --pmc passes (SQ set, lane set, FETCH_SIZE, WRITE_SIZE) -> gpurun_out/TAG_* summaries
#   profcfg:TAG:CONFIG         the same passes on tools/perf_kernel.py --config CONFIG (c3, c5, c1, c4: the other transport kernels)
#   profx:TAG:ARGS             the same passes on tools/perf_explore.py ARGS (e.g. a lone packet: --photons,1,--lanes,64,--block,64,--steps,6,--events,8192)
#   stats:TAG:CONFIG           only the kernel-trace --stats pass on tools/perf_kernel.py --config CONFIG
#   py:SCRIPT[:ARGS]           python SCRIPT ARGS (debug helpers under tools/dbg)
#   explore[:ARGS]             tools/perf_explore.py ARGS (throughput vs photon age, drain tail)
#   perflib:SUFFIXES:CONFIGS   tools/perf_kernel.py --config C with LART_GPU_LIB = liblart_gpu<SUFFIX>.so for every pair (SUFFIXES like default,_t6,_prof)
#   envperf:VAR=VALUE:CONFIGS  tools/perf_kernel.py --config C with one environment variable set (timing knobs of the library)
#   ab:VARIANTS:LANES          tuning builds x pool sizes through tools/perf_explore.py (VARIANTS like default,_w3 ; LANES like 0,786432)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { local name=$1 to=$2; shift 2
  echo "=== $name: $*" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "=== $name rc=$rc" | tee -a gpurun_out/steps.log; tail -n 12 "gpurun_out/$name.log" | cut -c1-1500
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping"; exit $rc; fi; return 0; }
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_ANY"
SQ2="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
SQ3="SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_FLAT SQ_INSTS_SMEM"
profile() { # tag kernel-filter events-per-step command...
  local tag=$1 filt=$2 eps=$3; shift 3
  step ${tag}_stats 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- "$@"
  step ${tag}_pmc_sq 900 rocprofv3 --pmc $SQ1 --output-format csv -d gpurun_out/${tag}_pmc_sq -- "$@"
  step ${tag}_pmc_lane 900 rocprofv3 --pmc $SQ2 --output-format csv -d gpurun_out/${tag}_pmc_lane -- "$@"
  step ${tag}_pmc_mem 900 rocprofv3 --pmc $SQ3 --output-format csv -d gpurun_out/${tag}_pmc_mem -- "$@"
  step ${tag}_pmc_fetch 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -- "$@"
  step ${tag}_pmc_write 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -- "$@"
  step ${tag}_summary 120 python3 tools/pmc_summary.py gpurun_out/${tag}_pmc_summary.json $eps "$*" gpurun_out/${tag}_pmc_sq gpurun_out/${tag}_pmc_lane gpurun_out/${tag}_pmc_mem gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
  local f; f=$(find gpurun_out/${tag}_stats -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
  rm -rf gpurun_out/${tag}_stats gpurun_out/${tag}_pmc_sq gpurun_out/${tag}_pmc_lane gpurun_out/${tag}_pmc_mem gpurun_out/${tag}_pmc_mem gpurun_out/${tag}_pmc_fetch gpurun_out/${tag}_pmc_write
}
for spec in "$@"; do
  IFS=: read -r what a1 a2 <<< "$spec"
  case $what in
    smoke)   step smoke 400 python __graft_entry__.py smoke ;;
    avail)   step avail 120 rocprofv3 --list-avail ;;
    tests)   if [ -n "$a1" ]; then step tests 1100 python -m pytest tests -m gpu -q --timeout=900 --durations=40 -k "$a1"; else step tests 1100 python -m pytest tests -m gpu -q --timeout=900 --durations=40; fi ;;
    slow)    ( while true; do sleep 60; echo "heartbeat $(date +%T)"; done ) & hb=$!
             LART_RUN_SLOW=1 step slow 1150 python -m pytest tests -m "gpu and slow" -q -rP --timeout=1100 -k "$a1"; kill $hb 2>/dev/null ;;
    pytest)  step pytest_$(echo "$a1" | tr -c 'A-Za-z0-9\n' '_' | cut -c1-40) 1100 python -m pytest ${a1//,/ } -m gpu -q -rP --timeout=900 --durations=15 ;;
    bench)   step bench 900 python bench.py ${a1//,/ } ;;
    configs) step configs 900 python tools/perf_configs.py ;;
    perf)    for c in ${a1//,/ }; do step perf_$c 400 python tools/perf_kernel.py --config $c ${a2//,/ }; done ;;
    prof)    profile "$a1" transport_kernel 4096 python3 bench.py --no-cpu-baseline --completion-photons 0 --no-other-configs ${a2//,/ } ;;
    profcfg) profile "$a1" transport_kernel 4096 python3 tools/perf_kernel.py --config "$a2" ;;
    profx)   profile "$a1" transport_kernel 8192 python3 tools/perf_explore.py ${a2//,/ } ;;
    stats)   step ${a1}_stats 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${a1}_stats -- python3 tools/perf_kernel.py --config "$a2"
             f=$(find gpurun_out/${a1}_stats -name "*kernel_stats.csv" | head -n 1); [ -n "$f" ] && cp "$f" gpurun_out/${a1}_kernel_stats.csv && head -n 8 gpurun_out/${a1}_kernel_stats.csv | cut -c1-300; rm -rf gpurun_out/${a1}_stats ;;
    py)      step py_$(basename ${a1%.py}) 600 python ${a1} ${a2//,/ } ;;
    explore) step explore 1100 python tools/perf_explore.py ${a1//,/ } ;;
    perflib) for v in ${a1//,/ }; do [ "$v" = default ] && v=""; for c in ${a2//,/ }; do
               LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu$v.so step "perf${v}_$c" 400 python tools/perf_kernel.py --config $c; done; done ;;
    envperf) for c in ${a2//,/ }; do export "$a1"; step "perf_${a1%%=*}_$c" 400 python tools/perf_kernel.py --config $c; unset "${a1%%=*}"; done ;;
    ab)      for v in ${a1//,/ }; do [ "$v" = default ] && v=""; for l in ${a2//,/ }; do
               LART_GPU_LIB=$PWD/lart_amd/csrc/liblart_gpu$v.so step "ab${v}_l$l" 300 python tools/perf_explore.py --steps 3 --events 4096 --lanes $l; done; done ;;
    *) echo "unknown step $spec"; exit 2 ;;
  esac
done
