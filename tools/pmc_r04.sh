#!/bin/bash
# HBM-traffic counters (round 4): FETCH_SIZE and WRITE_SIZE in SEPARATE passes, an SQ pass for MFMA utilisation; the
# vanilla step, the Instant-NGP STEADY-STATE loop and the Part 4 engine loop.  The program comes directly after `--`;
# every pass has its own time limit and a failed pass ends its workload's sequence.   usage: bash tools/pmc_r04.sh [workloads...]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
passes() {   # passes <workload> cmd...
  local w=$1; shift
  local dir=gpurun_out/pmc_r04_$w
  mkdir -p "$dir"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$dir" -o fetch -- "$@" > "$dir/fetch.log" 2>&1 \
   && timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$dir" -o write -- "$@" > "$dir/write.log" 2>&1 \
   && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$dir" -o sq -- "$@" > "$dir/sq.log" 2>&1
  echo "$w rc=$?"
  python3 tools/pmc_summarize.py "$dir" "$dir/summary.json" > "$dir/summary.txt" 2>&1
  find "$dir" \( -name '*_counter_collection.csv' -o -name '*_kernel_trace.csv' \) -delete
}
WL=${*:-vanilla instant part4}
for w in $WL; do
  case $w in
    vanilla) passes vanilla python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-instant --no-part4 --render-frames 1 ;;
    instant) python3 tools/instant_profile_loop.py train /tmp/instant_state.pt > gpurun_out/pmc_r04_instant_train.log 2>&1 \
             && passes instant python3 tools/instant_profile_loop.py loop /tmp/instant_state.pt 24 ;;
    part4) passes part4 python3 tools/part4_profile_loop.py 40 ;;
  esac
done
ls -la gpurun_out/pmc_r04_* | head -40
