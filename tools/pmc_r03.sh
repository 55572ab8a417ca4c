#!/bin/bash
# HBM-traffic counters (round 3): FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3,
# WRITE_SIZE 2), an SQ pass for MFMA utilisation; for the vanilla step, the Instant-NGP workload and the Part 4 engine loop.
# The program comes directly after `--` (no env / bash -c hop under the profiler); every pass has its own time limit and a
# failed pass ends its workload's sequence.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
passes() {   # passes <workload> cmd...
  local w=$1; shift
  local dir=gpurun_out/pmc_r03_$w
  mkdir -p "$dir"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$dir" -o fetch -- "$@" > "$dir/fetch.log" 2>&1 \
   && timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$dir" -o write -- "$@" > "$dir/write.log" 2>&1 \
   && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$dir" -o sq -- "$@" > "$dir/sq.log" 2>&1
  echo "$w rc=$?"
  python3 tools/pmc_summarize.py "$dir" "$dir/summary.json" > "$dir/summary.txt" 2>&1
}
passes vanilla python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-instant --no-part4 --render-frames 1
passes instant python3 bench.py --workload instant --steps 6 --render-frames 1
passes part4 python3 tools/part4_profile_loop.py 40
# only the summaries travel back (the per-dispatch counter files are large)
find gpurun_out/pmc_r03_vanilla gpurun_out/pmc_r03_instant gpurun_out/pmc_r03_part4 \( -name '*_counter_collection.csv' -o -name '*_kernel_trace.csv' \) -delete
ls -la gpurun_out/pmc_r03_vanilla gpurun_out/pmc_r03_instant gpurun_out/pmc_r03_part4 | head -40
