#!/bin/bash
# Ordered launch sequence of ONE steady-state step (kernel + memory-copy trace, no counters) of the Instant and the Part 4 engine:
# which tiny launches (fills, copies, one-workgroup kernels) sit between the kernels that do the work.
#   usage: bash tools/step_trace.sh [tag]   -> gpurun_out/step_trace[_tag]/{instant,part4}_sequence.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/step_trace${1:+_$1}
mkdir -p "$OUT"
python3 tools/instant_profile_loop.py train /tmp/instant_state.pt > "$OUT/instant_train.log" 2>&1 || { tail -5 "$OUT/instant_train.log"; exit 1; }
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o instant -- python3 tools/instant_profile_loop.py loop /tmp/instant_state.pt 24 > "$OUT/instant_stdout.log" 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o part4 -- python3 tools/part4_profile_loop.py 24 > "$OUT/part4_stdout.log" 2>&1
python3 tools/step_trace_summarize.py "$OUT" instant nerf::sample_compact > "$OUT/instant_sequence.txt"
python3 tools/step_trace_summarize.py "$OUT" part4 nerf::sample_compact > "$OUT/part4_sequence.txt"
find "$OUT" -name '*_trace.csv' -delete
tail -3 "$OUT"/*_stdout.log
cat "$OUT/instant_sequence.txt"
