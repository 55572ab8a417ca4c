#!/bin/bash
# rocprofv3 kernel-trace + stats (round 4): the default bench (vanilla step + render), the Instant-NGP STEADY-STATE loop
# (tools/instant_profile_loop.py: trained first, outside the profiler) and the Part 4 engine loop; only the per-kernel
# statistics travel back -> copy into profiles/.   usage: bash tools/profile_r04.sh [tag]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/prof_r04${1:+_$1}
mkdir -p "$OUT"
python3 tools/instant_profile_loop.py train /tmp/instant_state.pt > "$OUT/instant_train.log" 2>&1 || { tail -5 "$OUT/instant_train.log"; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o bench -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-instant --no-part4 > "$OUT/bench_stdout.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o instant -- python3 tools/instant_profile_loop.py loop /tmp/instant_state.pt 96 > "$OUT/instant_stdout.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o part4 -- python3 tools/part4_profile_loop.py 96 > "$OUT/part4_stdout.log" 2>&1
find "$OUT" -name '*_kernel_trace.csv' -delete
tail -2 "$OUT"/*_stdout.log
ls -la "$OUT" | head -30
