#!/bin/bash
# rocprofv3 kernel-trace + stats (round 3): the default bench (vanilla step at bf16 images + fp8 secondary + render), the
# Instant-NGP workload and the Part 4 engine loop; only the per-kernel statistics travel back -> copy into profiles/
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof_r03
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -o bench -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-instant --no-part4 > gpurun_out/prof_r03/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -o instant -- python3 bench.py --workload instant --steps 30 > gpurun_out/prof_r03/instant_stdout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -o part4 -- python3 tools/part4_profile_loop.py 96 > gpurun_out/prof_r03/part4_stdout.log 2>&1
# the per-dispatch traces are large: only the per-kernel statistics travel back
find gpurun_out/prof_r03 -name '*_kernel_trace.csv' -delete
ls -la gpurun_out/prof_r03 | head -30
