#!/bin/bash
# rocprofv3 kernel-trace + stats of the default bench (round 2): vanilla step + renders + the Instant-NGP block;
# summaries are copied into profiles/ afterwards
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r02
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o bench -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r02/bench_stdout.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o instant -- python3 bench.py --workload instant --steps 30 > gpurun_out/prof_r02/instant_stdout.log 2>&1
# the per-dispatch traces are large (the Instant run has >100 k rows): only the per-kernel statistics travel back
rm -f gpurun_out/prof_r02/*_kernel_trace.csv
ls -la gpurun_out/prof_r02 | head -30
