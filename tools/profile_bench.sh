#!/bin/bash
# rocprofv3 kernel-trace + stats of the bench (round 1); summaries are copied into profiles/ afterwards
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r02
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o bench -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r02/bench_stdout.log 2>&1
ls -R gpurun_out/prof_r02 | head -30
