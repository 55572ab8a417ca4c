#!/bin/bash
# rocprofv3 kernel-trace + stats of the Instant-NGP bench (engine path): per-kernel totals over 1000
# training steps, the steady-state timing loop and the 800x800 renders
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_instant
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_instant -o inst -- python3 bench.py --workload instant > gpurun_out/prof_instant/stdout.log 2>&1
ls gpurun_out/prof_instant | head
