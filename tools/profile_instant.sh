#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_instant
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_instant -o inst -- python3 tools/instant_profile_loop.py > gpurun_out/prof_instant/stdout.log 2>&1
grep "ms/step" gpurun_out/prof_instant/stdout.log
