#!/bin/bash
# per-kernel times of the hash-grid backward (atomic and binned forms); summary -> gpurun_out/r02/hb/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02/hb
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/hb -o hb -- python3 tools/time_hash_bwd.py ${1:-198000} > gpurun_out/r02/hb/out.txt 2>&1
rm -f gpurun_out/r02/hb/*_kernel_trace.csv
grep -v "^W2026\|^E2026" gpurun_out/r02/hb/out.txt | tail -5
cut -d, -f1-8 gpurun_out/r02/hb/hb_kernel_stats.csv | head -14
