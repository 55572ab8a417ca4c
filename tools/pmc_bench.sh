#!/bin/bash
# HBM-traffic counters of the bench kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), plus an SQ pass for MFMA utilisation.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r02
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --render-frames 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r02 -o fetch -- $CMD > gpurun_out/pmc_r02/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r02 -o write -- $CMD > gpurun_out/pmc_r02/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r02 -o sq -- $CMD > gpurun_out/pmc_r02/sq.log 2>&1
ls gpurun_out/pmc_r02 | head -30
