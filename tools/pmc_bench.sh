#!/bin/bash
# HBM-traffic counters of the bench kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), plus an SQ pass for MFMA utilisation;
# once for the vanilla step + render, once for the Instant-NGP workload.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r02 gpurun_out/pmc_r02i
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-instant --render-frames 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r02 -o fetch -- $CMD > gpurun_out/pmc_r02/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r02 -o write -- $CMD > gpurun_out/pmc_r02/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r02 -o sq -- $CMD > gpurun_out/pmc_r02/sq.log 2>&1
CMD="python3 bench.py --workload instant --steps 6 --render-frames 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r02i -o fetch -- $CMD > gpurun_out/pmc_r02i/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r02i -o write -- $CMD > gpurun_out/pmc_r02i/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r02i -o sq -- $CMD > gpurun_out/pmc_r02i/sq.log 2>&1
python3 tools/pmc_summarize.py gpurun_out/pmc_r02 gpurun_out/pmc_r02/summary.json > gpurun_out/pmc_r02/summary.txt 2>&1
python3 tools/pmc_summarize.py gpurun_out/pmc_r02i gpurun_out/pmc_r02i/summary.json > gpurun_out/pmc_r02i/summary.txt 2>&1
# only the summaries travel back (the per-dispatch counter files are large)
rm -f gpurun_out/pmc_r02/*_counter_collection.csv gpurun_out/pmc_r02/*_kernel_trace.csv gpurun_out/pmc_r02i/*_counter_collection.csv gpurun_out/pmc_r02i/*_kernel_trace.csv
ls -la gpurun_out/pmc_r02 gpurun_out/pmc_r02i | head -40
