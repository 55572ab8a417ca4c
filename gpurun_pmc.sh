#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_r01
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_r01 -o sq1 -- python3 tools_pmc_fwd.py > gpurun_out/pmc_r01/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_r01 -o sq2 -- python3 tools_pmc_fwd.py > gpurun_out/pmc_r01/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r01 -o tcc1 -- python3 tools_pmc_fwd.py > gpurun_out/pmc_r01/tcc1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_r01 -o tcc2 -- python3 tools_pmc_fwd.py > gpurun_out/pmc_r01/tcc2.log 2>&1
ls gpurun_out/pmc_r01
