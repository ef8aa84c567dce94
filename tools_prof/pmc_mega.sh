#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_mega_sq -- python3 tools_prof/prof_mega.py > gpurun_out/pmc_mega_sq.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_mega_fetch -- python3 tools_prof/prof_mega.py > gpurun_out/pmc_mega_fetch.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mega_write -- python3 tools_prof/prof_mega.py > gpurun_out/pmc_mega_write.log 2>&1 || exit 1
