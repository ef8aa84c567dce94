#!/bin/bash
# usage: pmc_ffn.sh <tag> ; env BD_FFN_DBG selects the kernel variant
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_$1 -- python3 tools_prof/prof_ffn.py > gpurun_out/pmc_$1.log 2>&1
