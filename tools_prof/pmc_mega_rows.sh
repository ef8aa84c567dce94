#!/bin/bash
# HBM-side traffic and SQ counters of the chosen-row forward (three separate passes: FETCH_SIZE and WRITE_SIZE do not fit one)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_rows_sq -- python3 tools_prof/prof_mega_rows.py > gpurun_out/pmc_rows_sq.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_rows_fetch -- python3 tools_prof/prof_mega_rows.py > gpurun_out/pmc_rows_fetch.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_rows_write -- python3 tools_prof/prof_mega_rows.py > gpurun_out/pmc_rows_write.log 2>&1 || exit 1
for k in backbone_mega_kernel mega_tail_kernel; do python3 tools_prof/pmc_to_json.py $k gpurun_out/pmc_rows_sq gpurun_out/pmc_rows_fetch gpurun_out/pmc_rows_write > gpurun_out/pmc_rows_$k.json; done
