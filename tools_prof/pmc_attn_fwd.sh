#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/pmc_attn_fwd_sq -- python3 tools_prof/prof_attn_fwd.py > gpurun_out/pmc_attn_fwd_sq.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_attn_fwd_sq2 -- python3 tools_prof/prof_attn_fwd.py > gpurun_out/pmc_attn_fwd_sq2.log 2>&1 || exit 1
python3 tools_prof/pmc_to_json.py attn_fwd_kernel gpurun_out/pmc_attn_fwd_sq gpurun_out/pmc_attn_fwd_sq2
