#!/bin/bash
# SQ counters and HBM-side traffic of the eight-wave GEMM (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_g8_sq gpurun_out/pmc_g8_fetch gpurun_out/pmc_g8_write
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_g8_sq -- python3 tools_prof/prof_gemm8p.py > gpurun_out/pmc_g8_sq.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_g8_fetch -- python3 tools_prof/prof_gemm8p.py > gpurun_out/pmc_g8_fetch.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_g8_write -- python3 tools_prof/prof_gemm8p.py > gpurun_out/pmc_g8_write.log 2>&1 || exit 1
for k in "gemm8p_kernel<1" "gemm8p_kernel<2"; do echo "== $k"; python3 tools_prof/pmc_to_json.py "$k" gpurun_out/pmc_g8_sq gpurun_out/pmc_g8_fetch gpurun_out/pmc_g8_write; done
