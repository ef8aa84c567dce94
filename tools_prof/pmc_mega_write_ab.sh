#!/bin/bash
# WRITE_SIZE of the single-launch forward with and without the table fetch by LDS-DMA (one --pmc pass each)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for m in 0 1; do
  export BD_MEGA_TABLES_DMA=$m
  timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w_$m -- python3 tools_prof/prof_mega.py > gpurun_out/pmc_w_$m.log 2>&1 || exit 1
  echo "tables by DMA = $m"; python3 tools_prof/pmc_to_json.py backbone_mega_kernel gpurun_out/pmc_w_$m
done
