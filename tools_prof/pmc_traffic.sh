#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for k in ffn attn; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${k}_fetch -- python3 tools_prof/prof_$k.py > gpurun_out/pmc_${k}_fetch.log 2>&1 || exit 1
  timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_${k}_write -- python3 tools_prof/prof_$k.py > gpurun_out/pmc_${k}_write.log 2>&1 || exit 1
done
