#!/bin/bash
# rocprofv3 kernel statistics of the default bench command (CPU baselines off: they are host work)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/benchprof
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/benchprof -o bench --output-format csv -- python3 bench.py --cpu-seqs 0 --cpu-train-steps 0 > gpurun_out/benchprof.json 2> gpurun_out/benchprof.err || { tail gpurun_out/benchprof.err; exit 1; }
f=$(find gpurun_out/benchprof -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/benchprof_kernel_stats.csv
rm -f gpurun_out/benchprof/*kernel_trace.csv
head -12 $f | cut -c1-150
