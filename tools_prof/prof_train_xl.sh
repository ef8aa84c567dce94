#!/bin/bash
# rocprofv3 kernel statistics of the training step at the reference's xl size (tools_prof/bench_train_sizes.py xl)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/xlprof
BD_TRAIN_SIZES_ONLY=${1:-xl} timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/xlprof -o xl --output-format csv -- python3 tools_prof/bench_train_sizes.py > gpurun_out/xlprof.log 2> gpurun_out/xlprof.err || { tail gpurun_out/xlprof.err; exit 1; }
f=$(find gpurun_out/xlprof -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/xl_kernel_stats.csv
rm -rf gpurun_out/xlprof
grep "packed rows" gpurun_out/xlprof.log
head -24 gpurun_out/xl_kernel_stats.csv | cut -c1-160
