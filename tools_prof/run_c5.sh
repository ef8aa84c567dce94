set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u tools_prof/bench_c5.py 8 256 10 > gpurun_out/c5_bench.log 2>&1 || { tail gpurun_out/c5_bench.log; exit 1; }
grep -v amdgpu.ids gpurun_out/c5_bench.log
rm -rf gpurun_out/c5prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/c5prof -o c5 --output-format csv -- python3 tools_prof/prof_c5.py > gpurun_out/c5_prof.log 2>&1 || { tail gpurun_out/c5_prof.log; exit 1; }
f=$(find gpurun_out/c5prof -name '*kernel_stats.csv' | head -1)
cut -c1-160 $f | head -25
