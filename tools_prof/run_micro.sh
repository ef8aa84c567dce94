#!/bin/bash
# Builds and runs the micro-benchmarks of tools_prof/micro/ on the GPU box, one after the other, each under its own
# time limit; the outputs go to gpurun_out/micro.txt (copied to profiles/ by hand).
cd $GRAFT_REPO_ROOT || exit 1
mkdir -p gpurun_out /tmp/micro
: > gpurun_out/micro.txt
for f in stream_rate dma_forms lds_rate mfma_fill mfma_order mfma_dma mfma_real store_pattern; do
  echo "== $f" >> gpurun_out/micro.txt
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/micro/$f tools_prof/micro/$f.hip 2> /dev/null || exit 1
  timeout -k 10 60 /tmp/micro/$f >> gpurun_out/micro.txt 2>&1 || exit 1
done
for t in bench_launch_floor bench_gemm_stride bench_attn_fwd; do
  echo "== $t" >> gpurun_out/micro.txt
  timeout -k 10 90 python tools_prof/$t.py 2>/dev/null >> gpurun_out/micro.txt || exit 1
done
tail -5 gpurun_out/micro.txt
