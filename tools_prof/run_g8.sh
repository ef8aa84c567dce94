set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 env BD_GEMM8P=2 python -u tools_prof/check_gemm8p.py > gpurun_out/g8_check.log 2>&1 || { tail -30 gpurun_out/g8_check.log; exit 1; }
tail -2 gpurun_out/g8_check.log
timeout -k 10 300 env BD_GEMM8P=2 BD_GEMM8P_SLOTS=16 python -u tools_prof/check_gemm8p.py > gpurun_out/g8_check16.log 2>&1 || { tail -30 gpurun_out/g8_check16.log; exit 1; }
tail -2 gpurun_out/g8_check16.log
timeout -k 10 200 python -u tools_prof/bench_gemm_big.py > gpurun_out/g8_new.log 2>&1 || { tail gpurun_out/g8_new.log; exit 1; }
grep -v amdgpu.ids gpurun_out/g8_new.log
timeout -k 10 250 python -u tools_prof/bench_sizes.py 2>&1 | grep -v amdgpu.ids | grep -E "forward|gemm"
