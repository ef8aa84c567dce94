set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 env BD_GEMM8P=2 python -u tools_prof/check_gemm8p.py > gpurun_out/g8_check.log 2>&1 || { tail -30 gpurun_out/g8_check.log; exit 1; }
tail -20 gpurun_out/g8_check.log
timeout -k 10 200 env BD_GEMM8P=0 python -u tools_prof/bench_gemm_big.py > gpurun_out/g8_bench_off.log 2>&1 && \
timeout -k 10 200 env BD_GEMM8P=1 python -u tools_prof/bench_gemm_big.py > gpurun_out/g8_bench_on.log 2>&1
echo "--- off"; cat gpurun_out/g8_bench_off.log; echo "--- on"; cat gpurun_out/g8_bench_on.log
