set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 env BD_GEMM8P=2 python -u tools_prof/check_gemm8p.py > gpurun_out/g8_check.log 2>&1 || { tail -30 gpurun_out/g8_check.log; exit 1; }
tail -3 gpurun_out/g8_check.log
timeout -k 10 300 env BD_GEMM8P=2 BD_GEMM8P_SLOTS=16 python -u tools_prof/check_gemm8p.py > gpurun_out/g8_check16.log 2>&1 || { tail -30 gpurun_out/g8_check16.log; exit 1; }
tail -3 gpurun_out/g8_check16.log
for v in 0 256; do
  timeout -k 10 200 env BD_GEMM8P_SLOTS=$v python -u tools_prof/bench_gemm_big.py > gpurun_out/g8_slots$v.log 2>&1 || { tail gpurun_out/g8_slots$v.log; exit 1; }
  echo "--- slots $v"; grep -v amdgpu.ids gpurun_out/g8_slots$v.log
done
