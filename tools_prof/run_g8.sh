set -o pipefail
echo "--- 8p"; timeout -k 10 200 python -u tools_prof/bench_gemm_epi.py 2>&1 | grep -v amdgpu.ids || exit 1
echo "--- gemm256 (4 waves)"; timeout -k 10 200 env BD_GEMM8P=0 python -u tools_prof/bench_gemm_epi.py 2>&1 | grep -v amdgpu.ids || exit 1
echo "--- 128x128 kernels"; timeout -k 10 200 env BD_GEMM8P=0 BD_GEMM256=0 python -u tools_prof/bench_gemm_epi.py 2>&1 | grep -v amdgpu.ids || exit 1
