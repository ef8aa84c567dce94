set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 env BD_GEMM8P=2 python -u tools_prof/check_gemm8p.py > gpurun_out/g8_check.log 2>&1 || { tail -30 gpurun_out/g8_check.log; exit 1; }
tail -1 gpurun_out/g8_check.log
timeout -k 10 300 env BD_GEMM8P=2 BD_GEMM8P_SLOTS=16 python -u tools_prof/check_gemm8p.py > gpurun_out/g8_check16.log 2>&1 || { tail -30 gpurun_out/g8_check16.log; exit 1; }
tail -1 gpurun_out/g8_check16.log
timeout -k 10 200 python -u tools_prof/bench_gemm_epi.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 500 env BD_GEMM8P=2 python -u -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_train.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -20 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
