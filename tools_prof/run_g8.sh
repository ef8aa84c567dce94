set -o pipefail
mkdir -p gpurun_out
for d in 0 10000 20000 30000 45000; do
  echo "--- delay $d cycles"
  timeout -k 10 200 env BD_G8_DELAY=$d python -u tools_prof/bench_gemm_epi.py 2>&1 | grep -v amdgpu.ids || exit 1
done
