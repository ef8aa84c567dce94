set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_train.py tests/test_gpu_train_packed.py tests/test_gpu_train_c4.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -3 gpurun_out/t.log
for w in 0 1 0 1; do
  echo "--- BD_WGRAD_GROUP=$w"
  timeout -k 10 300 env BD_WGRAD_GROUP=$w python -u bench.py --steps 1 --warmup 0 --train-steps 60 --no-configs --cpu-seqs 0 --cpu-train-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('ms/step', round(t['ms_per_step'],3), 'tok/s', round(t['value']), 'loss', t['final_loss']); k=t['kernels']; print({n:(round(v['avg_launch_ms']*1e3,1), v['launches']) for n,v in k.items() if 'wgrad' in n or 'gemm' in n})" || exit 1
done
