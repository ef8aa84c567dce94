set -o pipefail
for j in 0 1 0 1; do
  echo "--- BD_WGRAD_SKIP=$j"
  timeout -k 10 300 env BD_WGRAD_SKIP=$j python -u bench.py --steps 1 --warmup 0 --train-steps 60 --no-configs --cpu-seqs 0 --cpu-train-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('ms/step', round(t['ms_per_step'],3), 'tok/s', round(t['value'])); k=t['kernels']; print({n:(round(v['avg_launch_ms']*1e3,1), v['launches']) for n,v in k.items()})" || exit 1
done
