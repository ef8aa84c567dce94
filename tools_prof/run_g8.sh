set -o pipefail
for w in 0 1 0 1; do
  echo "--- wide=$w"
  timeout -k 10 300 env BD_GEMM8P_WIDE=$w python -u bench.py --steps 1 --warmup 0 --train-steps 60 --no-configs --cpu-seqs 0 --cpu-train-steps 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['train']; print('ms/step', round(t['ms_per_step'],3), 'tok/s', round(t['value']))" || exit 1
done
