#!/bin/bash
# usage: ab_train.sh "ENV1=a ENV2=b" "ENVX=c" ...   (each argument one arm; two rounds)
for i in 1 2; do
  for arm in "$@"; do
    r=$(env $arm timeout -k 10 300 python bench.py --steps 1 --no-configs --cpu-seqs 0 --cpu-train-steps 0 --train-steps 100 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['train']['ms_per_step'],4))") || exit 1
    echo "arm [$arm] train ms/step $r"
  done
done
