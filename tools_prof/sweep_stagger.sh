# headline sampling batch for several phase shifts between the CUs' first workgroups (BD_MEGA_STAGGER, default 3)
for u in 3 0 1 2 4 6 8 12 3; do
BD_MEGA_STAGGER=$u timeout -k 10 120 python bench.py --steps 2 --warmup 1 --cpu-seqs 0 --cpu-train-steps 0 --no-configs --train-steps 0 > gpurun_out/stg.json 2> gpurun_out/stg.err || exit 1
python -c "
import json;r=json.load(open('gpurun_out/stg.json'));print('stagger $u:', round(r['value'],1), 'seq/s, frac', round(r['roofline']['frac'],4))"
done
