# two training-leg runs of bench.py on this box (A/B material: box-to-box variation is larger than most single changes)
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --cpu-seqs 0 --cpu-train-steps 0 --no-configs --train-steps 60 > gpurun_out/ab_train.json 2> gpurun_out/ab_train.err || exit 1
python -c "
import json;r=json.load(open('gpurun_out/ab_train.json'));t=r['train'];print('train', t['value'], t['ms_per_step'], t.get('final_loss'))"
done
