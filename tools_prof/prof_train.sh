#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trainprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/trainprof -o tr --output-format csv -- python3 bench.py --steps 1 --warmup 0 --train-steps 40 --no-configs --cpu-seqs 0 --cpu-train-steps 0 > gpurun_out/trainprof.json 2> gpurun_out/trainprof.err || { tail gpurun_out/trainprof.err; exit 1; }
python3 - <<'PY'
import csv, glob, json
d = json.loads(open("gpurun_out/trainprof.json").read().strip().splitlines()[-1])
print("train ms/step", d["train"]["ms_per_step"], "tok/s", d["train"]["value"])
f = glob.glob("gpurun_out/trainprof/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n_steps = 40 + 10 + 5 + 3 + 10     # timed + warm-up + profiled + all-rows warm-up + all-rows timed (approx.)
for r in rows[:32]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):6d} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:8.1f}")
PY
