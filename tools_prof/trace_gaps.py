"""Idle time of a stream between consecutive kernels of a rocprofv3 --kernel-trace run, summed by (kernel before ->
kernel after) over the steps a marker kernel delimits.

usage: trace_gaps.py kernel_trace.csv marker first_step n_steps [stream]"""
import csv, sys
from collections import defaultdict

path, marker, first, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
with open(path) as fh:
    rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "0")))
                   for r in csv.DictReader(fh)), key=lambda r: r[1])
idx = [i for i, r in enumerate(rows) if marker in r[0]]
seg = rows[idx[first]:idx[first + steps]]
streams = defaultdict(list)
for r in seg:
    streams[r[3]].append(r)
main = sys.argv[5] if len(sys.argv) > 5 else max(streams, key=lambda k: len(streams[k]))
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40]
gaps, total = defaultdict(float), 0.0
cnt = defaultdict(int)
ks = streams[main]
for a, b in zip(ks, ks[1:]):
    g = (b[1] - a[2]) / 1e3
    if g > 0:
        gaps[(short(a[0]), short(b[0]))] += g; cnt[(short(a[0]), short(b[0]))] += 1; total += g
print(f"stream {main}: {len(ks) / steps:.0f} kernels per step, idle {total / steps:.1f} us per step")
for k, v in sorted(gaps.items(), key=lambda x: -x[1])[:25]:
    print(f"{v / steps:7.1f} us/step  {cnt[k] / steps:5.1f} x {v / cnt[k]:6.2f} us   {k[0]} -> {k[1]}")
