"""Per-step kernel table from a rocprofv3 --kernel-trace run (the sqlite database or the kernel_trace.csv): steps are
delimited by a marker kernel.

usage: rocpd_steps.py results.db|kernel_trace.csv marker first_step n_steps [out.csv]"""
import csv, sqlite3, sys
from collections import defaultdict

marker = sys.argv[2]; first = int(sys.argv[3]); steps = int(sys.argv[4])
if sys.argv[1].endswith(".csv"):
    with open(sys.argv[1]) as fh:
        rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "0")))
                       for r in csv.DictReader(fh)), key=lambda r: r[1])
else:
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
    rows = db.execute(f"select s.kernel_name, d.start, d.end, d.stream_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
idx = [i for i, r in enumerate(rows) if marker in r[0]]
seg = rows[idx[first]:idx[first + steps]]
t = defaultdict(float); c = defaultdict(int); by_stream = defaultdict(float)
for n, s, e, st in seg:
    key = n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].replace('_ZN12_GLOBAL__N_1', '')[:64]; t[key] += (e - s) / 1e3; c[key] += 1; by_stream[st] += (e - s) / 1e3
print(f"steps {first}..{first + steps} of {len(idx)}: span {(seg[-1][2] - seg[0][1]) / 1e3 / steps:.1f} us per step; busy per stream",
      {k: round(v / steps, 1) for k, v in by_stream.items()})
lines = ["kernel,calls_per_step,us_per_step,avg_us"]
for k, v in sorted(t.items(), key=lambda x: -x[1]):
    lines.append(f"{k},{c[k] / steps:.2f},{v / steps:.1f},{v / c[k]:.2f}")
print("\n".join(lines[:32]))
if len(sys.argv) > 5:
    open(sys.argv[5], "w").write("\n".join(lines) + "\n")
