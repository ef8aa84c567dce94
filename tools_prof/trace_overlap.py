"""How much of a rocprofv3 kernel trace's span has two or more kernels in flight (kernel_trace.csv; the second half of
the trace only, to leave warm-up out).  usage: trace_overlap.py kernel_trace.csv"""
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?")))
rows.sort()
rows = rows[len(rows) // 2:]
ev = []
for s, e, _, _ in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
depth = 0; last = ev[0][0]; t = {0: 0, 1: 0, 2: 0}
for ts, d in ev:
    t[min(depth, 2)] += ts - last
    last = ts; depth += d
span = ev[-1][0] - ev[0][0]
print(f"{len(rows)} kernels over {span/1e6:.1f} ms: idle {t[0]/span:.1%}, one kernel {t[1]/span:.1%}, two or more {t[2]/span:.1%}; "
      f"queues {sorted(set(r[3] for r in rows))}")
