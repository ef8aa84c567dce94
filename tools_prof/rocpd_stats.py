"""Kernel statistics (the columns of rocprofv3's kernel_stats.csv) from the sqlite database that
`rocprofv3 --kernel-trace --stats` writes when no csv output format is asked for.

usage: rocpd_stats.py results.db out.csv"""
import math
import shutil
import sqlite3
import subprocess
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
dur = defaultdict(list)
for name, start, end in db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"):
    dur[name].append(end - start)
filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
names = list(dur)
try:      # the database keeps mangled names; kernel_stats.csv has them demangled
    plain = subprocess.run([filt], input="\n".join(n[:-3] if n.endswith(".kd") else n for n in names), capture_output=True,
                           text=True, check=True).stdout.split("\n")
    dur = {plain[i]: dur[n] for i, n in enumerate(names)}
except (OSError, subprocess.CalledProcessError):
    pass
total = sum(sum(v) for v in dur.values())
lines = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"']
for name, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    n, s = len(v), sum(v)
    mean = s / n
    sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
    lines.append(f'"{name}",{n},{s},{mean:.6f},{100.0 * s / total:.4g},{min(v)},{max(v)},{sd:.6f}')
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:8]))
