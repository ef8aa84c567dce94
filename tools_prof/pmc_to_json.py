"""Per-launch counter sums of one kernel from rocprofv3 --pmc passes (newest counter_collection.csv of each directory).

usage: pmc_to_json.py kernel_substring dir [dir ...]   -> JSON on stdout"""
import csv, glob, json, os, sys
from collections import defaultdict

kern = sys.argv[1]
out = {}
for d in sys.argv[2:]:
    files = sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    sums, disp, dur = defaultdict(float), defaultdict(set), {}
    with open(files[-1]) as fh:
        for r in csv.DictReader(fh):
            if kern in r["Kernel_Name"]:
                sums[r["Counter_Name"]] += float(r["Counter_Value"])
                disp[r["Counter_Name"]].add(r["Dispatch_Id"])
                dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for c in sums:
        out[c] = sums[c] / max(1, len(disp[c]))
    if dur:
        out.setdefault("duration_us", {})[os.path.basename(d.rstrip("/"))] = sum(dur.values()) / len(dur)
print(json.dumps(out, indent=1))
