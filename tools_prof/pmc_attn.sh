#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_attn1 gpurun_out/pmc_attn2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_attn1 -- python3 tools_prof/prof_attn512.py > gpurun_out/pmc_attn1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_attn2 -- python3 tools_prof/prof_attn512.py > gpurun_out/pmc_attn2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_attn1", "pmc_attn2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "attn_fwd" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, sum(v) / len(v), len(v))
PY
