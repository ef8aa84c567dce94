# memory-side counters of the single-launch forward, one pass per run (results under gpurun_out/pmc_mem_*)
R=$GRAFT_REPO_ROOT
cd $R && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc_mem_a -- python3 tools_prof/prof_mega.py > $R/gpurun_out/pmc_mem_a.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_sum --output-format csv -d $R/gpurun_out/pmc_mem_b -- python3 tools_prof/prof_mega.py > $R/gpurun_out/pmc_mem_b.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_HIT_sum --output-format csv -d $R/gpurun_out/pmc_mem_c -- python3 tools_prof/prof_mega.py > $R/gpurun_out/pmc_mem_c.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, glob, collections
for tag in "abc":
    for f in glob.glob(f"gpurun_out/pmc_mem_{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "backbone_mega" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc:
            print(tag, k, acc[k] / max(1, n[k]) , "per dispatch row;", n[k], "rows")
PY
