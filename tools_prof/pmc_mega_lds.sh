# LDS-side counters of the single-launch forward (one pass)
R=$GRAFT_REPO_ROOT
cd $R && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_mem_d -- python3 tools_prof/prof_mega.py > $R/gpurun_out/pmc_mem_d.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/pmc_mem_e -- python3 tools_prof/prof_mega.py > $R/gpurun_out/pmc_mem_e.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections
for tag in "de":
    for f in glob.glob(f"gpurun_out/pmc_mem_{tag}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "backbone_mega" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc:
            print(tag, k, acc[k] / max(1, n[k]))
PY
