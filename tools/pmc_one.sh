cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_one/$name -- python3 tools/gpu_one.py 550 > gpurun_out/pmc_one_$name.log 2>&1; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_IFETCH
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("gpurun_out/pmc_one/*/**/*counter_collection.csv",recursive=True)):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mpc_solve" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
