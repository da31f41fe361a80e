#!/bin/bash
# PMC passes for the solve kernel (each pass its own rocprofv3 run, counters only - no trace domains mixed in; the program
# itself follows `--`).  usage: tools/pmc_run.sh <tag> [command...]   default command: the bench without side measurements
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r06}
shift || true
if [ $# -eq 0 ]; then set -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-side; fi
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT
mkdir -p $OUT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- "${CMD[@]}" > $OUT/$name.json 2> $OUT/$name.err || echo "pass $name failed"; }
CMD=("$@")
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH
run sq3 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_SALU
run sq4 SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
run fetch FETCH_SIZE
run write WRITE_SIZE
for d in $OUT/*/; do find $d -name "*counter_collection.csv" | head -1; done
