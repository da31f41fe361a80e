#!/bin/bash
# Copy the evidence of a tools/profile_session.sh run (merged back under gpurun_out/) into profiles/ (tracked).
# usage: tools/collect_profiles.sh [tag]     default tag r06
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r06}
S=gpurun_out/${TAG}s
P=profiles
cp "$(ls -t $S/stats/*/*_kernel_stats.csv | head -1)" $P/${TAG}_bench_kernel_stats.csv
cp "$(ls -t $S/ltv_stats/*/*_kernel_stats.csv | head -1)" $P/${TAG}_ltv_kernel_stats.csv
# the rollout's trace holds a hundred torch kernels: keep the engine's own and the ten largest others
python3 - "$(ls -t $S/rollout_stats/*/*_kernel_stats.csv | head -1)" $P/${TAG}_rollout_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
keep = [rows[0]] + [r for r in rows[1:] if "mpc_" in r[0]] + [r for r in rows[1:] if "mpc_" not in r[0]][:10]
csv.writer(open(sys.argv[2], "w", newline="")).writerows(keep)
PY
cp $S/bench_prof.json $P/${TAG}_bench_profiled_run.json
cp $S/bench.json $P/${TAG}_bench.json
for f in latency tail predict preamble_sections parity_sweep ltv_timing run_pure_mpc inflight stream_queues graph_step_trace; do cp $S/$f.txt $P/${TAG}_$f.txt; done
cp $S/rollout.jsonl $P/${TAG}_rollout.jsonl
cp $S/bench_rccl_1rank.json $P/${TAG}_bench_rccl_1rank.json
python3 tools/pmc_summary.py $TAG
python3 tools/pmc_summary.py ${TAG}_bulk
python3 tools/resource_usage.py > $P/${TAG}_resource_usage.txt
ls -la $P | grep ${TAG}_
