#!/bin/bash
# Copy the evidence of a tools/profile_session.sh run (merged back under gpurun_out/) into profiles/ (tracked).
# usage: tools/collect_profiles.sh [tag]     default tag r02
set -e
cd "$(dirname "$0")/.."
TAG=${1:-r02}
S=gpurun_out/${TAG}s
P=profiles
latest=$(ls -t $S/stats/*/*_kernel_stats.csv | head -1)
cp "$latest" $P/${TAG}_bench_kernel_stats.csv
cp $S/bench_prof.json $P/${TAG}_bench_profiled_run.json
cp $S/bench.json $P/${TAG}_bench.json
for f in latency wave_sections pmc_lone_wave predict parity_sweep ltv_timing run_pure_mpc; do cp $S/$f.txt $P/${TAG}_$f.txt; done
cp $S/rollout.jsonl $P/${TAG}_rollout.jsonl
cp $S/bench_rccl_1rank.json $P/${TAG}_bench_rccl_1rank.json
python3 tools/pmc_summary.py $TAG
python3 tools/pmc_summary.py ${TAG}_bulk
python3 tools/resource_usage.py > $P/${TAG}_resource_usage.txt
for f in dispatch_skew setprio residency residency2; do [ -f gpurun_out/${TAG}_$f.txt ] && cp gpurun_out/${TAG}_$f.txt $P/${TAG}_$f.txt; done
ls -la $P | grep ${TAG}_
