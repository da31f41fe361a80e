#!/bin/bash
# One GPU session that produces the evidence tracked under profiles/ (run through gpurun; tools/collect_profiles.sh copies
# the summaries from gpurun_out/ into profiles/).  usage: tools/profile_session.sh [tag]     default tag r06
# Every rocprofv3 command has the program itself after `--`; PMC passes are their own runs (tools/pmc_run.sh).
set -x
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}s
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side > $O/bench_prof.json 2> $O/bench_prof.err
bash tools/pmc_run.sh $TAG > $O/pmc.log 2>&1
python3 tools/pmc_summary.py $TAG >> $O/pmc.log 2>&1
bash tools/pmc_run.sh ${TAG}_bulk python3 tools/gpu_bulk.py > $O/pmc_bulk.log 2>&1
python3 tools/pmc_summary.py ${TAG}_bulk >> $O/pmc_bulk.log 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 tools/gpu_lat.py --bulk > $O/latency.txt 2>&1
python3 tools/gpu_capline.py >> $O/latency.txt 2>&1
python3 tools/gpu_inflight.py > $O/inflight.txt 2>&1
python3 tools/gpu_queues.py > $O/stream_queues.txt 2>&1
python3 tools/gpu_tail.py > $O/tail.txt 2>&1
python3 tools/gpu_predict_timing.py > $O/predict.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/ubench/libpreamble_sections.so tools/ubench/preamble_sections.hip > $O/preamble_sections.err 2>&1
python3 tools/gpu_preamble_sections.py > $O/preamble_sections.txt 2>> $O/preamble_sections.err
python3 tools/gpu_parity_sweep.py > $O/parity_sweep.txt 2>&1
ls $O
python3 tools/gpu_ltv_timing.py > $O/ltv_timing.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ltv_stats -- python3 tools/ltv_profile.py > $O/ltv_prof.txt 2> $O/ltv_prof.err
# rollouts: the collector's default (hipGraph step, fused environment) and the eager / torch-environment rows beside it
python3 tools/bench_rollout.py --envs 256 2048 8192 --steps 64 > $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 256 2048 --steps 64 --no-graph >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 256 --steps 64 --no-graph --env-backend torch >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 256 --steps 64 --version v1 >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 256 --steps 64 --version v1 --max-iter 1000 --tol 1e-6 >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 8192 --steps 32 --groups 4 >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 2048 --steps 64 --groups 2 >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 2048 --steps 32 --version v1 >> $O/rollout.jsonl 2>> $O/rollout.err
rocprofv3 --kernel-trace --output-format csv -d $O/steptrace -- python3 tools/bench_rollout.py --envs 256 > /dev/null 2>> $O/rollout.err
python3 tools/graph_step_gaps.py $O/steptrace > $O/graph_step_trace.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/rollout_stats -- python3 tools/bench_rollout.py --envs 256 --steps 64 --no-graph > $O/rollout_prof.json 2> $O/rollout_prof.err
BENCH_FORCE_DIST=1 python3 bench.py --steps 10 --no-side --no-cpu-baseline > $O/bench_rccl_1rank.json 2>> $O/rollout.err
python3 tools/run_pure_mpc.py > $O/run_pure_mpc.txt 2>&1
ls $O
