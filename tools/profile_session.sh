set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-side > $O/bench_prof.json 2> $O/bench_prof.err
bash tools/pmc_run.sh r02 > $O/pmc.log 2>&1
python3 tools/pmc_summary.py r02 >> $O/pmc.log 2>&1
bash tools/pmc_run.sh r02_bulk python3 tools/gpu_bulk.py > $O/pmc_bulk.log 2>&1
python3 tools/pmc_summary.py r02_bulk >> $O/pmc_bulk.log 2>&1
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 tools/gpu_lat.py --bulk > $O/latency.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/ubench/libwave_sections.so tools/ubench/wave_sections.hip 2> $O/wave_sections_build.err
python3 tools/gpu_wave_sections.py --batch > $O/wave_sections.txt 2>&1
bash tools/pmc_one.sh > $O/pmc_lone_wave.txt 2>&1
python3 tools/gpu_predict_timing.py > $O/predict.txt 2>&1
python3 tools/gpu_parity_sweep.py > $O/parity_sweep.txt 2>&1
ls $O
python3 tools/gpu_ltv_timing.py > $O/ltv_timing.txt 2>&1
python3 tools/bench_rollout.py --envs 256 --steps 64 > $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 2048 --steps 64 >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 8192 --steps 32 >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 256 --steps 64 --version v1 >> $O/rollout.jsonl 2>> $O/rollout.err
python3 tools/bench_rollout.py --envs 8192 --steps 32 --graph --groups 4 >> $O/rollout.jsonl 2>> $O/rollout.err
BENCH_FORCE_DIST=1 python3 bench.py --steps 10 --no-side --no-cpu-baseline > $O/bench_rccl_1rank.json 2>> $O/rollout.err
python3 tools/run_pure_mpc.py > $O/run_pure_mpc.txt 2>&1
ls $O
