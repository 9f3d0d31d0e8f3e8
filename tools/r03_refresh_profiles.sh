#!/bin/bash
# Everything profiles/ holds for the current build, in ONE gpurun call: the rocprofv3 passes of tools/profile_bench.sh for the headline
# workload and BASELINE configs 5, 3, 4; each roofline.json put in place (in this copy of the repo) BEFORE the bench lines are taken,
# so that the lines carry a fraction stamped with this build's kernel sources; then all five configs (tools/run_configs.py).
# Results under gpurun_out/r03_final/ and gpurun_out/prof_bench_*/ -- copy them into profiles/ as tools/README.md says.
F=gpurun_out/r03_final
mkdir -p $F
timeout -k 10 400 bash tools/profile_bench.sh r03 > $F/prof_r03.log 2>&1 && cp gpurun_out/prof_bench_r03/roofline.json profiles/roofline.json
timeout -k 10 400 bash tools/profile_bench.sh r03_config5 --workload config5 > $F/prof_c5.log 2>&1 && cp gpurun_out/prof_bench_r03_config5/roofline.json profiles/roofline_config5.json
timeout -k 10 600 bash tools/profile_bench.sh r03_config3 --workload config3 --steps 1 --warmup 1 > $F/prof_c3.log 2>&1 && cp gpurun_out/prof_bench_r03_config3/roofline.json profiles/roofline_config3.json
timeout -k 10 400 bash tools/profile_bench.sh r03_config4 --workload config4 > $F/prof_c4.log 2>&1 && cp gpurun_out/prof_bench_r03_config4/roofline.json profiles/roofline_config4.json
echo "profiles done"
timeout -k 10 500 python3 bench.py > $F/bench_line.json 2> $F/bench.err; echo "bench rc $?"; cut -c1-160 $F/bench_line.json
timeout -k 10 300 python3 bench.py --workload config5 --no-oneshot --no-cpu-baseline --steps 5 --warmup 1 > $F/bench_config5.json 2>/dev/null; cut -c120-200 $F/bench_config5.json
timeout -k 10 300 python3 bench.py --workload config3 --no-oneshot --no-cpu-baseline --steps 2 --warmup 1 > $F/bench_config3.json 2>/dev/null; cut -c120-200 $F/bench_config3.json
timeout -k 10 300 python3 bench.py --workload config4 --no-oneshot --no-cpu-baseline --steps 5 --warmup 1 > $F/bench_config4.json 2>/dev/null; cut -c120-200 $F/bench_config4.json
timeout -k 10 600 python3 tools/run_configs.py > $F/all_configs.jsonl 2>/dev/null; cut -c1-110 $F/all_configs.jsonl
