#!/bin/bash
# Round 3: the bench line, then the rocprofv3 passes for the headline workload and for BASELINE configs 5, 3 and 4, then all configs.
# Two gpurun calls' worth of work: PART=1 (bench + headline + config 5) and PART=2 (configs 3, 4 + run_configs).
mkdir -p gpurun_out/r03_final
if [ "${PART:-1}" = "1" ]; then
  timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 > gpurun_out/r03_final/bench_line.json 2> gpurun_out/r03_final/bench.err
  echo "bench rc $?"; cut -c1-300 gpurun_out/r03_final/bench_line.json
  timeout -k 10 500 bash tools/profile_bench.sh r03
  timeout -k 10 500 bash tools/profile_bench.sh r03_config5 --workload config5
else
  timeout -k 10 700 bash tools/profile_bench.sh r03_config3 --workload config3 --steps 1 --warmup 1
  timeout -k 10 400 bash tools/profile_bench.sh r03_config4 --workload config4
  timeout -k 10 600 python3 tools/run_configs.py > gpurun_out/r03_final/all_configs.jsonl 2>/dev/null
  cat gpurun_out/r03_final/all_configs.jsonl
fi
