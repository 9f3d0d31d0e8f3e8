#!/bin/bash
# Here, after tools/r03_refresh_profiles.sh has run on the GPU box: copy what it produced into profiles/ (the tracked copies).
set -e
F=gpurun_out/r03_final
cp gpurun_out/prof_bench_r03/roofline.json profiles/roofline.json
cp gpurun_out/prof_bench_r03/roofline.json profiles/r03_roofline.json
cp gpurun_out/prof_bench_r03/summary.txt profiles/r03_bench_rocprofv3_summary.txt
cp "$(ls -t gpurun_out/prof_bench_r03/trace/*/*_kernel_stats.csv | head -1)" profiles/r03_bench_kernel_stats.csv
for c in 3 4 5; do
  cp gpurun_out/prof_bench_r03_config$c/roofline.json profiles/roofline_config$c.json
  cp gpurun_out/prof_bench_r03_config$c/summary.txt profiles/r03_config${c}_rocprofv3_summary.txt
  tail -1 $F/bench_config$c.json > profiles/r03_config${c}_bench_line.json
done
tail -1 $F/bench_line.json > profiles/r03_bench_line.json
cp $F/all_configs.jsonl profiles/r03_all_configs.jsonl
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
sha = bench.csrc_sha()
for n in ("roofline", "roofline_config3", "roofline_config4", "roofline_config5"):
    assert json.load(open("profiles/%s.json" % n))["csrc_sha"] == sha, n
for n in ("r03_bench_line", "r03_config3_bench_line", "r03_config4_bench_line", "r03_config5_bench_line"):
    d = json.load(open("profiles/%s.json" % n)); r = d["roofline"]
    assert r["frac"] is not None and "stale_profile" not in r, n
    print(n, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["kernel_ms_per_step"])
print("profiles/ is consistent with kernel sources", sha)
PY
