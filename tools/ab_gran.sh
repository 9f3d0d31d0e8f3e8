#!/bin/bash
# A/B of the work-queue granularity knobs at full-frame (K=1) and 1/8-frame (K=8) launches
set -e
out=gpurun_out/ab_gran.txt; : > $out
run() { # label, env..., -- bench args
  label=$1; shift
  line=$(env "$@" timeout -k 10 240 python bench.py --no-cpu-baseline --gather-blocks $K 2>/dev/null | grep '^{')
  python - "$label" "$K" "$line" >> $out <<'PY'
import sys, json
j = json.loads(sys.argv[3]); print(sys.argv[1], "K=" + sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"])
PY
}
for K in 1 8; do
  run base_old DRT_TRACE_CHUNK=1024 DRT_SHADE_SUBS=0
  run default X=1
  run chunk128 DRT_TRACE_CHUNK=128
  run chunk256 DRT_TRACE_CHUNK=256
  run subs1 DRT_SHADE_SUBS=1
  run subs3 DRT_SHADE_SUBS=3
  run subs12 DRT_SHADE_SUBS=12
done
cat $out
