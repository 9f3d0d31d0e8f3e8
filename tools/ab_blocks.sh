#!/bin/bash
# block count x stream count sweep (N=1 emulation of the per-rank tile sizes of N>1)
set -e
out=gpurun_out/ab_blocks.txt; : > $out
for cfg in "1 1" "2 2" "4 2" "8 1" "8 2" "16 1" "16 2" "16 4" "32 1" "32 2" "32 4"; do
  set -- $cfg
  line=$(timeout -k 10 240 python bench.py --no-cpu-baseline --gather-blocks $1 --block-streams $2 2>/dev/null | grep '^{')
  python - "$1" "$2" "$line" >> $out <<'PY'
import sys, json
j = json.loads(sys.argv[3]); print("K=%s streams=%s" % (sys.argv[1], sys.argv[2]), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"])
PY
done
cat $out
