#!/bin/bash
# shade item split sweep at full frame (K=1, batch 64) and at the per-rank tile sizes of N = 2, 4, 8
set -e
out=gpurun_out/ab_batch.txt; : > $out
for kb in "1 64" "2 128" "4 256" "8 256"; do
set -- $kb; K=$1; B=$2
for subs in 0 1 2 3 4 6 12; do
  line=$(env DRT_SHADE_SUBS=$subs timeout -k 10 240 python bench.py --no-cpu-baseline --gather-blocks $K --batch $B 2>/dev/null | grep '^{')
  python - "$K" "$B" "$subs" "$line" >> $out <<'PY'
import sys, json
j = json.loads(sys.argv[4]); print("K=%s batch=%s subs=%s" % tuple(sys.argv[1:4]), j["value"], j["ms_per_step"], j["roofline"]["kernel_ms_per_step"])
PY
done
done
cat $out
