#!/bin/bash
# The one-shot call (drt_render_tile, host buffers, fresh process each) against the number of row blocks it goes out in and the
# samples per kernel pair its record pool is sized for.   bash tools/r03_oneshot_blocks.sh
mkdir -p gpurun_out/r03_oneshot
for rep in 1 2; do
for blocks in 1 2 4 8; do
for batch in 0 64 128 256; do
  echo "blocks $blocks: $(DRT_ONESHOT_BLOCKS=$blocks DRT_TEST_FLAGS=2 timeout -k 10 120 python tools/oneshot_batch.py $batch 1024 256 2>&1 | grep -v amdgpu.ids)"
done
done
done > gpurun_out/r03_oneshot/sweep.txt 2>&1
cat gpurun_out/r03_oneshot/sweep.txt
