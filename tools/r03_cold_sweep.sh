#!/bin/bash
# The cold one-shot call (first big drt_render_tile of a process) against the samples per kernel pair, i.e. the size of the
# record pool the process touches for the first time. Each run is a fresh process; DRT_TIMING=1 prints the stages.
mkdir -p gpurun_out/r03_cold
for rep in 1 2; do
for batch in 0 4 8 16 32 64; do
  DRT_TEST_FLAGS=2 DRT_TIMING=1 timeout -k 10 120 python tools/oneshot_batch.py $batch 1024 256 2>&1 | grep -v amdgpu.ids
done
done > gpurun_out/r03_cold/sweep.txt 2>&1
cat gpurun_out/r03_cold/sweep.txt
