#!/bin/bash
# quick A/B of knobs on the GPU box
export DRT_VERBOSE=1
for bpc in 1 2 4 8; do echo "== shade blocks/CU $bpc"; DRT_SHADE_BLOCKS_PER_CU=$bpc SPP=8 BATCH=8 python3 tools/prof_workload.py 2>&1 | tail -2; done
for bpc in 1 2 4 8; do echo "== trace blocks/CU $bpc"; DRT_TRACE_BLOCKS_PER_CU=$bpc SPP=8 BATCH=8 python3 tools/prof_workload.py 2>&1 | tail -1; done
for b in 1 2 4 16 32; do echo "== batch $b"; SPP=32 BATCH=$b python3 tools/prof_workload.py 2>&1 | tail -1; done
echo "== v1 library"; DRT_HIP_LIB=$PWD/ab/libdrt_hip_v1.so SPP=16 BATCH=8 python3 tools/prof_workload.py 2>&1 | tail -1
