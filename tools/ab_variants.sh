#!/bin/bash
# A/B of prebuilt library variants (variants/*.so, built with -D overrides) on the bench workload
for lib in daily-ray-trace_amd/libdrt_hip.so $(ls variants/*.so 2>/dev/null); do
  DRT_HIP_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-oneshot 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l[0] == chr(123)][0]); print('$lib', j['value'], j['roofline']['kernel_ms_per_step'])"
done
