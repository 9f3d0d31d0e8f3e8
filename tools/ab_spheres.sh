#!/bin/bash
# A/B of prebuilt library variants (variants/*.so) on BASELINE config 5 (10k spheres, 4096^2, 64 spp)
for lib in daily-ray-trace_amd/libdrt_hip.so variants/*.so; do
  echo -n "$lib "; DRT_HIP_LIB=$PWD/$lib ONLY=5 timeout -k 10 300 python tools/run_configs.py 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l[0] == chr(123)][0]); print(j['Mpaths_per_s'], 'trace', j['trace_ms'], 'shade', j['shade_ms'])"
done
