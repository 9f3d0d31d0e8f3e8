#!/bin/bash
for f in variants/*.so; do echo "== $f"; DRT_HIP_LIB=$PWD/$f SPP=64 BATCH=16 python3 tools/prof_workload.py 2>&1 | tail -1; done
