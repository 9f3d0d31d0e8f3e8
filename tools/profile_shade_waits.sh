#!/bin/bash
# Where the shade kernel's waves wait: PMC passes over the fixed workload with the 69-sample grid (main + tail pass) and with a
# 64-sample grid (main pass alone). bash tools/profile_shade_waits.sh [lib.so]
LIB=${1:-daily-ray-trace_amd/libdrt_hip.so}
export TMPDIR=/tmp SIZE=1024 SPP=64 BATCH=64 DRT_HIP_LIB=$PWD/$LIB
for wl in 695 720; do
  OUT=$PWD/gpurun_out/prof_waits_$wl
  rm -rf $OUT; mkdir -p $OUT
  export MAX_WL=$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_workload.py > $OUT/trace.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmcA -- python3 tools/prof_workload.py > $OUT/pmcA.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $OUT/pmcB -- python3 tools/prof_workload.py > $OUT/pmcB.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INSTS_VSKIPPED --output-format csv -d $OUT/pmcC -- python3 tools/prof_workload.py > $OUT/pmcC.log 2>&1
  python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  echo "== max_wl $wl"; grep -h "workload" $OUT/trace.log; grep -A9 "shade_kernel" $OUT/summary.txt | grep -v "^--"
done
