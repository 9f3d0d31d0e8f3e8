#!/bin/bash
# PMC probe of one library variant on a fixed workload (cornell 1024^2, 64 spp, one launch pair): bash tools/profile_variant.sh <lib.so> <tag>
LIB=$1; TAG=$2
OUT=$PWD/gpurun_out/prof_var_$TAG
mkdir -p $OUT
export TMPDIR=/tmp SIZE=1024 SPP=64 BATCH=64 DRT_HIP_LIB=$PWD/$LIB
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_workload.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmcA -- python3 tools/prof_workload.py > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAVES --output-format csv -d $OUT/pmcB -- python3 tools/prof_workload.py > $OUT/pmcB.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A9 "shade_kernel" $OUT/summary.txt | grep -v "^--"
