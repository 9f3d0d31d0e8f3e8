#!/bin/bash
TAG=${1:-p2}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_workload.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmcA -- python3 tools/prof_workload.py > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_LEVEL_WAVES --output-format csv -d $OUT/pmcB -- python3 tools/prof_workload.py > $OUT/pmcB.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmcC -- python3 tools/prof_workload.py > $OUT/pmcC.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A10 "shade_kernel\|trace_kernel" $OUT/summary.txt | grep -v "^--"
