#!/bin/bash
# PMC probe of the trace kernel on the 10k-sphere scene (BVH path): 2048^2, 8 spp
OUT=$PWD/gpurun_out/prof_spheres
mkdir -p $OUT
export TMPDIR=/tmp SPHERES=10000 SIZE=2048 SPP=8 BATCH=8
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_workload.py > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmcA -- python3 tools/prof_workload.py > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmcB -- python3 tools/prof_workload.py > $OUT/pmcB.log 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/pmcC -- python3 tools/prof_workload.py > $OUT/pmcC.log 2>&1
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -A12 "trace_kernel\|primary_kernel\|bounce_kernel" $OUT/summary.txt | grep -v "^--"
tail -2 $OUT/trace.log
