#!/bin/bash
# rocprofv3 passes over tools/prof_workload.py; outputs under gpurun_out/prof_<tag>/
set -e
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/prof_workload.py > $OUT/plain.log 2>&1
cat $OUT/plain.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/prof_workload.py > $OUT/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc1 -- python3 tools/prof_workload.py > $OUT/pmc1.log 2>&1 || echo "pmc1 failed"
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc2 -- python3 tools/prof_workload.py > $OUT/pmc2.log 2>&1 || echo "pmc2 failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 tools/prof_workload.py > $OUT/pmc3.log 2>&1 || echo "pmc3 failed"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc4 -- python3 tools/prof_workload.py > $OUT/pmc4.log 2>&1 || echo "pmc4 failed"
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
