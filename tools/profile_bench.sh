#!/bin/bash
# rocprofv3 over the default bench command; outputs under gpurun_out/prof_bench_<tag>/
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || echo "write failed"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $CMD > $OUT/pmc_sq1.log 2>&1 || echo "sq1 failed"
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1 || echo "sq2 failed"
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
tail -3 $OUT/trace.log
