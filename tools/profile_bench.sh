#!/bin/bash
# rocprofv3 over a bench command: kernel trace + stats, then separate PMC passes (counters are never mixed with tracing).
# Outputs under gpurun_out/prof_bench_<tag>/; tools/roofline_from_profiles.py folds them into <out>/roofline.json, which is
# copied to profiles/roofline.json (headline) or profiles/roofline_<workload>.json.
#   bash tools/profile_bench.sh r03                       (the headline workload; on the GPU box, from the repo root)
#   bash tools/profile_bench.sh r03_config5 --workload config5
TAG=${1:-r03}
shift
EXTRA="$@"
OUT=$PWD/gpurun_out/prof_bench_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-oneshot $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || echo "trace pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || echo "write failed"
# vector-pipe busy time against the SIMD cycles available, and the share of active lanes (same pass: the quotients need it)
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq1 -- $CMD > $OUT/pmc_sq1.log 2>&1 || echo "sq1 failed"
# what the vector instructions are: f64 adds / multiplies / FMAs / transcendentals; LDS and memory instructions
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq2 -- $CMD > $OUT/pmc_sq2.log 2>&1 || echo "sq2 failed"
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1 || true
python3 tools/roofline_from_profiles.py $OUT --bench-log $OUT/trace.log \
    --source "rocprofv3 passes of tools/profile_bench.sh $TAG over: $CMD" --out $OUT/roofline.json > /dev/null 2>$OUT/roofline.err || echo "roofline summary failed"
grep "^{" $OUT/trace.log | tail -1 > $OUT/bench_line.json
tail -3 $OUT/trace.log | cut -c1-400
