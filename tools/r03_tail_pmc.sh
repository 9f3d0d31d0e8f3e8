#!/bin/bash
# PMC probe of the shade kernel's tail pass ALONE (DRT_DEBUG_SHADE_MODE=2) and main pass alone (=1) on the headline frame
OUT=$PWD/gpurun_out/r03_tail_pmc
mkdir -p $OUT
export TMPDIR=/tmp SPP=256 BATCH=256
for mode in 2 1; do
  export DRT_DEBUG_SHADE_MODE=$mode
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/a$mode -- python3 tools/prof_workload.py > $OUT/a$mode.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/b$mode -- python3 tools/prof_workload.py > $OUT/b$mode.log 2>&1
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/c$mode -- python3 tools/prof_workload.py > $OUT/c$mode.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
for mode in ("2", "1"):
    agg = collections.defaultdict(float)
    for f in glob.glob(os.path.join(sys.argv[1], "?" + mode, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "drt_shade_kernel" in row["Kernel_Name"]:
                agg[row["Counter_Name"]] += float(row["Counter_Value"] or 0)
    print("mode", mode, "(2 = tail pass only, 1 = main pass only)")
    for k in sorted(agg): print("   %-32s %.4g" % (k, agg[k]))
PY
