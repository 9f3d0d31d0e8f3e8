#!/bin/bash
# A/B of library variants on the 10k-sphere scene: throughput of a 2048^2 x 16 spp probe and of BASELINE config 5, and the bounce
# kernel's lane activity / vector busy share from one PMC pass per variant.   bash tools/r03_spheres_ab.sh [PMC=1] lib...
OUT=$PWD/gpurun_out/r03_spheres
mkdir -p $OUT
export TMPDIR=/tmp
LIBS=${@:-daily-ray-trace_amd/libdrt_hip.so}
for lib in $LIBS; do
  tag=$(basename $lib .so)
  echo "== $lib"
  DRT_HIP_LIB=$PWD/$lib SPHERES=10000 SIZE=2048 SPP=16 BATCH=16 timeout -k 10 120 python3 tools/prof_workload.py 2>&1 | grep workload
  DRT_HIP_LIB=$PWD/$lib ONLY=5 timeout -k 10 300 python3 tools/run_configs.py 2>/dev/null | grep config
  if [ -n "$PMC" ]; then
    rm -rf $OUT/pmc_$tag
    DRT_HIP_LIB=$PWD/$lib SPHERES=10000 SIZE=2048 SPP=16 BATCH=16 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $OUT/pmc_$tag -- python3 tools/prof_workload.py > $OUT/pmc_$tag.log 2>&1
    python3 - $OUT/pmc_$tag <<'PY'
import csv, glob, os, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"] or 0)
for k, c in agg.items():
    if "bounce" in k or "primary" in k:
        a = c["SQ_ACTIVE_INST_VALU"]
        print("   %-22s lanes active %.3f  vector busy %.3f  waves waiting %.3f  VALU %.3g  SALU %.3g  VMEM_RD %.3g" % (
            k, c["SQ_THREAD_CYCLES_VALU"] / (64 * a), a * 4 / (c["SQ_BUSY_CYCLES"] * 32), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VALU"], c["SQ_INSTS_SALU"], c["SQ_INSTS_VMEM_RD"]))
PY
  fi
done
