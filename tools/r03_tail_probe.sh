for lib in daily-ray-trace_amd/libdrt_hip.so; do
for cfg in "0 4 0" "1 4 0" "2 4 0" "2 2 0" "2 1 0" "2 4 1" ; do set -- $cfg
 echo -n "mode $1 blocks/CU $2 phaseA_off $3: "; DRT_DEBUG_SHADE_MODE=$1 DRT_SHADE_BLOCKS_PER_CU=$2 DRT_DEBUG_TAIL_PHASE_A_OFF=$3 SPP=256 BATCH=256 timeout -k 10 120 python tools/prof_workload.py 2>&1 | grep workload | sed 's/.*trace/trace/'
done; done
