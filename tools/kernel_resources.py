#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy figures of libdrt_hip.so's kernels, from hipcc's own resource remarks
(-Rpass-analysis=kernel-resource-usage). Compiles csrc/drt_launcher.hip to an object in /tmp; extra arguments are passed on
(e.g. -DSHADE_PREFETCH_DEPTH=3).   python3 tools/kernel_resources.py [filter substring] [-D...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "daily-ray-trace_amd")
args = [a for a in sys.argv[1:] if a.startswith("-")]
flt = [a for a in sys.argv[1:] if not a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c++17",
       "-I" + os.path.join(REPO, "include"), "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(PKG, "csrc", "drt_launcher.hip"),
       "-o", "/tmp/drt_resources.o"] + args
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|TotalSGPRs|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" ")[0]] = v
print("%-60s %6s %6s %8s %6s %5s" % ("kernel", "VGPRs", "SGPRs", "scratch", "LDS", "occ"))
for name, r in rows.items():
    if flt and not any(f in name for f in flt):
        continue
    print("%-60s %6s %6s %8s %6s %5s" % (name[:60], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("LDS"), r.get("Occupancy")))
