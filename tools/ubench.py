import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import numpy as np, pydrt
os.environ["DRT_VERBOSE"] = "1"
n = 256 * 256 * 32
a = np.full(n, 0.5); b = np.full(n, 0.999)
for op in (5, 6, 5, 6):
    pydrt.selftest_arith(op, a, b)
