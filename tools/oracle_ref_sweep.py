"""CPU, this container only (needs oracle/_ref built from /root/reference): the oracle in reference arithmetic against the COMPILED
reference over hundreds or thousands of random scenes of tests/fuzz_scenes.py (the test suite pins 56 seeds), film and hit indices,
NaN for NaN.   usage: oracle_ref_sweep.py FIRST LAST [SIZE SPP DEPTH]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("daily-ray-trace_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(REPO, p))
import numpy as np, pydrt, oracle_py as O, fuzz_scenes
if not O.ref_available():
    sys.exit("oracle/_ref is not built: make -C oracle ref (needs /root/reference)")
first, last = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    fuzz_scenes.FUZZ_SIZE, fuzz_scenes.FUZZ_SPP, fuzz_scenes.FUZZ_DEPTH = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
bad, t0 = [], time.time()
for seed in range(first, last + 1):
    b, p = fuzz_scenes.load(seed, pydrt)
    rp, ra, rv = O.ref_render_tile(b, p)
    op, oa, ov, ohits, _ = O.oracle_render_tile(b, p, want_hits=True, math_mode=O.MATH_REFERENCE)
    ok = fuzz_scenes.same(op, rp) and fuzz_scenes.same(oa, ra) and fuzz_scenes.same(ov, rv)
    if ok and b.camera.aperture_radius == 0.0:  # the harness's hit log replays pinhole paths only (oracle/ref_harness.c)
        hits, replay, real = O.ref_trace_hits(b, p)
        ok = fuzz_scenes.same(replay, real) and np.array_equal(hits, ohits)
    if not ok:
        bad.append(seed)
        print("MISMATCH seed", seed, flush=True)
    if (seed - first) % 500 == 499: print("... seed %d, %d mismatches, %.1f s" % (seed, len(bad), time.time() - t0), flush=True)
print("seeds %d..%d: %d scenes, %d mismatches %s, %.1f s" % (first, last, last - first + 1, len(bad), bad, time.time() - t0))
