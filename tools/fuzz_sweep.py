"""One-off wider sweep of tests/fuzz_scenes.py on the GPU against the oracle (the test suite pins 56 seeds; this runs hundreds).
usage: fuzz_sweep.py FIRST LAST [SIZE SPP DEPTH]   -- seeds >= 100 are the crowded (BVH) kind"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("daily-ray-trace_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(REPO, p))
import numpy as np, pydrt, oracle_py as O, fuzz_scenes
first, last = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    fuzz_scenes.FUZZ_SIZE, fuzz_scenes.FUZZ_SPP, fuzz_scenes.FUZZ_DEPTH = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
bad = []
t0 = time.time()
for seed in range(first, last + 1):
    b, p = fuzz_scenes.load(seed, pydrt)
    p.flags = pydrt.FLAG_RECORD_HITS
    r = pydrt.Renderer(b, p); r.render(); hits = r.read_hit_indices(int(p.spp)); px, av, va = r.read_film(); st = r.stats(); r.close()
    opx, oav, ova, ohits, ost = O.oracle_render_tile(b, p, want_hits=True, math_mode=O.MATH_DEVICE, num_threads=8)
    ok = np.array_equal(hits, ohits) and st.rng_draws == ost.rng_draws and fuzz_scenes.same(px, opx, 1e-12) and fuzz_scenes.same(va, ova, 1e-12) and fuzz_scenes.same(av, oav, 1e-12)
    if not ok:
        bad.append(seed)
        print("MISMATCH seed", seed, "hits differ:", int((hits != ohits).any(axis=1).sum()), flush=True)
    if (seed - first) % 500 == 499: print("... seed %d, %d mismatches, %.1f s" % (seed, len(bad), time.time() - t0), flush=True)
print("seeds %d..%d: %d scenes, %d mismatches %s, %.1f s" % (first, last, last - first + 1, len(bad), bad, time.time() - t0))
