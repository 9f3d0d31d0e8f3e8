"""Debug aid: one fuzz scene on the GPU against the oracle; prints the first differing paths. usage: fuzz_debug.py SEED"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("daily-ray-trace_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(REPO, p))
import numpy as np, pydrt, oracle_py as O, fuzz_scenes
seed = int(sys.argv[1])
b, p = fuzz_scenes.load(seed, pydrt)
p.flags = pydrt.FLAG_RECORD_HITS
r = pydrt.Renderer(b, p); r.render(); hits = r.read_hit_indices(int(p.spp)); px, av, va = r.read_film(); r.close()
opx, oav, ova, ohits, ost = O.oracle_render_tile(b, p, want_hits=True, math_mode=O.MATH_DEVICE)
bad = np.argwhere((hits != ohits).any(axis=1)).ravel()
print("seed", seed, "surfaces", b.scene.num_surfaces, "paths differing:", len(bad), "of", len(hits), "env", {k: v for k, v in os.environ.items() if k.startswith("DRT_")})
for i in bad[:6]:
    print("  path", i, "gpu", hits[i], "oracle", ohits[i])
    d = int(np.argmax(hits[i] != ohits[i]))
    for idx in (hits[i][d], ohits[i][d]):
        if idx >= 0:
            s = b.scene.surfaces[idx]
            print("     surface", idx, "type", s.type, "pos", list(s.position), "radius", s.radius, "normal", list(s.normal))
