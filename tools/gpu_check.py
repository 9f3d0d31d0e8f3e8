"""GPU bring-up check: HIP path vs the CPU oracle on a few configs (run on the GPU box)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd")); sys.path.insert(0, os.path.join(REPO, "oracle"))
import numpy as np
import pydrt, oracle_py as O

rng = np.random.default_rng(0)
a = np.concatenate([rng.uniform(0, 100, 200000), 10.0 ** rng.uniform(-300, 300, 200000)])
b = np.concatenate([rng.uniform(-3, 3, 200000), 10.0 ** rng.uniform(-150, 150, 200000)])
print("sqrt bit-exact:", np.array_equal(pydrt.selftest_arith(0, a), np.sqrt(a)))
print("div  bit-exact:", np.array_equal(pydrt.selftest_arith(1, a, b), a / b))
t = rng.uniform(-1.0, 7.0, 400000)
sc = pydrt.selftest_arith(2, t).reshape(-1, 2)
os_, oc = np.empty_like(t), np.empty_like(t)
import ctypes as C
L = O.oracle_lib(); s_ = C.c_double(); c_ = C.c_double()
for i in range(0, t.size, 40):
    L.drt_oracle_sincos(t[i], C.byref(s_), C.byref(c_)); os_[i] = s_.value; oc[i] = c_.value
idx = np.arange(0, t.size, 40)
print("sincos bit-exact vs oracle:", np.array_equal(sc[idx, 0], os_[idx]) and np.array_equal(sc[idx, 1], oc[idx]),
      "max err vs libm", np.abs(sc[:, 0] - np.sin(t)).max(), np.abs(sc[:, 1] - np.cos(t)).max())
x = rng.uniform(0, 1, 400000); y = np.full_like(x, 100.0)
pw = pydrt.selftest_arith(3, x, y); ref = np.power(x, y)
m = ref > 1e-300
print("pow rel err vs libm:", np.max(np.abs(pw[m] - ref[m]) / ref[m]))

for (w, h, spp, depth, batch) in [(32, 32, 4, 8, 2), (64, 64, 8, 8, 3), (128, 96, 2, 4, 0)]:
    bundle = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), w, h)
    params = pydrt.make_params(w, h, spp=spp, max_depth=depth, seed=1, flags=pydrt.FLAG_RECORD_HITS, batch_spp=batch)
    t0 = time.time(); r = pydrt.Renderer(bundle, params); r.render(); px, av, va = r.read_film(); t1 = time.time()
    hits = r.read_hit_indices(spp); st = r.stats(); xyz = r.read_xyz(); r.close()
    opx, oav, ova, ohits, ost = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_DEVICE)
    oxyz = O.oracle_film_to_xyz(bundle, opx)
    print(w, h, spp, depth, "hits equal:", np.array_equal(hits, ohits), "n mismatch", int((hits != ohits).sum()),
          "pix rel", np.abs(px - opx).max() / np.abs(opx).max(), "avg rel", np.abs(av - oav).max() / np.abs(oav).max(),
          "var rel", np.abs(va - ova).max() / max(np.abs(ova).max(), 1e-300),
          "xyz rel", np.max(np.abs(xyz - oxyz) / np.maximum(np.abs(oxyz), 1e-12)),
          "bit-equal pix", np.array_equal(px, opx))
    print("   stats gpu", st.paths, st.closest_hit_scans, st.shaded_vertices, st.shadow_scans, st.rng_draws,
          "| oracle", ost.paths, ost.closest_hit_scans, ost.shaded_vertices, ost.shadow_scans, ost.rng_draws,
          "| trace ms %.3f shade ms %.3f wall %.3f" % (st.trace_ms, st.shade_ms, t1 - t0))

# throughput probe
w = h = 1024
bundle = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), w, h)
for batch in (4, 8):
    params = pydrt.make_params(w, h, spp=16, max_depth=8, seed=1, batch_spp=batch)
    r = pydrt.Renderer(bundle, params); r.render(0, batch); r.synchronize(); r.reset_film()
    t0 = time.time(); r.render(0, 16); r.synchronize(); t1 = time.time(); st = r.stats(); r.close()
    print("1024^2 x16spp batch", batch, "wall %.1f ms -> %.1f Mpaths/s | trace %.1f ms shade %.1f ms" % (
        (t1 - t0) * 1e3, w * h * 16 / (t1 - t0) / 1e6, st.trace_ms, st.shade_ms))
