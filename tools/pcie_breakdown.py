"""Where the one-shot host-buffer call spends its time: context creation (allocations), film upload, render, film download."""
import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
import ctypes as C
b = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), 1024, 1024)
pydrt.render_tile(b, pydrt.make_params(64, 64, spp=1, max_depth=8, seed=1))  # runtime + code object warm-up
S, n = b.S, 1024 * 1024
for batch in (0, 32, 16):
    p = pydrt.make_params(1024, 1024, spp=256, max_depth=8, seed=1, batch_spp=batch)
    px = np.zeros((n, S + 1)); av = np.zeros((n, S)); va = np.zeros((n, S))
    t = [time.time()]
    r = pydrt.Renderer(b, p); t.append(time.time())
    L, f64p = r.L, C.POINTER(C.c_double)
    L.drt_write_film(r.ctx, px.ctypes.data_as(f64p), av.ctypes.data_as(f64p), va.ctypes.data_as(f64p)); t.append(time.time())
    r.render(); r.synchronize(); t.append(time.time())
    L.drt_read_film(r.ctx, px.ctypes.data_as(f64p), av.ctypes.data_as(f64p), va.ctypes.data_as(f64p)); t.append(time.time())
    bs = r.batch_spp()
    r.close(); t.append(time.time())
    d = [(t[i + 1] - t[i]) * 1e3 for i in range(5)]
    print("batch %3d: create %.0f ms, upload %.0f ms, render %.0f ms, download %.0f ms, destroy %.0f ms; total %.0f ms" % (bs, *d, sum(d)), flush=True)
t0 = time.time(); pydrt.render_tile(b, pydrt.make_params(1024, 1024, spp=256, max_depth=8, seed=1)); t1 = time.time()
print("drt_render_tile one-shot: %.0f ms -> %.1f Mpaths/s" % ((t1 - t0) * 1e3, n * 256 / (t1 - t0) / 1e6))
