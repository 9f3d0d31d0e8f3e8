"""Fixed workload for rocprofv3: cornell_plane_light 1024^2, depth 8, SPP samples in batches of BATCH."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
spp = int(os.environ.get("SPP", "16")); batch = int(os.environ.get("BATCH", "8")); size = int(os.environ.get("SIZE", "1024"))
depth = int(os.environ.get("DEPTH", "8"))
max_wl = float(os.environ.get("MAX_WL", "720"))  # 695: a 64-sample grid, the shade kernel without its tail pass
spheres = int(os.environ.get("SPHERES", "0"))  # > 0: BASELINE config 5's generator instead of the Cornell box
bundle = pydrt.synthetic_sphere_scene(spheres, size, size) if spheres else pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), size, size, min_wl=380.0, max_wl=max_wl, wl_interval=5.0)
params = pydrt.make_params(size, size, spp=spp, max_depth=depth, seed=1, batch_spp=batch)
r = pydrt.Renderer(bundle, params)
r.render(0, batch); r.synchronize(); r.reset_film()
t0 = time.time(); r.render(0, spp); r.synchronize(); t1 = time.time()
st = r.stats()
print("workload %dx%d spp %d batch %d depth %d: wall %.2f ms, %.1f Mpaths/s, trace %.2f ms, shade %.2f ms" % (
    size, size, spp, batch, depth, (t1 - t0) * 1e3, size * size * spp / (t1 - t0) / 1e6, st.trace_ms, st.shade_ms))
r.close()
