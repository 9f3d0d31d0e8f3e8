import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
size = int(os.environ.get("SIZE", "512")); spp = int(os.environ.get("SPP", "4")); n = int(os.environ.get("NSPHERES", "10000"))
b = pydrt.synthetic_sphere_scene(n, size, size)
p = pydrt.make_params(size, size, spp=spp, max_depth=8, seed=1)
r = pydrt.Renderer(b, p); r.render(0, 1); r.synchronize(); r.reset_film()
t0 = time.time(); r.render(0, spp); r.synchronize(); t1 = time.time(); st = r.stats()
print("spheres %d, %dx%d x%d spp depth 8: wall %.1f ms -> %.2f Mpaths/s | trace %.1f ms shade %.1f ms | scans/path %.3f shaded/path %.3f" % (
    n, size, size, spp, (t1 - t0) * 1e3, size * size * spp / (t1 - t0) / 1e6, st.trace_ms, st.shade_ms, st.closest_hit_scans / st.paths, st.shaded_vertices / st.paths))
