import os, sys
REPO = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
for step in (4.0, 2.5, 1.5):
    b = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), 1024, 1024, min_wl=380.0, max_wl=720.0, wl_interval=step)
    p = pydrt.make_params(1024, 1024, spp=64, max_depth=8, seed=1, batch_spp=64)
    r = pydrt.Renderer(b, p)
    r.render(0, 64); r.synchronize(); r.reset_film()
    r.render(0, 64); r.synchronize()
    st = r.stats()
    print("S = %d: trace %.1f ms, shade %.1f ms" % (b.S, st.trace_ms, st.shade_ms), flush=True)
    r.close()
