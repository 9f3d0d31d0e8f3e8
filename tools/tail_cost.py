"""What the packed tail pass costs: the same frame with S = 64 (380..695 nm, no tail), S = 69 (the reference grid) and S = 69 with the tail
handled as a second register set (DRT_NO_TAIL_PASS=1 in the environment)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
for max_wl in (695.0, 720.0):
    b = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), 1024, 1024, min_wl=380.0, max_wl=max_wl, wl_interval=5.0)
    p = pydrt.make_params(1024, 1024, spp=256, max_depth=8, seed=1, batch_spp=64)
    r = pydrt.Renderer(b, p)
    r.render(0, 64); r.synchronize(); r.reset_film()
    r.render(0, 256); r.synchronize()
    st = r.stats()
    print("S = %d: trace %.1f ms, shade %.1f ms" % (b.S, st.trace_ms, st.shade_ms), flush=True)
    r.close()
