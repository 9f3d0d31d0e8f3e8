"""Per-sample and per-vertex cost of the shade kernel: the Cornell frame at depth 1, 2, 3, 4, 8, 16 on the 64-sample grid (main pass
alone) and the reference's 69-sample grid; shade ms against shaded vertices per path is a line whose intercept is the per-sample cost."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
scene = os.environ.get("SCENE", "cornell_plane_light.scn")  # SCENE=cornell_large_box.scn: BASELINE config 3's closed box (long paths)
depths = [int(d) for d in os.environ.get("DEPTHS", "1,2,3,4,8,16").split(",")]
for max_wl in (695.0, 720.0):
    b = pydrt.load_scene(os.path.join(REPO, "scenes", scene), 1024, 1024, min_wl=380.0, max_wl=max_wl, wl_interval=5.0)
    for depth in depths:
        p = pydrt.make_params(1024, 1024, spp=128, max_depth=depth, seed=1, batch_spp=64)
        r = pydrt.Renderer(b, p)
        r.render(0, 64); r.synchronize(); r.reset_film()
        r.render(0, 128); r.synchronize()
        st = r.stats()
        n = 1024 * 1024 * 128
        print("S = %d depth %2d: scans/path %.3f shaded/path %.3f  trace %.3f ns/path  shade %.3f ns/path" % (
            b.S, depth, st.closest_hit_scans / n, st.shaded_vertices / n, st.trace_ms * 1e6 / n, st.shade_ms * 1e6 / n), flush=True)
        r.close()
