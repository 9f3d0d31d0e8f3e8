"""Kernel time against launch size: rows of the headline frame spread evenly over the image (row_stride = 1024 / rows), 256 spp in ONE
kernel pair, trace and shade times from the library's HIP events -- the per-pair cost that does not scale with the paths."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
size, spp, depth = 1024, 256, 8
bundle = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), size, size)
full = None
for rows in [int(a) for a in sys.argv[1:]] or (1024, 512, 256, 128, 64, 32, 16, 8):
    p = pydrt.make_params(size, size, spp=spp, max_depth=depth, seed=1, y0=0, tile_h=rows, row_stride=size // rows, batch_spp=pydrt.BATCH_RESIDENT)
    r = pydrt.Renderer(bundle, p)
    best = (1e9, 0, 0)
    for rep in range(4):
        r.reset_film(); r.render(0, spp); r.synchronize(); st = r.stats()
        if st.total_ms < best[0]: best = (st.total_ms, st.trace_ms, st.shade_ms)
    paths = rows * size * spp / 1e6
    if full is None: full = (75.13 / 268.435456, 80.29 / 268.435456) if rows != 1024 else (best[1] / paths, best[2] / paths)
    print("%4d rows %6.1f M paths, %d launch(es): trace %7.3f ms (+%.3f over the full frame's rate), shade %7.3f ms (+%.3f)" % (
        rows, paths, st.launches, best[1], best[1] - full[0] * paths, best[2], best[2] - full[1] * paths), flush=True)
    r.close()
