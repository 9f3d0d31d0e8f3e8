"""What ONE rank of an N-rank bench renders, timed alone on one GPU: the row-cyclic share of the 1024^2 x 256 spp frame (rows r, r + N, ...)
in the row blocks bench.py would use -- the compute side of the strong-scaling curve (the gather is not in it).
usage: rank_share_probe.py [N ...]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt, torch
stream = torch.cuda.Stream()  # one stream for all of a rank's blocks, as in bench.py
size, spp, depth = 1024, 256, 8
bundle = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), size, size)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    rows = size // n
    max_tile = rows * size
    n_blocks = max(2, min(8, max_tile // (128 * 1024))) if n > 1 else 1
    per = (rows + n_blocks - 1) // n_blocks
    ctxs = []
    for b in range(n_blocks):
        j0 = b * per
        h = min(per, rows - j0)
        p = pydrt.make_params(size, size, spp=spp, max_depth=depth, seed=1, y0=0 + n * j0, tile_h=h, row_stride=n, batch_spp=pydrt.BATCH_RESIDENT)
        ctxs.append(pydrt.Renderer(bundle, p))
        ctxs[-1].set_stream(stream.cuda_stream)
    best = 1e9
    for rep in range(4):
        for r in ctxs: r.reset_film()
        t0 = time.perf_counter()
        for r in ctxs: r.render(0, spp)
        for r in ctxs: r.synchronize()
        best = min(best, time.perf_counter() - t0)
    sts = [r.stats() for r in ctxs]
    k = "trace %.2f + shade %.2f" % (sum(st.trace_ms for st in sts), sum(st.shade_ms for st in sts))
    print("N=%d: %d rows in %d block(s): %.2f ms (kernels %s) -> ideal 1/N of the one-GPU frame: %.2f ms; compute efficiency %.3f" % (
        n, rows, n_blocks, best * 1e3, k, 155.9 / n, 155.9 / n / (best * 1e3)), flush=True)
    for r in ctxs: r.close()
