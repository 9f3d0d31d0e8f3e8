"""One-shot drt_render_tile (host buffers) wall time, each run in a FRESH process. Device memory a process touches for the
first time is cleared by the driver (10-40 ms/GB, varies from run to run), so record-buffer size matters to a short job.
  python tools/oneshot_batch.py            -> sweep;  python tools/oneshot_batch.py BATCH SIZE SPP -> one run"""
import os, sys, time, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
    import pydrt
    batch, size, spp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    b = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), size, size)
    pydrt.render_tile(b, pydrt.make_params(64, 64, spp=1, max_depth=8, seed=1))  # runtime + code object warm-up
    flags = int(os.environ.get("DRT_TEST_FLAGS", "0"))
    t0 = time.time(); px, av, va, st = pydrt.render_tile(b, pydrt.make_params(size, size, spp=spp, max_depth=8, seed=1, batch_spp=batch, flags=flags)); t1 = time.time()
    print("size %d spp %d batch %3d flags %d: one-shot %.0f ms (device kernels %.0f ms) -> %.1f Mpaths/s" % (size, spp, batch, flags, (t1 - t0) * 1e3, st.total_ms, size * size * spp / (t1 - t0) / 1e6), flush=True)
else:
    for rep in range(3):
        for batch in (0, 64):
            for flags in (0, 2):
                os.environ["DRT_TEST_FLAGS"] = str(flags)
                subprocess.run([sys.executable, os.path.abspath(__file__), str(batch), "1024", "256"], env=dict(os.environ, DRT_TIMING="1"))
