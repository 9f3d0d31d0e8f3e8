"""All five BASELINE.json configs on one GPU at full size: throughput + path statistics (film stays on the device)."""
import os, sys, time, json
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
S = lambda n: os.path.join(REPO, "scenes", n)
configs = [
    ("1 init_cornell 256x256, 4 spp, depth 4", lambda: pydrt.load_scene(S("init_cornell.scn"), 256, 256), 256, 4, 4),
    ("2 cornell_plane_light 1024x1024, 256 spp, depth 8", lambda: pydrt.load_scene(S("cornell_plane_light.scn"), 1024, 1024), 1024, 256, 8),
    ("3 cornell_large_box 2048x2048, 1024 spp, depth 16 (one GPU, no tiling)", lambda: pydrt.load_scene(S("cornell_large_box.scn"), 2048, 2048), 2048, 1024, 16),
    ("4 cornell + smooth gold 1024x1024, 512 spp, depth 8", lambda: pydrt.load_scene(S("cornell_gold_mirror.scn"), 1024, 1024), 1024, 512, 8),
    ("5 10k spheres 4096x4096, 64 spp, depth 8", lambda: pydrt.synthetic_sphere_scene(10000, 4096, 4096), 4096, 64, 8),
]
only = os.environ.get("ONLY")
out = []
for name, load, size, spp, depth in configs:
    if only and not name.startswith(only):
        continue
    b = load()
    batch = pydrt.BATCH_RESIDENT  # long-lived context sizing (include/drt_hip.h)
    if os.environ.get("BATCH"):
        batch = int(os.environ["BATCH"])
    p = pydrt.make_params(size, size, spp=spp, max_depth=depth, seed=1, batch_spp=batch)
    r = pydrt.Renderer(b, p)
    r.render(0, min(spp, r.batch_spp())); r.synchronize(); r.reset_film()
    t0 = time.time(); r.render(0, spp); r.synchronize(); t1 = time.time()
    st = r.stats()
    row = {"config": name, "paths": st.paths, "seconds": round(t1 - t0, 3), "Mpaths_per_s": round(st.paths / (t1 - t0) / 1e6, 1),
           "trace_ms": round(st.trace_ms, 1), "shade_ms": round(st.shade_ms, 1), "scans_per_path": round(st.closest_hit_scans / st.paths, 3),
           "shaded_per_path": round(st.shaded_vertices / st.paths, 3), "batch_spp": r.batch_spp()}
    print(json.dumps(row), flush=True)
    out.append(row)
    r.close()
