"""PCIe-inclusive rate of the one-shot host-buffer form (drt_render_tile) on BASELINE config 2, and the drt_render program."""
import os, sys, time, subprocess
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt
b = pydrt.load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), 1024, 1024)
p = pydrt.make_params(1024, 1024, spp=256, max_depth=8, seed=1)
pydrt.render_tile(b, pydrt.make_params(1024, 1024, spp=1, max_depth=8, seed=1))
t0 = time.time(); px, av, va, st = pydrt.render_tile(b, p); t1 = time.time()
print("drt_render_tile 1024^2 x256 depth 8 with host buffers: wall %.1f ms -> %.1f Mpaths/s (device %.1f ms)" % ((t1 - t0) * 1e3, 1024 * 1024 * 256 / (t1 - t0) / 1e6, st.total_ms))
os.makedirs(os.path.join(REPO, "output"), exist_ok=True)
cfg = open(os.path.join(REPO, "config.cfg")).read().replace("output_width      800", "output_width      320").replace("output_height     600", "output_height     240")
open("/tmp/drt_test.cfg", "w").write(cfg)
r = subprocess.run([os.path.join(REPO, "daily-ray-trace_amd", "drt_render"), "/tmp/drt_test.cfg"], cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
print("drt_render exit", r.returncode); print("\n".join(r.stdout.splitlines()[-6:]))
print("spd sizes", [os.path.getsize(os.path.join(REPO, "output", f)) for f in ("output.spd", "average.spd", "variance.spd")], "expected", 40 + 320 * 240 * 70 * 8, 40 + 320 * 240 * 69 * 8)
