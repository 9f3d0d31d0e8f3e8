"""CPU: the POSIX C host (scene / config / CSV reader, scene build) and the C-ABI library surface."""
import ctypes as C
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import cases
import pydrt

REPO = cases.REPO


def test_hip_library_exports_every_declared_symbol():
    """include/drt_hip.h is the contract: every function it declares must be exported (no compute is called)."""
    header = open(os.path.join(REPO, "include", "drt_hip.h")).read()
    declared = set(re.findall(r"\b(drt_[a-z_0-9]+)\s*\(", header)) - {"drt_context"}
    assert declared == set(pydrt.HIP_SYMBOLS), declared ^ set(pydrt.HIP_SYMBOLS)
    L = pydrt.hip_lib()
    for sym in declared:
        assert getattr(L, sym) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(REPO, "daily-ray-trace_amd", "libdrt_hip.so")],
                         stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r" T (drt_[a-z_0-9]+)", out))
    assert declared <= exported


def test_struct_layouts_match_the_header():
    assert C.sizeof(pydrt.Surface) == 4 + 4 + 24 + 8 + 72
    assert C.sizeof(pydrt.Camera) == 20 * 8
    assert C.sizeof(pydrt.Stats) == 8 * 8 + 8 + 8 + 4 + 4 + 4 + 4 + 3 * 8  # + record pool: blocks, peak, block bytes, redone launches; + kernel pairs, pad, min / max / avg sample pass
    assert C.sizeof(pydrt.Params) == 72


def test_bdsf_list_names_and_order():
    """The X-macro list keeps the reference's 7 + 6 names in order (src/bdsf_list.h)."""
    text = open(os.path.join(REPO, "include", "bdsf_list.h")).read()
    assert re.findall(r"^BDSF\((\w+)\)", text, re.M) == pydrt.BDSF_NAMES
    assert re.findall(r"^DIRF\((\w+)\)", text, re.M) == pydrt.DIRF_NAMES
    H = pydrt.host_lib()
    names = (C.c_char_p * 7).in_dll(H, "bdsf_name_list")
    assert [n.decode() for n in names] == pydrt.BDSF_NAMES
    dnames = (C.c_char_p * 6).in_dll(H, "dir_func_name_list")
    assert [n.decode() for n in dnames] == pydrt.DIRF_NAMES


def test_cornell_plane_light_scene_build():
    b = pydrt.load_scene(cases.scene_path("cornell_plane_light.scn"), 1024, 1024)
    sc = b.scene
    assert (sc.num_surfaces, sc.num_materials, sc.num_wavelengths) == (9, 12, 69)  # 11 parsed + 1 trailing zero material
    names = b.material_names()
    assert names[:3] == ["vacuum", "escape", "blue_plastic"] and names[-1] == ""
    assert sc.base_material == 0 and sc.escape_material == 1
    assert sc.materials[1].is_black_body == 1  # escape is forced black-body (src/daily_ray_trace.c:148)
    gold = sc.materials[names.index("gold")]
    assert [gold.bdsfs[i] for i in range(gold.num_bdsfs)] == [pydrt.BDSF["ct_conductor_bdsf"]]
    assert gold.dir_func == pydrt.DIRF["sample_ct_direction"] and gold.roughness == 0.1
    light = sc.materials[names.index("light")]
    assert light.is_emissive == 1 and light.is_black_body == 1
    assert b.surface_names()[5:7] == ["gold_ball", "glass_ball"]
    s = sc.surfaces[0]  # back wall: u = pointu - position, n = normalise(u x v)
    assert list(s.u) == [6.0, 0.0, 0.0] and list(s.v) == [0.0, -6.0, 0.0] and list(s.normal) == [0.0, 0.0, -1.0]
    cam = b.camera
    assert list(cam.forward) == [0.0, 0.0, -1.0] and list(cam.up) == [0.0, 1.0, 0.0] and list(cam.right) == [1.0, 0.0, 0.0]
    # gold n/k come from micrometre CSVs: scaled x1000 and interpolated onto the grid
    spds = b.spds()
    assert 0.1 < spds[gold.refract_spd].min() and spds[gold.refract_spd].max() < 2.0
    assert np.abs(spds[sc.cmf_rw] - 1.0).max() < 1e-3  # white table (not exactly 1 everywhere)


def test_csv_reader_quirks(tmp_path):
    H = pydrt.host_lib()
    out = np.zeros(5)

    def resample(text, lo=400.0, step=10.0):
        f = tmp_path / "t.csv"
        f.write_bytes(text)
        assert H.drt_host_csv_to_spectrum(str(f).encode(), lo, step, 5, out.ctypes.data_as(C.POINTER(C.c_double))) == 1
        return out.copy()

    a = resample(b"wl,v\n400,1.0\n410,2.0\n420,3.0\n430,4.0\n440,5.0\n")
    assert list(a) == [1.0, 2.0, 3.0, 4.0, 5.0]
    b = resample(b"wl,v\n0.400,1.0\n0.420,3.0\n0.440,5.0\n")  # micrometres -> x1000, linear interpolation
    np.testing.assert_allclose(b, [1.0, 2.0, 3.0, 4.0, 5.0], rtol=1e-12)
    c = resample(b"wl,v\n400,-1.0\n440,-5.0\n")  # a number starts at the first DIGIT: the sign is lost
    assert list(c) == [1.0, 2.0, 3.0, 4.0, 5.0]
    d = resample(b"wl,v\n400,1.0\n420,3.0\n\x00")  # trailing NUL (cmf_*.csv); past the end: last segment extrapolated
    np.testing.assert_allclose(d, [1.0, 2.0, 3.0, 4.0, 5.0], rtol=1e-12)
    e = resample(b"wl,v\n420,3.0\n440,5.0\n")  # below the first entry: first segment extrapolated (as the reference)
    np.testing.assert_allclose(e, [1.0, 2.0, 3.0, 4.0, 5.0], rtol=1e-12)
    assert H.drt_host_csv_to_spectrum(b"/nonexistent.csv", 400.0, 10.0, 5, out.ctypes.data_as(C.POINTER(C.c_double))) == 0


@pytest.mark.parametrize("name,roll_deg", [("init_cornell.scn", 0), ("cornell_large_box.scn", 0), ("first_scene.scn", 0),
                                            ("example_scene.scn", 0), ("cornell_downward.scn", 180)])
def test_legacy_scenes_load(name, roll_deg):
    """The five shipped scenes in the old syntax (camera up/right/forward, no bdsfs/dir_func, no vacuum/escape)."""
    b = pydrt.load_scene(cases.scene_path(name), 64, 64)
    names = b.material_names()
    assert "vacuum" in names and "escape" in names
    text = open(cases.scene_path(name)).read()
    up = [float(x) for x in re.search(r"^up\s+(.*)$", text, re.M).group(1).replace(",", " ").split()]
    right = [float(x) for x in re.search(r"^right\s+(.*)$", text, re.M).group(1).replace(",", " ").split()]
    np.testing.assert_allclose(list(b.camera.up), up, atol=1e-12)
    np.testing.assert_allclose(list(b.camera.right), right, atol=1e-12)
    for i, n in enumerate(names):
        m = b.scene.materials[i]
        if m.diffuse_spd >= 0 and not m.is_black_body:
            assert [m.bdsfs[j] for j in range(m.num_bdsfs)] == [0, 1] and m.dir_func == 0


def test_scene_errors():
    with pytest.raises(RuntimeError):
        pydrt.load_scene("/nonexistent.scn", 8, 8)
    bad = "Camera\nposition 0 0 8\ntarget 0 0 0\nfov 90\nfdepth 6\nflength 0.3\nMaterial\nname m\ndiffuse csv missing.csv\n"
    with pytest.raises(RuntimeError):
        pydrt.load_scene_text(bad, 8, 8)
    # grammar errors keep the reference's behaviour: message + exit(-1)
    code = ("import sys; sys.path.insert(0, %r); import pydrt; pydrt.load_scene_text('Camera\\nbogus 1\\n', 8, 8)" % os.path.join(REPO, "daily-ray-trace_amd"))
    r = subprocess.run(["python", "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 255 and "parse error" in r.stdout


def test_config_parser_and_spd_writer(tmp_path):
    H = pydrt.host_lib()

    class Config(C.Structure):
        _fields_ = [("num_pixel_samples", C.c_uint32), ("max_cast_depth", C.c_uint32), ("output_width", C.c_uint32),
                    ("output_height", C.c_uint32), ("min_wl", C.c_double), ("max_wl", C.c_double), ("wl_interval", C.c_double)] + \
                   [(n, C.c_char * 64) for n in ("input_scene", "output_spd", "average_spd", "variance_spd", "output_bmp", "average_bmp",
                                                  "variance_bmp", "white_spd", "cmf_x", "cmf_y", "cmf_z", "red_spd", "green_spd",
                                                  "blue_spd", "cyan_spd", "magenta_spd", "yellow_spd")] + [("pixel_scheme", C.c_int)]

    assert C.sizeof(Config) == 1136  # same size as the reference's config_arguments
    cfg = Config()
    text = open(os.path.join(REPO, "config.cfg"), "rb").read().replace(b"scenes/cornell", b"scenes\\cornell")
    H.parse_config(text, len(text), C.byref(cfg))
    assert (cfg.num_pixel_samples, cfg.max_cast_depth, cfg.output_width, cfg.output_height) == (4, 4, 800, 600)
    assert (cfg.min_wl, cfg.max_wl, cfg.wl_interval) == (380.0, 720.0, 5.0)
    assert cfg.input_scene == b"scenes/cornell_plane_light.scn" and cfg.pixel_scheme == 2  # '\' accepted, stored as '/'
    # .spd round trip: 40-byte header + pixels
    px = np.arange(2 * 3 * 5, dtype=np.float64)
    path = str(tmp_path / "o.spd").encode()
    H.drt_host_write_spd.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, C.POINTER(C.c_double)]
    assert H.drt_host_write_spd(path, 3, 2, 4, 1, 380.0, 5.0, px.ctypes.data_as(C.POINTER(C.c_double))) == 0
    raw = open(path, "rb").read()
    assert len(raw) == 40 + px.nbytes
    ident, w, h, nwl, has_filter = struct.unpack_from("<5I", raw, 0)
    assert (ident, w, h, nwl, has_filter) == (0xEDFEEFBE, 3, 2, 4, 1)
    assert struct.unpack_from("<2d", raw, 24) == (380.0, 5.0)
    assert np.array_equal(np.frombuffer(raw, dtype=np.float64, offset=40), px)


def test_product_does_not_touch_the_oracle():
    """Nothing under daily-ray-trace_amd/ or include/ may include, link, load or import anything of oracle/
    (comments may cite it)."""
    patterns = [r'#\s*include\s*["<][^">]*oracle', r"^\s*(import|from)\s+oracle", r"libdrt_oracle", r"libdrt_ref", r"import\s+oracle_py",
                r"-ldrt_oracle", r"drt_oracle_[a-z_]+\s*\("]
    bad = []
    for root in ("daily-ray-trace_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(REPO, root)):
            for f in files:
                if f.endswith((".c", ".h", ".hip", ".py", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    for pat in patterns:
                        if re.search(pat, text, re.M):
                            bad.append((os.path.join(dirpath, f), pat))
    assert bad == [], bad


def test_spd_to_bmp_postprocess(tmp_path, golden_dir):
    """N2: .spd -> linear RGB -> 32-bit bottom-up BMP, with the reference's spectrum_to_rgb_f64 as known answer."""
    H = pydrt.host_lib()
    g = np.load(os.path.join(golden_dir, "unit_spectral.npz"), allow_pickle=False)
    S = g["tables"].shape[1]
    cmf = np.ascontiguousarray(g["tables"][0:4])
    f64p = C.POINTER(C.c_double)
    H.drt_host_spectrum_to_rgb.argtypes = [f64p, C.c_uint32, C.c_double, f64p, f64p]
    rgb = np.zeros(3)
    for i in range(len(g["rgbs"])):
        H.drt_host_spectrum_to_rgb(cmf.ctypes.data_as(f64p), S, 5.0, g["rgb_spd"][i].copy().ctypes.data_as(f64p), rgb.ctypes.data_as(f64p))
        assert np.array_equal(rgb, g["rgb_back"][i])  # == the compiled reference's spectrum_to_rgb_f64
    # a 3x2 film with filter sums: pixel i carries spectrum i (pre-multiplied by the filter value 2)
    w, h = 3, 2
    film = np.zeros((w * h, S + 1))
    film[:, :S] = g["rgb_spd"][:6] * 2.0
    film[:, S] = 2.0
    H.drt_host_write_spd.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, f64p]
    spd = str(tmp_path / "o.spd").encode()
    bmp = str(tmp_path / "o.bmp").encode()
    assert H.drt_host_write_spd(spd, w, h, S, 1, 380.0, 5.0, film.ctypes.data_as(f64p)) == 0
    H.drt_host_spd_file_to_bmp.argtypes = [C.c_char_p, C.c_char_p, f64p]
    assert H.drt_host_spd_file_to_bmp(spd, bmp, cmf.ctypes.data_as(f64p)) == 0
    raw = open(bmp, "rb").read()
    assert len(raw) == 14 + 40 + w * h * 4 and raw[:2] == b"BM"
    size, off = struct.unpack_from("<I4xI", raw, 2)
    assert (size, off) == (len(raw), 54)
    hdr = struct.unpack_from("<IiiHHIIiiII", raw, 14)
    assert hdr[0:3] == (40, w, h) and hdr[4] == 32 and hdr[7:9] == (3780, 3780)
    px = np.frombuffer(raw, dtype=np.uint8, offset=54).reshape(w * h, 4)
    expect = (np.clip(g["rgb_back"][:6], 0.0, 1.0) * 255.0).astype(np.uint8)  # clamp, truncate, no gamma
    assert np.array_equal(px[:, 2], expect[:, 0]) and np.array_equal(px[:, 1], expect[:, 1]) and np.array_equal(px[:, 0], expect[:, 2])
    assert np.all(px[:, 3] == 255)


def _config_for(workdir, scene="scenes/cornell_plane_light.scn", depth=4, scheme="pixel_random"):
    """A config_arguments (1136 bytes, the reference's layout) whose outputs live under workdir."""
    H = pydrt.host_lib()
    text = ("num_pixel_samples 6\nmax_cast_depth %d\noutput_width 5\noutput_height 3\nmin_wl 380.0\nmax_wl 720.0\nwl_interval 5.0\n"
            "pixel_scheme %s\ninput_scene %s\n"
            "output_spd %s/output.spd\naverage_spd %s/average.spd\nvariance_spd %s/variance.spd\n" % (depth, scheme, scene, workdir, workdir, workdir)).encode()
    buf = C.create_string_buffer(1136)
    tb = C.create_string_buffer(text, len(text) + 1)
    H.parse_config.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p]
    H.parse_config.restype = None
    H.parse_config(tb, len(text), buf)
    return buf


def test_checkpoint_sets_are_crash_safe_and_resume_refuses_mixed_files(tmp_path):
    """A checkpoint is three generation files + a manifest that names the generation; the manifest's rename is the one
    switch, so a kill at any moment leaves the previous complete checkpoint or the new one, and the loader
    refuses anything inconsistent with the files or with the job (size, grid, seed, depth, pixel scheme, scene file)."""
    import shutil
    H = pydrt.host_lib()
    f64p = C.POINTER(C.c_double)
    H.drt_host_write_outputs.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_double, C.c_double, f64p, f64p, f64p,
                                         C.c_int, C.c_uint32, C.c_uint64]
    H.drt_host_load_checkpoint.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, f64p, f64p, f64p,
                                           C.POINTER(C.c_uint32)]
    H.drt_host_checkpoint_error.restype = C.c_char_p
    w, h, S = 5, 3, 69
    n_px = w * h
    rng = np.random.default_rng(3)
    import tempfile
    short = tempfile.mkdtemp(prefix="ck", dir="/tmp")  # config_arguments path fields hold 63 characters (src/daily_ray_trace.h:36-52)
    scene = os.path.join(short, "s.scn")
    open(scene, "w").write("Camera\n")

    def film(n):
        """a film that is consistent with n samples: sums, filter = n, mean = sum / n, variance >= 0"""
        px = np.zeros((n_px, S + 1)); px[:, :S] = rng.uniform(0, 5, (n_px, S)) * n; px[:, S] = n
        return px, px[:, :S] / n, rng.uniform(0, 2, (n_px, S))

    def write(d, n, seed=1, raw=1):
        os.makedirs(d, exist_ok=True)
        cfg = _config_for(d, scene)
        px, av, va = film(n)
        assert H.drt_host_write_outputs(cfg, w, h, S, 380.0, 5.0, px.ctypes.data_as(f64p), av.ctypes.data_as(f64p), va.ctypes.data_as(f64p), raw, n, seed) == 0
        return cfg, px, av, va

    def load(cfg, seed=1, size=(w, h)):
        a, b, c = np.zeros((size[0] * size[1], S + 1)), np.zeros((size[0] * size[1], S)), np.zeros((size[0] * size[1], S))
        done = C.c_uint32(77)
        rc = H.drt_host_load_checkpoint(cfg, size[0], size[1], S, seed, a.ctypes.data_as(f64p), b.ctypes.data_as(f64p), c.ctypes.data_as(f64p), C.byref(done))
        return rc, done.value, a, b, c, H.drt_host_checkpoint_error().decode()

    def generation(d):
        return int([l for l in open(os.path.join(d, "output.spd.ckpt")).read().splitlines() if l.startswith("generation")][0].split()[1])

    d2 = os.path.join(short, "n4")
    cfg, px, av, va = write(d2, 4)
    assert not [f for f in os.listdir(d2) if f.endswith(".tmp")]          # nothing half-written is left behind
    assert sorted(os.listdir(d2)) == ["average.spd", "average.spd.ck0", "output.spd", "output.spd.ck0", "output.spd.ckpt", "variance.spd", "variance.spd.raw.ck0"]
    rc, done, a, b, c, why = load(cfg)
    assert rc == 0 and done == 4 and np.array_equal(a, px) and np.array_equal(b, av) and np.array_equal(c, va)
    # the reference's outputs under their own names hold the same film (and the normalised variance)
    assert open(os.path.join(d2, "output.spd"), "rb").read() == open(os.path.join(d2, "output.spd.ck0"), "rb").read()
    nv = np.fromfile(os.path.join(d2, "variance.spd"), dtype=np.float64, offset=40).reshape(n_px, S)
    assert np.array_equal(nv, va / va.max(axis=1, keepdims=True))
    # the next checkpoint goes to the other generation and retires this one
    cfg, px6, av6, va6 = write(d2, 6)
    assert generation(d2) == 1 and not os.path.exists(os.path.join(d2, "output.spd.ck0"))
    rc, done, a, *_ = load(cfg)
    assert rc == 0 and done == 6 and np.array_equal(a, px6)
    # a kill while the NEXT checkpoint's files are being written (generation 0 half there, manifest not switched yet):
    # the previous checkpoint still stands, whole
    open(os.path.join(d2, "output.spd.ck0"), "wb").write(b"half a file")
    open(os.path.join(d2, "average.spd.ck0"), "wb").write(b"")
    rc, done, a, b, c, why = load(cfg)
    assert rc == 0 and done == 6 and np.array_equal(a, px6) and np.array_equal(b, av6) and np.array_equal(c, va6)
    # ... and a kill right after the switch leaves the new one whole: the older generation's files may still lie around
    cfg, px8, *_ = write(d2, 8)
    assert generation(d2) == 0
    open(os.path.join(d2, "output.spd.ck1"), "wb").write(b"left over")
    rc, done, a, *_ = load(cfg)
    assert rc == 0 and done == 8 and np.array_equal(a, px8)
    os.remove(os.path.join(d2, "output.spd.ck1"))
    # files of two checkpoints mixed under one manifest: mean of N1 = 2 samples beside sums of N2 = 8
    d1 = os.path.join(short, "n2")
    write(d1, 2)
    shutil.copy(os.path.join(d1, "average.spd.ck0"), os.path.join(d2, "average.spd.ck0"))
    rc, done, *_, why = load(cfg)
    assert rc != 0 and done == 0 and "does not belong" in why
    # the whole older set under a newer manifest: the filter sums give it away
    shutil.rmtree(d2)
    cfg, *_ = write(d2, 4)
    shutil.copy(os.path.join(d1, "output.spd.ck0"), os.path.join(d2, "output.spd.ck0"))
    rc, done, *_, why = load(cfg)
    assert rc != 0 and "manifest says 4" in why
    # no manifest, no resume
    shutil.rmtree(d2)
    cfg, *_ = write(d2, 4)
    os.remove(os.path.join(d2, "output.spd.ckpt"))
    rc, done, *_, why = load(cfg)
    assert rc != 0 and "manifest" in why
    # a final write without the raw variance retires the checkpoint it overwrote
    shutil.rmtree(d2)
    cfg, *_ = write(d2, 4)
    write(d2, 6, raw=0)
    assert sorted(os.listdir(d2)) == ["average.spd", "output.spd", "variance.spd"] and load(cfg)[0] != 0
    # other job: another seed, size, depth, pixel scheme, scene file; a truncated file; a file of another size with a forged header
    cfg, *_ = write(d2, 4)
    assert load(cfg, seed=2)[0] != 0 and "seed" in load(cfg, seed=2)[5]
    assert load(cfg, size=(3, 5))[0] != 0
    assert "max_cast_depth" in load(_config_for(d2, scene, depth=5))[5]
    assert "pixel scheme" in load(_config_for(d2, scene, scheme="pixel_center"))[5]
    open(scene, "a").write("# edited\n")
    assert "another scene" in load(cfg)[5]
    cfg, *_ = write(d2, 4)
    assert load(cfg)[0] == 0
    raw_path = os.path.join(d2, "variance.spd.raw.ck%d" % generation(d2))
    data = open(raw_path, "rb").read()
    open(raw_path, "wb").write(data[:-8])
    rc, _, _, _, _, why = load(cfg)
    assert rc != 0 and "bytes" in why
    open(raw_path, "wb").write(data)
    assert load(cfg)[0] == 0
    hdr = np.frombuffer(data[:40], dtype=np.uint32).copy(); hdr[1] = 1 << 30; hdr[2] = 1 << 30   # a header that announces 2^60 pixels
    open(raw_path, "wb").write(hdr.tobytes() + data[40:])
    H.drt_host_read_spd.argtypes = [C.c_char_p, C.c_void_p, C.POINTER(f64p)]
    out = f64p()
    assert H.drt_host_read_spd(raw_path.encode(), C.create_string_buffer(40), C.byref(out)) != 0 and not out
    assert load(cfg)[0] != 0
    shutil.rmtree(short)


def test_bvh_depth_stays_within_the_traversal_stack():
    """the traversal pushes unchecked, one entry per level at most, so the BUILT tree's depth is what keeps it
    safe. Host-side check (drt_bvh_stats builds the same tree drt_create would, no GPU needed) on the scenes that stress the
    builder: the 10k-sphere config, collinear centres, thousands of identical surfaces (nothing for the SAH to separate),
    and an exponential cluster (the SAH peels one surface per level until its level cap hands over to median splits)."""
    def spheres(centres, radius=0.1):
        surf = [{"type": pydrt.GEO_SPHERE, "material": 1, "position": [float(c[0]), float(c[1]), float(c[2])], "radius": radius} for c in centres]
        mats = [{"is_black_body": 1}, {"diffuse_spd": 0, "bdsfs": [0], "dir_func": 0}]
        cam = pydrt.init_camera([0, 0, 30], [0, 0, 0], 0.0, 60.0, 6.0, 0.3, 0.0, 8, 8)
        return pydrt.build_scene(surf, mats, np.ones((4, 69)), 0, 0, cam)

    rng = np.random.default_rng(5)
    cases_ = {
        "config 5": pydrt.synthetic_sphere_scene(10000, 8, 8),
        "collinear": spheres(np.stack([np.linspace(-50, 50, 5000), np.zeros(5000), np.zeros(5000)], axis=1)),
        "identical": spheres(np.zeros((3000, 3))),
        "exponential": spheres(np.stack([2.0 ** -np.arange(0, 400, 0.25) * 40, rng.uniform(-1e-6, 1e-6, 1600), np.zeros(1600)], axis=1), radius=1e-9),
    }
    for name, bundle in cases_.items():
        nodes, in_leaves, depth, stack = pydrt.bvh_stats(bundle)
        n = sum(1 for i in range(int(bundle.scene.num_surfaces)) if bundle.scene.surfaces[i].type in (pydrt.GEO_SPHERE, pydrt.GEO_PLANE))
        assert in_leaves == n, name                   # every surface is in exactly one leaf
        assert 1 <= depth <= stack == 32, (name, depth)
        assert nodes <= max(1, n), name
