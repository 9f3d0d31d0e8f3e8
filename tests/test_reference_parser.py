"""CPU, only where oracle/_ref was built (this container): the host loader (host/drt_scene.c, host/drt_spectrum.c)
against the reference's OWN tokenizer + parse_scene + parse_config (src/read_scene.c:1-796) and its own CSV resampling
(the body of load_csv_file_to_spectrum, :812-end), compiled from the reference's files by oracle/Makefile.

init_scene / init_spd (src/daily_ray_trace.c:79-211) need the Win32 alloc and file calls and are not built; what they do
with the parsed input is restated below in a few lines of Python over the reference's own primitives (rgb_f64_to_spectrum,
generate_blackbody_spectrum, spectrum_normalise, const_spectrum, spectral_mul_by_scalar, create_plane_from_points,
init_camera), and the host loader's finished drt_scene / drt_camera must equal that bit for bit."""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import fuzz_scenes
import oracle_py as O
import pydrt

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")

SPECTRA = os.path.join(cases.REPO, "spectra")
TABLE_CSVS = ["white_rgb_to_spd", "cmf_x", "cmf_y", "cmf_z", "white_rgb_to_spd", "red_rgb_to_spd", "green_rgb_to_spd",
              "blue_rgb_to_spd", "cyan_rgb_to_spd", "magenta_rgb_to_spd", "yellow_rgb_to_spd"]  # rw, x, y, z, white, r, g, b, c, m, y


def ref_tables(n, min_wl, interval):
    """init_spd_tables (src/spectrum.c:1-47): the 11 tables through the reference's CSV resampling."""
    R = O.ref_lib()
    R.ref_set_grid(n, min_wl, interval)
    t = np.zeros((11, n))
    for k, name in enumerate(TABLE_CSVS):
        assert R.ref_csv_to_spectrum(os.path.join(SPECTRA, name + ".csv").encode(), O._ptr(t[k])) == 1
    R.ref_set_tables(O._ptr(t))
    return t


def ref_init_spd(inp, n):
    """init_spd, src/daily_ray_trace.c:79-123, over the reference's own spectral functions. None = no spectrum."""
    R = O.ref_lib()
    s = np.zeros(n)
    if inp.method == O.SPD_METHOD_RGB:
        R.ref_rgb_to_spectrum(O._v3(inp.value), O._ptr(s))
    elif inp.method == O.SPD_METHOD_CSV:
        assert R.ref_csv_to_spectrum(os.path.join(SPECTRA, inp.csv.decode()).encode(), O._ptr(s)) == 1
    elif inp.method == O.SPD_METHOD_BLACKBODY:
        R.ref_blackbody(inp.value[0], O._ptr(s))
        R.ref_spectrum_normalise(O._ptr(s))
    elif inp.method == O.SPD_METHOD_CONST:
        R.ref_const_spectrum(O._ptr(s), inp.value[0])
    else:
        return None
    if inp.has_scale_factor:
        R.ref_spectral_mul_by_scalar(O._ptr(s), inp.scale_factor)
    return s


def check_scene_against_reference_parser(text, w, h, grid=(380.0, 720.0, 5.0)):
    R = O.ref_lib()
    rc, cam, mats, surfs = O.ref_parse_scene(text)
    assert rc == 0
    bundle = pydrt.load_scene_text(text, w, h, min_wl=grid[0], max_wl=grid[1], wl_interval=grid[2])
    sc, n = bundle.scene, bundle.S
    assert n == int(((grid[1] - grid[0]) / grid[2]) + 1.0)  # src/spectrum.c:3
    spds = bundle.spds()
    tables = ref_tables(n, grid[0], grid[2])
    assert np.array_equal(spds[:11], tables)  # the host's CSV reader == the reference's, on all 11 tables
    assert (sc.cmf_rw, sc.cmf_x, sc.cmf_y, sc.cmf_z) == (0, 1, 2, 3)
    # camera: init_camera (reference code) on the reference parser's camera block
    rcam = pydrt.Camera()
    R.ref_init_camera(C.byref(rcam), O._v3(cam[3:6]), O._v3(cam[0:3]), cam[6], cam[7], cam[8], cam[9], cam[10], w, h)
    assert bytes(rcam) == bytes(bundle.camera)
    # materials: init_scene :133-162 -- parsed + 1 trailing all-zero material, escape forced black body
    assert sc.num_materials == len(mats) + 1
    names = bundle.material_names()
    base = escape = None
    for i, m in enumerate(mats):
        hm = sc.materials[i]
        assert names[i].encode() == m.name
        assert hm.is_black_body == (1 if m.is_escape_material else m.is_black_body)
        assert hm.is_emissive == m.is_emissive
        assert hm.shininess == m.shininess and hm.roughness == m.roughness
        assert hm.num_bdsfs == m.num_bdsfs
        assert list(hm.bdsfs)[:m.num_bdsfs] == list(m.bdsfs)[:m.num_bdsfs]
        if m.dir_func >= 0:
            assert hm.dir_func == m.dir_func
        for k, field in enumerate(("emission_spd", "diffuse_spd", "glossy_spd", "mirror_spd", "refract_spd", "extinct_spd")):
            want = ref_init_spd(m.spd[k], n)
            idx = getattr(hm, field)
            if want is None:
                assert idx == -1, (m.name, field)
            else:
                assert idx >= 11 and np.array_equal(spds[idx], want), (m.name, field)
        if m.is_escape_material:
            escape = i
        if m.is_base_material:
            base = i
    last = sc.materials[len(mats)]
    assert (last.is_black_body, last.is_emissive, last.num_bdsfs) == (0, 0, 0)
    assert all(getattr(last, f) == -1 for f in ("emission_spd", "diffuse_spd", "glossy_spd", "mirror_spd", "refract_spd", "extinct_spd"))
    assert sc.base_material == base and sc.escape_material == escape
    # surfaces: init_scene :172-210 -- planes through create_plane_from_points, material by first name match (else 0)
    assert sc.num_surfaces == len(surfs)
    ref_names = [m.name for m in mats] + [b""]
    for i, s in enumerate(surfs):
        hs = sc.surfaces[i]
        assert hs.type == s.type and list(hs.position) == list(s.position)
        if s.type == pydrt.GEO_SPHERE:
            assert hs.radius == s.radius
        elif s.type == pydrt.GEO_PLANE:
            u, v, nrm = np.zeros(3), np.zeros(3), np.zeros(3)
            R.ref_create_plane(O._v3(s.position), O._v3(s.u), O._v3(s.v), O._ptr(u), O._ptr(v), O._ptr(nrm))
            assert list(hs.u) == list(u) and list(hs.v) == list(v) and list(hs.normal) == list(nrm)
        want_mat = ref_names.index(s.material_name) if s.material_name in ref_names else 0
        assert hs.material == want_mat
    return bundle


def test_cornell_plane_light_loads_like_the_reference():
    """The one shipped scene the reference's parser accepts, at the bench size and at config.cfg's."""
    text = open(cases.scene_path("cornell_plane_light.scn")).read()
    check_scene_against_reference_parser(text, 1024, 1024)
    check_scene_against_reference_parser(text, 800, 600)


def test_authored_modern_scenes_load_like_the_reference():
    """The scenes this repo authored in the current grammar (gold mirror, lights, thin lens): <= 16 materials and surfaces."""
    for name, grid in (("cornell_gold_mirror.scn", (380.0, 720.0, 5.0)), ("test_lights.scn", (380.0, 720.0, 5.0)),
                       ("test_lens.scn", (380.0, 720.0, 10.0)), ("cornell_plane_light.scn", (380.0, 720.0, 4.0))):
        text = open(cases.scene_path(name)).read()
        if text.count("Material") > 16 or text.count("Surface") > 16:
            continue
        check_scene_against_reference_parser(text, 64, 48, grid)


@pytest.mark.parametrize("seed", fuzz_scenes.FUZZ_SEEDS[:24])
def test_random_scenes_load_like_the_reference(seed):
    """Random scenes in the .scn grammar (every BDSF, sampler, SPD method, light kind) that fit the reference's fixed arrays."""
    text = fuzz_scenes.random_scene_text(seed)
    if text.count("Material") > 16 or text.count("Surface") > 16:
        pytest.skip("more than 16 materials or surfaces: the reference's parser arrays overflow (src/read_scene.h:85-86)")
    check_scene_against_reference_parser(text, 40, 30)


def test_legacy_scenes_are_rejected_by_the_reference_parser():
    """the five legacy-syntax scenes end in parse_error() -> exit(-1) in the reference (the host loads them
    through its superset grammar: tests/test_host.py)."""
    for name in ("init_cornell.scn", "cornell_large_box.scn", "cornell_downward.scn", "first_scene.scn", "example_scene.scn"):
        rc = O.ref_parse_scene(open(cases.scene_path(name)).read())[0]
        assert rc != 0, name


def test_config_cfg_parses_like_the_reference():
    """parse_config on the reference's own config.cfg (Windows `\\` paths): the same 1136-byte config_arguments, path
    separators aside (the POSIX host stores `/`)."""
    H = pydrt.host_lib()
    ref_cfg = os.path.join("/root/reference", "config.cfg")
    for path in (ref_cfg, os.path.join(cases.REPO, "config.cfg")):
        text = open(path, "rb").read()
        if b"/" in text and path != ref_cfg:
            # `/` is not a word character of the reference's tokenizer (src/read_scene.c:76-79): give it the `\` form
            ref_text = text.replace(b"/", b"\\")
        else:
            ref_text = text
        rc, raw = O.ref_parse_config(ref_text)
        assert rc == 0 and len(raw) == 1136
        buf = C.create_string_buffer(1136)
        tb = C.create_string_buffer(text, len(text) + 1)
        H.parse_config.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p]
        H.parse_config.restype = None
        H.parse_config(tb, len(text), buf)
        assert buf.raw == raw.replace(b"\\", b"/")
    assert raw[:16] == np.array([4, 4, 800, 600], dtype=np.uint32).tobytes()  # num_pixel_samples, depth, width, height


def test_csv_resampling_matches_the_reference_on_every_table():
    """load_csv_file_to_spectrum's body on every spectra/*.csv (nm and um files, NUL-terminated ones, the magenta table with
    its duplicated rows) on three grids: the host's reader gives the same samples bit for bit."""
    R = O.ref_lib()
    H = pydrt.host_lib()
    for (lo, hi, step) in ((380.0, 720.0, 5.0), (380.0, 720.0, 4.0), (400.0, 700.0, 2.5)):
        n = int(((hi - lo) / step) + 1.0)
        R.ref_set_grid(n, lo, step)
        for f in sorted(os.listdir(SPECTRA)):
            if not f.endswith(".csv"):
                continue
            p = os.path.join(SPECTRA, f).encode()
            a, b = np.zeros(n), np.zeros(n)
            assert R.ref_csv_to_spectrum(p, O._ptr(a)) == 1
            assert H.drt_host_csv_to_spectrum(p, lo, step, n, O._ptr(b)) == 1
            assert np.array_equal(a, b, equal_nan=True), (f, lo, step)
