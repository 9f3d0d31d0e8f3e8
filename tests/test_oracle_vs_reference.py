"""CPU, only where oracle/_ref was built (this container): the oracle against the REAL reference code,
bit for bit, at sizes the golden files do not hold (incl. BASELINE config 1 in full: 256x256, 4 spp, depth 4)."""
import numpy as np
import pytest

import cases
import fuzz_scenes
import oracle_py as O
import pydrt

pytestmark = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("name", list(cases.RENDER_CASES))
def test_film_bit_identical(name):
    bundle, params = cases.load_case(name)
    rp, ra, rv = O.ref_render_tile(bundle, params)
    op, oa, ov, _, _ = O.oracle_render_tile(bundle, params, math_mode=O.MATH_REFERENCE)
    assert np.array_equal(rp, op) and np.array_equal(ra, oa) and np.array_equal(rv, ov, equal_nan=True)


@pytest.mark.parametrize("name", list(cases.NAN_CASES))
def test_film_bit_identical_nan_for_nan(name):
    """example_scene.scn: a camera without fov / fdepth / flength makes every ray NaN (cases.NAN_CASES); the reference's film is
    NaN in every wavelength and the filter sums still count the samples -- and so is the oracle's, NaN for NaN."""
    bundle, params = cases.load_case(name)
    rp, ra, rv = O.ref_render_tile(bundle, params)
    op, oa, ov, _, st = O.oracle_render_tile(bundle, params, math_mode=O.MATH_REFERENCE)
    assert fuzz_scenes.same(op, rp) and fuzz_scenes.same(oa, ra) and fuzz_scenes.same(ov, rv)
    assert np.isnan(rp[:, :bundle.S]).all() and np.all(rp[:, bundle.S] == float(params.spp)) and st.shaded_vertices == 0


def test_config1_full_size_bit_identical():
    """BASELINE.json configs[0]: init_cornell.scn 256x256, 4 spp, depth 4, single thread, fixed seed."""
    bundle = pydrt.load_scene(cases.scene_path("init_cornell.scn"), 256, 256)
    params = pydrt.make_params(256, 256, spp=4, max_depth=4, seed=1)
    rp, ra, rv = O.ref_render_tile(bundle, params)
    op, oa, ov, _, st = O.oracle_render_tile(bundle, params, math_mode=O.MATH_REFERENCE)
    assert st.paths == 256 * 256 * 4
    assert np.array_equal(rp, op) and np.array_equal(ra, oa) and np.array_equal(rv, ov)


def test_hit_sequences_and_replay_identity():
    """The hit-index log comes from replaying cast_ray with the reference's own functions; the replay must give
    the bits of the reference's real cast_ray, and the oracle must log the same surfaces."""
    bundle, params = cases.load_case("plane_light_16")
    hits, replay, real = O.ref_trace_hits(bundle, params)
    assert np.array_equal(replay, real)
    _, _, _, ohits, _ = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_REFERENCE)
    assert np.array_equal(hits, ohits)
    assert (hits[:, 0] >= -1).all() and (hits == -1).any() and (hits >= 0).any()


def test_traversal_order_independence_of_the_reference():
    """with per-path seeding the reference's own sample_scene gives the same spectrum whether the
    pixels are visited forward or in reverse."""
    import ctypes as C
    bundle, params = cases.load_case("plane_light_16")
    R = O.ref_lib()
    R.ref_set_scene(C.byref(bundle.scene))
    S = bundle.S
    filt = C.c_double()
    fwd = np.zeros((16 * 16, S)); rev = np.zeros((16 * 16, S))
    order = [(x, y) for y in range(16) for x in range(16)]
    for buf, seq in ((fwd, order), (rev, order[::-1])):
        for (x, y) in seq:
            R.ref_sample_scene(C.byref(bundle.camera), C.byref(params), x, y, 2, buf[y * 16 + x].ctypes.data_as(C.POINTER(C.c_double)), C.byref(filt))
    assert np.array_equal(fwd, rev)


@pytest.mark.parametrize("seed", fuzz_scenes.FUZZ_SEEDS)
def test_random_scenes_bit_identical(seed):
    """Random scenes (tests/fuzz_scenes.py: every BDSF and sampler, mixed lights, nested media, lens cameras): the oracle in
    reference arithmetic against the compiled reference, film and hit indices, NaN for NaN."""
    bundle, params = fuzz_scenes.load(seed, pydrt)
    rp, ra, rv = O.ref_render_tile(bundle, params)
    op, oa, ov, ohits, _ = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_REFERENCE)
    assert fuzz_scenes.same(op, rp) and fuzz_scenes.same(oa, ra) and fuzz_scenes.same(ov, rv)
    if bundle.camera.aperture_radius == 0.0:  # the harness's hit log replays pinhole paths only (oracle/ref_harness.c)
        hits, replay, real = O.ref_trace_hits(bundle, params)
        assert fuzz_scenes.same(replay, real) and np.array_equal(hits, ohits)


@pytest.mark.parametrize("name", list(cases.degenerate_scenes()))
def test_degenerate_geometry_and_materials_bit_identical(name):
    """tests/cases.py degenerate_scenes() -- coincident surfaces, radius 0 and negative, planes with parallel or zero edges, a 900 km
    sphere, roughness 0 (NaN films), the camera inside glass: whatever the reference's arithmetic makes of them, the oracle in reference
    arithmetic makes the same, film and hit indices, NaN for NaN."""
    bundle = pydrt.load_scene_text(cases.degenerate_scenes()[name], 24, 24)
    params = pydrt.make_params(24, 24, spp=4, max_depth=8, seed=3)
    rp, ra, rv = O.ref_render_tile(bundle, params)
    op, oa, ov, ohits, _ = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_REFERENCE)
    assert fuzz_scenes.same(op, rp) and fuzz_scenes.same(oa, ra) and fuzz_scenes.same(ov, rv)
    if bundle.camera.aperture_radius == 0.0:  # the harness's hit log replays pinhole paths only (oracle/ref_harness.c)
        hits, replay, real = O.ref_trace_hits(bundle, params)
        assert fuzz_scenes.same(replay, real) and np.array_equal(hits, ohits)


@pytest.mark.parametrize("S", list(fuzz_scenes.FUZZ_GRIDS))
def test_random_scenes_on_other_wavelength_grids_bit_identical(S):
    """The same comparison on grids from 2 to 256 wavelengths (incl. grids that end below the 630 nm the dielectric
    sampler looks up, src/daily_ray_trace.c:381)."""
    bundle, params = fuzz_scenes.load(3 + S % 7, pydrt, fuzz_scenes.FUZZ_GRIDS[S])
    assert bundle.S == S
    rp, ra, rv = O.ref_render_tile(bundle, params)
    op, oa, ov, _, _ = O.oracle_render_tile(bundle, params, math_mode=O.MATH_REFERENCE)
    assert fuzz_scenes.same(op, rp) and fuzz_scenes.same(oa, ra) and fuzz_scenes.same(ov, rv)


def test_host_scene_build_pieces_on_random_inputs():
    """The product's host code (host/drt_scene.c, host/drt_spectrum.c) against the reference's own init_camera,
    rgb_f64_to_spectrum, generate_blackbody_spectrum and spectrum_to_rgb on random inputs, on the reference grid and on a
    10 nm one. rgb -> spectrum is bit for bit; the camera and the black body go through tan / expl / powl and are held to 1e-13."""
    import ctypes as C
    H, R = pydrt.host_lib(), O.ref_lib()
    f64p = C.POINTER(C.c_double)
    p = lambda a: a.ctypes.data_as(f64p)
    H.drt_host_spectrum_to_rgb.argtypes = [f64p, C.c_uint32, C.c_double, f64p, f64p]
    r = np.random.default_rng(99)
    for wl_interval in (5.0, 10.0):
        bundle = pydrt.load_scene(cases.scene_path("cornell_plane_light.scn"), 32, 32, wl_interval=wl_interval)
        S = bundle.S
        sc = bundle.scene
        block = np.ctypeslib.as_array(sc.spds, shape=(sc.num_spds, S))
        tables = np.ascontiguousarray(np.stack([block[sc.cmf_rw], block[sc.cmf_x], block[sc.cmf_y], block[sc.cmf_z]] +
                                               [block[sc.cmf_z + 1 + k] for k in range(7)]))  # host/drt_scene.c: 11 adjacent rows
        R.ref_set_grid(S, 380.0, wl_interval)
        R.ref_set_tables(p(tables))
        rgb_tables = np.ascontiguousarray(tables[4:11])
        a, b = np.zeros(S), np.zeros(S)
        for _ in range(200):
            rgb = r.uniform(0.0, 1.2, 3)
            if r.random() < 0.3:
                rgb[int(r.integers(0, 3))] = rgb[int(r.integers(0, 3))]  # ties between channels pick the branches' edges
            H.drt_host_rgb_to_spectrum(p(rgb_tables), S, p(rgb.copy()), p(a))
            R.ref_rgb_to_spectrum(p(rgb.copy()), p(b))
            assert np.array_equal(a, b)
            t = float(r.uniform(800.0, 12000.0))
            H.drt_host_blackbody_spectrum(380.0, wl_interval, S, t, p(a))
            R.ref_blackbody(t, p(b))
            np.testing.assert_allclose(a, b, rtol=1e-13)
            spd = r.uniform(0.0, 3.0, S)
            rgb_h, rgb_r = np.zeros(3), np.zeros(3)
            H.drt_host_spectrum_to_rgb(p(np.ascontiguousarray(tables[0:4])), S, wl_interval, p(spd), p(rgb_h))
            R.ref_spectrum_to_rgb(p(spd.copy()), p(rgb_r))
            assert np.array_equal(rgb_h, rgb_r)
    for _ in range(300):
        pos = r.uniform(-10, 10, 3); tgt = r.uniform(-10, 10, 3)
        roll, fov = float(r.uniform(-180, 180)), float(r.uniform(10, 140))
        fdepth, flength, aperture = float(r.uniform(0.5, 20)), float(r.uniform(0.05, 2)), float(r.choice([0.0, r.uniform(0.01, 0.5)]))
        w, h = int(r.integers(1, 3000)), int(r.integers(1, 3000))
        cam_h = pydrt.init_camera(pos, tgt, roll, fov, fdepth, flength, aperture, w, h)
        cam_r = pydrt.Camera()
        R.ref_init_camera(C.byref(cam_r), p(pos.copy()), p(tgt.copy()), roll, fov, fdepth, flength, aperture, w, h)
        np.testing.assert_allclose(np.frombuffer(bytes(cam_h), dtype=np.float64), np.frombuffer(bytes(cam_r), dtype=np.float64),
                                   rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("name", ["plane_light_16", "gold_mirror", "lights", "init_cornell"])
def test_integration_stub_round_trips_a_scene(name):
    """INTEGRATION.md's reference-side binding (include/drt_reference_binding.inc, compiled here against the reference's own
    headers as a maintainer would add it to src/daily_ray_trace.c): the reference's scene_data / camera_data / config go in,
    the boundary structs of include/drt_hip.h come out and reach drt_render_tile -- here the oracle behind that signature.
    The film must be, bit for bit, the film of the scene the boundary structs came from, i.e. flattening loses nothing."""
    import ctypes as C
    bundle, params = cases.load_case(name)
    R = O.ref_lib()
    R.ref_set_scene(C.byref(bundle.scene))
    S = bundle.S
    n = int(params.width) * int(params.height)
    seen = {}

    def render_tile(sc, cam, p, px, av, va, st):
        sc, cam, p = sc.contents, cam.contents, p.contents
        seen.update(num_spds=int(sc.num_spds), num_materials=int(sc.num_materials), num_surfaces=int(sc.num_surfaces),
                    cmf=(int(sc.cmf_rw), int(sc.cmf_x), int(sc.cmf_y), int(sc.cmf_z)), flags=int(p.flags), spp=int(p.spp),
                    base=int(sc.base_material), escape=int(sc.escape_material))
        b2 = pydrt.SceneBundle(sc, cam)
        opx, oav, ova, _, ost = O.oracle_render_tile(b2, p, math_mode=O.MATH_REFERENCE)
        for dst, src in ((px, opx), (av, oav), (va, ova)):
            C.memmove(dst, src.ctypes.data, src.nbytes)
        st.contents.paths = ost.paths
        return 0

    px, av, va = np.zeros((n, S + 1)), np.zeros((n, S)), np.zeros((n, S))
    p1 = pydrt.make_params(int(params.width), int(params.height), spp=int(params.spp), max_depth=int(params.max_depth), seed=1,
                           pixel_scheme=int(params.pixel_scheme))
    rc = R.ref_run_binding(C.byref(bundle.camera), C.byref(p1), O.RENDER_TILE_FN(render_tile), O._ptr(px), O._ptr(av), O._ptr(va))
    assert rc == 0
    sc = bundle.scene
    assert seen["num_materials"] == sc.num_materials and seen["num_surfaces"] == sc.num_surfaces
    assert seen["num_spds"] == 11 + sc.num_spds + 1 and seen["cmf"] == (0, 1, 2, 3)  # tables, the scene's rows, the zero row
    assert seen["base"] == sc.base_material and seen["escape"] == sc.escape_material
    assert seen["flags"] == pydrt.FLAG_FILM_ZERO and seen["spp"] == int(params.spp)
    opx, oav, ova, _, _ = O.oracle_render_tile(bundle, p1, math_mode=O.MATH_REFERENCE)
    assert np.array_equal(px, opx) and np.array_equal(av, oav) and np.array_equal(va, ova)
    # and the reference's own pixel loop on the same scene agrees (the stub replaces exactly that loop)
    rp, ra, rv = O.ref_render_tile(bundle, p1)
    assert np.array_equal(px, rp) and np.array_equal(av, ra) and np.array_equal(va, rv)
    # a failing launcher takes the stub's exit(-1) path
    assert R.ref_run_binding(C.byref(bundle.camera), C.byref(p1), O.RENDER_TILE_FN(lambda *a: -7), O._ptr(px), O._ptr(av), O._ptr(va)) != 0


def test_integration_md_shows_the_compiled_stub():
    """The code block in INTEGRATION.md is the file that is compiled above, character for character."""
    import os
    stub = open(os.path.join(cases.REPO, "include", "drt_reference_binding.inc")).read().strip()
    doc = open(os.path.join(cases.REPO, "INTEGRATION.md")).read()
    assert stub in doc
