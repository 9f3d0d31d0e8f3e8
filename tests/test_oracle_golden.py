"""CPU: the oracle restatement against the committed golden vectors.

tests/golden/*.npz were produced by oracle/make_golden.py from the REAL reference path compiled from
the reference's own sources (oracle/_ref). REFERENCE math mode of the oracle is compared bit-for-bit
where only IEEE + - * / sqrt are involved, and to 1e-12 where libm (sin/cos/pow/exp) takes part, since
glibc picks CPU-specific variants of those at run time.
"""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import oracle_py as O
import pydrt

f64p = C.POINTER(C.c_double)


def p(a):
    return a.ctypes.data_as(f64p)


@pytest.fixture(scope="module")
def L():
    O.set_math_mode(O.MATH_REFERENCE)
    return O.oracle_lib()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_intersectors_bit_exact(L, golden_dir):
    g = load(golden_dir, "unit_geometry.npz")
    n = len(g["sph_t"])
    sph = np.array([L.drt_oracle_line_sphere(p(g["sph_o"][i].copy()), p(g["sph_d"][i].copy()), p(g["sph_c"][i].copy()), float(g["sph_r"][i]))
                    for i in range(n)])
    assert np.array_equal(sph, g["sph_t"])
    pl = np.array([L.drt_oracle_line_plane(p(g["pl_o"][i].copy()), p(g["pl_d"][i].copy()), p(g["pl_p"][i].copy()), p(g["pl_n"][i].copy()),
                                           p(g["pl_u"][i].copy()), p(g["pl_v"][i].copy())) for i in range(n)])
    assert np.array_equal(pl, g["pl_t"])
    # the analytic answers of the reference's own visual test (src/test.c:160-256)
    assert g["sph_t"][150] == 0.0 and np.all(g["pl_t"][900:907] == 1.0)
    assert np.isinf(g["pl_t"][910:930]).all()  # rays parallel to the plane


def test_reflect_transmit_rotations_bit_exact(L, golden_dir):
    g = load(golden_dir, "unit_geometry.npz")
    n = len(g["rf_ir"])
    out = np.zeros(3); m = np.zeros(9); z = np.array([0.0, 0.0, 1.0])
    for i in range(n):
        L.drt_oracle_reflect(p(g["rf_v"][i].copy()), p(g["rf_n"][i].copy()), p(out))
        assert np.array_equal(out, g["rf_reflect"][i])
        L.drt_oracle_transmit(p(g["rf_v"][i].copy()), p(g["rf_n"][i].copy()), float(g["rf_ir"][i]), float(g["rf_tr"][i]), p(out))
        assert np.array_equal(out, g["rf_transmit"][i], equal_nan=True)
        L.drt_oracle_rotation_between(p(z), p(g["rot_w"][i].copy()), p(m))
        assert np.array_equal(m, g["rot_m"][i])
        L.drt_oracle_rotation_about_axis(p(g["rot_w"][i].copy()), float(g["rax_angle"][i]), p(m))
        np.testing.assert_allclose(m, g["rax_m"][i], rtol=1e-13, atol=1e-15)
    assert np.isnan(g["rf_transmit"]).any()  # total internal reflection cases are in the set
    assert np.array_equal(g["rot_m"][0], -np.eye(3).ravel())  # antiparallel -> -I


def test_rng_and_shape_samplers(L, golden_dir):
    g = load(golden_dir, "unit_sampling.npz")
    out = np.zeros(3)
    for i in range(len(g["keys"])):
        L.drt_oracle_seed_path(int(g["keys"][i]))
        assert L.drt_oracle_get_rng_state() == int(g["state"][i])
        assert L.drt_oracle_rng() == g["first"][i]
        L.drt_oracle_uniform_sample_sphere(p(out))
        np.testing.assert_allclose(out, g["sphere"][i], rtol=1e-13, atol=1e-16)
        L.drt_oracle_uniform_sample_disc(p(out))
        np.testing.assert_allclose(out, g["disc"][i], rtol=1e-13, atol=1e-16)
        assert L.drt_oracle_get_rng_state() == int(g["state_after"][i])
    assert 0.0 <= g["first"].min() and g["first"].max() <= 1.0


def test_spectral_pieces(L, golden_dir):
    g = load(golden_dir, "unit_spectral.npz")
    bundle = pydrt.load_scene(cases.scene_path("cornell_plane_light.scn"), 64, 64)
    S = bundle.S
    assert np.array_equal(bundle.spds()[:11], g["tables"])  # host CSV resampling == the tables the fixtures used
    xyz = np.zeros(3)
    for i in range(len(g["rgbs"])):
        L.drt_oracle_spectrum_to_xyz(C.byref(bundle.scene), p(g["rgb_spd"][i].copy()), p(xyz))
        assert np.array_equal(xyz, g["xyz"][i])
    out = np.zeros(S)
    for i, cth in enumerate(g["cosines"]):
        L.drt_oracle_fs_dielectric_reflectance(p(g["vac"].copy()), p(g["glass"].copy()), float(cth), S, p(out))
        assert np.array_equal(out, g["diel_r"][i])
        L.drt_oracle_fs_dielectric_reflectance(p(g["glass"].copy()), p(g["vac"].copy()), float(cth), S, p(out))
        assert np.array_equal(out, g["diel_r_inside"][i])
        L.drt_oracle_fs_conductor_reflectance(p(g["vac"].copy()), p(g["au_n"].copy()), p(g["au_k"].copy()), float(cth), S, p(out))
        assert np.array_equal(out, g["cond_r"][i])
    assert (g["diel_r_inside"] == 1.0).any()  # total internal reflection branch exercised
    for i, wl in enumerate(g["wls"]):
        assert L.drt_oracle_value_at_wl(C.byref(bundle.scene), p(g["glass"].copy()), float(wl)) == g["value_at_wl"][i]
    for i in range(len(g["ggx"])):
        a = L.drt_oracle_ggx(p(g["ggx_sn"][i].copy()), p(g["ggx_mn"][i].copy()), float(g["ggx_rough"][i]))
        b = L.drt_oracle_ggx_att(p(g["ggx_v"][i].copy()), p(g["ggx_sn"][i].copy()), p(g["ggx_mn"][i].copy()), float(g["ggx_rough"][i]))
        assert a == g["ggx"][i] and b == g["ggx_att"][i]


def test_host_rgb_blackbody_camera_match_reference(golden_dir):
    """Host-side scene build (product code in host/) against the reference's rgb_f64_to_spectrum,
    generate_blackbody_spectrum and init_camera."""
    g = load(golden_dir, "unit_spectral.npz")
    H = pydrt.host_lib()
    S = g["tables"].shape[1]
    tables = np.ascontiguousarray(g["tables"][4:11])
    out = np.zeros(S)
    for i in range(len(g["rgbs"])):
        H.drt_host_rgb_to_spectrum(p(tables), S, p(g["rgbs"][i].copy()), p(out))
        assert np.array_equal(out, g["rgb_spd"][i])
    for i, t in enumerate(g["temps"]):
        H.drt_host_blackbody_spectrum(380.0, 5.0, S, float(t), p(out))
        np.testing.assert_allclose(out, g["blackbody"][i], rtol=1e-13)
    # the known-answer statistics of the reference's RGB->SPD->RGB round trip (src/test.c:45-142)
    np.testing.assert_allclose(g["roundtrip"][2:5], [0.132717, 0.208796, 0.123103], atol=5e-7)
    np.testing.assert_allclose(g["roundtrip"][5:8], [0.042695, 0.066948, 0.041327], atol=5e-7)
    gg = load(golden_dir, "unit_geometry.npz")
    for ci, co in zip(gg["cam_in"], gg["cam_out"]):
        cam = pydrt.init_camera(ci[0:3], ci[3:6], ci[6], ci[7], ci[8], ci[9], ci[10], int(ci[11]), int(ci[12]))
        np.testing.assert_allclose(np.frombuffer(bytes(cam), dtype=np.float64), co, rtol=1e-13, atol=1e-15)


def test_bdsfs_and_direction_samplers(L, golden_dir):
    g = load(golden_dir, "unit_bdsf.npz")
    bundle = pydrt.load_scene(cases.scene_path("cornell_plane_light.scn"), 64, 64)
    S = bundle.S
    sc = C.byref(bundle.scene)
    out = np.zeros(S); d = np.zeros(3); pdf = C.c_double()
    n = len(g["points"])
    seen_carry_over = False
    for i in range(n):
        pt = g["points"][i]
        m = g["materials"][i]
        P = O.make_point(pt[0:3], pt[3:6], pt[6:9], pt[9], int(m[0]), int(m[1]), int(m[2]), pt[10])
        rin = g["random_in"][i].copy()
        for b in range(7):
            out[:] = 0.0
            L.drt_oracle_bdsf_func(sc, b, C.byref(P), p(rin), p(out))
            np.testing.assert_allclose(out, g["per_func"][i, b], rtol=1e-12, atol=1e-300, err_msg="bdsf %d point %d" % (b, i))
        L.drt_oracle_bdsf(sc, C.byref(P), p(rin), p(out))
        np.testing.assert_allclose(out, g["sum_random"][i], rtol=1e-12, atol=1e-300)
        mat = bundle.scene.materials[int(m[0])]
        L.drt_oracle_set_rng_state(int(g["rng_state"][i]))
        L.drt_oracle_dir_func(sc, int(mat.dir_func), C.byref(P), p(d), C.byref(pdf))
        np.testing.assert_allclose(d, g["sampled_dir"][i], rtol=1e-12, atol=1e-15, equal_nan=True)
        np.testing.assert_allclose(pdf.value, g["sampled_pdf"][i], rtol=1e-12, equal_nan=True)
        assert L.drt_oracle_get_rng_state() == int(g["state_after"][i])
        sd = g["sampled_dir"][i].copy()  # feed the REFERENCE's direction so the exact-equality tests see the same bits
        L.drt_oracle_bdsf(sc, C.byref(P), p(sd), p(out))
        np.testing.assert_allclose(out, g["sum_sampled"][i], rtol=1e-12, atol=1e-300, equal_nan=True)
        for dfn in range(6):
            L.drt_oracle_set_rng_state(int(g["rng_state"][i]))
            L.drt_oracle_dir_func(sc, dfn, C.byref(P), p(d), C.byref(pdf))
            np.testing.assert_allclose(d, g["all_dirs"][i, dfn], rtol=1e-12, atol=1e-15, equal_nan=True)
            np.testing.assert_allclose(pdf.value, g["all_pdfs"][i, dfn], rtol=1e-12, equal_nan=True)
        L.drt_oracle_set_rng_state(int(g["rng_state"][i]))
        L.drt_oracle_direct_light(sc, C.byref(P), p(out))
        np.testing.assert_allclose(out, g["direct"][i], rtol=1e-12, atol=1e-300)
        assert L.drt_oracle_get_rng_state() == int(g["direct_state"][i])
        # quirk Q1: a dielectric evaluated at its sampled reflection returns 2R (carry-over), at transmission T
        if bundle.material_names()[int(m[0])] == "dielectric" and np.isfinite(g["sum_sampled"][i]).all():
            r = g["per_func"][i]  # not the sampled direction; just make sure the sums are non-trivial
            seen_carry_over = seen_carry_over or g["sum_sampled"][i].max() > 0
    assert seen_carry_over


@pytest.mark.parametrize("name", list(cases.RENDER_CASES))
def test_render_matches_reference_film(L, golden_dir, name):
    g = load(golden_dir, "render_%s.npz" % name)
    bundle, params = cases.load_case(name)
    px, av, va, hits, st = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_REFERENCE)
    S = bundle.S
    if "hits" in g.files:
        assert np.array_equal(hits, g["hits"]), "closest-hit surface indices differ from the reference"
    assert np.array_equal(px[:, S], g["filter"])
    step = max(1, px.shape[0] // 16)
    for got, key in ((px[::step], "pix_sample"), (av[::step], "avg_sample"), (va[::step], "var_sample")):
        assert cases.rel_err(got, g[key]) <= 1e-12
    for got, key in ((px[:, :S].sum(axis=1), "pix_sum"), (av.sum(axis=1), "avg_sum"), (va.sum(axis=1), "var_sum")):
        assert cases.rel_err(got, g[key]) <= 1e-12
    xyz = O.oracle_film_to_xyz(bundle, px)
    assert cases.xyz_rel_err(xyz, g["xyz"]) <= 1e-10
    if "pixels" in g.files:
        for got, key in ((px, "pixels"), (av, "avgs"), (va, "vars")):
            assert cases.rel_err(got, g[key]) <= 1e-12
    # DEVICE arithmetic (what the GPU runs) stays within 1e-9 of the reference and follows the same paths
    px2, _, _, hits2, st2 = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_DEVICE)
    assert np.array_equal(hits2, hits)
    assert cases.xyz_rel_err(O.oracle_film_to_xyz(bundle, px2), g["xyz"]) <= 1e-9
    assert (st.paths, st.rng_draws) == (st2.paths, st2.rng_draws)
    O.set_math_mode(O.MATH_REFERENCE)


@pytest.mark.parametrize("name", list(cases.NAN_CASES))
def test_render_matches_reference_film_nan_for_nan(L, golden_dir, name):
    """example_scene.scn (cases.NAN_CASES): the reference's film for a camera that is all NaN, kept whole in the fixture."""
    import fuzz_scenes
    g = load(golden_dir, "render_%s.npz" % name)
    bundle, params = cases.load_case(name)
    for mode in (O.MATH_REFERENCE, O.MATH_DEVICE):
        px, av, va, hits, st = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=mode)
        assert fuzz_scenes.same(px, g["pixels"]) and fuzz_scenes.same(av, g["avgs"]) and fuzz_scenes.same(va, g["vars"])
        assert (hits[:, 0] == -1).all() and st.closest_hit_scans == st.paths
    O.set_math_mode(O.MATH_REFERENCE)


def test_threads_and_tiles_reproduce_single_thread(L):
    """Per-path seeding makes the result independent of traversal order: rows split over threads, and a
    row-cyclic pair of tiles, give the bits of the single-thread full-frame render."""
    bundle, params = cases.load_case("plane_light_48")
    px, av, va, hits, st = O.oracle_render_tile(bundle, params, want_hits=True)
    px4, av4, va4, hits4, st4 = O.oracle_render_tile(bundle, params, want_hits=True, num_threads=4)
    assert np.array_equal(px, px4) and np.array_equal(av, av4) and np.array_equal(va, va4) and np.array_equal(hits, hits4)
    assert st.rng_draws == st4.rng_draws
    w, h = int(params.width), int(params.height)
    S = bundle.S
    full = px.reshape(h, w, S + 1)
    for r in range(2):
        pt = pydrt.make_params(w, h, spp=int(params.spp), max_depth=int(params.max_depth), seed=int(params.seed), y0=r, tile_h=h // 2,
                               row_stride=2)
        tp, _, _, _, _ = O.oracle_render_tile(bundle, pt)
        assert np.array_equal(tp.reshape(h // 2, w, S + 1), full[r::2])


def test_device_and_reference_arithmetic_fork_no_path():
    """How often does the kernels' arithmetic (IEEE f64 throughout, the path's own sincos) send a path somewhere else than
    the reference's (x87 long double wherever PI appears, libm sin/cos)? Counted here on BASELINE config 1 in full
    (init_cornell 256x256, 4 spp, depth 4: 262 144 paths) and on 2.1 M paths of the bench scene (cornell_plane_light
    512x512, 8 spp, depth 8): a path FORKS when its hit-index sequence differs between the oracle's two arithmetic modes
    (REFERENCE mode is bit-identical to the compiled reference: tests/test_oracle_vs_reference.py). Measured: 0 of
    2 359 296 paths, same number of rng draws; the films differ by <= 1e-12 of the brightest value."""
    total = forked = 0
    for scene, size, spp, depth in (("init_cornell.scn", 256, 4, 4), ("cornell_plane_light.scn", 512, 8, 8)):
        bundle = pydrt.load_scene(cases.scene_path(scene), size, size)
        params = pydrt.make_params(size, size, spp=spp, max_depth=depth, seed=1)
        rp, _, _, rh, rs = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_REFERENCE, num_threads=8)
        dp, _, _, dh, ds = O.oracle_render_tile(bundle, params, want_hits=True, math_mode=O.MATH_DEVICE, num_threads=8)
        forked += int((rh != dh).any(axis=1).sum())
        total += int(rs.paths)
        assert rs.rng_draws == ds.rng_draws and rs.closest_hit_scans == ds.closest_hit_scans
        assert cases.rel_err(dp, rp) <= 1e-11
    print("forked paths: %d of %d" % (forked, total))
    assert total == 256 * 256 * 4 + 512 * 512 * 8 and forked == 0
