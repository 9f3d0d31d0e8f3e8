"""Generate tests/golden/*.npz from the REAL reference path (oracle/_ref/libdrt_ref.so).

Run in a container where /root/reference exists:  make -C oracle ref && python oracle/make_golden.py [render case ...]
The fixtures are data only (inputs + the reference's outputs); they let the oracle be checked on
machines where the reference cannot be built (the GPU box). Every array is float64/int32/uint64,
loaded with numpy.load(allow_pickle=False).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(REPO, "daily-ray-trace_amd"), HERE, os.path.join(REPO, "tests")]
import pydrt  # noqa: E402
import oracle_py as O  # noqa: E402
import cases  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
f64p = C.POINTER(C.c_double)


def p(a):
    return a.ctypes.data_as(f64p)


def v3(a):
    return (C.c_double * 3)(*[float(x) for x in a])


def unit_vectors(rng, n):
    v = rng.normal(size=(n, 3))
    return v / np.sqrt((v * v).sum(axis=1))[:, None]


def gen_geometry(R):
    rng = np.random.default_rng(101)
    n = 1000
    # ---- ray / sphere: random + tangent, inside, behind, exact-hit edge cases
    o = rng.uniform(-4, 4, (n, 3)); d = unit_vectors(rng, n); c = rng.uniform(-3, 3, (n, 3)); r = rng.uniform(0.1, 2.0, n)
    o[:50] = c[:50] + d[:50] * 0.3 * r[:50, None]          # origin inside the sphere
    o[50:100] = c[50:100] + d[50:100] * 3.0 * r[50:100, None]  # sphere behind the ray
    t = np.cross(d[100:150], unit_vectors(rng, 50)); t /= np.sqrt((t * t).sum(axis=1))[:, None]
    o[100:150] = c[100:150] + t * r[100:150, None] - d[100:150] * 2.0    # tangent rays
    o[150] = (0, 0, 1); d[150] = (0, 0, -1); c[150] = (0, 0, 0); r[150] = 1.0  # test.c's orthographic case
    sph = np.array([R.ref_line_sphere(p(o[i]), p(d[i]), p(c[i]), float(r[i])) for i in range(n)])
    # ---- ray / plane: random + parallel, on-boundary, behind
    po = rng.uniform(-3, 3, (n, 3)); pu = po + rng.uniform(-3, 3, (n, 3)); pv = po + rng.uniform(-3, 3, (n, 3))
    U = np.zeros((n, 3)); V = np.zeros((n, 3)); N = np.zeros((n, 3))
    for i in range(n):
        R.ref_create_plane(p(po[i]), p(pu[i]), p(pv[i]), p(U[i]), p(V[i]), p(N[i]))
    o2 = rng.uniform(-5, 5, (n, 3)); d2 = unit_vectors(rng, n)
    hit_pts = po + U * rng.uniform(-0.2, 1.2, (n, 1)) + V * rng.uniform(-0.2, 1.2, (n, 1))
    d2[:700] = hit_pts[:700] - o2[:700]; d2[:700] /= np.sqrt((d2[:700] ** 2).sum(axis=1))[:, None]
    # exactly on the edges / corners of an axis-aligned unit plane (inclusive bounds)
    for k, (a, b) in enumerate([(0.0, 0.3), (1.0, 0.3), (0.3, 0.0), (0.3, 1.0), (0.0, 0.0), (1.0, 1.0), (0.5, 0.5)]):
        i = 900 + k
        po[i] = (-0.5, -0.5, 0.0); pu[i] = (0.5, -0.5, 0.0); pv[i] = (-0.5, 0.5, 0.0)
        R.ref_create_plane(p(po[i]), p(pu[i]), p(pv[i]), p(U[i]), p(V[i]), p(N[i]))
        o2[i] = (-0.5 + a, -0.5 + b, 1.0); d2[i] = (0.0, 0.0, -1.0)
    for i in range(910, 930):  # parallel to the plane
        d2[i] = U[i] / np.sqrt((U[i] ** 2).sum())
    pl = np.array([R.ref_line_plane(p(o2[i]), p(d2[i]), p(po[i]), p(N[i]), p(U[i]), p(V[i])) for i in range(n)])
    # ---- reflect / transmit / rotations
    vin = unit_vectors(rng, n); nn = unit_vectors(rng, n); ir = rng.uniform(1.0, 1.6, n); tr = rng.uniform(1.0, 1.6, n)
    refl = np.zeros((n, 3)); trans = np.zeros((n, 3)); rot = np.zeros((n, 9)); rax = np.zeros((n, 9))
    w = unit_vectors(rng, n); ang = rng.uniform(-4, 4, n)
    w[0] = (0, 0, -1); w[1] = (0, 0, 1)  # antiparallel / parallel to the z axis
    z = np.array([0.0, 0.0, 1.0])
    for i in range(n):
        R.ref_reflect(p(vin[i]), p(nn[i]), p(refl[i]))
        R.ref_transmit(p(vin[i]), p(nn[i]), float(ir[i]), float(tr[i]), p(trans[i]))
        R.ref_rotation_between(p(z), p(w[i]), p(rot[i]))
        R.ref_rotation_about_axis(p(w[i]), float(ang[i]), p(rax[i]))
    # ---- init_camera
    cams_in = np.array([[0, 0, 8, 0, 0, 0, 0, 90, 6, 0.3, 0, 64, 64], [0, 0, 8, 0, 0, 0, 0, 90, 8, 0.5, 0, 256, 256],
                        [0, 2.5, -1.5, 0, 1.5, -1.5, 180, 90, 8, 0.5, 0, 800, 600], [1, 2, 3, -1, 0.5, -2, 33, 55, 5, 0.2, 0.1, 1024, 768],
                        [0, 0, 30, 0, 0, -20, 0, 60, 6, 0.3, 0, 4096, 4096]], dtype=np.float64)
    cams_out = np.zeros((len(cams_in), 20))
    for i, ci in enumerate(cams_in):
        cam = pydrt.Camera()
        R.ref_init_camera(C.byref(cam), v3(ci[0:3]), v3(ci[3:6]), ci[6], ci[7], ci[8], ci[9], ci[10], int(ci[11]), int(ci[12]))
        cams_out[i] = np.frombuffer(bytes(cam), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "unit_geometry.npz"), sph_o=o, sph_d=d, sph_c=c, sph_r=r, sph_t=sph, pl_o=o2, pl_d=d2,
                        pl_p=po, pl_pu=pu, pl_pv=pv, pl_u=U, pl_v=V, pl_n=N, pl_t=pl, rf_v=vin, rf_n=nn, rf_ir=ir, rf_tr=tr,
                        rf_reflect=refl, rf_transmit=trans, rot_w=w, rot_m=rot, rax_angle=ang, rax_m=rax, cam_in=cams_in,
                        cam_out=cams_out)


def gen_sampling(R):
    rng = np.random.default_rng(202)
    n = 2000
    keys = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
    first = np.zeros(n); state = np.zeros(n, dtype=np.uint64); sph = np.zeros((n, 3)); disc = np.zeros((n, 3))
    state_after = np.zeros(n, dtype=np.uint64)
    for i in range(n):
        R.ref_seed_path(int(keys[i]))
        state[i] = R.ref_get_rng_state()
        first[i] = R.ref_rng()
        R.ref_uniform_sample_sphere(p(sph[i]))
        R.ref_uniform_sample_disc(p(disc[i]))
        state_after[i] = R.ref_get_rng_state()
    np.savez_compressed(os.path.join(OUT, "unit_sampling.npz"), keys=keys, state=state, first=first, sphere=sph, disc=disc,
                        state_after=state_after)


def gen_spectral(R, bundle):
    rng = np.random.default_rng(303)
    S = bundle.S
    spds = bundle.spds()
    R.ref_set_scene(C.byref(bundle.scene))
    rgbs = np.vstack([rng.uniform(0, 1, (40, 3)), [[0.2, 0.2, 0.8], [0.5, 0.0, 0.0], [1, 1, 1], [0, 0, 0], [0.3, 0.3, 0.3], [0.7, 0.2, 0.7]]])
    rgb_spd = np.zeros((len(rgbs), S)); xyz = np.zeros((len(rgbs), 3)); rgb_back = np.zeros((len(rgbs), 3))
    for i in range(len(rgbs)):
        R.ref_rgb_to_spectrum(p(rgbs[i]), p(rgb_spd[i]))
        R.ref_spectrum_to_xyz(p(rgb_spd[i]), p(xyz[i]))
        R.ref_spectrum_to_rgb(p(rgb_spd[i]), p(rgb_back[i]))
    temps = np.array([2000.0, 3200.0, 4000.0, 6500.0])
    bb = np.zeros((len(temps), S))
    for i, t in enumerate(temps):
        R.ref_blackbody(float(t), p(bb[i]))
    # test.c's RGB -> SPD -> RGB round trip over the 11^3 grid (src/test.c:45-142): error statistics
    grid = np.arange(0, 1.0001, 0.1)
    errs = []
    tmp = np.zeros(S); back = np.zeros(3)
    for r_ in grid:
        for g_ in grid:
            for b_ in grid:
                rgb = np.array([r_, g_, b_])
                R.ref_rgb_to_spectrum(p(rgb), p(tmp)); R.ref_spectrum_to_rgb(p(tmp), p(back))
                errs.append(np.abs(back - rgb))
    errs = np.array(errs)
    roundtrip = np.array([errs.max(), errs.mean(), *errs.max(axis=0), *errs.mean(axis=0)])
    wls = np.array([630.0, 632.5, 381.0, 555.0, 700.0])
    glass = spds[int(bundle.scene.materials[bundle.material_names().index("dielectric")].refract_spd)]
    vat = np.array([R.ref_value_at_wl(p(glass), float(w)) for w in wls])
    # Fresnel terms on the scene's real tables
    mats = bundle.material_names()
    gold = bundle.scene.materials[mats.index("gold")]
    vac = spds[int(bundle.scene.materials[mats.index("vacuum")].refract_spd)]
    au_n, au_k = spds[int(gold.refract_spd)], spds[int(gold.extinct_spd)]
    cosines = np.concatenate([rng.uniform(0, 1, 30), [0.0, 1.0, 1e-6, 0.999999]])
    d_r = np.zeros((len(cosines), S)); d_r_in = np.zeros((len(cosines), S)); c_r = np.zeros((len(cosines), S))
    for i, cth in enumerate(cosines):
        R.ref_fs_dielectric_reflectance(p(vac), p(glass), float(cth), p(d_r[i]))
        R.ref_fs_dielectric_reflectance(p(glass), p(vac), float(cth), p(d_r_in[i]))  # inside the glass: TIR
        R.ref_fs_conductor_reflectance(p(vac), p(au_n), p(au_k), float(cth), p(c_r[i]))
    n = 400
    sn = unit_vectors(rng, n); mn = unit_vectors(rng, n); vv = unit_vectors(rng, n); rough = rng.uniform(0.02, 0.9, n)
    g = np.array([R.ref_ggx(p(sn[i]), p(mn[i]), float(rough[i])) for i in range(n)])
    ga = np.array([R.ref_ggx_att(p(vv[i]), p(sn[i]), p(mn[i]), float(rough[i])) for i in range(n)])
    np.savez_compressed(os.path.join(OUT, "unit_spectral.npz"), tables=spds[:11], rgbs=rgbs, rgb_spd=rgb_spd, xyz=xyz, rgb_back=rgb_back,
                        temps=temps, blackbody=bb, roundtrip=roundtrip, wls=wls, glass=glass, value_at_wl=vat, vac=vac, au_n=au_n,
                        au_k=au_k, cosines=cosines, diel_r=d_r, diel_r_inside=d_r_in, cond_r=c_r, ggx_sn=sn, ggx_mn=mn, ggx_v=vv,
                        ggx_rough=rough, ggx=g, ggx_att=ga)


def gen_bdsf(R, bundle):
    """Every BDSF and direction sampler on random surface points of cornell_plane_light's materials, plus the
    bdsf() sums for (a) a random direction, (b) the direction the material's own sampler returns."""
    rng = np.random.default_rng(404)
    S = bundle.S
    R.ref_set_scene(C.byref(bundle.scene))
    names = bundle.material_names()
    vac = names.index("vacuum")
    rows = []
    n_per = 24
    for mi, name in enumerate(names):
        m = bundle.scene.materials[mi]
        if m.num_bdsfs == 0:
            continue
        for k in range(n_per):
            nrm = unit_vectors(rng, 1)[0]
            out = unit_vectors(rng, 1)[0]
            if np.dot(out, nrm) < 0:
                out = -out
            inside = (name == "dielectric" and k % 3 == 0)
            pt = O.make_point(rng.uniform(-2, 2, 3), nrm, out, float(np.dot(nrm, out)), mi, mi if inside else vac, vac if inside else mi)
            rows.append((mi, pt, unit_vectors(rng, 1)[0], int(rng.integers(1, 2 ** 62))))
    n = len(rows)
    pts = np.zeros((n, 14)); rnd_in = np.zeros((n, 3)); states = np.zeros(n, dtype=np.uint64)
    mat_idx = np.zeros((n, 3), dtype=np.int32)
    per_func = np.zeros((n, pydrt.DRT_MAX_BDSFS if False else 7, S))  # each BDSF applied to a zeroed result, random direction
    sum_rnd = np.zeros((n, S)); sum_smp = np.zeros((n, S)); smp_dir = np.zeros((n, 3)); smp_pdf = np.zeros(n)
    state_after = np.zeros(n, dtype=np.uint64)
    all_dirs = np.zeros((n, 6, 3)); all_pdfs = np.zeros((n, 6)); direct = np.zeros((n, S)); direct_state = np.zeros(n, dtype=np.uint64)
    for i, (mi, pt, rin, st) in enumerate(rows):
        pts[i, 0:3] = pt.position[:]; pts[i, 3:6] = pt.normal[:]; pts[i, 6:9] = pt.out[:]; pts[i, 9] = pt.on_dot; pts[i, 10] = pt.trans_wl
        mat_idx[i] = (pt.surface_material, pt.incident_material, pt.transmit_material)
        rnd_in[i] = rin; states[i] = st
        for b in range(7):
            R.ref_bdsf_func(b, C.byref(pt), p(rnd_in[i]), p(per_func[i, b]))
        R.ref_bdsf(C.byref(pt), p(rnd_in[i]), p(sum_rnd[i]))
        m = bundle.scene.materials[mi]
        R.ref_set_rng_state(int(st))
        pdf = C.c_double()
        R.ref_dir_func(int(m.dir_func), C.byref(pt), p(smp_dir[i]), C.byref(pdf))
        smp_pdf[i] = pdf.value
        state_after[i] = R.ref_get_rng_state()
        R.ref_bdsf(C.byref(pt), p(smp_dir[i]), p(sum_smp[i]))
        for dfn in range(6):
            if dfn in (3, 4) and m.refract_spd < 0 and name != "dielectric":
                pass
            R.ref_set_rng_state(int(st))
            R.ref_dir_func(dfn, C.byref(pt), p(all_dirs[i, dfn]), C.byref(pdf))
            all_pdfs[i, dfn] = pdf.value
        # direct lighting at points placed inside the box (position matters for the shadow ray)
        R.ref_set_rng_state(int(st))
        R.ref_direct_light(C.byref(pt), p(direct[i]))
        direct_state[i] = R.ref_get_rng_state()
    np.savez_compressed(os.path.join(OUT, "unit_bdsf.npz"), points=pts, materials=mat_idx, random_in=rnd_in, rng_state=states,
                        per_func=per_func, sum_random=sum_rnd, sampled_dir=smp_dir, sampled_pdf=smp_pdf, sum_sampled=sum_smp,
                        state_after=state_after, all_dirs=all_dirs, all_pdfs=all_pdfs, direct=direct, direct_state=direct_state)


def gen_renders(R, only=None):
    for name in list(cases.RENDER_CASES) + list(cases.NAN_CASES):
        if only and name not in only:
            continue
        bundle, params = cases.load_case(name)
        px, av, va = O.ref_render_tile(bundle, params)
        hits, spec_replay, spec_real = O.ref_trace_hits(bundle, params) if bundle.camera.aperture_radius == 0.0 else (None, None, None)
        if hits is not None:
            assert np.array_equal(spec_replay, spec_real, equal_nan=True), name + ": replayed cast_ray differs from the reference's own"
        S = bundle.S
        xyz = np.zeros((px.shape[0], 3)); tmp = np.zeros(S)
        R.ref_set_scene(C.byref(bundle.scene))
        for i in range(px.shape[0]):
            tmp[:] = px[i, :S] / px[i, S]
            R.ref_spectrum_to_xyz(p(tmp), p(xyz[i]))
        full = name in ("plane_light_16",) or name in cases.NAN_CASES
        data = dict(xyz=xyz, pix_sum=px[:, :S].sum(axis=1), avg_sum=av.sum(axis=1), var_sum=va.sum(axis=1), filter=px[:, S],
                    pix_sample=px[:: max(1, px.shape[0] // 16)], avg_sample=av[:: max(1, px.shape[0] // 16)],
                    var_sample=va[:: max(1, px.shape[0] // 16)])
        if hits is not None:
            data["hits"] = hits
        if full:
            data.update(pixels=px, avgs=av, vars=va)
        np.savez_compressed(os.path.join(OUT, "render_%s.npz" % name), **data)
        print("render", name, "ok", "paths", px.shape[0] * int(params.spp))


def main():
    os.makedirs(OUT, exist_ok=True)
    R = O.ref_lib()
    bundle = pydrt.load_scene(cases.scene_path("cornell_plane_light.scn"), 64, 64)
    R.ref_set_scene(C.byref(bundle.scene))
    if len(sys.argv) > 1:  # python oracle/make_golden.py <render case> ...: only those fixtures
        gen_renders(R, only=set(sys.argv[1:]))
        return
    gen_geometry(R)
    gen_sampling(R)
    gen_spectral(R, bundle)
    gen_bdsf(R, bundle)
    gen_renders(R)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, "total %.1f KB" % (total / 1024))


if __name__ == "__main__":
    main()
