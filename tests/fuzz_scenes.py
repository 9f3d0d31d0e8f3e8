"""Random scenes in the reference's .scn grammar, for parity tests beyond the shipped and authored scenes: every BDSF
and direction sampler, plane / sphere / point lights (several at once), overlapping spheres (nested media), thin lens
and pinhole cameras, odd shininess and roughness values. The same text goes through the C host's parser, so the loader
is exercised too. Deterministic per seed."""
import numpy as np

HEADER = """Material
name vacuum
refract constant 1.0
base_material

Material
name escape
escape_material
"""


def _rgb(r):
    return "rgb %.3f, %.3f, %.3f" % tuple(r.uniform(0.05, 0.9, 3))


def random_scene_text(seed):
    r = np.random.default_rng(seed)
    r2 = np.random.default_rng(10 ** 6 + seed)  # later additions draw from their own stream, so older seeds keep their geometry
    sky = r2.random() < 0.35                    # an emitting escape material: paths that leave the scene pick up light
    glow = r2.random() < 0.4                    # a surface that emits AND scatters (emissive, not a black body)
    out = []
    # the box first (half size h), then a camera inside it near the open/front side, looking at the middle
    h = r.uniform(3.5, 5.0)
    pos = np.array([r.uniform(-1.0, 1.0), r.uniform(-1.0, 1.0), h * r.uniform(0.75, 0.95)])
    tgt = r.uniform(-0.7, 0.7, 3)
    lens = r.random() < 0.3
    out.append("Camera\nposition %.4f, %.4f, %.4f\ntarget %.4f, %.4f, %.4f\nroll %.2f\nfov %.1f\nfdepth %.3f\nflength 0.3\naperture %.3f\n" % (
        *pos, *tgt, r.uniform(-30, 30), r.uniform(45, 95), np.linalg.norm(pos - tgt), 0.06 if lens else 0.0))
    out.append(HEADER.replace("escape_material", "emission constant %.3f\nescape_material" % r2.uniform(0.05, 0.6)) if sky else HEADER)
    mats = []

    def mat(name, body):
        mats.append(name)
        out.append("Material\nname %s\n%s\n" % (name, body))

    for k in range(3):
        mat("plastic%d" % k, "diffuse %s\nglossy %s\nshininess %s\nbdsfs bp_diffuse_bdsf, bp_glossy_bdsf\ndir_func cos_weighted_sample_hemisphere" % (
            _rgb(r), _rgb(r), r.choice(["8.0", "32.5", "100.0", "1.0"])))
    mat("matte", "diffuse %s\nbdsfs bp_diffuse_bdsf\ndir_func uniform_sample_hemisphere" % _rgb(r))
    mat("mirror", "mirror %s\nbdsfs mirror_bdsf\ndir_func sample_specular_direction" % _rgb(r))
    mat("smooth_gold", "refract csv au_spec_n.csv\nextinct csv au_spec_k.csv\nbdsfs fs_conductor_bdsf\ndir_func sample_specular_direction")
    mat("rough_gold", "refract csv au_spec_n.csv\nextinct csv au_spec_k.csv\nroughness %.3f\nbdsfs ct_conductor_bdsf\ndir_func sample_ct_direction" % r.uniform(0.05, 0.4))
    mat("glass", "refract csv glass.csv\nbdsfs fs_dielectric_reflectance_bdsf, fs_dielectric_transmittance_bdsf\ndir_func sample_reflect_or_transmit_direction")
    mat("thin_glass", "refract csv glass.csv\nbdsfs fs_dielectric_transmittance_bdsf\ndir_func sample_transmit_direction")
    mat("dense", "refract constant %.3f\nbdsfs fs_dielectric_reflectance_bdsf, fs_dielectric_transmittance_bdsf\ndir_func sample_reflect_or_transmit_direction" % r.uniform(1.1, 2.4))
    lights = []
    for k, body in enumerate(["emission blackbody %.0f scale %.2f" % (r.uniform(2500, 7000), r.uniform(1, 5)),
                              "emission %s scale %.2f" % (_rgb(r), r.uniform(1, 6)),
                              "emission constant %.3f" % r.uniform(0.05, 2.0)]):
        out.append("Material\nname light%d\n%s\nis_black_body true\n" % (k, body))
        lights.append("light%d" % k)
    if glow:
        out.append("Material\nname glow\nemission %s scale %.2f\ndiffuse %s\nglossy %s\nshininess 12.0\nbdsfs bp_diffuse_bdsf, bp_glossy_bdsf\n"
                   "dir_func cos_weighted_sample_hemisphere\n" % (_rgb(r2), r2.uniform(0.5, 3.0), _rgb(r2), _rgb(r2)))
    surf = []

    def plane(name, p, u, v, m):
        surf.append("Surface\nname %s\ntype plane\nposition %.4f, %.4f, %.4f\npointu %.4f, %.4f, %.4f\npointv %.4f, %.4f, %.4f\nmaterial %s\n" % (name, *p, *u, *v, m))

    walls = [("floor", (-h, -h, -h), (h, -h, -h), (-h, -h, h)), ("ceil", (-h, h, -h), (h, h, -h), (-h, h, h)),
             ("back", (-h, -h, -h), (h, -h, -h), (-h, h, -h)), ("left", (-h, -h, -h), (-h, -h, h), (-h, h, -h)),
             ("right", (h, -h, -h), (h, -h, h), (h, h, -h)), ("front", (-h, -h, h), (h, -h, h), (-h, h, h))]
    for name, p, u, v in walls:
        if r.random() < 0.85:
            plane(name, p, u, v, r.choice(mats[:5]))
    crowded = seed >= 100  # > 96 surfaces: the tables leave LDS and the BVH (with planes and every material in it) takes over
    for k in range(int(r.integers(100, 160)) if crowded else int(r.integers(3, 9))):
        c = r.uniform(-2.6, 2.6, 3) if crowded else r.uniform(-2.2, 2.2, 3)
        rad = r.uniform(0.08, 0.45) if crowded else r.uniform(0.3, 1.3)
        surf.append("Surface\nname ball%d\ntype sphere\nposition %.4f, %.4f, %.4f\nradius %.4f\nmaterial %s\n" % (k, *c, rad, r.choice(mats)))
    n_lights = int(r.integers(1, 4))
    for k in range(n_lights):
        kind = r.choice(["plane", "sphere", "point"])
        m = lights[int(r.integers(0, 3))]
        c = r.uniform(-2.5, 2.5, 3)
        if kind == "plane":
            e1 = r.uniform(-1, 1, 3); e2 = r.uniform(-1, 1, 3)
            plane("lamp%d" % k, c, c + e1, c + e2, m)
        elif kind == "sphere":
            surf.append("Surface\nname lamp%d\ntype sphere\nposition %.4f, %.4f, %.4f\nradius %.3f\nmaterial %s\n" % (k, *c, r.uniform(0.1, 0.5), m))
        else:
            surf.append("Surface\nname lamp%d\ntype point\nposition %.4f, %.4f, %.4f\nmaterial %s\n" % (k, *c, m))
    if glow:
        c = r2.uniform(-2.0, 2.0, 3)
        surf.append("Surface\nname glowball\ntype sphere\nposition %.4f, %.4f, %.4f\nradius %.3f\nmaterial glow\n" % (*c, r2.uniform(0.3, 0.9)))
    order = r.permutation(len(surf))  # lights and geometry interleaved: light order follows surface order in the reference
    return "\n".join(out) + "\n" + "\n".join(surf[i] for i in order)


FUZZ_SEEDS = list(range(1, 49)) + list(range(100, 108))
FUZZ_SIZE, FUZZ_SPP, FUZZ_DEPTH = 12, 3, 6


def load(seed, pydrt, grid=None):
    """grid: (min_wl, max_wl, interval) or None for the reference's 380..720 step 5"""
    kw = dict(min_wl=grid[0], max_wl=grid[1], wl_interval=grid[2]) if grid else {}
    bundle = pydrt.load_scene_text(random_scene_text(seed), FUZZ_SIZE, FUZZ_SIZE, **kw)
    params = pydrt.make_params(FUZZ_SIZE, FUZZ_SIZE, spp=FUZZ_SPP, max_depth=FUZZ_DEPTH, seed=1000 + seed)
    return bundle, params


def same(a, b, tol=0.0):
    """equal where finite (bit for bit at tol 0, else scale-relative), NaN / inf in the same places"""
    a, b = np.asarray(a), np.asarray(b)
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return False
    if not np.array_equal(np.isnan(a), np.isnan(b)) or not np.array_equal(a[~fa & ~np.isnan(a)], b[~fb & ~np.isnan(b)]):
        return False
    if tol == 0.0:
        return bool(np.array_equal(a[fa], b[fb]))
    scale = float(np.max(np.abs(b[fb]))) if fb.any() else 1.0
    return bool(np.max(np.abs(a[fa] - b[fb]), initial=0.0) <= tol * (scale if scale > 0 else 1.0))


# wavelength grids by sample count S: lane sets x tail widths of the shade kernel (S = 64k + r: r <= 16 is a packed tail)
# (every grid brackets 630 nm, which the dielectric sampler looks up: the reference reads past the array otherwise and
# drt_create refuses such a grid)
FUZZ_GRIDS = {S: (400.0, 400.0 + (S - 1) * step, step) for S, step in
              ((2, 250.0), (5, 60.0), (63, 5.0), (64, 5.0), (65, 5.0), (70, 4.0), (80, 4.0), (81, 4.0), (128, 2.0), (129, 2.0),
               (144, 2.0), (145, 2.0), (192, 1.5), (200, 1.5), (256, 1.0))}
