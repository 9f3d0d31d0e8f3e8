"""Shared test cases: the scenes/configs every parity test and the golden generator iterate over."""
import os

import numpy as np

import pydrt

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scene_path(name):
    return os.path.join(REPO, "scenes", name)


# name -> (scene file or generator tag, width, height, spp, depth, seed, pixel scheme)
RENDER_CASES = {
    "plane_light_16": ("cornell_plane_light.scn", 16, 16, 4, 8, 1, pydrt.FILM_SAMPLE_RANDOM),
    "plane_light_48": ("cornell_plane_light.scn", 48, 48, 4, 8, 1, pydrt.FILM_SAMPLE_RANDOM),
    "plane_light_d1": ("cornell_plane_light.scn", 32, 24, 3, 1, 7, pydrt.FILM_SAMPLE_RANDOM),
    "plane_light_d2": ("cornell_plane_light.scn", 32, 24, 3, 2, 7, pydrt.FILM_SAMPLE_RANDOM),
    "plane_light_d16": ("cornell_plane_light.scn", 24, 24, 2, 16, 3, pydrt.FILM_SAMPLE_RANDOM),
    "plane_light_center": ("cornell_plane_light.scn", 24, 24, 2, 4, 1, pydrt.FILM_SAMPLE_CENTER),
    "init_cornell": ("init_cornell.scn", 32, 32, 4, 4, 1, pydrt.FILM_SAMPLE_RANDOM),  # config 1 scene (legacy syntax)
    "large_box": ("cornell_large_box.scn", 32, 32, 3, 16, 1, pydrt.FILM_SAMPLE_RANDOM),  # config 3 scene
    "gold_mirror": ("cornell_gold_mirror.scn", 32, 32, 4, 8, 1, pydrt.FILM_SAMPLE_RANDOM),  # config 4 scene
    "downward": ("cornell_downward.scn", 24, 24, 2, 4, 1, pydrt.FILM_SAMPLE_RANDOM),  # legacy, roll 180
    "lights": ("test_lights.scn", 32, 32, 4, 6, 5, pydrt.FILM_SAMPLE_RANDOM),  # plane + sphere + point light
    "lens": ("test_lens.scn", 24, 24, 3, 4, 2, pydrt.FILM_SAMPLE_RANDOM),  # thin-lens camera
    "many_lights": ("test_many_lights.scn", 24, 24, 2, 5, 4, pydrt.FILM_SAMPLE_RANDOM),  # 12 lights: vertex records wider than a 64-word register
    "deep_paths": ("cornell_large_box.scn", 16, 16, 2, 40, 6, pydrt.FILM_SAMPLE_RANDOM),  # more vertices than the shade kernel prefetches
    # other wavelength grids: S = 35 (one partial set), 86 (two sets, no tail pass), 137 (two sets + a 9-wide tail);
    # with the 4 nm grid trans_wl = 630 falls between two samples, so value_at_wl really interpolates
    "grid_10nm": ("cornell_plane_light.scn", 20, 20, 3, 6, 8, pydrt.FILM_SAMPLE_RANDOM, (380.0, 720.0, 10.0)),
    "grid_4nm": ("cornell_plane_light.scn", 20, 20, 3, 6, 8, pydrt.FILM_SAMPLE_RANDOM, (380.0, 720.0, 4.0)),
    "grid_2p5nm": ("cornell_plane_light.scn", 20, 20, 3, 6, 8, pydrt.FILM_SAMPLE_RANDOM, (380.0, 720.0, 2.5)),
    "spheres_1500": ("@spheres:1500", 32, 32, 2, 6, 9, pydrt.FILM_SAMPLE_RANDOM),  # config 5 generator, reduced
    # the two shipped scenes without an enclosing box (legacy syntax): a plastic sphere under a point light, 63 % of the camera rays escape
    "first_scene": ("first_scene.scn", 32, 32, 4, 4, 1, pydrt.FILM_SAMPLE_RANDOM),
}

# example_scene.scn gives its camera no fov / fdepth / flength: init_camera (src/daily_ray_trace.c:49-77) then divides 0 by 0, every
# camera ray is NaN, misses every surface and leaves NaN x 0 in the film -- what the reference's arithmetic does with that input is
# the expected output here too: NaN for NaN, compared with fuzz_scenes.same()
NAN_CASES = {
    "example_scene": ("example_scene.scn", 16, 16, 3, 4, 1, pydrt.FILM_SAMPLE_RANDOM),
}


def load_case(name):
    case = RENDER_CASES[name] if name in RENDER_CASES else NAN_CASES[name]
    scene, w, h, spp, depth, seed, scheme = case[:7]
    grid = case[7] if len(case) > 7 else (380.0, 720.0, 5.0)
    if scene.startswith("@spheres:"):
        bundle = pydrt.synthetic_sphere_scene(int(scene.split(":")[1]), w, h)
    else:
        bundle = pydrt.load_scene(scene_path(scene), w, h, min_wl=grid[0], max_wl=grid[1], wl_interval=grid[2])
    params = pydrt.make_params(w, h, spp=spp, max_depth=depth, seed=seed, pixel_scheme=scheme)
    return bundle, params


def rel_err(a, b):
    """max |a-b| / max|b| -- the scale-relative error used for film buffers."""
    scale = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b))) / (scale if scale > 0 else 1.0)


def xyz_rel_err(a, b, floor=1e-9):
    """per-pixel, per-channel relative error of XYZ (the metric BASELINE.json names), with an absolute floor."""
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))


def _surface(name, kind, body, material):
    return "\nSurface\nname %s\ntype %s\n%s\nmaterial %s\n" % (name, kind, body, material)


def degenerate_scenes():
    """cornell_plane_light.scn with geometry and materials a loader would rather not see, as .scn text by name: what the reference's
    arithmetic makes of them (ties by surface order, NaN from a 0 / 0, a sphere of radius 0 ...) is the expected output."""
    base = open(scene_path("cornell_plane_light.scn")).read()
    odd = ("\nMaterial\nname odd1\ndiffuse rgb 0.5, 0.5, 0.5\nglossy rgb 0.3, 0.3, 0.3\nshininess 2.5\nbdsfs bp_diffuse_bdsf, bp_glossy_bdsf\n"
           "dir_func cos_weighted_sample_hemisphere\n\nMaterial\nname odd2\ndiffuse rgb 0.5, 0.5, 0.5\nglossy rgb 0.3, 0.3, 0.3\nshininess 1000000.0\n"
           "bdsfs bp_diffuse_bdsf, bp_glossy_bdsf\ndir_func uniform_sample_hemisphere\n")
    return {
        "coincident_surfaces": base + _surface("floor2", "plane", "position -3.0, -3.0, -3.0\npointu 3.0, -3.0, -3.0\npointv -3.0, -3.0, 3.0", "red_plastic")
                                    + _surface("gold2", "sphere", "position 2.0, -2.0, -0.5\nradius 0.75", "dielectric"),
        "zero_and_negative_radius": base + _surface("z", "sphere", "position 0.0, 0.0, 0.0\nradius 0.0", "red_plastic")
                                         + _surface("n", "sphere", "position 0.5, -1.0, 0.5\nradius -0.6", "green_plastic"),
        "plane_with_parallel_edges": base + _surface("deg", "plane", "position -1.0, -1.0, 0.0\npointu 0.0, -1.0, 0.0\npointv 1.0, -1.0, 0.0", "red_plastic"),
        "plane_with_a_zero_edge": base + _surface("deg0", "plane", "position -1.0, -1.0, 0.0\npointu -1.0, -1.0, 0.0\npointv 1.0, -1.0, 0.5", "red_plastic"),
        "huge_and_tiny_spheres": base + _surface("far", "sphere", "position 0.0, 0.0, -900000.0\nradius 899997.5", "green_plastic")
                                      + _surface("dust", "sphere", "position 0.2, 0.1, 1.0\nradius 0.000001", "red_plastic"),
        "roughness_0_and_odd_shininess": base.replace("roughness 0.1", "roughness 0.0").replace("shininess 16.0", "shininess 0.0") + odd
                                         + _surface("o1", "sphere", "position 0.0, -2.0, 1.0\nradius 0.5", "odd1") + _surface("o2", "sphere", "position 1.0, -2.3, 1.5\nradius 0.5", "odd2"),
        "camera_inside_the_glass_ball": base.replace("position 0.0, 0.0, 8.0", "position -1.5, -1.8, 2.2"),
        # lights: a point light ON a wall and one inside the glass ball, a sphere light that swallows the gold ball, a plane light of no area
        "awkward_lights": base + "\nMaterial\nname extra_light\nemission constant 0.7\nis_black_body true\n"
                               + _surface("on_wall", "point", "position -3.0, 0.5, 0.5", "extra_light") + _surface("in_glass", "point", "position -1.5, -1.8, 2.0", "extra_light")
                               + _surface("big_bulb", "sphere", "position 2.0, -2.0, -0.5\nradius 0.9", "extra_light")
                               + _surface("no_area", "plane", "position 0.0, 2.5, 0.0\npointu 0.0, 2.5, 0.0\npointv 0.5, 2.5, 0.5", "extra_light"),
        # a thin lens as wide as the room, focused on the back wall
        "enormous_aperture": base.replace("aperture 0.0", "aperture 2.5"),
    }
