"""ctypes bindings of the TEST-ONLY checkers: oracle/libdrt_oracle.so (CPU restatement) and, where it
was built, oracle/_ref/libdrt_ref.so (the real reference path). Imported by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() only -- never by the product.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(REPO, "daily-ray-trace_amd"))
import pydrt  # noqa: E402  (struct definitions of the boundary only)

MATH_REFERENCE, MATH_DEVICE = 0, 1
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)


class Point(C.Structure):
    _fields_ = [("position", C.c_double * 3), ("normal", C.c_double * 3), ("out", C.c_double * 3),
                ("on_dot", C.c_double), ("trans_wl", C.c_double), ("surface_material", C.c_uint32),
                ("incident_material", C.c_uint32), ("transmit_material", C.c_uint32)]


def make_point(position, normal, out, on_dot, surface_material, incident_material, transmit_material, trans_wl=630.0):
    p = Point()
    p.position = (C.c_double * 3)(*position)
    p.normal = (C.c_double * 3)(*normal)
    p.out = (C.c_double * 3)(*out)
    p.on_dot, p.trans_wl = on_dot, trans_wl
    p.surface_material, p.incident_material, p.transmit_material = surface_material, incident_material, transmit_material
    return p


def _v3(a):
    return (C.c_double * 3)(*[float(x) for x in a])


def _ptr(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(HERE, "libdrt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libdrt_oracle.so is not built: make -C oracle")
        L = C.CDLL(path)
        S, Cm, P = C.POINTER(pydrt.Scene), C.POINTER(pydrt.Camera), C.POINTER(pydrt.Params)
        L.drt_oracle_render_tile.argtypes = [S, Cm, P, f64p, f64p, f64p, i32p, C.POINTER(pydrt.Stats), C.c_int]
        L.drt_oracle_sample_scene.argtypes = [S, Cm, P, C.c_uint32, C.c_uint32, C.c_uint32, f64p, f64p, i32p]
        L.drt_oracle_spectrum_to_xyz.argtypes = [S, f64p, f64p]
        L.drt_oracle_film_to_xyz.argtypes = [S, f64p, C.c_uint64, f64p]
        L.drt_oracle_line_sphere.restype = C.c_double
        L.drt_oracle_line_sphere.argtypes = [f64p, f64p, f64p, C.c_double]
        L.drt_oracle_line_plane.restype = C.c_double
        L.drt_oracle_line_plane.argtypes = [f64p] * 6
        L.drt_oracle_reflect.argtypes = [f64p, f64p, f64p]
        L.drt_oracle_transmit.argtypes = [f64p, f64p, C.c_double, C.c_double, f64p]
        L.drt_oracle_rotation_between.argtypes = [f64p, f64p, f64p]
        L.drt_oracle_rotation_about_axis.argtypes = [f64p, C.c_double, f64p]
        L.drt_oracle_seed_path.argtypes = [C.c_uint64]
        L.drt_oracle_set_rng_state.argtypes = [C.c_uint64]
        L.drt_oracle_get_rng_state.restype = C.c_uint64
        L.drt_oracle_rng.restype = C.c_double
        L.drt_oracle_path_key.restype = C.c_uint64
        L.drt_oracle_path_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.drt_oracle_uniform_sample_sphere.argtypes = [f64p]
        L.drt_oracle_uniform_sample_disc.argtypes = [f64p]
        L.drt_oracle_sincos.argtypes = [C.c_double, f64p, f64p]
        L.drt_oracle_bdsf_func.argtypes = [S, C.c_uint32, C.POINTER(Point), f64p, f64p]
        L.drt_oracle_bdsf.argtypes = [S, C.POINTER(Point), f64p, f64p]
        L.drt_oracle_dir_func.argtypes = [S, C.c_uint32, C.POINTER(Point), f64p, f64p]
        L.drt_oracle_ggx.restype = C.c_double
        L.drt_oracle_ggx.argtypes = [f64p, f64p, C.c_double]
        L.drt_oracle_ggx_att.restype = C.c_double
        L.drt_oracle_ggx_att.argtypes = [f64p, f64p, f64p, C.c_double]
        L.drt_oracle_fs_dielectric_reflectance.argtypes = [f64p, f64p, C.c_double, C.c_uint32, f64p]
        L.drt_oracle_fs_conductor_reflectance.argtypes = [f64p, f64p, f64p, C.c_double, C.c_uint32, f64p]
        L.drt_oracle_value_at_wl.restype = C.c_double
        L.drt_oracle_value_at_wl.argtypes = [S, f64p, C.c_double]
        L.drt_oracle_find_ray_intersection.argtypes = [S, f64p, f64p, C.POINTER(Point)]
        L.drt_oracle_points_mutually_visible.argtypes = [S, f64p, f64p]
        L.drt_oracle_direct_light.argtypes = [S, C.POINTER(Point), f64p]
        _oracle = L
    return _oracle


def set_math_mode(mode):
    oracle_lib().drt_oracle_set_math_mode(mode)


def oracle_render_tile(bundle, params, want_hits=False, num_threads=1, math_mode=None):
    """CPU twin of drt_render_tile. Returns (pixels, avgs, vars, hits|None, Stats)."""
    L = oracle_lib()
    if math_mode is not None:
        set_math_mode(math_mode)
    n = int(params.tile_w) * int(params.tile_h)
    S = bundle.S
    px = np.zeros((n, S + 1))
    av = np.zeros((n, S))
    va = np.zeros((n, S))
    hits = np.full((n * int(params.spp), int(params.max_depth)), -2, dtype=np.int32) if want_hits else None
    st = pydrt.Stats()
    rc = L.drt_oracle_render_tile(C.byref(bundle.scene), C.byref(bundle.camera), C.byref(params), _ptr(px), _ptr(av),
                                  _ptr(va), _ptr(hits, C.c_int32), C.byref(st), num_threads)
    if rc != 0:
        raise RuntimeError("drt_oracle_render_tile failed: %d" % rc)
    return px, av, va, hits, st


def oracle_film_to_xyz(bundle, pixels):
    L = oracle_lib()
    pixels = np.ascontiguousarray(pixels, dtype=np.float64)
    n = pixels.shape[0]
    xyz = np.empty((n, 3))
    L.drt_oracle_film_to_xyz(C.byref(bundle.scene), _ptr(pixels), n, _ptr(xyz))
    return xyz


# ------------------------------------------------------------------------------------------------
class RefSpdInput(C.Structure):
    _fields_ = [("method", C.c_uint32), ("has_scale_factor", C.c_uint32), ("scale_factor", C.c_double),
                ("value", C.c_double * 3), ("csv", C.c_char * 64)]


class RefMaterialInput(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("is_base_material", C.c_uint32), ("is_escape_material", C.c_uint32),
                ("is_black_body", C.c_uint32), ("is_emissive", C.c_uint32), ("shininess", C.c_double),
                ("roughness", C.c_double), ("spd", RefSpdInput * 6), ("num_bdsfs", C.c_uint32),
                ("bdsfs", C.c_int32 * 16), ("dir_func", C.c_int32)]


class RefSurfaceInput(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("material_name", C.c_char * 32), ("type", C.c_uint32), ("pad", C.c_uint32),
                ("position", C.c_double * 3), ("radius", C.c_double), ("normal", C.c_double * 3),
                ("u", C.c_double * 3), ("v", C.c_double * 3)]


# int drt_render_tile(scene, camera, params, pixels, avgs, vars, stats): the boundary's one-shot entry point (include/drt_hip.h)
RENDER_TILE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(pydrt.Scene), C.POINTER(pydrt.Camera), C.POINTER(pydrt.Params), f64p, f64p, f64p,
                             C.POINTER(pydrt.Stats))

SPD_METHOD_NONE, SPD_METHOD_RGB, SPD_METHOD_CSV, SPD_METHOD_BLACKBODY, SPD_METHOD_CONST = range(5)  # src/read_scene.h:15-23

_ref = None


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libdrt_ref.so"))


def ref_lib():
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libdrt_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libdrt_ref.so is not built (needs /root/reference): make -C oracle ref")
        L = C.CDLL(path)
        S, Cm, P = C.POINTER(pydrt.Scene), C.POINTER(pydrt.Camera), C.POINTER(pydrt.Params)
        L.ref_seed_path.argtypes = [C.c_uint64]
        L.ref_set_rng_state.argtypes = [C.c_uint64]
        L.ref_get_rng_state.restype = C.c_uint64
        L.ref_rng_draws.restype = C.c_uint64
        L.ref_rng.restype = C.c_double
        L.ref_set_grid.argtypes = [C.c_uint32, C.c_double, C.c_double]
        L.ref_set_tables.argtypes = [f64p]
        L.ref_line_sphere.restype = C.c_double
        L.ref_line_sphere.argtypes = [f64p, f64p, f64p, C.c_double]
        L.ref_line_plane.restype = C.c_double
        L.ref_line_plane.argtypes = [f64p] * 6
        L.ref_reflect.argtypes = [f64p, f64p, f64p]
        L.ref_transmit.argtypes = [f64p, f64p, C.c_double, C.c_double, f64p]
        L.ref_rotation_between.argtypes = [f64p, f64p, f64p]
        L.ref_rotation_about_axis.argtypes = [f64p, C.c_double, f64p]
        L.ref_create_plane.argtypes = [f64p] * 6
        L.ref_uniform_sample_sphere.argtypes = [f64p]
        L.ref_uniform_sample_disc.argtypes = [f64p]
        L.ref_rgb_to_spectrum.argtypes = [f64p, f64p]
        L.ref_spectrum_to_xyz.argtypes = [f64p, f64p]
        L.ref_spectrum_to_rgb.argtypes = [f64p, f64p]
        L.ref_blackbody.argtypes = [C.c_double, f64p]
        L.ref_value_at_wl.restype = C.c_double
        L.ref_value_at_wl.argtypes = [f64p, C.c_double]
        L.ref_init_camera.argtypes = [Cm, f64p, f64p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_uint32, C.c_uint32]
        L.ref_ggx.restype = C.c_double
        L.ref_ggx.argtypes = [f64p, f64p, C.c_double]
        L.ref_ggx_att.restype = C.c_double
        L.ref_ggx_att.argtypes = [f64p, f64p, f64p, C.c_double]
        L.ref_fs_dielectric_reflectance.argtypes = [f64p, f64p, C.c_double, f64p]
        L.ref_fs_conductor_reflectance.argtypes = [f64p, f64p, f64p, C.c_double, f64p]
        L.ref_set_scene.argtypes = [S]
        L.ref_bdsf_func.argtypes = [C.c_uint32, C.POINTER(Point), f64p, f64p]
        L.ref_bdsf.argtypes = [C.POINTER(Point), f64p, f64p]
        L.ref_dir_func.argtypes = [C.c_uint32, C.POINTER(Point), f64p, f64p]
        L.ref_find_ray_intersection.argtypes = [f64p, f64p, C.POINTER(Point)]
        L.ref_points_mutually_visible.argtypes = [f64p, f64p]
        L.ref_direct_light.argtypes = [C.POINTER(Point), f64p]
        L.ref_sample_scene.argtypes = [Cm, P, C.c_uint32, C.c_uint32, C.c_uint32, f64p, f64p]
        L.ref_trace_hits.argtypes = [Cm, P, C.c_uint32, C.c_uint32, C.c_uint32, i32p, f64p]
        L.ref_render_tile.argtypes = [Cm, P, f64p, f64p, f64p]
        # the reference's own parser and CSV resampling (harness version 2)
        u32p = C.POINTER(C.c_uint32)
        L.ref_parse_scene.argtypes = [C.c_char_p, C.c_uint32, u32p, u32p]
        L.ref_parsed_camera.argtypes = [f64p]
        L.ref_parsed_material.argtypes = [C.c_uint32, C.POINTER(RefMaterialInput)]
        L.ref_parsed_surface.argtypes = [C.c_uint32, C.POINTER(RefSurfaceInput)]
        L.ref_parse_config.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.ref_sizeof_config.restype = C.c_uint32
        L.ref_csv_to_spectrum.argtypes = [C.c_char_p, f64p]
        L.ref_spectrum_normalise.argtypes = [f64p]
        L.ref_spectral_mul_by_scalar.argtypes = [f64p, C.c_double]
        L.ref_const_spectrum.argtypes = [f64p, C.c_double]
        L.ref_run_binding.argtypes = [Cm, P, RENDER_TILE_FN, f64p, f64p, f64p]
        _ref = L
    return _ref


def ref_render_tile(bundle, params):
    """The reference's own pixel loop over the tile (real reference code). Returns (pixels, avgs, vars)."""
    L = ref_lib()
    L.ref_set_scene(C.byref(bundle.scene))
    n = int(params.tile_w) * int(params.tile_h)
    S = bundle.S
    px = np.zeros((n, S + 1))
    av = np.zeros((n, S))
    va = np.zeros((n, S))
    L.ref_render_tile(C.byref(bundle.camera), C.byref(params), _ptr(px), _ptr(av), _ptr(va))
    return px, av, va


def ref_trace_hits(bundle, params):
    """Hit-index sequences + per-path spectra through the reference's own functions, ordered
    (sample, row, col) like the oracle. Also returns the spectra of sample_scene() for the same
    paths so the caller can assert both are bitwise equal."""
    L = ref_lib()
    L.ref_set_scene(C.byref(bundle.scene))
    S = bundle.S
    n = int(params.tile_w) * int(params.tile_h) * int(params.spp)
    hits = np.full((n, int(params.max_depth)), -2, dtype=np.int32)
    spec_replay = np.zeros((n, S))
    spec_real = np.zeros((n, S))
    filt = C.c_double()
    stride = int(params.row_stride) or 1
    k = 0
    for s in range(int(params.spp)):
        for j in range(int(params.tile_h)):
            for i in range(int(params.tile_w)):
                x, y, smp = int(params.x0) + i, int(params.y0) + j * stride, int(params.first_sample) + s
                L.ref_trace_hits(C.byref(bundle.camera), C.byref(params), x, y, smp,
                                 hits[k].ctypes.data_as(i32p), spec_replay[k].ctypes.data_as(f64p))
                L.ref_sample_scene(C.byref(bundle.camera), C.byref(params), x, y, smp,
                                   spec_real[k].ctypes.data_as(f64p), C.byref(filt))
                k += 1
    return hits, spec_replay, spec_real


def ref_parse_scene(text):
    """The reference's own parse_scene (src/read_scene.c) on a .scn text. Returns (rc, camera[11], materials, surfaces);
    rc != 0 is what parse_error()'s exit() was called with. The reference's arrays hold 16 materials / 16 surfaces
    with no bounds check (src/read_scene.h:85-86), so the caller must not pass larger scenes."""
    L = ref_lib()
    b = text.encode() if isinstance(text, str) else text
    nm, ns = C.c_uint32(), C.c_uint32()
    rc = L.ref_parse_scene(b, len(b), C.byref(nm), C.byref(ns))
    cam = np.zeros(11)
    L.ref_parsed_camera(_ptr(cam))
    mats, surfs = [], []
    if rc == 0:
        for i in range(min(nm.value, 16)):
            m = RefMaterialInput()
            L.ref_parsed_material(i, C.byref(m))
            mats.append(m)
        for i in range(min(ns.value, 16)):
            su = RefSurfaceInput()
            L.ref_parsed_surface(i, C.byref(su))
            surfs.append(su)
    return rc, cam, mats, surfs


def ref_parse_config(text):
    """The reference's parse_config into its own 1136-byte config_arguments; returns (rc, bytes)."""
    L = ref_lib()
    b = text.encode() if isinstance(text, str) else text
    n = L.ref_sizeof_config()
    buf = C.create_string_buffer(n)
    rc = L.ref_parse_config(b, len(b), buf, n)
    return rc, buf.raw
