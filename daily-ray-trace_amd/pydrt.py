"""ctypes bindings of the product: libdrt_hip.so (HIP launcher, include/drt_hip.h) and
libdrt_host.so (POSIX C host: .scn / CSV / camera, host/drt_host.h).

Used by tests/, bench.py and __graft_entry__.py. Python is plumbing only: every compute call
goes through the C-ABI. Nothing here touches oracle/; there is no CPU fallback -- if the HIP
library is missing, `hip_lib()` raises.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)

DRT_MAX_BDSFS = 16
GEO_POINT, GEO_SPHERE, GEO_PLANE = 1, 2, 3
FILM_SAMPLE_CENTER, FILM_SAMPLE_RANDOM = 1, 2
MODE_SPECTRAL, MODE_XYZ = 0, 1
FLAG_RECORD_HITS = 1
BATCH_RESIDENT = 0xFFFFFFFF  # make_params(batch_spp=...): launches sized for a context kept across frames
FLAG_FILM_ZERO = 2
PATH_BVH, PATH_TRACE_TAIL = 1, 2  # Stats.path_flags

# names and order of include/bdsf_list.h
BDSF_NAMES = ["bp_diffuse_bdsf", "bp_glossy_bdsf", "mirror_bdsf", "fs_conductor_bdsf",
              "fs_dielectric_reflectance_bdsf", "fs_dielectric_transmittance_bdsf", "ct_conductor_bdsf"]
DIRF_NAMES = ["cos_weighted_sample_hemisphere", "uniform_sample_hemisphere", "sample_specular_direction",
              "sample_transmit_direction", "sample_reflect_or_transmit_direction", "sample_ct_direction"]
BDSF = {n: i for i, n in enumerate(BDSF_NAMES)}
DIRF = {n: i for i, n in enumerate(DIRF_NAMES)}

f64x3 = C.c_double * 3


class Surface(C.Structure):
    _fields_ = [("type", C.c_uint32), ("material", C.c_uint32), ("position", f64x3), ("radius", C.c_double),
                ("normal", f64x3), ("u", f64x3), ("v", f64x3)]


class Material(C.Structure):
    _fields_ = [("is_black_body", C.c_uint32), ("is_emissive", C.c_uint32), ("shininess", C.c_double),
                ("roughness", C.c_double), ("emission_spd", C.c_int32), ("diffuse_spd", C.c_int32),
                ("glossy_spd", C.c_int32), ("mirror_spd", C.c_int32), ("refract_spd", C.c_int32),
                ("extinct_spd", C.c_int32), ("num_bdsfs", C.c_uint32), ("bdsfs", C.c_uint32 * DRT_MAX_BDSFS),
                ("dir_func", C.c_uint32)]


class Scene(C.Structure):
    _fields_ = [("num_surfaces", C.c_uint32), ("surfaces", C.POINTER(Surface)), ("num_materials", C.c_uint32),
                ("materials", C.POINTER(Material)), ("base_material", C.c_uint32), ("escape_material", C.c_uint32),
                ("num_spds", C.c_uint32), ("num_wavelengths", C.c_uint32), ("spds", C.POINTER(C.c_double)),
                ("min_wavelength", C.c_double), ("wavelength_interval", C.c_double), ("cmf_rw", C.c_uint32),
                ("cmf_x", C.c_uint32), ("cmf_y", C.c_uint32), ("cmf_z", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("forward", f64x3), ("right", f64x3), ("up", f64x3), ("aperture_position", f64x3),
                ("aperture_radius", C.c_double), ("focal_depth", C.c_double), ("focal_length", C.c_double),
                ("film_bottom_left", f64x3), ("pixel_width", C.c_double), ("pixel_height", C.c_double)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("x0", C.c_uint32), ("y0", C.c_uint32),
                ("tile_w", C.c_uint32), ("tile_h", C.c_uint32), ("row_stride", C.c_uint32), ("spp", C.c_uint32),
                ("first_sample", C.c_uint32), ("max_depth", C.c_uint32), ("pixel_scheme", C.c_uint32),
                ("seed", C.c_uint64), ("mode", C.c_uint32), ("device", C.c_int32), ("batch_spp", C.c_uint32),
                ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("closest_hit_scans", C.c_uint64), ("shaded_vertices", C.c_uint64),
                ("shadow_scans", C.c_uint64), ("rng_draws", C.c_uint64), ("trace_ms", C.c_double),
                ("shade_ms", C.c_double), ("total_ms", C.c_double), ("record_pool_blocks", C.c_uint64),
                ("record_pool_peak", C.c_uint64), ("record_block_bytes", C.c_uint32), ("redone_launches", C.c_uint32),
                ("launches", C.c_uint32), ("path_flags", C.c_uint32), ("min_sample_ms", C.c_double), ("max_sample_ms", C.c_double),
                ("avg_sample_ms", C.c_double)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


def make_params(width, height, spp, max_depth, seed=1, x0=0, y0=0, tile_w=None, tile_h=None, row_stride=1,
                first_sample=0, pixel_scheme=FILM_SAMPLE_RANDOM, mode=MODE_SPECTRAL, device=0, batch_spp=0, flags=0):
    p = Params()
    p.width, p.height = width, height
    p.x0, p.y0 = x0, y0
    p.tile_w = width if tile_w is None else tile_w
    p.tile_h = height if tile_h is None else tile_h
    p.row_stride = row_stride
    p.spp, p.first_sample, p.max_depth, p.pixel_scheme = spp, first_sample, max_depth, pixel_scheme
    p.seed, p.mode, p.device, p.batch_spp, p.flags = seed, mode, device, batch_spp, flags
    return p


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype)) if a is not None else None


# ------------------------------------------------------------------------------------------------
# libdrt_host.so

_host = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(HERE, "libdrt_host.so")
        if not os.path.exists(path):
            raise RuntimeError("libdrt_host.so is not built: run __graft_entry__.build() / make -C daily-ray-trace_amd host")
        L = C.CDLL(path)
        L.drt_host_load_scene.restype = C.c_void_p
        L.drt_host_load_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double,
                                          C.c_double, C.c_double]
        L.drt_host_load_scene_text.restype = C.c_void_p
        L.drt_host_load_scene_text.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                               C.c_double, C.c_double, C.c_double]
        L.drt_host_free_scene.argtypes = [C.c_void_p]
        L.drt_host_scene_data.restype = C.POINTER(Scene)
        L.drt_host_scene_data.argtypes = [C.c_void_p]
        L.drt_host_camera_data.restype = C.POINTER(Camera)
        L.drt_host_camera_data.argtypes = [C.c_void_p]
        L.drt_host_material_name.restype = C.c_char_p
        L.drt_host_material_name.argtypes = [C.c_void_p, C.c_uint32]
        L.drt_host_surface_name.restype = C.c_char_p
        L.drt_host_surface_name.argtypes = [C.c_void_p, C.c_uint32]
        L.drt_host_last_error.restype = C.c_char_p
        L.drt_host_csv_to_spectrum.restype = C.c_uint32
        L.drt_host_csv_to_spectrum.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_uint32, C.POINTER(C.c_double)]
        L.drt_host_rgb_to_spectrum.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.drt_host_blackbody_spectrum.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_double, C.POINTER(C.c_double)]
        L.drt_host_init_camera.argtypes = [C.POINTER(Camera), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                           C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32]
        _host = L
    return _host


class SceneBundle:
    """A scene + camera pair ready for the C-ABI. Keeps every backing array alive."""

    def __init__(self, scene, camera, keep=None, handle=None):
        self.scene = scene
        self.camera = camera
        self._keep = keep
        self._handle = handle

    @property
    def S(self):
        return int(self.scene.num_wavelengths)

    def spds(self):
        n = int(self.scene.num_spds) * self.S
        return np.ctypeslib.as_array(self.scene.spds, shape=(n,)).reshape(int(self.scene.num_spds), self.S).copy()

    def material_names(self):
        if self._handle is None:
            return [""] * int(self.scene.num_materials)
        return [host_lib().drt_host_material_name(self._handle, i).decode() for i in range(int(self.scene.num_materials))]

    def surface_names(self):
        if self._handle is None:
            return [""] * int(self.scene.num_surfaces)
        return [host_lib().drt_host_surface_name(self._handle, i).decode() for i in range(int(self.scene.num_surfaces))]

    def __del__(self):
        try:
            if self._handle is not None and _host is not None:
                _host.drt_host_free_scene(self._handle)
                self._handle = None
        except Exception:
            pass


def load_scene(scene_path, width, height, spectra_dir=None, min_wl=380.0, max_wl=720.0, wl_interval=5.0):
    """parse .scn + build scene/camera through the C host (host/drt_scene.c)."""
    L = host_lib()
    spectra_dir = spectra_dir or os.path.join(REPO, "spectra")
    h = L.drt_host_load_scene(scene_path.encode(), spectra_dir.encode(), None, width, height, min_wl, max_wl, wl_interval)
    if not h:
        raise RuntimeError("drt_host_load_scene: " + L.drt_host_last_error().decode())
    return SceneBundle(L.drt_host_scene_data(h).contents, L.drt_host_camera_data(h).contents, handle=h)


def load_scene_text(text, width, height, spectra_dir=None, min_wl=380.0, max_wl=720.0, wl_interval=5.0):
    L = host_lib()
    spectra_dir = spectra_dir or os.path.join(REPO, "spectra")
    b = text.encode()
    h = L.drt_host_load_scene_text(b, len(b), spectra_dir.encode(), None, width, height, min_wl, max_wl, wl_interval)
    if not h:
        raise RuntimeError("drt_host_load_scene_text: " + L.drt_host_last_error().decode())
    return SceneBundle(L.drt_host_scene_data(h).contents, L.drt_host_camera_data(h).contents, handle=h)


def init_camera(position, target, roll, fov, fdepth, flength, aperture, width, height):
    cam = Camera()
    pos = (C.c_double * 3)(*position)
    tgt = (C.c_double * 3)(*target)
    host_lib().drt_host_init_camera(C.byref(cam), pos, tgt, roll, fov, fdepth, flength, aperture, width, height)
    return cam


def build_scene(surfaces, materials, spds, base_material, escape_material, camera, min_wl=380.0, wl_interval=5.0,
                cmf=(0, 1, 2, 3)):
    """Assemble a Scene from Python data (synthetic scenes).

    surfaces: list of dicts {type, material, position, radius?, normal?, u?, v?}
    materials: list of dicts with Material field names (bdsfs as list of ids)
    spds: float64 array [n_spd][S]
    """
    spds = np.ascontiguousarray(spds, dtype=np.float64)
    sa = (Surface * max(1, len(surfaces)))()
    for i, s in enumerate(surfaces):
        sa[i].type = s["type"]
        sa[i].material = s["material"]
        sa[i].position = f64x3(*s["position"])
        sa[i].radius = float(s.get("radius", 0.0))
        for k in ("normal", "u", "v"):
            if k in s:
                setattr(sa[i], k, f64x3(*s[k]))
    ma = (Material * max(1, len(materials)))()
    for i, m in enumerate(materials):
        for k in ("emission_spd", "diffuse_spd", "glossy_spd", "mirror_spd", "refract_spd", "extinct_spd"):
            setattr(ma[i], k, int(m.get(k, -1)))
        ma[i].is_black_body = int(m.get("is_black_body", 0))
        ma[i].is_emissive = int(m.get("is_emissive", 0))
        ma[i].shininess = float(m.get("shininess", 0.0))
        ma[i].roughness = float(m.get("roughness", 0.0))
        b = m.get("bdsfs", [])
        ma[i].num_bdsfs = len(b)
        for j, v in enumerate(b):
            ma[i].bdsfs[j] = v
        ma[i].dir_func = int(m.get("dir_func", 0))
    sc = Scene()
    sc.num_surfaces = len(surfaces)
    sc.surfaces = C.cast(sa, C.POINTER(Surface))
    sc.num_materials = len(materials)
    sc.materials = C.cast(ma, C.POINTER(Material))
    sc.base_material, sc.escape_material = base_material, escape_material
    sc.num_spds, sc.num_wavelengths = spds.shape
    sc.spds = _ptr(spds, C.c_double)
    sc.min_wavelength, sc.wavelength_interval = min_wl, wl_interval
    sc.cmf_rw, sc.cmf_x, sc.cmf_y, sc.cmf_z = cmf
    return SceneBundle(sc, camera, keep=(sa, ma, spds))


def plane_from_points(o, pu, pv):
    """create_plane_from_points: returns (u, v, n) for a surface dict."""
    o, pu, pv = (np.asarray(a, dtype=np.float64) for a in (o, pu, pv))
    u, v = pu - o, pv - o
    n = np.array([u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]])
    n = n / np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
    return u, v, n


# ------------------------------------------------------------------------------------------------
# libdrt_hip.so

_hip = None


def _share_torch_hip_runtime():
    """One HIP runtime per process. A PyTorch-ROCm wheel carries its own libamdhip64.so (soname libamdhip64.so.7, like
    /opt/rocm's) and loads it by file name; if libdrt_hip.so has pulled in /opt/rocm's copy first, the process ends up with
    two runtimes and the second one finds no GPU ("No HIP GPUs are available"). So when such a wheel is installed, its copy
    is loaded first -- without importing torch -- and libdrt_hip.so binds to it by soname; `import torch` later finds the
    same file already mapped. DRT_NO_TORCH_RUNTIME=1 turns this off (then never import torch after the first render)."""
    if os.environ.get("DRT_NO_TORCH_RUNTIME") or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def hip_lib():
    global _hip
    if _hip is None:
        path = os.environ.get("DRT_HIP_LIB") or os.path.join(HERE, "libdrt_hip.so")  # DRT_HIP_LIB: A/B builds when profiling
        if not os.path.exists(path):
            raise RuntimeError("libdrt_hip.so is not built (no fallback exists): run __graft_entry__.build()")
        _share_torch_hip_runtime()
        L = C.CDLL(path)
        L.drt_last_error.restype = C.c_char_p
        L.drt_device_count.restype = C.c_int
        L.drt_create.restype = C.c_void_p
        L.drt_create.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.POINTER(Params)]
        L.drt_destroy.argtypes = [C.c_void_p]
        L.drt_bind_film.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.drt_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.drt_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.drt_synchronize.argtypes = [C.c_void_p]
        L.drt_reset_film.argtypes = [C.c_void_p]
        L.drt_film_device_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.drt_read_film.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.drt_read_xyz.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.drt_read_bgra.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8)]
        L.drt_group_read_bgra.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint8)]
        f64p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L.drt_group_create.restype = C.c_void_p
        L.drt_group_create.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.POINTER(Params), i32p, C.c_uint32]
        L.drt_group_destroy.argtypes = [C.c_void_p]
        L.drt_group_destroy.restype = None
        L.drt_group_size.argtypes = [C.c_void_p]
        L.drt_group_size.restype = C.c_uint32
        L.drt_group_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.drt_group_synchronize.argtypes = [C.c_void_p]
        L.drt_group_read_film.argtypes = [C.c_void_p, f64p, f64p, f64p]
        L.drt_group_write_film.argtypes = [C.c_void_p, f64p, f64p, f64p]
        L.drt_group_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.drt_render_tile_multi.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.POINTER(Params), i32p, C.c_uint32, f64p, f64p, f64p,
                                            C.POINTER(Stats)]
        L.drt_write_film.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.drt_read_hit_indices.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint64]
        L.drt_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.drt_batch_spp.restype = C.c_uint32
        L.drt_batch_spp.argtypes = [C.c_void_p]
        L.drt_render_tile.argtypes = [C.POINTER(Scene), C.POINTER(Camera), C.POINTER(Params), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(Stats)]
        L.drt_selftest_arith.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.c_uint64]
        L.drt_bvh_stats.argtypes = [C.POINTER(Scene)] + [C.POINTER(C.c_uint32)] * 4
        L.drt_selftest_unit.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_double), C.c_uint32,
                                        C.c_uint64]
        _hip = L
    return _hip


HIP_SYMBOLS = ["drt_last_error", "drt_device_count", "drt_create", "drt_destroy", "drt_bind_film", "drt_set_stream",
               "drt_render", "drt_synchronize", "drt_reset_film", "drt_film_device_ptrs", "drt_read_film", "drt_write_film",
               "drt_read_xyz", "drt_read_bgra", "drt_read_hit_indices", "drt_get_stats", "drt_batch_spp", "drt_render_tile", "drt_selftest_arith",
               "drt_selftest_unit", "drt_bvh_stats",
               "drt_group_create", "drt_group_destroy", "drt_group_size", "drt_group_render", "drt_group_synchronize",
               "drt_group_read_film", "drt_group_write_film", "drt_group_read_bgra", "drt_group_get_stats", "drt_render_tile_multi"]


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, hip_lib().drt_last_error().decode()))


class Renderer:
    """Session form of the C-ABI: scene resident on the device, film accumulated on the device."""

    def __init__(self, bundle, params):
        self.L = hip_lib()
        self.bundle = bundle
        self.params = params
        self.S = bundle.S
        self.n_pixels = int(params.tile_w) * int(params.tile_h)
        self.ctx = self.L.drt_create(C.byref(bundle.scene), C.byref(bundle.camera), C.byref(params))
        if not self.ctx:
            raise RuntimeError("drt_create failed: " + self.L.drt_last_error().decode())

    def close(self):
        if self.ctx:
            self.L.drt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind_film(self, d_pixels, d_avgs, d_vars):
        _check(self.L.drt_bind_film(self.ctx, d_pixels, d_avgs, d_vars), "drt_bind_film")

    def set_stream(self, stream_handle):
        _check(self.L.drt_set_stream(self.ctx, stream_handle), "drt_set_stream")

    def render(self, first_sample=None, num_samples=None):
        fs = int(self.params.first_sample) if first_sample is None else first_sample
        ns = int(self.params.spp) if num_samples is None else num_samples
        _check(self.L.drt_render(self.ctx, fs, ns), "drt_render")

    def synchronize(self):
        _check(self.L.drt_synchronize(self.ctx), "drt_synchronize")

    def reset_film(self):
        _check(self.L.drt_reset_film(self.ctx), "drt_reset_film")

    def film_device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(self.L.drt_film_device_ptrs(self.ctx, C.byref(a), C.byref(b), C.byref(c)), "drt_film_device_ptrs")
        return a.value, b.value, c.value

    def read_xyz_film(self):
        """XYZ film mode: the raw [n][8] accumulators (X, Y, Z, filter sum, tail X, Y, Z, 0)."""
        acc = np.empty((self.n_pixels, 8), dtype=np.float64)
        _check(self.L.drt_read_film(self.ctx, _ptr(acc, C.c_double), None, None), "drt_read_film")
        return acc

    def write_xyz_film(self, acc):
        acc = np.ascontiguousarray(acc, dtype=np.float64)
        _check(self.L.drt_write_film(self.ctx, _ptr(acc, C.c_double), None, None), "drt_write_film")

    def read_film(self):
        px = np.empty((self.n_pixels, self.S + 1), dtype=np.float64)
        av = np.empty((self.n_pixels, self.S), dtype=np.float64)
        va = np.empty((self.n_pixels, self.S), dtype=np.float64)
        _check(self.L.drt_read_film(self.ctx, _ptr(px, C.c_double), _ptr(av, C.c_double), _ptr(va, C.c_double)), "drt_read_film")
        return px, av, va

    def write_film(self, px, av, va):
        px, av, va = (np.ascontiguousarray(a, dtype=np.float64) for a in (px, av, va))
        _check(self.L.drt_write_film(self.ctx, _ptr(px, C.c_double), _ptr(av, C.c_double), _ptr(va, C.c_double)), "drt_write_film")

    def read_xyz(self):
        xyz = np.empty((self.n_pixels, 3), dtype=np.float64)
        _check(self.L.drt_read_xyz(self.ctx, _ptr(xyz, C.c_double)), "drt_read_xyz")
        return xyz

    def read_bgra(self, which=0):
        """BMP pixel bytes [n][4] = B, G, R, 255 of the sum (0), mean (1) or normalised variance (2) film."""
        out = np.empty((self.n_pixels, 4), dtype=np.uint8)
        _check(self.L.drt_read_bgra(self.ctx, int(which), _ptr(out, C.c_uint8)), "drt_read_bgra")
        return out

    def read_hit_indices(self, num_samples):
        n = self.n_pixels * num_samples
        out = np.empty((n, int(self.params.max_depth)), dtype=np.int32)
        _check(self.L.drt_read_hit_indices(self.ctx, _ptr(out, C.c_int32), n), "drt_read_hit_indices")
        return out

    def batch_spp(self):
        return int(self.L.drt_batch_spp(self.ctx))

    def stats(self):
        st = Stats()
        _check(self.L.drt_get_stats(self.ctx, C.byref(st)), "drt_get_stats")
        return st


class Group:
    """drt_group_*: one host thread, several GPUs; the tile's rows dealt cyclically over `devices` (None: all visible)."""

    def __init__(self, bundle, params, devices=None):
        self.L = hip_lib()
        self.bundle, self.params, self.S = bundle, params, bundle.S
        self.n_pixels = int(params.tile_w) * int(params.tile_h)
        if devices is None:
            arr, n = None, 0
        else:
            arr, n = (C.c_int32 * len(devices))(*devices), len(devices)
        self.g = self.L.drt_group_create(C.byref(bundle.scene), C.byref(bundle.camera), C.byref(params), arr, n)
        if not self.g:
            raise RuntimeError("drt_group_create: " + self.L.drt_last_error().decode())

    def size(self):
        return int(self.L.drt_group_size(self.g))

    def render(self, first_sample=None, num_samples=None):
        fs = int(self.params.first_sample) if first_sample is None else first_sample
        ns = int(self.params.spp) if num_samples is None else num_samples
        _check(self.L.drt_group_render(self.g, fs, ns), "drt_group_render")

    def read_film(self):
        px = np.empty((self.n_pixels, self.S + 1)); av = np.empty((self.n_pixels, self.S)); va = np.empty((self.n_pixels, self.S))
        f64p = C.POINTER(C.c_double)
        _check(self.L.drt_group_read_film(self.g, px.ctypes.data_as(f64p), av.ctypes.data_as(f64p), va.ctypes.data_as(f64p)), "drt_group_read_film")
        return px, av, va

    def write_film(self, px, av, va):
        f64p = C.POINTER(C.c_double)
        px, av, va = (np.ascontiguousarray(a, dtype=np.float64) for a in (px, av, va))
        _check(self.L.drt_group_write_film(self.g, px.ctypes.data_as(f64p), av.ctypes.data_as(f64p), va.ctypes.data_as(f64p)), "drt_group_write_film")

    def stats(self):
        st = Stats()
        _check(self.L.drt_group_get_stats(self.g, C.byref(st)), "drt_group_get_stats")
        return st

    def close(self):
        if self.g:
            self.L.drt_group_destroy(self.g)
            self.g = None


def render_tile(bundle, params):
    """One-shot drt_render_tile with host buffers. Returns (pixels, avgs, vars, stats)."""
    L = hip_lib()
    n = int(params.tile_w) * int(params.tile_h)
    S = bundle.S
    px = np.zeros((n, S + 1), dtype=np.float64)
    av = np.zeros((n, S), dtype=np.float64)
    va = np.zeros((n, S), dtype=np.float64)
    st = Stats()
    rc = L.drt_render_tile(C.byref(bundle.scene), C.byref(bundle.camera), C.byref(params), _ptr(px, C.c_double),
                           _ptr(av, C.c_double), _ptr(va, C.c_double), C.byref(st))
    _check(rc, "drt_render_tile")
    return px, av, va, st


def selftest_arith(op, a, b=None, device=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    n = a.size
    b = np.ascontiguousarray(b if b is not None else np.zeros_like(a), dtype=np.float64)
    out = np.empty(2 * n if op == 2 else n, dtype=np.float64)
    _check(hip_lib().drt_selftest_arith(device, op, _ptr(a, C.c_double), _ptr(b, C.c_double), _ptr(out, C.c_double), n),
           "drt_selftest_arith")
    return out


(UNIT_LINE_SPHERE, UNIT_LINE_PLANE, UNIT_REFLECT, UNIT_TRANSMIT, UNIT_ROTATION_BETWEEN, UNIT_SAMPLE_SPHERE, UNIT_SAMPLE_DISC,
 UNIT_GGX, UNIT_GGX_ATT, UNIT_FS_DIELECTRIC, UNIT_FS_CONDUCTOR, UNIT_SEED_AND_DRAW, UNIT_BVH_BOX) = range(13)
_UNIT_OUT = {UNIT_LINE_SPHERE: 1, UNIT_LINE_PLANE: 1, UNIT_REFLECT: 3, UNIT_TRANSMIT: 3, UNIT_ROTATION_BETWEEN: 9,
             UNIT_SAMPLE_SPHERE: 4, UNIT_SAMPLE_DISC: 4, UNIT_GGX: 1, UNIT_GGX_ATT: 1, UNIT_FS_DIELECTRIC: 1, UNIT_FS_CONDUCTOR: 1,
             UNIT_SEED_AND_DRAW: 2, UNIT_BVH_BOX: 1}


def bvh_stats(bundle):
    """(nodes, surfaces in leaves, levels, stack capacity) of the hierarchy the library builds for this scene; no GPU needed."""
    v = [C.c_uint32() for _ in range(4)]
    _check(hip_lib().drt_bvh_stats(C.byref(bundle.scene), *[C.byref(x) for x in v]), "drt_bvh_stats")
    return tuple(x.value for x in v)


def selftest_unit(func, records, device=0):
    """One of the path's device functions over `records` ([n][k] doubles; u64 arguments as their bit patterns). Returns [n][m]."""
    rec = np.ascontiguousarray(records, dtype=np.float64)
    if rec.ndim == 1:
        rec = rec.reshape(-1, 1)
    n, k = rec.shape
    m = _UNIT_OUT[func]
    out = np.zeros((n, m), dtype=np.float64)
    _check(hip_lib().drt_selftest_unit(device, func, _ptr(rec, C.c_double), k, _ptr(out, C.c_double), m, n), "drt_selftest_unit")
    return out


# ------------------------------------------------------------------------------------------------
# Synthetic scenes

def _xorshift64_stream(seed):
    x = seed & 0xFFFFFFFFFFFFFFFF
    while True:
        x ^= (x << 13) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 7
        x ^= (x << 17) & 0xFFFFFFFFFFFFFFFF
        yield (x >> 33) / 2147483647.0


def synthetic_sphere_scene(n_spheres, width, height, seed=0x5EED, spectra_dir=None):
    """The many-sphere scene of BASELINE config 5: n spheres, centres uniform in [-20,20]x[-20,20]x[-40,0], radii
    uniform [0.05,0.35] from xorshift64 (draw order cx,cy,cz,r), materials round-robin over {blue, green, red, white,
    teal plastic, mirror, rough gold}, one 10x10 plane light at y=25 (emission constant 1), vacuum base + escape;
    pinhole camera (0,0,30) -> (0,0,-20), fov 60. Materials/SPDs come from cornell_plane_light.scn via the host loader."""
    base = load_scene(os.path.join(REPO, "scenes", "cornell_plane_light.scn"), width, height, spectra_dir=spectra_dir)
    names = base.material_names()
    mats = []
    for i in range(int(base.scene.num_materials)):
        m = base.scene.materials[i]
        mats.append({k: getattr(m, k) for k in ("is_black_body", "is_emissive", "shininess", "roughness", "emission_spd",
                                                 "diffuse_spd", "glossy_spd", "mirror_spd", "refract_spd", "extinct_spd",
                                                 "dir_func")} | {"bdsfs": [m.bdsfs[j] for j in range(m.num_bdsfs)]})
    cycle = [names.index(n) for n in ("blue_plastic", "green_plastic", "red_plastic", "white_plastic", "teal_plastic",
                                      "mirror", "gold")]
    g = _xorshift64_stream(seed)
    surfaces = []
    for i in range(n_spheres):
        cx = -20.0 + 40.0 * next(g)
        cy = -20.0 + 40.0 * next(g)
        cz = -40.0 + 40.0 * next(g)
        r = 0.05 + 0.30 * next(g)
        surfaces.append({"type": GEO_SPHERE, "material": cycle[i % len(cycle)], "position": (cx, cy, cz), "radius": r})
    u, v, n = plane_from_points((-5.0, 25.0, 5.0), (5.0, 25.0, 5.0), (-5.0, 25.0, -5.0))
    surfaces.append({"type": GEO_PLANE, "material": names.index("light"), "position": (-5.0, 25.0, 5.0), "u": u, "v": v, "normal": n})
    cam = init_camera((0.0, 0.0, 30.0), (0.0, 0.0, -20.0), 0.0, 60.0, 6.0, 0.3, 0.0, width, height)
    return build_scene(surfaces, mats, base.spds(), int(base.scene.base_material), int(base.scene.escape_material), cam)
