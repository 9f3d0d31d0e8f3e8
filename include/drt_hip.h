/*
 * drt_hip.h -- C-ABI of the MI355X render launcher (libdrt_hip.so).
 *
 * This is the drop-in boundary for the per-pixel render loop of daily-ray-trace:
 * the host (plain C, POSIX) keeps loading .scn scenes and spectra CSVs exactly as
 * before, flattens them into the plain structs below, and calls drt_render_tile()
 * (or the drt_create/drt_render/drt_read_film session form) where the reference
 * runs its `for sample / for y / for x` loop.
 *
 * Reference interfaces replaced (paths relative to the reference tree):
 *   - the pixel loop of render_image()           src/daily_ray_trace.c:710-745
 *   - sample_scene()                             src/daily_ray_trace.c:571-618
 *   - cast_ray() and everything below it         src/daily_ray_trace.c:215-479
 *   - the BDSF / direction-sampling plugin tables src/bdsf.h:1-52, src/bdsf_list.h
 *   - rng()/seed_rng()                           src/rng.h:1-2
 *   - spectrum_to_xyz()                          src/spectrum.c:49-70
 *
 * Everything here is plain C: pointers, sizes, doubles. No C++ or torch types.
 * All arithmetic on the path is IEEE f64; integers are u32/u64.
 *
 * The oracle (oracle/drt_oracle.h) consumes the SAME structs, so a parity test
 * hands one scene to both sides.
 */
#ifndef DRT_HIP_H
#define DRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRT_MAX_BDSFS 16 /* object_material.bdsfs[16], src/daily_ray_trace.h:109 */

/* Surface kinds keep the reference's numeric values (src/daily_ray_trace.h:7-14). */
enum
{
    DRT_GEO_NONE   = 0,
    DRT_GEO_POINT  = 1,
    DRT_GEO_SPHERE = 2,
    DRT_GEO_PLANE  = 3
};

/* film_sample_scheme, src/daily_ray_trace.h:20-26 */
enum
{
    DRT_FILM_SAMPLE_CENTER = 1,
    DRT_FILM_SAMPLE_RANDOM = 2
};

/*
 * Material plugin IDs. The names and their ORDER come from bdsf_list.h (the same
 * X-macro file the reference expands into function-pointer tables); here the list
 * expands to integer IDs because function pointers do not cross to the GPU.
 */
#define BDSF(name) DRT_BDSF_##name,
#define DIRF(name)
enum
{
#include "bdsf_list.h"
    DRT_NUM_BDSFS
};
#undef BDSF
#undef DIRF
#define BDSF(name)
#define DIRF(name) DRT_DIRF_##name,
enum
{
#include "bdsf_list.h"
    DRT_NUM_DIRFS
};
#undef BDSF
#undef DIRF

/* One surface: object_geometry (src/daily_ray_trace.h:78-93) without the name.
 * Planes carry the derived normal/u/v of create_plane_from_points (src/geometry.c:203-209). */
typedef struct drt_surface
{
    uint32_t type;     /* DRT_GEO_* */
    uint32_t material; /* index into drt_scene.materials */
    double   position[3];
    double   radius;   /* spheres */
    double   normal[3];
    double   u[3];
    double   v[3];
} drt_surface;

/* One material: object_material (src/daily_ray_trace.h:95-111). SPDs are indices into
 * drt_scene.spds; -1 means "not given" (a NULL spectrum in the reference) and reads as zeros. */
typedef struct drt_material
{
    uint32_t is_black_body;
    uint32_t is_emissive;
    double   shininess;
    double   roughness;
    int32_t  emission_spd;
    int32_t  diffuse_spd;
    int32_t  glossy_spd;
    int32_t  mirror_spd;
    int32_t  refract_spd;
    int32_t  extinct_spd;
    uint32_t num_bdsfs;
    uint32_t bdsfs[DRT_MAX_BDSFS]; /* DRT_BDSF_* */
    uint32_t dir_func;             /* DRT_DIRF_* */
} drt_material;

/* scene_data (src/daily_ray_trace.h:146-156) plus the spectral tables the path reads. */
typedef struct drt_scene
{
    uint32_t           num_surfaces;
    const drt_surface *surfaces;
    uint32_t           num_materials;
    const drt_material *materials;
    uint32_t           base_material;   /* scene_data.base_material   */
    uint32_t           escape_material; /* scene_data.escape_material */

    uint32_t      num_spds;
    uint32_t      num_wavelengths; /* number_of_spectrum_samples, src/spectrum.h:3 */
    const double *spds;            /* [num_spds][num_wavelengths] */
    double        min_wavelength;  /* smallest_wavelength (nm)     */
    double        wavelength_interval;
    /* colour-matching tables (cmfs, src/spectrum.h:17-23) as SPD indices */
    uint32_t cmf_rw, cmf_x, cmf_y, cmf_z;
} drt_scene;

/* camera_data, src/daily_ray_trace.h:158-170 -- same fields, same meaning. */
typedef struct drt_camera
{
    double forward[3];
    double right[3];
    double up[3];
    double aperture_position[3];
    double aperture_radius;
    double focal_depth;
    double focal_length;
    double film_bottom_left[3];
    double pixel_width;
    double pixel_height;
} drt_camera;

enum
{
    DRT_MODE_SPECTRAL = 0, /* film = sum(+filter), mean, variance per wavelength (the reference's output) */
    DRT_MODE_XYZ      = 1  /* XYZ-only film ("a different mode", reported as such): per pixel 8 doubles
                              {X, Y, Z numerators of wavelengths 0..63.., filter sum, X, Y, Z of the tail wavelengths, 0};
                              spectrum_to_xyz's sums are taken per kernel pair and added up, so no spectrum is kept, there
                              is no mean / variance, and 64 instead of 1664 bytes per pixel cross the links. In this mode
                              the "pixels" buffer of every call below is [tile_h*tile_w][8] and avgs / vars are NULL;
                              drt_read_xyz() gives the same XYZ as the spectral film's to rounding (order of sums). */
};

/*
 * What to render. The tile is the pixel set {(x0+i, y0+j*row_stride) : i<tile_w, j<tile_h} of a
 * width x height image (row_stride>1 gives the row-cyclic multi-GPU partition). Path RNG key
 * (SURVEY 8a-R): seed + ((sample*height + y)*width + x) as u64, xorshift64 seeded through splitmix64.
 */
typedef struct drt_params
{
    uint32_t width, height;
    uint32_t x0, y0, tile_w, tile_h, row_stride;
    uint32_t spp;          /* num_pixel_samples  */
    uint32_t first_sample; /* index of the first sample (resume support) */
    uint32_t max_depth;    /* max_cast_depth     */
    uint32_t pixel_scheme; /* DRT_FILM_SAMPLE_*  */
    uint64_t seed;
    uint32_t mode;         /* DRT_MODE_* */
    int32_t  device;       /* HIP device ordinal */
    uint32_t batch_spp;    /* samples traced per launch pair. 0 = sized for a job of `spp` samples (about 32 launch pairs, or as many samples
                              as about 16 GB of vertex records hold, whichever is more); DRT_BATCH_RESIDENT = sized for a context kept across
                              many frames (up to 256 M paths per launch, at least 16 samples per pixel, as memory allows) */
    uint32_t flags;        /* DRT_FLAG_* */
} drt_params;

#define DRT_BATCH_RESIDENT 0xFFFFFFFFu /* drt_params.batch_spp: let the library size launches for a long-lived context */

enum
{
    DRT_FLAG_RECORD_HITS = 1u, /* keep closest-hit surface indices per path vertex (parity tests) */
    DRT_FLAG_FILM_ZERO   = 2u  /* one-shot forms only: the caller's buffers are zero-filled (as the reference's alloc() leaves
                                  them, src/daily_ray_trace.c:689-691), so they are not uploaded before rendering */
};

typedef struct drt_stats
{
    uint64_t paths;
    uint64_t closest_hit_scans; /* find_ray_intersection calls (V_int)       */
    uint64_t shaded_vertices;   /* direct_light_contribution calls (V_shade) */
    uint64_t shadow_scans;      /* points_mutually_visible calls             */
    uint64_t rng_draws;
    double   trace_ms;          /* path-geometry kernel, HIP-event time      */
    double   shade_ms;          /* spectral shade + film kernel              */
    double   total_ms;
    /* the pool of vertex records (four vertices per block): its size, the most a launch has used, and how many launches had to be
     * rendered again in worst-case-sized pieces because the pool ran out (0 unless the tile's paths grew after the context measured them) */
    uint64_t record_pool_blocks;
    uint64_t record_pool_peak;
    uint32_t record_block_bytes;
    uint32_t redone_launches;
    /* the reference's per-pass report (min / max / avg time of one sample pass over the image, src/daily_ray_trace.c:746-756): a
     * kernel pair renders several samples of every pixel it covers, so a pair's HIP-event time is scaled to one sample of the
     * whole tile -- ms x tile pixels / (pixels x samples of the pair) -- and min / max / avg run over the pairs */
    uint32_t launches;   /* kernel pairs timed */
    uint32_t path_flags; /* which kernels this context runs: DRT_PATH_* (a group: the devices' flags or'ed) */
    double   min_sample_ms, max_sample_ms, avg_sample_ms;
} drt_stats;

enum
{
    DRT_PATH_BVH        = 1u, /* the scene is behind the bounding-volume hierarchy: drt_primary_kernel + drt_bounce_kernel trace it */
    DRT_PATH_TRACE_TAIL = 2u  /* the trace kernel carries every path's tail wavelengths (all-plastic scenes); the shade kernel's tail pass only updates the film */
};

typedef struct drt_context drt_context;

/* Last error text of the calling thread ("" when none). */
const char *drt_last_error(void);
/* Number of HIP devices visible; negative on error. */
int drt_device_count(void);

/* Session form. drt_create copies the scene to the device (SoA) and allocates the film
 * (zero-filled, like the reference's VirtualAlloc'ed accumulators, src/daily_ray_trace.c:689-691). */
drt_context *drt_create(const drt_scene *scene, const drt_camera *camera, const drt_params *params);
void         drt_destroy(drt_context *ctx);
/* Use caller-owned DEVICE buffers for the film instead of the library's own
 * ([tile_h*tile_w][S+1], [..][S], [..][S] doubles). The caller zero-fills them. */
int drt_bind_film(drt_context *ctx, void *d_pixels, void *d_avgs, void *d_vars);
/* Launch on this hipStream_t (NULL = the context's own stream). */
int drt_set_stream(drt_context *ctx, void *hip_stream);
/* Enqueue samples [first_sample, first_sample+num_samples) for every tile pixel. Asynchronous. */
int drt_render(drt_context *ctx, uint32_t first_sample, uint32_t num_samples);
int drt_synchronize(drt_context *ctx);
/* Zero the film and the statistics (for repeated timed runs). */
int drt_reset_film(drt_context *ctx);
/* Device pointers of the film buffers (for collectives on them). */
int drt_film_device_ptrs(drt_context *ctx, void **d_pixels, void **d_avgs, void **d_vars);
/* Copy the film to host buffers (any may be NULL). Synchronises. */
int drt_read_film(drt_context *ctx, double *pixels, double *avgs, double *vars);
/* Replace the film with host data (any pointer may be NULL to leave that buffer as is): resuming from a checkpoint. */
int drt_write_film(drt_context *ctx, const double *pixels, const double *avgs, const double *vars);
/* Per-pixel XYZ of sum/filter (spectrum_to_xyz on the device), [tile_h*tile_w][3]. Synchronises. */
int drt_read_xyz(drt_context *ctx, double *xyz);
/* One film buffer as the pixel bytes of the reference's .bmp outputs, [tile_h*tile_w][4] = B, G, R, 255, tile row 0 first (the
 * order a bottom-up BMP stores them): which = 0 sum / filter, 1 running mean, 2 variance / its largest sample. Replaces
 * spd_file_to_rgb_f64_pixels (src/daily_ray_trace.c:1-28) + spectrum_to_rgb_f64 (src/spectrum.c:72-82) + rgb_f64_to_rgb_u8
 * (src/win32_platform.c:136-147) with one kernel over the resident film. DRT_MODE_SPECTRAL only. Synchronises. */
int drt_read_bgra(drt_context *ctx, int which, uint8_t *bgra);
/* Closest-hit surface indices of the LAST rendered sample batch: [n][max_depth] int32 per path
 * (-1 miss, -2 vertex not reached); needs DRT_FLAG_RECORD_HITS. Paths are ordered
 * (sample - first_sample_of_last_call, tile row, tile column). */
int drt_read_hit_indices(drt_context *ctx, int32_t *dst, uint64_t capacity_paths);
int drt_get_stats(drt_context *ctx, drt_stats *out);
/* Samples per pixel one trace+shade kernel pair processes (the launch granularity); 0 on error. */
uint32_t drt_batch_spp(drt_context *ctx);

/*
 * One-shot form matching the reference's loop: host buffers, caller-owned,
 * accumulated INTO (so the caller zero-fills them, as alloc() does in the reference).
 * Returns 0 on success, negative on failure (see drt_last_error()).
 */
int drt_render_tile(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                    double *dst_pixels, double *dst_avgs, double *dst_vars, drt_stats *stats);

/*
 * Several GPUs from ONE host thread (SURVEY 8b "Threading": the launcher may drive 1..8 devices, one stream each).
 * The tile's rows are dealt cyclically over the devices -- device k of n owns tile rows k, k+n, ... for all samples,
 * the same partition the multi-process callers use -- each device has its own context and stream, all of them render
 * concurrently, and the film comes back into the caller's buffers in image order (one strided copy per device and
 * buffer). `devices` lists HIP device ordinals (a device may appear more than once: that many contexts share it);
 * devices == NULL means 0..n_devices-1, n_devices == 0 means every visible device; drt_params.device is ignored.
 * Results are bit-identical to a single context's, whatever the device list.
 */
typedef struct drt_group drt_group;
drt_group *drt_group_create(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                            const int32_t *devices, uint32_t n_devices);
void     drt_group_destroy(drt_group *g);
uint32_t drt_group_size(drt_group *g);
/* Enqueue samples [first_sample, first_sample+num_samples) on every device. Asynchronous. */
int drt_group_render(drt_group *g, uint32_t first_sample, uint32_t num_samples);
int drt_group_synchronize(drt_group *g);
/* Whole-tile film buffers ([tile_h*tile_w][S+1], [..][S], [..][S]) <-> the devices' row sets. NULL skips a buffer. */
int drt_group_read_film(drt_group *g, double *pixels, double *avgs, double *vars);
int drt_group_write_film(drt_group *g, const double *pixels, const double *avgs, const double *vars);
int drt_group_read_bgra(drt_group *g, int which, uint8_t *bgra); /* drt_read_bgra over the whole image */
/* Counters summed over the devices; trace_ms / shade_ms / total_ms are the slowest device's. */
int drt_group_get_stats(drt_group *g, drt_stats *out);
/* One-shot form of drt_render_tile over a device list. */
int drt_render_tile_multi(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                          const int32_t *devices, uint32_t n_devices,
                          double *dst_pixels, double *dst_avgs, double *dst_vars, drt_stats *stats);

/* Shape of the bounding-volume hierarchy drt_create() builds for scenes too large for LDS: node count, surfaces in
 * leaves, levels, and the traversal stack's capacity in entries (one per level at most; drt_create() refuses a deeper tree).
 * Host only: runs without a GPU. */
int drt_bvh_stats(const drt_scene *scene, uint32_t *nodes, uint32_t *leaf_surfaces, uint32_t *depth, uint32_t *stack_entries);

/* Arithmetic self-test kernels: evaluate op over n inputs on the device so tests can check
 * that f64 sqrt / divide / the path's sincos are bit-identical to the host. op: 0 sqrt(a),
 * 1 a/b, 2 sincos(a) -> out[2*i], out[2*i+1], 3 pow(a,b), 4 rng stream from key a (as u64 bits). */
int drt_selftest_arith(int device, int op, const double *a, const double *b, double *out, uint64_t n);

/* Device-function self-test: runs ONE of the path's device functions -- the very __device__ function the trace / shade
 * kernels call -- over n records (`in_stride` doubles in, `out_stride` doubles out per record), so that the edge cases of
 * the reference's functions (tangent / parallel / on-boundary rays, antiparallel rotation, disc centre, total internal
 * reflection) meet the HIP code directly and not only when a random scene happens to produce them. func:
 *   0 line_sphere_intersection  src/geometry.c:123-146   in o[3] d[3] c[3] r                 out t
 *   1 line_plane_intersection   src/geometry.c:157-182   in o[3] d[3] p[3] n[3] u[3] v[3]    out t
 *   2 vec3_reflect              src/geometry.c:85-90     in v[3] n[3]                        out r[3]
 *   3 vec3_transmit             src/geometry.c:92-106    in v[3] n[3] ir tr                  out t[3] (NaN on total internal reflection)
 *   4 find_rotation_between_vectors src/geometry.c:263-295  in v[3] w[3]                     out m[9], columns
 *   5 uniform_sample_sphere     src/rng.c:14-23          in rng state (u64 bits)             out p[3], state after (u64 bits)
 *   6 uniform_sample_disc       src/rng.c:25-51          in rng state (u64 bits)             out p[3], state after (u64 bits)
 *   7 ggx                       src/bdsf.c:3-20          in sn[3] mn[3] roughness            out D
 *   8 ggx_att                   src/bdsf.c:22-42         in v[3] sn[3] mn[3] roughness       out D * G1
 *   9 fs_dielectric_reflectance src/bdsf.c:44-67         in ir tr cos (one wavelength)       out R
 *  10 fs_conductor_reflectance  src/bdsf.c:78-101        in ir tr te cos (one wavelength)    out R
 *  11 seed_rng + rng            src/rng.c:1-12 (8a-R)    in path key (u64 bits)              out state (u64 bits), first rng()
 *  12 the hierarchy's f32 box test (prunes the scan of src/daily_ray_trace.c:340-364; no reference counterpart)
 *                                                          in o[3] d[3] lo[3] hi[3]            out lower bound of the entry distance, < 0: rejected */
int drt_selftest_unit(int device, int func, const double *in, uint32_t in_stride, double *out, uint32_t out_stride, uint64_t n);

#ifdef __cplusplus
}
#endif
#endif /* DRT_HIP_H */
