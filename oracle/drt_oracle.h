/*
 * drt_oracle.h -- CPU restatement of daily-ray-trace's per-pixel render path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing in the product (daily-ray-trace_amd/, include/) may include,
 * link or call this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker / the reported CPU baseline.
 *
 * Pinning: validated against the compiled reference (oracle/_ref, built by oracle/Makefile from
 * the reference's own source files) and against the committed fixtures under tests/golden/
 * that were generated from it (oracle/make_golden.py). See DESIGN.md "Oracle".
 *
 * The functions consume the boundary structs of include/drt_hip.h.
 */
#ifndef DRT_ORACLE_H
#define DRT_ORACLE_H

#include <stdint.h>
#include "../include/drt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * Arithmetic mode.
 *  DRT_ORACLE_MATH_REFERENCE: what gcc/x86-64 makes of the reference source -- every expression
 *      containing PI (an `L` literal, src/types.h:1) is evaluated in x87 long double, and
 *      sin/cos/pow come from libm. This is the mode checked against oracle/_ref (bit-exact).
 *  DRT_ORACLE_MATH_DEVICE: the same expressions in IEEE f64 only, with the path's own
 *      range-reduced sincos (spec below). This is the arithmetic the HIP kernels use, so
 *      hit indices and XYZ can be compared bit-for-bit with the GPU (pow excepted: libm vs ocml).
 */
enum
{
    DRT_ORACLE_MATH_REFERENCE = 0,
    DRT_ORACLE_MATH_DEVICE    = 1
};
void drt_oracle_set_math_mode(int mode);
int  drt_oracle_get_math_mode(void);

/* ---- whole path ---------------------------------------------------------------------------- */

/* CPU twin of drt_render_tile(): same arguments, same accumulate-into semantics
 * (src/daily_ray_trace.c:710-745). hit_indices (optional): [spp*tile_h*tile_w][max_depth] int32,
 * closest-hit surface index per find_ray_intersection call (-1 miss, -2 not reached), ordered
 * (sample, tile row, tile column). num_threads<=1: single thread (the reference's configuration);
 * >1: rows split across that many pthreads (per-path RNG makes the result identical). */
int drt_oracle_render_tile(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                           double *dst_pixels, double *dst_avgs, double *dst_vars,
                           int32_t *hit_indices, drt_stats *stats, int num_threads);

/* One path: sample_scene() (src/daily_ray_trace.c:571-618) with the per-path seed of (x, y, sample).
 * contribution: [S] doubles. hit_seq: [max_depth] or NULL. Returns the number of closest-hit scans. */
int drt_oracle_sample_scene(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                            uint32_t x, uint32_t y, uint32_t sample,
                            double *contribution, double *filter, int32_t *hit_seq);

/* spectrum_to_xyz (src/spectrum.c:49-70). */
void drt_oracle_spectrum_to_xyz(const drt_scene *scene, const double *spd, double xyz[3]);
/* Per-pixel XYZ of sum/filter for a [n_pixels][S+1] film (src/daily_ray_trace.c:15-23 + spectrum_to_xyz). */
void drt_oracle_film_to_xyz(const drt_scene *scene, const double *pixels, uint64_t n_pixels, double *xyz);

/* ---- unit-level entry points (pinned one by one against oracle/_ref) ----------------------- */

double drt_oracle_line_sphere(const double o[3], const double d[3], const double c[3], double r);
double drt_oracle_line_plane(const double o[3], const double d[3], const double p[3], const double n[3],
                             const double u[3], const double v[3]);
void   drt_oracle_reflect(const double v[3], const double n[3], double out[3]);
void   drt_oracle_transmit(const double v[3], const double n[3], double ir, double tr, double out[3]);
void   drt_oracle_rotation_between(const double v[3], const double w[3], double m_cols[9]);
void   drt_oracle_rotation_about_axis(const double axis[3], double angle, double m_cols[9]);

/* RNG of SURVEY 8a-R. */
void   drt_oracle_seed_path(uint64_t key);
void   drt_oracle_set_rng_state(uint64_t state);
uint64_t drt_oracle_get_rng_state(void);
double drt_oracle_rng(void);
uint64_t drt_oracle_path_key(uint64_t seed, uint32_t width, uint32_t height, uint32_t x, uint32_t y, uint32_t sample);
void   drt_oracle_uniform_sample_sphere(double out[3]);
void   drt_oracle_uniform_sample_disc(double out[3]);
/* the path's own sincos (DEVICE mode arithmetic) */
void   drt_oracle_sincos(double t, double *s, double *c);

/* A surface point for the BDSF-level entry points (scene_point, src/daily_ray_trace.h:113-125). */
typedef struct drt_oracle_point
{
    double   position[3];
    double   normal[3];
    double   out[3];
    double   on_dot;
    double   trans_wl;
    uint32_t surface_material;
    uint32_t incident_material;
    uint32_t transmit_material;
} drt_oracle_point;

/* One BDSF by ID. `result` is [S] and is read-modify-write: functions that early-out leave it
 * untouched, exactly like the reference (quirk Q1). */
void drt_oracle_bdsf_func(const drt_scene *scene, uint32_t bdsf_id, const drt_oracle_point *p,
                          const double incoming[3], double *result);
/* bdsf() dispatcher, src/daily_ray_trace.c:215-229. */
void drt_oracle_bdsf(const drt_scene *scene, const drt_oracle_point *p, const double incoming[3], double *reflectance);
/* One direction sampler by ID (uses the oracle RNG state). */
void drt_oracle_dir_func(const drt_scene *scene, uint32_t dirf_id, const drt_oracle_point *p,
                         double dir[3], double *recip_pdf);
double drt_oracle_ggx(const double sn[3], const double mn[3], double r);
double drt_oracle_ggx_att(const double v[3], const double sn[3], const double mn[3], double r);
void   drt_oracle_fs_dielectric_reflectance(const double *ir, const double *tr, double inc_cos, uint32_t n, double *out);
void   drt_oracle_fs_conductor_reflectance(const double *ir, const double *tr, const double *te, double inc_cos, uint32_t n, double *out);
double drt_oracle_value_at_wl(const drt_scene *scene, const double *spd, double wl);

/* find_ray_intersection (src/daily_ray_trace.c:334-403); returns surface index or -1. */
int drt_oracle_find_ray_intersection(const drt_scene *scene, const double o[3], const double d[3], drt_oracle_point *p);
/* points_mutually_visible (src/daily_ray_trace.c:238-270). */
int drt_oracle_points_mutually_visible(const drt_scene *scene, const double p0[3], const double p1[3]);
/* direct_light_contribution (src/daily_ray_trace.c:272-332); contribution: [S]. */
void drt_oracle_direct_light(const drt_scene *scene, const drt_oracle_point *p, double *contribution);

#ifdef __cplusplus
}
#endif
#endif
