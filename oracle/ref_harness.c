/*
 * ref_harness.c -- drives the REAL reference code of the render path, compiled from the
 * reference's own source files where they lie (/root/reference/src). TEST INFRASTRUCTURE ONLY.
 *
 * This file contains no reference code. It is built by oracle/Makefile (target `ref`) into
 * oracle/_ref/libdrt_ref.so, only in a container where /root/reference exists. The Makefile
 * hands it two generated include files (kept in a temp dir, never committed, never shipped):
 *   drt_ref_types.h  = src/daily_ray_trace.h minus the five #include lines that pull in the
 *                      Win32 platform layer, the .scn parser and the render driver
 *                      (<Windows.h>, win32_platform.h/.c, read_scene.c, daily_ray_trace.c);
 *   drt_ref_path.inc = src/daily_ray_trace.c lines 49-77 (init_camera), 213-479 (bdsf ..
 *                      cast_ray) and 545-618 (sample_pixel_point, sample_scene), verbatim;
 *   drt_ref_pixel_body.inc = src/daily_ray_trace.c lines 729-743, verbatim: the body of
 *                      render_image's pixel loop (sample_scene call, accumulation, running
 *                      mean / variance), compiled inside ref_render_tile's loop below;
 *   drt_ref_parser.inc = src/read_scene.c lines 1-796 (tokenizer, parse_scene, parse_config),
 *                      verbatim but for the spelling of `#include "Keywords.h"` (the file is
 *                      keywords.h; Linux is case-sensitive);
 *   drt_ref_csv_body.inc = src/read_scene.c lines 812-840 and 843-end: the body of
 *                      load_csv_file_to_spectrum after its Win32 file read, without the
 *                      unalloc() call (the buffer is this harness's own).
 * Every other reference file used (types.h utils.[ch] spectrum.[ch] geometry.[ch] rng.[ch]
 * bdsf.[ch] bdsf_list.h read_scene.h) is included whole, unmodified, straight from the tree.
 * No stand-in is written for any header, library or function the image lacks: the platform
 * layer, init_scene / init_spd (they need its alloc and file calls) and render_image's set-up
 * are simply not part of this build (their callers are dropped by --gc-sections), and scene
 * data arrives through the boundary structs instead. printf and exit are macro-renamed around
 * the parser (it is chatty, and its parse_error() ends the process: here it ends the parse).
 *
 * rand()/srand() are macro-renamed to the build's per-path xorshift64;
 * glibc's RAND_MAX (2^31-1) is what rng() divides by.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REF_API __attribute__((visibility("default")))

static uint64_t g_ref_rng_state = 1;
static uint64_t g_ref_rng_draws = 0;
static int drt_probe_rand(void)
{
    uint64_t x = g_ref_rng_state;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    g_ref_rng_state = x;
    g_ref_rng_draws += 1;
    return (int)(uint32_t)(x >> 33);
}
static void drt_probe_srand(unsigned seed) { g_ref_rng_state = seed ? seed : 1; }
#define rand drt_probe_rand
#define srand drt_probe_srand

#include "drt_ref_types.h" /* generated: see header comment */
#include "drt_ref_path.inc"

#undef rand
#undef srand

/* ---- the reference's tokenizer + parse_scene + parse_config, and its CSV resampling -------- */
#include <setjmp.h>
static jmp_buf g_parse_jmp;
static int g_parse_active = 0;
static void ref_parse_exit(int code)
{
    if (g_parse_active) longjmp(g_parse_jmp, code ? code : 1);
    abort();
}
static int ref_quiet_printf(const char *fmt, ...) { (void)fmt; return 0; }
#define printf ref_quiet_printf
#define exit ref_parse_exit
#include "drt_ref_parser.inc"
static u32 ref_csv_body(spectrum dst, char *csv_file_buffer, u32 csv_file_size)
{
    (void)csv_file_size;
#include "drt_ref_csv_body.inc"
#undef printf
#undef exit

#include "../include/drt_hip.h"

/* ---- RNG ---------------------------------------------------------------------------------- */
static uint64_t splitmix64(uint64_t k)
{
    uint64_t z = k + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z ? z : 1ull;
}
REF_API void ref_seed_path(uint64_t key) { g_ref_rng_state = splitmix64(key); }
REF_API void ref_set_rng_state(uint64_t s) { g_ref_rng_state = s; }
REF_API uint64_t ref_get_rng_state(void) { return g_ref_rng_state; }
REF_API uint64_t ref_rng_draws(void) { return g_ref_rng_draws; }
REF_API double ref_rng(void) { return rng(); }

/* ---- spectral grid and tables (the reference's globals, filled from the boundary structs) ---- */
static f64 *g_block = NULL;

static void set_grid_rows(uint32_t n, double min_wl, double interval, uint32_t scene_rows)
{
    number_of_spectrum_samples = n;
    smallest_wavelength = min_wl;
    sample_interval = interval;
    largest_wavelength = min_wl + (n - 1) * interval;
    spectrum_size = n * sizeof(f64);
    free(g_block);
    /* one block like init_spd_tables' (src/spectrum.c:10-34): 4 cmfs, 7 rgb tables, then the stack; `scene_rows` spectra
     * of a scene are taken from the bottom of the stack, as init_spd's alloc_spd() calls would */
    uint32_t stack_capacity = 32 + scene_rows; /* src/daily_ray_trace.c:658 says 32; scenes built here may hold more spectra */
    g_block = (f64 *)calloc((size_t)(11 + stack_capacity) * n, sizeof(f64));
    spd_alloc_table.size = (11 + stack_capacity) * spectrum_size;
    spd_alloc_table.base = g_block;
    spd_alloc_table.cmf_base = g_block;
    spd_alloc_table.rgb_base = g_block + 4 * n;
    spd_alloc_table.stack_base = g_block + 11 * n;
    cmfs.rw.samples = g_block + 0 * n;
    cmfs.x.samples = g_block + 1 * n;
    cmfs.y.samples = g_block + 2 * n;
    cmfs.z.samples = g_block + 3 * n;
    rgb_spds.white.samples = g_block + 4 * n;
    rgb_spds.red.samples = g_block + 5 * n;
    rgb_spds.green.samples = g_block + 6 * n;
    rgb_spds.blue.samples = g_block + 7 * n;
    rgb_spds.cyan.samples = g_block + 8 * n;
    rgb_spds.magenta.samples = g_block + 9 * n;
    rgb_spds.yellow.samples = g_block + 10 * n;
    spd_stack.capacity = stack_capacity;
    spd_stack.allocated = scene_rows;
    spd_stack.base = g_block + 11 * n;
    spd_stack.next = spd_stack.base + (size_t)scene_rows * n;
}
REF_API void ref_set_grid(uint32_t n, double min_wl, double interval) { set_grid_rows(n, min_wl, interval, 0); }
/* tables: [11][n] = rw, x, y, z, white, red, green, blue, cyan, magenta, yellow */
REF_API void ref_set_tables(const double *tables) { memcpy(g_block, tables, (size_t)11 * number_of_spectrum_samples * sizeof(f64)); }

/* ---- geometry.c / rng.c / spectrum.c, whole files ------------------------------------------ */
static vec3 v3_in(const double a[3]) { vec3 v; v.x = a[0]; v.y = a[1]; v.z = a[2]; return v; }
static void v3_out(vec3 v, double o[3]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
static void m_out(mat3x3 m, double o[9])
{
    for (int c = 0; c < 3; c += 1) for (int r = 0; r < 3; r += 1) o[3 * c + r] = m.columns[c].xyz[r];
}
REF_API double ref_line_sphere(const double o[3], const double d[3], const double c[3], double r)
{
    return line_sphere_intersection(v3_in(o), v3_in(d), v3_in(c), r);
}
REF_API double ref_line_plane(const double o[3], const double d[3], const double p[3], const double n[3], const double u[3], const double v[3])
{
    return line_plane_intersection(v3_in(o), v3_in(d), v3_in(p), v3_in(n), v3_in(u), v3_in(v));
}
REF_API void ref_reflect(const double v[3], const double n[3], double out[3]) { v3_out(vec3_reflect(v3_in(v), v3_in(n)), out); }
REF_API void ref_transmit(const double v[3], const double n[3], double ir, double tr, double out[3])
{
    v3_out(vec3_transmit(v3_in(v), v3_in(n), ir, tr), out);
}
REF_API void ref_rotation_between(const double v[3], const double w[3], double m[9]) { m_out(find_rotation_between_vectors(v3_in(v), v3_in(w)), m); }
REF_API void ref_rotation_about_axis(const double a[3], double angle, double m[9]) { m_out(rotation_about_axis(v3_in(a), angle), m); }
REF_API void ref_create_plane(const double o[3], const double pu[3], const double pv[3], double u[3], double v[3], double n[3])
{
    vec3 po, u_, v_, n_;
    create_plane_from_points(v3_in(o), v3_in(pu), v3_in(pv), &po, &u_, &v_, &n_);
    v3_out(u_, u); v3_out(v_, v); v3_out(n_, n);
}
REF_API void ref_uniform_sample_sphere(double out[3]) { v3_out(uniform_sample_sphere(), out); }
REF_API void ref_uniform_sample_disc(double out[3]) { v3_out(uniform_sample_disc(), out); }
REF_API void ref_rgb_to_spectrum(const double rgb[3], double *dst)
{
    rgb_f64 c; c.r = rgb[0]; c.g = rgb[1]; c.b = rgb[2];
    spectrum s; s.samples = dst;
    rgb_f64_to_spectrum(c, s);
}
REF_API void ref_spectrum_to_xyz(const double *spd, double xyz[3])
{
    spectrum s; s.samples = (f64 *)spd;
    rgb_f64 r = spectrum_to_xyz(s);
    xyz[0] = r.x; xyz[1] = r.y; xyz[2] = r.z;
}
REF_API void ref_spectrum_to_rgb(const double *spd, double rgb[3])
{
    spectrum s; s.samples = (f64 *)spd;
    rgb_f64 r = spectrum_to_rgb_f64(s);
    rgb[0] = r.r; rgb[1] = r.g; rgb[2] = r.b;
}
REF_API void ref_blackbody(double temperature, double *dst)
{
    spectrum s; s.samples = dst;
    generate_blackbody_spectrum(s, temperature);
}
REF_API double ref_value_at_wl(const double *spd, double wl)
{
    spectrum s; s.samples = (f64 *)spd;
    return value_at_wl(s, wl);
}
REF_API void ref_init_camera(drt_camera *out, const double position[3], const double target[3], double roll, double fov,
                             double fdepth, double flength, double aperture, uint32_t w, uint32_t h)
{
    camera_input_data in;
    memset(&in, 0, sizeof(in));
    in.position = v3_in(position); in.target = v3_in(target);
    in.roll = roll; in.fov = fov; in.fdepth = fdepth; in.flength = flength; in.aperture = aperture;
    in.width_px = w; in.height_px = h;
    camera_data cam;
    memset(&cam, 0, sizeof(cam));
    init_camera(&cam, &in);
    v3_out(cam.forward, out->forward); v3_out(cam.right, out->right); v3_out(cam.up, out->up);
    v3_out(cam.aperture_position, out->aperture_position);
    out->aperture_radius = cam.aperture_radius; out->focal_depth = cam.focal_depth; out->focal_length = cam.focal_length;
    v3_out(cam.film_bottom_left, out->film_bottom_left);
    out->pixel_width = cam.pixel_width; out->pixel_height = cam.pixel_height;
}

/* ---- bdsf.c ---------------------------------------------------------------------------------- */
REF_API double ref_ggx(const double sn[3], const double mn[3], double r) { return ggx(v3_in(sn), v3_in(mn), r); }
REF_API double ref_ggx_att(const double v[3], const double sn[3], const double mn[3], double r) { return ggx_att(v3_in(v), v3_in(sn), v3_in(mn), r); }
REF_API void ref_fs_dielectric_reflectance(const double *ir, const double *tr, double inc_cos, double *out)
{
    spectrum o, i, t; o.samples = out; i.samples = (f64 *)ir; t.samples = (f64 *)tr;
    fs_dielectric_reflectance(o, i, t, inc_cos);
}
REF_API void ref_fs_conductor_reflectance(const double *ir, const double *tr, const double *te, double inc_cos, double *out)
{
    spectrum o, i, t, e; o.samples = out; i.samples = (f64 *)ir; t.samples = (f64 *)tr; e.samples = (f64 *)te;
    fs_conductor_reflectance(o, i, t, e, inc_cos);
}

/* ---- scene in the reference's own structs ---------------------------------------------------- */
static scene_data g_scene;
static f64 *g_scene_spds = NULL; /* the scene's spectra: rows 11.. of the one block, where init_spd would have put them */
static f64 *g_zero = NULL;       /* one all-zero row after them: stands for a spectrum the scene does not give (the boundary's -1) */

static spectrum spd_at(const drt_scene *sc, int32_t idx)
{
    spectrum s;
    s.samples = idx < 0 ? g_zero : g_scene_spds + (size_t)idx * sc->num_wavelengths;
    return s;
}

/* Builds scene_data from the boundary structs; sets the spectral grid and the 11 tables from
 * SPD indices 0..10 of the block (the host loader's layout: rw,x,y,z,white,r,g,b,c,m,y). */
REF_API void ref_set_scene(const drt_scene *sc)
{
    uint32_t S = sc->num_wavelengths;
    set_grid_rows(S, sc->min_wavelength, sc->wavelength_interval, sc->num_spds + 1);
    if (sc->num_spds >= 11) ref_set_tables(sc->spds);
    /* colour-matching tables may sit elsewhere in the block */
    memcpy(cmfs.rw.samples, sc->spds + (size_t)sc->cmf_rw * S, S * sizeof(f64));
    memcpy(cmfs.x.samples, sc->spds + (size_t)sc->cmf_x * S, S * sizeof(f64));
    memcpy(cmfs.y.samples, sc->spds + (size_t)sc->cmf_y * S, S * sizeof(f64));
    memcpy(cmfs.z.samples, sc->spds + (size_t)sc->cmf_z * S, S * sizeof(f64));
    free(g_scene.surfaces); free(g_scene.scene_materials);
    g_scene_spds = spd_stack.base;
    memcpy(g_scene_spds, sc->spds, (size_t)sc->num_spds * S * sizeof(f64));
    g_zero = g_scene_spds + (size_t)sc->num_spds * S; /* zero-filled by calloc */
    memset(&g_scene, 0, sizeof(g_scene));
    g_scene.num_surfaces = sc->num_surfaces;
    g_scene.num_scene_materials = sc->num_materials;
    char *sbuf = (char *)calloc(sc->num_surfaces ? sc->num_surfaces : 1, sizeof(object_geometry) + sizeof(u32));
    g_scene.surfaces = (object_geometry *)sbuf;
    g_scene.surface_material_indices = (u32 *)(sbuf + (size_t)(sc->num_surfaces ? sc->num_surfaces : 1) * sizeof(object_geometry));
    g_scene.scene_materials = (object_material *)calloc(sc->num_materials, sizeof(object_material));
    for (uint32_t i = 0; i < sc->num_materials; i += 1)
    {
        const drt_material *m = &sc->materials[i];
        object_material *d = &g_scene.scene_materials[i];
        d->is_black_body = m->is_black_body;
        d->is_emissive = m->is_emissive;
        d->shininess = m->shininess;
        d->roughness = m->roughness;
        d->emission_spd = spd_at(sc, m->emission_spd);
        d->diffuse_spd = spd_at(sc, m->diffuse_spd);
        d->glossy_spd = spd_at(sc, m->glossy_spd);
        d->mirror_spd = spd_at(sc, m->mirror_spd);
        d->refract_spd = spd_at(sc, m->refract_spd);
        d->extinct_spd = spd_at(sc, m->extinct_spd);
        d->num_bdsfs = m->num_bdsfs;
        for (uint32_t j = 0; j < m->num_bdsfs; j += 1) d->bdsfs[j] = bdsf_list[m->bdsfs[j]];
        d->sample_direction = dir_func_list[m->dir_func];
    }
    g_scene.base_material = &g_scene.scene_materials[sc->base_material];
    g_scene.escape_material = &g_scene.scene_materials[sc->escape_material];
    for (uint32_t i = 0; i < sc->num_surfaces; i += 1)
    {
        const drt_surface *s = &sc->surfaces[i];
        object_geometry *d = &g_scene.surfaces[i];
        d->type = (geometry_type)s->type;
        d->position = v3_in(s->position);
        if (s->type == GEO_TYPE_SPHERE) d->radius = s->radius;
        else if (s->type == GEO_TYPE_PLANE)
        {
            d->normal = v3_in(s->normal);
            d->u = v3_in(s->u);
            d->v = v3_in(s->v);
        }
        g_scene.surface_material_indices[i] = s->material;
    }
}

static camera_data cam_in(const drt_camera *c)
{
    camera_data cam;
    memset(&cam, 0, sizeof(cam));
    cam.forward = v3_in(c->forward); cam.right = v3_in(c->right); cam.up = v3_in(c->up);
    cam.aperture_position = v3_in(c->aperture_position);
    cam.aperture_radius = c->aperture_radius; cam.focal_depth = c->focal_depth; cam.focal_length = c->focal_length;
    cam.film_bottom_left = v3_in(c->film_bottom_left);
    cam.pixel_width = c->pixel_width; cam.pixel_height = c->pixel_height;
    return cam;
}

/* The same plain point record as the oracle's drt_oracle_point. */
typedef struct
{
    double position[3], normal[3], out[3];
    double on_dot, trans_wl;
    uint32_t surface_material, incident_material, transmit_material;
} ref_point;

static scene_point point_in(const ref_point *a)
{
    scene_point p;
    memset(&p, 0, sizeof(p));
    p.position = v3_in(a->position); p.normal = v3_in(a->normal); p.out = v3_in(a->out);
    p.on_dot = a->on_dot; p.trans_wl = a->trans_wl;
    p.surface_material = &g_scene.scene_materials[a->surface_material];
    p.incident_material = &g_scene.scene_materials[a->incident_material];
    p.transmit_material = &g_scene.scene_materials[a->transmit_material];
    return p;
}
static void point_out(const scene_point *p, ref_point *a)
{
    v3_out(p->position, a->position); v3_out(p->normal, a->normal); v3_out(p->out, a->out);
    a->on_dot = p->on_dot; a->trans_wl = p->trans_wl;
    a->surface_material = (uint32_t)(p->surface_material - g_scene.scene_materials);
    a->incident_material = p->incident_material ? (uint32_t)(p->incident_material - g_scene.scene_materials) : 0;
    a->transmit_material = p->transmit_material ? (uint32_t)(p->transmit_material - g_scene.scene_materials) : 0;
}

REF_API void ref_bdsf_func(uint32_t id, const ref_point *ap, const double incoming[3], double *result)
{
    scene_point p = point_in(ap);
    spectrum r; r.samples = result;
    bdsf_list[id](r, &p, v3_in(incoming));
}
REF_API void ref_bdsf(const ref_point *ap, const double incoming[3], double *reflectance)
{
    scene_point p = point_in(ap);
    spectrum r; r.samples = reflectance;
    bdsf(r, &p, v3_in(incoming));
}
REF_API void ref_dir_func(uint32_t id, const ref_point *ap, double dir[3], double *recip_pdf)
{
    scene_point p = point_in(ap);
    vec3 v; v.x = v.y = v.z = 0.0;
    f64 pdf = 0.0;
    dir_func_list[id](&v, &pdf, &p);
    v3_out(v, dir);
    *recip_pdf = pdf;
}
REF_API int ref_find_ray_intersection(const double o[3], const double d[3], ref_point *ap)
{
    scene_point p;
    memset(&p, 0, sizeof(p));
    find_ray_intersection(&p, &g_scene, v3_in(o), v3_in(d));
    if (ap) point_out(&p, ap);
    if (p.surface_material == g_scene.escape_material && !p.surface) return -1;
    return (int)(p.surface - g_scene.surfaces);
}
REF_API int ref_points_mutually_visible(const double p0[3], const double p1[3]) { return (int)points_mutually_visible(v3_in(p0), v3_in(p1), &g_scene); }
REF_API void ref_direct_light(const ref_point *ap, double *contribution)
{
    scene_point p = point_in(ap);
    spectrum c; c.samples = contribution;
    direct_light_contribution(c, &p, &g_scene);
}

/* One path through the reference's own sample_scene(). */
REF_API void ref_sample_scene(const drt_camera *c, const drt_params *p, uint32_t x, uint32_t y, uint32_t sample,
                              double *contribution, double *filter)
{
    camera_data cam = cam_in(c);
    ref_seed_path(p->seed + (((uint64_t)sample * p->height + y) * (uint64_t)p->width + x));
    spectrum s; s.samples = contribution;
    sample_scene(s, filter, x, y, &g_scene, &cam, p->max_depth, (film_sample_scheme)p->pixel_scheme);
}

/*
 * Closest-hit index sequence of one path. cast_ray() does not report which surfaces it hit, so
 * this replays the same seed through a loop of the reference's OWN functions in cast_ray's order
 * (find_ray_intersection / direct_light_contribution / sample_direction / bdsf) and logs
 * intersection.surface. The caller checks that the spectrum it produces is bitwise the one
 * sample_scene() gives, which ties the logged sequence to the real cast_ray.
 */
REF_API int ref_trace_hits(const drt_camera *c, const drt_params *p, uint32_t x, uint32_t y, uint32_t sample,
                           int32_t *hit_seq, double *contribution)
{
    camera_data cam = cam_in(c);
    ref_seed_path(p->seed + (((uint64_t)sample * p->height + y) * (uint64_t)p->width + x));
    uint32_t S = number_of_spectrum_samples;
    vec3 ps = sample_pixel_point((film_sample_scheme)p->pixel_scheme);
    f64 film_x = ((f64)x + ps.x) * cam.pixel_width;
    f64 film_y = ((f64)y + ps.y) * cam.pixel_height;
    vec3 pp = vec3_sum(vec3_sum(vec3_mul_by_f64(cam.right, film_x), vec3_mul_by_f64(cam.up, film_y)), cam.film_bottom_left);
    if (cam.aperture_radius > 0.0) return -1; /* pinhole only; thin-lens paths are compared through spectra */
    vec3 ro = pp;
    vec3 rd = vec3_normalise(vec3_sub(cam.aperture_position, ro));
    vec3 rd0 = rd;
    spectrum dst; dst.samples = contribution;
    zero_spectrum(dst);
    spectrum contrib = alloc_spd(), thr = alloc_spd(), refl = alloc_spd(), tmp = alloc_spd();
    const_spectrum(thr, 1.0);
    scene_point ip;
    memset(&ip, 0, sizeof(ip));
    int scans = 0;
    for (uint32_t d = 0; d < p->max_depth; d += 1) hit_seq[d] = -2;
    for (uint32_t depth = 0; depth < p->max_depth; depth += 1)
    {
        ip.surface = NULL;
        find_ray_intersection(&ip, &g_scene, ro, rd);
        int missed = (ip.surface_material == g_scene.escape_material && ip.surface == NULL);
        hit_seq[depth] = missed ? -1 : (int32_t)(ip.surface - g_scene.surfaces);
        scans += 1;
        object_material *mat = ip.surface_material;
        if (mat->is_black_body && !mat->is_emissive) break;
        else if (mat->is_black_body && mat->is_emissive)
        {
            spectral_mul_by_spectrum(tmp, thr, mat->emission_spd);
            spectral_sum(dst, dst, tmp);
            break;
        }
        direct_light_contribution(contrib, &ip, &g_scene);
        spectral_mul_by_spectrum(tmp, thr, contrib);
        spectral_sum(dst, dst, tmp);
        vec3 in; f64 dir_pdf;
        mat->sample_direction(&in, &dir_pdf, &ip);
        bdsf(refl, &ip, in);
        spectral_mul_by_scalar(refl, refl, dir_pdf);
        spectral_mul_by_spectrum(thr, thr, refl);
        rd = in;
        ro = ip.position;
    }
    free_spd(tmp); free_spd(refl); free_spd(thr); free_spd(contrib);
    spectral_mul_by_scalar(dst, dst, vec3_dot(rd0, cam.forward) * 1.0);
    (void)S;
    return scans;
}

/* The pixel loop of render_image (src/daily_ray_trace.c:710-745) over a tile. The loop's BODY is the reference's own text
 * (lines 729-743, compiled in place from drt_ref_pixel_body.inc); this function only supplies the variables that body
 * names -- with the reference's names and types -- and seeds the per-path generator before it. */
REF_API void ref_render_tile(const drt_camera *c, const drt_params *p, double *pixels, double *avgs, double *vars)
{
    camera_data camera = cam_in(c);
    scene_data scene = g_scene; /* shallow copy: the body says &scene */
    uint32_t S = number_of_spectrum_samples;
    f64 *contribution_buffer = (f64 *)calloc(S + 1, sizeof(f64));
    spectrum contribution; contribution.samples = contribution_buffer;
    f64 *filter = &contribution_buffer[number_of_spectrum_samples];
    spectrum tmp_0_spd = alloc_spd(), tmp_1_spd = alloc_spd();
    u32 max_cast_depth = p->max_depth;
    config_arguments config_storage;
    config_arguments *config = &config_storage;
    config->pixel_scheme = (film_sample_scheme)p->pixel_scheme;
    uint32_t stride = p->row_stride ? p->row_stride : 1;
    for (uint32_t s = 0; s < p->spp; s += 1)
    {
        u32 sample = p->first_sample + s;
        for (uint32_t j = 0; j < p->tile_h; j += 1)
        {
            u32 y = p->y0 + j * stride;
            for (uint32_t i = 0; i < p->tile_w; i += 1)
            {
                u32 x = p->x0 + i;
                uint64_t off = (uint64_t)j * p->tile_w + i;
                f64 *dst_pixel = pixels + off * (S + 1);
                spectrum dst_pixel_spd, dst_pixel_avg, dst_pixel_var;
                dst_pixel_spd.samples = dst_pixel;
                dst_pixel_avg.samples = avgs + off * S;
                dst_pixel_var.samples = vars + off * S;
                ref_seed_path(p->seed + (((uint64_t)sample * p->height + y) * (uint64_t)p->width + x));
#include "drt_ref_pixel_body.inc"
            }
        }
    }
    free_spd(tmp_1_spd); free_spd(tmp_0_spd);
    free(contribution_buffer);
}

/* ---- parser exports --------------------------------------------------------------------------- */
static camera_input_data g_parsed_camera;
static scene_input_data  g_parsed_scene;

/* parse_scene on a copy of `text`. Returns 0, or the code parse_error() would have ended the process with
 * (its exit(-1) arrives here as 255 / -1 -> nonzero). Counts come back through the pointers. */
REF_API int ref_parse_scene(const char *text, uint32_t size, uint32_t *num_materials, uint32_t *num_surfaces)
{
    char *copy = (char *)calloc((size_t)size + 2, 1);
    memcpy(copy, text, size);
    memset(&g_parsed_camera, 0, sizeof(g_parsed_camera));
    memset(&g_parsed_scene, 0, sizeof(g_parsed_scene));
    memset(&tokeniser, 0, sizeof(tokeniser));
    int rc = 0;
    g_parse_active = 1;
    if ((rc = setjmp(g_parse_jmp)) == 0) parse_scene(copy, size, &g_parsed_camera, &g_parsed_scene);
    g_parse_active = 0;
    free(copy);
    if (num_materials) *num_materials = g_parsed_scene.num_scene_materials;
    if (num_surfaces) *num_surfaces = g_parsed_scene.num_surfaces;
    return rc;
}
/* target, position, roll, fov, fdepth, flength, aperture */
REF_API void ref_parsed_camera(double out[11])
{
    v3_out(g_parsed_camera.target, out); v3_out(g_parsed_camera.position, out + 3);
    out[6] = g_parsed_camera.roll; out[7] = g_parsed_camera.fov; out[8] = g_parsed_camera.fdepth;
    out[9] = g_parsed_camera.flength; out[10] = g_parsed_camera.aperture;
}
typedef struct
{
    uint32_t method, has_scale_factor;
    double   scale_factor;
    double   value[3]; /* rgb; or value[0] = blackbody temperature / constant */
    char     csv[64];
} ref_spd_input;
typedef struct
{
    char     name[32];
    uint32_t is_base_material, is_escape_material, is_black_body, is_emissive;
    double   shininess, roughness;
    ref_spd_input spd[6]; /* emission, diffuse, glossy, mirror, refract, extinct */
    uint32_t num_bdsfs;
    int32_t  bdsfs[16]; /* index into bdsf_list, -1 when the pointer is none of them */
    int32_t  dir_func;
} ref_material_input;
typedef struct
{
    char     name[32], material_name[32];
    uint32_t type, pad;
    double   position[3], radius, normal[3], u[3], v[3];
} ref_surface_input;

static void spd_input_out(const spd_input_data *in, ref_spd_input *o)
{
    memset(o, 0, sizeof(*o));
    o->method = (uint32_t)in->method;
    o->has_scale_factor = in->has_scale_factor;
    o->scale_factor = in->scale_factor;
    switch (in->method)
    {
        case SPD_METHOD_RGB: o->value[0] = in->rgb.r; o->value[1] = in->rgb.g; o->value[2] = in->rgb.b; break;
        case SPD_METHOD_CSV: memcpy(o->csv, in->csv, 64); break;
        case SPD_METHOD_BLACKBODY: o->value[0] = in->blackbody_temp; break;
        case SPD_METHOD_CONST: o->value[0] = in->constant; break;
        default: break;
    }
}
REF_API int ref_parsed_material(uint32_t i, ref_material_input *o)
{
    if (i >= 16) return -1;
    const material_input_data *m = &g_parsed_scene.scene_materials[i];
    memset(o, 0, sizeof(*o));
    memcpy(o->name, m->name, 32);
    o->is_base_material = m->is_base_material; o->is_escape_material = m->is_escape_material;
    o->is_black_body = m->is_black_body; o->is_emissive = m->is_emissive;
    o->shininess = m->shininess; o->roughness = m->roughness;
    spd_input_out(&m->emission_input, &o->spd[0]); spd_input_out(&m->diffuse_input, &o->spd[1]);
    spd_input_out(&m->glossy_input, &o->spd[2]); spd_input_out(&m->mirror_input, &o->spd[3]);
    spd_input_out(&m->refract_input, &o->spd[4]); spd_input_out(&m->extinct_input, &o->spd[5]);
    o->num_bdsfs = m->num_bdsfs;
    for (uint32_t j = 0; j < 16; j += 1)
    {
        o->bdsfs[j] = -1;
        for (uint32_t k = 0; k < num_bdsfs_defined; k += 1) if (m->bdsfs[j] == bdsf_list[k]) o->bdsfs[j] = (int32_t)k;
    }
    o->dir_func = -1;
    for (uint32_t k = 0; k < num_dir_funcs_defined; k += 1) if (m->sample_direction_function == dir_func_list[k]) o->dir_func = (int32_t)k;
    return 0;
}
REF_API int ref_parsed_surface(uint32_t i, ref_surface_input *o)
{
    if (i >= 16) return -1;
    const surface_input_data *s = &g_parsed_scene.surfaces[i];
    memset(o, 0, sizeof(*o));
    memcpy(o->name, s->name, 32);
    memcpy(o->material_name, s->material_name, 32);
    o->type = (uint32_t)s->type;
    v3_out(s->position, o->position);
    if (s->type == GEO_TYPE_SPHERE) o->radius = s->radius;
    else
    {
        v3_out(s->normal, o->normal); v3_out(s->u, o->u); v3_out(s->v, o->v);
    }
    return 0;
}
/* parse_config into the reference's own 1136-byte config_arguments (the caller's buffer is zeroed first). */
REF_API int ref_parse_config(const char *text, uint32_t size, void *config_out, uint32_t config_size)
{
    if (config_size != sizeof(config_arguments)) return -2;
    char *copy = (char *)calloc((size_t)size + 2, 1);
    memcpy(copy, text, size);
    memset(config_out, 0, sizeof(config_arguments));
    memset(&tokeniser, 0, sizeof(tokeniser));
    int rc = 0;
    g_parse_active = 1;
    if ((rc = setjmp(g_parse_jmp)) == 0) parse_config(copy, size, (config_arguments *)config_out);
    g_parse_active = 0;
    free(copy);
    return rc;
}
REF_API uint32_t ref_sizeof_config(void) { return (uint32_t)sizeof(config_arguments); }

/* The reference's CSV resampling (the body of load_csv_file_to_spectrum, src/read_scene.c:812-872) on the bytes of `path`,
 * read here with stdio into a zero-filled buffer one byte larger than the file, as the reference's alloc() + read leave it.
 * The grid is the one ref_set_grid() set. Returns 0 when the file cannot be read. */
REF_API int ref_csv_to_spectrum(const char *path, double *dst)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)calloc((size_t)n + 4, 1);
    if (fread(buf, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(buf); return 0; }
    fclose(f);
    spectrum s; s.samples = dst;
    u32 ok = ref_csv_body(s, buf, (u32)n + 1);
    free(buf);
    return (int)ok;
}
REF_API void ref_spectrum_normalise(double *spd) { spectrum s; s.samples = spd; spectrum_normalise(s); }
REF_API void ref_spectral_mul_by_scalar(double *spd, double f) { spectrum s; s.samples = spd; spectral_mul_by_scalar(s, s, f); }
REF_API void ref_const_spectrum(double *spd, double f) { spectrum s; s.samples = spd; const_spectrum(s, f); }

/* ---- the reference-side binding of INTEGRATION.md, compiled as the maintainer would add it ------------------ */
/* include/drt_reference_binding.inc is the text INTEGRATION.md shows. Its one call into the library goes through a
 * pointer here, so that a test can hand it any implementation of the boundary (libdrt_hip.so's, or the oracle behind the
 * same signature) without this library linking against either. */
typedef int (*render_tile_fn)(const drt_scene *, const drt_camera *, const drt_params *, double *, double *, double *, drt_stats *);
static render_tile_fn g_render_tile = NULL;
static const char *binding_last_error(void) { return "the boundary implementation handed to ref_run_binding failed"; }
#define drt_render_tile g_render_tile
#define drt_last_error binding_last_error
#define printf ref_quiet_printf
#define exit ref_parse_exit
#include "../include/drt_reference_binding.inc"
#undef exit
#undef printf
#undef drt_last_error
#undef drt_render_tile

/* Runs render_pixels_mi355x() on the scene ref_set_scene() built, with config_arguments / camera_data made from the
 * boundary structs, and `fn` as drt_render_tile. Returns 0, or nonzero when the stub took its exit(-1) path. */
REF_API int ref_run_binding(const drt_camera *c, const drt_params *p, render_tile_fn fn, double *pixels, double *avgs, double *vars)
{
    camera_data camera = cam_in(c);
    config_arguments config;
    memset(&config, 0, sizeof(config));
    config.output_width = p->width;
    config.output_height = p->height;
    config.num_pixel_samples = p->spp;
    config.max_cast_depth = p->max_depth;
    config.pixel_scheme = (film_sample_scheme)p->pixel_scheme;
    g_render_tile = fn;
    int rc = 0;
    g_parse_active = 1;
    if ((rc = setjmp(g_parse_jmp)) == 0) render_pixels_mi355x(&config, &g_scene, &camera, pixels, avgs, vars);
    g_parse_active = 0;
    g_render_tile = NULL;
    return rc;
}

REF_API int ref_version(void) { return 2; }
