/*
 * drt_oracle.c -- CPU restatement (plain C, scalar f64) of daily-ray-trace's render path.
 *
 * TEST INFRASTRUCTURE ONLY -- see drt_oracle.h. Every function cites the reference lines it
 * follows (paths relative to the reference tree). The restatement keeps every quirk of the
 * reference: BDSF carry-over, sin^4 in the dielectric Fresnel, area-only
 * light weights, multiplicative multi-light accumulation, upper-hemisphere sphere lights,
 * exact vec3 equality tests, trans_wl = 630, rng() in [0,1] inclusive.
 *
 * Build: gcc -O2 -ffp-contract=off (x86-64/SSE2; long double = x87 80-bit for REFERENCE mode).
 */
#include "drt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y, z; } v3;
typedef struct { v3 c[3]; } m33; /* columns, like mat3x3 (src/geometry.h:32-35) */

#define PI_L 3.1415926535897932385L /* src/types.h:1 */
static const double PI_D    = 3.14159265358979323846;
static const double VIS_FUDGE = 0.0001; /* src/daily_ray_trace.c:237 */

static int g_math_mode = DRT_ORACLE_MATH_REFERENCE;
void drt_oracle_set_math_mode(int mode) { g_math_mode = mode; }
int  drt_oracle_get_math_mode(void) { return g_math_mode; }
#define REFMODE (g_math_mode == DRT_ORACLE_MATH_REFERENCE)

/* per-thread state: the reference keeps one global libc stream; the build's RNG is per path */
static __thread uint64_t t_rng_state = 1;
static __thread uint64_t t_rng_draws = 0;
static __thread uint64_t t_closest_scans = 0;
static __thread uint64_t t_shadow_scans = 0;
static __thread uint64_t t_shaded = 0;

/* ------------------------------------------------------------------------------------------ */
/* vec3 / mat3 : src/geometry.c:6-106, 211-295                                                 */

static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 from3(const double a[3]) { return V(a[0], a[1], a[2]); }
static void to3(v3 a, double o[3]) { o[0] = a.x; o[1] = a.y; o[2] = a.z; }
static int v_equal(v3 a, v3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }     /* :6-9   */
static v3 v_sum(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }             /* :11-18 */
static v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }             /* :20-27 */
static double v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          /* :29-33 */
static v3 v_cross(v3 a, v3 b)                                                           /* :35-42 */
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static v3 v_mul(v3 v, double f) { return V(f * v.x, f * v.y, f * v.z); }               /* :44-51 */
static v3 v_div(v3 v, double f) { return V(v.x / f, v.y / f, v.z / f); }               /* :53-60 */
static double v_length(v3 v) { return sqrt(v_dot(v, v)); }                             /* :62-67 */
static v3 v_normalise(v3 v) { return v_div(v, v_length(v)); }                          /* :69-74 */
static v3 v_reverse(v3 v) { return V(-v.x, -v.y, -v.z); }                              /* :76-83 */

/* vec3_reflect, src/geometry.c:85-90 */
static v3 v_reflect(v3 v, v3 n)
{
    double f = 2.0 * v_dot(v, n);
    return v_sub(v, v_mul(n, f));
}

/* vec3_transmit, src/geometry.c:92-106 (NaN on total internal reflection) */
static v3 v_transmit(v3 v, v3 n, double ir, double tr)
{
    double vn_dot  = v_dot(v, n);
    double rel_ref = ir / tr;
    v3 m = v_mul(n, vn_dot);
    v = v_sub(m, v);
    v3 perpend = v_reverse(v_mul(v, rel_ref));
    double perpend_dot = -sqrt(1.0 - v_dot(perpend, perpend));
    v3 parallel = v_mul(n, perpend_dot);
    return v_sum(perpend, parallel);
}

static double m_get(const m33 *m, int col, int row)
{
    const v3 *c = &m->c[col];
    return row == 0 ? c->x : (row == 1 ? c->y : c->z);
}
static void m_set(m33 *m, int col, int row, double f)
{
    v3 *c = &m->c[col];
    if (row == 0) c->x = f; else if (row == 1) c->y = f; else c->z = f;
}
static v3 m_row(const m33 *m, int r) { return V(m_get(m, 0, r), m_get(m, 1, r), m_get(m, 2, r)); } /* :211-218 */
static v3 m_vmul(const m33 *m, v3 v)                                                               /* :220-228 */
{
    return V(v_dot(m_row(m, 0), v), v_dot(m_row(m, 1), v), v_dot(m_row(m, 2), v));
}

/* find_rotation_between_vectors, src/geometry.c:263-295 (Rodrigues; antiparallel -> -I).
 * mat3x3_mul (:240-252) stores row(m,i).col(n,j) at columns[i].xyz[j]; kept as written. */
static m33 rotation_between(v3 v, v3 w)
{
    v3 n = v_cross(v, w);
    double c = v_dot(v, w);
    m33 r = {{{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}}};
    if (v_dot(n, n) == 0.0 && c <= 0.0)
    {
        r.c[0].x = -1.0;
        r.c[1].y = -1.0;
        r.c[2].z = -1.0;
    }
    else
    {
        m33 m;
        m.c[0] = V(0.0, n.z, -n.y);
        m.c[1] = V(-n.z, 0.0, n.x);
        m.c[2] = V(n.y, -n.x, 0.0);
        m33 mm;
        for (int i = 0; i < 3; i += 1)
            for (int j = 0; j < 3; j += 1)
                m_set(&mm, i, j, v_dot(m_row(&m, i), m.c[j]));
        double f = 1.0 / (1.0 + c);
        for (int i = 0; i < 3; i += 1) mm.c[i] = v_mul(mm.c[i], f);
        m33 id = {{{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}}};
        for (int i = 0; i < 3; i += 1) r.c[i] = v_sum(v_sum(id.c[i], m.c[i]), mm.c[i]);
    }
    return r;
}

/* rotation_about_axis, src/geometry.c:297-313. Column 0 row z uses axis.z*axis.z (as written). */
static m33 rotation_about_axis(v3 axis, double angle_rad)
{
    double cos_th = cos(angle_rad);
    double sin_th = sin(angle_rad);
    m33 r;
    r.c[0].x = cos_th + (axis.x * axis.x) * (1 - cos_th);
    r.c[0].y = axis.y * axis.x * (1 - cos_th) + axis.z * sin_th;
    r.c[0].z = axis.z * axis.z * (1 - cos_th) - axis.y * sin_th;
    r.c[1].x = axis.x * axis.y * (1 - cos_th) - axis.z * sin_th;
    r.c[1].y = cos_th + (axis.y * axis.y) * (1 - cos_th);
    r.c[1].z = axis.z * axis.y * (1 - cos_th) + axis.x * sin_th;
    r.c[2].x = axis.x * axis.z * (1 - cos_th) + axis.y * sin_th;
    r.c[2].y = axis.y * axis.z * (1 - cos_th) - axis.x * sin_th;
    r.c[2].z = cos_th + axis.z * axis.z * (1 - cos_th);
    return r;
}

/* ------------------------------------------------------------------------------------------ */
/* Intersectors: src/geometry.c:123-182                                                         */

static double line_sphere(v3 o, v3 d, v3 sc, double sr) /* :123-146 */
{
    v3 c_to_o = v_sub(o, sc);
    double a = 1.0;
    double b = -2.0 * v_dot(c_to_o, d);
    double c = v_dot(c_to_o, c_to_o) - sr * sr;
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant < 0.0) return INFINITY;
    double sq = sqrt(discriminant);
    double a_2 = 2.0 * a;
    double s0 = (b + sq) / a_2;
    double s1 = (b - sq) / a_2;
    if (s0 < 0.0 && s1 < 0.0) return INFINITY;
    else if (s0 >= 0.0 && s1 < 0.0) return s0;
    else if (s1 >= 0.0 && s0 < 0.0) return s1;
    else if (s0 <= s1) return s0;
    else return s1;
}

static double line_plane(v3 o, v3 d, v3 pp, v3 pn, v3 pu, v3 pv) /* :157-182 */
{
    if (v_dot(d, pn) == 0.0) return INFINITY;
    v3 o_to_p = v_sub(pp, o);
    double l = v_dot(o_to_p, pn) / v_dot(d, pn);
    v3 i = v_sum(o, v_mul(d, l));
    v3 j = v_sub(i, pp);
    double ul = v_length(pu);
    double vl = v_length(pv);
    v3 un = v_normalise(pu);
    v3 vn = v_normalise(pv);
    double ju = v_dot(j, un);
    double jv = v_dot(j, vn);
    if (l >= 0.0 && 0.0 <= ju && ju <= ul && 0.0 <= jv && jv <= vl) return l;
    return INFINITY;
}

/* ------------------------------------------------------------------------------------------ */
/* RNG: the seam is src/rng.h:1-2; the generator is the build's               */

static uint64_t splitmix64(uint64_t k)
{
    uint64_t z = k + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z ? z : 1ull;
}
void drt_oracle_seed_path(uint64_t key) { t_rng_state = splitmix64(key); }
void drt_oracle_set_rng_state(uint64_t s) { t_rng_state = s; }
uint64_t drt_oracle_get_rng_state(void) { return t_rng_state; }
uint64_t drt_oracle_path_key(uint64_t seed, uint32_t width, uint32_t height, uint32_t x, uint32_t y, uint32_t sample)
{
    return seed + (((uint64_t)sample * (uint64_t)height + (uint64_t)y) * (uint64_t)width + (uint64_t)x);
}
/* rng(), src/rng.c:2-7 with rand() := xorshift64 >> 33, RAND_MAX := 2^31-1 (glibc's) */
static double rng(void)
{
    uint64_t x = t_rng_state;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    t_rng_state = x;
    t_rng_draws += 1;
    double r = (double)(uint32_t)(x >> 33);
    return r / 2147483647.0;
}
double drt_oracle_rng(void) { return rng(); }

/*
 * The path's sincos (DEVICE arithmetic). Spec, identical op for op in the HIP kernels:
 *   fn = rint(t * 2/pi); r = t - fn*PIO2_1; w = fn*PIO2_1T; y0 = r - w; y1 = (r - y0) - w
 *   z = y0*y0
 *   ks = y0 - ((z*(0.5*y1 - (z*y0)*(S2+z*(S3+z*(S4+z*(S5+z*S6))))) - y1) - (z*y0)*S1)
 *   kc = w1 + (((1-w1) - 0.5*z) + (z*(z*(C1+z*(C2+z*(C3+z*(C4+z*(C5+z*C6)))))) - y0*y1)),  w1 = 1 - 0.5*z
 *   quadrant n = (int)fn & 3: (s,c) = (ks,kc) | (kc,-ks) | (-ks,-kc) | (-kc,ks)
 * Constants are the classic Sun fdlibm minimax coefficients; valid for |t| < ~1e5, < 1 ulp.
 */
static const double SC_TWO_OVER_PI = 6.36619772367581382433e-01;
static const double SC_PIO2_1  = 1.57079632673412561417e+00;
static const double SC_PIO2_1T = 6.07710050650619224932e-11;
static const double SC_S1 = -1.66666666666666324348e-01, SC_S2 = 8.33333333332248946124e-03,
                    SC_S3 = -1.98412698298579493134e-04, SC_S4 = 2.75573137070700676789e-06,
                    SC_S5 = -2.50507602534068634195e-08, SC_S6 = 1.58969099521155010221e-10;
static const double SC_C1 = 4.16666666666666019037e-02, SC_C2 = -1.38888888888741095749e-03,
                    SC_C3 = 2.48015872894767294178e-05, SC_C4 = -2.75573143513906633035e-07,
                    SC_C5 = 2.08757232129817482790e-09, SC_C6 = -1.13596475577881948265e-11;

void drt_oracle_sincos(double t, double *s, double *c)
{
    double fn = rint(t * SC_TWO_OVER_PI);
    double r  = t - fn * SC_PIO2_1;
    double w  = fn * SC_PIO2_1T;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double z  = y0 * y0;
    double v  = z * y0;
    double rs = SC_S2 + z * (SC_S3 + z * (SC_S4 + z * (SC_S5 + z * SC_S6)));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * SC_S1);
    double rc = z * (SC_C1 + z * (SC_C2 + z * (SC_C3 + z * (SC_C4 + z * (SC_C5 + z * SC_C6)))));
    double hz = 0.5 * z;
    double w1 = 1.0 - hz;
    double kc = w1 + (((1.0 - w1) - hz) + (z * rc - y0 * y1));
    int n = (int)fn & 3;
    if (n == 0) { *s = ks; *c = kc; }
    else if (n == 1) { *s = kc; *c = -ks; }
    else if (n == 2) { *s = -ks; *c = -kc; }
    else { *s = -kc; *c = ks; }
}

static void path_sincos(double t, double *s, double *c)
{
    if (REFMODE) { *c = cos(t); *s = sin(t); }
    else drt_oracle_sincos(t, s, c);
}

/* t = 2.0 * PI * v  (src/rng.c:19, src/bdsf.c:268, src/daily_ray_trace.c:301) */
static double two_pi_times(double v)
{
    if (REFMODE) return (double)(2.0 * PI_L * v);
    return (2.0 * PI_D) * v;
}

/* uniform_sample_sphere, src/rng.c:14-23 */
static v3 uniform_sample_sphere(void)
{
    double u = rng();
    double v = rng();
    double r = sqrt(1.0 - u * u);
    double t = two_pi_times(v);
    double st, ct;
    path_sincos(t, &st, &ct);
    return V(r * ct, r * st, u);
}

/* uniform_sample_disc, src/rng.c:25-51 (Shirley concentric map) */
static v3 uniform_sample_disc(void)
{
    v3 v = {0.0, 0.0, 0.0};
    double r_x = rng();
    double r_y = rng();
    double o_x = 2.0 * r_x - 1.0;
    double o_y = 2.0 * r_y - 1.0;
    if (o_x == 0.0 && o_y == 0.0) return v;
    double r, t;
    if (fabs(o_x) > fabs(o_y))
    {
        r = o_x;
        if (REFMODE) t = (double)((PI_L / 4.0) * (o_y / o_x));
        else t = (PI_D / 4.0) * (o_y / o_x);
    }
    else
    {
        r = o_y;
        if (REFMODE) t = (double)((PI_L / 2.0) - (PI_L / 4.0) * (o_x / o_y));
        else t = (PI_D / 2.0) - (PI_D / 4.0) * (o_x / o_y);
    }
    double st, ct;
    path_sincos(t, &st, &ct);
    v.x = r * ct;
    v.y = r * st;
    return v;
}
void drt_oracle_uniform_sample_sphere(double out[3]) { to3(uniform_sample_sphere(), out); }
void drt_oracle_uniform_sample_disc(double out[3]) { to3(uniform_sample_disc(), out); }

/* ------------------------------------------------------------------------------------------ */
/* Scene access                                                                                 */

#define MAX_S 512
static const double g_zero_spd[MAX_S];

static const double *spd_of(const drt_scene *sc, int32_t idx)
{
    if (idx < 0) return g_zero_spd; /* NULL spectrum in the reference */
    return sc->spds + (size_t)idx * sc->num_wavelengths;
}

/* lerp, src/utils.c:1-4 */
static double lerp(double x, double x0, double x1, double y0, double y1)
{
    return y0 + ((x - x0) * ((y1 - y0) / (x1 - x0)));
}

/* value_at_wl, src/spectrum.c:150-162 */
static double value_at_wl(const drt_scene *sc, const double *spd, double wl)
{
    uint32_t i_0 = (uint32_t)((wl - sc->min_wavelength) / sc->wavelength_interval);
    uint32_t i_1 = i_0 + 1;
    double w_0 = sc->min_wavelength + i_0 * sc->wavelength_interval;
    double w_1 = sc->min_wavelength + i_1 * sc->wavelength_interval;
    return lerp(wl, w_0, w_1, spd[i_0], spd[i_1]);
}
double drt_oracle_value_at_wl(const drt_scene *sc, const double *spd, double wl) { return value_at_wl(sc, spd, wl); }

/* The working point of the integrator: scene_point, src/daily_ray_trace.h:113-125 */
typedef struct
{
    v3 position, normal, out;
    double on_dot, trans_wl;
    const drt_material *surface_material, *incident_material, *transmit_material;
    int surface_index;
} point;

static void point_from_api(const drt_scene *sc, const drt_oracle_point *a, point *p)
{
    p->position = from3(a->position);
    p->normal = from3(a->normal);
    p->out = from3(a->out);
    p->on_dot = a->on_dot;
    p->trans_wl = a->trans_wl;
    p->surface_material = &sc->materials[a->surface_material];
    p->incident_material = &sc->materials[a->incident_material];
    p->transmit_material = &sc->materials[a->transmit_material];
    p->surface_index = -1;
}

/* ------------------------------------------------------------------------------------------ */
/* src/bdsf.c                                                                                   */

static double ggx(v3 sn, v3 mn, double r) /* :3-20 */
{
    double g;
    double d = v_dot(sn, mn);
    double r_2 = r * r;
    if (d <= 0.0) g = 0.0;
    else
    {
        double d_2 = d * d;
        double d_4 = d_2 * d_2;
        double tan_sq = (1.0 / d_2) - 1.0;
        if (REFMODE) g = (double)(r_2 / (PI_L * d_4 * (r_2 + tan_sq) * (r_2 + tan_sq)));
        else g = r_2 / (((PI_D * d_4) * (r_2 + tan_sq)) * (r_2 + tan_sq));
    }
    return g;
}

static double ggx_att(v3 v, v3 sn, v3 mn, double r) /* :22-42 */
{
    double att;
    double g = ggx(sn, mn, r);
    double v_mn = v_dot(v, mn);
    double v_sn = v_dot(v, sn);
    double dot_quot = fabs(v_mn / v_sn);
    double r_2 = r * r;
    if (dot_quot <= 0.0) att = 0.0;
    else
    {
        double vn_tan_sq = (1.0 / (v_sn * v_sn)) - 1.0;
        att = 2.0 / (1.0 + sqrt(1.0 + r_2 * vn_tan_sq));
    }
    return g * att;
}

/* fs_dielectric_reflectance, :44-67. ts_cos squares the already squared sine (Q2). */
static void fs_dielectric_reflectance(double *refl, const double *ir, const double *tr, double inc_cos, uint32_t S)
{
    double inc_sin_sq = 1.0 - inc_cos * inc_cos;
    for (uint32_t i = 0; i < S; i += 1)
    {
        double rel = ir[i] / tr[i];
        double ts_sin_sq = rel * rel * inc_sin_sq;
        if (ts_sin_sq >= 1.0)
        {
            refl[i] = 1.0;
            continue;
        }
        double ts_cos = sqrt(1.0 - ts_sin_sq * ts_sin_sq);
        double tr_on = tr[i] * inc_cos;
        double tr_ts = tr[i] * ts_cos;
        double ir_on = ir[i] * inc_cos;
        double ir_ts = ir[i] * ts_cos;
        double par = (tr_on - ir_ts) / (tr_on + ir_ts);
        double per = (ir_on - tr_ts) / (ir_on + tr_ts);
        par *= par;
        per *= per;
        refl[i] = 0.5 * (par + per);
    }
}

static void fs_dielectric_transmittance(double *t, const double *ir, const double *tr, double inc_cos, uint32_t S) /* :69-76 */
{
    fs_dielectric_reflectance(t, ir, tr, inc_cos, S);
    for (uint32_t i = 0; i < S; i += 1) t[i] = 1.0 - t[i];
}

static void fs_conductor_reflectance(double *refl, const double *ir, const double *tr, const double *te, double inc_cos, uint32_t S) /* :78-101 */
{
    double inc_cos_sq = inc_cos * inc_cos;
    double inc_sin_sq = 1.0 - inc_cos_sq;
    for (uint32_t i = 0; i < S; i += 1)
    {
        double rr = tr[i] / ir[i];
        double re = te[i] / ir[i];
        double rr_sq = rr * rr;
        double re_sq = re * re;
        double r = rr_sq - re_sq - inc_sin_sq;
        double apb_sq = sqrt(r * r + 4.0 * rr_sq * re_sq);
        double a = sqrt(0.5 * (apb_sq + r));
        double s = apb_sq + inc_cos_sq;
        double t = 2.0 * a * inc_cos;
        double u = inc_cos_sq * apb_sq + inc_sin_sq * inc_sin_sq;
        double v = t * inc_sin_sq;
        double par = (s - t) / (s + t);
        double per = par * (u - v) / (u + v);
        refl[i] = 0.5 * (par + per);
    }
}

static void bp_diffuse_bdsf(const drt_scene *sc, double *out, const point *p, v3 in) /* :105-109 */
{
    uint32_t S = sc->num_wavelengths;
    const double *d = spd_of(sc, p->surface_material->diffuse_spd);
    double inv_pi = REFMODE ? (double)(1.0 / PI_L) : 1.0 / PI_D;
    for (uint32_t i = 0; i < S; i += 1) out[i] = d[i] * inv_pi;
    double a = fabs(v_dot(p->normal, in));
    for (uint32_t i = 0; i < S; i += 1) out[i] = out[i] * a;
}

static void bp_glossy_bdsf(const drt_scene *sc, double *out, const point *p, v3 in) /* :111-119 */
{
    uint32_t S = sc->num_wavelengths;
    v3 bisector = v_normalise(v_sum(p->out, in));
    double nb = v_dot(p->normal, bisector);
    double spec = pow((0.0 > nb) ? 0.0 : nb, p->surface_material->shininess); /* f64_max, src/utils.c:13-16 */
    const double *g = spd_of(sc, p->surface_material->glossy_spd);
    for (uint32_t i = 0; i < S; i += 1) out[i] = g[i] * spec;
    double a = fabs(v_dot(p->normal, in));
    for (uint32_t i = 0; i < S; i += 1) out[i] = out[i] * a;
}

static void mirror_bdsf(const drt_scene *sc, double *out, const point *p, v3 in) /* :121-132 */
{
    uint32_t S = sc->num_wavelengths;
    v3 outgoing = v_reverse(p->out);
    if (v_equal(in, v_reflect(outgoing, p->normal)))
        memcpy(out, spd_of(sc, p->surface_material->mirror_spd), S * sizeof(double));
    else
        memset(out, 0, S * sizeof(double));
}

static void fs_conductor_bdsf(const drt_scene *sc, double *out, const point *p, v3 in) /* :134-146 */
{
    v3 outgoing = v_reverse(p->out);
    if (v_equal(in, v_reflect(outgoing, p->normal)))
    {
        fs_conductor_reflectance(out, spd_of(sc, p->incident_material->refract_spd),
                                 spd_of(sc, p->transmit_material->refract_spd),
                                 spd_of(sc, p->transmit_material->extinct_spd), p->on_dot, sc->num_wavelengths);
    }
}

static void fs_dielectric_reflectance_bdsf(const drt_scene *sc, double *out, const point *p, v3 in) /* :148-159 */
{
    v3 outgoing = v_reverse(p->out);
    if (v_equal(in, v_reflect(outgoing, p->normal)))
    {
        fs_dielectric_reflectance(out, spd_of(sc, p->incident_material->refract_spd),
                                  spd_of(sc, p->transmit_material->refract_spd), p->on_dot, sc->num_wavelengths);
    }
}

static void fs_dielectric_transmittance_bdsf(const drt_scene *sc, double *out, const point *p, v3 in) /* :161-172 */
{
    v3 outgoing = v_reverse(p->out);
    const double *ir_spd = spd_of(sc, p->incident_material->refract_spd);
    const double *tr_spd = spd_of(sc, p->transmit_material->refract_spd);
    double ir = value_at_wl(sc, ir_spd, p->trans_wl);
    double tr = value_at_wl(sc, tr_spd, p->trans_wl);
    if (v_equal(in, v_transmit(outgoing, p->normal, ir, tr)))
        fs_dielectric_transmittance(out, ir_spd, tr_spd, p->on_dot, sc->num_wavelengths);
}

static void ct_conductor_bdsf(const drt_scene *sc, double *out, const point *p, v3 in) /* :174-186 */
{
    uint32_t S = sc->num_wavelengths;
    v3 micro_normal = v_normalise(v_sum(p->out, in));
    double mn_dot = fabs(v_dot(p->normal, micro_normal));
    fs_conductor_reflectance(out, spd_of(sc, p->incident_material->refract_spd),
                             spd_of(sc, p->transmit_material->refract_spd),
                             spd_of(sc, p->transmit_material->extinct_spd), mn_dot, S);
    double coef = ggx_att(p->out, p->normal, micro_normal, p->surface_material->roughness) * (1.0 / (4.0 * p->on_dot));
    for (uint32_t i = 0; i < S; i += 1) out[i] = out[i] * coef;
}

static void bdsf_call(const drt_scene *sc, uint32_t id, double *out, const point *p, v3 in)
{
    switch (id)
    {
        case DRT_BDSF_bp_diffuse_bdsf: bp_diffuse_bdsf(sc, out, p, in); break;
        case DRT_BDSF_bp_glossy_bdsf: bp_glossy_bdsf(sc, out, p, in); break;
        case DRT_BDSF_mirror_bdsf: mirror_bdsf(sc, out, p, in); break;
        case DRT_BDSF_fs_conductor_bdsf: fs_conductor_bdsf(sc, out, p, in); break;
        case DRT_BDSF_fs_dielectric_reflectance_bdsf: fs_dielectric_reflectance_bdsf(sc, out, p, in); break;
        case DRT_BDSF_fs_dielectric_transmittance_bdsf: fs_dielectric_transmittance_bdsf(sc, out, p, in); break;
        case DRT_BDSF_ct_conductor_bdsf: ct_conductor_bdsf(sc, out, p, in); break;
        default: break;
    }
}

/* bdsf(), src/daily_ray_trace.c:215-229: bdsf_result is zeroed ONCE; a function that early-outs
 * leaves the previous function's result in it, and it is added again (Q1). */
static void bdsf(const drt_scene *sc, double *reflectance, const point *p, v3 in)
{
    uint32_t S = sc->num_wavelengths;
    double bdsf_result[MAX_S];
    memset(bdsf_result, 0, S * sizeof(double));
    memset(reflectance, 0, S * sizeof(double));
    const drt_material *mat = p->surface_material;
    for (uint32_t i = 0; i < mat->num_bdsfs; i += 1)
    {
        bdsf_call(sc, mat->bdsfs[i], bdsf_result, p, in);
        for (uint32_t k = 0; k < S; k += 1) reflectance[k] = bdsf_result[k] + reflectance[k];
    }
}

/* Direction samplers, src/bdsf.c:188-292. They return the RECIPROCAL pdf (:190). */

static void uniform_sample_hemisphere(v3 *v, double *pdf, const point *p) /* :191-198 */
{
    *v = uniform_sample_sphere();
    m33 r = rotation_between(V(0.0, 0.0, 1.0), p->normal);
    *v = m_vmul(&r, *v);
    *pdf = REFMODE ? (double)(2.0 * PI_L) : 2.0 * PI_D;
}

static void cos_weighted_sample_hemisphere(v3 *v, double *pdf, const point *p) /* :200-213 */
{
    v3 q;
    for (;;)
    {
        q = uniform_sample_disc();
        if (v_dot(q, q) < 1.0) break;
    }
    q.z = sqrt(1.0 - v_dot(q, q));
    m33 r = rotation_between(V(0.0, 0.0, 1.0), p->normal);
    *v = m_vmul(&r, q);
    if (REFMODE) *pdf = (double)(PI_L / v_dot(p->normal, *v));
    else *pdf = PI_D / v_dot(p->normal, *v);
}

static void sample_specular_direction(v3 *v, double *pdf, const point *p) /* :215-220 */
{
    *v = v_reflect(v_reverse(p->out), p->normal);
    *pdf = 1.0;
}

static void sample_transmit_direction(const drt_scene *sc, v3 *v, double *pdf, const point *p) /* :222-234 */
{
    double ir = value_at_wl(sc, spd_of(sc, p->incident_material->refract_spd), p->trans_wl);
    double tr = value_at_wl(sc, spd_of(sc, p->transmit_material->refract_spd), p->trans_wl);
    *v = v_transmit(v_reverse(p->out), p->normal, ir, tr);
    *pdf = 1.0;
}

static void sample_reflect_or_transmit_direction(const drt_scene *sc, v3 *v, double *pdf, const point *p) /* :236-259 */
{
    double reflectance[MAX_S];
    const double *ir_spd = spd_of(sc, p->incident_material->refract_spd);
    const double *tr_spd = spd_of(sc, p->transmit_material->refract_spd);
    fs_dielectric_reflectance(reflectance, ir_spd, tr_spd, p->on_dot, sc->num_wavelengths);
    double rd = value_at_wl(sc, reflectance, p->trans_wl);
    double ir = value_at_wl(sc, ir_spd, p->trans_wl);
    double tr = value_at_wl(sc, tr_spd, p->trans_wl);
    double f = rng();
    v3 w = v_reverse(p->out);
    if (f < rd)
    {
        *v = v_reflect(w, p->normal);
        *pdf = 1.0 / rd;
    }
    else
    {
        *v = v_transmit(w, p->normal, ir, tr);
        *pdf = 1.0 / (1.0 - rd);
    }
}

static void sample_ct_direction(v3 *v, double *pdf, const point *p) /* :261-292 */
{
    do
    {
        double f = rng();
        double g = rng();
        double phi_mn = two_pi_times(g);
        double tan_mn = (p->surface_material->roughness * sqrt(f)) / sqrt(1.0 - f);
        double cos_mn = 1.0 / sqrt(1.0 + tan_mn * tan_mn);
        double sin_mn = sqrt(1.0 - cos_mn * cos_mn);
        double sp, cp;
        path_sincos(phi_mn, &sp, &cp);
        v3 micro_normal = V(sin_mn * cp, sin_mn * sp, cos_mn);
        m33 r = rotation_between(V(0.0, 0.0, 1.0), p->normal);
        micro_normal = m_vmul(&r, micro_normal);
        double sn_mn_dot = v_dot(p->normal, micro_normal);
        if (sn_mn_dot < 0.0)
        {
            micro_normal = v_reverse(micro_normal);
            sn_mn_dot = -sn_mn_dot;
        }
        double o_mn_dot = v_dot(p->out, micro_normal);
        v3 w = v_reverse(p->out);
        *v = v_reflect(w, micro_normal);
        double d = ggx(p->normal, micro_normal, p->surface_material->roughness) * sn_mn_dot;
        *pdf = ((4.0 * o_mn_dot) / d);
    } while (v_dot(*v, p->normal) < 0.0);
}

static void dirf_call(const drt_scene *sc, uint32_t id, v3 *v, double *pdf, const point *p)
{
    switch (id)
    {
        case DRT_DIRF_cos_weighted_sample_hemisphere: cos_weighted_sample_hemisphere(v, pdf, p); break;
        case DRT_DIRF_uniform_sample_hemisphere: uniform_sample_hemisphere(v, pdf, p); break;
        case DRT_DIRF_sample_specular_direction: sample_specular_direction(v, pdf, p); break;
        case DRT_DIRF_sample_transmit_direction: sample_transmit_direction(sc, v, pdf, p); break;
        case DRT_DIRF_sample_reflect_or_transmit_direction: sample_reflect_or_transmit_direction(sc, v, pdf, p); break;
        case DRT_DIRF_sample_ct_direction: sample_ct_direction(v, pdf, p); break;
        default: break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* src/daily_ray_trace.c:238-479                                                                */

static double surface_distance(const drt_surface *s, v3 o, v3 d)
{
    if (s->type == DRT_GEO_SPHERE) return line_sphere(o, d, from3(s->position), s->radius);
    return line_plane(o, d, from3(s->position), from3(s->normal), from3(s->u), from3(s->v));
}

/* points_mutually_visible, :238-270 */
static int points_mutually_visible(const drt_scene *sc, v3 p0, v3 p1)
{
    t_shadow_scans += 1;
    int visible = 1;
    v3 ray_direction = v_normalise(v_sub(p1, p0));
    v3 ray_origin = v_sum(p0, v_mul(ray_direction, VIS_FUDGE));
    double vis_dist = v_length(v_sub(p1, ray_origin)) - VIS_FUDGE;
    for (uint32_t i = 0; i < sc->num_surfaces; i += 1)
    {
        const drt_surface *s = &sc->surfaces[i];
        if (s->type != DRT_GEO_SPHERE && s->type != DRT_GEO_PLANE) continue; /* GEO_TYPE_POINT: continue */
        double dist = surface_distance(s, ray_origin, ray_direction);
        if (dist < vis_dist)
        {
            visible = 0;
            break;
        }
    }
    return visible;
}

/* find_ray_intersection, :334-403 */
static int find_ray_intersection(const drt_scene *sc, point *ip, v3 ray_origin, v3 ray_direction)
{
    t_closest_scans += 1;
    double min_dist = INFINITY;
    int index = -1;
    ray_origin = v_sum(ray_origin, v_mul(ray_direction, VIS_FUDGE));
    for (uint32_t i = 0; i < sc->num_surfaces; i += 1)
    {
        const drt_surface *s = &sc->surfaces[i];
        if (s->type != DRT_GEO_SPHERE && s->type != DRT_GEO_PLANE) continue;
        double dist = surface_distance(s, ray_origin, ray_direction);
        if (dist < min_dist)
        {
            min_dist = dist;
            index = (int)i;
        }
    }
    if (index >= 0)
    {
        const drt_surface *s = &sc->surfaces[index];
        const drt_material *smat = &sc->materials[s->material];
        ip->position = v_sum(ray_origin, v_mul(ray_direction, min_dist));
        if (s->type == DRT_GEO_SPHERE) ip->normal = v_normalise(v_sub(ip->position, from3(s->position)));
        else ip->normal = from3(s->normal);
        ip->trans_wl = 630.0;
        ip->out = v_reverse(ray_direction);
        ip->on_dot = v_dot(ip->normal, ip->out);
        ip->transmit_material = smat;
        ip->incident_material = &sc->materials[sc->base_material];
        if (ip->on_dot < 0.0)
        {
            if (s->type != DRT_GEO_PLANE)
            {
                ip->transmit_material = &sc->materials[sc->base_material];
                ip->incident_material = smat;
            }
            ip->normal = v_reverse(ip->normal);
            ip->on_dot = v_dot(ip->normal, ip->out);
        }
        ip->surface_material = smat;
        ip->surface_index = index;
    }
    else
    {
        ip->surface_material = &sc->materials[sc->escape_material];
        ip->surface_index = -1;
    }
    return index;
}

/* direct_light_contribution, :272-332. RNG draws happen before the visibility test;
 * a later light multiplies everything accumulated so far (Q4); weight = atten * area (Q3). */
static void direct_light_contribution(const drt_scene *sc, double *contribution, const point *ip)
{
    t_shaded += 1;
    uint32_t S = sc->num_wavelengths;
    double reflectance[MAX_S];
    memset(reflectance, 0, S * sizeof(double));
    memset(contribution, 0, S * sizeof(double));
    for (uint32_t i = 0; i < sc->num_surfaces; i += 1)
    {
        const drt_surface *ls = &sc->surfaces[i];
        const drt_material *lm = &sc->materials[ls->material];
        if (!lm->is_emissive) continue;
        double light_pdf = 0.0; /* uninitialised in the reference for other types; never reached */
        double attenuation_factor = 1.0;
        v3 light_position = V(0.0, 0.0, 0.0);
        switch (ls->type)
        {
            case DRT_GEO_POINT:
            {
                light_position = from3(ls->position);
                double dist = v_length(v_sub(light_position, ip->position));
                light_pdf = 1.0;
                if (REFMODE) attenuation_factor = (double)(4.0 * PI_L * dist * dist);
                else attenuation_factor = ((4.0 * PI_D) * dist) * dist;
                break;
            }
            case DRT_GEO_SPHERE:
            {
                double u = rng();
                double v = rng();
                double r = sqrt(1.0 - u * u);
                double t = two_pi_times(v);
                double st, ct;
                path_sincos(t, &st, &ct);
                v3 sphere_point = V(r * ct, r * st, u);
                light_position = v_sum(from3(ls->position), v_mul(sphere_point, ls->radius));
                if (REFMODE) light_pdf = (double)(4.0 * PI_L * ls->radius * ls->radius);
                else light_pdf = ((4.0 * PI_D) * ls->radius) * ls->radius;
                break;
            }
            case DRT_GEO_PLANE:
            {
                double u = rng();
                double v = rng();
                v3 u_pos = v_mul(from3(ls->u), u);
                v3 v_pos = v_mul(from3(ls->v), v);
                light_position = v_sum(v_sum(from3(ls->position), u_pos), v_pos);
                light_pdf = v_length(v_cross(from3(ls->u), from3(ls->v)));
                break;
            }
            default: break;
        }
        if (points_mutually_visible(sc, ip->position, light_position))
        {
            v3 incoming = v_normalise(v_sub(light_position, ip->position));
            bdsf(sc, reflectance, ip, incoming);
            const double *em = spd_of(sc, lm->emission_spd);
            double c = attenuation_factor * (light_pdf);
            for (uint32_t k = 0; k < S; k += 1) contribution[k] = contribution[k] + reflectance[k];
            for (uint32_t k = 0; k < S; k += 1) contribution[k] = contribution[k] * em[k];
            for (uint32_t k = 0; k < S; k += 1) contribution[k] = contribution[k] * c;
        }
    }
}

/* cast_ray, :432-479 */
static int cast_ray(const drt_scene *sc, double *dst, v3 ray_origin, v3 ray_direction, uint32_t max_depth, int32_t *hit_seq)
{
    uint32_t S = sc->num_wavelengths;
    double contribution[MAX_S], throughput[MAX_S], reflectance[MAX_S];
    for (uint32_t k = 0; k < S; k += 1) throughput[k] = 1.0;
    v3 in;
    point ip;
    memset(&ip, 0, sizeof(ip));
    int scans = 0;
    for (uint32_t depth = 0; depth < max_depth; depth += 1)
    {
        int idx = find_ray_intersection(sc, &ip, ray_origin, ray_direction);
        if (hit_seq) hit_seq[depth] = idx;
        scans += 1;
        const drt_material *mat = ip.surface_material;
        if (mat->is_black_body && !mat->is_emissive) break;
        else if (mat->is_black_body && mat->is_emissive)
        {
            const double *em = spd_of(sc, mat->emission_spd);
            for (uint32_t k = 0; k < S; k += 1) dst[k] = dst[k] + throughput[k] * em[k];
            break;
        }
        else
        {
            direct_light_contribution(sc, contribution, &ip);
            for (uint32_t k = 0; k < S; k += 1) dst[k] = dst[k] + throughput[k] * contribution[k];
            double dir_pdf = 0.0;
            dirf_call(sc, mat->dir_func, &in, &dir_pdf, &ip);
            bdsf(sc, reflectance, &ip, in);
            for (uint32_t k = 0; k < S; k += 1) reflectance[k] = reflectance[k] * dir_pdf;
            for (uint32_t k = 0; k < S; k += 1) throughput[k] = throughput[k] * reflectance[k];
            ray_direction = in;
            ray_origin = ip.position;
        }
    }
    return scans;
}

/* sample_pixel_point :550-569 and sample_scene :571-618 */
static int sample_scene(const drt_scene *sc, const drt_camera *cam, double *contribution, double *filter,
                        uint32_t x, uint32_t y, uint32_t max_depth, uint32_t scheme, int32_t *hit_seq)
{
    uint32_t S = sc->num_wavelengths;
    memset(contribution, 0, S * sizeof(double));
    double px = 0.0, py = 0.0;
    if (scheme == DRT_FILM_SAMPLE_CENTER) { px = 0.5; py = 0.5; }
    else if (scheme == DRT_FILM_SAMPLE_RANDOM) { px = rng(); py = rng(); }
    double film_x = ((double)x + px) * cam->pixel_width;
    double film_y = ((double)y + py) * cam->pixel_height;
    v3 forward = from3(cam->forward);
    v3 bottom = v_mul(from3(cam->up), film_y);
    v3 left = v_mul(from3(cam->right), film_x);
    v3 bl = v_sum(left, bottom);
    v3 pixel_point = v_sum(bl, from3(cam->film_bottom_left));
    v3 ray_origin, ray_direction;
    v3 aperture_position = from3(cam->aperture_position);
    if (cam->aperture_radius > 0.0)
    {
        v3 focus_dir = v_normalise(v_sub(aperture_position, pixel_point));
        focus_dir = v_mul(focus_dir, cam->focal_depth / v_dot(focus_dir, forward));
        v3 focus_point = v_sum(pixel_point, focus_dir);
        m33 r = rotation_between(V(0.0, 0.0, 1.0), forward);
        v3 disc_point = v_mul(uniform_sample_disc(), cam->aperture_radius);
        v3 lens_point = m_vmul(&r, disc_point);
        ray_origin = v_sum(aperture_position, lens_point);
        ray_direction = v_normalise(v_sub(focus_point, ray_origin));
    }
    else
    {
        ray_origin = pixel_point;
        ray_direction = v_normalise(v_sub(aperture_position, ray_origin));
    }
    if (hit_seq) for (uint32_t d = 0; d < max_depth; d += 1) hit_seq[d] = -2;
    int scans = cast_ray(sc, contribution, ray_origin, ray_direction, max_depth, hit_seq);
    double pixel_filter_value = 1.0;
    double vignette_factor = v_dot(ray_direction, forward);
    double m = vignette_factor * pixel_filter_value;
    for (uint32_t k = 0; k < S; k += 1) contribution[k] = contribution[k] * m;
    *filter = pixel_filter_value;
    return scans;
}

int drt_oracle_sample_scene(const drt_scene *sc, const drt_camera *cam, const drt_params *p,
                            uint32_t x, uint32_t y, uint32_t sample, double *contribution, double *filter, int32_t *hit_seq)
{
    drt_oracle_seed_path(drt_oracle_path_key(p->seed, p->width, p->height, x, y, sample));
    return sample_scene(sc, cam, contribution, filter, x, y, p->max_depth, p->pixel_scheme, hit_seq);
}

/* The pixel loop of render_image, :710-745, over one tile. */
typedef struct
{
    const drt_scene *sc;
    const drt_camera *cam;
    const drt_params *p;
    double *pixels, *avgs, *vars;
    int32_t *hits;
    uint32_t row_begin, row_end;
    drt_stats stats;
} tile_job;

static void *render_rows(void *arg)
{
    tile_job *job = (tile_job *)arg;
    const drt_params *p = job->p;
    uint32_t S = job->sc->num_wavelengths;
    double contribution[MAX_S + 1];
    double tmp0[MAX_S], tmp1[MAX_S];
    t_rng_draws = t_closest_scans = t_shadow_scans = t_shaded = 0;
    uint64_t paths = 0;
    uint32_t stride = p->row_stride ? p->row_stride : 1;
    for (uint32_t s = 0; s < p->spp; s += 1)
    {
        uint32_t sample = p->first_sample + s;
        for (uint32_t j = job->row_begin; j < job->row_end; j += 1)
        {
            uint32_t y = p->y0 + j * stride;
            for (uint32_t i = 0; i < p->tile_w; i += 1)
            {
                uint32_t x = p->x0 + i;
                uint64_t pixel_offset = (uint64_t)j * p->tile_w + i;
                double *dst_pixel = job->pixels + (uint64_t)(S + 1) * pixel_offset;
                double *dst_avg = job->avgs ? job->avgs + (uint64_t)S * pixel_offset : NULL;
                double *dst_var = job->vars ? job->vars + (uint64_t)S * pixel_offset : NULL;
                int32_t *hit_seq = job->hits
                    ? job->hits + (((uint64_t)s * p->tile_h + j) * p->tile_w + i) * p->max_depth : NULL;
                double filter = 0.0;
                drt_oracle_seed_path(drt_oracle_path_key(p->seed, p->width, p->height, x, y, sample));
                sample_scene(job->sc, job->cam, contribution, &filter, x, y, p->max_depth, p->pixel_scheme, hit_seq);
                paths += 1;
                for (uint32_t k = 0; k < S; k += 1) dst_pixel[k] = dst_pixel[k] + contribution[k]; /* :732 */
                dst_pixel[S] += filter;                                                            /* :733 */
                if (dst_avg && dst_var)                                                             /* :736-743 */
                {
                    for (uint32_t k = 0; k < S; k += 1) tmp0[k] = contribution[k] - dst_avg[k];
                    for (uint32_t k = 0; k < S; k += 1) tmp1[k] = tmp0[k];
                    for (uint32_t k = 0; k < S; k += 1) tmp0[k] = tmp0[k] / (double)(sample + 1);
                    for (uint32_t k = 0; k < S; k += 1) dst_avg[k] = dst_avg[k] + tmp0[k];
                    for (uint32_t k = 0; k < S; k += 1) tmp0[k] = contribution[k] - dst_avg[k];
                    for (uint32_t k = 0; k < S; k += 1) tmp0[k] = tmp1[k] * tmp0[k];
                    for (uint32_t k = 0; k < S; k += 1) dst_var[k] = dst_var[k] + tmp0[k];
                }
            }
        }
    }
    job->stats.paths = paths;
    job->stats.closest_hit_scans = t_closest_scans;
    job->stats.shaded_vertices = t_shaded;
    job->stats.shadow_scans = t_shadow_scans;
    job->stats.rng_draws = t_rng_draws;
    return NULL;
}

int drt_oracle_render_tile(const drt_scene *sc, const drt_camera *cam, const drt_params *p,
                           double *dst_pixels, double *dst_avgs, double *dst_vars,
                           int32_t *hit_indices, drt_stats *stats, int num_threads)
{
    if (!sc || !cam || !p || !dst_pixels) return -1;
    if (sc->num_wavelengths > MAX_S) return -2;
    if (num_threads < 1) num_threads = 1;
    if ((uint32_t)num_threads > p->tile_h) num_threads = (int)(p->tile_h ? p->tile_h : 1);
    tile_job *jobs = (tile_job *)calloc((size_t)num_threads, sizeof(tile_job));
    pthread_t *threads = (pthread_t *)calloc((size_t)num_threads, sizeof(pthread_t));
    for (int t = 0; t < num_threads; t += 1)
    {
        jobs[t].sc = sc; jobs[t].cam = cam; jobs[t].p = p;
        jobs[t].pixels = dst_pixels; jobs[t].avgs = dst_avgs; jobs[t].vars = dst_vars; jobs[t].hits = hit_indices;
        jobs[t].row_begin = (uint32_t)(((uint64_t)p->tile_h * (uint64_t)t) / (uint64_t)num_threads);
        jobs[t].row_end = (uint32_t)(((uint64_t)p->tile_h * (uint64_t)(t + 1)) / (uint64_t)num_threads);
    }
    if (num_threads == 1) render_rows(&jobs[0]);
    else
    {
        for (int t = 0; t < num_threads; t += 1) pthread_create(&threads[t], NULL, render_rows, &jobs[t]);
        for (int t = 0; t < num_threads; t += 1) pthread_join(threads[t], NULL);
    }
    if (stats)
    {
        memset(stats, 0, sizeof(*stats));
        for (int t = 0; t < num_threads; t += 1)
        {
            stats->paths += jobs[t].stats.paths;
            stats->closest_hit_scans += jobs[t].stats.closest_hit_scans;
            stats->shaded_vertices += jobs[t].stats.shaded_vertices;
            stats->shadow_scans += jobs[t].stats.shadow_scans;
            stats->rng_draws += jobs[t].stats.rng_draws;
        }
    }
    free(threads);
    free(jobs);
    return 0;
}

/* spectrum_to_xyz, src/spectrum.c:49-70 */
void drt_oracle_spectrum_to_xyz(const drt_scene *sc, const double *spd, double xyz[3])
{
    uint32_t S = sc->num_wavelengths;
    const double *rw = spd_of(sc, (int32_t)sc->cmf_rw), *cx = spd_of(sc, (int32_t)sc->cmf_x);
    const double *cy = spd_of(sc, (int32_t)sc->cmf_y), *cz = spd_of(sc, (int32_t)sc->cmf_z);
    double X = 0.0, Y = 0.0, Z = 0.0, n = 0.0;
    for (uint32_t i = 0; i < S; i += 1) n += (cy[i] * rw[i]);
    n *= sc->wavelength_interval;
    for (uint32_t i = 0; i < S; i += 1)
    {
        X += (cx[i] * spd[i] * rw[i]);
        Y += (cy[i] * spd[i] * rw[i]);
        Z += (cz[i] * spd[i] * rw[i]);
    }
    xyz[0] = X * (sc->wavelength_interval / n);
    xyz[1] = Y * (sc->wavelength_interval / n);
    xyz[2] = Z * (sc->wavelength_interval / n);
}

/* src/daily_ray_trace.c:15-23: divide by the filter sum, then spectrum_to_xyz */
void drt_oracle_film_to_xyz(const drt_scene *sc, const double *pixels, uint64_t n_pixels, double *xyz)
{
    uint32_t S = sc->num_wavelengths;
    double tmp[MAX_S];
    for (uint64_t p = 0; p < n_pixels; p += 1)
    {
        const double *px = pixels + p * (S + 1);
        double f = px[S];
        for (uint32_t k = 0; k < S; k += 1) tmp[k] = px[k] / f;
        drt_oracle_spectrum_to_xyz(sc, tmp, xyz + 3 * p);
    }
}

/* ---- unit-level wrappers ------------------------------------------------------------------- */

double drt_oracle_line_sphere(const double o[3], const double d[3], const double c[3], double r)
{
    return line_sphere(from3(o), from3(d), from3(c), r);
}
double drt_oracle_line_plane(const double o[3], const double d[3], const double p[3], const double n[3], const double u[3], const double v[3])
{
    return line_plane(from3(o), from3(d), from3(p), from3(n), from3(u), from3(v));
}
void drt_oracle_reflect(const double v[3], const double n[3], double out[3]) { to3(v_reflect(from3(v), from3(n)), out); }
void drt_oracle_transmit(const double v[3], const double n[3], double ir, double tr, double out[3])
{
    to3(v_transmit(from3(v), from3(n), ir, tr), out);
}
static void m_out(const m33 *m, double o[9])
{
    for (int c = 0; c < 3; c += 1) { o[3 * c] = m->c[c].x; o[3 * c + 1] = m->c[c].y; o[3 * c + 2] = m->c[c].z; }
}
void drt_oracle_rotation_between(const double v[3], const double w[3], double m_cols[9])
{
    m33 m = rotation_between(from3(v), from3(w));
    m_out(&m, m_cols);
}
void drt_oracle_rotation_about_axis(const double axis[3], double angle, double m_cols[9])
{
    m33 m = rotation_about_axis(from3(axis), angle);
    m_out(&m, m_cols);
}
void drt_oracle_bdsf_func(const drt_scene *sc, uint32_t id, const drt_oracle_point *ap, const double incoming[3], double *result)
{
    point p;
    point_from_api(sc, ap, &p);
    bdsf_call(sc, id, result, &p, from3(incoming));
}
void drt_oracle_bdsf(const drt_scene *sc, const drt_oracle_point *ap, const double incoming[3], double *reflectance)
{
    point p;
    point_from_api(sc, ap, &p);
    bdsf(sc, reflectance, &p, from3(incoming));
}
void drt_oracle_dir_func(const drt_scene *sc, uint32_t id, const drt_oracle_point *ap, double dir[3], double *recip_pdf)
{
    point p;
    point_from_api(sc, ap, &p);
    v3 v = {0.0, 0.0, 0.0};
    double pdf = 0.0;
    dirf_call(sc, id, &v, &pdf, &p);
    to3(v, dir);
    *recip_pdf = pdf;
}
double drt_oracle_ggx(const double sn[3], const double mn[3], double r) { return ggx(from3(sn), from3(mn), r); }
double drt_oracle_ggx_att(const double v[3], const double sn[3], const double mn[3], double r)
{
    return ggx_att(from3(v), from3(sn), from3(mn), r);
}
void drt_oracle_fs_dielectric_reflectance(const double *ir, const double *tr, double inc_cos, uint32_t n, double *out)
{
    fs_dielectric_reflectance(out, ir, tr, inc_cos, n);
}
void drt_oracle_fs_conductor_reflectance(const double *ir, const double *tr, const double *te, double inc_cos, uint32_t n, double *out)
{
    fs_conductor_reflectance(out, ir, tr, te, inc_cos, n);
}
static void point_to_api(const drt_scene *sc, const point *p, drt_oracle_point *a)
{
    to3(p->position, a->position);
    to3(p->normal, a->normal);
    to3(p->out, a->out);
    a->on_dot = p->on_dot;
    a->trans_wl = p->trans_wl;
    a->surface_material = (uint32_t)(p->surface_material - sc->materials);
    a->incident_material = p->incident_material ? (uint32_t)(p->incident_material - sc->materials) : 0;
    a->transmit_material = p->transmit_material ? (uint32_t)(p->transmit_material - sc->materials) : 0;
}
int drt_oracle_find_ray_intersection(const drt_scene *sc, const double o[3], const double d[3], drt_oracle_point *ap)
{
    point p;
    memset(&p, 0, sizeof(p));
    int idx = find_ray_intersection(sc, &p, from3(o), from3(d));
    if (ap) point_to_api(sc, &p, ap);
    return idx;
}
int drt_oracle_points_mutually_visible(const drt_scene *sc, const double p0[3], const double p1[3])
{
    return points_mutually_visible(sc, from3(p0), from3(p1));
}
void drt_oracle_direct_light(const drt_scene *sc, const drt_oracle_point *ap, double *contribution)
{
    point p;
    point_from_api(sc, ap, &p);
    direct_light_contribution(sc, contribution, &p);
}
