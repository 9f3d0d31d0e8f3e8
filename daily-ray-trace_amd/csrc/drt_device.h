/*
 * drt_device.h -- device-side math of the render path (gfx950 HIP, IEEE f64, no FMA contraction).
 *
 * The path is split in two along the one axis the reference's arithmetic allows: everything that
 * decides WHERE a path goes (intersections, light samples, sampled directions, RNG) depends on a
 * handful of per-path scalars, never on the 69-wide spectra; everything spectral is elementwise
 * per wavelength with those scalars as coefficients. So
 *   - the trace kernel (one ray per lane) runs geometry and writes a compact vertex record,
 *   - the shade kernel (one wavelength per lane, one pixel per wave) replays the records over
 *     the wavelengths and accumulates the film,
 * and each spectral value still goes through exactly the f64 operations, in the order, that the
 * reference applies to it (src/daily_ray_trace.c:215-479, src/bdsf.c, src/spectrum.c:189-243).
 *
 * Arithmetic contract ("DEVICE mode" of oracle/drt_oracle.c): expressions the reference
 * evaluates in x87 long double because PI is an L literal are evaluated here in f64 in the same
 * association order; sin/cos use the range-reduced kernel below (same spec as the oracle);
 * sqrt and / are the correctly rounded IEEE operations.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define DRT_PI 3.14159265358979323846
#define DRT_VIS_FUDGE 0.0001 /* src/daily_ray_trace.c:237 */
#define DRT_INF __builtin_huge_val()

struct V3
{
    double x, y, z;
};
struct M33
{
    V3 c[3]; /* columns */
};

__device__ __forceinline__ V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ bool v_equal(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
__device__ __forceinline__ V3 v_sum(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 v_sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ double v_dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 v_cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 v_mul(V3 v, double f) { return v3(f * v.x, f * v.y, f * v.z); }
__device__ __forceinline__ V3 v_div(V3 v, double f) { return v3(v.x / f, v.y / f, v.z / f); }
__device__ __forceinline__ double v_length(V3 v) { return __builtin_sqrt(v_dot(v, v)); }
__device__ __forceinline__ V3 v_normalise(V3 v) { return v_div(v, v_length(v)); }
__device__ __forceinline__ V3 v_reverse(V3 v) { return v3(-v.x, -v.y, -v.z); }

/* vec3_reflect, src/geometry.c:85-90 */
__device__ __forceinline__ V3 v_reflect(V3 v, V3 n)
{
    double f = 2.0 * v_dot(v, n);
    return v_sub(v, v_mul(n, f));
}

/* vec3_transmit, src/geometry.c:92-106 (NaN on total internal reflection, as in the reference) */
__device__ __forceinline__ V3 v_transmit(V3 v, V3 n, double ir, double tr)
{
    double vn_dot = v_dot(v, n);
    double rel_ref = ir / tr;
    V3 m = v_mul(n, vn_dot);
    v = v_sub(m, v);
    V3 perpend = v_reverse(v_mul(v, rel_ref));
    double perpend_dot = -__builtin_sqrt(1.0 - v_dot(perpend, perpend));
    V3 parallel = v_mul(n, perpend_dot);
    return v_sum(perpend, parallel);
}

__device__ __forceinline__ V3 m_vmul(const M33 &m, V3 v)
{
    V3 r0 = v3(m.c[0].x, m.c[1].x, m.c[2].x);
    V3 r1 = v3(m.c[0].y, m.c[1].y, m.c[2].y);
    V3 r2 = v3(m.c[0].z, m.c[1].z, m.c[2].z);
    return v3(v_dot(r0, v), v_dot(r1, v), v_dot(r2, v));
}

/* find_rotation_between_vectors, src/geometry.c:263-295 with v = (0,0,1) folded in is NOT done:
 * the general Rodrigues form is kept so every product and sum matches the reference's. */
__device__ __forceinline__ M33 rotation_between(V3 v, V3 w)
{
    V3 n = v_cross(v, w);
    double c = v_dot(v, w);
    M33 r;
    if (v_dot(n, n) == 0.0 && c <= 0.0)
    {
        r.c[0] = v3(-1.0, 0.0, 0.0);
        r.c[1] = v3(0.0, -1.0, 0.0);
        r.c[2] = v3(0.0, 0.0, -1.0);
        return r;
    }
    M33 m;
    m.c[0] = v3(0.0, n.z, -n.y);
    m.c[1] = v3(-n.z, 0.0, n.x);
    m.c[2] = v3(n.y, -n.x, 0.0);
    V3 row0 = v3(m.c[0].x, m.c[1].x, m.c[2].x);
    V3 row1 = v3(m.c[0].y, m.c[1].y, m.c[2].y);
    V3 row2 = v3(m.c[0].z, m.c[1].z, m.c[2].z);
    double f = 1.0 / (1.0 + c);
    /* mat3x3_mul stores row(m,i).col(m,j) at columns[i].xyz[j] (src/geometry.c:240-252) */
    V3 mm0 = v_mul(v3(v_dot(row0, m.c[0]), v_dot(row0, m.c[1]), v_dot(row0, m.c[2])), f);
    V3 mm1 = v_mul(v3(v_dot(row1, m.c[0]), v_dot(row1, m.c[1]), v_dot(row1, m.c[2])), f);
    V3 mm2 = v_mul(v3(v_dot(row2, m.c[0]), v_dot(row2, m.c[1]), v_dot(row2, m.c[2])), f);
    r.c[0] = v_sum(v_sum(v3(1.0, 0.0, 0.0), m.c[0]), mm0);
    r.c[1] = v_sum(v_sum(v3(0.0, 1.0, 0.0), m.c[1]), mm1);
    r.c[2] = v_sum(v_sum(v3(0.0, 0.0, 1.0), m.c[2]), mm2);
    return r;
}

/* ---- RNG: per-path xorshift64, seeded through splitmix64 ---------------------- */
__device__ __forceinline__ uint64_t drt_splitmix64(uint64_t k)
{
    uint64_t z = k + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z ? z : 1ull;
}
/* rng(): rand()/RAND_MAX with rand() := state>>33, RAND_MAX := 2^31-1; range [0,1] inclusive */
__device__ __forceinline__ double drt_rng(uint64_t &state, uint32_t &draws)
{
    uint64_t x = state;
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    state = x;
    draws += 1;
    return (double)(uint32_t)(x >> 33) / 2147483647.0;
}

/* ---- sincos: the specification the oracle's sincos restates (see DESIGN.md "Arithmetic contract") ---- */
__device__ __forceinline__ void drt_sincos(double t, double &s, double &c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632673412561417e+00, PIO2_1T = 6.07710050650619224932e-11;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double fn = __builtin_rint(t * TWO_OVER_PI);
    double r = t - fn * PIO2_1;
    double w = fn * PIO2_1T;
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double z = y0 * y0;
    double v = z * y0;
    double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double hz = 0.5 * z;
    double w1 = 1.0 - hz;
    double kc = w1 + (((1.0 - w1) - hz) + (z * rc - y0 * y1));
    int n = (int)fn & 3;
    s = (n == 0) ? ks : (n == 1) ? kc : (n == 2) ? -ks : -kc;
    c = (n == 0) ? kc : (n == 1) ? -ks : (n == 2) ? -kc : ks;
}

/* uniform_sample_sphere, src/rng.c:14-23 */
__device__ __forceinline__ V3 uniform_sample_sphere(uint64_t &rs, uint32_t &draws)
{
    double u = drt_rng(rs, draws);
    double v = drt_rng(rs, draws);
    double r = __builtin_sqrt(1.0 - u * u);
    double t = (2.0 * DRT_PI) * v;
    double st, ct;
    drt_sincos(t, st, ct);
    return v3(r * ct, r * st, u);
}

/* uniform_sample_disc, src/rng.c:25-51 */
__device__ __forceinline__ V3 uniform_sample_disc(uint64_t &rs, uint32_t &draws)
{
    double r_x = drt_rng(rs, draws);
    double r_y = drt_rng(rs, draws);
    double o_x = 2.0 * r_x - 1.0;
    double o_y = 2.0 * r_y - 1.0;
    if (o_x == 0.0 && o_y == 0.0) return v3(0.0, 0.0, 0.0);
    double r, t;
    if (__builtin_fabs(o_x) > __builtin_fabs(o_y))
    {
        r = o_x;
        t = (DRT_PI / 4.0) * (o_y / o_x);
    }
    else
    {
        r = o_y;
        t = (DRT_PI / 2.0) - (DRT_PI / 4.0) * (o_x / o_y);
    }
    double st, ct;
    drt_sincos(t, st, ct);
    return v3(r * ct, r * st, 0.0);
}

/* ---- intersectors ----------------------------------------------------------------------------- */

/* line_sphere_intersection, src/geometry.c:123-146 (a = 1: 4*a*c == 4*c and /(2a) == *0.5 exactly) */
__device__ __forceinline__ double line_sphere(V3 o, V3 d, V3 sc, double sr)
{
    V3 c_to_o = v_sub(o, sc);
    double b = -2.0 * v_dot(c_to_o, d);
    double c = v_dot(c_to_o, c_to_o) - sr * sr;
    double disc = b * b - 4.0 * c;
    if (disc < 0.0) return DRT_INF;
    double sq = __builtin_sqrt(disc);
    double s0 = (b + sq) * 0.5;
    double s1 = (b - sq) * 0.5;
    if (s0 < 0.0 && s1 < 0.0) return DRT_INF;
    else if (s0 >= 0.0 && s1 < 0.0) return s0;
    else if (s1 >= 0.0 && s0 < 0.0) return s1;
    else if (s0 <= s1) return s0;
    else return s1;
}

/* line_plane_intersection, src/geometry.c:157-182. |u|, |v|, u/|u|, v/|v| are per-plane constants
 * the reference recomputes per ray; they arrive precomputed (same IEEE sqrt and divisions). */
__device__ __forceinline__ double line_plane(V3 o, V3 d, V3 pp, V3 pn, V3 un, V3 vn, double ul, double vl)
{
    double dn = v_dot(d, pn);
    if (dn == 0.0) return DRT_INF;
    V3 o_to_p = v_sub(pp, o);
    double l = v_dot(o_to_p, pn) / dn;
    V3 i = v_sum(o, v_mul(d, l));
    V3 j = v_sub(i, pp);
    double ju = v_dot(j, un);
    double jv = v_dot(j, vn);
    if (l >= 0.0 && 0.0 <= ju && ju <= ul && 0.0 <= jv && jv <= vl) return l;
    return DRT_INF;
}

/* ---- pow(x, y) as bp_glossy_bdsf uses it (src/bdsf.c:116): x in [0, 1], y the material's shininess ------------------------ */
/* For an integer shininess (what scenes carry: 100, 32, ...) by repeated squaring in double-double arithmetic: the value carried is
 * hi + lo with |lo| <= ulp(hi) / 2, every product is exact to 2^-104, so after the <= 20 products of an exponent below 1024 the
 * result rounds to the double nearest x^y except within ~2^-98 of a rounding boundary -- at least as close to libm's pow (which the
 * reference calls and which is not correctly rounded either) as the general-purpose pow it replaces here (within 2 ulp of glibc's,
 * tests/test_gpu_parity.py), at about half the instructions. Any other exponent goes to pow(). */
__device__ __forceinline__ double drt_pow_shininess(double x, double y)
{
    const uint32_t n = (uint32_t)y;
    if (!(y >= 0.0 && y < 1024.0 && (double)n == y) || !(x >= 0.0 && x <= 1.0)) return pow(x, y);
    double rh = 1.0, rl = 0.0; /* the result so far */
    double bh = x, bl = 0.0;   /* x^(2^k) */
    for (uint32_t k = n; __any(k != 0u); k >>= 1)
    {
        if (k & 1u)
        {
            /* (rh + rl) * (bh + bl) */
            const double p = rh * bh;
            double e = __builtin_fma(rh, bh, -p);
            e = __builtin_fma(rh, bl, e);
            e = __builtin_fma(rl, bh, e);
            rh = p + e;
            rl = e - (rh - p);
        }
        if (k > 1u)
        {
            /* (bh + bl)^2 */
            const double p = bh * bh;
            double e = __builtin_fma(bh, bh, -p);
            e = __builtin_fma(bh + bh, bl, e);
            bh = p + e;
            bl = e - (bh - p);
        }
    }
    return rh + rl;
}

/* ---- Fresnel terms, one wavelength (the loop bodies of src/bdsf.c:44-101) --------------------- */

/* fs_dielectric_reflectance body; ts_cos squares the already squared sine (quirk Q2) */
__device__ __forceinline__ double dielectric_reflectance(double ir, double tr, double inc_cos, double inc_sin_sq)
{
    double rel = ir / tr;
    double ts_sin_sq = rel * rel * inc_sin_sq;
    if (ts_sin_sq >= 1.0) return 1.0;
    double ts_cos = __builtin_sqrt(1.0 - ts_sin_sq * ts_sin_sq);
    double tr_on = tr * inc_cos;
    double tr_ts = tr * ts_cos;
    double ir_on = ir * inc_cos;
    double ir_ts = ir * ts_cos;
    double par = (tr_on - ir_ts) / (tr_on + ir_ts);
    double per = (ir_on - tr_ts) / (ir_on + tr_ts);
    par *= par;
    per *= per;
    return 0.5 * (par + per);
}

/* fs_conductor_reflectance body */
__device__ __forceinline__ double conductor_reflectance(double ir, double tr, double te, double inc_cos, double inc_cos_sq,
                                                        double inc_sin_sq)
{
    double rr = tr / ir;
    double re = te / ir;
    double rr_sq = rr * rr;
    double re_sq = re * re;
    double r = rr_sq - re_sq - inc_sin_sq;
    double apb_sq = __builtin_sqrt(r * r + 4.0 * rr_sq * re_sq);
    double a = __builtin_sqrt(0.5 * (apb_sq + r));
    double s = apb_sq + inc_cos_sq;
    double t = 2.0 * a * inc_cos;
    double u = inc_cos_sq * apb_sq + inc_sin_sq * inc_sin_sq;
    double v = t * inc_sin_sq;
    double par = (s - t) / (s + t);
    double per = par * (u - v) / (u + v);
    return 0.5 * (par + per);
}

/* The same two terms with what depends on the PAIR of media only taken out of the loop over vertices: for a vertex whose
 * incident / transmit materials are the scene's base material and the surface's own (every vertex but those met from inside a
 * second object), the launcher tabulates per wavelength
 *     rel_sq = (ir / tr) * (ir / tr)                              fs_dielectric_reflectance, src/bdsf.c:52-56
 *     cA = (tr/ir)^2 - (te/ir)^2,  cB = (4 (tr/ir)^2) (te/ir)^2   fs_conductor_reflectance, src/bdsf.c:84-91
 * with the reference's own operations in the reference's order (IEEE division and multiplication give the same bits on the
 * host as here), so with `paired` these return bit for bit what the functions above return: one division less per dielectric
 * term, two less per conductor term. The pair's rows arrive in the registers of the inputs they replace: a dielectric's as
 * (ir, tr, rel_sq in te's place), a conductor's as (cA in ir's place, cB in tr's). */
__device__ __forceinline__ double dielectric_reflectance_sel(bool paired, double ir, double tr, double te_or_rel_sq, double inc_cos, double inc_sin_sq)
{
    double rel_sq = te_or_rel_sq;
    if (!paired)
    {
        double rel = ir / tr;
        rel_sq = rel * rel;
    }
    double ts_sin_sq = rel_sq * inc_sin_sq;
    if (ts_sin_sq >= 1.0) return 1.0;
    double ts_cos = __builtin_sqrt(1.0 - ts_sin_sq * ts_sin_sq);
    double tr_on = tr * inc_cos;
    double tr_ts = tr * ts_cos;
    double ir_on = ir * inc_cos;
    double ir_ts = ir * ts_cos;
    double par = (tr_on - ir_ts) / (tr_on + ir_ts);
    double per = (ir_on - tr_ts) / (ir_on + tr_ts);
    par *= par;
    per *= per;
    return 0.5 * (par + per);
}
__device__ __forceinline__ double conductor_reflectance_sel(bool paired, double ir_or_cA, double tr_or_cB, double te, double inc_cos, double inc_cos_sq,
                                                            double inc_sin_sq)
{
    double cA = ir_or_cA, cB = tr_or_cB;
    if (!paired)
    {
        double rr = tr_or_cB / ir_or_cA;
        double re = te / ir_or_cA;
        double rr_sq = rr * rr;
        double re_sq = re * re;
        cA = rr_sq - re_sq;
        cB = 4.0 * rr_sq * re_sq;
    }
    double r = cA - inc_sin_sq;
    double apb_sq = __builtin_sqrt(r * r + cB);
    double a = __builtin_sqrt(0.5 * (apb_sq + r));
    double s = apb_sq + inc_cos_sq;
    double t = 2.0 * a * inc_cos;
    double u = inc_cos_sq * apb_sq + inc_sin_sq * inc_sin_sq;
    double v = t * inc_sin_sq;
    double par = (s - t) / (s + t);
    double per = par * (u - v) / (u + v);
    return 0.5 * (par + per);
}

/* ggx / ggx_att, src/bdsf.c:3-42 */
__device__ __forceinline__ double ggx(V3 sn, V3 mn, double r)
{
    double d = v_dot(sn, mn);
    double r_2 = r * r;
    if (d <= 0.0) return 0.0;
    double d_2 = d * d;
    double d_4 = d_2 * d_2;
    double tan_sq = (1.0 / d_2) - 1.0;
    return r_2 / (((DRT_PI * d_4) * (r_2 + tan_sq)) * (r_2 + tan_sq));
}
__device__ __forceinline__ double ggx_att(V3 v, V3 sn, V3 mn, double r)
{
    double att;
    double g = ggx(sn, mn, r);
    double v_mn = v_dot(v, mn);
    double v_sn = v_dot(v, sn);
    double dot_quot = __builtin_fabs(v_mn / v_sn);
    double r_2 = r * r;
    if (dot_quot <= 0.0) att = 0.0;
    else
    {
        double vn_tan_sq = (1.0 / (v_sn * v_sn)) - 1.0;
        att = 2.0 / (1.0 + __builtin_sqrt(1.0 + r_2 * vn_tan_sq));
    }
    return g * att;
}

/* lerp (src/utils.c:1-4) as value_at_wl uses it (src/spectrum.c:150-162) */
__device__ __forceinline__ double drt_lerp(double x, double x0, double x1, double y0, double y1)
{
    return y0 + ((x - x0) * ((y1 - y0) / (x1 - x0)));
}
