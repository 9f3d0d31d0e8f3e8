/*
 * drt_spectrum.c -- host-side SPD construction: CSV resampling, RGB -> spectrum, blackbody.
 * Behaviour follows src/read_scene.c:797-872 and src/spectrum.c:84-119, :245-273 of the
 * reference, including what its CSV reader does with signs, duplicated rows, micrometre files
 * and wavelengths past the end of a file (see comments). Arrays are sized dynamically.
 */
#include "drt_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.1415926535897932385L /* a long double literal in the reference (src/types.h:1) */

static f64 lerp(f64 x, f64 x0, f64 x1, f64 y0, f64 y1) /* src/utils.c:1-4 */
{
    return y0 + ((x - x0) * ((y1 - y0) / (x1 - x0)));
}

/* The reference scans with three helpers that stop at NUL or at a 0xFF byte (char EOF). */
static int at_stop(const char *c) { return *c == 0 || *c == (char)EOF; }
static const char *next_newline(const char *c)
{
    for (;; c += 1)
    {
        if (*c == '\n') return c;
        if (at_stop(c)) return NULL;
    }
}
static const char *next_digit(const char *c)
{
    for (;; c += 1)
    {
        if (*c >= '0' && *c <= '9') return c;
        if (at_stop(c)) return NULL;
    }
}
static const char *next_char(const char *c, char want)
{
    for (;; c += 1)
    {
        if (*c == want) return c;
        if (at_stop(c)) return NULL;
    }
}

/*
 * CSV layout: first line is a heading; every other line is "wavelength, value".
 * As in the reference:
 *  - a sample is counted for every newline that still has a digit somewhere after it;
 *  - a number starts at the first DIGIT found, so a leading '-' or '.' is not part of it;
 *  - if the first wavelength is < 10 the file is taken to be in micrometres (x1000);
 *  - the table is walked forward only. A grid wavelength below the file's first entry extrapolates
 *    the first segment (as in the reference). One above the file's last entry makes the reference
 *    walk off its 128-entry arrays (undefined behaviour); here it extrapolates the last segment.
 */
u32 drt_host_csv_to_spectrum(const char *csv_path, f64 min_wl, f64 wl_interval, u32 num_samples, f64 *dst)
{
    FILE *f = fopen(csv_path, "rb");
    if (!f) return 0;
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)calloc((size_t)size + 2, 1);
    if (fread(buf, 1, (size_t)size, f) != (size_t)size) { fclose(f); free(buf); return 0; }
    fclose(f);

    u32 count = 0;
    for (const char *c = next_newline(buf); c != NULL; c = next_newline(c))
    {
        c += 1;
        if (next_digit(c)) count += 1;
    }
    f64 *wl = (f64 *)calloc((size_t)count + 2, sizeof(f64));
    f64 *val = (f64 *)calloc((size_t)count + 2, sizeof(f64));
    const char *c = next_newline(buf);
    c = c ? c + 1 : buf;
    for (u32 i = 0; i < count && c; i += 1)
    {
        c = next_digit(c);
        if (!c) break;
        wl[i] = atof(c);
        c = next_char(c, ',');
        if (!c) break;
        c = next_digit(c);
        if (!c) break;
        val[i] = atof(c);
        c = next_newline(c);
    }
    free(buf);

    if (count > 0 && wl[0] < 10.0)
        for (u32 i = 0; i < count; i += 1) wl[i] *= 1000.0;

    u32 k = 0;
    for (u32 s = 0; s < num_samples; s += 1)
    {
        f64 sample_wl = min_wl + ((f64)s) * wl_interval;
        for (; k + 2 < count && wl[k + 1] < sample_wl; k += 1);
        dst[s] = lerp(sample_wl, wl[k], wl[k + 1], val[k], val[k + 1]);
    }
    free(wl);
    free(val);
    return 1;
}

/* rgb_f64_to_spectrum, src/spectrum.c:84-119 (Smits-style: white*min + cmy*(mid-min) + rgb*(max-mid)).
 * tables: [7][S] in the order white, red, green, blue, cyan, magenta, yellow. */
void drt_host_rgb_to_spectrum(const f64 *t, u32 S, const f64 rgb[3], f64 *dst)
{
    const f64 *white = t;
    const f64 *rgb_spectra[3] = {t + 1 * S, t + 2 * S, t + 3 * S};
    const f64 *cmy_spectra[3] = {t + 4 * S, t + 5 * S, t + 6 * S};
    u32 idx[3] = {0, 1, 2};
    u32 tmp;
    if (rgb[idx[0]] > rgb[idx[1]]) { tmp = idx[1]; idx[1] = idx[0]; idx[0] = tmp; }
    if (rgb[idx[1]] > rgb[idx[2]]) { tmp = idx[2]; idx[2] = idx[1]; idx[1] = tmp; }
    if (rgb[idx[0]] > rgb[idx[1]]) { tmp = idx[1]; idx[1] = idx[0]; idx[0] = tmp; }
    u32 small = idx[0], mid = idx[1], large = idx[2];
    f64 diff_mid_small = rgb[mid] - rgb[small];
    f64 diff_large_mid = rgb[large] - rgb[mid];
    for (u32 i = 0; i < S; i += 1) dst[i] = white[i] * rgb[small];
    for (u32 i = 0; i < S; i += 1) dst[i] += cmy_spectra[small][i] * diff_mid_small;
    for (u32 i = 0; i < S; i += 1) dst[i] += rgb_spectra[large][i] * diff_large_mid;
}

/* compute_blackbody_power / generate_blackbody_spectrum, src/spectrum.c:245-273 (long double) */
void drt_host_blackbody_spectrum(f64 min_wl, f64 wl_interval, u32 S, f64 temperature, f64 *dst)
{
    const long double c = 2.99792458e8L;
    const long double h = 6.626176e-34L;
    const long double k = 1.380662e-23L;
    long double temp = (long double)temperature;
    for (u32 i = 0; i < S; i += 1)
    {
        long double wl_nm = (long double)(min_wl + (i * wl_interval));
        long double wl_m = wl_nm * 1e-9L;
        long double numerator = 2.0L * PI * h * c * c;
        long double lambda_5 = powl(wl_m, 5.0L);
        long double e_power = ((h * c) / k) / (temp * wl_m);
        long double e_term = expl(e_power);
        long double denominator = lambda_5 * (e_term - 1.0L);
        long double power = numerator / denominator;
        dst[i] = (f64)(power * 1e9L);
    }
}
