/*
 * drt_checkpoint.c -- the .spd outputs of render_image() (src/daily_ray_trace.c:758-770) written so that a kill at any
 * moment leaves either the previous complete set or the new complete set, and the test a resumed run applies before
 * it trusts a set (the reference only sketches progressive accumulation, src/daily_ray_trace.c:620-633).
 *
 * A set is four files -- sum+filter, mean, max-normalised variance (the reference's three outputs) and the un-normalised
 * variance a resumed run needs (<variance_spd>.raw) -- plus a manifest (<output_spd>.ckpt) naming the samples they hold.
 * Writing a set: every file goes to <path>.tmp first; only when all of them are complete is the old manifest removed,
 * the files renamed into place and the new manifest written (itself through a .tmp + rename). A resumed run accepts a set
 * only with a manifest that agrees with the job (size, wavelength grid, seed) and with the files (headers, file sizes,
 * filter sums of every pixel = the manifest's sample count, mean = sum / n on a sample of pixels); anything else restarts
 * from sample 0 with a message saying why.
 */
#include "drt_host.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_ckpt_error[256];
const char *drt_host_checkpoint_error(void) { return g_ckpt_error; }
static int refuse(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ckpt_error, sizeof(g_ckpt_error), fmt, ap);
    va_end(ap);
    return -1;
}

static void tmp_name(char *dst, size_t cap, const char *path) { snprintf(dst, cap, "%s.tmp", path); }

static int manifest_path(char *dst, size_t cap, const config_arguments *config) { return snprintf(dst, cap, "%s.ckpt", config->output_spd) < (int)cap ? 0 : -1; }

int drt_host_write_outputs(const config_arguments *config, u32 width, u32 height, u32 S, f64 min_wl, f64 interval,
                           const f64 *dst_pixels, const f64 *dst_avgs, const f64 *dst_vars, int with_raw_variance,
                           u32 samples_done, u64 seed)
{
    g_ckpt_error[0] = 0;
    u64 num_pixels = (u64)width * height;
    char raw_path[96], mpath[96];
    snprintf(raw_path, sizeof(raw_path), "%s.raw", config->variance_spd);
    if (manifest_path(mpath, sizeof(mpath), config)) return refuse("output path too long");
    /* the variance is max-normalised per pixel before writing (src/daily_ray_trace.c:766-769), on a copy */
    f64 *norm = (f64 *)malloc(num_pixels * S * sizeof(f64));
    if (!norm) return refuse("out of memory");
    for (u64 px = 0; px < num_pixels; px += 1)
    {
        const f64 *v = dst_vars + px * S;
        f64 *o = norm + px * S;
        f64 highest = 0.0;
        for (u32 i = 0; i < S; i += 1) if (v[i] > highest) highest = v[i];
        for (u32 i = 0; i < S; i += 1) o[i] = v[i] / highest;
    }
    const char *paths[4] = {config->output_spd, config->variance_spd, config->average_spd, with_raw_variance ? raw_path : NULL};
    const f64 *data[4] = {dst_pixels, norm, dst_avgs, dst_vars};
    char tmp[4][112];
    int rc = 0;
    for (int k = 0; k < 4 && !rc; k += 1)
    {
        if (!paths[k]) continue;
        tmp_name(tmp[k], sizeof(tmp[k]), paths[k]);
        if (drt_host_write_spd(tmp[k], width, height, S, k == 0, min_wl, interval, data[k])) rc = refuse("could not write %s", tmp[k]);
    }
    free(norm);
    if (rc)
    {
        for (int k = 0; k < 4; k += 1) if (paths[k]) { tmp_name(tmp[k], sizeof(tmp[k]), paths[k]); remove(tmp[k]); }
        return rc;
    }
    /* all four are complete: from here to the new manifest there is no manifest, so a kill in between is seen by resume */
    remove(mpath);
    for (int k = 0; k < 4; k += 1)
        if (paths[k] && rename(tmp[k], paths[k]) != 0) return refuse("could not move %s into place", tmp[k]);
    if (with_raw_variance)
    {
        char mtmp[112];
        tmp_name(mtmp, sizeof(mtmp), mpath);
        FILE *f = fopen(mtmp, "w");
        if (!f) return refuse("could not write %s", mtmp);
        int ok = fprintf(f, "drt-checkpoint 1\nsamples %u\nwidth %u\nheight %u\nwavelengths %u\nseed %llu\n", samples_done, width, height, S,
                         (unsigned long long)seed) > 0;
        ok = (fclose(f) == 0) && ok;
        if (!ok || rename(mtmp, mpath) != 0) return refuse("could not write %s", mpath);
    }
    return 0;
}

static long file_size(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fclose(f);
    return n;
}

int drt_host_load_checkpoint(const config_arguments *config, u32 width, u32 height, u32 S, u64 seed, f64 *dst_pixels,
                             f64 *dst_avgs, f64 *dst_vars, u32 *samples_done)
{
    g_ckpt_error[0] = 0;
    *samples_done = 0;
    char raw_path[96], mpath[96];
    snprintf(raw_path, sizeof(raw_path), "%s.raw", config->variance_spd);
    if (manifest_path(mpath, sizeof(mpath), config)) return refuse("output path too long");
    FILE *mf = fopen(mpath, "r");
    if (!mf) return refuse("no checkpoint manifest %s (no checkpoint, or one that was cut short)", mpath);
    unsigned version = 0, n = 0, w = 0, h = 0, s = 0;
    unsigned long long sd = 0;
    int got = fscanf(mf, "drt-checkpoint %u samples %u width %u height %u wavelengths %u seed %llu", &version, &n, &w, &h, &s, &sd);
    fclose(mf);
    if (got != 6 || version != 1) return refuse("%s is not a checkpoint manifest", mpath);
    if (w != width || h != height || s != S) return refuse("checkpoint is %ux%u with %u wavelengths, the job %ux%u with %u", w, h, s, width, height, S);
    if (sd != seed) return refuse("checkpoint was rendered with seed %llu, the job uses %llu", sd, (unsigned long long)seed);
    if (n == 0) return refuse("checkpoint holds no samples");
    const u64 num_pixels = (u64)width * height;
    const char *paths[3] = {config->output_spd, config->average_spd, raw_path};
    f64 *dst[3] = {dst_pixels, dst_avgs, dst_vars};
    for (int k = 0; k < 3; k += 1)
    {
        const u32 has_filter = k == 0;
        const u64 want = 40 + num_pixels * (S + has_filter) * sizeof(f64);
        long sz = file_size(paths[k]);
        if (sz < 0) return refuse("%s is missing", paths[k]);
        if ((u64)sz != want) return refuse("%s has %ld bytes, a %ux%u x %u file has %llu", paths[k], sz, width, height, S, (unsigned long long)want);
        spd_file_header hd;
        f64 *px = NULL;
        if (drt_host_read_spd(paths[k], &hd, &px) != 0) return refuse("%s cannot be read", paths[k]);
        if (hd.width_in_pixels != width || hd.height_in_pixels != height || hd.number_of_wavelengths != S || (hd.has_filter_values != 0) != (has_filter != 0))
        {
            free(px);
            return refuse("%s has another size or layout than the job", paths[k]);
        }
        memcpy(dst[k], px, num_pixels * (S + has_filter) * sizeof(f64));
        free(px);
    }
    /* every pixel's filter sum is the sample count (the filter value is 1.0, src/daily_ray_trace.c:616) */
    for (u64 px = 0; px < num_pixels; px += 1)
        if (dst_pixels[px * (S + 1) + S] != (f64)n) return refuse("%s holds %g samples at pixel %llu, the manifest says %u", config->output_spd, dst_pixels[px * (S + 1) + S], (unsigned long long)px, n);
    /* the mean must be the mean of THESE sums: mean = sum / n to rounding, on a spread of pixels */
    const u64 stride = num_pixels > 4096 ? num_pixels / 4096 : 1;
    for (u64 px = 0; px < num_pixels; px += stride)
        for (u32 i = 0; i < S; i += 1)
        {
            f64 sum = dst_pixels[px * (S + 1) + i], mean = dst_avgs[px * S + i];
            f64 tol = 1e-9 * (fabs(sum) / n) + 1e-300;
            if (!(fabs(mean - sum / n) <= tol) && isfinite(sum))
                return refuse("%s does not belong to %s (pixel %llu: mean %g, sum / %u = %g)", config->average_spd, config->output_spd, (unsigned long long)px, mean, n, sum / n);
            if (dst_vars[px * S + i] < 0.0) return refuse("%s holds a negative variance sum", raw_path);
        }
    *samples_done = n;
    return 0;
}
