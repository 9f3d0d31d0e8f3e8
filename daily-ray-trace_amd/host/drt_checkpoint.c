/*
 * drt_checkpoint.c -- the .spd outputs of render_image() (src/daily_ray_trace.c:758-770) and the checkpoints a resumed run
 * continues from (the reference only sketches progressive accumulation, src/daily_ray_trace.c:620-633).
 *
 * A CHECKPOINT is three data files -- sum+filter, mean, un-normalised variance -- under generation names
 * (<output_spd>.ck<g>, <average_spd>.ck<g>, <variance_spd>.raw.ck<g>, g = 0 or 1) plus ONE manifest (<output_spd>.ckpt) that
 * names the generation, the samples it holds and the job it belongs to. Writing one: the three files of the generation the
 * manifest does NOT name are written and fsync'ed; then the new manifest goes to a temporary name, is fsync'ed and renamed
 * over the old one -- that single rename is the switch -- and the directory is fsync'ed; only then are the older
 * generation's files removed. A kill at any moment therefore leaves the previous complete checkpoint (before the rename) or
 * the new complete one (after it), never neither and never a mixture. The reference's three outputs under their own names
 * (sum+filter, mean, max-normalised variance) are written after the switch, each through a temporary name + rename; the
 * first two are hard links to the generation files where the file system allows, so a checkpoint moves the film to disk
 * once. They are what a viewer reads and are NOT what a resumed run trusts.
 *
 * A resumed run accepts a checkpoint only with a manifest that agrees with the job -- image size, wavelength grid (count,
 * first wavelength, interval), seed, max_cast_depth, pixel scheme and a hash of the scene file -- and with the files (headers,
 * sizes, filter sum of every pixel = the manifest's sample count, mean = sum / n on a spread of pixels); anything else
 * restarts from sample 0 with a message saying why.
 */
#include "drt_host.h"

#include <fcntl.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

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

#define CK_PATH 128
static void tmp_name(char *dst, size_t cap, const char *path) { snprintf(dst, cap, "%s.tmp", path); }
static int manifest_path(char *dst, size_t cap, const config_arguments *config) { return snprintf(dst, cap, "%s.ckpt", config->output_spd) < (int)cap ? 0 : -1; }
/* the three data files of generation g: sum+filter, mean, un-normalised variance */
static void generation_paths(const config_arguments *config, int g, char out[3][CK_PATH])
{
    snprintf(out[0], CK_PATH, "%s.ck%d", config->output_spd, g);
    snprintf(out[1], CK_PATH, "%s.ck%d", config->average_spd, g);
    snprintf(out[2], CK_PATH, "%s.raw.ck%d", config->variance_spd, g);
}

static int sync_path(const char *path, int directory)
{
    int fd = open(path, directory ? (O_RDONLY | O_DIRECTORY) : O_RDONLY);
    if (fd < 0) return -1;
    int rc = fsync(fd);
    close(fd);
    return rc;
}
static void sync_parent(const char *path)
{
    char dir[CK_PATH];
    snprintf(dir, sizeof(dir), "%s", path);
    char *slash = strrchr(dir, '/');
    if (slash) *slash = 0; else snprintf(dir, sizeof(dir), ".");
    (void)sync_path(dir[0] ? dir : "/", 1);
}

/* FNV-1a over the scene file's bytes (0 when it cannot be read): a resumed run must be the same scene */
static u64 scene_hash(const char *path)
{
    FILE *f = fopen(path, "rb");
    if (!f) return 0;
    u64 h = 0xcbf29ce484222325ull;
    int c;
    while ((c = fgetc(f)) != EOF) h = (h ^ (u64)(unsigned char)c) * 0x100000001b3ull;
    fclose(f);
    return h ? h : 1;
}

typedef struct
{
    unsigned version, samples, width, height, wavelengths, generation, max_depth, pixel_scheme;
    unsigned long long seed, scene;
    double min_wl, interval;
} manifest;

static int read_manifest(const char *mpath, manifest *m)
{
    FILE *mf = fopen(mpath, "r");
    if (!mf) return -1;
    memset(m, 0, sizeof(*m));
    int got = fscanf(mf, "drt-checkpoint %u samples %u width %u height %u wavelengths %u seed %llu generation %u max_depth %u pixel_scheme %u min_wl %lf interval %lf scene %llx",
                     &m->version, &m->samples, &m->width, &m->height, &m->wavelengths, &m->seed, &m->generation, &m->max_depth, &m->pixel_scheme,
                     &m->min_wl, &m->interval, &m->scene);
    fclose(mf);
    return (got == 12 && m->version == 2 && m->generation <= 1) ? 0 : -2;
}

/* one of the reference's outputs under its own name: a hard link to `same_as` when given and possible, else written out */
static int publish(const char *path, const char *same_as, u32 width, u32 height, u32 S, u32 has_filter, f64 min_wl, f64 interval, const f64 *data)
{
    char tmp[CK_PATH];
    tmp_name(tmp, sizeof(tmp), path);
    remove(tmp);
    if (!(same_as && link(same_as, tmp) == 0) && drt_host_write_spd(tmp, width, height, S, has_filter, min_wl, interval, data)) return refuse("could not write %s", tmp);
    if (rename(tmp, path) != 0) return refuse("could not move %s into place", tmp);
    return 0;
}

int drt_host_write_outputs(const config_arguments *config, u32 width, u32 height, u32 S, f64 min_wl, f64 interval,
                           const f64 *dst_pixels, const f64 *dst_avgs, const f64 *dst_vars, int with_raw_variance,
                           u32 samples_done, u64 seed)
{
    g_ckpt_error[0] = 0;
    u64 num_pixels = (u64)width * height;
    char mpath[CK_PATH], gen[3][CK_PATH], old[3][CK_PATH];
    if (manifest_path(mpath, sizeof(mpath), config)) return refuse("output path too long");
    manifest prev;
    const int have_prev = read_manifest(mpath, &prev) == 0;
    const int g = have_prev ? 1 - (int)prev.generation : 0;
    generation_paths(config, g, gen);
    generation_paths(config, 1 - g, old);
    if (with_raw_variance)
    {
        /* the new generation, complete and on disk, before the manifest names it */
        const f64 *data[3] = {dst_pixels, dst_avgs, dst_vars};
        for (int k = 0; k < 3; k += 1)
            if (drt_host_write_spd(gen[k], width, height, S, k == 0, min_wl, interval, data[k]) || sync_path(gen[k], 0))
            {
                for (int j = 0; j <= k; j += 1) remove(gen[j]);
                return refuse("could not write %s", gen[k]);
            }
        char mtmp[CK_PATH];
        tmp_name(mtmp, sizeof(mtmp), mpath);
        FILE *f = fopen(mtmp, "w");
        if (!f) return refuse("could not write %s", mtmp);
        int ok = fprintf(f, "drt-checkpoint 2\nsamples %u\nwidth %u\nheight %u\nwavelengths %u\nseed %llu\ngeneration %d\nmax_depth %u\npixel_scheme %u\n"
                            "min_wl %.17g\ninterval %.17g\nscene %llx\n",
                         samples_done, width, height, S, (unsigned long long)seed, g, config->max_cast_depth, (unsigned)config->pixel_scheme, min_wl, interval,
                         (unsigned long long)scene_hash(config->input_scene)) > 0;
        ok = (fflush(f) == 0) && ok;
        ok = (fsync(fileno(f)) == 0) && ok;
        ok = (fclose(f) == 0) && ok;
        if (!ok || rename(mtmp, mpath) != 0) return refuse("could not write %s", mpath); /* THE switch: before it the old checkpoint stands, after it the new one */
        sync_parent(mpath);
        for (int k = 0; k < 3; k += 1) remove(old[k]);
    }
    /* the reference's three outputs under their own names; the variance is max-normalised per pixel before writing
     * (src/daily_ray_trace.c:766-769), on a copy */
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
    int rc = publish(config->output_spd, with_raw_variance ? gen[0] : NULL, width, height, S, 1, min_wl, interval, dst_pixels);
    if (!rc) rc = publish(config->average_spd, with_raw_variance ? gen[1] : NULL, width, height, S, 0, min_wl, interval, dst_avgs);
    if (!rc) rc = publish(config->variance_spd, NULL, width, height, S, 0, min_wl, interval, norm);
    free(norm);
    if (rc) return rc;
    if (!with_raw_variance)
    {
        /* a final write outside checkpoint mode: whatever checkpoint there was described an earlier state of these outputs */
        remove(mpath);
        for (int k = 0; k < 3; k += 1) { remove(gen[k]); remove(old[k]); }
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
    char mpath[CK_PATH], paths[3][CK_PATH];
    if (manifest_path(mpath, sizeof(mpath), config)) return refuse("output path too long");
    manifest m;
    int mrc = read_manifest(mpath, &m);
    if (mrc == -1) return refuse("no checkpoint manifest %s", mpath);
    if (mrc) return refuse("%s is not a checkpoint manifest of this version", mpath);
    const u32 n = m.samples;
    if (m.width != width || m.height != height || m.wavelengths != S)
        return refuse("checkpoint is %ux%u with %u wavelengths, the job %ux%u with %u", m.width, m.height, m.wavelengths, width, height, S);
    if (m.seed != seed) return refuse("checkpoint was rendered with seed %llu, the job uses %llu", m.seed, (unsigned long long)seed);
    if (m.max_depth != config->max_cast_depth) return refuse("checkpoint was rendered with max_cast_depth %u, the job uses %u", m.max_depth, config->max_cast_depth);
    if (m.pixel_scheme != (unsigned)config->pixel_scheme) return refuse("checkpoint was rendered with pixel scheme %u, the job uses %u", m.pixel_scheme, (unsigned)config->pixel_scheme);
    if (m.min_wl != config->min_wl || m.interval != config->wl_interval)
        return refuse("checkpoint is on the wavelength grid %g + k %g, the job on %g + k %g", m.min_wl, m.interval, config->min_wl, config->wl_interval);
    if (m.scene != scene_hash(config->input_scene)) return refuse("checkpoint was rendered from another scene file than %s is now", config->input_scene);
    if (n == 0) return refuse("checkpoint holds no samples");
    const u64 num_pixels = (u64)width * height;
    generation_paths(config, (int)m.generation, paths);
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
        if (dst_pixels[px * (S + 1) + S] != (f64)n) return refuse("%s holds %g samples at pixel %llu, the manifest says %u", paths[0], dst_pixels[px * (S + 1) + S], (unsigned long long)px, n);
    /* the mean must be the mean of THESE sums: mean = sum / n to rounding, on a spread of pixels */
    const u64 stride = num_pixels > 4096 ? num_pixels / 4096 : 1;
    for (u64 px = 0; px < num_pixels; px += stride)
        for (u32 i = 0; i < S; i += 1)
        {
            f64 sum = dst_pixels[px * (S + 1) + i], mean = dst_avgs[px * S + i];
            f64 tol = 1e-9 * (fabs(sum) / n) + 1e-300;
            if (!(fabs(mean - sum / n) <= tol) && isfinite(sum))
                return refuse("%s does not belong to %s (pixel %llu: mean %g, sum / %u = %g)", paths[1], paths[0], (unsigned long long)px, mean, n, sum / n);
            if (dst_vars[px * S + i] < 0.0) return refuse("%s holds a negative variance sum", paths[2]);
        }
    *samples_done = n;
    return 0;
}
