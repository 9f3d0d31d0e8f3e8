/*
 * drt_render.c -- render_image() for the POSIX + HIP host (replaces src/daily_ray_trace.c:635-777).
 *
 * Same inputs (config_arguments) and outputs (three .spd files, then the three BMPs of the reference's
 * main()). The `for sample / for y / for x` loop (src/daily_ray_trace.c:710-745) is the C-ABI launcher
 * (include/drt_hip.h: drt_group_create / drt_group_render / drt_group_read_film, the session form of
 * drt_render_tile over one or several GPUs, so the film can stay on the device between checkpoints); everything around it stays plain C.
 * There is no CPU fallback: if the launcher fails, render_image reports it and exits.
 */
#include "drt_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static f64 now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (f64)ts.tv_sec * 1000.0 + (f64)ts.tv_nsec * 1e-6;
}

int render_image_ex(config_arguments *config, const drt_host_options *opt, drt_stats *stats_out)
{
    u32 width = config->output_width, height = config->output_height;
    spd_tables_csvs csvs;
    csvs.white = config->white_spd;  csvs.cmf_x = config->cmf_x;      csvs.cmf_y = config->cmf_y;
    csvs.cmf_z = config->cmf_z;      csvs.rgb_red = config->red_spd;  csvs.rgb_green = config->green_spd;
    csvs.rgb_blue = config->blue_spd; csvs.rgb_cyan = config->cyan_spd; csvs.rgb_magenta = config->magenta_spd;
    csvs.rgb_yellow = config->yellow_spd;

    /* material `csv` entries live next to the table CSVs (the reference hard-codes "spectra\") */
    char spectra_dir[128];
    snprintf(spectra_dir, sizeof(spectra_dir), "%s", config->white_spd);
    char *slash = strrchr(spectra_dir, '/');
    if (slash) *slash = 0; else snprintf(spectra_dir, sizeof(spectra_dir), "spectra");

    drt_host_scene *hs = drt_host_load_scene(config->input_scene, spectra_dir, &csvs, width, height,
                                             config->min_wl, config->max_wl, config->wl_interval);
    if (!hs)
    {
        fprintf(stderr, "render_image: %s\n", drt_host_last_error());
        return -1;
    }
    const drt_scene *scene = drt_host_scene_data(hs);
    u32 S = scene->num_wavelengths;
    u64 num_pixels = (u64)width * height;

    /* zero-filled accumulators (VirtualAlloc semantics, src/daily_ray_trace.c:689-691), 64-bit sizes */
    f64 *dst_pixels = (f64 *)calloc(num_pixels * (S + 1), sizeof(f64));
    f64 *dst_avgs = (f64 *)calloc(num_pixels * S, sizeof(f64));
    f64 *dst_vars = (f64 *)calloc(num_pixels * S, sizeof(f64));
    if (!dst_pixels || !dst_avgs || !dst_vars)
    {
        fprintf(stderr, "render_image: out of memory for %llu pixels\n", (unsigned long long)num_pixels);
        return -1;
    }

    drt_params p;
    memset(&p, 0, sizeof(p));
    p.width = width;
    p.height = height;
    p.tile_w = width;
    p.tile_h = height;
    p.row_stride = 1;
    p.spp = config->num_pixel_samples;
    p.max_depth = config->max_cast_depth;
    p.pixel_scheme = (u32)config->pixel_scheme;
    p.seed = opt ? opt->seed : 1;
    p.mode = DRT_MODE_SPECTRAL;
    p.device = opt ? opt->device : 0;
    p.batch_spp = opt ? opt->batch_spp : 0;

    /*
     * Progressive accumulation (the reference only sketches it, src/daily_ray_trace.c:620-633): the
     * film lives on the device for the whole render; every `checkpoint_spp` samples the three .spd files are
     * rewritten, plus the un-normalised variance (<variance_spd>.raw) that a resumed run needs. A resumed run
     * (opt->resume) reloads those files, takes the number of samples done from the filter sum, and continues --
     * bit-identical to an uninterrupted run, because sample k always uses the same per-path seeds.
     */
    drt_stats stats;
    memset(&stats, 0, sizeof(stats));
    u32 done = 0;
    const u64 seed = p.seed;
    if (opt && opt->resume)
    {
        /* host/drt_checkpoint.c: accepted only when manifest, headers, sizes, filter sums and means agree with the job */
        if (drt_host_load_checkpoint(config, width, height, S, seed, dst_pixels, dst_avgs, dst_vars, &done) == 0)
        {
            if (!(opt && opt->quiet)) printf("Resuming after %u samples\n", done);
        }
        else
        {
            fprintf(stderr, "render_image: not resuming (%s), starting at sample 0\n", drt_host_checkpoint_error());
            memset(dst_pixels, 0, num_pixels * (S + 1) * sizeof(f64));
            memset(dst_avgs, 0, num_pixels * S * sizeof(f64));
            memset(dst_vars, 0, num_pixels * S * sizeof(f64));
            done = 0;
        }
    }
    f64 t0 = now_ms();
    int rc = 0;
    /* one device, or several at once with the image rows dealt cyclically over them (one host thread, drt_group_*) */
    int32_t one_device = p.device;
    const int32_t *devices = &one_device;
    u32 n_devices = 1;
    if (opt && opt->all_devices) { devices = NULL; n_devices = 0; }
    else if (opt && opt->n_devices) { devices = opt->devices; n_devices = opt->n_devices; }
    drt_group *ctx = drt_group_create(scene, drt_host_camera_data(hs), &p, devices, n_devices);
    if (!ctx) rc = -1;
    if (!rc && !(opt && opt->quiet) && drt_group_size(ctx) > 1) printf("Rendering on %u devices\n", drt_group_size(ctx));
    if (!rc && done) rc = drt_group_write_film(ctx, dst_pixels, dst_avgs, dst_vars);
    u32 step = (opt && opt->checkpoint_spp) ? opt->checkpoint_spp : p.spp;
    while (!rc && done < p.spp)
    {
        u32 n = (p.spp - done < step) ? p.spp - done : step;
        if ((rc = drt_group_render(ctx, done, n))) break;
        done += n;
        if (done < p.spp) /* a checkpoint: the final write below uses the same code */
        {
            if ((rc = drt_group_read_film(ctx, dst_pixels, dst_avgs, dst_vars))) break;
            if (drt_host_write_outputs(config, width, height, S, scene->min_wavelength, scene->wavelength_interval, dst_pixels, dst_avgs, dst_vars, 1, done, seed))
                fprintf(stderr, "render_image: checkpoint write failed: %s\n", drt_host_checkpoint_error());
            if (!(opt && opt->quiet)) printf("Checkpoint at %u / %u samples\n", done, p.spp);
        }
    }
    if (!rc) rc = drt_group_read_film(ctx, dst_pixels, dst_avgs, dst_vars);
    if (!rc) rc = drt_group_get_stats(ctx, &stats);
    /* the .bmp pixels come from the film while it is still on the device (drt_read_bgra: the same bytes as converting the
     * .spd files on the host, host/drt_bmp.c, without reading 1.7 GB back from disk) */
    u8 *bgra[3] = { NULL, NULL, NULL };
    const char *bmp_path[3] = { config->output_bmp, config->average_bmp, config->variance_bmp };
    for (int k = 0; !rc && k < 3 && config->output_bmp[0]; k += 1)
    {
        if (!bmp_path[k][0]) continue;
        bgra[k] = (u8 *)malloc(num_pixels * 4 + 4);
        if (!bgra[k]) { rc = -3; break; }
        rc = drt_group_read_bgra(ctx, k, bgra[k]);
    }
    drt_group_destroy(ctx);
    f64 t1 = now_ms();
    if (rc != 0)
    {
        fprintf(stderr, "render_image: the HIP launcher failed (%d): %s\n", rc, drt_last_error());
        for (int k = 0; k < 3; k += 1) free(bgra[k]);
        free(dst_vars);
        free(dst_avgs);
        free(dst_pixels);
        drt_host_free_scene(hs);
        return rc;
    }
    if (!(opt && opt->quiet))
    {
        /* the reference's report, src/daily_ray_trace.c:753-756; a "sample" is one sample pass over the image at the rate of one
         * kernel pair (drt_stats, include/drt_hip.h) */
        printf("Min sample time: %fms\n", stats.min_sample_ms);
        printf("Max sample time: %fms\n", stats.max_sample_ms);
        printf("Avg sample time: %fms\n", stats.avg_sample_ms);
        printf("Total render time: %fms (device %fms: trace %fms, shade+film %fms, %u kernel pairs)\n", t1 - t0, stats.total_ms,
               stats.trace_ms, stats.shade_ms, stats.launches);
        printf("Paths: %llu  closest-hit scans/path: %.3f  shaded vertices/path: %.3f  Mpaths/s (device): %.2f\n",
               (unsigned long long)stats.paths, (f64)stats.closest_hit_scans / (f64)stats.paths,
               (f64)stats.shaded_vertices / (f64)stats.paths, (f64)stats.paths / (stats.total_ms * 1e3));
    }

    int wrc = drt_host_write_outputs(config, width, height, S, scene->min_wavelength, scene->wavelength_interval, dst_pixels, dst_avgs, dst_vars,
                                     (opt && opt->checkpoint_spp) ? 1 : 0, done, seed);
    if (wrc) fprintf(stderr, "render_image: %s\n", drt_host_checkpoint_error());
    int w0 = wrc, w1 = 0, w2 = 0;
    /* post-process like the reference's main(): each film -> linear RGB -> BMP (src/win32_main.c:150-152) */
    if (!(w0 || w1 || w2))
    {
        int bad = 0;
        for (int k = 0; k < 3; k += 1)
            if (bgra[k]) bad |= drt_host_write_bmp_bgra(bmp_path[k], width, height, bgra[k]);
        if (bad) fprintf(stderr, "render_image: could not write one of the .bmp outputs\n");
    }
    for (int k = 0; k < 3; k += 1) free(bgra[k]);
    if (stats_out) *stats_out = stats;
    free(dst_vars);
    free(dst_avgs);
    free(dst_pixels);
    drt_host_free_scene(hs);
    return (w0 || w1 || w2) ? -2 : 0;
}

void render_image(config_arguments *config)
{
    drt_host_options opt;
    memset(&opt, 0, sizeof(opt));
    const char *e;
    opt.seed = 1;
    if ((e = getenv("DRT_DEVICE"))) opt.device = atoi(e);
    if ((e = getenv("DRT_DEVICES")))
    {
        if (strcmp(e, "all") == 0) opt.all_devices = 1;
        else
            for (const char *c = e; *c && opt.n_devices < 16;)
            {
                char *end;
                long d = strtol(c, &end, 10);
                if (end == c) break;
                opt.devices[opt.n_devices++] = (int32_t)d;
                c = (*end == ',') ? end + 1 : end;
            }
    }
    if ((e = getenv("DRT_SEED"))) opt.seed = strtoull(e, NULL, 0);
    if ((e = getenv("DRT_BATCH_SPP"))) opt.batch_spp = (u32)atoi(e);
    if ((e = getenv("DRT_CHECKPOINT_SPP"))) opt.checkpoint_spp = (u32)atoi(e);
    if ((e = getenv("DRT_RESUME"))) opt.resume = (u32)atoi(e);
    if (render_image_ex(config, &opt, NULL) != 0) exit(-1);
}
