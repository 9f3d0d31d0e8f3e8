/*
 * drt_host.h -- the POSIX C host of the MI355X build: the part of daily-ray-trace that stays on
 * the CPU (config + .scn reading, SPD tables, scene/camera build, .spd writing) and hands the
 * per-pixel loop to libdrt_hip.so through include/drt_hip.h.
 *
 * Mirrors, with the same names and argument meaning:
 *   config_arguments / render_image()   src/daily_ray_trace.h:28-55, :177; src/daily_ray_trace.c:635
 *   parse_config()                      src/read_scene.c:604-765
 *   parse_scene() (superset grammar)    src/read_scene.c:345-602
 *   load_csv_file_to_spectrum()         src/read_scene.c:801-872
 *   init_spd_tables / rgb_f64_to_spectrum / generate_blackbody_spectrum   src/spectrum.c
 *   init_camera / init_scene / init_spd src/daily_ray_trace.c:49-211
 *   win32_platform.c file/timer/alloc   -> POSIX (calloc, stdio, clock_gettime)
 */
#ifndef DRT_HOST_H
#define DRT_HOST_H

#include <stdint.h>
#include "../../include/drt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t  u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef double   f64;

typedef enum
{
    FILM_SAMPLE_NONE,
    FILM_SAMPLE_CENTER,
    FILM_SAMPLE_RANDOM,
    FILM_SAMPLE_COUNT
} film_sample_scheme;

/* Same fields, order and sizes as the reference's config_arguments (1136 bytes). */
typedef struct
{
    u32  num_pixel_samples;
    u32  max_cast_depth;
    u32  output_width;
    u32  output_height;
    f64  min_wl;
    f64  max_wl;
    f64  wl_interval;
    char input_scene[64];
    char output_spd[64];
    char average_spd[64];
    char variance_spd[64];
    char output_bmp[64];
    char average_bmp[64];
    char variance_bmp[64];
    char white_spd[64];
    char cmf_x[64];
    char cmf_y[64];
    char cmf_z[64];
    char red_spd[64];
    char green_spd[64];
    char blue_spd[64];
    char cyan_spd[64];
    char magenta_spd[64];
    char yellow_spd[64];
    film_sample_scheme pixel_scheme;
} config_arguments;

/* .spd header, src/daily_ray_trace.h:59-68 (40 bytes) */
typedef struct
{
    u32 id;
    u32 width_in_pixels;
    u32 height_in_pixels;
    u32 number_of_wavelengths;
    u32 has_filter_values;
    f64 min_wavelength;
    f64 wavelength_interval;
} spd_file_header;

/* Host extras that the reference has no field for (device choice, RNG seed, batch size);
 * read from the environment by render_image(): DRT_DEVICE, DRT_DEVICES, DRT_SEED, DRT_BATCH_SPP, DRT_CHECKPOINT_SPP, DRT_RESUME. */
typedef struct
{
    int32_t  device;
    uint64_t seed;
    uint32_t batch_spp;
    uint32_t quiet;
    uint32_t checkpoint_spp; /* rewrite the .spd files every this many samples (0: only at the end) */
    uint32_t resume;         /* continue from the .spd files of an earlier (checkpointed) run */
    uint32_t n_devices;      /* > 0: render on devices[0..n_devices) at once, image rows dealt cyclically (drt_group_*); */
    int32_t  devices[16];    /* 0: the single `device` above. DRT_DEVICES="0,1,2,3" or "all" (every visible device) */
    uint32_t all_devices;
} drt_host_options;

/* Fills *config from the text of a config.cfg. Unknown keys are fatal (exit(-1)), like the reference.
 * Paths may use '\' or '/'. */
void parse_config(char *config_contents, u32 config_contents_size, config_arguments *config);
void print_config_arguments(config_arguments *config);

/* The drop-in: same signature as the reference. Writes the three .spd files. */
void render_image(config_arguments *config);
/* Same, with explicit options and statistics; returns 0 on success. */
int render_image_ex(config_arguments *config, const drt_host_options *opt, drt_stats *stats);

/* A loaded scene: owns every array the drt_scene/drt_camera inside point to. */
typedef struct drt_host_scene drt_host_scene;

typedef struct
{
    const char *white, *cmf_x, *cmf_y, *cmf_z, *rgb_red, *rgb_green, *rgb_blue, *rgb_cyan, *rgb_magenta, *rgb_yellow;
} spd_tables_csvs;

/*
 * init_spd_tables + load_scene in one call. `spectra_dir` is the directory `csv <file>` material
 * entries are looked up in (the reference hard-codes "spectra\\", src/daily_ray_trace.c:93).
 * tables==NULL uses <spectra_dir>/{white_rgb_to_spd,cmf_x,...}.csv.
 * Returns NULL and sets drt_host_last_error() when a file is missing; grammar errors exit(-1).
 */
drt_host_scene *drt_host_load_scene(const char *scene_path, const char *spectra_dir, const spd_tables_csvs *tables,
                                    u32 width_px, u32 height_px, f64 min_wl, f64 max_wl, f64 wl_interval);
/* Same from memory (the text of a .scn). */
drt_host_scene *drt_host_load_scene_text(const char *scene_text, u32 scene_size, const char *spectra_dir,
                                         const spd_tables_csvs *tables, u32 width_px, u32 height_px,
                                         f64 min_wl, f64 max_wl, f64 wl_interval);
void              drt_host_free_scene(drt_host_scene *s);
const drt_scene  *drt_host_scene_data(const drt_host_scene *s);
const drt_camera *drt_host_camera_data(const drt_host_scene *s);
const char       *drt_host_material_name(const drt_host_scene *s, u32 i);
const char       *drt_host_surface_name(const drt_host_scene *s, u32 i);
const char       *drt_host_last_error(void);

/* Standalone pieces, exported for the parity tests. */
/* load_csv_file_to_spectrum: resample a CSV onto the grid; returns 1, or 0 if the file is missing. */
u32  drt_host_csv_to_spectrum(const char *csv_path, f64 min_wl, f64 wl_interval, u32 num_samples, f64 *dst);
/* rgb_f64_to_spectrum with the 7 rgb tables given as [7][num_samples] (white,red,green,blue,cyan,magenta,yellow). */
void drt_host_rgb_to_spectrum(const f64 *rgb_tables, u32 num_samples, const f64 rgb[3], f64 *dst);
void drt_host_blackbody_spectrum(f64 min_wl, f64 wl_interval, u32 num_samples, f64 temperature, f64 *dst);
/* init_camera: fills *camera from position/target/roll/fov/fdepth/flength/aperture and the image size. */
void drt_host_init_camera(drt_camera *camera, const f64 position[3], const f64 target[3], f64 roll, f64 fov,
                          f64 fdepth, f64 flength, f64 aperture, u32 width_px, u32 height_px);

/* Name tables expanded from include/bdsf_list.h (same role as bdsf_name_list / dir_func_name_list). */
extern const char *bdsf_name_list[];
extern const u32   num_bdsfs_defined;
extern const char *dir_func_name_list[];
extern const u32   num_dir_funcs_defined;

/* .spd files */
int drt_host_write_spd(const char *path, u32 width, u32 height, u32 num_wl, u32 has_filter, f64 min_wl, f64 interval,
                       const f64 *pixels);
/* Reads a .spd; *pixels is malloc'ed. Returns 0 on success. */
int drt_host_read_spd(const char *path, spd_file_header *header, f64 **pixels);

/* The outputs of render_image() as one crash-safe set (host/drt_checkpoint.c): the three .spd files, with
 * with_raw_variance also <variance_spd>.raw and the manifest <output_spd>.ckpt a resumed run needs. Returns 0 on success. */
int drt_host_write_outputs(const config_arguments *config, u32 width, u32 height, u32 S, f64 min_wl, f64 interval,
                           const f64 *dst_pixels, const f64 *dst_avgs, const f64 *dst_vars, int with_raw_variance,
                           u32 samples_done, u64 seed);
/* Loads a checkpointed set into the (caller-allocated, full-frame) film buffers if, and only if, manifest, headers, file
 * sizes, filter sums and means all agree with each other and with the job; returns 0 and the samples held, or nonzero
 * (buffers then hold garbage: clear them) with the reason in drt_host_checkpoint_error(). */
int drt_host_load_checkpoint(const config_arguments *config, u32 width, u32 height, u32 S, u64 seed, f64 *dst_pixels,
                             f64 *dst_avgs, f64 *dst_vars, u32 *samples_done);
const char *drt_host_checkpoint_error(void);

/* .spd -> BMP post-process (spd_file_to_bmp, src/win32_main.c:115-121). cmf: [4][S] rows rw, x, y, z. */
void drt_host_spectrum_to_rgb(const f64 *cmf, u32 S, f64 interval, const f64 *spd, f64 rgb[3]);
int  drt_host_write_bmp(const char *path, u32 width, u32 height, const f64 *rgb);
int  drt_host_write_bmp_bgra(const char *path, u32 width, u32 height, const u8 *bgra);
int  drt_host_spd_file_to_bmp(const char *spd_path, const char *bmp_path, const f64 *cmf);

#ifdef __cplusplus
}
#endif
#endif
