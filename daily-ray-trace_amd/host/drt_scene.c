/*
 * drt_scene.c -- .scn / config.cfg reader and scene build for the POSIX host.
 *
 * Grammar: the reference's block grammar (src/read_scene.c:345-602, keywords src/keywords.h) as a
 * SUPERSET:
 *   - legacy camera keys `up/right/forward` (the five old shipped scenes) are accepted and mapped
 *     to target/roll; materials without `bdsfs/dir_func` get the plastic defaults; `vacuum`
 *     (base) and `escape` materials are synthesised when absent;
 *   - `center/origin/point_u/point_v/is_blackbody` aliases of example_scene.scn are accepted;
 *   - words may contain '/' and '-' (POSIX paths), array sizes are dynamic, names are bounded.
 * Scene build follows init_camera / init_spd / init_scene (src/daily_ray_trace.c:49-211):
 * one trailing all-zero material, escape forced black-body, unmatched material name -> index 0.
 * Output is the flat drt_scene / drt_camera of include/drt_hip.h.
 */
#include "drt_host.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.1415926535897932385L

static __thread char g_host_error[512];
const char *drt_host_last_error(void) { return g_host_error; }
static void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_host_error, sizeof(g_host_error), fmt, ap);
    va_end(ap);
}

/* ---- name tables from the X-macro list (same role as src/bdsf.h:10-17, :35-42) ---- */
#define BDSF(name) #name,
#define DIRF(name)
const char *bdsf_name_list[] = {
#include "../../include/bdsf_list.h"
};
#undef BDSF
#undef DIRF
#define BDSF(name)
#define DIRF(name) #name,
const char *dir_func_name_list[] = {
#include "../../include/bdsf_list.h"
};
#undef BDSF
#undef DIRF
const u32 num_bdsfs_defined = sizeof(bdsf_name_list) / sizeof(bdsf_name_list[0]);
const u32 num_dir_funcs_defined = sizeof(dir_func_name_list) / sizeof(dir_func_name_list[0]);

/* ------------------------------------------------------------------------------------------ */
/* Token stream: items separated by whitespace and commas. A token is a number when strtod     */
/* consumes all of it, otherwise a word.                                                        */

typedef struct
{
    const char *text, *end, *loc;
    char  word[256];
    f64   value;
    int   is_number;
    int   at_end;
    const char *what; /* "scene" or "config" for messages */
} token_stream;

static int is_sep(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == ',' || c == 0 || c == '\f' || c == '\v'; }

static void ts_init(token_stream *ts, const char *text, u32 size, const char *what)
{
    memset(ts, 0, sizeof(*ts));
    ts->text = text;
    ts->end = text + size;
    ts->loc = text;
    ts->what = what;
}

/* Reads the token at *loc (does not move the stream); returns the position after it. */
static const char *ts_scan(const token_stream *ts, const char *loc, char *word, f64 *value, int *is_number, int *at_end)
{
    while (loc < ts->end && is_sep(*loc)) loc += 1;
    if (loc >= ts->end)
    {
        *at_end = 1;
        word[0] = 0;
        *is_number = 0;
        return loc;
    }
    *at_end = 0;
    size_t n = 0;
    while (loc < ts->end && !is_sep(*loc))
    {
        if (n + 1 < 256) word[n++] = *loc;
        loc += 1;
    }
    word[n] = 0;
    char *num_end = NULL;
    f64 v = strtod(word, &num_end);
    *is_number = (num_end != word && *num_end == 0);
    *value = *is_number ? v : 0.0;
    return loc;
}

static void ts_next(token_stream *ts) { ts->loc = ts_scan(ts, ts->loc, ts->word, &ts->value, &ts->is_number, &ts->at_end); }
static void ts_peek(const token_stream *ts, char *word, int *at_end)
{
    f64 v;
    int isnum;
    ts_scan(ts, ts->loc, word, &v, &isnum, at_end);
}

/* parse_error(), src/read_scene.c:196-203: report and exit(-1) */
static void parse_error(const token_stream *ts, const char *expect)
{
    u32 line = 1;
    for (const char *c = ts->text; c < ts->loc && c < ts->end; c += 1) if (*c == '\n') line += 1;
    printf("ERROR: %s parse error near line %u at token \"%s\"%s%s\n", ts->what, line, ts->word,
           expect ? ", expected " : "", expect ? expect : "");
    exit(-1);
}

static f64 parse_float(token_stream *ts)
{
    ts_next(ts);
    if (ts->at_end || !ts->is_number) parse_error(ts, "a number");
    return ts->value;
}
static u32 parse_uint(token_stream *ts) { return (u32)parse_float(ts); }
static void parse_vec3(token_stream *ts, f64 dst[3]) { dst[0] = parse_float(ts); dst[1] = parse_float(ts); dst[2] = parse_float(ts); }
static void parse_word(token_stream *ts, char *dst, size_t cap)
{
    ts_next(ts);
    if (ts->at_end) parse_error(ts, "a word");
    snprintf(dst, cap, "%s", ts->word);
}
static u32 parse_bool(token_stream *ts)
{
    ts_next(ts);
    if (strcmp(ts->word, "true") == 0) return 1;
    if (strcmp(ts->word, "false") == 0) return 0;
    parse_error(ts, "true or false");
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Parsed (input) form                                                                          */

typedef enum { SPD_METHOD_NONE, SPD_METHOD_RGB, SPD_METHOD_CSV, SPD_METHOD_BLACKBODY, SPD_METHOD_CONST } spd_input_method;

typedef struct
{
    spd_input_method method;
    u32  has_scale_factor;
    f64  scale_factor;
    f64  rgb[3];
    char csv[128];
    f64  blackbody_temp;
    f64  constant;
} spd_input;

typedef struct
{
    char name[64];
    u32  is_base_material, is_escape_material, is_black_body, is_emissive;
    f64  shininess, roughness;
    spd_input emission, diffuse, glossy, mirror, refract, extinct;
    u32  num_bdsfs, has_bdsfs;
    u32  bdsfs[DRT_MAX_BDSFS];
    u32  dir_func, has_dir_func;
} material_input;

typedef struct
{
    char name[64];
    u32  type;
    f64  position[3];
    char material_name[64];
    f64  radius;
    f64  u[3], v[3];
} surface_input;

typedef struct
{
    f64 target[3], position[3];
    f64 roll, fov, fdepth, flength, aperture;
    u32 has_target, has_legacy;
    f64 up[3], right[3], forward[3];
} camera_input;

typedef struct
{
    camera_input    camera;
    material_input *materials;
    u32             num_materials, cap_materials;
    surface_input  *surfaces;
    u32             num_surfaces, cap_surfaces;
} scene_input;

static int is_block_word(const char *w) { return strcmp(w, "Camera") == 0 || strcmp(w, "Material") == 0 || strcmp(w, "Surface") == 0; }

/* parse_spd_method, src/read_scene.c:263-305 */
static void parse_spd_method(token_stream *ts, spd_input *dst)
{
    ts_next(ts);
    if (strcmp(ts->word, "rgb") == 0) { dst->method = SPD_METHOD_RGB; parse_vec3(ts, dst->rgb); }
    else if (strcmp(ts->word, "csv") == 0) { dst->method = SPD_METHOD_CSV; parse_word(ts, dst->csv, sizeof(dst->csv)); }
    else if (strcmp(ts->word, "blackbody") == 0) { dst->method = SPD_METHOD_BLACKBODY; dst->blackbody_temp = parse_float(ts); }
    else if (strcmp(ts->word, "constant") == 0) { dst->method = SPD_METHOD_CONST; dst->constant = parse_float(ts); }
    else parse_error(ts, "rgb, csv, blackbody or constant");
    char look[256];
    int at_end;
    ts_peek(ts, look, &at_end);
    if (!at_end && strcmp(look, "scale") == 0)
    {
        ts_next(ts);
        dst->has_scale_factor = 1;
        dst->scale_factor = parse_float(ts);
    }
}

static int lookup_name(const char *name, const char **table, u32 n)
{
    for (u32 i = 0; i < n; i += 1) if (strcmp(name, table[i]) == 0) return (int)i;
    return -1;
}

/* parse_camera, src/read_scene.c:345-396, plus the legacy keys */
static void parse_camera(token_stream *ts, camera_input *cam)
{
    char look[256];
    int at_end;
    for (ts_peek(ts, look, &at_end); !at_end && !is_block_word(look); ts_peek(ts, look, &at_end))
    {
        ts_next(ts);
        const char *w = ts->word;
        if (strcmp(w, "position") == 0) parse_vec3(ts, cam->position);
        else if (strcmp(w, "target") == 0) { parse_vec3(ts, cam->target); cam->has_target = 1; }
        else if (strcmp(w, "roll") == 0) cam->roll = parse_float(ts);
        else if (strcmp(w, "fov") == 0) cam->fov = parse_float(ts);
        else if (strcmp(w, "fdepth") == 0) cam->fdepth = parse_float(ts);
        else if (strcmp(w, "flength") == 0) cam->flength = parse_float(ts);
        else if (strcmp(w, "aperture") == 0) cam->aperture = parse_float(ts);
        else if (strcmp(w, "up") == 0) { parse_vec3(ts, cam->up); cam->has_legacy |= 1; }
        else if (strcmp(w, "right") == 0) { parse_vec3(ts, cam->right); cam->has_legacy |= 2; }
        else if (strcmp(w, "forward") == 0) { parse_vec3(ts, cam->forward); cam->has_legacy |= 4; }
        else parse_error(ts, "a camera key");
    }
}

/* parse_material, src/read_scene.c:398-488 */
static void parse_material(token_stream *ts, scene_input *scene)
{
    if (scene->num_materials == scene->cap_materials)
    {
        scene->cap_materials = scene->cap_materials ? scene->cap_materials * 2 : 16;
        scene->materials = (material_input *)realloc(scene->materials, scene->cap_materials * sizeof(material_input));
    }
    material_input *m = &scene->materials[scene->num_materials++];
    memset(m, 0, sizeof(*m));
    char look[256];
    int at_end;
    for (ts_peek(ts, look, &at_end); !at_end && !is_block_word(look); ts_peek(ts, look, &at_end))
    {
        ts_next(ts);
        const char *w = ts->word;
        if (strcmp(w, "name") == 0) parse_word(ts, m->name, sizeof(m->name));
        else if (strcmp(w, "diffuse") == 0) parse_spd_method(ts, &m->diffuse);
        else if (strcmp(w, "glossy") == 0) parse_spd_method(ts, &m->glossy);
        else if (strcmp(w, "emission") == 0) { parse_spd_method(ts, &m->emission); m->is_emissive = 1; }
        else if (strcmp(w, "mirror") == 0) parse_spd_method(ts, &m->mirror);
        else if (strcmp(w, "refract") == 0) parse_spd_method(ts, &m->refract);
        else if (strcmp(w, "extinct") == 0) parse_spd_method(ts, &m->extinct);
        else if (strcmp(w, "is_black_body") == 0 || strcmp(w, "is_blackbody") == 0) m->is_black_body = parse_bool(ts);
        else if (strcmp(w, "shininess") == 0) m->shininess = parse_float(ts);
        else if (strcmp(w, "roughness") == 0) m->roughness = parse_float(ts);
        else if (strcmp(w, "base_material") == 0) m->is_base_material = 1;
        else if (strcmp(w, "escape_material") == 0) m->is_escape_material = 1;
        else if (strcmp(w, "bdsfs") == 0)
        {
            /* parse_bdsfs, src/read_scene.c:308-328: names up to the next key */
            m->has_bdsfs = 1;
            m->num_bdsfs = 0;
            for (ts_peek(ts, look, &at_end); !at_end; ts_peek(ts, look, &at_end))
            {
                int id = lookup_name(look, bdsf_name_list, num_bdsfs_defined);
                if (id < 0) break;
                ts_next(ts);
                if (m->num_bdsfs == DRT_MAX_BDSFS) parse_error(ts, "at most 16 bdsfs");
                m->bdsfs[m->num_bdsfs++] = (u32)id;
            }
            if (m->num_bdsfs == 0) { ts_next(ts); parse_error(ts, "a bdsf name from bdsf_list.h"); }
        }
        else if (strcmp(w, "dir_func") == 0)
        {
            char name[128];
            parse_word(ts, name, sizeof(name));
            int id = lookup_name(name, dir_func_name_list, num_dir_funcs_defined);
            if (id < 0) parse_error(ts, "a dir_func name from bdsf_list.h");
            m->dir_func = (u32)id;
            m->has_dir_func = 1;
        }
        else parse_error(ts, "a material key");
    }
}

/* parse_surface, src/read_scene.c:490-567 */
static void parse_surface(token_stream *ts, scene_input *scene)
{
    if (scene->num_surfaces == scene->cap_surfaces)
    {
        scene->cap_surfaces = scene->cap_surfaces ? scene->cap_surfaces * 2 : 16;
        scene->surfaces = (surface_input *)realloc(scene->surfaces, scene->cap_surfaces * sizeof(surface_input));
    }
    surface_input *s = &scene->surfaces[scene->num_surfaces++];
    memset(s, 0, sizeof(*s));
    char look[256];
    int at_end;
    for (ts_peek(ts, look, &at_end); !at_end && !is_block_word(look); ts_peek(ts, look, &at_end))
    {
        ts_next(ts);
        const char *w = ts->word;
        if (strcmp(w, "name") == 0) parse_word(ts, s->name, sizeof(s->name));
        else if (strcmp(w, "type") == 0)
        {
            ts_next(ts);
            if (strcmp(ts->word, "point") == 0) s->type = DRT_GEO_POINT;
            else if (strcmp(ts->word, "sphere") == 0) s->type = DRT_GEO_SPHERE;
            else if (strcmp(ts->word, "plane") == 0) s->type = DRT_GEO_PLANE;
            else parse_error(ts, "point, sphere or plane");
        }
        else if (strcmp(w, "position") == 0 || strcmp(w, "center") == 0 || strcmp(w, "origin") == 0) parse_vec3(ts, s->position);
        else if (strcmp(w, "radius") == 0) s->radius = parse_float(ts);
        else if (strcmp(w, "pointu") == 0 || strcmp(w, "point_u") == 0) parse_vec3(ts, s->u);
        else if (strcmp(w, "pointv") == 0 || strcmp(w, "point_v") == 0) parse_vec3(ts, s->v);
        else if (strcmp(w, "material") == 0) parse_word(ts, s->material_name, sizeof(s->material_name));
        else parse_error(ts, "a surface key");
    }
}

/* parse_scene, src/read_scene.c:569-602 */
static void parse_scene(const char *text, u32 size, scene_input *scene)
{
    token_stream ts;
    ts_init(&ts, text, size, "scene");
    for (ts_next(&ts); !ts.at_end; ts_next(&ts))
    {
        if (strcmp(ts.word, "Camera") == 0) parse_camera(&ts, &scene->camera);
        else if (strcmp(ts.word, "Material") == 0) parse_material(&ts, scene);
        else if (strcmp(ts.word, "Surface") == 0) parse_surface(&ts, scene);
        else parse_error(&ts, "Camera, Material or Surface");
    }
}

/* ------------------------------------------------------------------------------------------ */
/* parse_config, src/read_scene.c:604-765                                                       */

static void copy_path(char *dst, const char *src)
{
    size_t n = strlen(src);
    if (n > 63) n = 63;
    for (size_t i = 0; i < n; i += 1) dst[i] = (src[i] == '\\') ? '/' : src[i];
    dst[n] = 0;
}

void parse_config(char *config_contents, u32 config_contents_size, config_arguments *config)
{
    token_stream ts;
    ts_init(&ts, config_contents, config_contents_size, "config");
    struct { const char *key; char *dst; } paths[] = {
        {"input_scene", config->input_scene}, {"output_spd", config->output_spd}, {"average_spd", config->average_spd},
        {"variance_spd", config->variance_spd}, {"output_bmp", config->output_bmp}, {"average_bmp", config->average_bmp},
        {"variance_bmp", config->variance_bmp}, {"white_spd", config->white_spd}, {"cmf_x", config->cmf_x},
        {"cmf_y", config->cmf_y}, {"cmf_z", config->cmf_z}, {"red_spd", config->red_spd}, {"green_spd", config->green_spd},
        {"blue_spd", config->blue_spd}, {"cyan_spd", config->cyan_spd}, {"magenta_spd", config->magenta_spd},
        {"yellow_spd", config->yellow_spd}};
    for (ts_next(&ts); !ts.at_end; ts_next(&ts))
    {
        const char *w = ts.word;
        if (strcmp(w, "num_pixel_samples") == 0) config->num_pixel_samples = parse_uint(&ts);
        else if (strcmp(w, "max_cast_depth") == 0) config->max_cast_depth = parse_uint(&ts);
        else if (strcmp(w, "output_width") == 0) config->output_width = parse_uint(&ts);
        else if (strcmp(w, "output_height") == 0) config->output_height = parse_uint(&ts);
        else if (strcmp(w, "min_wl") == 0) config->min_wl = parse_float(&ts);
        else if (strcmp(w, "max_wl") == 0) config->max_wl = parse_float(&ts);
        else if (strcmp(w, "wl_interval") == 0) config->wl_interval = parse_float(&ts);
        else if (strcmp(w, "pixel_scheme") == 0)
        {
            ts_next(&ts);
            if (strcmp(ts.word, "pixel_random") == 0) config->pixel_scheme = FILM_SAMPLE_RANDOM;
            else if (strcmp(ts.word, "pixel_center") == 0) config->pixel_scheme = FILM_SAMPLE_CENTER;
            else parse_error(&ts, "pixel_random or pixel_center");
        }
        else
        {
            int found = 0;
            for (size_t i = 0; i < sizeof(paths) / sizeof(paths[0]); i += 1)
            {
                if (strcmp(w, paths[i].key) == 0)
                {
                    char tmp[256];
                    parse_word(&ts, tmp, sizeof(tmp));
                    copy_path(paths[i].dst, tmp);
                    found = 1;
                    break;
                }
            }
            if (!found) parse_error(&ts, "a config key");
        }
    }
}

void print_config_arguments(config_arguments *config) /* src/win32_platform.c:163-178 */
{
    printf("Num pixel samples:   %u\n", config->num_pixel_samples);
    printf("Output width:        %u\n", config->output_width);
    printf("Output height:       %u\n", config->output_height);
    printf("Min wavelength:      %f\n", config->min_wl);
    printf("Max wavelength:      %f\n", config->max_wl);
    printf("Wavelength interval: %f\n", config->wl_interval);
    printf("Input scene path:    %s\n", config->input_scene);
    printf("Output spd path:     %s\n", config->output_spd);
    printf("Average spd path:    %s\n", config->average_spd);
    printf("Variance spd path:   %s\n", config->variance_spd);
    printf("Output bmp path:     %s\n", config->output_bmp);
    printf("Average bmp path:    %s\n", config->average_bmp);
    printf("Variance bmp path:   %s\n", config->variance_bmp);
}

/* ------------------------------------------------------------------------------------------ */
/* vec3 / mat3 for the camera build (src/geometry.c)                                            */

typedef struct { f64 x, y, z; } v3;
typedef struct { v3 c[3]; } m33;
static v3 V(f64 x, f64 y, f64 z) { v3 r = {x, y, z}; return r; }
static v3 v_sum(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static f64 v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 v_cross(v3 a, v3 b) { return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static v3 v_mul(v3 v, f64 f) { return V(f * v.x, f * v.y, f * v.z); }
static f64 v_length(v3 v) { return sqrt(v_dot(v, v)); }
static v3 v_normalise(v3 v) { f64 l = v_length(v); return V(v.x / l, v.y / l, v.z / l); }
static f64 m_at(const m33 *m, int col, int row) { const v3 *c = &m->c[col]; return row == 0 ? c->x : (row == 1 ? c->y : c->z); }
static v3 m_row(const m33 *m, int r) { return V(m_at(m, 0, r), m_at(m, 1, r), m_at(m, 2, r)); }
static v3 m_vmul(const m33 *m, v3 v) { return V(v_dot(m_row(m, 0), v), v_dot(m_row(m, 1), v), v_dot(m_row(m, 2), v)); }

static m33 rotation_between(v3 v, v3 w) /* src/geometry.c:263-295 */
{
    v3 n = v_cross(v, w);
    f64 c = v_dot(v, w);
    m33 r = {{{1.0, 0.0, 0.0}, {0.0, 1.0, 0.0}, {0.0, 0.0, 1.0}}};
    if (v_dot(n, n) == 0.0 && c <= 0.0)
    {
        r.c[0].x = -1.0; r.c[1].y = -1.0; r.c[2].z = -1.0;
        return r;
    }
    m33 m = {{{0.0, n.z, -n.y}, {-n.z, 0.0, n.x}, {n.y, -n.x, 0.0}}};
    f64 f = 1.0 / (1.0 + c);
    for (int i = 0; i < 3; i += 1)
    {
        /* (m*m) is stored with row/column swapped in the reference's mat3x3_mul; it is symmetric */
        v3 mm = V(v_dot(m_row(&m, i), m.c[0]), v_dot(m_row(&m, i), m.c[1]), v_dot(m_row(&m, i), m.c[2]));
        mm = v_mul(mm, f);
        v3 id = V(i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0);
        r.c[i] = v_sum(v_sum(id, m.c[i]), mm);
    }
    return r;
}

static m33 rotation_about_axis(v3 a, f64 angle_rad) /* src/geometry.c:297-313, as written */
{
    f64 c = cos(angle_rad), s = sin(angle_rad);
    m33 r;
    r.c[0].x = c + (a.x * a.x) * (1 - c);
    r.c[0].y = a.y * a.x * (1 - c) + a.z * s;
    r.c[0].z = a.z * a.z * (1 - c) - a.y * s;
    r.c[1].x = a.x * a.y * (1 - c) - a.z * s;
    r.c[1].y = c + (a.y * a.y) * (1 - c);
    r.c[1].z = a.z * a.y * (1 - c) + a.x * s;
    r.c[2].x = a.x * a.z * (1 - c) + a.y * s;
    r.c[2].y = a.y * a.z * (1 - c) - a.x * s;
    r.c[2].z = c + a.z * a.z * (1 - c);
    return r;
}

static void put3(f64 dst[3], v3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

/* init_camera, src/daily_ray_trace.c:49-77 */
void drt_host_init_camera(drt_camera *camera, const f64 position[3], const f64 target[3], f64 roll, f64 fov,
                          f64 fdepth, f64 flength, f64 aperture, u32 width_px, u32 height_px)
{
    v3 pos = V(position[0], position[1], position[2]);
    v3 tgt = V(target[0], target[1], target[2]);
    v3 ref_forward = {0.0, 0.0, -1.0};
    v3 ref_up = {0.0, 1.0, 0.0};
    v3 forward = v_normalise(v_sub(tgt, pos));
    m33 orient = rotation_between(ref_forward, forward);
    f64 roll_rad = roll * (PI / 180.0);
    m33 rr = rotation_about_axis(forward, roll_rad);
    v3 up = m_vmul(&rr, m_vmul(&orient, ref_up));
    v3 right = v_normalise(v_cross(forward, up));

    f64 aperture_distance = (flength * fdepth) / (flength + fdepth);
    v3 aperture_position = v_sum(pos, v_mul(forward, aperture_distance));
    f64 fov_rad = fov * (PI / 180.0);
    f64 aspect_ratio = (f64)width_px / (f64)height_px;
    f64 film_width = 2.0 * aperture_distance * tan(fov_rad / 2.0);
    f64 film_height = film_width / aspect_ratio;
    v3 film_right = v_mul(right, 0.5 * film_width);
    v3 film_top = v_mul(up, 0.5 * film_height);

    memset(camera, 0, sizeof(*camera));
    put3(camera->forward, forward);
    put3(camera->right, right);
    put3(camera->up, up);
    put3(camera->aperture_position, aperture_position);
    camera->aperture_radius = aperture;
    camera->focal_depth = fdepth;
    camera->focal_length = flength;
    put3(camera->film_bottom_left, v_sub(v_sub(pos, film_right), film_top));
    camera->pixel_width = film_width / (f64)width_px;
    camera->pixel_height = film_height / (f64)height_px;
}

/* ------------------------------------------------------------------------------------------ */
/* Scene build                                                                                  */

struct drt_host_scene
{
    drt_scene     scene;
    drt_camera    camera;
    drt_surface  *surfaces;
    drt_material *materials;
    f64          *spds;
    u32           num_spds, cap_spds, S;
    char        (*material_names)[64];
    char        (*surface_names)[64];
    f64           min_wl, interval;
    char          spectra_dir[256];
};

static f64 *new_spd(drt_host_scene *h, int32_t *index)
{
    if (h->num_spds == h->cap_spds)
    {
        h->cap_spds = h->cap_spds ? h->cap_spds * 2 : 32;
        h->spds = (f64 *)realloc(h->spds, (size_t)h->cap_spds * h->S * sizeof(f64));
    }
    *index = (int32_t)h->num_spds;
    f64 *p = h->spds + (size_t)h->num_spds * h->S;
    memset(p, 0, h->S * sizeof(f64));
    h->num_spds += 1;
    return p;
}

enum { T_RW = 0, T_X, T_Y, T_Z, T_WHITE, T_RED, T_GREEN, T_BLUE, T_CYAN, T_MAGENTA, T_YELLOW, T_COUNT };

/* init_spd, src/daily_ray_trace.c:79-123. Returns -1 for "no spectrum". */
static int32_t init_spd(drt_host_scene *h, const spd_input *in, int *ok)
{
    if (in->method == SPD_METHOD_NONE) return -1;
    int32_t idx;
    f64 *dst = new_spd(h, &idx);
    u32 S = h->S;
    switch (in->method)
    {
        case SPD_METHOD_RGB:
            drt_host_rgb_to_spectrum(h->spds + (size_t)T_WHITE * S, S, in->rgb, dst);
            break;
        case SPD_METHOD_CSV:
        {
            char path[512];
            char name[128];
            copy_path(name, in->csv);
            snprintf(path, sizeof(path), "%s/%s", h->spectra_dir, name);
            if (!drt_host_csv_to_spectrum(path, h->min_wl, h->interval, S, dst))
            {
                set_error("cannot read spectrum csv %s", path);
                *ok = 0;
            }
            break;
        }
        case SPD_METHOD_BLACKBODY:
        {
            drt_host_blackbody_spectrum(h->min_wl, h->interval, S, in->blackbody_temp, dst);
            f64 highest = 0.0; /* spectrum_normalise, src/spectrum.c:182-187 */
            for (u32 i = 0; i < S; i += 1) if (dst[i] > highest) highest = dst[i];
            for (u32 i = 0; i < S; i += 1) dst[i] /= highest;
            break;
        }
        case SPD_METHOD_CONST:
            for (u32 i = 0; i < S; i += 1) dst[i] = in->constant;
            break;
        default: break;
    }
    if (in->has_scale_factor)
        for (u32 i = 0; i < S; i += 1) dst[i] = dst[i] * in->scale_factor;
    return idx;
}

/* Legacy scenes: fill what the old grammar left implicit. */
static void apply_legacy_defaults(scene_input *in)
{
    camera_input *c = &in->camera;
    if (!c->has_target && (c->has_legacy & 4))
    {
        for (int i = 0; i < 3; i += 1) c->target[i] = c->position[i] + c->forward[i];
        c->has_target = 1;
        if (c->has_legacy & 1)
        {
            /* roll = angle between the given `up` and the up init_camera derives at roll 0 */
            drt_camera tmp;
            drt_host_init_camera(&tmp, c->position, c->target, 0.0, 90.0, 1.0, 1.0, 0.0, 1, 1);
            v3 up0 = V(tmp.up[0], tmp.up[1], tmp.up[2]);
            v3 upl = v_normalise(V(c->up[0], c->up[1], c->up[2]));
            v3 fwd = V(tmp.forward[0], tmp.forward[1], tmp.forward[2]);
            f64 cosang = v_dot(up0, upl);
            f64 sinang = v_dot(v_cross(up0, upl), fwd);
            f64 ang = atan2(sinang, cosang) * (f64)(180.0 / PI);
            if (fabs(ang) < 1e-9) ang = 0.0;
            if (fabs(fabs(ang) - 180.0) < 1e-9) ang = 180.0;
            c->roll = ang;
        }
    }
    int have_base = 0, have_escape = 0;
    for (u32 i = 0; i < in->num_materials; i += 1)
    {
        material_input *m = &in->materials[i];
        if (m->is_base_material) have_base = 1;
        if (m->is_escape_material) have_escape = 1;
        if (!m->has_bdsfs && !m->is_black_body && !m->is_escape_material && !m->is_base_material &&
            (m->diffuse.method != SPD_METHOD_NONE || m->glossy.method != SPD_METHOD_NONE))
        {
            m->num_bdsfs = 2;
            m->bdsfs[0] = DRT_BDSF_bp_diffuse_bdsf;
            m->bdsfs[1] = DRT_BDSF_bp_glossy_bdsf;
            m->has_bdsfs = 1;
            if (m->diffuse.method == SPD_METHOD_NONE) { m->diffuse.method = SPD_METHOD_CONST; m->diffuse.constant = 0.0; }
            if (m->glossy.method == SPD_METHOD_NONE) { m->glossy.method = SPD_METHOD_CONST; m->glossy.constant = 0.0; }
        }
        if (m->has_bdsfs && !m->has_dir_func)
        {
            m->dir_func = DRT_DIRF_cos_weighted_sample_hemisphere;
            m->has_dir_func = 1;
        }
    }
    for (int pass = 0; pass < 2; pass += 1)
    {
        if ((pass == 0 && have_base) || (pass == 1 && have_escape)) continue;
        if (in->num_materials == in->cap_materials)
        {
            in->cap_materials = in->cap_materials ? in->cap_materials * 2 : 16;
            in->materials = (material_input *)realloc(in->materials, in->cap_materials * sizeof(material_input));
        }
        material_input *m = &in->materials[in->num_materials++];
        memset(m, 0, sizeof(*m));
        if (pass == 0)
        {
            snprintf(m->name, sizeof(m->name), "vacuum");
            m->refract.method = SPD_METHOD_CONST;
            m->refract.constant = 1.0;
            m->is_base_material = 1;
        }
        else
        {
            snprintf(m->name, sizeof(m->name), "escape");
            m->is_escape_material = 1;
        }
    }
}

static drt_host_scene *build_scene(scene_input *in, const char *spectra_dir, const spd_tables_csvs *tables,
                                   u32 width_px, u32 height_px, f64 min_wl, f64 max_wl, f64 wl_interval)
{
    drt_host_scene *h = (drt_host_scene *)calloc(1, sizeof(*h));
    h->S = (u32)(((max_wl - min_wl) / wl_interval) + 1.0); /* init_spd_tables, src/spectrum.c:3 */
    h->min_wl = min_wl;
    h->interval = wl_interval;
    snprintf(h->spectra_dir, sizeof(h->spectra_dir), "%s", spectra_dir ? spectra_dir : "spectra");
    int ok = 1;

    /* init_spd_tables, src/spectrum.c:36-46: 4 colour-matching + 7 rgb tables */
    char defaults[10][512];
    const char *def_names[10] = {"white_rgb_to_spd.csv", "cmf_x.csv", "cmf_y.csv", "cmf_z.csv", "red_rgb_to_spd.csv",
                                 "green_rgb_to_spd.csv", "blue_rgb_to_spd.csv", "cyan_rgb_to_spd.csv",
                                 "magenta_rgb_to_spd.csv", "yellow_rgb_to_spd.csv"};
    for (int i = 0; i < 10; i += 1) snprintf(defaults[i], sizeof(defaults[i]), "%s/%s", h->spectra_dir, def_names[i]);
    const char *white = tables && tables->white ? tables->white : defaults[0];
    const char *files[T_COUNT];
    files[T_RW] = white;
    files[T_X] = tables && tables->cmf_x ? tables->cmf_x : defaults[1];
    files[T_Y] = tables && tables->cmf_y ? tables->cmf_y : defaults[2];
    files[T_Z] = tables && tables->cmf_z ? tables->cmf_z : defaults[3];
    files[T_WHITE] = white;
    files[T_RED] = tables && tables->rgb_red ? tables->rgb_red : defaults[4];
    files[T_GREEN] = tables && tables->rgb_green ? tables->rgb_green : defaults[5];
    files[T_BLUE] = tables && tables->rgb_blue ? tables->rgb_blue : defaults[6];
    files[T_CYAN] = tables && tables->rgb_cyan ? tables->rgb_cyan : defaults[7];
    files[T_MAGENTA] = tables && tables->rgb_magenta ? tables->rgb_magenta : defaults[8];
    files[T_YELLOW] = tables && tables->rgb_yellow ? tables->rgb_yellow : defaults[9];
    for (int t = 0; t < T_COUNT; t += 1)
    {
        int32_t idx;
        f64 *dst = new_spd(h, &idx);
        if (!drt_host_csv_to_spectrum(files[t], min_wl, wl_interval, h->S, dst))
        {
            set_error("cannot read spectrum table %s", files[t]);
            ok = 0;
        }
    }

    apply_legacy_defaults(in);

    /* init_scene, src/daily_ray_trace.c:125-211: parsed materials + one trailing all-zero material */
    u32 nm = in->num_materials + 1;
    h->materials = (drt_material *)calloc(nm, sizeof(drt_material));
    h->material_names = (char(*)[64])calloc(nm, 64);
    u32 base = 0, escape = 0;
    int have_base = 0, have_escape = 0;
    for (u32 i = 0; i < nm; i += 1)
    {
        drt_material *dst = &h->materials[i];
        dst->emission_spd = dst->diffuse_spd = dst->glossy_spd = dst->mirror_spd = dst->refract_spd = dst->extinct_spd = -1;
        if (i == in->num_materials) break;
        const material_input *m = &in->materials[i];
        snprintf(h->material_names[i], 64, "%s", m->name);
        dst->is_black_body = m->is_escape_material ? 1 : m->is_black_body;
        dst->is_emissive = m->is_emissive;
        dst->shininess = m->shininess;
        dst->roughness = m->roughness;
        dst->dir_func = m->dir_func;
        dst->num_bdsfs = m->num_bdsfs;
        for (u32 j = 0; j < m->num_bdsfs; j += 1) dst->bdsfs[j] = m->bdsfs[j];
        dst->emission_spd = init_spd(h, &m->emission, &ok);
        dst->diffuse_spd = init_spd(h, &m->diffuse, &ok);
        dst->glossy_spd = init_spd(h, &m->glossy, &ok);
        dst->mirror_spd = init_spd(h, &m->mirror, &ok);
        dst->refract_spd = init_spd(h, &m->refract, &ok);
        dst->extinct_spd = init_spd(h, &m->extinct, &ok);
        if (m->is_escape_material) { escape = i; have_escape = 1; }
        if (m->is_base_material) { base = i; have_base = 1; }
    }
    (void)have_base;
    (void)have_escape;

    h->surfaces = (drt_surface *)calloc(in->num_surfaces ? in->num_surfaces : 1, sizeof(drt_surface));
    h->surface_names = (char(*)[64])calloc(in->num_surfaces ? in->num_surfaces : 1, 64);
    for (u32 i = 0; i < in->num_surfaces; i += 1)
    {
        const surface_input *s = &in->surfaces[i];
        drt_surface *dst = &h->surfaces[i];
        snprintf(h->surface_names[i], 64, "%s", s->name);
        dst->type = s->type;
        memcpy(dst->position, s->position, sizeof(dst->position));
        if (s->type == DRT_GEO_SPHERE) dst->radius = s->radius;
        else if (s->type == DRT_GEO_PLANE)
        {
            /* create_plane_from_points, src/geometry.c:203-209 */
            v3 o = V(s->position[0], s->position[1], s->position[2]);
            v3 u = v_sub(V(s->u[0], s->u[1], s->u[2]), o);
            v3 v = v_sub(V(s->v[0], s->v[1], s->v[2]), o);
            v3 n = v_normalise(v_cross(u, v));
            put3(dst->u, u);
            put3(dst->v, v);
            put3(dst->normal, n);
        }
        dst->material = 0; /* unmatched name -> index 0, as the zero-filled table gives in the reference */
        int matched = 0;
        for (u32 j = 0; j < nm; j += 1)
        {
            if (strcmp(s->material_name, h->material_names[j]) == 0)
            {
                dst->material = j;
                matched = 1;
                break;
            }
        }
        if (!matched) fprintf(stderr, "warning: surface %s: unknown material \"%s\", using material 0\n", s->name, s->material_name);
    }

    const camera_input *c = &in->camera;
    drt_host_init_camera(&h->camera, c->position, c->target, c->roll, c->fov, c->fdepth, c->flength, c->aperture, width_px, height_px);

    h->scene.num_surfaces = in->num_surfaces;
    h->scene.surfaces = h->surfaces;
    h->scene.num_materials = nm;
    h->scene.materials = h->materials;
    h->scene.base_material = base;
    h->scene.escape_material = escape;
    h->scene.num_spds = h->num_spds;
    h->scene.num_wavelengths = h->S;
    h->scene.spds = h->spds;
    h->scene.min_wavelength = min_wl;
    h->scene.wavelength_interval = wl_interval;
    h->scene.cmf_rw = T_RW;
    h->scene.cmf_x = T_X;
    h->scene.cmf_y = T_Y;
    h->scene.cmf_z = T_Z;
    if (!ok)
    {
        drt_host_free_scene(h);
        return NULL;
    }
    return h;
}

drt_host_scene *drt_host_load_scene_text(const char *scene_text, u32 scene_size, const char *spectra_dir,
                                         const spd_tables_csvs *tables, u32 width_px, u32 height_px,
                                         f64 min_wl, f64 max_wl, f64 wl_interval)
{
    g_host_error[0] = 0;
    scene_input in;
    memset(&in, 0, sizeof(in));
    parse_scene(scene_text, scene_size, &in);
    drt_host_scene *h = build_scene(&in, spectra_dir, tables, width_px, height_px, min_wl, max_wl, wl_interval);
    free(in.materials);
    free(in.surfaces);
    return h;
}

drt_host_scene *drt_host_load_scene(const char *scene_path, const char *spectra_dir, const spd_tables_csvs *tables,
                                    u32 width_px, u32 height_px, f64 min_wl, f64 max_wl, f64 wl_interval)
{
    char path[512];
    snprintf(path, sizeof(path), "%s", scene_path);
    for (char *c = path; *c; c += 1) if (*c == '\\') *c = '/';
    FILE *f = fopen(path, "rb");
    if (!f)
    {
        set_error("cannot open scene %s", path);
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)calloc((size_t)size + 1, 1);
    size_t got = fread(buf, 1, (size_t)size, f);
    fclose(f);
    drt_host_scene *h = drt_host_load_scene_text(buf, (u32)got, spectra_dir, tables, width_px, height_px, min_wl, max_wl, wl_interval);
    free(buf);
    return h;
}

void drt_host_free_scene(drt_host_scene *h)
{
    if (!h) return;
    free(h->surfaces);
    free(h->materials);
    free(h->spds);
    free(h->material_names);
    free(h->surface_names);
    free(h);
}
const drt_scene *drt_host_scene_data(const drt_host_scene *h) { return &h->scene; }
const drt_camera *drt_host_camera_data(const drt_host_scene *h) { return &h->camera; }
const char *drt_host_material_name(const drt_host_scene *h, u32 i) { return i < h->scene.num_materials ? h->material_names[i] : ""; }
const char *drt_host_surface_name(const drt_host_scene *h, u32 i) { return i < h->scene.num_surfaces ? h->surface_names[i] : ""; }
