/*
 * drt_launcher.hip -- the C-ABI of include/drt_hip.h: scene upload (AoS boundary structs -> SoA
 * device tables), film ownership, batch scheduling of the trace and shade kernels on one HIP
 * stream, HIP-event timing, statistics. gfx950 only; there is no CPU path in this library.
 */
#include "drt_kernels.h"
#include "drt_bvh_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <chrono>
#include <vector>

static thread_local std::string g_last_error;

#define DRT_DEFAULT_MAX_BATCH 256 /* samples per kernel pair when the caller leaves batch_spp = 0 */

/* d_counters, in 8-byte words: [0, DRT_NUM_COUNTERS) the statistics of complete kernel pairs; then the WORK words of the pair in
 * flight -- +0 trace queue, +1 shade queue, +2 bounce queue length, +3 spare, +4 pool cursor (zeroed before every trace launch),
 * +5 overflow flag, +6 sequence number of the last complete pair, +7 peak of the pool cursor -- then, at DRT_PAIR_COUNTERS, the
 * statistics of the pair in flight: the kernels count there, and drt_mark_pair_kernel adds them to the totals only when the pair
 * was complete (a pair whose pool ran out is rendered again, and would be counted twice) */
#define DRT_PAIR_COUNTERS (DRT_NUM_COUNTERS + 8)
#define DRT_COUNTER_WORDS (DRT_PAIR_COUNTERS + DRT_NUM_COUNTERS)

static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                      \
    do                                                                                                     \
    {                                                                                                      \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(-100 - (int)e_, "%s: %s", #expr, hipGetErrorString(e_));         \
    } while (0)

struct drt_context
{
    int         device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;
    drt_params  params{};
    DevScene    dsc{};
    DevCamera   dcam{};
    uint32_t    cmf_rw = 0, cmf_x = 0, cmf_y = 0, cmf_z = 0;
    double      interval = 0.0;

    std::vector<void *> allocations; /* scene tables */
    double *d_pixels = nullptr, *d_avgs = nullptr, *d_vars = nullptr; /* XYZ film mode: d_pixels is [n_pix][XYZ_FILM_WORDS], the others stay null */
    bool    xyz_mode = false;
    bool    own_film = false;
    uint64_t *d_records = nullptr, *d_headers = nullptr;
    uint32_t  light0_em_spd = 0;      /* emission SPD row of the first light (0 when there is none) */
    double   *d_tail_stage = nullptr; /* [n_pix * batch][tail_count]: per-sample results of the shade kernel's tail pass */
    uint32_t  batch_spp = 1;
    uint32_t  vertex_words = 0, vertex_shift = 0, block_words = 0; /* a vertex record, its log2, a pool block (four vertices), in 8-byte words */
    uint64_t  pool_blocks = 0;          /* blocks in d_records */
    uint32_t  worst_blocks_per_path = 0; /* what a path of max_depth vertices takes (table block included) */
    double    est_blocks_per_path = 0.0; /* measured on a sample of the tile when the context is created */
    struct Batch { uint32_t first_sample, n_samples; uint64_t seq; uint32_t row0, rows, hits_sample_offset, stride; }; /* rows == 0: the whole tile */
    std::vector<Batch> inflight;         /* kernel pairs enqueued since the last synchronisation (redone if the pool ran out) */
    uint64_t  next_seq = 1;
    uint64_t  redone_batches = 0, pool_peak = 0;
    int32_t  *d_hits = nullptr;
    uint64_t  hits_capacity = 0; /* in paths */
    uint32_t  hits_samples = 0;
    unsigned long long *d_counters = nullptr; /* DRT_COUNTER_WORDS words: layout at DRT_PAIR_COUNTERS above */
    PrimaryHit *d_primary = nullptr;          /* BVH pipeline: closest hit of every path's camera ray (drt_bvh_kernels.h) */
    uint64_t   *d_queue = nullptr;            /* BVH pipeline: ids of the paths that go on after their first hit */
    bool        bvh_pipeline = false;
    int         primary_grid_cap = 0, bounce_grid_cap = 0;
    double   *d_xyz = nullptr;
    uint8_t  *d_bgra = nullptr;
    bool      trace_tail = false;     /* the trace kernel carries the tail wavelengths of the paths it can (drt_trace_kernel<true, true>): those of plastic
                                         and mirror vertices only, and those without a vertex */
    bool      tail_all_staged = false; /* ... and in this scene that is every path: the shade kernel's tail pass has nothing to replay */
    bool      dark_skip = true;        /* the shade kernel's instantiation that passes over samples worth 0 in pixels nothing has reached yet */
    bool      simple_bdsfs = false;    /* no material lists anything but bp_diffuse_bdsf, bp_glossy_bdsf, mirror_bdsf: the shade kernel without the Fresnel code */
    const double *d_spd_tail = nullptr; /* [n_spd][tail_count]: the SPD table's tail columns */

    bool   scene_in_lds = true, spds_in_lds = true, use_bvh = false;
    size_t trace_lds = 0, shade_lds = 0;
    int    trace_grid_cap = 0, shade_grid_cap = 0;
    uint32_t trace_chunk_override = 0, shade_subs_override = ~0u, tail_period_override = 0; /* DRT_TRACE_CHUNK, DRT_SHADE_SUBS tuning knobs */
    uint32_t shade_sets = 1, tail_first = 0, tail_count = 0;
    uint64_t n_pix = 0;

    std::vector<hipEvent_t> ev; /* triples: trace start, trace end / shade start, shade end */
    std::vector<double> ev_paths; /* per triple: the paths of that kernel pair */
    size_t ev_used = 0;
    double trace_ms = 0.0, shade_ms = 0.0;
    uint64_t timed_pairs = 0; /* per sample pass over the tile (src/daily_ray_trace.c:746-756): min, max, running mean over the pairs */
    double min_sample_ms = 0.0, max_sample_ms = 0.0, avg_sample_ms = 0.0;
};

static size_t trace_lds_bytes(uint32_t n_surf, uint32_t n_lights, uint32_t n_mat)
{
    size_t b = (size_t)SF_COUNT * n_surf * 8 + (size_t)LF_COUNT * n_lights * 8;
    b += ((size_t)(2 * n_surf + 2 * n_lights) * 4 + 7) & ~(size_t)7;
    b += (size_t)n_mat * sizeof(DevMaterial);
    return b;
}

template <typename T>
static int upload(drt_context *ctx, const std::vector<T> &host, const T **dev)
{
    void *p = nullptr;
    size_t bytes = std::max<size_t>(host.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    ctx->allocations.push_back(p);
    if (!host.empty()) HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    *dev = (const T *)p;
    return 0;
}

static V3 hv(const double a[3]) { V3 r; r.x = a[0]; r.y = a[1]; r.z = a[2]; return r; }

/* host-side twins of the device vector ops used for per-surface constants (same IEEE ops) */
static double h_dot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double h_length(const double a[3]) { return std::sqrt(h_dot(a, a)); }

/* ---- BVH over the surfaces of a large scene (host build: median split, padded boxes) ---- */
struct BuildPrim
{
    double   lo[3], hi[3], c[3];
    uint32_t idx;
};

/* Box of everything a surface's intersector can report a hit on. A plane accepts the points j (relative to its
 * `position`) of the plane through it with 0 <= j.u^ <= |u| and 0 <= j.v^ <= |v| (src/geometry.c:157-182): the rectangle
 * position + a u + b v only when u and v are perpendicular; for slanted edge vectors it is the parallelogram whose
 * corners solve [u^; v^; n] j = (a, b, 0) at the four (a, b) extremes -- bounded here by solving exactly that. */
static void prim_bounds(const drt_surface &s, double lo[3], double hi[3])
{
    if (s.type == DRT_GEO_SPHERE)
    {
        for (int k = 0; k < 3; k += 1)
        {
            lo[k] = s.position[k] - std::fabs(s.radius);
            hi[k] = s.position[k] + std::fabs(s.radius);
        }
    }
    else
    {
        const double ul = h_length(s.u), vl = h_length(s.v);
        const double un[3] = {s.u[0] / ul, s.u[1] / ul, s.u[2] / ul}, vn[3] = {s.v[0] / vl, s.v[1] / vl, s.v[2] / vl};
        const double *n = s.normal;
        /* rows of M = u^, v^, n; its inverse by cofactors */
        const double c0[3] = {vn[1] * n[2] - vn[2] * n[1], vn[2] * n[0] - vn[0] * n[2], vn[0] * n[1] - vn[1] * n[0]}; /* v^ x n */
        const double c1[3] = {n[1] * un[2] - n[2] * un[1], n[2] * un[0] - n[0] * un[2], n[0] * un[1] - n[1] * un[0]}; /* n x u^ */
        const double det = un[0] * c0[0] + un[1] * c0[1] + un[2] * c0[2];
        const bool ok = std::isfinite(det) && std::fabs(det) > 1e-6 && std::isfinite(ul) && std::isfinite(vl);
        for (int k = 0; k < 3; k += 1)
        {
            lo[k] = HUGE_VAL;
            hi[k] = -HUGE_VAL;
        }
        for (int corner = 0; corner < 4 && ok; corner += 1)
        {
            const double a = (corner & 1) ? ul : 0.0, b = (corner & 2) ? vl : 0.0;
            for (int k = 0; k < 3; k += 1)
            {
                double j = (a * c0[k] + b * c1[k]) / det; /* M^-1 (a, b, 0) */
                lo[k] = std::min(lo[k], s.position[k] + j);
                hi[k] = std::max(hi[k], s.position[k] + j);
            }
        }
        if (!ok) /* edge vectors (nearly) parallel, or a normal in their span: the accepted region is unbounded */
            for (int k = 0; k < 3; k += 1)
            {
                lo[k] = -1e300;
                hi[k] = 1e300;
            }
    }
    for (int k = 0; k < 3; k += 1)
    {
        /* pad: a hit the intersector COMPUTES (rounding included) must stay inside the box */
        double pad = 1e-5 + 1e-9 * std::max(std::fabs(lo[k]), std::fabs(hi[k]));
        lo[k] -= pad;
        hi[k] += pad;
    }
}

struct BvhBuilder
{
    std::vector<BuildPrim> prims;
    std::vector<BvhNode>   nodes;
    std::vector<uint32_t>  order;
    int LEAF = 1; /* surfaces per leaf: ONE, which is what the kernels' leaf steps are written for (the reference packs a count <= 8 in 3 bits).
                     Config 5, trace stage: 757 ms with 1, 855 with 2, 948 with 4 */
    int max_depth = 0;    /* deepest level holding a node: the traversal pushes at most one entry per level */
    double pad32 = 0.0;   /* extra padding of the STORED boxes that pays for testing them in f32 (drt_kernels.h, Ray32) */
    double extent = 0.0;  /* largest |coordinate| of the surfaces' boxes and of the camera */

    void bounds(size_t b, size_t e, double lo[3], double hi[3]) const
    {
        for (int k = 0; k < 3; k += 1) { lo[k] = HUGE_VAL; hi[k] = -HUGE_VAL; }
        for (size_t i = b; i < e; i += 1)
            for (int k = 0; k < 3; k += 1)
            {
                lo[k] = std::min(lo[k], prims[i].lo[k]);
                hi[k] = std::max(hi[k], prims[i].hi[k]);
            }
    }
    /* fills child slot c of node `parent` with the subtree over prims [b, e) */
    void set_child(int parent, int c, size_t b, size_t e, int depth)
    {
        double lo[3], hi[3];
        bounds(b, e, lo, hi);
        max_depth = std::max(max_depth, depth + 1);
        for (int k = 0; k < 3; k += 1)
        {
            lo[k] -= pad32;
            hi[k] += pad32;
            /* outward to f32: the stored box must contain the (already padded) f64 box */
            float fl = (float)lo[k], fh = (float)hi[k];
            if ((double)fl > lo[k]) fl = std::nextafterf(fl, -INFINITY);
            if ((double)fh < hi[k]) fh = std::nextafterf(fh, INFINITY);
            nodes[parent].lo[c][k] = fl;
            nodes[parent].hi[c][k] = fh;
        }
        if (e - b <= (size_t)LEAF)
        {
            nodes[parent].child[c] = -2 - ((int32_t)order.size() * 8 + ((int32_t)(e - b) - 1)); /* bvh_leaf_ref(first slot, count) */
            nodes[parent].count[c] = (int32_t)(e - b);
            for (size_t i = b; i < e; i += 1) order.push_back(prims[i].idx);
            return;
        }
        int me = (int)nodes.size();
        nodes.push_back(BvhNode());
        nodes[parent].child[c] = me;
        nodes[parent].count[c] = 0;
        split(me, b, e, depth + 1);
    }
    static double half_area(const double lo[3], const double hi[3])
    {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
    /* Binned surface-area heuristic (32 bins per axis, all three axes): the split that minimises
     * area(left)*n(left) + area(right)*n(right); falls back to the median along the widest axis when the
     * centroids do not separate. Only the amount of pruning depends on this, never a result. */
    void split(int node, size_t b, size_t e, int depth)
    {
        const int BINS = 32;
        double clo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, chi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
        for (size_t i = b; i < e; i += 1)
            for (int k = 0; k < 3; k += 1)
            {
                clo[k] = std::min(clo[k], prims[i].c[k]);
                chi[k] = std::max(chi[k], prims[i].c[k]);
            }
        int best_axis = -1, best_bin = -1;
        double best_cost = HUGE_VAL;
        /* SAH trees have no depth bound of their own, and the traversal stacks hold BVH_STACK entries, one per level at most.
         * A median-split subtree over m surfaces is at most 1 + ceil(log2 m) levels deep, so the SAH may split this node only
         * while a median-split subtree below its children would still fit. */
        int log2m = 0;
        while (((size_t)1 << log2m) < e - b) log2m += 1;
        if (!getenv("DRT_BVH_MEDIAN") && depth + 2 + log2m < BVH_STACK)
            for (int axis = 0; axis < 3; axis += 1)
            {
                double ext = chi[axis] - clo[axis];
                if (!(ext > 0.0)) continue;
                double blo[BINS][3], bhi[BINS][3];
                size_t cnt[BINS];
                for (int k = 0; k < BINS; k += 1)
                {
                    cnt[k] = 0;
                    for (int a = 0; a < 3; a += 1) { blo[k][a] = HUGE_VAL; bhi[k][a] = -HUGE_VAL; }
                }
                for (size_t i = b; i < e; i += 1)
                {
                    int k = std::min(BINS - 1, (int)((prims[i].c[axis] - clo[axis]) / ext * BINS));
                    cnt[k] += 1;
                    for (int a = 0; a < 3; a += 1)
                    {
                        blo[k][a] = std::min(blo[k][a], prims[i].lo[a]);
                        bhi[k][a] = std::max(bhi[k][a], prims[i].hi[a]);
                    }
                }
                /* sweep: right-side areas from the back, then left side from the front */
                double r_area[BINS];
                size_t r_cnt[BINS];
                double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
                size_t n = 0;
                for (int k = BINS - 1; k > 0; k -= 1)
                {
                    if (cnt[k]) for (int a = 0; a < 3; a += 1) { lo[a] = std::min(lo[a], blo[k][a]); hi[a] = std::max(hi[a], bhi[k][a]); }
                    n += cnt[k];
                    r_area[k] = n ? half_area(lo, hi) : 0.0;
                    r_cnt[k] = n;
                }
                for (int a = 0; a < 3; a += 1) { lo[a] = HUGE_VAL; hi[a] = -HUGE_VAL; }
                n = 0;
                for (int k = 0; k + 1 < BINS; k += 1) /* split after bin k */
                {
                    if (cnt[k]) for (int a = 0; a < 3; a += 1) { lo[a] = std::min(lo[a], blo[k][a]); hi[a] = std::max(hi[a], bhi[k][a]); }
                    n += cnt[k];
                    if (n == 0 || r_cnt[k + 1] == 0) continue;
                    double cost = half_area(lo, hi) * (double)n + r_area[k + 1] * (double)r_cnt[k + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = k; }
                }
            }
        size_t mid;
        if (best_axis >= 0)
        {
            const int axis = best_axis, bin = best_bin;
            const double lo0 = clo[axis], ext = chi[axis] - clo[axis];
            auto it = std::stable_partition(prims.begin() + b, prims.begin() + e, [=](const BuildPrim &x) {
                return std::min(BINS - 1, (int)((x.c[axis] - lo0) / ext * BINS)) <= bin;
            });
            mid = (size_t)(it - prims.begin());
        }
        else
        {
            int axis = 0;
            for (int k = 1; k < 3; k += 1) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
            mid = (b + e) / 2;
            std::nth_element(prims.begin() + b, prims.begin() + mid, prims.begin() + e,
                             [axis](const BuildPrim &x, const BuildPrim &y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.idx < y.idx); });
        }
        set_child(node, 0, b, mid, depth);
        set_child(node, 1, mid, e, depth);
    }
    /* `reach`: the largest |coordinate| a ray origin outside the surfaces can have (the camera) */
    void build(const drt_scene *scene, double reach)
    {
        extent = reach;
        for (uint32_t i = 0; i < scene->num_surfaces; i += 1)
        {
            const drt_surface &s = scene->surfaces[i];
            if (s.type != DRT_GEO_SPHERE && s.type != DRT_GEO_PLANE) continue; /* points are never intersected */
            BuildPrim p;
            prim_bounds(s, p.lo, p.hi);
            for (int k = 0; k < 3; k += 1)
            {
                p.c[k] = 0.5 * (p.lo[k] + p.hi[k]);
                if (std::fabs(p.lo[k]) < 1e299) extent = std::max(extent, std::fabs(p.lo[k]));
                if (std::fabs(p.hi[k]) < 1e299) extent = std::max(extent, std::fabs(p.hi[k]));
            }
            p.idx = i;
            prims.push_back(p);
        }
        /* the f32 box test moves a slab plane by < 6 * 2^-24 * extent (drt_kernels.h, Ray32): pad by 2^-19 * extent, 5x that */
        pad32 = std::ldexp(extent, -19);
        nodes.push_back(BvhNode());
        nodes[0].count[0] = nodes[0].count[1] = -1;
        nodes[0].child[0] = nodes[0].child[1] = BVH_DONE; /* no child */
        if (prims.empty()) return;
        if (prims.size() <= (size_t)LEAF) set_child(0, 0, 0, prims.size(), 0);
        else split(0, 0, prims.size(), 0);
    }
};

static void shade_sets(uint32_t S, uint32_t *n_sets, uint32_t *tail_first, uint32_t *tail_count);

static int build_device_scene(drt_context *ctx, const drt_scene *scene, double reach)
{
    const uint32_t S = scene->num_wavelengths;
    const uint32_t n_surf = scene->num_surfaces;
    if (S == 0 || scene->num_spds == 0 || !scene->spds) return fail(-2, "scene has no spectral tables");
    if (scene->base_material >= scene->num_materials || scene->escape_material >= scene->num_materials)
        return fail(-2, "base/escape material index out of range");

    DevScene &d = ctx->dsc;
    d.n_surf = n_surf;
    d.n_mat = scene->num_materials;
    d.S = S;
    d.n_spd = scene->num_spds;
    d.base_mat = scene->base_material;
    d.escape_mat = scene->escape_material;
    /* value_at_wl(., trans_wl = 630), src/spectrum.c:150-162 and src/daily_ray_trace.c:381 */
    d.trans_wl = 630.0;
    d.trans_i0 = (uint32_t)((d.trans_wl - scene->min_wavelength) / scene->wavelength_interval);
    if (d.trans_i0 + 1 >= S) return fail(-2, "wavelength grid does not bracket trans_wl = 630 nm");
    d.trans_w0 = scene->min_wavelength + d.trans_i0 * scene->wavelength_interval;
    d.trans_w1 = scene->min_wavelength + (d.trans_i0 + 1) * scene->wavelength_interval;

    std::vector<double> surf((size_t)SF_COUNT * n_surf, 0.0);
    std::vector<uint32_t> stype(n_surf), smat(n_surf);
    std::vector<double> lights;
    std::vector<uint32_t> ltype, lmat;
    std::vector<uint32_t> light_surfaces;
    for (uint32_t i = 0; i < n_surf; i += 1)
    {
        const drt_surface &s = scene->surfaces[i];
        if (s.material >= scene->num_materials) return fail(-2, "surface %u: material index out of range", i);
        stype[i] = s.type;
        smat[i] = s.material;
        surf[(size_t)SF_PX * n_surf + i] = s.position[0];
        surf[(size_t)SF_PY * n_surf + i] = s.position[1];
        surf[(size_t)SF_PZ * n_surf + i] = s.position[2];
        surf[(size_t)SF_RADIUS * n_surf + i] = s.radius;
        if (s.type == DRT_GEO_PLANE)
        {
            /* |u|, |v|, u/|u|, v/|v|: what line_plane_intersection recomputes per ray (src/geometry.c:166-170) */
            double ul = h_length(s.u), vl = h_length(s.v);
            surf[(size_t)SF_NX * n_surf + i] = s.normal[0];
            surf[(size_t)SF_NY * n_surf + i] = s.normal[1];
            surf[(size_t)SF_NZ * n_surf + i] = s.normal[2];
            for (int k = 0; k < 3; k += 1)
            {
                surf[(size_t)(SF_UNX + k) * n_surf + i] = s.u[k] / ul;
                surf[(size_t)(SF_VNX + k) * n_surf + i] = s.v[k] / vl;
            }
            surf[(size_t)SF_ULEN * n_surf + i] = ul;
            surf[(size_t)SF_VLEN * n_surf + i] = vl;
        }
        if (scene->materials[s.material].is_emissive) light_surfaces.push_back(i);
    }
    const uint32_t n_lights = (uint32_t)light_surfaces.size();
    d.n_lights = n_lights;
    lights.assign((size_t)LF_COUNT * std::max<uint32_t>(n_lights, 1), 0.0);
    ltype.resize(n_lights);
    lmat.resize(n_lights);
    for (uint32_t l = 0; l < n_lights; l += 1)
    {
        const drt_surface &s = scene->surfaces[light_surfaces[l]];
        ltype[l] = s.type;
        lmat[l] = s.material;
        for (int k = 0; k < 3; k += 1)
        {
            lights[(size_t)(LF_PX + k) * n_lights + l] = s.position[k];
            lights[(size_t)(LF_UX + k) * n_lights + l] = s.u[k];
            lights[(size_t)(LF_VX + k) * n_lights + l] = s.v[k];
        }
        lights[(size_t)LF_RADIUS * n_lights + l] = s.radius;
        double pdf = 1.0; /* src/daily_ray_trace.c:292, :304, :314 */
        if (s.type == DRT_GEO_SPHERE) pdf = ((4.0 * DRT_PI) * s.radius) * s.radius;
        else if (s.type == DRT_GEO_PLANE)
        {
            double c[3] = {s.u[1] * s.v[2] - s.u[2] * s.v[1], s.u[2] * s.v[0] - s.u[0] * s.v[2], s.u[0] * s.v[1] - s.u[1] * s.v[0]};
            pdf = h_length(c);
        }
        lights[(size_t)LF_PDF * n_lights + l] = pdf;
    }

    std::vector<double> spds(scene->spds, scene->spds + (size_t)scene->num_spds * S);
    std::map<int32_t, uint32_t> diffuse_pi_row;
    /* the zero row is appended last; derived rows go between: reserve its index now */
    uint32_t n_diffuse = 0;
    {
        std::map<int32_t, int> seen;
        for (uint32_t i = 0; i < scene->num_materials; i += 1)
            if (scene->materials[i].diffuse_spd >= 0 && !seen[scene->materials[i].diffuse_spd]++) n_diffuse += 1;
    }
    /* Rows tabulated per pair of media for the Fresnel terms (drt_device.h, *_reflectance_sel; drt_kernels.h, PAIR_*): a material
     * whose list holds dielectric Fresnel functions gets one row (rel_sq) for rays entering it from the base material and one for
     * rays leaving it; one that holds conductor functions two rows (cA, cB) for rays arriving from the base material. A list
     * with both kinds gets none (the shade kernel tells the kind from the pair field, not per function). */
    auto fresnel_kind = [&](const drt_material &m) -> int { /* 0 none or mixed, 1 dielectric, 2 conductor */
        bool d = false, c = false;
        for (uint32_t j = 0; j < std::min<uint32_t>(m.num_bdsfs, DRT_MAX_BDSFS); j += 1)
        {
            const uint32_t b = m.bdsfs[j];
            d = d || b == DRT_BDSF_fs_dielectric_reflectance_bdsf || b == DRT_BDSF_fs_dielectric_transmittance_bdsf;
            c = c || b == DRT_BDSF_fs_conductor_bdsf || b == DRT_BDSF_ct_conductor_bdsf;
        }
        if (getenv("DRT_NO_PAIR_ROWS")) return 0; /* A/B knob of the parity tests: every term from ir, tr, te */
        return (d && !c) ? 1 : (c && !d) ? 2 : 0;
    };
    uint32_t n_pair_rows = 0;
    for (uint32_t i = 0; i < scene->num_materials; i += 1) n_pair_rows += 2u * (uint32_t)(fresnel_kind(scene->materials[i]) != 0);
    const uint32_t zero_row = scene->num_spds + n_diffuse + n_pair_rows;
    if (zero_row >= PAIR_CONDUCTOR) n_pair_rows = 0; /* row numbers must fit below the kind bit: (never with real scenes) no pair rows then */
    std::vector<DevMaterial> mats(scene->num_materials);
    for (uint32_t i = 0; i < scene->num_materials; i += 1)
    {
        const drt_material &m = scene->materials[i];
        DevMaterial &dm = mats[i];
        memset(&dm, 0, sizeof(dm));
        dm.is_black_body = m.is_black_body;
        dm.is_emissive = m.is_emissive;
        dm.num_bdsfs = std::min<uint32_t>(m.num_bdsfs, DRT_MAX_BDSFS);
        dm.dir_func = m.dir_func;
        const int32_t idx[6] = {m.emission_spd, m.diffuse_spd, m.glossy_spd, m.mirror_spd, m.refract_spd, m.extinct_spd};
        for (int k = 0; k < 6; k += 1)
            if (idx[k] >= (int32_t)scene->num_spds) return fail(-2, "material %u: SPD index out of range", i);
        /* device rows: scene rows as they are; a missing spectrum -> the all-zero row; diffuse -> its derived
         * diffuse * (1/PI) row (bp_diffuse_bdsf's first product, src/bdsf.c:107, is a per-material constant) */
        auto row = [&](int32_t i) { return i >= 0 ? i : (int32_t)zero_row; };
        dm.emission_spd = row(m.emission_spd); dm.glossy_spd = row(m.glossy_spd);
        dm.mirror_spd = row(m.mirror_spd); dm.refract_spd = row(m.refract_spd); dm.extinct_spd = row(m.extinct_spd);
        dm.diffuse_spd = (int32_t)zero_row;
        if (m.diffuse_spd >= 0)
        {
            auto it = diffuse_pi_row.find(m.diffuse_spd);
            if (it == diffuse_pi_row.end())
            {
                const double inv_pi = 1.0 / DRT_PI;
                uint32_t r = (uint32_t)(spds.size() / S);
                for (uint32_t k = 0; k < S; k += 1) spds.push_back(scene->spds[(size_t)m.diffuse_spd * S + k] * inv_pi);
                it = diffuse_pi_row.emplace(m.diffuse_spd, r).first;
            }
            dm.diffuse_spd = (int32_t)it->second;
        }
        dm.shininess = m.shininess;
        dm.roughness = m.roughness;
        if (m.refract_spd >= 0)
        {
            dm.refract_i0 = scene->spds[(size_t)m.refract_spd * S + d.trans_i0];
            dm.refract_i1 = scene->spds[(size_t)m.refract_spd * S + d.trans_i0 + 1];
        }
        for (uint32_t j = 0; j < dm.num_bdsfs; j += 1)
        {
            uint32_t b = m.bdsfs[j];
            if (b >= DRT_NUM_BDSFS) return fail(-2, "material %u: unknown bdsf id %u", i, b);
            dm.bdsfs[j] = b;
            dm.bdsf_packed |= (uint64_t)b << (4 * j);
            if (b == DRT_BDSF_bp_glossy_bdsf) dm.needs |= NEED_GLOSSY;
            if (b == DRT_BDSF_mirror_bdsf || b == DRT_BDSF_fs_conductor_bdsf || b == DRT_BDSF_fs_dielectric_reflectance_bdsf) dm.needs |= NEED_EQR;
            if (b == DRT_BDSF_fs_dielectric_transmittance_bdsf) dm.needs |= NEED_EQT;
            if (b == DRT_BDSF_ct_conductor_bdsf) dm.needs |= NEED_CT;
        }
        if (dm.num_bdsfs == 2 && dm.bdsfs[0] == DRT_BDSF_bp_diffuse_bdsf && dm.bdsfs[1] == DRT_BDSF_bp_glossy_bdsf) dm.vertex_flags = FLAG_PLASTIC;
        dm.pair_out = dm.pair_in = PAIR_NONE;
        const int kind = n_pair_rows ? fresnel_kind(m) : 0;
        if (kind)
        {
            /* a spectrum that is not given reads as zeros, as everywhere (the table's all-zero row) */
            const drt_material &bm = scene->materials[scene->base_material];
            auto at = [&](int32_t spd, uint32_t k) { return spd >= 0 ? scene->spds[(size_t)spd * S + k] : 0.0; };
            const uint32_t r0 = (uint32_t)(spds.size() / S);
            if (kind == 1)
            {
                /* rel_sq = (ir / tr) (ir / tr), src/bdsf.c:52-56: entering (ir the base material's, tr this one's), then leaving */
                for (uint32_t k = 0; k < S; k += 1) { const double rel = at(bm.refract_spd, k) / at(m.refract_spd, k); spds.push_back(rel * rel); }
                for (uint32_t k = 0; k < S; k += 1) { const double rel = at(m.refract_spd, k) / at(bm.refract_spd, k); spds.push_back(rel * rel); }
                dm.pair_out = (uint16_t)r0;
                dm.pair_in = (uint16_t)(r0 + 1u);
            }
            else
            {
                /* cA = rr_sq - re_sq, cB = 4 rr_sq re_sq with rr = tr / ir, re = te / ir, src/bdsf.c:84-91 */
                std::vector<double> cB(S);
                for (uint32_t k = 0; k < S; k += 1)
                {
                    const double ir = at(bm.refract_spd, k), rr = at(m.refract_spd, k) / ir, re = at(m.extinct_spd, k) / ir;
                    const double rr_sq = rr * rr, re_sq = re * re;
                    spds.push_back(rr_sq - re_sq);
                    cB[k] = 4.0 * rr_sq * re_sq;
                }
                spds.insert(spds.end(), cB.begin(), cB.end());
                dm.pair_out = (uint16_t)(r0 | PAIR_CONDUCTOR);
            }
        }
        if (!m.is_black_body && dm.dir_func >= DRT_NUM_DIRFS) return fail(-2, "material %u: unknown dir_func id %u", i, dm.dir_func);
    }
    spds.resize((size_t)(zero_row + 1) * S, 0.0); /* + the all-zero row */
    d.n_spd = zero_row + 1;

    int rc;
    if ((rc = upload(ctx, surf, &d.surf))) return rc;
    if ((rc = upload(ctx, stype, &d.surf_type))) return rc;
    if ((rc = upload(ctx, smat, &d.surf_mat))) return rc;
    if ((rc = upload(ctx, lights, &d.lights))) return rc;
    if ((rc = upload(ctx, ltype, &d.light_type))) return rc;
    if ((rc = upload(ctx, lmat, &d.light_mat))) return rc;
    ctx->light0_em_spd = n_lights ? ((uint32_t)mats[lmat[0]].emission_spd & 0xFFFFu) : 0u; /* what the trace kernel writes into light 0's blocks */
    if ((rc = upload(ctx, mats, &d.mats))) return rc;
    if ((rc = upload(ctx, spds, &d.spds))) return rc;

    ctx->cmf_rw = scene->cmf_rw; ctx->cmf_x = scene->cmf_x; ctx->cmf_y = scene->cmf_y; ctx->cmf_z = scene->cmf_z;
    ctx->interval = scene->wavelength_interval;
    if (std::max(std::max(ctx->cmf_rw, ctx->cmf_x), std::max(ctx->cmf_y, ctx->cmf_z)) >= scene->num_spds)
        return fail(-2, "colour-matching SPD index out of range");

    /* LDS budgets: keep the scene in LDS when it leaves room for >= 2 workgroups per CU */
    ctx->trace_lds = trace_lds_bytes(n_surf, n_lights, scene->num_materials);
    /* small scenes: whole scan out of LDS; large scenes: tables stay in HBM/L2 and a BVH prunes the scan */
    ctx->scene_in_lds = ctx->trace_lds <= 64 * 1024 && n_surf <= 96 && !getenv("DRT_FORCE_BVH");
    ctx->use_bvh = !ctx->scene_in_lds; /* (round 1's brute-force scan from HBM, DRT_NO_BVH, is gone: the two BVH kernels are the only path for such scenes) */
    d.bvh_nodes = nullptr;
    d.bvh_leaf = nullptr;
    if (ctx->use_bvh)
    {
        BvhBuilder bb;
        bb.build(scene, reach);
        /* the traversal stacks hold BVH_STACK entries, one per level at most, and push unchecked */
        if (bb.max_depth > BVH_STACK) return fail(-2, "BVH of %zu surfaces is %d levels deep, the traversal stack holds %d", bb.prims.size(), bb.max_depth, BVH_STACK);
        /* the f32 box test multiplies coordinates by 1/d capped at 2^100 (drt_kernels.h, bvh_inv32) */
        if (!(bb.extent < 134217728.0)) return fail(-2, "scene or camera coordinates reach %g: the hierarchy's f32 box test holds up to 2^27", bb.extent);
        std::vector<BvhLeafPrim> leaf(std::max<size_t>(bb.order.size(), 1));
        memset(leaf.data(), 0, leaf.size() * sizeof(BvhLeafPrim));
        for (size_t k = 0; k < bb.order.size(); k += 1)
        {
            const uint32_t i = bb.order[k];
            leaf[k].index = i;
            leaf[k].type = stype[i];
            for (int f = 0; f < 4; f += 1) leaf[k].f[f] = surf[(size_t)f * n_surf + i]; /* SF_PX, SF_PY, SF_PZ, SF_RADIUS */
            leaf[k].reach32 = INFINITY; /* planes: never "certainly missed" */
            if (stype[i] == DRT_GEO_SPHERE)
            {
                for (int f = 0; f < 3; f += 1) leaf[k].c32[f] = (float)leaf[k].f[f];
                /* radius + 64 u E (drt_kernels.h, sphere_certainly_missed), rounded up twice over */
                float reach = (float)(std::fabs(leaf[k].f[3]) + std::ldexp(bb.extent, -18));
                leaf[k].reach32 = std::nextafterf(std::nextafterf(reach, INFINITY), INFINITY);
            }
        }
        if ((rc = upload(ctx, bb.nodes, &d.bvh_nodes))) return rc;
        if ((rc = upload(ctx, leaf, &d.bvh_leaf))) return rc;
    }
    if (!ctx->scene_in_lds) ctx->trace_lds = 0;
    ctx->spds_in_lds = (size_t)d.n_spd * S * 8 <= 64 * 1024;
    /* The tail wavelengths in the trace kernel (csrc/drt_kernels.h, drt_trace_kernel<true, true>): scenes scanned out of LDS with one
     * light and a tail of at most 8 wavelengths, when the table's tail columns and the waves' running values ([2 R][64 lanes] each)
     * fit beside the scene. Paths of two-lobe plastic and mirror vertices (and those without a vertex) are carried there; where
     * every surface is a light, a black body, plastic or a mirror that is EVERY path, and the shade kernel's tail pass has nothing
     * to replay (tail_all_staged). With glass or gold in the scene the paths that touch them stay with the tail pass, as tasks. */
    {
        uint32_t sets = 0, tf = 0, tc = 0;
        shade_sets(S, &sets, &tf, &tc);
        const size_t extra = (size_t)d.n_spd * tc * 8 + (size_t)(TRACE_BLOCK / 64) * 2 * tc * 64 * 8;
        /* every list of every material (used by a surface or not: a ray can only meet a surface's, but the check is cheap) */
        ctx->simple_bdsfs = !getenv("DRT_NO_SIMPLE_SHADE");
        for (uint32_t i = 0; i < scene->num_materials; i += 1)
            for (uint32_t j = 0; j < mats[i].num_bdsfs; j += 1)
            {
                const uint32_t b = mats[i].bdsfs[j];
                if (b != DRT_BDSF_bp_diffuse_bdsf && b != DRT_BDSF_bp_glossy_bdsf && b != DRT_BDSF_mirror_bdsf) ctx->simple_bdsfs = false;
            }
        bool all_simple = true;
        for (uint32_t i = 0; i < n_surf; i += 1)
        {
            const DevMaterial &m = mats[smat[i]];
            const bool mirror_only = m.num_bdsfs == 1u && m.bdsfs[0] == DRT_BDSF_mirror_bdsf;
            if (!(m.is_black_body || (m.vertex_flags & FLAG_PLASTIC) || mirror_only)) all_simple = false;
        }
        /* DRT_TRACE_TAIL: 0 = never (every path's tail through the shade kernel's tail pass), 2 = only in scenes where every path can be
         * carried (round 2's rule); default: whenever the kernel can run -- paths that meet glass or gold stay with the tail pass, which
         * takes them as tasks while the others, three quarters and more, cost it nothing */
        const char *e = getenv("DRT_TRACE_TAIL");
        ctx->trace_tail = ctx->scene_in_lds && sets == 1 && tc > 0 && tc <= 8 && n_lights == 1 && ctx->trace_lds + extra <= 48 * 1024 && !(e && *e == '0') &&
                          (all_simple || !(e && *e == '2'));
        ctx->tail_all_staged = ctx->trace_tail && all_simple;
        if (ctx->trace_tail)
        {
            std::vector<double> cols((size_t)d.n_spd * tc);
            for (uint32_t r = 0; r < d.n_spd; r += 1)
                for (uint32_t j = 0; j < tc; j += 1) cols[(size_t)r * tc + j] = spds[(size_t)r * S + tf + j];
            if ((rc = upload(ctx, cols, &ctx->d_spd_tail))) return rc;
            ctx->trace_lds += extra;
        }
    }
    return 0;
}

static bool shade_simple(const drt_context *ctx) { return ctx->simple_bdsfs && ctx->spds_in_lds && ctx->shade_sets == 1; }

template <int NSETS, bool XYZ>
static int launch_shade_mode(drt_context *ctx, uint32_t grid, const ShadeParams &sp, double *const film[3])
{
#define DRT_LAUNCH_SHADE(LDS, DARK, SIMPLE)                                                                                                       \
    hipLaunchKernelGGL((drt_shade_kernel<NSETS, LDS, XYZ, DARK, SIMPLE>), dim3(grid), dim3(SHADE_BLOCK), ctx->shade_lds, ctx->stream, ctx->dsc, sp, \
                       ctx->d_records, ctx->d_headers, film[0], film[1], film[2], ctx->d_counters + DRT_NUM_COUNTERS + 1)
    /* SIMPLE: scenes without a Fresnel function, one wavelength set per lane, tables in LDS (shade_simple()) */
    if (NSETS == 1 && shade_simple(ctx))
    {
        if (ctx->dark_skip) DRT_LAUNCH_SHADE(true, true, (NSETS == 1));
        else DRT_LAUNCH_SHADE(true, false, (NSETS == 1));
    }
    else if (ctx->spds_in_lds && ctx->dark_skip) DRT_LAUNCH_SHADE(true, true, false);
    else if (ctx->spds_in_lds) DRT_LAUNCH_SHADE(true, false, false);
    else if (ctx->dark_skip) DRT_LAUNCH_SHADE(false, true, false);
    else DRT_LAUNCH_SHADE(false, false, false);
#undef DRT_LAUNCH_SHADE
    return 0;
}

template <int NSETS>
static int launch_shade_sets(drt_context *ctx, uint32_t grid, const ShadeParams &sp, double *const film[3])
{
    return ctx->xyz_mode ? launch_shade_mode<NSETS, true>(ctx, grid, sp, film) : launch_shade_mode<NSETS, false>(ctx, grid, sp, film);
}

/* How the S wavelengths map to lanes: n full 64-lane sets, plus (when the remainder is small) a packed tail pass */
static void shade_sets(uint32_t S, uint32_t *n_sets, uint32_t *tail_first, uint32_t *tail_count)
{
    uint32_t full = S / 64, rem = S % 64;
    *tail_first = 0;
    *tail_count = 0;
    if (S <= 64 || rem == 0) *n_sets = (S + 63) / 64;
    else if (rem <= 16 && !getenv("DRT_NO_TAIL_PASS"))
    {
        *n_sets = full;
        *tail_first = 64 * full;
        *tail_count = rem;
    }
    else *n_sets = full + 1;
}

/* film: the three buffers from the first pixel the launch covers on (a launch may cover a range of the tile's rows) */
static int launch_shade(drt_context *ctx, uint32_t grid, const ShadeParams &sp, double *const film[3])
{
    switch (ctx->shade_sets)
    {
        case 1: return launch_shade_sets<1>(ctx, grid, sp, film);
        case 2: return launch_shade_sets<2>(ctx, grid, sp, film);
        case 3: return launch_shade_sets<3>(ctx, grid, sp, film);
        default: return launch_shade_sets<4>(ctx, grid, sp, film);
    }
}

template <int NSETS, bool XYZ>
static int shade_occupancy_mode(drt_context *ctx, int *per_cu)
{
    /* (the DARK instantiations have the same registers and LDS) */
    if (NSETS == 1 && shade_simple(ctx))
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, drt_shade_kernel<NSETS, true, XYZ, true, (NSETS == 1)>, SHADE_BLOCK, ctx->shade_lds));
    else if (ctx->spds_in_lds)
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, drt_shade_kernel<NSETS, true, XYZ, true>, SHADE_BLOCK, ctx->shade_lds));
    else
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, drt_shade_kernel<NSETS, false, XYZ, true>, SHADE_BLOCK, ctx->shade_lds));
    return 0;
}

template <int NSETS>
static int shade_occupancy(drt_context *ctx, int *per_cu)
{
    return ctx->xyz_mode ? shade_occupancy_mode<NSETS, true>(ctx, per_cu) : shade_occupancy_mode<NSETS, false>(ctx, per_cu);
}

extern "C" const char *drt_last_error(void) { return g_last_error.c_str(); }

/* Host only (no HIP call): builds the hierarchy build_device_scene() would build and reports its shape. */
extern "C" int drt_bvh_stats(const drt_scene *scene, uint32_t *nodes, uint32_t *leaf_surfaces, uint32_t *depth, uint32_t *stack_entries)
{
    g_last_error.clear();
    if (!scene) return fail(-1, "null argument");
    BvhBuilder bb;
    bb.build(scene, 0.0);
    if (nodes) *nodes = (uint32_t)bb.nodes.size();
    if (leaf_surfaces) *leaf_surfaces = (uint32_t)bb.order.size();
    if (depth) *depth = (uint32_t)bb.max_depth;
    if (stack_entries) *stack_entries = BVH_STACK;
    return 0;
}

extern "C" int drt_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(-1, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

/* bytes of the first film buffer: sum+filter rows, or the XYZ accumulators */
static size_t pixels_bytes(const drt_context *ctx)
{
    return (size_t)ctx->n_pix * (ctx->xyz_mode ? (size_t)XYZ_FILM_WORDS : (size_t)ctx->dsc.S + 1) * 8;
}

static int enqueue_trace(drt_context *ctx, uint32_t first_sample, uint32_t n, uint32_t hits_sample_offset, uint32_t row0 = 0, uint32_t rows = 0, uint32_t stride = 0);

/* waves of the trace-stage kernel that takes record blocks, for a launch of n_paths */
static uint64_t trace_waves(const drt_context *ctx, uint64_t n_paths)
{
    const uint64_t cap = (uint64_t)(ctx->bvh_pipeline ? ctx->bounce_grid_cap : ctx->trace_grid_cap);
    return std::min<uint64_t>(cap, (n_paths + TRACE_BLOCK - 1) / TRACE_BLOCK) * (TRACE_BLOCK / 64);
}
/* blocks that are enough for ANY launch of n_paths: every path max_depth vertices; fewer than 64 blocks of every POOL_CHUNK
 * stay unused when a wave moves on to a new chunk, every wave ends on a part-used one, and its lanes on a spare (or two) each */
static uint64_t blocks_worst_case(const drt_context *ctx, uint64_t n_paths)
{
    return (uint64_t)((double)n_paths * ctx->worst_blocks_per_path * (1.0 + 64.0 / POOL_CHUNK)) + trace_waves(ctx, n_paths) * (POOL_CHUNK + 2 * 64) + 1;
}

static int create_impl(drt_context *ctx, const drt_scene *scene, const drt_camera *camera, const drt_params *params)
{
    if (!scene || !camera || !params) return fail(-1, "null argument");
    if (params->tile_w == 0 || params->tile_h == 0 || params->max_depth == 0) return fail(-1, "empty tile or zero depth");
    if (params->mode != DRT_MODE_SPECTRAL && params->mode != DRT_MODE_XYZ) return fail(-1, "unknown film mode %u", params->mode);
    ctx->xyz_mode = params->mode == DRT_MODE_XYZ;
    if ((uint64_t)params->x0 + params->tile_w > params->width || (uint64_t)params->y0 + (uint64_t)(params->tile_h - 1) * (params->row_stride ? params->row_stride : 1) >= params->height)
        return fail(-1, "tile does not fit the %ux%u image", params->width, params->height);
    ctx->params = *params;
    if (ctx->params.row_stride == 0) ctx->params.row_stride = 1;
    ctx->device = params->device;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    /* ray origins that are not on a surface are on the camera: its aperture and its film */
    double reach = 0.0;
    for (int k = 0; k < 3; k += 1)
    {
        const double a = camera->aperture_position[k], f = camera->film_bottom_left[k];
        reach = std::max(reach, std::fabs(a) + 2.0 * std::fabs(f - a) + std::fabs(camera->aperture_radius));
    }
    int rc = build_device_scene(ctx, scene, reach);
    if (rc) return rc;

    DevCamera &c = ctx->dcam;
    c.forward = hv(camera->forward); c.right = hv(camera->right); c.up = hv(camera->up);
    c.aperture_position = hv(camera->aperture_position); c.film_bottom_left = hv(camera->film_bottom_left);
    c.aperture_radius = camera->aperture_radius; c.focal_depth = camera->focal_depth;
    c.pixel_width = camera->pixel_width; c.pixel_height = camera->pixel_height;

    const uint32_t S = scene->num_wavelengths;
    ctx->n_pix = (uint64_t)params->tile_w * params->tile_h;
    /* vertex record stride: fixed part + one block per light, rounded up to a power of two (>= 16 words) so a
     * vertex never straddles the shade kernel's 64-word prefetch registers */
    ctx->vertex_words = 16;
    while (ctx->vertex_words < REC_VERTEX_WORDS + REC_LIGHT_WORDS * ctx->dsc.n_lights) ctx->vertex_words *= 2;
    ctx->vertex_shift = 0;
    while ((1u << ctx->vertex_shift) < ctx->vertex_words) ctx->vertex_shift += 1;
    ctx->block_words = REC_BLOCK_VERTICES * ctx->vertex_words;
    /* a path of max_depth vertices: a block per four, plus the table block once the header's three are used up */
    const uint32_t deepest = (params->max_depth + REC_BLOCK_VERTICES - 1) / REC_BLOCK_VERTICES;
    ctx->worst_blocks_per_path = deepest + (deepest > REC_HEADER_BLOCKS ? 1u : 0u);
    if (deepest > REC_HEADER_BLOCKS + 2 * ctx->block_words)
        return fail(-2, "max_depth %u: a path's table block holds %u blocks beyond the header's %d", params->max_depth, 2 * ctx->block_words, REC_HEADER_BLOCKS);
    if (ctx->dsc.n_spd >= 0xFFFFu) return fail(-2, "too many SPDs for the 16-bit record indices");
    if (S > 64 * SHADE_MAX_SETS) return fail(-2, "more than %d wavelengths", 64 * SHADE_MAX_SETS);
    {
        uint32_t sets = 0, tf = 0, tc = 0;
        shade_sets(S, &sets, &tf, &tc);
        ctx->tail_count = tc;
    }
    /* scenes behind the hierarchy: camera rays walk it a wave at a time, the rest of each path runs from a queue (drt_bvh_kernels.h) */
    ctx->bvh_pipeline = !ctx->scene_in_lds;

    HIP_TRY(hipMalloc((void **)&ctx->d_pixels, pixels_bytes(ctx)));
    if (!ctx->xyz_mode)
    {
        HIP_TRY(hipMalloc((void **)&ctx->d_avgs, (size_t)ctx->n_pix * S * 8));
        HIP_TRY(hipMalloc((void **)&ctx->d_vars, (size_t)ctx->n_pix * S * 8));
    }
    ctx->own_film = true;
    HIP_TRY(hipMemsetAsync(ctx->d_pixels, 0, pixels_bytes(ctx), ctx->stream));
    if (ctx->d_avgs) HIP_TRY(hipMemsetAsync(ctx->d_avgs, 0, (size_t)ctx->n_pix * S * 8, ctx->stream));
    if (ctx->d_vars) HIP_TRY(hipMemsetAsync(ctx->d_vars, 0, (size_t)ctx->n_pix * S * 8, ctx->stream));
    HIP_TRY(hipMalloc((void **)&ctx->d_counters, DRT_COUNTER_WORDS * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, DRT_COUNTER_WORDS * sizeof(unsigned long long), ctx->stream));

    /* persistent trace grid: as many workgroups as the chip keeps resident */
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    int per_cu = 0;
    if (ctx->scene_in_lds && ctx->trace_tail) /* the instantiation that will be launched: its registers and LDS decide the grid */
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, drt_trace_kernel<true, true>, TRACE_BLOCK, ctx->trace_lds));
    else if (ctx->scene_in_lds)
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, drt_trace_kernel<true, false>, TRACE_BLOCK, ctx->trace_lds));
    if (per_cu < 1) per_cu = 1;
    if (const char *e = getenv("DRT_TRACE_BLOCKS_PER_CU")) per_cu = std::max(1, atoi(e)); /* tuning knob */
    ctx->trace_grid_cap = prop.multiProcessorCount * per_cu;
    if (ctx->bvh_pipeline)
    {
        int p_cu = 0, b_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&p_cu, drt_primary_kernel, PRIMARY_BLOCK, 0));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_cu, drt_bounce_kernel, BOUNCE_BLOCK, 0));
        if (const char *e = getenv("DRT_TRACE_BLOCKS_PER_CU")) b_cu = std::max(1, atoi(e));
        ctx->primary_grid_cap = prop.multiProcessorCount * std::max(1, p_cu);
        ctx->bounce_grid_cap = prop.multiProcessorCount * std::max(1, b_cu);
    }
    /* shade kernel LDS: SPD tables + two record buffers per wave */
    ctx->shade_lds = ctx->spds_in_lds ? (size_t)ctx->dsc.n_spd * S * 8 : 0;
    ctx->shade_lds += (size_t)SHADE_WAVES * SHADE_WAVE_LDS_WORDS * 8; /* a wave's own region: two record slots in the main pass (the coefficient words of plastic
                                                                          vertices are read back from them), a window of headers in the tail pass */
    int s_per_cu = 0;
    shade_sets(S, &ctx->shade_sets, &ctx->tail_first, &ctx->tail_count);
    switch (ctx->shade_sets)
    {
        case 1: rc = shade_occupancy<1>(ctx, &s_per_cu); break;
        case 2: rc = shade_occupancy<2>(ctx, &s_per_cu); break;
        case 3: rc = shade_occupancy<3>(ctx, &s_per_cu); break;
        default: rc = shade_occupancy<4>(ctx, &s_per_cu); break;
    }
    if (rc) return rc;
    if (s_per_cu < 1) s_per_cu = 1;
    if (const char *e = getenv("DRT_SHADE_BLOCKS_PER_CU")) s_per_cu = std::max(1, atoi(e)); /* tuning knob */
    ctx->shade_grid_cap = prop.multiProcessorCount * s_per_cu;
    /* Both grids are persistent and fill every wave slot of the chip, so a kernel of another stream -- a collective's,
     * a copy's -- starts only when one of them ends. DRT_RESERVE_BLOCKS=n leaves n workgroup slots free for such
     * company (bench.py sets it when a gather runs behind the rendering). */
    if (const char *e = getenv("DRT_RESERVE_BLOCKS"))
    {
        int n = std::max(0, atoi(e));
        ctx->trace_grid_cap = std::max(1, ctx->trace_grid_cap - n);
        ctx->shade_grid_cap = std::max(1, ctx->shade_grid_cap - n);
    }
    if (const char *e = getenv("DRT_TRACE_CHUNK")) ctx->trace_chunk_override = (uint32_t)std::min(1 << 20, std::max(64, atoi(e) / 64 * 64));
    if (const char *e = getenv("DRT_TAIL_PERIOD")) ctx->tail_period_override = (uint32_t)std::max(0, atoi(e));
    if (const char *e = getenv("DRT_SHADE_SUBS")) ctx->shade_subs_override = (uint32_t)std::max(0, atoi(e));
    /*
     * The record pool and the launch size. A launch of N paths needs about N * b blocks, b = blocks per path on THIS tile of
     * THIS scene (0.9 on the Cornell frame, where a worst-case path would take 4): b is measured here, once, on a sample of the
     * tile (every k-th row, one sample per pixel, traced into a small pool sized for the worst case), and the pool gets 1.2 b
     * per path, a quarter more for what the waves' chunks leave unused, and a margin for small launches -- so 64 M paths take
     * 28 GB instead of 68. If a launch runs out after all, its
     * shade kernel and everything queued behind it do nothing and the host renders those samples again in launches sized for the
     * worst case (redo_batches): slower, never wrong. Launch size: large launches are the efficient ones (their last round is
     * amortised: DESIGN.md, work queues); the default (batch_spp = 0, a one-shot job) follows the job announced in params->spp -- about
     * 32 kernel pairs, at least 16 GB of records, at most 64 M paths per launch. A larger pool would save launches (one pair per row
     * block of the one-shot call instead of two: kernels 179 -> 171 ms with 40 GB) but a process that starts right after another has
     * freed tens of GB waits for the driver to scrub them, 1-2 s per 40 GB (tools/r03_oneshot_blocks.sh): 16 GB stays.
     * Callers that keep a context across many frames pass DRT_BATCH_RESIDENT (below) or a batch_spp of their own.
     */
    const size_t block_bytes = (size_t)ctx->block_words * 8;
    const uint64_t npx = std::max<uint64_t>(ctx->n_pix, 1);
    const size_t per_path_fixed = REC_HEADER_WORDS * 8 + (size_t)ctx->tail_count * 8 + (ctx->bvh_pipeline ? sizeof(PrimaryHit) + sizeof(uint64_t) : 0);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    ctx->est_blocks_per_path = (double)ctx->worst_blocks_per_path;
    if (npx * (uint64_t)ctx->worst_blocks_per_path * block_bytes > (256ull << 20) && !getenv("DRT_POOL_WORST_CASE"))
    {
        /* the sample: tile rows 0, k, 2k, ... with k such that it holds about 128 k paths */
        const uint32_t k = (uint32_t)std::max<uint64_t>(1, npx / (128u << 10));
        drt_params pp = ctx->params;
        pp.tile_h = (ctx->params.tile_h + k - 1) / k;
        pp.row_stride = ctx->params.row_stride * k;
        const uint64_t n_sample = (uint64_t)pp.tile_w * pp.tile_h;
        ctx->n_pix = n_sample;
        ctx->pool_blocks = blocks_worst_case(ctx, n_sample);
        ctx->n_pix = npx;
        HIP_TRY(hipMalloc((void **)&ctx->d_records, ctx->pool_blocks * block_bytes));
        HIP_TRY(hipMalloc((void **)&ctx->d_headers, n_sample * REC_HEADER_WORDS * 8));
        if (ctx->bvh_pipeline)
        {
            HIP_TRY(hipMalloc((void **)&ctx->d_primary, n_sample * sizeof(PrimaryHit)));
            HIP_TRY(hipMalloc((void **)&ctx->d_queue, n_sample * sizeof(uint64_t)));
        }
        const drt_params keep = ctx->params;
        const uint64_t keep_pix = ctx->n_pix;
        ctx->params = pp;
        ctx->params.flags &= ~(uint32_t)DRT_FLAG_RECORD_HITS;
        ctx->n_pix = n_sample;
        ctx->batch_spp = 1;
        rc = enqueue_trace(ctx, keep.first_sample, 1, 0);
        ctx->params = keep;
        ctx->n_pix = keep_pix;
        if (rc) return rc;
        unsigned long long used = 0; /* blocks handed to paths (the cursor also counts the waves' part-used chunks) */
        HIP_TRY(hipMemcpyAsync(&used, ctx->d_counters + DRT_PAIR_COUNTERS + 5, sizeof(used), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->est_blocks_per_path = (double)used / (double)n_sample;
        {
            /* vertices per path on the sample: a scene whose paths average three vertices and more is a closed one -- hardly a pixel
             * stays dark there, and the shade kernel without the test for it is the faster one (config 3: 1767 against 1806-1837 ms) */
            unsigned long long cnt[3] = {0, 0, 0}; /* paths, scans, shaded vertices of the sample */
            HIP_TRY(hipMemcpy(cnt, ctx->d_counters + DRT_PAIR_COUNTERS, sizeof(cnt), hipMemcpyDeviceToHost));
            if (cnt[0] > 0) ctx->dark_skip = (double)cnt[2] / (double)cnt[0] < 3.0;
        }
        (void)hipFree(ctx->d_records); ctx->d_records = nullptr;
        (void)hipFree(ctx->d_headers); ctx->d_headers = nullptr;
        (void)hipFree(ctx->d_primary); ctx->d_primary = nullptr;
        (void)hipFree(ctx->d_queue); ctx->d_queue = nullptr;
        HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, DRT_COUNTER_WORDS * sizeof(unsigned long long), ctx->stream)); /* the sample does not count */
        ctx->ev_used = 0;
        ctx->ev_paths.clear();
    }
    /* blocks a launch of n paths is given: 1.3 x the expectation, 8 sigma of a sum of n on top (a path's block count has a
     * standard deviation below 2), never less than one sample per pixel in the worst case (what redo_batches launches) */
    auto blocks_for = [&](uint64_t n_paths) -> uint64_t {
        const double want = 1.2 * ctx->est_blocks_per_path * (double)n_paths + 16.0 * std::sqrt((double)n_paths) + 4096.0;
        const uint64_t expect = (uint64_t)(want * (1.0 + 64.0 / POOL_CHUNK)) + trace_waves(ctx, n_paths) * (POOL_CHUNK + 2 * 64) + 1; /* + the lanes' spares */
        return std::max<uint64_t>(std::min<uint64_t>(expect, blocks_worst_case(ctx, n_paths)), blocks_worst_case(ctx, npx));
    };
    uint32_t batch = params->batch_spp;
    if (batch == DRT_BATCH_RESIDENT)
    {
        /* a context kept across many frames: up to 256 M paths per kernel pair -- every launch ends on a partly idle chip (work
         * queues), and 288 GB of HBM hold the 116 GB of records that takes on the Cornell frame: 1024^2 x 256 spp in one pair instead of
         * four, 1525 -> 1568 Mpaths/s (the loop below halves the launch until its records fit half of what is free) -- but never fewer
         * than 16 samples per pixel and launch, because the film is read and written once per launch: 3328 bytes per pixel, which at
         * 4 samples a launch (the 4096^2 frame of config 5) was a third of the frame */
        const uint64_t resident_paths = getenv("DRT_RESIDENT_PATHS_M") ? (uint64_t)std::max(1, atoi(getenv("DRT_RESIDENT_PATHS_M"))) << 20 : (256ull << 20);
        const uint64_t by_paths = std::max<uint64_t>(1, std::min<uint64_t>(DRT_DEFAULT_MAX_BATCH, resident_paths / npx));
        batch = (uint32_t)std::max<uint64_t>(by_paths, 16);
        if (params->spp) batch = std::min(batch, params->spp);
    }
    else if (batch == 0)
    {
        const double path_bytes = 1.2 * (1.0 + 64.0 / POOL_CHUNK) * ctx->est_blocks_per_path * (double)block_bytes + (double)per_path_fixed;
        uint64_t by_job = (std::max<uint32_t>(params->spp, 1) + 31) / 32;
        uint64_t by_floor = (uint64_t)((double)(16ull << 30) / (path_bytes * (double)npx));
        uint64_t cap = std::max<uint64_t>(1, std::min<uint64_t>(DRT_DEFAULT_MAX_BATCH, (64ull << 20) / npx));
        batch = (uint32_t)std::max<uint64_t>(1, std::min(std::max(by_job, by_floor), cap));
        if (params->spp) batch = std::min(batch, params->spp);
    }
    batch = std::min<uint32_t>(batch, 4096);
    const size_t film_bytes = ctx->xyz_mode ? (size_t)ctx->n_pix * XYZ_FILM_WORDS * 8 : (size_t)ctx->n_pix * (3 * (size_t)S + 1) * 8;
    (void)film_bytes; /* allocated above: free_b already excludes it */
    auto bytes_for = [&](uint32_t b) -> size_t { return blocks_for(npx * b) * block_bytes + (size_t)npx * b * per_path_fixed; };
    while (batch > 1 && bytes_for(batch) > free_b / 2) batch /= 2;
    if (bytes_for(batch) > free_b) return fail(-3, "not enough device memory: need %zu bytes", bytes_for(batch));
    ctx->batch_spp = batch;
    ctx->pool_blocks = blocks_for(npx * batch);
    if (const char *e = getenv("DRT_POOL_BLOCKS")) /* test knob: a pool this small (never below one worst-case sample per pixel) */
        ctx->pool_blocks = std::max<uint64_t>(blocks_worst_case(ctx, npx), std::min<uint64_t>(ctx->pool_blocks, strtoull(e, nullptr, 0)));
    if (ctx->pool_blocks >= 0xFFFFFFF0ull) return fail(-3, "record pool of %llu blocks: block numbers are 32 bits", (unsigned long long)ctx->pool_blocks);
    HIP_TRY(hipMalloc((void **)&ctx->d_records, ctx->pool_blocks * block_bytes));
    HIP_TRY(hipMalloc((void **)&ctx->d_headers, (size_t)npx * batch * REC_HEADER_WORDS * 8));
    if (ctx->bvh_pipeline)
    {
        HIP_TRY(hipMalloc((void **)&ctx->d_primary, (size_t)npx * batch * sizeof(PrimaryHit)));
        HIP_TRY(hipMalloc((void **)&ctx->d_queue, (size_t)npx * batch * sizeof(uint64_t)));
    }
    if (ctx->tail_count) HIP_TRY(hipMalloc((void **)&ctx->d_tail_stage, (size_t)npx * batch * ctx->tail_count * 8));
    if (const char *e = getenv("DRT_DARK_SKIP")) ctx->dark_skip = atoi(e) != 0; /* A/B switch of the parity tests: same film either way */
    if (getenv("DRT_VERBOSE"))
        fprintf(stderr, "drt: %d CUs, trace %d blocks/CU (lds %zu), shade %d blocks/CU (lds %zu), batch %u, %.3f blocks/path measured (worst %u), pool %.2f GB\n",
                prop.multiProcessorCount, per_cu, ctx->trace_lds, s_per_cu, ctx->shade_lds, ctx->batch_spp, ctx->est_blocks_per_path,
                ctx->worst_blocks_per_path, (double)(ctx->pool_blocks * block_bytes) / 1e9);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" drt_context *drt_create(const drt_scene *scene, const drt_camera *camera, const drt_params *params)
{
    g_last_error.clear();
    drt_context *ctx = new drt_context();
    int rc = create_impl(ctx, scene, camera, params);
    if (rc != 0)
    {
        std::string keep = g_last_error;
        drt_destroy(ctx);
        (void)hipGetLastError();
        g_last_error = keep;
        return nullptr;
    }
    return ctx;
}

extern "C" void drt_destroy(drt_context *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (void *p : ctx->allocations) (void)hipFree(p);
    if (ctx->own_film)
    {
        (void)hipFree(ctx->d_pixels);
        (void)hipFree(ctx->d_avgs);
        (void)hipFree(ctx->d_vars);
    }
    (void)hipFree(ctx->d_records);
    (void)hipFree(ctx->d_headers);
    (void)hipFree(ctx->d_primary);
    (void)hipFree(ctx->d_queue);
    (void)hipFree(ctx->d_tail_stage);
    (void)hipFree(ctx->d_hits);
    (void)hipFree(ctx->d_counters);
    (void)hipFree(ctx->d_xyz);
    (void)hipFree(ctx->d_bgra);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" int drt_bind_film(drt_context *ctx, void *d_pixels, void *d_avgs, void *d_vars)
{
    if (!ctx || !d_pixels || (!ctx->xyz_mode && (!d_avgs || !d_vars))) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->own_film)
    {
        (void)hipFree(ctx->d_pixels);
        (void)hipFree(ctx->d_avgs);
        (void)hipFree(ctx->d_vars);
        ctx->own_film = false;
    }
    ctx->d_pixels = (double *)d_pixels;
    ctx->d_avgs = (double *)d_avgs;
    ctx->d_vars = (double *)d_vars;
    return 0;
}

extern "C" int drt_set_stream(drt_context *ctx, void *hip_stream)
{
    if (!ctx) return fail(-1, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return 0;
}

extern "C" int drt_synchronize(drt_context *ctx);

static int next_events(drt_context *ctx, hipEvent_t out[3])
{
    while (ctx->ev.size() < ctx->ev_used + 3)
    {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        ctx->ev.push_back(e);
    }
    for (int k = 0; k < 3; k += 1) out[k] = ctx->ev[ctx->ev_used + k];
    ctx->ev_used += 3;
    return 0;
}

/* fold finished event triples into trace_ms / shade_ms (stream must be idle) */
static int collect_timings(drt_context *ctx)
{
    for (size_t k = 0; k + 3 <= ctx->ev_used; k += 3)
    {
        float a = 0.f, b = 0.f;
        HIP_TRY(hipEventElapsedTime(&a, ctx->ev[k], ctx->ev[k + 1]));
        HIP_TRY(hipEventElapsedTime(&b, ctx->ev[k + 1], ctx->ev[k + 2]));
        ctx->trace_ms += a;
        ctx->shade_ms += b;
        const double paths = k / 3 < ctx->ev_paths.size() ? ctx->ev_paths[k / 3] : 0.0;
        if (paths > 0.0)
        {
            const double t = ((double)a + (double)b) * (double)ctx->n_pix / paths; /* one sample of every tile pixel at this pair's rate */
            if (ctx->timed_pairs == 0 || t < ctx->min_sample_ms) ctx->min_sample_ms = t;
            if (ctx->timed_pairs == 0 || t > ctx->max_sample_ms) ctx->max_sample_ms = t;
            ctx->timed_pairs += 1;
            ctx->avg_sample_ms += (t - ctx->avg_sample_ms) / (double)ctx->timed_pairs;
        }
    }
    ctx->ev_used = 0;
    ctx->ev_paths.clear();
    return 0;
}

/* after a kernel pair: if the pool did not run out, this pair is the last complete one; the pool's high-water mark */
__global__ void drt_mark_pair_kernel(unsigned long long *totals, unsigned long long seq)
{
    /* state[0] pool cursor of this pair, [1] overflow flag, [2] last complete pair, [3] peak of the cursor */
    unsigned long long *state = totals + DRT_NUM_COUNTERS + 4, *pair = totals + DRT_PAIR_COUNTERS;
    const bool complete = *(const uint32_t *)(state + 1) == 0u;
    if (complete) state[2] = seq;
    if (state[0] > state[3]) state[3] = state[0];
    for (int k = 0; k < DRT_NUM_COUNTERS; k += 1)
    {
        if (complete) totals[k] += pair[k]; /* an incomplete pair is rendered again: its paths are counted then */
        pair[k] = 0;
    }
}

/* The trace stage of one kernel pair over samples [first_sample, first_sample + n) of every tile pixel: work queues and the
 * pool cursor reset, then drt_trace_kernel (scene in LDS) or drt_primary_kernel + drt_bounce_kernel (scene behind the hierarchy). */
static int enqueue_trace(drt_context *ctx, uint32_t first_sample, uint32_t n, uint32_t hits_sample_offset, uint32_t row0, uint32_t rows, uint32_t stride)
{
    const drt_params &p = ctx->params;
    TraceParams tp{};
    /* rows [row0, row0 + rows) of the tile (rows == 0: all of it): the kernels see a tile that starts there; headers, records and the
     * staging buffers are indexed from the launch's first pixel */
    const bool whole = rows == 0;
    if (whole) { row0 = 0; rows = p.tile_h; }
    tp.width = p.width; tp.height = p.height; tp.x0 = p.x0; tp.y0 = p.y0 + row0 * p.row_stride;
    tp.tile_w = p.tile_w; tp.tile_h = rows; tp.row_stride = p.row_stride;
    tp.first_sample = first_sample;
    tp.n_samples = n;
    tp.max_depth = p.max_depth;
    tp.pixel_scheme = p.pixel_scheme;
    tp.record_hits = (p.flags & DRT_FLAG_RECORD_HITS) ? 1u : 0u;
    tp.seed = p.seed;
    tp.n_pix = whole ? ctx->n_pix : (uint64_t)rows * p.tile_w;
    tp.n_paths = tp.n_pix * n;
    tp.vertex_words = ctx->vertex_words;
    tp.block_words = ctx->block_words;
    tp.hits_sample_offset = hits_sample_offset;
    tp.batch = stride ? stride : ctx->batch_spp; /* sample slots per pixel in the header array (a launch over fewer rows may take more samples) */
    tp.pool_blocks = (uint32_t)ctx->pool_blocks;
    tp.tail_stage = ctx->trace_tail ? ctx->d_tail_stage : nullptr;
    tp.spd_tail = ctx->d_spd_tail;
    tp.tail_count = ctx->tail_count;
    tp.n_spd = ctx->dsc.n_spd;
    unsigned long long *work = ctx->d_counters + DRT_NUM_COUNTERS;
    tp.pool_cursor = work + 4;
    tp.overflow = (uint32_t *)(work + 5);
    HIP_TRY(hipMemsetAsync(work, 0, 5 * sizeof(unsigned long long), ctx->stream)); /* trace + shade work queues, bounce queue length, spare, pool cursor */
    uint64_t blocks_needed = (tp.n_paths + TRACE_BLOCK - 1) / TRACE_BLOCK;
    const int grid_cap = ctx->bvh_pipeline ? ctx->bounce_grid_cap : ctx->trace_grid_cap;
    uint32_t grid = (uint32_t)std::min<uint64_t>(blocks_needed, (uint64_t)grid_cap);
    /* work-queue granularity: about 16 draws per wave, so that the last draws finish together; 64..1024 path ids */
    {
        uint64_t waves = (uint64_t)grid * (TRACE_BLOCK / 64);
        uint64_t c = tp.n_paths / (waves * 16) / 64 * 64;
        tp.chunk = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(c, 64), ctx->bvh_pipeline ? 256 : 1024); /* queued paths all cost alike: small draws (config 5: 827 ms at 64-256, 855 at 1024) */
        if (ctx->trace_chunk_override) tp.chunk = ctx->trace_chunk_override;
    }
    if (ctx->bvh_pipeline)
    {
        /* camera rays: a wave per 64 path ids; then the queued paths, one per lane */
        const uint64_t packets = (tp.n_paths + 63) / 64;
        const uint32_t pgrid = (uint32_t)std::min<uint64_t>((packets + PRIMARY_BLOCK / 64 - 1) / (PRIMARY_BLOCK / 64), (uint64_t)ctx->primary_grid_cap);
        hipLaunchKernelGGL(drt_primary_kernel, dim3(pgrid), dim3(PRIMARY_BLOCK), 0, ctx->stream, ctx->dsc, ctx->dcam, tp, ctx->d_headers,
                           ctx->d_hits, ctx->d_counters + DRT_PAIR_COUNTERS, ctx->d_primary, ctx->d_queue, work + 2);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(drt_bounce_kernel, dim3(grid), dim3(BOUNCE_BLOCK), 0, ctx->stream, ctx->dsc, ctx->dcam, tp, ctx->d_records,
                           ctx->d_headers, ctx->d_hits, ctx->d_counters + DRT_PAIR_COUNTERS, work, ctx->d_primary, ctx->d_queue, work + 2);
    }
    else if (ctx->trace_tail && tp.tail_stage)
        hipLaunchKernelGGL((drt_trace_kernel<true, true>), dim3(grid), dim3(TRACE_BLOCK), ctx->trace_lds, ctx->stream, ctx->dsc,
                           ctx->dcam, tp, ctx->d_records, ctx->d_headers, ctx->d_hits, ctx->d_counters + DRT_PAIR_COUNTERS, work);
    else
        hipLaunchKernelGGL((drt_trace_kernel<true, false>), dim3(grid), dim3(TRACE_BLOCK), ctx->trace_lds, ctx->stream, ctx->dsc,
                           ctx->dcam, tp, ctx->d_records, ctx->d_headers, ctx->d_hits, ctx->d_counters + DRT_PAIR_COUNTERS, work);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* One kernel pair: trace, shade + film, and the mark that tells the host whether the pair was complete. */
static int enqueue_pair(drt_context *ctx, uint32_t first_sample, uint32_t n, uint32_t hits_sample_offset, uint32_t row0 = 0, uint32_t rows = 0, uint32_t stride = 0)
{
    hipEvent_t ev[3];
    int rc = next_events(ctx, ev);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    if ((rc = enqueue_trace(ctx, first_sample, n, hits_sample_offset, row0, rows, stride))) return rc;
    const uint64_t n_pix = rows ? (uint64_t)rows * ctx->params.tile_w : ctx->n_pix; /* pixels of this launch */
    const uint64_t pix0 = rows ? (uint64_t)row0 * ctx->params.tile_w : 0;           /* its first pixel in the tile */
    HIP_TRY(hipEventRecord(ev[1], ctx->stream));

    unsigned long long *work = ctx->d_counters + DRT_NUM_COUNTERS;
    ShadeParams sp{};
    sp.n_pix = n_pix;
    sp.n_samples = n;
    sp.first_sample = first_sample;
    sp.vertex_words = ctx->vertex_words;
    sp.vertex_shift = ctx->vertex_shift;
    sp.block_words = ctx->block_words;
    sp.overflow = (const uint32_t *)(work + 5);
    sp.n_lights = ctx->dsc.n_lights;
    sp.batch = stride ? stride : ctx->batch_spp;
    sp.tail_first = ctx->tail_first;
    sp.tail_count = ctx->tail_count;
    sp.tail_stage = ctx->d_tail_stage;
    sp.light0_em_spd = ctx->light0_em_spd;
    sp.tail_staged = (ctx->tail_all_staged && ctx->d_tail_stage) ? 1u : 0u;
    if (const char *e = getenv("DRT_DEBUG_SHADE_MODE")) sp.mode = (uint32_t)atoi(e); /* timing probe: 1 main pass only, 2 tail pass only */
    if (const char *e = getenv("DRT_DEBUG_TAIL_PHASE_A_OFF")) sp.tail_staged = (uint32_t)atoi(e) ? 1u : sp.tail_staged;
    sp.cmf_rw = ctx->cmf_rw; sp.cmf_x = ctx->cmf_x; sp.cmf_y = ctx->cmf_y; sp.cmf_z = ctx->cmf_z;
    sp.chunk = ctx->tail_count ? 64u / ctx->tail_count : SHADE_PIXEL_CHUNK;
    if (ctx->tail_count && !sp.tail_staged)
    {
        /* A group's tail-pass item is the longest item of the queue: the replay of every path of its pixels that the trace kernel did
         * not carry, 1-4 ms where glass fills them -- the floor under a small launch. Its lane groups take PATHS, not pixels, so a group
         * of fewer pixels is replayed by the same 64 / R lane groups in that much less time, and only the short film phase runs with
         * lanes to spare. Worth it where a wave gets fewer than two such items (64 rows x 1024 px x 256 spp: shade 6.8 -> 6.1 ms with
         * half the pixels per group, 8 rows: 3.9 -> 1.6 with a quarter); the whole frame loses (80.2 -> 82.7: more, emptier film phases). */
        const uint64_t tail_items = (n_pix + sp.chunk - 1) / sp.chunk, waves_ = (uint64_t)ctx->shade_grid_cap * SHADE_WAVES;
        if (2 * tail_items < waves_) sp.chunk = std::max(1u, sp.chunk / 4);
        else if (tail_items < 2 * waves_) sp.chunk = std::max(1u, sp.chunk / 2);
    }
    if (const char *e = getenv("DRT_SHADE_CHUNK")) sp.chunk = std::max(1u, std::min(ctx->tail_count ? 64u / ctx->tail_count : SHADE_PIXEL_CHUNK, (uint32_t)atoi(e))); /* tuning knob */
    uint64_t groups = (n_pix + sp.chunk - 1) / sp.chunk;
    const bool inline_tail = ctx->tail_count != 0;
    /* work items: a group's main pass in pieces of sub_pixels pixels (+ its tail pass as an item of its own) when the
     * groups alone are too few to keep the last round of the persistent waves short */
    {
        const uint64_t waves = (uint64_t)ctx->shade_grid_cap * SHADE_WAVES;
        uint32_t subs = ctx->shade_subs_override;
        if (subs == ~0u)
        {
            /* pixels per main-pass item: small enough for >= 64 items per wave (short last round), large enough for
             * >= 256 paths per item (the queue is one atomic counter) */
            uint64_t p_balance = n_pix / (waves * 64);
            uint64_t p_atomic = (256 + n - 1) / n;
            uint32_t P = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(std::max(p_balance, p_atomic), 1), sp.chunk);
            subs = (sp.chunk + P - 1) / P;
            if (subs == 1 && !inline_tail) subs = 0;
        }
        subs = std::min(subs, sp.chunk);
        if (subs == 0 || (subs == 1 && !inline_tail) || groups * (uint64_t)(sp.chunk + 1) >= 0xFFFFFFFFull)
        {
            sp.sub_pixels = sp.chunk;
            sp.items_per_group = 1;
        }
        else
        {
            sp.sub_pixels = (sp.chunk + subs - 1) / subs;
            sp.items_per_group = (sp.chunk + sp.sub_pixels - 1) / sp.sub_pixels + (inline_tail ? 1 : 0);
        }
        if (groups * sp.items_per_group >= 0xFFFFFFFFull) return fail(-1, "tile too large for the shade work queue");
        sp.n_items = (uint32_t)(groups * sp.items_per_group);
        if (inline_tail && sp.items_per_group > 1)
        {
            uint32_t mains = sp.items_per_group - 1;
            /* tail items (the longest ones, about a millisecond each) evenly through the queue when every wave gets many of them; when a
             * wave gets only a few (a rank's share of a frame, a row block of the one-shot call), they go out in the first third of the
             * queue and the launch ends on main-pass pieces only (64 rows x 1024 px x 256 spp: shade 7.0 -> 6.2 ms with the tails in the
             * first 1/6 to 1/2 of the queue, 7.8 when spread over all of it; the whole frame does not care: 79.9-80.1 ms at any setting) */
            sp.tail_period_mains = (groups >= 8 * waves) ? mains : std::max<uint32_t>(1, mains / 3);
            if (ctx->tail_period_override) sp.tail_period_mains = std::min(mains, ctx->tail_period_override);
        }
    }
    uint32_t sgrid = (uint32_t)std::min<uint64_t>(((uint64_t)sp.n_items + SHADE_WAVES - 1) / SHADE_WAVES, (uint64_t)ctx->shade_grid_cap);
    const size_t S_ = ctx->dsc.S;
    double *const film[3] = {ctx->d_pixels + pix0 * (ctx->xyz_mode ? (size_t)XYZ_FILM_WORDS : S_ + 1),
                             ctx->d_avgs ? ctx->d_avgs + pix0 * S_ : nullptr, ctx->d_vars ? ctx->d_vars + pix0 * S_ : nullptr};
    rc = launch_shade(ctx, sgrid, sp, film);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    const uint64_t seq = ctx->next_seq++;
    hipLaunchKernelGGL(drt_mark_pair_kernel, dim3(1), dim3(1), 0, ctx->stream, ctx->d_counters, (unsigned long long)seq);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[2], ctx->stream));
    ctx->ev_paths.resize(ctx->ev_used / 3, 0.0);
    ctx->ev_paths[ctx->ev_used / 3 - 1] = (double)n_pix * (double)n;
    ctx->inflight.push_back({first_sample, n, seq, row0, rows, hits_sample_offset, stride});
    return 0;
}

/* The pool ran out in some kernel pair: that pair's shade kernel and every later kernel did nothing. Render those samples again,
 * in launches small enough for the worst case (every path max_depth vertices), synchronously. */
static int redo_batches(drt_context *ctx, uint64_t last_good_seq)
{
    std::vector<drt_context::Batch> todo;
    for (const auto &b : ctx->inflight) if (b.seq > last_good_seq) todo.push_back(b);
    ctx->inflight.clear();
    unsigned long long *work = ctx->d_counters + DRT_NUM_COUNTERS;
    HIP_TRY(hipMemsetAsync(work + 5, 0, sizeof(unsigned long long), ctx->stream)); /* the overflow flag */
    /* samples per launch that fit the pool whatever the paths do (one always does: the pool is never smaller, create_impl) */
    uint32_t safe = 1;
    while (safe < ctx->batch_spp && blocks_worst_case(ctx, ctx->n_pix * (uint64_t)(safe + 1)) <= ctx->pool_blocks) safe += 1;
    for (const auto &b : todo)
    {
        ctx->redone_batches += 1;
        for (uint32_t done = 0; done < b.n_samples; done += safe)
        {
            /* (the hit log is indexed by sample offset within the caller's drt_render call: the batch remembers its own) */
            int rc = enqueue_pair(ctx, b.first_sample + done, std::min(safe, b.n_samples - done), b.hits_sample_offset + done, b.row0, b.rows);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    unsigned long long st[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(st, work + 4, sizeof(st), hipMemcpyDeviceToHost));
    if ((uint32_t)st[1] != 0u) return fail(-6, "record pool exhausted in a launch sized for the worst case (%llu blocks)", (unsigned long long)ctx->pool_blocks);
    ctx->inflight.clear();
    return 0;
}

/* Samples [first_sample, first_sample + num_samples) of the tile in n_blocks row blocks, each with ALL the samples before the next
 * block starts: a pair then covers fewer pixels and more samples of each (the record pool holds the same number of paths), and the
 * film -- read and written once per pair, 3328 bytes a pixel -- is touched that much less often. Samples per pair: five eighths of
 * what the pool's size would allow, because it is sized from the tile's AVERAGE path and the rows of an image that hold its objects
 * run above that (a block that runs out anyway is rendered again: redo_batches). done[k], if asked for, is recorded behind block k. */
static int enqueue_blocks(drt_context *ctx, uint32_t first_sample, uint32_t num_samples, uint32_t n_blocks, std::vector<hipEvent_t> *done)
{
    const uint32_t H = ctx->params.tile_h, W = ctx->params.tile_w;
    const uint32_t per = (H + n_blocks - 1) / n_blocks;
    const uint32_t fit = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint32_t>(num_samples, 4096), ctx->n_pix * (uint64_t)ctx->batch_spp * 5 / 8 / ((uint64_t)per * W)));
    /* ... in pairs of equal size: 256 samples where 125 fit go out as 86 + 85 + 85, not 125 + 125 + 6 (a launch of 6 samples fills the chip for a moment only) */
    const uint32_t n_pairs = (num_samples + fit - 1) / fit;
    const uint32_t n_blk = n_pairs ? (num_samples + n_pairs - 1) / n_pairs : 1u;
    for (uint32_t r0 = 0; r0 < H; r0 += per)
    {
        const uint32_t rows = std::min(per, H - r0);
        for (uint32_t at = 0; at < num_samples; at += n_blk)
        {
            const uint32_t n = std::min(n_blk, num_samples - at);
            int rc = enqueue_pair(ctx, first_sample + at, n, 0, r0, rows, n);
            if (rc) return rc;
        }
        if (done)
        {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            done->push_back(e);
            HIP_TRY(hipEventRecord(e, ctx->stream));
        }
    }
    return 0;
}

extern "C" int drt_render(drt_context *ctx, uint32_t first_sample, uint32_t num_samples)
{
    if (!ctx) return fail(-1, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    (void)hipGetLastError(); /* drop a stale error of an earlier, unrelated call: launches below are checked against a clean slate */
    const drt_params &p = ctx->params;
    if (p.flags & DRT_FLAG_RECORD_HITS)
    {
        uint64_t need = ctx->n_pix * (uint64_t)num_samples;
        if (need > ctx->hits_capacity)
        {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_hits);
            ctx->d_hits = nullptr;
            HIP_TRY(hipMalloc((void **)&ctx->d_hits, std::max<uint64_t>(need, 1) * p.max_depth * sizeof(int32_t)));
            ctx->hits_capacity = need;
        }
        ctx->hits_samples = num_samples;
    }
    /* keep the number of pending timing events bounded */
    if (ctx->ev_used >= 3 * 256)
    {
        int rc = drt_synchronize(ctx);
        if (rc) return rc;
    }
    /* More samples than one pair takes over the whole tile, on a tile so large that a pair takes few samples of each pixel: row blocks,
     * so that the film is passed over fewer times (config 5, 4096^2 at 16 samples a pair: 208 bytes of film per path; in blocks
     * 1141 -> 1196 Mpaths/s). Where a pair takes 32 samples or more the film is a small part of the traffic and the smaller launches
     * cost more than they save (1024^2, 64 a pair: 1493 -> 1452). Not with the hit log on, which is laid out by sample of the whole tile. */
    if (!(p.flags & DRT_FLAG_RECORD_HITS) && num_samples > ctx->batch_spp && ctx->batch_spp < 32 && p.tile_h >= 64 && !getenv("DRT_NO_ROW_BLOCKS"))
    {
        const uint64_t want = ((uint64_t)num_samples * 8 + (uint64_t)ctx->batch_spp * 5 - 1) / ((uint64_t)ctx->batch_spp * 5); /* blocks for one pair each */
        const uint32_t n_blocks = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(want, 16), p.tile_h / 16);
        if (n_blocks > 1) return enqueue_blocks(ctx, first_sample, num_samples, n_blocks, nullptr);
    }
    /* (pairs of equal size here too: 100 samples where 64 fit are 50 + 50) */
    const uint32_t n_pairs = (num_samples + ctx->batch_spp - 1) / ctx->batch_spp;
    const uint32_t each = n_pairs ? (num_samples + n_pairs - 1) / n_pairs : 1u;
    for (uint32_t done = 0; done < num_samples; done += each)
    {
        int rc = enqueue_pair(ctx, first_sample + done, std::min(each, num_samples - done), done);
        if (rc) return rc;
    }
    return 0;
}

extern "C" int drt_synchronize(drt_context *ctx)
{
    if (!ctx) return fail(-1, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    int rc = collect_timings(ctx);
    if (rc) return rc;
    if (!ctx->inflight.empty())
    {
        unsigned long long st[4] = {0, 0, 0, 0}; /* pool cursor, overflow flag, last complete pair, peak of the cursor */
        HIP_TRY(hipMemcpy(st, ctx->d_counters + DRT_NUM_COUNTERS + 4, sizeof(st), hipMemcpyDeviceToHost));
        ctx->pool_peak = std::max<uint64_t>(ctx->pool_peak, st[3]);
        if ((uint32_t)st[1] != 0u)
        {
            if ((rc = redo_batches(ctx, st[2]))) return rc;
            if ((rc = collect_timings(ctx))) return rc;
        }
        ctx->inflight.clear();
    }
    return 0;
}

extern "C" int drt_reset_film(drt_context *ctx)
{
    if (!ctx) return fail(-1, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = drt_synchronize(ctx);
    if (rc) return rc;
    const size_t S = ctx->dsc.S;
    HIP_TRY(hipMemsetAsync(ctx->d_pixels, 0, pixels_bytes(ctx), ctx->stream));
    if (ctx->d_avgs) HIP_TRY(hipMemsetAsync(ctx->d_avgs, 0, (size_t)ctx->n_pix * S * 8, ctx->stream));
    if (ctx->d_vars) HIP_TRY(hipMemsetAsync(ctx->d_vars, 0, (size_t)ctx->n_pix * S * 8, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->d_counters, 0, DRT_COUNTER_WORDS * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->trace_ms = ctx->shade_ms = 0.0;
    ctx->timed_pairs = 0;
    ctx->min_sample_ms = ctx->max_sample_ms = ctx->avg_sample_ms = 0.0;
    return 0;
}

extern "C" int drt_film_device_ptrs(drt_context *ctx, void **d_pixels, void **d_avgs, void **d_vars)
{
    if (!ctx) return fail(-1, "null context");
    if (d_pixels) *d_pixels = ctx->d_pixels;
    if (d_avgs) *d_avgs = ctx->d_avgs;
    if (d_vars) *d_vars = ctx->d_vars;
    return 0;
}

extern "C" int drt_read_film(drt_context *ctx, double *pixels, double *avgs, double *vars)
{
    if (!ctx) return fail(-1, "null context");
    int rc = drt_synchronize(ctx);
    if (rc) return rc;
    const size_t S = ctx->dsc.S;
    if (ctx->xyz_mode && (avgs || vars)) return fail(-4, "the XYZ film has no mean / variance buffers");
    if (pixels) HIP_TRY(hipMemcpy(pixels, ctx->d_pixels, pixels_bytes(ctx), hipMemcpyDeviceToHost));
    if (avgs) HIP_TRY(hipMemcpy(avgs, ctx->d_avgs, (size_t)ctx->n_pix * S * 8, hipMemcpyDeviceToHost));
    if (vars) HIP_TRY(hipMemcpy(vars, ctx->d_vars, (size_t)ctx->n_pix * S * 8, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int drt_write_film(drt_context *ctx, const double *pixels, const double *avgs, const double *vars)
{
    if (!ctx) return fail(-1, "null context");
    int rc = drt_synchronize(ctx);
    if (rc) return rc;
    const size_t S = ctx->dsc.S;
    if (ctx->xyz_mode && (avgs || vars)) return fail(-4, "the XYZ film has no mean / variance buffers");
    if (pixels) HIP_TRY(hipMemcpy(ctx->d_pixels, pixels, pixels_bytes(ctx), hipMemcpyHostToDevice));
    if (avgs) HIP_TRY(hipMemcpy(ctx->d_avgs, avgs, (size_t)ctx->n_pix * S * 8, hipMemcpyHostToDevice));
    if (vars) HIP_TRY(hipMemcpy(ctx->d_vars, vars, (size_t)ctx->n_pix * S * 8, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int drt_read_xyz(drt_context *ctx, double *xyz)
{
    if (!ctx || !xyz) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    /* first the film itself: drt_synchronize is where a record pool that ran out is noticed and those samples are rendered again --
     * a conversion enqueued before it would read the incomplete film */
    int rc = drt_synchronize(ctx);
    if (rc) return rc;
    if (!ctx->d_xyz) HIP_TRY(hipMalloc((void **)&ctx->d_xyz, (size_t)ctx->n_pix * 3 * 8));
    uint32_t grid = (uint32_t)((ctx->n_pix + 255) / 256);
    if (ctx->xyz_mode)
        hipLaunchKernelGGL(drt_xyz_finish_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->dsc, ctx->cmf_rw, ctx->cmf_y, ctx->interval,
                           ctx->n_pix, ctx->d_pixels, ctx->d_xyz);
    else
        hipLaunchKernelGGL(drt_film_xyz_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->dsc, ctx->cmf_rw, ctx->cmf_x, ctx->cmf_y,
                           ctx->cmf_z, ctx->interval, ctx->n_pix, ctx->d_pixels, ctx->d_xyz);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(xyz, ctx->d_xyz, (size_t)ctx->n_pix * 3 * 8, hipMemcpyDeviceToHost));
    return 0;
}

/* one film buffer as BMP pixel bytes, left on the device in ctx->d_bgra */
static int film_to_bgra(drt_context *ctx, int which)
{
    if (which < 0 || which > 2) return fail(-1, "which = %d: 0 sum, 1 mean, 2 variance", which);
    if (ctx->xyz_mode) return fail(-4, "the XYZ film keeps no spectra: render in DRT_MODE_SPECTRAL for .bmp pixels");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = drt_synchronize(ctx); /* the complete film first (see drt_read_xyz) */
    if (rc) return rc;
    if (!ctx->d_bgra) HIP_TRY(hipMalloc((void **)&ctx->d_bgra, (size_t)ctx->n_pix * 4));
    const double *film = which == 0 ? ctx->d_pixels : which == 1 ? ctx->d_avgs : ctx->d_vars;
    uint32_t grid = (uint32_t)((ctx->n_pix + 255) / 256);
    hipLaunchKernelGGL(drt_film_bgra_kernel, dim3(grid), dim3(256), 0, ctx->stream, ctx->dsc, ctx->cmf_rw, ctx->cmf_x, ctx->cmf_y,
                       ctx->cmf_z, ctx->interval, ctx->n_pix, film, which, ctx->d_bgra);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int drt_read_bgra(drt_context *ctx, int which, uint8_t *bgra)
{
    if (!ctx || !bgra) return fail(-1, "null argument");
    int rc = film_to_bgra(ctx, which);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(bgra, ctx->d_bgra, (size_t)ctx->n_pix * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int drt_read_hit_indices(drt_context *ctx, int32_t *dst, uint64_t capacity_paths)
{
    if (!ctx || !dst) return fail(-1, "null argument");
    if (!(ctx->params.flags & DRT_FLAG_RECORD_HITS) || !ctx->d_hits) return fail(-4, "hit recording is off (DRT_FLAG_RECORD_HITS)");
    int rc = drt_synchronize(ctx);
    if (rc) return rc;
    uint64_t n = std::min<uint64_t>(capacity_paths, ctx->n_pix * (uint64_t)ctx->hits_samples);
    HIP_TRY(hipMemcpy(dst, ctx->d_hits, n * ctx->params.max_depth * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int drt_get_stats(drt_context *ctx, drt_stats *out)
{
    if (!ctx || !out) return fail(-1, "null argument");
    int rc = drt_synchronize(ctx);
    if (rc) return rc;
    unsigned long long c[DRT_NUM_COUNTERS];
    HIP_TRY(hipMemcpy(c, ctx->d_counters, sizeof(c), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    out->paths = c[0];
    out->closest_hit_scans = c[1];
    out->shaded_vertices = c[2];
    out->shadow_scans = c[3];
    out->rng_draws = c[4];
    out->trace_ms = ctx->trace_ms;
    out->shade_ms = ctx->shade_ms;
    out->total_ms = ctx->trace_ms + ctx->shade_ms;
    out->record_pool_blocks = ctx->pool_blocks;
    out->record_pool_peak = ctx->pool_peak;
    out->record_block_bytes = ctx->block_words * 8;
    out->redone_launches = (uint32_t)ctx->redone_batches;
    out->launches = (uint32_t)std::min<uint64_t>(ctx->timed_pairs, 0xFFFFFFFFull);
    out->path_flags = (ctx->bvh_pipeline ? DRT_PATH_BVH : 0u) | ((ctx->trace_tail && ctx->d_tail_stage) ? DRT_PATH_TRACE_TAIL : 0u);
    out->min_sample_ms = ctx->min_sample_ms;
    out->max_sample_ms = ctx->max_sample_ms;
    out->avg_sample_ms = ctx->avg_sample_ms;
    return 0;
}

extern "C" uint32_t drt_batch_spp(drt_context *ctx) { return ctx ? ctx->batch_spp : 0; }

static double wall_ms()
{
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

extern "C" int drt_render_tile(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                               double *dst_pixels, double *dst_avgs, double *dst_vars, drt_stats *stats)
{
    g_last_error.clear();
    const bool xyz = params && params->mode == DRT_MODE_XYZ; /* then dst_pixels is [n][8] and the other two are not used */
    if (!dst_pixels || (!xyz && (!dst_avgs || !dst_vars))) return fail(-1, "null film buffer");
    if (xyz) dst_avgs = dst_vars = nullptr;
    const bool verbose = getenv("DRT_VERBOSE") != nullptr || getenv("DRT_TIMING") != nullptr;
    double t[6] = {wall_ms(), 0, 0, 0, 0, 0};
    drt_context *ctx = drt_create(scene, camera, params);
    if (!ctx) return -1;
    t[1] = wall_ms();
    int rc = 0;
    do
    {
        /* accumulate INTO the caller's buffers: start from their contents (unless the caller vouches they are zero) */
        if (!(params->flags & DRT_FLAG_FILM_ZERO) && (rc = drt_write_film(ctx, dst_pixels, dst_avgs, dst_vars))) break;
        t[2] = wall_ms();
        /* The tile goes out in row blocks, each with all its samples (as many per kernel pair as the record pool was sized for: a
         * quarter of the rows takes four times the samples), so that a block's film rows cross PCIe while the next block renders:
         * what is left exposed of the 1.7 GB download is its last quarter. (A pool that runs out on the way: drt_synchronize
         * renders again from there, and the film is fetched whole.) */
        uint32_t n_blocks = 8; /* 1024^2 x 256 spp, wall: 226 ms in one piece, 219 / 214 / 206 ms in 2 / 4 / 8 blocks (kernels 187 -> 197 ms: smaller launches) */
        if (getenv("DRT_ONESHOT_BLOCKS")) n_blocks = (uint32_t)std::max(1, atoi(getenv("DRT_ONESHOT_BLOCKS")));
        const uint32_t per = (ctx->params.tile_h + n_blocks - 1) / n_blocks; /* rows per block */
        const bool blocks_ok = params->spp != 0 && n_blocks > 1 && !(params->flags & DRT_FLAG_RECORD_HITS) && ctx->params.tile_h >= 16 * n_blocks &&
                               (uint64_t)ctx->params.tile_w * ctx->params.tile_h >= (1u << 18);
        std::vector<hipEvent_t> block_done;
        if (!blocks_ok)
        {
            if ((rc = drt_render(ctx, params->first_sample, params->spp))) break;
        }
        else if ((rc = enqueue_blocks(ctx, params->first_sample, params->spp, n_blocks, &block_done))) break;
        if (params->flags & DRT_FLAG_FILM_ZERO)
        {
            /* Buffers that come zero-filled (the reference's alloc()) have usually never been touched: the download would then
             * pay a page fault per 4 KB (100 ms for 1.7 GB instead of 30). The kernels are running and this thread has nothing to
             * do, so it touches the pages now -- writing the zero the caller vouched for into one byte of each. */
            const size_t n = (size_t)params->tile_w * params->tile_h, S = scene->num_wavelengths;
            char *bufs[3] = {(char *)dst_pixels, (char *)dst_avgs, (char *)dst_vars};
            const size_t sizes[3] = {n * (xyz ? (size_t)XYZ_FILM_WORDS : S + 1) * 8, n * S * 8, n * S * 8};
            for (int k = 0; k < 3; k += 1)
            {
                if (!bufs[k]) continue;
                for (size_t off = 0; off < sizes[k]; off += 4096) ((volatile char *)bufs[k])[off] = 0;
                if (sizes[k]) ((volatile char *)bufs[k])[sizes[k] - 1] = 0;
            }
        }
        if (verbose && !blocks_ok && (rc = drt_synchronize(ctx))) break;
        t[3] = wall_ms();
        bool fetched = false;
        if (blocks_ok && block_done.size() == (size_t)((ctx->params.tile_h + per - 1) / per))
        {
            const size_t W = ctx->params.tile_w, S = scene->num_wavelengths;
            const size_t words[3] = {xyz ? (size_t)XYZ_FILM_WORDS : S + 1, S, S};
            double *host[3] = {dst_pixels, dst_avgs, dst_vars};
            double *dev[3] = {ctx->d_pixels, ctx->d_avgs, ctx->d_vars};
            fetched = true;
            for (size_t k = 0; k < block_done.size() && fetched; k += 1)
            {
                if (hipEventSynchronize(block_done[k]) != hipSuccess) { fetched = false; break; }
                unsigned long long st[4] = {0, 0, 0, 0}; /* pool cursor, overflow flag, last complete pair, peak */
                if (hipMemcpy(st, ctx->d_counters + DRT_NUM_COUNTERS + 4, sizeof(st), hipMemcpyDeviceToHost) != hipSuccess || (uint32_t)st[1] != 0u) { fetched = false; break; }
                const size_t r0 = k * per, rows = std::min<size_t>(per, ctx->params.tile_h - r0);
                for (int f = 0; f < 3; f += 1)
                    if (host[f] && dev[f] && hipMemcpy(host[f] + r0 * W * words[f], dev[f] + r0 * W * words[f], rows * W * words[f] * 8, hipMemcpyDeviceToHost) != hipSuccess) fetched = false;
            }
        }
        for (hipEvent_t e : block_done) (void)hipEventDestroy(e);
        if ((rc = drt_synchronize(ctx))) break; /* timings, and what a pool that ran out left undone */
        if (!fetched && (rc = drt_read_film(ctx, dst_pixels, dst_avgs, dst_vars))) break;
        if (stats && (rc = drt_get_stats(ctx, stats))) break;
        t[4] = wall_ms();
    } while (0);
    std::string keep = g_last_error;
    drt_destroy(ctx);
    g_last_error = keep;
    t[5] = wall_ms();
    if (verbose && rc == 0)
        fprintf(stderr, "drt_render_tile: create %.1f ms, film upload %.1f ms, render %.1f ms, film download %.1f ms, destroy %.1f ms\n",
                t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]);
    return rc;
}

/* ---------------------------------------------------------------------------------------------- */
/* Device groups: one host thread, several GPUs                                                     */

struct drt_group
{
    std::vector<drt_context *> ctx; /* nullptr for a device that got no rows */
    std::vector<uint32_t>      rows;
    uint32_t tile_w = 0, tile_h = 0, S = 0;
    bool     xyz_mode = false;
};

extern "C" void drt_group_destroy(drt_group *g)
{
    if (!g) return;
    std::string keep = g_last_error;
    for (drt_context *c : g->ctx) drt_destroy(c);
    delete g;
    g_last_error = keep;
}

extern "C" drt_group *drt_group_create(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                                       const int32_t *devices, uint32_t n_devices)
{
    g_last_error.clear();
    if (!scene || !camera || !params)
    {
        fail(-1, "null argument");
        return nullptr;
    }
    if (n_devices == 0)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        {
            (void)hipGetLastError();
            fail(-5, "no HIP device is visible");
            return nullptr;
        }
        n_devices = (uint32_t)n;
        devices = nullptr;
    }
    if (n_devices > 64)
    {
        fail(-1, "more than 64 devices in a group");
        return nullptr;
    }
    if (params->flags & DRT_FLAG_RECORD_HITS)
    {
        fail(-1, "hit recording is per context: use drt_create for it");
        return nullptr;
    }
    drt_group *g = new drt_group();
    g->tile_w = params->tile_w;
    g->tile_h = params->tile_h;
    g->S = scene->num_wavelengths;
    g->xyz_mode = params->mode == DRT_MODE_XYZ;
    for (uint32_t k = 0; k < n_devices; k += 1)
    {
        drt_params p = *params;
        p.device = devices ? devices[k] : (int32_t)k;
        p.y0 = params->y0 + k * params->row_stride;
        p.row_stride = params->row_stride * n_devices;
        p.tile_h = params->tile_h > k ? (params->tile_h - k + n_devices - 1) / n_devices : 0;
        g->rows.push_back(p.tile_h);
        if (p.tile_h == 0)
        {
            g->ctx.push_back(nullptr);
            continue;
        }
        drt_context *c = drt_create(scene, camera, &p);
        if (!c)
        {
            drt_group_destroy(g);
            return nullptr;
        }
        g->ctx.push_back(c);
    }
    return g;
}

extern "C" uint32_t drt_group_size(drt_group *g) { return g ? (uint32_t)g->ctx.size() : 0; }

extern "C" int drt_group_render(drt_group *g, uint32_t first_sample, uint32_t num_samples)
{
    if (!g) return fail(-1, "null group");
    for (drt_context *c : g->ctx)
        if (c)
        {
            int rc = drt_render(c, first_sample, num_samples); /* asynchronous: the devices run side by side */
            if (rc) return rc;
        }
    return 0;
}

extern "C" int drt_group_synchronize(drt_group *g)
{
    if (!g) return fail(-1, "null group");
    for (drt_context *c : g->ctx)
        if (c)
        {
            int rc = drt_synchronize(c);
            if (rc) return rc;
        }
    return 0;
}

/* rows k, k+n, ... of a whole-tile host buffer <-> device k's contiguous rows: one strided copy */
static int group_copy(drt_group *g, double *host, int which, bool to_device)
{
    if (!host) return 0;
    const size_t n = g->ctx.size();
    const size_t C = which == 0 ? (g->xyz_mode ? (size_t)XYZ_FILM_WORDS : (size_t)g->S + 1) : (size_t)g->S;
    if (g->xyz_mode && which != 0) return fail(-4, "the XYZ film has no mean / variance buffers");
    const size_t row_bytes = (size_t)g->tile_w * C * 8;
    for (size_t k = 0; k < n; k += 1)
    {
        drt_context *c = g->ctx[k];
        if (!c) continue;
        HIP_TRY(hipSetDevice(c->device));
        void *dev = which == 0 ? (void *)c->d_pixels : which == 1 ? (void *)c->d_avgs : (void *)c->d_vars;
        char *h = (char *)host + k * row_bytes;
        if (to_device)
            HIP_TRY(hipMemcpy2D(dev, row_bytes, h, n * row_bytes, row_bytes, g->rows[k], hipMemcpyHostToDevice));
        else
            HIP_TRY(hipMemcpy2D(h, n * row_bytes, dev, row_bytes, row_bytes, g->rows[k], hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int drt_group_read_film(drt_group *g, double *pixels, double *avgs, double *vars)
{
    int rc = drt_group_synchronize(g);
    if (rc) return rc;
    if ((rc = group_copy(g, pixels, 0, false))) return rc;
    if ((rc = group_copy(g, avgs, 1, false))) return rc;
    return group_copy(g, vars, 2, false);
}

extern "C" int drt_group_write_film(drt_group *g, const double *pixels, const double *avgs, const double *vars)
{
    int rc = drt_group_synchronize(g);
    if (rc) return rc;
    if ((rc = group_copy(g, const_cast<double *>(pixels), 0, true))) return rc;
    if ((rc = group_copy(g, const_cast<double *>(avgs), 1, true))) return rc;
    return group_copy(g, const_cast<double *>(vars), 2, true);
}

extern "C" int drt_group_read_bgra(drt_group *g, int which, uint8_t *bgra)
{
    if (!g || !bgra) return fail(-1, "null argument");
    const size_t n = g->ctx.size();
    const size_t row_bytes = (size_t)g->tile_w * 4;
    for (size_t k = 0; k < n; k += 1)
    {
        drt_context *c = g->ctx[k];
        if (!c) continue;
        int rc = film_to_bgra(c, which);
        if (rc) return rc;
        HIP_TRY(hipMemcpy2D(bgra + k * row_bytes, n * row_bytes, c->d_bgra, row_bytes, row_bytes, g->rows[k], hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int drt_group_get_stats(drt_group *g, drt_stats *out)
{
    if (!g || !out) return fail(-1, "null argument");
    memset(out, 0, sizeof(*out));
    for (drt_context *c : g->ctx)
    {
        if (!c) continue;
        drt_stats st;
        int rc = drt_get_stats(c, &st);
        if (rc) return rc;
        out->paths += st.paths;
        out->closest_hit_scans += st.closest_hit_scans;
        out->shaded_vertices += st.shaded_vertices;
        out->shadow_scans += st.shadow_scans;
        out->rng_draws += st.rng_draws;
        out->trace_ms = std::max(out->trace_ms, st.trace_ms);
        out->shade_ms = std::max(out->shade_ms, st.shade_ms);
        out->total_ms = std::max(out->total_ms, st.total_ms);
        out->record_pool_blocks += st.record_pool_blocks; /* all the pools together; the fullest any of them has been */
        out->record_pool_peak = std::max(out->record_pool_peak, st.record_pool_peak);
        out->record_block_bytes = st.record_block_bytes;
        out->redone_launches += st.redone_launches;
        /* the devices run side by side: a sample pass over the whole tile takes as long as the slowest device's share */
        out->launches += st.launches;
        out->path_flags |= st.path_flags;
        out->min_sample_ms = std::max(out->min_sample_ms, st.min_sample_ms);
        out->max_sample_ms = std::max(out->max_sample_ms, st.max_sample_ms);
        out->avg_sample_ms = std::max(out->avg_sample_ms, st.avg_sample_ms);
    }
    return 0;
}

extern "C" int drt_render_tile_multi(const drt_scene *scene, const drt_camera *camera, const drt_params *params,
                                     const int32_t *devices, uint32_t n_devices, double *dst_pixels, double *dst_avgs,
                                     double *dst_vars, drt_stats *stats)
{
    g_last_error.clear();
    const bool xyz = params && params->mode == DRT_MODE_XYZ;
    if (!dst_pixels || (!xyz && (!dst_avgs || !dst_vars))) return fail(-1, "null film buffer");
    if (xyz) dst_avgs = dst_vars = nullptr;
    drt_group *g = drt_group_create(scene, camera, params, devices, n_devices);
    if (!g) return -1;
    int rc = 0;
    do
    {
        if (!(params->flags & DRT_FLAG_FILM_ZERO) && (rc = drt_group_write_film(g, dst_pixels, dst_avgs, dst_vars))) break;
        if ((rc = drt_group_render(g, params->first_sample, params->spp))) break;
        if ((rc = drt_group_read_film(g, dst_pixels, dst_avgs, dst_vars))) break;
        if (stats && (rc = drt_group_get_stats(g, stats))) break;
    } while (0);
    drt_group_destroy(g);
    return rc;
}

extern "C" int drt_selftest_arith(int device, int op, const double *a, const double *b, double *out, uint64_t n)
{
    g_last_error.clear();
    HIP_TRY(hipSetDevice(device));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    size_t out_n = (op == 2) ? 2 * n : n;
    HIP_TRY(hipMalloc((void **)&da, std::max<uint64_t>(n, 1) * 8));
    HIP_TRY(hipMalloc((void **)&db, std::max<uint64_t>(n, 1) * 8));
    HIP_TRY(hipMalloc((void **)&dout, std::max<size_t>(out_n, 1) * 8));
    HIP_TRY(hipMemcpy(da, a, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db, b, n * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(drt_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, op, da, db, dout, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, 0));
    HIP_TRY(hipDeviceSynchronize());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (getenv("DRT_VERBOSE")) fprintf(stderr, "drt_selftest_arith op %d n %llu: %.3f ms\n", op, (unsigned long long)n, ms);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    HIP_TRY(hipMemcpy(out, dout, out_n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    return 0;
}

extern "C" int drt_selftest_unit(int device, int func, const double *in, uint32_t in_stride, double *out, uint32_t out_stride, uint64_t n)
{
    g_last_error.clear();
    if (func < 0 || func >= DRT_UNIT_COUNT) return fail(-1, "unknown unit function %d", func);
    if (!in || !out || in_stride == 0 || out_stride == 0) return fail(-1, "null argument");
    /* what each function reads and writes per record: a caller with narrower records would make the kernel read past its buffers */
    static const uint32_t need_in[DRT_UNIT_COUNT] = {10, 18, 6, 8, 6, 1, 1, 7, 10, 3, 4, 1, 12};
    static const uint32_t need_out[DRT_UNIT_COUNT] = {1, 1, 3, 3, 9, 4, 4, 1, 1, 1, 1, 2, 1};
    if (in_stride < need_in[func] || out_stride < need_out[func])
        return fail(-1, "unit function %d needs %u doubles in and %u out per record", func, need_in[func], need_out[func]);
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(device));
    double *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc((void **)&din, n * in_stride * 8));
    HIP_TRY(hipMalloc((void **)&dout, n * out_stride * 8));
    HIP_TRY(hipMemcpy(din, in, n * in_stride * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(dout, 0, n * out_stride * 8));
    hipLaunchKernelGGL(drt_unit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, func, din, in_stride, dout, out_stride, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout, n * out_stride * 8, hipMemcpyDeviceToHost));
    (void)hipFree(din);
    (void)hipFree(dout);
    return 0;
}
