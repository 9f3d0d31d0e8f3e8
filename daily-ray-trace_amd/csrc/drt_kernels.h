/*
 * drt_kernels.h -- the two hot kernels and their data layouts.
 *
 *   drt_trace_kernel   one ray per lane. sample_scene's ray generation + cast_ray's geometry:
 *                      closest hit, next-event light samples + shadow rays, BDSF coefficient
 *                      evaluation, direction sampling, xorshift RNG. Scene primitives, lights and
 *                      materials are SoA in HBM, staged into LDS once per workgroup. Lanes whose
 *                      path has ended pick up the next path (wave ballot + prefix count), so a
 *                      wave stays full until the batch runs out.
 *                      Output: one compact record per shaded vertex (layout below).
 *   drt_shade_kernel   one wavelength per lane, one pixel per wave. Replays the vertex records of
 *                      the pixel's samples over the wavelengths (SPD tables in LDS), i.e. bdsf()
 *                      sums, light weighting, throughput products, emissive hits, vignette; then
 *                      the film update of render_image (sum, filter sum, running mean / variance).
 *                      Throughput and radiance live in registers; only the film touches HBM.
 *
 * Reference lines each piece follows are cited at the piece.
 */
#pragma once

#include "drt_device.h"
#include "../../include/drt_hip.h"

/* ---------------------------------------------------------------------------------------------- */
/* Device-side scene                                                                               */

/* surface SoA field indices: surf[f * n_surf + i] */
enum
{
    SF_PX = 0, SF_PY, SF_PZ, SF_RADIUS,
    SF_NX, SF_NY, SF_NZ,
    SF_UNX, SF_UNY, SF_UNZ, /* u / |u| */
    SF_VNX, SF_VNY, SF_VNZ, /* v / |v| */
    SF_ULEN, SF_VLEN,
    SF_COUNT
};
/* light SoA field indices: lights[f * n_lights + l] */
enum
{
    LF_PX = 0, LF_PY, LF_PZ, LF_RADIUS,
    LF_UX, LF_UY, LF_UZ,
    LF_VX, LF_VY, LF_VZ,
    LF_PDF, /* sphere: (4 pi R) R ; plane: |u x v| ; point: 1 */
    LF_COUNT
};

/* what a material's bdsf list needs evaluated per (vertex, incoming direction) */
enum
{
    NEED_GLOSSY = 1u, /* bp_glossy_bdsf: pow term           */
    NEED_EQR    = 2u, /* exact test against the mirror direction (mirror / fs_conductor / fs_dielectric_reflectance) */
    NEED_EQT    = 4u, /* exact test against the refracted direction (fs_dielectric_transmittance)                  */
    NEED_CT     = 8u  /* ct_conductor_bdsf: half-vector cosine + GGX coefficient */
};

/* Bounding-volume hierarchy over the surfaces of a large scene. It only PRUNES the surface
 * scan: leaves call the same intersectors, and the closest hit is the minimum distance with the lowest surface
 * index on ties -- exactly what the reference's linear scan with its strict `<` returns
 * (src/daily_ray_trace.c:340-364) -- so hit indices do not change by a bit. Boxes are padded, so a surface whose
 * COMPUTED distance is finite always lies inside its boxes. */
struct BvhNode
{
    float   lo[2][3], hi[2][3]; /* the two children's boxes, rounded OUTWARD to f32 (they only prune; tests run in f64) */
    int32_t child[2];           /* the traversal's reference to the child, ready to use: node index (>= 0), a leaf (bvh_leaf_ref: first
                                   slot in bvh_leaf and count), or BVH_DONE for no child */
    int32_t count[2];           /* 0: inner child, > 0: leaf with that many surfaces, < 0: no child (host side and statistics) */
};                              /* 64 bytes: half a cache line per visit */

/* A surface as the leaves see it: a sphere's four numbers gathered into one 64-byte record, records of a leaf adjacent --
 * a lane walking the tree on its own touches half a cache line per sphere instead of one line per SoA field. */
struct BvhLeafPrim
{
    uint32_t index, type;
    double   f[4]; /* a sphere's centre and radius; a plane's many fields are read from the SoA tables (planes are few) */
    float    c32[3];  /* the centre again in f32, and the radius plus the f32 test's error bound, rounded up (planes: 0 and +inf): */
    float    reach32; /* sphere_certainly_missed() spares the f64 intersector most of the spheres a ray passes by */
    double   pad;
};             /* 64 bytes */

/* Rows tabulated for a vertex's pair of media (drt_device.h, *_reflectance_sel), as bits 48-63 of record word 2:
 * PAIR_NONE, or the first row's index (< 0x8000), with PAIR_CONDUCTOR set when the rows are a conductor's (cA, cB: two rows)
 * and clear when they are a dielectric's (rel_sq: one row). A material gets rows only when its Fresnel functions are all of
 * one kind, so the kind says which of the *_sel forms every function of the vertex's list takes. */
#define PAIR_NONE 0xFFFFu
#define PAIR_CONDUCTOR 0x8000u
struct DevMaterial
{
    uint32_t is_black_body, is_emissive, num_bdsfs, dir_func;
    uint32_t needs;
    uint16_t pair_out, pair_in; /* the pair field for (incident = base material, transmit = this one) and for the other way round
                                   (a ray leaving a sphere of this material) */
    int32_t  emission_spd, diffuse_spd, glossy_spd, mirror_spd, refract_spd, extinct_spd;
    double   shininess, roughness;
    double   refract_i0, refract_i1; /* refract_spd at the two samples around trans_wl (value_at_wl) */
    uint32_t bdsfs[DRT_MAX_BDSFS];
    uint64_t bdsf_packed; /* bdsfs[] as 4-bit fields */
    uint32_t vertex_flags; /* FLAG_PLASTIC when the list is exactly {bp_diffuse_bdsf, bp_glossy_bdsf} */
    uint32_t pad1;
};

struct DevScene
{
    uint32_t n_surf, n_lights, n_mat, S;
    uint32_t n_spd, base_mat, escape_mat, trans_i0;
    double   trans_wl, trans_w0, trans_w1; /* value_at_wl at 630 nm: i0, i0+1 and their wavelengths */
    const double      *surf;       /* [SF_COUNT][n_surf] */
    const uint32_t    *surf_type;  /* [n_surf] */
    const uint32_t    *surf_mat;   /* [n_surf] */
    const double      *lights;     /* [LF_COUNT][n_lights] */
    const uint32_t    *light_type; /* [n_lights] */
    const uint32_t    *light_mat;  /* [n_lights] material of the emissive surface */
    const DevMaterial *mats;       /* [n_mat] */
    const double      *spds;       /* [n_spd][S]: scene rows, derived diffuse/pi rows, one zero row */
    const BvhNode     *bvh_nodes;  /* NULL: scan every surface */
    const BvhLeafPrim *bvh_leaf;   /* the surfaces grouped by leaf */
};

struct DevCamera
{
    V3     forward, right, up, aperture_position, film_bottom_left;
    double aperture_radius, focal_depth, pixel_width, pixel_height;
};

/* ---------------------------------------------------------------------------------------------- */
/* Vertex records (trace -> shade). All 8-byte words; self-contained, so the shade kernel never  */
/* looks a material up: the trace kernel resolves it into a packed BDSF list + SPD indices.        */
/*                                                                                                  */
/* Records live in a POOL of blocks of four vertices (4 * vertex_words words). A path takes a block */
/* when it reaches its 1st, 5th, 9th ... vertex -- one atomic add per wave and loop iteration, the  */
/* lanes' shares by ballot + prefix count (pool_alloc) -- so the pool holds what paths really use   */
/* (1.65 of 8 possible vertices on the Cornell frame) instead of max_depth slots per path. The     */
/* path's header says where its blocks are. A pool that runs out raises `overflow`: the shade       */
/* kernel of that launch and everything queued behind it become no-ops, and the host renders those  */
/* samples again with launches sized for the worst case (drt_launcher.hip, redo_batches).           */
/*                                                                                                  */
/*  path header (4 words, own array, indexed by slot = pixel * batch + sample_in_batch):           */
/*     w0 = n_shaded (bits 0-15) | term (16-23: 0 none/escape, 1 emissive hit) | light 0 visible from vertex v < 8 (24-31) |
 *          emission SPD (32-47) | vertex v < 16 has the two-lobe plastic list (48-63)                                      */
/*     w1 = vignette (double)                                                                      */
/*     w2 = block of vertices 0-3 (bits 0-31) | block of vertices 4-7 (32-63)                       */
/*     w3 = block of vertices 8-11 (bits 0-31) | TABLE block (32-63): a pool block used as an array */
/*          of 32-bit block numbers for vertices 12-15, 16-19, ... (deep paths only)                */
/*  vertex fixed part (10 words):                                                                  */
/*     w0 = BDSF list, 4 bits per entry (the material's bdsfs[] in order)                          */
/*     w1 = num_bdsfs (0-7) | sampled-direction flags (8-15) | diffuse SPD (16-31) | glossy SPD (32-47) | mirror SPD (48-63) */
/*     w2 = incident refract SPD (0-15) | transmit refract SPD (16-31) | transmit extinct SPD (32-47) |
 *          rows tabulated for this pair of media (48-63: PAIR_NONE, or first row | PAIR_CONDUCTOR)                           */
/*     w3 = on_dot, w4 = 1/pdf of the sampled direction                                            */
/*     w5..w8 = |n.in|, glossy pow term, |n.m|, GGX coefficient (sampled direction), w9 unused      */
/*  per light (6 words):                                                                           */
/*     w0 = emission SPD (0-15) | flags (16-23), w1 = atten * area, w2..w5 = the four coefficients  */
/*  flags: bit0 in == mirror direction, bit1 in == refracted direction, bit2 light visible,        */
/*         bit3 (vertex) plain two-lobe plastic: shade kernel takes its straight-line path          */
/*  SPD indices are rows of the DEVICE table: the scene's rows, then one derived row diffuse*(1/pi) */
/*  per diffuse SPD (w1's "diffuse" field points at it), then one all-zero row that stands for a   */
/*  missing spectrum (NULL in the reference).                                                       */

#define REC_HEADER_WORDS 4
#define REC_BLOCK_SHIFT 2   /* log2 of the vertices per pool block: four, what one 64-word prefetch register of the shade kernel holds at 16 words per vertex */
#define REC_BLOCK_VERTICES (1u << REC_BLOCK_SHIFT)
#define REC_HEADER_BLOCKS 3 /* blocks named in the header itself; further ones through the table block */
#define HDR_TERM_NOT_DONE 0x80u /* header `term` byte of a path that found the pool exhausted */
#define HDR_TERM_TAIL_STAGED 0x40u /* header `term` byte, or'ed in: the trace kernel has put the path's tail wavelengths into tail_stage */
#define HDR_TERM_MASK 0x3Fu
#define REC_VERTEX_WORDS 10
#define REC_LIGHT_WORDS 6
#define FLAG_EQR 1u
#define FLAG_EQT 2u
#define FLAG_VISIBLE 4u
#define FLAG_PLASTIC 8u /* vertex flags only: the BDSF list is exactly {bp_diffuse_bdsf, bp_glossy_bdsf} */

struct TraceParams
{
    uint32_t width, height, x0, y0, tile_w, tile_h, row_stride;
    uint32_t first_sample, n_samples, max_depth, pixel_scheme, record_hits;
    uint64_t seed;
    uint64_t n_pix, n_paths;
    uint32_t vertex_words, block_words; /* a vertex record and a pool block (four vertices) in 8-byte words */
    uint32_t hits_sample_offset, batch;  /* batch = sample slots per pixel in the header array */
    uint32_t chunk, pool_blocks;         /* path ids a wave draws from the work counter at a time (multiple of 64); blocks in the pool */
    unsigned long long *pool_cursor;     /* next free block of the pool (reset before every trace launch) */
    uint32_t *overflow;                  /* set when the pool ran out: this launch's records are incomplete */
    /* the tail wavelengths (S mod 64 of them) of plastic-only paths, done here: see "Tail wavelengths in the trace kernel" */
    double       *tail_stage;            /* [n_pix * batch][tail_count] per-sample values the shade kernel's film phase reads; NULL: off */
    const double *spd_tail;              /* [n_spd][tail_count]: the SPD table's tail columns */
    uint32_t      tail_count, n_spd;
};

/* A wave's share of the pool. Waves take POOL_CHUNK blocks at a time from the global cursor and hand them to their lanes by
 * ballot + prefix count: one atomic per ~250 blocks instead of one per loop iteration (a single word takes 100-200 atomics per
 * microsecond: with one per iteration the trace kernel took three times as long), and the blocks of a wave's paths -- the
 * samples of one pixel, mostly -- stay together in memory, which the shade kernel's reads of a pixel's samples want. */
#define POOL_CHUNK 256u
struct WavePool
{
    uint32_t base, left; /* wave-uniform: first block of the current chunk not handed out yet, blocks left in it */
    uint32_t taken;      /* per lane: blocks this lane's paths have opened (statistics; spares held at the end are not in it) */
};

/* Blocks for the lanes that `need` one. Must be called by the whole wave. Returns the lane's block, or ~0u when it needed one
 * and the pool is exhausted (the caller raises tp.overflow). */
__device__ __forceinline__ uint32_t pool_alloc(const TraceParams &tp, WavePool &wp, bool need, uint32_t lane)
{
    const unsigned long long m = __ballot(need);
    if (m == 0ull) return ~0u;
    const uint32_t n = (uint32_t)__popcll(m);
    if (n > wp.left)
    {
        /* a new chunk (what is left of the old one, fewer than 64 blocks, stays unused) */
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(tp.pool_cursor, (unsigned long long)POOL_CHUNK);
        base = __shfl(base, 0);
        if (base + POOL_CHUNK > (unsigned long long)tp.pool_blocks)
        {
            wp.left = 0;
            return ~0u;
        }
        wp.base = (uint32_t)base;
        wp.left = POOL_CHUNK;
    }
    const uint32_t id = wp.base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    wp.base += n;
    wp.left -= n;
    return need ? id : ~0u;
}

/* Lanes keep one SPARE block, taken at the top of a loop iteration -- where little else is live -- whenever the path could
 * start a new block at the vertex this iteration may reach (vertex number a multiple of four); a spare that is not used stays
 * with the lane for its next path. Whole wave. Returns false for a lane whose path cannot go on (pool exhausted). */
__device__ __forceinline__ bool path_spare_block(const TraceParams &tp, WavePool &wp, bool alive, uint32_t shaded, uint32_t &spare, uint32_t &spare_tbl,
                                                 uint32_t lane)
{
    const bool want = alive && (shaded & (REC_BLOCK_VERTICES - 1u)) == 0u && spare == ~0u;
    const bool want_tbl = alive && shaded == REC_HEADER_BLOCKS * REC_BLOCK_VERTICES && spare_tbl == ~0u;
    const uint32_t id = pool_alloc(tp, wp, want, lane);
    const uint32_t tid = pool_alloc(tp, wp, want_tbl, lane);
    if (want) spare = id;
    if (want_tbl) spare_tbl = tid;
    if ((want && id == ~0u) || (want_tbl && tid == ~0u))
    {
        atomicExch(tp.overflow, 1u);
        return false;
    }
    return true;
}
/* the path starts vertex number `shaded`: when that opens a block, the spare becomes it and is noted in the header (or the table) */
__device__ __forceinline__ void path_open_vertex(const TraceParams &tp, WavePool &wp, uint64_t *__restrict__ records, uint32_t shaded, uint64_t *hdr,
                                                 uint32_t &blk, uint32_t &tbl, uint32_t &spare, uint32_t &spare_tbl)
{
    if ((shaded & (REC_BLOCK_VERTICES - 1u)) != 0u) return;
    const uint32_t b = shaded >> REC_BLOCK_SHIFT;
    blk = spare;
    spare = ~0u;
    wp.taken += 1;
    uint32_t *h32 = (uint32_t *)hdr;
    if (b < REC_HEADER_BLOCKS) h32[4 + b] = blk; /* low / high half of w2, low half of w3 */
    else
    {
        if (b == REC_HEADER_BLOCKS)
        {
            tbl = spare_tbl;
            spare_tbl = ~0u;
            h32[7] = tbl;
            wp.taken += 1;
        }
        ((uint32_t *)(records + (uint64_t)tbl * tp.block_words))[b - REC_HEADER_BLOCKS] = blk;
    }
}

struct EvalCoef
{
    double   a_in, spec, mn_dot, ct_coef;
    uint32_t flags;
};

/* counters: [0] paths [1] closest-hit scans [2] shaded vertices [3] shadow scans [4] rng draws [5] record blocks taken */
#define DRT_NUM_COUNTERS 8

/* ---------------------------------------------------------------------------------------------- */
/* Trace kernel pieces                                                                             */

struct SceneView /* pointers into LDS (or HBM when the scene does not fit) */
{
    const double      *surf;
    const uint32_t    *surf_type, *surf_mat;
    const double      *lights;
    const uint32_t    *light_type, *light_mat;
    const DevMaterial *mats;
    const BvhNode     *bvh_nodes;
    const BvhLeafPrim *bvh_leaf;
    uint32_t           n_surf, n_lights;
};

__device__ __forceinline__ V3 sf3(const SceneView &sv, int f, uint32_t i)
{
    return v3(sv.surf[(f + 0) * sv.n_surf + i], sv.surf[(f + 1) * sv.n_surf + i], sv.surf[(f + 2) * sv.n_surf + i]);
}

__device__ __forceinline__ double surface_distance(const SceneView &sv, uint32_t i, uint32_t type, V3 o, V3 d)
{
    if (type == DRT_GEO_SPHERE) return line_sphere(o, d, sf3(sv, SF_PX, i), sv.surf[SF_RADIUS * sv.n_surf + i]);
    return line_plane(o, d, sf3(sv, SF_PX, i), sf3(sv, SF_NX, i), sf3(sv, SF_UNX, i), sf3(sv, SF_VNX, i),
                      sv.surf[SF_ULEN * sv.n_surf + i], sv.surf[SF_VLEN * sv.n_surf + i]);
}

#define BVH_STACK 32 /* levels of the deepest tree the builder hands out = entries a traversal stack can need */

__device__ __forceinline__ double leaf_distance(const SceneView &sv, const BvhLeafPrim &lp, V3 o, V3 d)
{
    if (lp.type == DRT_GEO_SPHERE) return line_sphere(o, d, v3(lp.f[SF_PX], lp.f[SF_PY], lp.f[SF_PZ]), lp.f[SF_RADIUS]);
    return surface_distance(sv, lp.index, lp.type, o, d);
}

/* A ray as the box tests see it: origin and reciprocal direction in f32. The boxes only PRUNE, so their test may be as coarse
 * as it likes as long as it never rejects a box the exact test would accept. Error budget of t = lo * inv - o * inv in f32
 * (u = 2^-24): the conversions of o and 1/d, the product o * inv and the fused multiply-add together move a slab plane by at
 * most 2u |o| + 2u |lo - o| in position, i.e. < 4.6e-5 for coordinates within 64; the builder pads every stored box by
 * 2^-19 x (largest coordinate of the scene and the camera) on top of its f64 padding, more than twice that (see
 * BvhBuilder::build).
 * An axis the ray is (nearly) parallel to: 1/d would be +-inf (or so large that lo * inv overflows), and then ONE of the two
 * slab planes gives inf - inf = NaN while the other gives -+inf -- a box that straddles the origin's coordinate would be
 * rejected. So |1/d| is capped at 2^100: the slab's interval becomes [(lo - o) 2^100, (hi - o) 2^100] instead of the true,
 * still longer one. A hit inside the box lies at a distance t <= 4 E with o within the unpadded box on that axis, i.e.
 * hi - o and o - lo >= 2^-19 E less the rounding above: the capped interval still contains t, so the box is still accepted,
 * and an origin outside the slab still rejects it (both ends on one side of 0). Coordinates stay below 2^27 (E x 2^100 must
 * not overflow; the builder refuses larger scenes). A NaN direction (total internal reflection, src/geometry.c:92-106) stays
 * NaN and rejects every box, as it misses every surface in the reference. */
#define BVH_INV_CAP 1.2676506e30f /* 2^100 */
struct Ray32
{
    float ix, iy, iz, nx, ny, nz; /* 1/d and -o/d */
    float ox, oy, oz, dx, dy, dz; /* o and d */
};
__device__ __forceinline__ float bvh_inv32(double d)
{
    const float i = __builtin_amdgcn_rcpf((float)d); /* v_rcp_f32: 1 ulp, inside the budget above */
    return __builtin_fabsf(i) > BVH_INV_CAP ? __builtin_copysignf(BVH_INV_CAP, i) : i;
}
__device__ __forceinline__ Ray32 bvh_ray32(V3 o, V3 d)
{
    Ray32 r;
    r.ix = bvh_inv32(d.x);
    r.iy = bvh_inv32(d.y);
    r.iz = bvh_inv32(d.z);
    r.nx = -((float)o.x * r.ix);
    r.ny = -((float)o.y * r.iy);
    r.nz = -((float)o.z * r.iz);
    r.ox = (float)o.x; r.oy = (float)o.y; r.oz = (float)o.z;
    r.dx = (float)d.x; r.dy = (float)d.y; r.dz = (float)d.z;
    return r;
}

/* A sphere the ray certainly does not hit within `lim`: its centre is farther from the ray's line than radius + bound, or it
 * lies wholly behind the origin, or wholly beyond the limit. All in f32; with u = 2^-24 and E the largest coordinate, the
 * rounding of centre, origin, direction and the three fused dot products moves the foot point by < 40 u E and the distance
 * along the ray by < 17 u E; reach32 = radius + 64 u E (rounded up) pays for both, so whenever this says "missed" the f64
 * intersector (line_sphere) would have returned no hit, or one beyond `lim`, and skipping it changes no result. Anything
 * not finite compares false: not skipped. */
__device__ __forceinline__ bool sphere_certainly_missed(const BvhLeafPrim &lp, const Ray32 &r, float lim)
{
    const float cx = lp.c32[0] - r.ox, cy = lp.c32[1] - r.oy, cz = lp.c32[2] - r.oz;
    const float t = __builtin_fmaf(cz, r.dz, __builtin_fmaf(cy, r.dy, cx * r.dx));
    const float px = __builtin_fmaf(-t, r.dx, cx), py = __builtin_fmaf(-t, r.dy, cy), pz = __builtin_fmaf(-t, r.dz, cz);
    const float p2 = __builtin_fmaf(pz, pz, __builtin_fmaf(py, py, px * px));
    const float R = lp.reach32;
    return p2 > R * R || t + R < 0.0f || t - R > lim;
}
/* a distance limit for f32 comparisons, rounded UP (inf stays inf, 0 stays 0) */
__device__ __forceinline__ float bvh_limit32(double limit) { return (float)limit * 1.00000024f; }

/* ray vs padded box: a LOWER bound of the entry distance (clamped at 0), or a negative number when the box is certainly
 * missed or behind the ray */
__device__ __forceinline__ float bvh_box_entry(const BvhNode &n, int c, const Ray32 &r)
{
    float t0x = __builtin_fmaf(n.lo[c][0], r.ix, r.nx), t1x = __builtin_fmaf(n.hi[c][0], r.ix, r.nx);
    float t0y = __builtin_fmaf(n.lo[c][1], r.iy, r.ny), t1y = __builtin_fmaf(n.hi[c][1], r.iy, r.ny);
    float t0z = __builtin_fmaf(n.lo[c][2], r.iz, r.nz), t1z = __builtin_fmaf(n.hi[c][2], r.iz, r.nz);
    float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)), __builtin_fminf(t0z, t1z));
    float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)), __builtin_fmaxf(t0z, t1z));
    if (!(tmax >= 0.0f) || !(tmin <= tmax)) return -1.0f;
    return tmin > 0.0f ? tmin : 0.0f;
}

/* Traversal state: a reference is an inner node index (>= 0), a leaf (< BVH_DONE: first slot and count packed), or
 * BVH_DONE. Both loops are "while-while" (Aila & Laine): every lane first walks inner nodes until it holds a leaf,
 * then the wave tests leaf surfaces together -- box tests and surface tests are not interleaved lane by lane, which is
 * what kept 5 lanes in 6 idle in the one-loop form (rocprofv3: SQ_THREAD_CYCLES_VALU / 64 SQ_ACTIVE_INST_VALU = 17 %). */
#define BVH_DONE (-1)
__device__ __forceinline__ int bvh_leaf_ref(int first, int count) { return -2 - (first * 8 + (count - 1)); } /* count 1..8 */
__device__ __forceinline__ bool bvh_is_leaf(int ref) { return ref < BVH_DONE; }

/* the two children of inner node `node` against the ray: which may be entered within `limit` (an f32 upper bound of the
 * distance limit), and a lower bound of where */
__device__ __forceinline__ void bvh_children(const BvhNode &n, const Ray32 &r, float limit, int ref[2], float t[2], bool hit[2])
{
#pragma unroll
    for (int c = 0; c < 2; c += 1)
    {
        ref[c] = n.child[c];
        t[c] = bvh_box_entry(n, c, r);
        hit[c] = ref[c] != BVH_DONE && t[c] >= 0.0f && t[c] <= limit;
    }
}

/* points_mutually_visible, src/daily_ray_trace.c:238-270 */
__device__ __forceinline__ bool points_mutually_visible(const SceneView &sv, V3 p0, V3 p1)
{
    V3 dir = v_normalise(v_sub(p1, p0));
    V3 o = v_sum(p0, v_mul(dir, DRT_VIS_FUDGE));
    double vis_dist = v_length(v_sub(p1, o)) - DRT_VIS_FUDGE;
    bool visible = true;
    for (uint32_t i = 0; i < sv.n_surf; i += 1)
    {
        uint32_t type = sv.surf_type[i];
        if (type != DRT_GEO_SPHERE && type != DRT_GEO_PLANE) continue;
        double dist = surface_distance(sv, i, type, o, dir);
        if (visible && dist < vis_dist) visible = false; /* the reference breaks here; later surfaces cannot undo it */
        if (!__any(visible)) break;
    }
    return visible;
}

struct HitPoint /* scene_point, src/daily_ray_trace.h:113-125 */
{
    V3       position, normal, out;
    double   on_dot;
    uint32_t surface_mat, incident_mat, transmit_mat;
    int      index;
};

/* find_ray_intersection, src/daily_ray_trace.c:334-403 */
__device__ __forceinline__ void find_ray_intersection(const SceneView &sv, const DevScene &sc, HitPoint &ip, V3 ro, V3 rd)
{
    double min_dist = DRT_INF;
    int index = -1;
    ro = v_sum(ro, v_mul(rd, DRT_VIS_FUDGE));
    {
        for (uint32_t i = 0; i < sv.n_surf; i += 1)
        {
            uint32_t type = sv.surf_type[i];
            if (type != DRT_GEO_SPHERE && type != DRT_GEO_PLANE) continue;
            double dist = surface_distance(sv, i, type, ro, rd);
            if (dist < min_dist)
            {
                min_dist = dist;
                index = (int)i;
            }
        }
    }
    ip.index = index;
    if (index >= 0)
    {
        uint32_t type = sv.surf_type[index];
        uint32_t smat = sv.surf_mat[index];
        ip.position = v_sum(ro, v_mul(rd, min_dist));
        if (type == DRT_GEO_SPHERE) ip.normal = v_normalise(v_sub(ip.position, sf3(sv, SF_PX, index)));
        else ip.normal = sf3(sv, SF_NX, index);
        ip.out = v_reverse(rd);
        ip.on_dot = v_dot(ip.normal, ip.out);
        ip.transmit_mat = smat;
        ip.incident_mat = sc.base_mat;
        if (ip.on_dot < 0.0)
        {
            if (type != DRT_GEO_PLANE)
            {
                ip.transmit_mat = sc.base_mat;
                ip.incident_mat = smat;
            }
            ip.normal = v_reverse(ip.normal);
            ip.on_dot = v_dot(ip.normal, ip.out);
        }
        ip.surface_mat = smat;
    }
    else ip.surface_mat = sc.escape_mat;
}

/* refractive indices at trans_wl: value_at_wl(refract_spd, 630), src/spectrum.c:150-162 */
__device__ __forceinline__ double refract_at_trans_wl(const DevScene &sc, const DevMaterial &m)
{
    return drt_lerp(sc.trans_wl, sc.trans_w0, sc.trans_w1, m.refract_i0, m.refract_i1);
}

/* vertex record word 2: the SPD rows of the media either side of the hit, and the rows tabulated for that pair (if any) */
__device__ __forceinline__ uint64_t record_media_word(const DevScene &sc, const SceneView &sv, const HitPoint &ip)
{
    const DevMaterial &sm = sv.mats[ip.surface_mat];
    uint32_t pair = PAIR_NONE;
    if (ip.incident_mat == sc.base_mat && ip.transmit_mat == ip.surface_mat) pair = sm.pair_out;
    else if (ip.incident_mat == ip.surface_mat && ip.transmit_mat == sc.base_mat) pair = sm.pair_in;
    return (uint64_t)((uint32_t)sv.mats[ip.incident_mat].refract_spd & 0xFFFFu) |
           ((uint64_t)((uint32_t)sv.mats[ip.transmit_mat].refract_spd & 0xFFFFu) << 16) |
           ((uint64_t)((uint32_t)sv.mats[ip.transmit_mat].extinct_spd & 0xFFFFu) << 32) | ((uint64_t)pair << 48);
}
/* which three table rows a vertex's per-wavelength Fresnel inputs come from: (ir, tr, te) as the reference has them, or --
 * pair rows tabulated -- (ir, tr, rel_sq) for a dielectric and (cA, cB, -) for a conductor: same registers, same three loads */
__device__ __forceinline__ void fresnel_rows(uint64_t w2, uint32_t &i0, uint32_t &i1, uint32_t &i2, bool &paired)
{
    const uint32_t i_ir = (uint32_t)(w2)&0xFFFFu, i_tr = (uint32_t)(w2 >> 16) & 0xFFFFu, i_te = (uint32_t)(w2 >> 32) & 0xFFFFu;
    const uint32_t pair = (uint32_t)(w2 >> 48) & 0xFFFFu;
    paired = pair != PAIR_NONE;
    const bool conductor = paired && (pair & PAIR_CONDUCTOR) != 0u;
    const uint32_t row = pair & (PAIR_CONDUCTOR - 1u);
    i0 = conductor ? row : i_ir;
    i1 = conductor ? row + 1u : i_tr;
    i2 = (paired && !conductor) ? row : i_te;
}

/* The per-direction scalars of every BDSF in the material's list (src/bdsf.c:105-186):
 * what remains of bdsf(p, in) once the per-wavelength work is taken out. */
__device__ __forceinline__ EvalCoef eval_coefficients(const DevScene &sc, const SceneView &sv, const HitPoint &ip, V3 in)
{
    const DevMaterial &mat = sv.mats[ip.surface_mat];
    EvalCoef e;
    e.a_in = __builtin_fabs(v_dot(ip.normal, in)); /* bp_diffuse :108, bp_glossy :118 */
    e.spec = 0.0;
    e.mn_dot = 0.0;
    e.ct_coef = 0.0;
    e.flags = 0;
    uint32_t needs = mat.needs;
    if (needs & NEED_GLOSSY) /* :113-115 */
    {
        V3 bisector = v_normalise(v_sum(ip.out, in));
        double nb = v_dot(ip.normal, bisector);
        e.spec = drt_pow_shininess((0.0 > nb) ? 0.0 : nb, mat.shininess);
    }
    if (needs & NEED_EQR) /* :123-124, :136-137, :150-151 */
    {
        if (v_equal(in, v_reflect(v_reverse(ip.out), ip.normal))) e.flags |= FLAG_EQR;
    }
    if (needs & NEED_EQT) /* :163-168 */
    {
        double ir = refract_at_trans_wl(sc, sv.mats[ip.incident_mat]);
        double tr = refract_at_trans_wl(sc, sv.mats[ip.transmit_mat]);
        if (v_equal(in, v_transmit(v_reverse(ip.out), ip.normal, ir, tr))) e.flags |= FLAG_EQT;
    }
    if (needs & NEED_CT) /* :176-184 */
    {
        V3 micro_normal = v_normalise(v_sum(ip.out, in));
        e.mn_dot = __builtin_fabs(v_dot(ip.normal, micro_normal));
        e.ct_coef = ggx_att(ip.out, ip.normal, micro_normal, mat.roughness) * (1.0 / (4.0 * ip.on_dot));
    }
    return e;
}

/* Direction samplers, src/bdsf.c:188-292; they return the reciprocal pdf */
__device__ __forceinline__ void sample_direction(const DevScene &sc, const SceneView &sv, const HitPoint &ip, uint64_t &rs,
                                                 uint32_t &draws, V3 &dir, double &recip_pdf)
{
    const DevMaterial &mat = sv.mats[ip.surface_mat];
    const V3 zaxis = v3(0.0, 0.0, 1.0);
    switch (mat.dir_func)
    {
        case DRT_DIRF_cos_weighted_sample_hemisphere: /* :200-213 */
        {
            V3 q;
            for (;;)
            {
                q = uniform_sample_disc(rs, draws);
                if (v_dot(q, q) < 1.0) break;
            }
            q.z = __builtin_sqrt(1.0 - v_dot(q, q));
            M33 r = rotation_between(zaxis, ip.normal);
            dir = m_vmul(r, q);
            recip_pdf = DRT_PI / v_dot(ip.normal, dir);
            break;
        }
        case DRT_DIRF_uniform_sample_hemisphere: /* :191-198 */
        {
            V3 s = uniform_sample_sphere(rs, draws);
            M33 r = rotation_between(zaxis, ip.normal);
            dir = m_vmul(r, s);
            recip_pdf = 2.0 * DRT_PI;
            break;
        }
        case DRT_DIRF_sample_specular_direction: /* :215-220 */
        {
            dir = v_reflect(v_reverse(ip.out), ip.normal);
            recip_pdf = 1.0;
            break;
        }
        case DRT_DIRF_sample_transmit_direction: /* :222-234 */
        {
            double ir = refract_at_trans_wl(sc, sv.mats[ip.incident_mat]);
            double tr = refract_at_trans_wl(sc, sv.mats[ip.transmit_mat]);
            dir = v_transmit(v_reverse(ip.out), ip.normal, ir, tr);
            recip_pdf = 1.0;
            break;
        }
        case DRT_DIRF_sample_reflect_or_transmit_direction: /* :236-259 */
        {
            const DevMaterial &im = sv.mats[ip.incident_mat];
            const DevMaterial &tm = sv.mats[ip.transmit_mat];
            /* value_at_wl(reflectance spectrum, 630) needs the Fresnel term at the two bracketing samples */
            double inc_sin_sq = 1.0 - ip.on_dot * ip.on_dot;
            double r0 = dielectric_reflectance(im.refract_i0, tm.refract_i0, ip.on_dot, inc_sin_sq);
            double r1 = dielectric_reflectance(im.refract_i1, tm.refract_i1, ip.on_dot, inc_sin_sq);
            double rd = drt_lerp(sc.trans_wl, sc.trans_w0, sc.trans_w1, r0, r1);
            double ir = refract_at_trans_wl(sc, im);
            double tr = refract_at_trans_wl(sc, tm);
            double f = drt_rng(rs, draws);
            V3 w = v_reverse(ip.out);
            if (f < rd)
            {
                dir = v_reflect(w, ip.normal);
                recip_pdf = 1.0 / rd;
            }
            else
            {
                dir = v_transmit(w, ip.normal, ir, tr);
                recip_pdf = 1.0 / (1.0 - rd);
            }
            break;
        }
        case DRT_DIRF_sample_ct_direction: /* :261-292 */
        {
            do
            {
                double f = drt_rng(rs, draws);
                double g = drt_rng(rs, draws);
                double phi_mn = (2.0 * DRT_PI) * g;
                double tan_mn = (mat.roughness * __builtin_sqrt(f)) / __builtin_sqrt(1.0 - f);
                double cos_mn = 1.0 / __builtin_sqrt(1.0 + tan_mn * tan_mn);
                double sin_mn = __builtin_sqrt(1.0 - cos_mn * cos_mn);
                double sp, cp;
                drt_sincos(phi_mn, sp, cp);
                V3 micro_normal = v3(sin_mn * cp, sin_mn * sp, cos_mn);
                M33 r = rotation_between(zaxis, ip.normal);
                micro_normal = m_vmul(r, micro_normal);
                double sn_mn_dot = v_dot(ip.normal, micro_normal);
                if (sn_mn_dot < 0.0)
                {
                    micro_normal = v_reverse(micro_normal);
                    sn_mn_dot = -sn_mn_dot;
                }
                double o_mn_dot = v_dot(ip.out, micro_normal);
                dir = v_reflect(v_reverse(ip.out), micro_normal);
                double d = ggx(ip.normal, micro_normal, mat.roughness) * sn_mn_dot;
                recip_pdf = ((4.0 * o_mn_dot) / d);
            } while (v_dot(dir, ip.normal) < 0.0);
            break;
        }
        default:
            dir = v3(0.0, 0.0, 0.0);
            recip_pdf = 0.0;
            break;
    }
}

__device__ __forceinline__ void store_coef(uint64_t *w, const EvalCoef &e)
{
    w[0] = (uint64_t)__double_as_longlong(e.a_in);
    w[1] = (uint64_t)__double_as_longlong(e.spec);
    w[2] = (uint64_t)__double_as_longlong(e.mn_dot);
    w[3] = (uint64_t)__double_as_longlong(e.ct_coef);
}

/* Camera ray: sample_pixel_point + sample_scene's ray set-up, src/daily_ray_trace.c:550-607 */
__device__ __forceinline__ void camera_ray(const DevCamera &cam, uint32_t scheme, uint32_t x, uint32_t y, uint64_t &rs,
                                           uint32_t &draws, V3 &ro, V3 &rd)
{
    double px = 0.0, py = 0.0;
    if (scheme == DRT_FILM_SAMPLE_CENTER) { px = 0.5; py = 0.5; }
    else if (scheme == DRT_FILM_SAMPLE_RANDOM) { px = drt_rng(rs, draws); py = drt_rng(rs, draws); }
    double film_x = ((double)x + px) * cam.pixel_width;
    double film_y = ((double)y + py) * cam.pixel_height;
    V3 bottom = v_mul(cam.up, film_y);
    V3 left = v_mul(cam.right, film_x);
    V3 pixel_point = v_sum(v_sum(left, bottom), cam.film_bottom_left);
    if (cam.aperture_radius > 0.0)
    {
        V3 focus_dir = v_normalise(v_sub(cam.aperture_position, pixel_point));
        focus_dir = v_mul(focus_dir, cam.focal_depth / v_dot(focus_dir, cam.forward));
        V3 focus_point = v_sum(pixel_point, focus_dir);
        M33 r = rotation_between(v3(0.0, 0.0, 1.0), cam.forward);
        V3 disc_point = v_mul(uniform_sample_disc(rs, draws), cam.aperture_radius);
        V3 lens_point = m_vmul(r, disc_point);
        ro = v_sum(cam.aperture_position, lens_point);
        rd = v_normalise(v_sub(focus_point, ro));
    }
    else
    {
        ro = pixel_point;
        rd = v_normalise(v_sub(cam.aperture_position, ro));
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* The trace kernel                                                                                */

#define TRACE_BLOCK 256
#define DRT_TRACE_TAIL_MAX 8u /* tail wavelengths the trace kernel carries at most (the launcher's rule: S mod 64 <= 8) */
#ifndef DRT_TRACE_WAVES_PER_SIMD
#define DRT_TRACE_WAVES_PER_SIMD 3 /* register budget: launch_bounds' 2nd argument is waves per SIMD */
#endif

/* LDS carve-up (8-byte aligned): surfaces, lights, then u32 tables, then materials (see trace_lds_bytes in the launcher) */

template <bool SCENE_IN_LDS, bool TAIL = false>
__global__ __launch_bounds__(TRACE_BLOCK, DRT_TRACE_WAVES_PER_SIMD) void drt_trace_kernel(DevScene sc, DevCamera cam, TraceParams tp,
                                                                 uint64_t *__restrict__ records, uint64_t *__restrict__ headers,
                                                                 int32_t *__restrict__ hits, unsigned long long *__restrict__ counters,
                                                                 unsigned long long *__restrict__ work_counter)
{
    extern __shared__ double lds_raw[];
    SceneView sv;
    sv.n_surf = sc.n_surf;
    sv.n_lights = sc.n_lights;
    if (SCENE_IN_LDS)
    {
        /* stage the SoA scene: coalesced HBM reads, one pass per table */
        double *l_surf = lds_raw;
        double *l_lights = l_surf + (size_t)SF_COUNT * sc.n_surf;
        uint32_t *l_u32 = (uint32_t *)(l_lights + (size_t)LF_COUNT * sc.n_lights);
        uint32_t n_u32 = 2 * sc.n_surf + 2 * sc.n_lights;
        DevMaterial *l_mats = (DevMaterial *)((char *)l_u32 + (((size_t)n_u32 * 4 + 7) & ~(size_t)7));
        for (uint32_t k = threadIdx.x; k < SF_COUNT * sc.n_surf; k += TRACE_BLOCK) l_surf[k] = sc.surf[k];
        for (uint32_t k = threadIdx.x; k < LF_COUNT * sc.n_lights; k += TRACE_BLOCK) l_lights[k] = sc.lights[k];
        for (uint32_t k = threadIdx.x; k < sc.n_surf; k += TRACE_BLOCK)
        {
            l_u32[k] = sc.surf_type[k];
            l_u32[sc.n_surf + k] = sc.surf_mat[k];
        }
        for (uint32_t k = threadIdx.x; k < sc.n_lights; k += TRACE_BLOCK)
        {
            l_u32[2 * sc.n_surf + k] = sc.light_type[k];
            l_u32[2 * sc.n_surf + sc.n_lights + k] = sc.light_mat[k];
        }
        const uint64_t *src = (const uint64_t *)sc.mats;
        uint64_t *dst = (uint64_t *)l_mats;
        for (uint32_t k = threadIdx.x; k < sc.n_mat * (uint32_t)(sizeof(DevMaterial) / 8); k += TRACE_BLOCK) dst[k] = src[k];
        __syncthreads();
        sv.surf = l_surf;
        sv.lights = l_lights;
        sv.surf_type = l_u32;
        sv.surf_mat = l_u32 + sc.n_surf;
        sv.light_type = l_u32 + 2 * sc.n_surf;
        sv.light_mat = l_u32 + 2 * sc.n_surf + sc.n_lights;
        sv.mats = l_mats;
        sv.bvh_nodes = nullptr; /* a scene that fits LDS is scanned whole */
        sv.bvh_leaf = nullptr;
    }
    else
    {
        sv.surf = sc.surf;
        sv.lights = sc.lights;
        sv.surf_type = sc.surf_type;
        sv.surf_mat = sc.surf_mat;
        sv.light_type = sc.light_type;
        sv.light_mat = sc.light_mat;
        sv.mats = sc.mats;
        sv.bvh_nodes = sc.bvh_nodes;
        sv.bvh_leaf = sc.bvh_leaf;
    }

    /*
     * Tail wavelengths in the trace kernel. The shade kernel's lanes are wavelengths, and the S mod 64 that do not fill a wave (5 on
     * the reference grid) cost it a pass of their own in which a lane must decode records by itself. HERE a lane is a path and
     * already holds every number a vertex contributes, so the same arithmetic for those few wavelengths is a handful of f64
     * operations per vertex at full lane use. It is done for paths that consist of two-lobe plastic and mirror vertices only (and
     * for paths without a vertex): running throughput and radiance per tail wavelength live in LDS (a wave's block: [2 R][64 lanes]),
     * the table's tail columns too, and when the path ends its values go to tail_stage, flagged in the header
     * (HDR_TERM_TAIL_STAGED). A path that meets any other material stays unflagged and is replayed by the shade kernel's tail pass,
     * which takes such paths as tasks and skips the flagged ones; in a scene where EVERY path is carried here the launcher says so
     * (ShadeParams::tail_staged) and that pass only updates the film. The launcher takes this instantiation for every scene scanned
     * out of LDS with one light and a tail of at most 8 wavelengths (DRT_TRACE_TAIL=0 turns it off, the parity tests' A/B).
     * Same operations in the same order as drt_shade_kernel's (src/daily_ray_trace.c:440-472, :615).
     */
    const uint32_t TR = tp.tail_count;
    const bool tail_on = TAIL && SCENE_IN_LDS && tp.tail_stage != nullptr; /* TAIL: an instantiation of its own, so that scenes that cannot use it run the kernel without any of this */
    double *l_spd_tail = nullptr, *tail_state = nullptr;
    if (tail_on)
    {
        l_spd_tail = (double *)(sv.mats + sc.n_mat); /* behind the LDS copy of the materials, the last of the scene's tables */
        for (uint32_t k = threadIdx.x; k < tp.n_spd * TR; k += TRACE_BLOCK) l_spd_tail[k] = tp.spd_tail[k];
        tail_state = l_spd_tail + (size_t)tp.n_spd * TR + (size_t)(threadIdx.x >> 6) * (2u * TR * 64u) + (threadIdx.x & 63u);
        __syncthreads();
    }
    bool tail_ok = false; /* per lane: the path's tail wavelengths are being carried here */
    const uint32_t lane = threadIdx.x & 63u;
    /* per-wave work queue: a wave draws chunks of consecutive path ids from the global counter and
     * deals them to its idle lanes by ballot + prefix count */
    const uint64_t CHUNK = tp.chunk;
    uint64_t chunk_next = 0, chunk_end = 0; /* wave-uniform */

    uint32_t n_scans = 0, n_shaded = 0, n_shadow = 0, n_draws = 0, n_paths = 0;

    /* per-lane path state */
    bool alive = false;
    bool exhausted = false; /* wave-uniform: no more work to draw */
    uint64_t pid = 0, rs = 1, hit_row = 0; /* hit_row: the path's row in the hit log, ordered (sample, pixel) */
    uint32_t depth = 0, shaded = 0;
    uint32_t vis0_mask = 0;    /* bit v: light 0 is visible from shaded vertex v (< 8); header bits 24-31 */
    uint32_t plastic_mask = 0; /* bit v: shaded vertex v (< 16) has the two-lobe plastic list; header bits 48-63, read by the shade kernel's tail pass */
    V3 ro = v3(0, 0, 0), rd = v3(0, 0, 0);
    uint64_t *hdr = nullptr;
    uint32_t blk = 0, tbl = 0;              /* the pool block of the current four vertices; the path's table block (deep paths) */
    uint32_t spare = ~0u, spare_tbl = ~0u; /* a block (and, for deep paths, a table block) held ready: see path_spare_block */
    WavePool wp = {0u, 0u, 0u};
    if (*tp.overflow) return; /* an earlier launch ran out of record blocks: the host renders from there again */

    for (;;)
    {
        /* ---- refill idle lanes ---- */
        unsigned long long idle_mask = __ballot(!alive);
        if (idle_mask != 0ull && !exhausted)
        {
            uint32_t want = (uint32_t)__popcll(idle_mask);
            uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull)); /* prefix count among idle lanes */
            uint64_t avail = chunk_end - chunk_next;
            if (avail < want)
            {
                /* hand out what is left of the chunk first, then draw a new chunk */
                if (!alive && rank < avail)
                {
                    pid = chunk_next + rank;
                    alive = true;
                }
                uint32_t taken = (uint32_t)avail;
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)CHUNK);
                base = __shfl(base, 0);
                if (base >= tp.n_paths)
                {
                    exhausted = true;
                    chunk_next = chunk_end = 0;
                }
                else
                {
                    chunk_next = base;
                    chunk_end = (base + CHUNK < tp.n_paths) ? base + CHUNK : tp.n_paths;
                    uint64_t avail2 = chunk_end - chunk_next;
                    bool fresh = false;
                    if (!alive && rank >= taken && (uint64_t)(rank - taken) < avail2)
                    {
                        pid = chunk_next + (rank - taken);
                        alive = true;
                        fresh = true;
                    }
                    uint32_t used = (want - taken < avail2) ? (want - taken) : (uint32_t)avail2;
                    chunk_next += used;
                    (void)fresh;
                }
                /* lanes that got a path in either step start it below */
            }
            else
            {
                if (!alive)
                {
                    pid = chunk_next + rank;
                    alive = true;
                }
                chunk_next += want;
            }
            /* start the newly assigned paths: lanes that were idle in idle_mask and are alive now */
            bool started = alive && ((idle_mask >> lane) & 1ull);
            if (started)
            {
                uint64_t q = pid / tp.n_samples; /* consecutive ids: the samples of one pixel */
                uint64_t s_local = pid - q * tp.n_samples;
                hit_row = s_local * tp.n_pix + q;
                uint32_t j = (uint32_t)(q / tp.tile_w);
                uint32_t i = (uint32_t)(q - (uint64_t)j * tp.tile_w);
                uint32_t x = tp.x0 + i;
                uint32_t y = tp.y0 + j * tp.row_stride;
                uint32_t sample = tp.first_sample + (uint32_t)s_local;
                uint64_t key = tp.seed + (((uint64_t)sample * (uint64_t)tp.height + (uint64_t)y) * (uint64_t)tp.width + (uint64_t)x);
                rs = drt_splitmix64(key);
                camera_ray(cam, tp.pixel_scheme, x, y, rs, n_draws, ro, rd);
                depth = 0;
                shaded = 0;
                plastic_mask = 0;
                vis0_mask = 0;
                tail_ok = tail_on;
                if (tail_on)
                {
#pragma unroll
                    for (uint32_t j = 0; j < DRT_TRACE_TAIL_MAX; j += 1)
                    {
                        if (j >= TR) break;
                        tail_state[(2u * j) * 64u] = 1.0;      /* throughput, const_spectrum(throughput, 1.0), :440 */
                        tail_state[(2u * j + 1u) * 64u] = 0.0; /* dst */
                    }
                }
                uint64_t slot = q * (uint64_t)tp.batch + s_local;
                hdr = headers + slot * REC_HEADER_WORDS;
                /* vignette: dot(ray_direction, forward) of the PRIMARY ray, src/daily_ray_trace.c:614 */
                hdr[1] = (uint64_t)__double_as_longlong(v_dot(rd, cam.forward) * 1.0);
                n_paths += 1;
                if (tp.record_hits)
                {
                    int32_t *h = hits + ((uint64_t)tp.hits_sample_offset * tp.n_pix + hit_row) * tp.max_depth;
                    for (uint32_t d = 0; d < tp.max_depth; d += 1) h[d] = -2;
                }
            }
        }
        if (!__any(alive)) break;

        /* a spare record block for the lanes whose path may open one at this iteration's vertex (the whole wave takes part) */
        if (!path_spare_block(tp, wp, alive, shaded, spare, spare_tbl, lane))
        {
            hdr[0] = (uint64_t)HDR_TERM_NOT_DONE << 16;
            alive = false;
        }
        if (alive)
        {
            /* ---- one iteration of cast_ray's loop, src/daily_ray_trace.c:446-474 ---- */
            HitPoint ip;
            find_ray_intersection(sv, sc, ip, ro, rd);
            n_scans += 1;
            if (tp.record_hits) hits[((uint64_t)tp.hits_sample_offset * tp.n_pix + hit_row) * tp.max_depth + depth] = ip.index;
            const DevMaterial &mat = sv.mats[ip.surface_mat];
            bool terminal = false;
            uint32_t term = 0, term_spd = 0;
            if (mat.is_black_body && !mat.is_emissive) terminal = true;
            else if (mat.is_black_body && mat.is_emissive)
            {
                terminal = true;
                term = 1;
                term_spd = (uint32_t)mat.emission_spd;
            }
            else
            {
                path_open_vertex(tp, wp, records, shaded, hdr, blk, tbl, spare, spare_tbl);
                uint64_t *vrec = records + (uint64_t)blk * tp.block_words + (uint64_t)(shaded & (REC_BLOCK_VERTICES - 1u)) * tp.vertex_words;
                /* direct_light_contribution, :272-332 -- light samples are drawn before the shadow test */
                n_shaded += 1;
                bool t_vis = false; /* light 0 as the tail arithmetic below wants it (the kernel carries tails only in one-light scenes) */
                double t_c = 0.0, t_a_in = 0.0, t_spec = 0.0;
                uint32_t t_em = 0, t_flags = 0;
                for (uint32_t l = 0; l < sv.n_lights; l += 1)
                {
                    uint32_t ltype = sv.light_type[l];
                    V3 lpos = v3(sv.lights[LF_PX * sv.n_lights + l], sv.lights[LF_PY * sv.n_lights + l], sv.lights[LF_PZ * sv.n_lights + l]);
                    double light_pdf = sv.lights[LF_PDF * sv.n_lights + l];
                    double attenuation = 1.0;
                    V3 light_position = lpos;
                    if (ltype == DRT_GEO_POINT)
                    {
                        double dist = v_length(v_sub(light_position, ip.position));
                        attenuation = ((4.0 * DRT_PI) * dist) * dist;
                    }
                    else if (ltype == DRT_GEO_SPHERE)
                    {
                        double u = drt_rng(rs, n_draws);
                        double v = drt_rng(rs, n_draws);
                        double r = __builtin_sqrt(1.0 - u * u);
                        double t = (2.0 * DRT_PI) * v;
                        double st, ct;
                        drt_sincos(t, st, ct);
                        V3 sp = v3(r * ct, r * st, u);
                        light_position = v_sum(lpos, v_mul(sp, sv.lights[LF_RADIUS * sv.n_lights + l]));
                    }
                    else if (ltype == DRT_GEO_PLANE)
                    {
                        double u = drt_rng(rs, n_draws);
                        double v = drt_rng(rs, n_draws);
                        V3 lu = v3(sv.lights[LF_UX * sv.n_lights + l], sv.lights[LF_UY * sv.n_lights + l], sv.lights[LF_UZ * sv.n_lights + l]);
                        V3 lv = v3(sv.lights[LF_VX * sv.n_lights + l], sv.lights[LF_VY * sv.n_lights + l], sv.lights[LF_VZ * sv.n_lights + l]);
                        light_position = v_sum(v_sum(lpos, v_mul(lu, u)), v_mul(lv, v));
                    }
                    n_shadow += 1;
                    bool visible = points_mutually_visible(sv, ip.position, light_position);
                    uint64_t *lrec = vrec + REC_VERTEX_WORDS + (uint64_t)l * REC_LIGHT_WORDS;
                    uint32_t lflags = 0;
                    if (visible)
                    {
                        V3 incoming = v_normalise(v_sub(light_position, ip.position));
                        EvalCoef e = eval_coefficients(sc, sv, ip, incoming);
                        lflags = e.flags | FLAG_VISIBLE;
                        if (l == 0 && shaded < 8u) vis0_mask |= 1u << shaded;
                        double c = attenuation * (light_pdf);
                        lrec[1] = (uint64_t)__double_as_longlong(c);
                        store_coef(lrec + 2, e);
                        if (l == 0) { t_vis = true; t_c = c; t_a_in = e.a_in; t_spec = e.spec; t_flags = e.flags; }
                    }
                    uint32_t em_spd = (uint32_t)sv.mats[sv.light_mat[l]].emission_spd & 0xFFFFu;
                    lrec[0] = (uint64_t)em_spd | ((uint64_t)lflags << 16);
                    if (l == 0) t_em = em_spd;
                }
                /* sampled continuation, :464-472 */
                V3 in;
                double dir_pdf;
                sample_direction(sc, sv, ip, rs, n_draws, in, dir_pdf);
                EvalCoef e = eval_coefficients(sc, sv, ip, in);
                if (tail_ok)
                {
                    if (mat.vertex_flags & FLAG_PLASTIC)
                    {
                        /* bdsf() over {bp_diffuse_bdsf, bp_glossy_bdsf} at the tail wavelengths, as drt_shade_kernel's plastic_vertex */
                        const double *row_d = l_spd_tail + ((uint32_t)mat.diffuse_spd & 0xFFFFu) * TR;
                        const double *row_g = l_spd_tail + ((uint32_t)mat.glossy_spd & 0xFFFFu) * TR;
                        const double *row_e = l_spd_tail + t_em * TR;
                        /* (a constant trip count, so that the compiler unrolls: the wavelengths' chains of dependent f64 operations then
                         *  run side by side instead of one after the other) */
#pragma unroll
                        for (uint32_t j = 0; j < DRT_TRACE_TAIL_MAX; j += 1)
                        {
                            if (j >= TR) break;
                            const double diffuse_pi = row_d[j], glossy = row_g[j];
                            double throughput = tail_state[(2u * j) * 64u], dst = tail_state[(2u * j + 1u) * 64u];
                            double contribution = 0.0;
                            if (t_vis)
                            {
                                double reflectance = diffuse_pi * t_a_in + 0.0;
                                reflectance = (glossy * t_spec) * t_a_in + reflectance;
                                contribution = contribution + reflectance; /* :323 */
                                contribution = contribution * row_e[j];    /* :324 */
                                contribution = contribution * t_c;         /* :326-327 */
                            }
                            dst = dst + throughput * contribution; /* :461-462 */
                            double reflectance = diffuse_pi * e.a_in + 0.0;
                            reflectance = (glossy * e.spec) * e.a_in + reflectance;
                            reflectance = reflectance * dir_pdf; /* :468 */
                            throughput = throughput * reflectance; /* :469 */
                            tail_state[(2u * j) * 64u] = throughput;
                            tail_state[(2u * j + 1u) * 64u] = dst;
                        }
                    }
                    else if (mat.num_bdsfs == 1u && mat.bdsfs[0] == DRT_BDSF_mirror_bdsf)
                    {
                        /* bdsf() over {mirror_bdsf}: the mirror's spectrum where the direction is the mirror direction exactly, else 0
                         * (src/bdsf.c:121-132) -- as bdsf_at_wavelength() in drt_shade_kernel */
                        const double *row_m = l_spd_tail + ((uint32_t)mat.mirror_spd & 0xFFFFu) * TR;
                        const double *row_e = l_spd_tail + t_em * TR;
#pragma unroll
                        for (uint32_t j = 0; j < DRT_TRACE_TAIL_MAX; j += 1)
                        {
                            if (j >= TR) break;
                            const double mirror = row_m[j];
                            double throughput = tail_state[(2u * j) * 64u], dst = tail_state[(2u * j + 1u) * 64u];
                            double contribution = 0.0;
                            if (t_vis)
                            {
                                double bdsf_result = (t_flags & FLAG_EQR) ? mirror : 0.0;
                                double reflectance = bdsf_result + 0.0;
                                contribution = contribution + reflectance;
                                contribution = contribution * row_e[j];
                                contribution = contribution * t_c;
                            }
                            dst = dst + throughput * contribution;
                            double bdsf_result = (e.flags & FLAG_EQR) ? mirror : 0.0;
                            double reflectance = bdsf_result + 0.0;
                            reflectance = reflectance * dir_pdf;
                            throughput = throughput * reflectance;
                            tail_state[(2u * j) * 64u] = throughput;
                            tail_state[(2u * j + 1u) * 64u] = dst;
                        }
                    }
                    else tail_ok = false; /* any other material: the shade kernel's tail pass replays this path */
                }
                vrec[0] = mat.bdsf_packed;
                vrec[1] = (uint64_t)mat.num_bdsfs | ((uint64_t)(e.flags | mat.vertex_flags) << 8) | ((uint64_t)((uint32_t)mat.diffuse_spd & 0xFFFFu) << 16) |
                          ((uint64_t)((uint32_t)mat.glossy_spd & 0xFFFFu) << 32) | ((uint64_t)((uint32_t)mat.mirror_spd & 0xFFFFu) << 48);
                vrec[2] = record_media_word(sc, sv, ip);
                vrec[3] = (uint64_t)__double_as_longlong(ip.on_dot);
                vrec[4] = (uint64_t)__double_as_longlong(dir_pdf);
                store_coef(vrec + 5, e);
                if ((mat.vertex_flags & FLAG_PLASTIC) && shaded < 16u) plastic_mask |= 1u << shaded;
                shaded += 1;
                rd = in;
                ro = ip.position;
            }
            depth += 1;
            if (terminal || depth >= tp.max_depth)
            {
                uint32_t staged = 0;
                if (tail_ok)
                {
                    /* the sample's value at the tail wavelengths: emission of what the path ended on, vignette (:452-457, :615) */
                    const double vignette = __longlong_as_double((long long)hdr[1]);
                    double *st = tp.tail_stage + (uint64_t)(hdr - headers) / REC_HEADER_WORDS * TR;
                    const double *row_t = l_spd_tail + (term_spd & 0xFFFFu) * TR;
#pragma unroll
                    for (uint32_t j = 0; j < DRT_TRACE_TAIL_MAX; j += 1)
                    {
                        if (j >= TR) break;
                        double dst = tail_state[(2u * j + 1u) * 64u];
                        if (term == 1) dst = dst + tail_state[(2u * j) * 64u] * row_t[j];
                        st[j] = dst * vignette;
                    }
                    staged = HDR_TERM_TAIL_STAGED;
                }
                hdr[0] = (uint64_t)shaded | ((uint64_t)(term | staged) << 16) | ((uint64_t)(term_spd & 0xFFFFu) << 32) | ((uint64_t)plastic_mask << 48) | ((uint64_t)vis0_mask << 24);
                alive = false;
            }
        }
    }

    /* statistics: wave reduction, one atomic per wave and counter */
    uint64_t vals[6] = {n_paths, n_scans, n_shaded, n_shadow, n_draws, wp.taken};
    for (int k = 0; k < 6; k += 1)
    {
        uint64_t v = vals[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && v) atomicAdd(&counters[k], (unsigned long long)v);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* The shade + film kernel                                                                         */

#define SHADE_BLOCK 256
#ifndef DRT_SHADE_WAVES_PER_SIMD
#define DRT_SHADE_WAVES_PER_SIMD 4
#endif
#define SHADE_WAVES (SHADE_BLOCK / 64)
/* one wavelength set per lane fits 128 registers (4 waves per SIMD); two or more sets hold two or more of every per-wavelength
 * value and spill at that budget (116-250 bytes of scratch): they get the registers of 3 or 2 waves per SIMD instead */
#ifndef DRT_SHADE_WAVES_MULTISET
#define DRT_SHADE_WAVES_MULTISET(n) ((n) == 2 ? 3 : 2)
#endif
#define SHADE_WAVES_PER_SIMD_FOR(n) ((n) == 1 ? DRT_SHADE_WAVES_PER_SIMD : DRT_SHADE_WAVES_MULTISET(n))
#ifndef DRT_SHADE_WAVES_SIMPLE
#define DRT_SHADE_WAVES_SIMPLE 5 /* the SIMPLE instantiation (no Fresnel code): 5 waves per SIMD without a spill; 6 (80 registers, 40 bytes of scratch)
                                    measured the same (config 3: shade 1772 -> 1692 ms at 6, 1691 at 5) */
#endif
#define SHADE_MAX_SETS 4 /* wavelengths per lane: S <= 64 * SHADE_MAX_SETS */

struct ShadeParams
{
    uint64_t n_pix;
    uint32_t n_samples, first_sample, vertex_words, block_words; /* a vertex record and a pool block (four vertices) in 8-byte words */
    uint32_t n_lights, batch;
    const uint32_t *overflow; /* the trace launch ran out of record blocks: nothing here is touched */
    uint32_t vertex_shift, mode; /* log2(vertex_words); DIAGNOSTIC mode: 1 main pass only, 2 tail pass only (timing probes; the film is then incomplete) */
    uint32_t tail_first, tail_count; /* wavelengths [tail_first, tail_first + tail_count) go through the packed tail pass (0: none) */
    uint32_t chunk, sub_pixels;      /* pixels per group (= tail packing size when there is a tail); pixels per main-pass work item */
    uint32_t light0_em_spd, tail_staged;  /* emission SPD row of light 0 (what every light block of light 0 says); tail_staged: the trace
                                             kernel has put every path's tail wavelengths into tail_stage, the tail pass only updates the film */
    double  *tail_stage;                  /* [n_pix * batch][tail_count]: per-sample results of the tail pass (see the kernel) */
    uint32_t cmf_rw, cmf_x, cmf_y, cmf_z; /* XYZ film mode: SPD rows of the white table and the colour-matching functions */
    uint32_t tail_period_mains, pad1;  /* split queue with a tail: main-pass items between two tail items (<= main items per group) */
    uint32_t items_per_group, n_items; /* work items: per group of `chunk` pixels, ceil(chunk/sub_pixels) main-pass items and, with a tail,
                                          one tail-pass item; items_per_group == 1: one item does the group's main pass and then its tail */
};

__device__ __forceinline__ double word_as_double(uint64_t w) { return __longlong_as_double((long long)w); }

__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l)
{
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

/* where vertex v of a path lives, from the header's block words (see the record layout above) */
__device__ __forceinline__ const uint64_t *path_vertex(const uint64_t *__restrict__ pool, uint32_t block_words, uint32_t vw, uint64_t h2, uint64_t h3, uint32_t v)
{
    const uint32_t b = v >> REC_BLOCK_SHIFT;
    uint32_t id;
    if (b == 0) id = (uint32_t)h2;
    else if (b == 1) id = (uint32_t)(h2 >> 32);
    else if (b == 2) id = (uint32_t)h3;
    else id = ((const uint32_t *)(pool + (uint64_t)(uint32_t)(h3 >> 32) * block_words))[b - REC_HEADER_BLOCKS];
    return pool + (uint64_t)id * block_words + (uint64_t)(v & (REC_BLOCK_VERTICES - 1u)) * vw;
}

/* value of SPD row `idx` at wavelength `lam` */
template <typename SpdPtr>
__device__ __forceinline__ double spd_at(SpdPtr spds, uint32_t S, uint32_t idx, uint32_t lam)
{
    return spds[idx * S + lam]; /* a missing spectrum is the table's all-zero row */
}

/* One BDSF sum for one wavelength: bdsf(), src/daily_ray_trace.c:215-229. `bdsf_result` is zeroed
 * once and carried from function to function; functions whose direction test fails leave it (Q1). */
/* `paired`: the vertex's pair of media has tabulated rows (fresnel_rows): then te is rel_sq for the dielectric functions, and
 * (ir, tr) are (cA, cB) for the conductor functions -- the *_sel forms of drt_device.h, same bits with fewer divisions. */
/* SIMPLE: an instantiation for scenes whose materials list nothing but bp_diffuse_bdsf, bp_glossy_bdsf and mirror_bdsf (the launcher
 * checks): the Fresnel cases, their divisions and square roots and the registers they live in are not compiled in, and the shade kernel
 * that is left fits five waves per SIMD instead of four (config 3, all plastic: shade 1772 -> 1691 ms). Same bits: the cases left
 * are the same code (tests/test_gpu_parity.py flips DRT_NO_SIMPLE_SHADE). */
template <bool SIMPLE = false>
__device__ __forceinline__ double bdsf_at_wavelength(uint64_t list, uint32_t num_bdsfs, double diffuse_pi, double glossy, double mirror,
                                                     double ir, double tr, double te, double on_dot, double a_in, double spec,
                                                     double mn_dot, double ct_coef, uint32_t flags, bool paired)
{
    double bdsf_result = 0.0;
    double reflectance = 0.0;
/* The Fresnel terms are pure f64 arithmetic, so the optimiser would hoist them out of this loop and
 * run them (divisions, square roots) for every vertex of every material. The empty asm ties each one
 * to its `case`: only vertices whose material lists the function pay for it. */
#define DRT_PIN_HERE(x) asm volatile("" : "+v"(x))
    for (uint32_t i = 0; i < num_bdsfs; i += 1)
    {
        switch ((uint32_t)(list >> (4 * i)) & 15u)
        {
            case DRT_BDSF_bp_diffuse_bdsf: /* src/bdsf.c:105-109; diffuse_pi = diffuse_spd * (1/PI), a per-material table */
                bdsf_result = diffuse_pi * a_in;
                break;
            case DRT_BDSF_bp_glossy_bdsf: /* :111-119 */
                bdsf_result = (glossy * spec) * a_in;
                break;
            case DRT_BDSF_mirror_bdsf: /* :121-132 */
                bdsf_result = (flags & FLAG_EQR) ? mirror : 0.0;
                break;
            case DRT_BDSF_fs_conductor_bdsf: /* :134-146 */
                if (SIMPLE) break;
                if (flags & FLAG_EQR)
                {
                    DRT_PIN_HERE(ir);
                    double c2 = on_dot * on_dot;
                    bdsf_result = conductor_reflectance_sel(paired, ir, tr, te, on_dot, c2, 1.0 - c2);
                }
                break;
            case DRT_BDSF_fs_dielectric_reflectance_bdsf: /* :148-159 */
                if (SIMPLE) break;
                if (flags & FLAG_EQR)
                {
                    DRT_PIN_HERE(ir);
                    bdsf_result = dielectric_reflectance_sel(paired, ir, tr, te, on_dot, 1.0 - on_dot * on_dot);
                }
                break;
            case DRT_BDSF_fs_dielectric_transmittance_bdsf: /* :161-172, :69-76 */
                if (SIMPLE) break;
                if (flags & FLAG_EQT)
                {
                    DRT_PIN_HERE(ir);
                    bdsf_result = 1.0 - dielectric_reflectance_sel(paired, ir, tr, te, on_dot, 1.0 - on_dot * on_dot);
                }
                break;
            case DRT_BDSF_ct_conductor_bdsf: /* :174-186 */
            {
                if (SIMPLE) break;
                DRT_PIN_HERE(ir);
                double c2 = mn_dot * mn_dot;
                bdsf_result = conductor_reflectance_sel(paired, ir, tr, te, mn_dot, c2, 1.0 - c2) * ct_coef;
                break;
            }
            default: break;
        }
        reflectance = bdsf_result + reflectance;
    }
#undef DRT_PIN_HERE
    return reflectance;
}

/*
 * One pixel per wave, lanes = wavelengths (lane, 64 + lane, ...).
 *
 * For each of the pixel's samples in the batch, in order: the path's vertex records arrive by one
 * coalesced vector load per 64 words (the NEXT sample's load is already in flight while this one is
 * replayed), every field is lifted into SGPRs with v_readlane, so the record decode and the BDSF
 * dispatch run on the scalar unit and the vector unit only does the per-wavelength f64 arithmetic:
 * cast_ray's spectral side (src/daily_ray_trace.c:440-472 with direct_light_contribution :272-332
 * inlined), the vignette (:615) and render_image's film update (:732-743) on accumulators kept in
 * registers. The film is read and written once per batch. Waves are persistent (SPD tables staged
 * into LDS once) and draw work items (pixel groups, or pieces of them: see the queue below) from a global counter,
 * because pixel cost varies.
 */
#ifndef SHADE_PREFETCH_REGS
#define SHADE_PREFETCH_REGS 2 /* 64-word registers per path: 128 record words are prefetched, deeper paths fall back */
#endif
#define SHADE_PREFETCH_DEPTH 1 /* samples whose record loads are in flight ahead of the one being replayed. Latency is hidden by the other
                                  waves at any depth (the kernel is bound by what it issues); one ahead is the fewest
                                  register moves and LDS writes: 106.0 ms against 106.8 with two */
#define SHADE_PIXEL_CHUNK 16 /* pixels per group when there is no tail pass (with one: 64 / tail wavelengths) */
/* A wave's own LDS region behind the SPD table: the main pass keeps two record slots there (2 x 64 x SHADE_PREFETCH_REGS words); the
 * tail pass, which a wave runs at other times, the headers of a window of samples of its group's pixels (TAIL_WINDOW_WORDS). */
#define TAIL_WINDOW_ENTRIES 112u /* headers of a window: three words each (words 0, 1, 2), and a 16-bit task list entry */
#define TAIL_WINDOW_WORDS (3u * TAIL_WINDOW_ENTRIES + TAIL_WINDOW_ENTRIES / 4u) /* 2.9 KB */
#define SHADE_WAVE_LDS_WORDS ((2u * 64u * SHADE_PREFETCH_REGS) > TAIL_WINDOW_WORDS ? (2u * 64u * SHADE_PREFETCH_REGS) : TAIL_WINDOW_WORDS)

#define XYZ_FILM_WORDS 8 /* XYZ film mode, per pixel: X, Y, Z numerators of the main pass, filter sum, X, Y, Z of the tail pass, unused */

/*
 * Tail pass. S = 69 leaves 5 wavelengths beyond the 64 lanes; giving them a second register set would cost every vector
 * instruction again for 5 useful lanes. Instead the tails of a group's pixels are packed into one wave: lane = (pixel g of
 * the group, tail wavelength j). Each lane replays ITS pixel's records, read per lane, with the same per-wavelength
 * arithmetic in the same order. It is a work item of the shade kernel's queue, sharing the SIMDs with main-pass waves
 * (as a kernel of its own, at 4 to 8 waves per SIMD, it was 12-27 ms slower: measured).
 */
template <bool SPDS_IN_LDS, bool XYZ, bool SIMPLE>
__device__ __forceinline__ void shade_tail_group(const DevScene &sc, const ShadeParams &sp, const double *lds, uint64_t *wave_lds, const uint64_t *__restrict__ records,
                                                 const uint64_t *__restrict__ headers, double *__restrict__ film_pixels,
                                                 double *__restrict__ film_avgs, double *__restrict__ film_vars, uint64_t chunk_base,
                                                 uint64_t chunk_end, uint32_t lane)
{
    const uint32_t S = sc.S;
    const uint32_t vw = sp.vertex_words;
    const uint32_t R = sp.tail_count;
    const uint32_t g = lane / R, j = lane - g * R;
    const uint64_t pix_l = chunk_base + g;
    const bool act = g < (uint32_t)(chunk_end - chunk_base) && g < 64u / R;
    const uint32_t lam = sp.tail_first + j; /* < S by construction */
    const double *table = SPDS_IN_LDS ? (const double *)lds : sc.spds;
    /*
     * Phase A -- radiance, paced per pixel. If the 12 pixels stepped through their samples together, every sample
     * would cost the longest of 12 paths (about 4.5 vertices where the average is 1.65), and every vertex step both
     * the two-lobe plastic's code and the general BDSF code, because some pixel or other is always on glass or gold.
     * Instead each pixel's lanes keep their own (sample, vertex) cursor; the header says which of the path's first 16
     * vertices are plastic (bits 48-63), so an iteration is EITHER a plastic step or a general step -- whichever more
     * lanes are waiting for -- and then closes the samples that are complete (emission, vignette; result parked in
     * tail_stage). The wave runs for the pixel with the most vertices in the batch, and the general code only runs
     * when it is what most lanes need.
     */
    /*
     * Phase A -- radiance. The work is a list of PATHS, dealt to whichever lane group is free.
     * A window of samples of the group's pixels is taken at a time (TAIL_WINDOW_ENTRIES headers: nine samples of twelve pixels at five
     * tail wavelengths). The whole wave fetches the window's headers in coalesced loads; a lane that finds its header without a vertex
     * -- three in five -- closes that sample at once (emission of what the path ended on, vignette: :452-457, :615, a few LDS reads and
     * stores), the others are entered in a task list in LDS (ballot + prefix count) with their header words beside it. Then every
     * group of R lanes -- one lane per tail wavelength -- takes the next task from the list whenever it is free, replays that path
     * vertex by vertex with its own cursor, parks the result in tail_stage and takes the next. (Before: a group walked ITS pixel's samples
     * in order, a memory round trip for every header -- the empty ones too -- and the wave ran as long as its pixel with the most
     * vertices: 220 iterations for twelve pixels' 64 samples where the vertices alone are 106 a group.) The header says which of a
     * path's first 16 vertices are plastic (bits 48-63), so an iteration is EITHER a plastic step or a general step -- whichever more
     * lanes are waiting for.
     */
    const uint32_t n_groups = 64u / R;                 /* lane groups = the pixels of a full chunk */
    const uint32_t win = TAIL_WINDOW_ENTRIES / n_groups; /* samples per window */
    uint16_t *task_list = (uint16_t *)(wave_lds + 3u * TAIL_WINDOW_ENTRIES);
    const uint32_t leader = g * R;                     /* the group's first lane */
    const bool in_group = g < n_groups;
    uint32_t v = 0, n_shaded = 0;
    uint64_t ph0 = 0; /* the header's first word: vertex count, how the path ended, flags (fields taken out where they are used) */
    double vignette = 0.0, throughput = 1.0, dst = 0.0;
    uint64_t ph2 = 0; /* the path's first two block words */
    uint64_t slot = 0; /* the path's header / staging slot: pixel * batch + sample */
    const uint64_t *vcur = records; /* the record of vertex v */
    const bool one_light = sp.n_lights == 1u;
    const double em0 = spd_at(table, S, one_light ? sp.light0_em_spd : 0u, lam); /* light 0's emission at this lane's wavelength */
    /* header word 3 (the blocks of vertices 8 and up) is fetched when a path gets that far */
    auto deep_blocks = [&](uint32_t vertex) -> uint64_t { return vertex >= 2u * REC_BLOCK_VERTICES ? headers[slot * REC_HEADER_WORDS + 3] : 0ull; };
    for (uint32_t s0 = sp.tail_staged ? sp.n_samples : 0u; s0 < sp.n_samples; s0 += win) /* tail_staged: nothing to replay, drt_trace_kernel<true, true> staged every sample */
    {
        const uint32_t s_end = s0 + win < sp.n_samples ? s0 + win : sp.n_samples;
        const uint32_t n_win = s_end - s0, n_px = (uint32_t)(chunk_end - chunk_base);
        uint32_t n_tasks = 0; /* wave-uniform */
        for (uint32_t e0 = 0; e0 < n_px * n_win; e0 += 64u)
        {
            const uint32_t e = e0 + lane;
            const bool valid = e < n_px * n_win;
            const uint32_t ge = valid ? e / n_win : 0u, off = valid ? e - ge * n_win : 0u;
            const uint64_t eslot = (chunk_base + ge) * (uint64_t)sp.batch + s0 + off;
            uint64_t h0 = 0, h1 = 0, h2 = 0;
            if (valid)
            {
                const uint64_t *h = headers + eslot * REC_HEADER_WORDS;
                h0 = h[0];
                h1 = h[1];
                h2 = h[2];
            }
            /* a path whose tail wavelengths the trace kernel has carried itself is done: its values are in tail_stage already */
            const bool staged = ((uint32_t)(h0 >> 16) & HDR_TERM_TAIL_STAGED) != 0u;
            const bool has_path = valid && !staged && (h0 & 0xFFFFu) != 0u;
            if (valid && !staged && !has_path)
            {
                /* no vertex: the sample is what the path ended on, with throughput 1 and nothing gathered (:452-457, :615) */
                const bool emissive = ((uint32_t)(h0 >> 16) & HDR_TERM_MASK) == 1u;
                const uint32_t row = (uint32_t)(h0 >> 32) & 0xFFFFu;
                for (uint32_t t = 0; t < R; t += 1)
                {
                    double d = 0.0;
                    if (emissive) d = d + 1.0 * spd_at(table, S, row, sp.tail_first + t);
                    sp.tail_stage[eslot * R + t] = d * word_as_double(h1);
                }
            }
            const unsigned long long m = __ballot(has_path);
            if (has_path)
            {
                const uint32_t k = n_tasks + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                task_list[k] = (uint16_t)e;
                uint64_t *w = wave_lds + e * 3u;
                w[0] = h0;
                w[1] = h1;
                w[2] = h2;
            }
            n_tasks += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier(); /* the other lanes' LDS writes are in order before this lane's reads (one wave, in-order LDS) */
        uint32_t next_task = 0; /* wave-uniform */
        bool busy = false;      /* the group is on a path */
        for (;;)
        {
            /* groups without a path take the next ones off the list, in lane order */
            const unsigned long long want = __ballot(!busy && in_group && lane == leader);
            if (want != 0ull)
            {
                const uint32_t mine = next_task + (uint32_t)__popcll(want & ((1ull << leader) - 1ull));
                if (!busy && in_group && mine < n_tasks)
                {
                    const uint32_t e = task_list[mine];
                    const uint64_t *w = wave_lds + e * 3u;
                    const uint32_t ge = e / n_win, off = e - ge * n_win;
                    slot = (chunk_base + ge) * (uint64_t)sp.batch + s0 + off;
                    ph0 = w[0];
                    vignette = word_as_double(w[1]);
                    ph2 = w[2];
                    n_shaded = (uint32_t)(ph0 & 0xFFFFu);
                    v = 0;
                    vcur = records + (uint64_t)(uint32_t)ph2 * sp.block_words; /* vertex 0 opens the header's first block */
                    throughput = 1.0;
                    dst = 0.0;
                    busy = true;
                }
                next_task += (uint32_t)__popcll(want);
            }
            if (!__any(busy)) break;
            const bool has_vertex = busy && v < n_shaded;
            const bool is_plastic = has_vertex && v < 16u && (((uint32_t)(ph0 >> 48) >> v) & 1u); /* bits 48-63: two-lobe plastic */
            const bool is_general = has_vertex && !is_plastic;
            const uint32_t n_plastic = (uint32_t)__popcll(__ballot(is_plastic)), n_general = (uint32_t)__popcll(__ballot(is_general));
            if (n_plastic > 0 && n_plastic >= n_general)
            {
                if (is_plastic)
                {
                    /* bdsf() over {bp_diffuse_bdsf, bp_glossy_bdsf}, straight-line (as in the main pass) */
                    const uint64_t *vrec = vcur;
                    const uint64_t w1 = vrec[1];
                    const double dir_pdf = word_as_double(vrec[4]);
                    const double s_a_in = word_as_double(vrec[5]), s_spec = word_as_double(vrec[6]);
                    const uint32_t i_diffuse = (uint32_t)(w1 >> 16) & 0xFFFFu, i_glossy = (uint32_t)(w1 >> 32) & 0xFFFFu;
                    const double diffuse_pi = spd_at(table, S, i_diffuse, lam), glossy = spd_at(table, S, i_glossy, lam);
                    double contribution = 0.0;
                    if (one_light && v < 8u)
                    {
                        /* one light, and the header says whether it is visible: its three numbers fetched together with the vertex's own
                         * words (no round trip for a flag word first), its emission row known in advance */
                        if (((uint32_t)(ph0 >> 24) >> v) & 1u) /* bits 24-31: light 0 visible */
                        {
                            const uint64_t *lrec = vrec + REC_VERTEX_WORDS;
                            const double c = word_as_double(lrec[1]), a_in = word_as_double(lrec[2]), spec = word_as_double(lrec[3]);
                            double reflectance = diffuse_pi * a_in + 0.0;
                            reflectance = (glossy * spec) * a_in + reflectance;
                            contribution = contribution + reflectance;
                            contribution = contribution * em0;
                            contribution = contribution * c;
                        }
                    }
                    else if (one_light)
                    {
                        /* later vertices: the flag word says whether the light is visible; it travels WITH the light's three numbers and
                         * the vertex's own words, and the result is computed either way and then chosen -- behind a branch the loads
                         * would wait for the flag word's round trip first */
                        const uint64_t *lrec = vrec + REC_VERTEX_WORDS;
                        const uint64_t lw0 = lrec[0];
                        const double c = word_as_double(lrec[1]), a_in = word_as_double(lrec[2]), spec = word_as_double(lrec[3]);
                        double reflectance = diffuse_pi * a_in + 0.0;
                        reflectance = (glossy * spec) * a_in + reflectance;
                        double lit = 0.0 + reflectance;
                        lit = lit * em0;
                        lit = lit * c;
                        contribution = ((uint32_t)(lw0 >> 16) & FLAG_VISIBLE) ? lit : 0.0;
                    }
                    else
                    for (uint32_t l = 0; l < sp.n_lights; l += 1)
                    {
                        const uint64_t *lrec = vrec + REC_VERTEX_WORDS + l * REC_LIGHT_WORDS;
                        const uint64_t lw0 = lrec[0];
                        if (!((uint32_t)(lw0 >> 16) & FLAG_VISIBLE)) continue;
                        const double c = word_as_double(lrec[1]), a_in = word_as_double(lrec[2]), spec = word_as_double(lrec[3]);
                        double reflectance = diffuse_pi * a_in + 0.0;
                        reflectance = (glossy * spec) * a_in + reflectance;
                        contribution = contribution + reflectance;
                        contribution = contribution * spd_at(table, S, (uint32_t)(lw0 & 0xFFFFu), lam);
                        contribution = contribution * c;
                    }
                    dst = dst + throughput * contribution;
                    double reflectance = diffuse_pi * s_a_in + 0.0;
                    reflectance = (glossy * s_spec) * s_a_in + reflectance;
                    reflectance = reflectance * dir_pdf;
                    throughput = throughput * reflectance;
                    v += 1;
                    if (v < n_shaded) vcur = (v & (REC_BLOCK_VERTICES - 1u)) != 0u ? vcur + vw : path_vertex(records, sp.block_words, vw, ph2, deep_blocks(v), v); /* the next record: the one behind, or a new block */
                }
            }
            else if (n_general > 0)
            {
                if (is_general)
                {
                    /* any BDSF list (also plastic vertices beyond the header's 16 flags) */
                    const uint64_t *vrec = vcur;
                    const uint64_t list = vrec[0], w1 = vrec[1], w2 = vrec[2];
                    const double on_dot = word_as_double(vrec[3]), dir_pdf = word_as_double(vrec[4]);
                    const double s_a_in = word_as_double(vrec[5]), s_spec = word_as_double(vrec[6]);
                    const uint32_t num_bdsfs = (uint32_t)(w1 & 0xFFu);
                    const uint32_t sflags = (uint32_t)(w1 >> 8) & 0xFFu;
                    const uint32_t i_diffuse = (uint32_t)(w1 >> 16) & 0xFFFFu, i_glossy = (uint32_t)(w1 >> 32) & 0xFFFFu;
                    const double diffuse_pi = spd_at(table, S, i_diffuse, lam), glossy = spd_at(table, S, i_glossy, lam);
                    const double mirror = spd_at(table, S, (uint32_t)(w1 >> 48) & 0xFFFFu, lam);
                    uint32_t f0, f1, f2;
                    bool paired;
                    fresnel_rows(w2, f0, f1, f2, paired);
                    const double ir = SIMPLE ? 0.0 : spd_at(table, S, f0, lam), tr = SIMPLE ? 0.0 : spd_at(table, S, f1, lam), te = SIMPLE ? 0.0 : spd_at(table, S, f2, lam);
                    double contribution = 0.0;
                    for (uint32_t l = 0; l < sp.n_lights; l += 1)
                    {
                        const uint64_t *lrec = vrec + REC_VERTEX_WORDS + l * REC_LIGHT_WORDS;
                        const uint64_t lw0 = lrec[0];
                        const uint32_t lflags = (uint32_t)(lw0 >> 16) & 0xFFu;
                        if (!(lflags & FLAG_VISIBLE)) continue;
                        double reflectance = bdsf_at_wavelength<SIMPLE>(list, num_bdsfs, diffuse_pi, glossy, mirror, ir, tr, te, on_dot,
                                                                word_as_double(lrec[2]), word_as_double(lrec[3]), word_as_double(lrec[4]),
                                                                word_as_double(lrec[5]), lflags, paired);
                        contribution = contribution + reflectance;
                        contribution = contribution * spd_at(table, S, (uint32_t)(lw0 & 0xFFFFu), lam);
                        contribution = contribution * word_as_double(lrec[1]);
                    }
                    dst = dst + throughput * contribution;
                    double reflectance = bdsf_at_wavelength<SIMPLE>(list, num_bdsfs, diffuse_pi, glossy, mirror, ir, tr, te, on_dot, s_a_in, s_spec,
                                                            word_as_double(vrec[7]), word_as_double(vrec[8]), sflags, paired);
                    reflectance = reflectance * dir_pdf;
                    throughput = throughput * reflectance;
                    v += 1;
                    if (v < n_shaded) vcur = (v & (REC_BLOCK_VERTICES - 1u)) != 0u ? vcur + vw : path_vertex(records, sp.block_words, vw, ph2, deep_blocks(v), v); /* the next record: the one behind, or a new block */
                }
            }
            if (busy && v >= n_shaded)
            {
                /* the path's last vertex is done: close the sample, :452-457 and :615 */
                if (((uint32_t)(ph0 >> 16) & HDR_TERM_MASK) == 1u) dst = dst + throughput * spd_at(table, S, (uint32_t)(ph0 >> 32) & 0xFFFFu, lam);
                sp.tail_stage[slot * R + j] = dst * vignette;
                busy = false;
            }
        }
        __builtin_amdgcn_wave_barrier(); /* every lane is through the window before its headers are overwritten */
    }
    double *px = film_pixels + pix_l * (uint64_t)(XYZ ? XYZ_FILM_WORDS : S + 1);
    double *pa = XYZ ? nullptr : film_avgs + pix_l * (uint64_t)S;
    double *pv = XYZ ? nullptr : film_vars + pix_l * (uint64_t)S;
    const double *stage = sp.tail_stage + (pix_l * (uint64_t)sp.batch) * R + j;
    /* Phase B -- the film update (src/daily_ray_trace.c:732-743), the pixels' samples in order, all pixels in step */
    double f_sum = (act && !XYZ) ? px[lam] : 0.0, f_avg = (act && !XYZ) ? pa[lam] : 0.0, f_var = (act && !XYZ) ? pv[lam] : 0.0;
#pragma unroll 4
    for (uint32_t k = 0; k < sp.n_samples; k += 1)
    {
        const double contribution = act ? stage[(uint64_t)k * R] : 0.0;
        const double denom = (double)(sp.first_sample + k + 1);
        f_sum = f_sum + contribution;
        if (!XYZ)
        {
            double t0 = contribution - f_avg;
            double t1 = t0;
            t0 = t0 / denom;
            f_avg = f_avg + t0;
            t0 = contribution - f_avg;
            t0 = t1 * t0;
            f_var = f_var + t0;
        }
    }
    if (XYZ)
    {
        /* the pixel's tail wavelengths, summed over its R lanes by the group's first lane */
        const double rw = spd_at(table, S, sp.cmf_rw, lam);
        const double x = act ? (spd_at(table, S, sp.cmf_x, lam) * f_sum * rw) : 0.0;
        const double y = act ? (spd_at(table, S, sp.cmf_y, lam) * f_sum * rw) : 0.0;
        const double z = act ? (spd_at(table, S, sp.cmf_z, lam) * f_sum * rw) : 0.0;
        double X = 0.0, Y = 0.0, Z = 0.0;
        for (uint32_t t = 0; t < R; t += 1)
        {
            const int src = (int)((lane - j + t) & 63u);
            X += __shfl(x, src);
            Y += __shfl(y, src);
            Z += __shfl(z, src);
        }
        if (act && j == 0)
        {
            px[4] += X;
            px[5] += Y;
            px[6] += Z;
        }
    }
    else if (act)
    {
        px[lam] = f_sum;
        pa[lam] = f_avg;
        pv[lam] = f_var;
    }
}


template <int NSETS, bool SPDS_IN_LDS, bool XYZ, bool DARK, bool SIMPLE = false>
__global__ __launch_bounds__(SHADE_BLOCK, (SIMPLE && NSETS == 1) ? DRT_SHADE_WAVES_SIMPLE : SHADE_WAVES_PER_SIMD_FOR(NSETS)) void drt_shade_kernel(DevScene sc, ShadeParams sp, const uint64_t *__restrict__ records,
                                                                 const uint64_t *__restrict__ headers, double *__restrict__ film_pixels,
                                                                 double *__restrict__ film_avgs, double *__restrict__ film_vars,
                                                                 unsigned long long *__restrict__ work_counter)
{
    extern __shared__ double lds[];
    const uint32_t S = sc.S;
    if (*sp.overflow) return; /* the records of this launch are incomplete: leave the film as it is (see the record layout) */
    if (SPDS_IN_LDS)
    {
        /* the SPD block [n_spd][S] is contiguous: coalesced copy */
        const uint32_t spd_words = sc.n_spd * S;
        for (uint32_t k = threadIdx.x; k < spd_words; k += SHADE_BLOCK) lds[k] = sc.spds[k];
        __syncthreads();
    }
    const uint32_t lane = threadIdx.x & 63u;
    /* behind the SPD table: two record slots per wave (see the sample loop) */
    uint64_t *rec_lds = (uint64_t *)(lds + (SPDS_IN_LDS ? (size_t)sc.n_spd * S : 0)) + (size_t)(threadIdx.x >> 6) * SHADE_WAVE_LDS_WORDS;
    const uint32_t vw = sp.vertex_words;                 /* power of two >= 16 */
    const uint32_t vpr = vw <= 64 ? 64u / vw : 0u;       /* vertices per 64-word register (0: records wider than a register) */
    const uint32_t n_fast_regs = vpr * SHADE_PREFETCH_REGS;
    const uint32_t n_fast = n_fast_regs < 2u * REC_BLOCK_VERTICES ? n_fast_regs : 2u * REC_BLOCK_VERTICES; /* vertices covered by the prefetch registers (blocks 0 and 1) */

    uint32_t lam_c[NSETS];
#pragma unroll
    for (int k = 0; k < NSETS; k += 1)
    {
        const uint32_t lam = 64u * k + lane;
        lam_c[k] = lam < S ? lam : 0; /* lanes past the table read row element 0 and are never stored */
    }

    const uint32_t S_main = sp.tail_count ? sp.tail_first : S; /* wavelengths the lane-per-wavelength pass covers */
    for (;;)
    {
        /* draw the next work item (wave-uniform): part of a pixel group's main pass, its tail pass, or both */
        uint32_t item = 0;
        if (lane == 0) item = (uint32_t)atomicAdd(work_counter, 1ull);
        item = (uint32_t)__builtin_amdgcn_readfirstlane((int)item);
        if (item >= sp.n_items) break;
        /* split queue: a group's tail pass is the longest item, so the tail items are dealt out early -- one at the head of
         * every period of 1 + tail_period_mains items, between main-pass pieces so that latency-bound tail waves and
         * arithmetic-bound main waves share the SIMDs -- and the launch ends on main-pass pieces only */
        const bool split = sp.items_per_group > 1;
        const bool inline_tail = sp.tail_count != 0;
        const uint32_t mains_per_group = sp.items_per_group - (inline_tail ? 1u : 0u);
        bool tail_item = inline_tail;
        uint32_t group = item, sub = 0;
        if (split)
        {
            uint32_t m = item; /* index among the main-pass pieces */
            if (inline_tail)
            {
                const uint32_t n_groups = sp.n_items / sp.items_per_group;
                const uint32_t q = sp.tail_period_mains, mixed = n_groups * (q + 1u);
                if (item < mixed)
                {
                    const uint32_t period = item / (q + 1u), r = item - period * (q + 1u);
                    tail_item = r == 0u;
                    group = period;
                    m = period * q + (r - 1u);
                }
                else
                {
                    tail_item = false;
                    m = n_groups * q + (item - mixed);
                }
            }
            if (!tail_item)
            {
                group = m / mains_per_group;
                sub = m - group * mains_per_group;
            }
        }
        const uint64_t chunk_base = (uint64_t)group * sp.chunk;
        const uint64_t chunk_end = (chunk_base + sp.chunk < sp.n_pix) ? chunk_base + sp.chunk : sp.n_pix;
        uint64_t main_base = chunk_base, main_end = chunk_end;
        if (split)
        {
            main_base = chunk_base + (uint64_t)sub * sp.sub_pixels;
            main_end = (main_base + sp.sub_pixels < chunk_end) ? main_base + sp.sub_pixels : chunk_end;
            if (tail_item || main_base > chunk_end) main_base = main_end = chunk_end; /* nothing for the main pass */
        }

      if (sp.mode == 2u) main_base = main_end;
      for (uint64_t pix = main_base; pix < main_end; pix += 1)
      {

        /* spectral film: the pixel's [S+1] / [S] / [S] rows; XYZ film: its XYZ_FILM_WORDS accumulators, and the batch's
         * spectral sum starts from zero and is folded into them at the end */
        double *px = film_pixels + pix * (uint64_t)(XYZ ? XYZ_FILM_WORDS : S + 1);
        double *pa = XYZ ? nullptr : film_avgs + pix * (uint64_t)S;
        double *pv = XYZ ? nullptr : film_vars + pix * (uint64_t)S;
        double f_sum[NSETS], f_avg[NSETS], f_var[NSETS];
#pragma unroll
        for (int k = 0; k < NSETS; k += 1)
        {
            const uint32_t lam = 64u * k + lane;
            const bool active = lam < S_main;
            f_sum[k] = (active && !XYZ) ? px[lam] : 0.0;
            f_avg[k] = (active && !XYZ) ? pa[lam] : 0.0;
            f_var[k] = (active && !XYZ) ? pv[lam] : 0.0;
        }
        /* A pixel nothing has reached yet: every accumulator is +0 (bit pattern zero). A sample worth +0 or -0 then changes nothing
         * -- sum 0 + (+-0) = +0; mean: (+-0 - 0) / n = +-0, 0 + (+-0) = +0; variance: (+-0)(+-0) = +0, 0 + 0 = +0 -- so its update
         * (seven f64 operations, one of them a division) is skipped, exactly. The camera rays of most pixels that see nothing all
         * leave the scene: three samples in five on the Cornell frame, more among the 10 000 spheres. (DARK: an instantiation of its
         * own -- in a closed scene, where no pixel stays dark, the same loop with the test in it measured 2-3 % slower; the launcher
         * chooses by the vertices per path it measures when the context is created.) */
        bool dark = false;
        if (DARK)
        {
            bool some = false;
#pragma unroll
            for (int k = 0; k < NSETS; k += 1)
                some = some || ((uint64_t)__double_as_longlong(f_sum[k]) | (uint64_t)__double_as_longlong(f_avg[k]) | (uint64_t)__double_as_longlong(f_var[k])) != 0ull;
            dark = !__any(some);
        }
        /* the pixel's samples in windows of 64: the headers of a window arrive in one coalesced load, lane s <- sample s */
      for (uint32_t s0 = 0; s0 < sp.n_samples; s0 += 64u)
      {
        const uint32_t n_win = (uint32_t)__builtin_amdgcn_readfirstlane((int)((sp.n_samples - s0 < 64u) ? sp.n_samples - s0 : 64u)); /* 1..64, and in a scalar register: the loop's end test is then scalar too */
        __builtin_assume(n_win >= 1u);
        uint64_t h0 = 0, h1 = 0, h2 = 0, h3 = 0;
        if (lane < n_win)
        {
            const uint64_t *h = headers + (pix * sp.batch + s0 + lane) * REC_HEADER_WORDS;
            h0 = h[0];
            h1 = h[1];
            h2 = h[2];
            h3 = h[3];
        }
        /* the window's samples that are worth +-0 whatever the wavelength: no vertex, ended on nothing that emits, and a finite vignette
         * (0 x vignette must be a zero: a NaN or an infinity has to go through the arithmetic) -- one ballot over the headers */
        const unsigned long long worthless = !DARK ? 0ull : __ballot(lane < n_win && (h0 & 0xFFFFu) == 0ull && ((uint32_t)(h0 >> 16) & HDR_TERM_MASK) != 1u &&
                                                                      __builtin_isfinite(word_as_double(h1)));
        /* a whole window of them in a pixel that is still dark: nothing to do for any of its samples */
        if (DARK && dark && (uint32_t)__popcll(worthless) == n_win) continue;
        /* The first vertices of a path as the prefetch registers see them: register k, lane l holds word 64 k + l of the path's
         * vertices laid end to end. A block is four vertices, 64 words or more, so a register never straddles blocks: one block
         * number per register, from the header (scalar), and consecutive lanes read consecutive words. Only the header's own
         * blocks are prefetched; deeper vertices are fetched when they are replayed. */
        auto prefetch = [&](uint32_t sa, uint32_t nw, uint64_t *dst) {
            if (nw == 0u) return; /* three paths in five have no shaded vertex: nothing to fetch, and what the registers hold is never read */
            const uint64_t b2 = readlane64(h2, sa);
#pragma unroll
            for (int k = 0; k < SHADE_PREFETCH_REGS; k += 1)
            {
                const uint32_t b = (64u * k) >> (sp.vertex_shift + REC_BLOCK_SHIFT); /* 0 or 1: SHADE_PREFETCH_REGS <= 2 */
                const uint32_t id = b == 0 ? (uint32_t)b2 : (uint32_t)(b2 >> 32);
                const uint64_t *base = records + (uint64_t)id * sp.block_words + (64u * k - (b << (sp.vertex_shift + REC_BLOCK_SHIFT)));
                dst[k] = (64u * k + lane < nw) ? base[lane] : 0;
            }
        };
        /* ring of prefetched records: ring[0] = the sample being replayed, ring[d] = d samples ahead. Memory
         * latency (~2 us) is several samples of replay, so the loads run SHADE_PREFETCH_DEPTH samples ahead. */
        uint64_t ring[SHADE_PREFETCH_DEPTH + 1][SHADE_PREFETCH_REGS];
#pragma unroll
        for (int d = 0; d < SHADE_PREFETCH_DEPTH; d += 1)
        {
            const uint32_t nw = ((uint32_t)d < n_win) ? (uint32_t)(readlane64(h0, d) & 0xFFFFu) * vw : 0u;
            prefetch((uint32_t)d < n_win ? (uint32_t)d : 0u, nw, ring[d + 1]);
        }
        for (uint32_t s = 0; s < n_win; s += 1)
        {
            const uint64_t hs = readlane64(h0, s);
            const double vignette = word_as_double(readlane64(h1, s));
            const uint32_t n_shaded = (uint32_t)(hs & 0xFFFFu);
            const uint32_t term = (uint32_t)(hs >> 16) & HDR_TERM_MASK;
            const uint32_t term_spd = (uint32_t)(hs >> 32) & 0xFFFFu;
            const uint32_t plastic_mask = (uint32_t)(hs >> 48); /* bit v: vertex v has the two-lobe plastic list */
            const uint32_t vis0_mask = (uint32_t)(hs >> 24) & 0xFFu; /* bit v (< 8): light 0 visible from vertex v */
            /* rotate the ring, then start the load for the sample SHADE_PREFETCH_DEPTH ahead */
#pragma unroll
            for (int d = 0; d < SHADE_PREFETCH_DEPTH; d += 1)
#pragma unroll
                for (int k = 0; k < SHADE_PREFETCH_REGS; k += 1) ring[d][k] = ring[d + 1][k];
            {
                const uint32_t sa = s + SHADE_PREFETCH_DEPTH;
                const uint32_t nw = (sa < n_win) ? (uint32_t)(readlane64(h0, sa) & 0xFFFFu) * vw : 0u;
                prefetch(sa < n_win ? sa : 0u, nw, ring[SHADE_PREFETCH_DEPTH]);
            }
            const uint64_t *cur = ring[0];
            /* This sample's records (requested a sample ago) also go to this wave's LDS slot s & 1: the coefficient words of a
             * plastic vertex are then read back as broadcast LDS loads -- one instruction per 64-bit word, result in a vector
             * register where the f64 operations want it -- instead of two v_readlane each. */
            if (n_shaded != 0u)
            {
                uint64_t *slot_now = rec_lds + (s & 1u) * (64u * SHADE_PREFETCH_REGS);
#pragma unroll
                for (int k = 0; k < SHADE_PREFETCH_REGS; k += 1) slot_now[64u * k + lane] = ring[0][k];
            }
            const uint64_t *rec_words = rec_lds + (s & 1u) * (64u * SHADE_PREFETCH_REGS);
            /* nothing gathered, nothing emitted, into a pixel that is all +0: the film update would change no bit */
            if (DARK && dark && ((worthless >> s) & 1ull) != 0ull) continue;

            double throughput[NSETS], dst[NSETS];
#pragma unroll
            for (int k = 0; k < NSETS; k += 1)
            {
                throughput[k] = 1.0; /* const_spectrum(throughput, 1.0), :440 */
                dst[k] = 0.0;
            }
            uint32_t v_first = 0; /* vertices [0, v_first) are done by the loop for plastic runs below */
            if (NSETS == 1 && sp.n_lights == 1u && vpr != 0u)
            {
                /* The path's leading run of two-lobe plastic vertices -- most paths are nothing else -- in a loop of its own: the header's
                 * flags say how long the run is before a word of a record is decoded, so per vertex there is one word to lift to the
                 * scalar side (the SPD indices), one bit to test (light 0 visible), five LDS reads and the arithmetic; none of the
                 * general loop's questions (which register? plastic? how many lights? beyond the prefetch?). Same operations in the
                 * same order as plastic_vertex below. Measured and left out: issuing a vertex's LDS reads a vertex ahead, which this
                 * shape allows (113.7 against 113.5 ms), and taking the later runs of a path the same way (116.2). */
                uint32_t run = (uint32_t)__builtin_ctz(~plastic_mask | 0x10000u); /* leading plastic vertices (<= 16) */
                run = run < n_shaded ? run : n_shaded;
                run = run < n_fast ? run : n_fast;
                run = run < 8u ? run : 8u; /* vis0_mask has 8 bits */
                if (run != 0u)
                {
                const double *table1 = SPDS_IN_LDS ? (const double *)lds : sc.spds;
                const double em0 = spd_at(table1, S, sp.light0_em_spd, lam_c[0]);
                /* one vertex of a run: w1 = the record's word 1 (SPD indices), vwords = its words in LDS */
                auto run_vertex = [&](uint64_t w1, const uint64_t *vwords, bool visible) {
                    const double diffuse_pi = spd_at(table1, S, (uint32_t)(w1 >> 16) & 0xFFFFu, lam_c[0]);
                    const double glossy = spd_at(table1, S, (uint32_t)(w1 >> 32) & 0xFFFFu, lam_c[0]);
                    const double dir_pdf = word_as_double(vwords[4]), s_a_in = word_as_double(vwords[5]), s_spec = word_as_double(vwords[6]);
                    const double c = word_as_double(vwords[REC_VERTEX_WORDS + 1]), a_in = word_as_double(vwords[REC_VERTEX_WORDS + 2]);
                    const double spec = word_as_double(vwords[REC_VERTEX_WORDS + 3]);
                    double contribution = 0.0;
                    if (visible)
                    {
                        double reflectance = diffuse_pi * a_in + 0.0;
                        reflectance = (glossy * spec) * a_in + reflectance;
                        contribution = contribution + reflectance; /* :323 */
                        contribution = contribution * em0;         /* :324 */
                        contribution = contribution * c;           /* :326-327 */
                    }
                    dst[0] = dst[0] + throughput[0] * contribution; /* :461-462 */
                    double reflectance = diffuse_pi * s_a_in + 0.0;
                    reflectance = (glossy * s_spec) * s_a_in + reflectance;
                    reflectance = reflectance * dir_pdf;         /* :468 */
                    throughput[0] = throughput[0] * reflectance; /* :469 */
                };
                /* the vertices the first prefetch register holds, then those of the second: the word with the SPD indices comes out of
                 * ONE register by v_readlane, no choosing between two */
                const uint32_t run0 = run < vpr ? run : vpr;
                for (uint32_t v = 0; v < run0; v += 1)
                    run_vertex(readlane64(cur[0], v * vw + 1u), rec_words + v * vw, ((vis0_mask >> v) & 1u) != 0u);
                for (uint32_t v = run0; v < run; v += 1)
                    run_vertex(readlane64(cur[SHADE_PREFETCH_REGS > 1 ? 1 : 0], (v - vpr) * vw + 1u), rec_words + v * vw, ((vis0_mask >> v) & 1u) != 0u);
                v_first = run;
                /* Beyond the prefetched records (vertices 8 and up: long paths in closed scenes), when everything so far was a run: the
                 * path's further blocks one at a time -- four vertices by ONE coalesced load, which takes the place of vertices 0-3
                 * in this sample's LDS slot, then the same loop (visibility from the light's flag word, the header has 8 bits). A
                 * round trip per block instead of one per vertex, and none of the general loop's work. */
                if (run == n_fast && n_shaded > run && vw * REC_BLOCK_VERTICES == 64u)
                {
                    const uint64_t h3s = readlane64(h3, s);
                    for (uint32_t b = n_fast / REC_BLOCK_VERTICES; b * REC_BLOCK_VERTICES < n_shaded && b < 4u; b += 1) /* plastic_mask covers 16 vertices */
                    {
                        uint32_t id = (uint32_t)h3s;
                        if (b == 1u) id = (uint32_t)(readlane64(h2, s) >> 32);
                        if (b >= REC_HEADER_BLOCKS) id = ((const uint32_t *)(records + (uint64_t)(uint32_t)(h3s >> 32) * sp.block_words))[b - REC_HEADER_BLOCKS];
                        const uint64_t blockw = records[(uint64_t)id * sp.block_words + lane];
                        uint64_t *slot = rec_lds + (s & 1u) * (64u * SHADE_PREFETCH_REGS);
                        slot[lane] = blockw;
                        const uint32_t v_lo = b * REC_BLOCK_VERTICES;
                        const uint32_t v_hi = n_shaded < v_lo + REC_BLOCK_VERTICES ? n_shaded : v_lo + REC_BLOCK_VERTICES;
                        uint32_t v = v_lo;
                        for (; v < v_hi && ((plastic_mask >> v) & 1u); v += 1)
                        {
                            const uint32_t w0 = (v - v_lo) * vw;
                            const uint64_t lw0 = readlane64(blockw, w0 + REC_VERTEX_WORDS);
                            run_vertex(readlane64(blockw, w0 + 1u), slot + w0, ((uint32_t)(lw0 >> 16) & FLAG_VISIBLE) != 0u);
                        }
                        v_first = v;
                        if (v < v_hi) break; /* something else than plastic: the general loop takes over */
                    }
                }
                }
            }
            for (uint32_t v = v_first; v < n_shaded; v += 1)
            {
                /* the register (and the lane offset in it) that holds this vertex's record */
                uint64_t src;
                uint32_t lane0;
                if (v < n_fast)
                {
                    const uint32_t sel = v / vpr;
                    lane0 = (v - sel * vpr) * vw;
                    src = cur[0];
                    if (SHADE_PREFETCH_REGS > 1 && sel != 0) src = cur[SHADE_PREFETCH_REGS > 1 ? 1 : 0];
                }
                else
                {
                    /* deep path or a record wider than a register: fetch the first 64 words of the vertex now */
                    lane0 = 0;
                    src = (lane < vw) ? path_vertex(records, sp.block_words, vw, readlane64(h2, s), readlane64(h3, s), v)[lane] : 0;
                }
                const double *table = SPDS_IN_LDS ? (const double *)lds : sc.spds;
                double contribution[NSETS];
#pragma unroll
                for (int k = 0; k < NSETS; k += 1) contribution[k] = 0.0;
                /* the two-lobe plastic, known from the header's flags before a word of the vertex is decoded: only the
                 * words that list needs travel to the scalar unit */
                auto plastic_vertex = [&](uint64_t w1, double dir_pdf, double s_a_in, double s_spec) {
                    const uint32_t i_diffuse = (uint32_t)(w1 >> 16) & 0xFFFFu, i_glossy = (uint32_t)(w1 >> 32) & 0xFFFFu;
                    /* the common material, straight-line: bdsf() over {bp_diffuse_bdsf, bp_glossy_bdsf}
                         * = (glossy*spec)*a + ((diffuse/pi)*a + 0), src/bdsf.c:105-119 through src/daily_ray_trace.c:215-229 */
                        double diffuse_pi[NSETS], glossy[NSETS];
#pragma unroll
                        for (int k = 0; k < NSETS; k += 1)
                        {
                            diffuse_pi[k] = spd_at(table, S, i_diffuse, lam_c[k]);
                            glossy[k] = spd_at(table, S, i_glossy, lam_c[k]);
                        }
                        if (sp.n_lights == 1u && v < 8u && v < n_fast)
                        {
                            /* one light, and the header already says whether it is visible: its three numbers, nothing else
                             * (the emission row of light 0 is the same in every block: sp.light0_em_spd) */
                            if ((vis0_mask >> v) & 1u)
                            {
                                const uint32_t off = REC_VERTEX_WORDS;
                                const uint64_t *lw = rec_words + v * vw + off;
                                const double c = word_as_double(lw[1]), a_in = word_as_double(lw[2]), spec = word_as_double(lw[3]);
#pragma unroll
                                for (int k = 0; k < NSETS; k += 1)
                                {
                                    double reflectance = diffuse_pi[k] * a_in + 0.0;
                                    reflectance = (glossy[k] * spec) * a_in + reflectance;
                                    contribution[k] = contribution[k] + reflectance;                                    /* :323 */
                                    contribution[k] = contribution[k] * spd_at(table, S, sp.light0_em_spd, lam_c[k]); /* :324 */
                                    contribution[k] = contribution[k] * c;                                              /* :326-327 */
                                }
                            }
                        }
                        else
                        for (uint32_t l = 0; l < sp.n_lights; l += 1) /* direct_light_contribution, :272-332 */
                        {
                            if (l == 0 && v < 8u && !((vis0_mask >> v) & 1u)) continue; /* known from the header: not visible, nothing to read */
                            const uint32_t off = REC_VERTEX_WORDS + l * REC_LIGHT_WORDS;
                            uint64_t lw[4];
                            if (off + REC_LIGHT_WORDS <= 64 || v < n_fast)
                            {
#pragma unroll
                                for (int k = 0; k < 4; k += 1) lw[k] = readlane64(src, lane0 + off + k);
                            }
                            else
                            {
                                const uint64_t *lp = path_vertex(records, sp.block_words, vw, readlane64(h2, s), readlane64(h3, s), v) + off;
#pragma unroll
                                for (int k = 0; k < 4; k += 1) lw[k] = readlane64(lp[k], 0);
                            }
                            if (!((uint32_t)(lw[0] >> 16) & FLAG_VISIBLE)) continue;
                            const uint32_t i_em = (uint32_t)(lw[0] & 0xFFFFu);
                            const double c = word_as_double(lw[1]), a_in = word_as_double(lw[2]), spec = word_as_double(lw[3]);
#pragma unroll
                            for (int k = 0; k < NSETS; k += 1)
                            {
                                double reflectance = diffuse_pi[k] * a_in + 0.0;
                                reflectance = (glossy[k] * spec) * a_in + reflectance;
                                contribution[k] = contribution[k] + reflectance;                      /* :323 */
                                contribution[k] = contribution[k] * spd_at(table, S, i_em, lam_c[k]); /* :324 */
                                contribution[k] = contribution[k] * c;                                /* :326-327 */
                            }
                        }
#pragma unroll
                        for (int k = 0; k < NSETS; k += 1)
                        {
                            dst[k] = dst[k] + throughput[k] * contribution[k]; /* :461-462 */
                            double reflectance = diffuse_pi[k] * s_a_in + 0.0;
                            reflectance = (glossy[k] * s_spec) * s_a_in + reflectance;
                            reflectance = reflectance * dir_pdf;         /* :468 */
                            throughput[k] = throughput[k] * reflectance; /* :469 */
                        }
                };
                if (v < 16u && ((plastic_mask >> v) & 1u))
                {
                    const uint64_t w1p = readlane64(src, lane0 + 1);
                    if (v < n_fast)
                    {
                        const uint64_t *vwords = rec_words + v * vw;
                        plastic_vertex(w1p, word_as_double(vwords[4]), word_as_double(vwords[5]), word_as_double(vwords[6]));
                        continue;
                    }
                    const double dir_pdf_p = word_as_double(readlane64(src, lane0 + 4));
                    const double s_a_in_p = word_as_double(readlane64(src, lane0 + 5)), s_spec_p = word_as_double(readlane64(src, lane0 + 6));
                    plastic_vertex(w1p, dir_pdf_p, s_a_in_p, s_spec_p);
                    continue;
                }
                const uint64_t list = readlane64(src, lane0 + 0);
                const uint64_t w1 = readlane64(src, lane0 + 1);
                const uint64_t w2 = readlane64(src, lane0 + 2);
                const double on_dot = word_as_double(readlane64(src, lane0 + 3));
                const double dir_pdf = word_as_double(readlane64(src, lane0 + 4));
                const uint32_t num_bdsfs = (uint32_t)(w1 & 0xFFu);
                const uint32_t sflags = (uint32_t)(w1 >> 8) & 0xFFu;
                const uint32_t i_diffuse = (uint32_t)(w1 >> 16) & 0xFFFFu, i_glossy = (uint32_t)(w1 >> 32) & 0xFFFFu;
                const uint32_t i_mirror = (uint32_t)(w1 >> 48) & 0xFFFFu;
                uint32_t i_ir, i_tr, i_te; /* or the pair's rows in their place */
                bool paired;               /* wave-uniform: the record word came through v_readlane */
                fresnel_rows(w2, i_ir, i_tr, i_te, paired);
                const double s_a_in = word_as_double(readlane64(src, lane0 + 5)), s_spec = word_as_double(readlane64(src, lane0 + 6));

                if (sflags & FLAG_PLASTIC) /* a plastic vertex beyond the header's 16 flags */
                {
                    plastic_vertex(w1, dir_pdf, s_a_in, s_spec);
                    continue;
                }

                /* every other material: the general BDSF list */
                double diffuse[NSETS], glossy[NSETS], mirror[NSETS], ir[NSETS], tr[NSETS], te[NSETS];
#pragma unroll
                for (int k = 0; k < NSETS; k += 1)
                {
                    diffuse[k] = spd_at(table, S, i_diffuse, lam_c[k]); glossy[k] = spd_at(table, S, i_glossy, lam_c[k]);
                    mirror[k] = spd_at(table, S, i_mirror, lam_c[k]);   ir[k] = SIMPLE ? 0.0 : spd_at(table, S, i_ir, lam_c[k]);
                    tr[k] = SIMPLE ? 0.0 : spd_at(table, S, i_tr, lam_c[k]); te[k] = SIMPLE ? 0.0 : spd_at(table, S, i_te, lam_c[k]);
                }
                for (uint32_t l = 0; l < sp.n_lights; l += 1) /* direct_light_contribution, :272-332 */
                {
                    const uint32_t off = REC_VERTEX_WORDS + l * REC_LIGHT_WORDS;
                    uint64_t lw[REC_LIGHT_WORDS];
                    if (off + REC_LIGHT_WORDS <= 64 || v < n_fast)
                    {
#pragma unroll
                        for (int k = 0; k < REC_LIGHT_WORDS; k += 1) lw[k] = readlane64(src, lane0 + off + k);
                    }
                    else
                    {
                        /* light block beyond the register: uniform loads straight from the record */
                        const uint64_t *lp = path_vertex(records, sp.block_words, vw, readlane64(h2, s), readlane64(h3, s), v) + off;
#pragma unroll
                        for (int k = 0; k < REC_LIGHT_WORDS; k += 1) lw[k] = readlane64(lp[k], 0);
                    }
                    const uint32_t lflags = (uint32_t)(lw[0] >> 16) & 0xFFu;
                    if (!(lflags & FLAG_VISIBLE)) continue;
                    const uint32_t i_em = (uint32_t)(lw[0] & 0xFFFFu);
                    const double c = word_as_double(lw[1]);
#pragma unroll
                    for (int k = 0; k < NSETS; k += 1)
                    {
                        double reflectance = bdsf_at_wavelength<SIMPLE>(list, num_bdsfs, diffuse[k], glossy[k], mirror[k], ir[k], tr[k], te[k], on_dot,
                                                                word_as_double(lw[2]), word_as_double(lw[3]), word_as_double(lw[4]),
                                                                word_as_double(lw[5]), lflags, paired);
                        contribution[k] = contribution[k] + reflectance;                      /* :323 */
                        contribution[k] = contribution[k] * spd_at(table, S, i_em, lam_c[k]); /* :324 */
                        contribution[k] = contribution[k] * c;                                /* :326-327 */
                    }
                }
                const double s_mn = word_as_double(readlane64(src, lane0 + 7)), s_ct = word_as_double(readlane64(src, lane0 + 8));
#pragma unroll
                for (int k = 0; k < NSETS; k += 1)
                {
                    dst[k] = dst[k] + throughput[k] * contribution[k]; /* :461-462 */
                    double reflectance = bdsf_at_wavelength<SIMPLE>(list, num_bdsfs, diffuse[k], glossy[k], mirror[k], ir[k], tr[k], te[k], on_dot,
                                                            s_a_in, s_spec, s_mn, s_ct, sflags, paired);
                    reflectance = reflectance * dir_pdf;         /* :468 */
                    throughput[k] = throughput[k] * reflectance; /* :469 */
                }
            }
            const double denom = (double)(sp.first_sample + s0 + s + 1);
            if (DARK) dark = false; /* (a sample that gets this far may leave something in the accumulators) */
#pragma unroll
            for (int k = 0; k < NSETS; k += 1)
            {
                if (term == 1) /* emissive black body hit, :452-457 */
                {
                    const double em = SPDS_IN_LDS ? spd_at((const double *)lds, S, term_spd, lam_c[k]) : spd_at(sc.spds, S, term_spd, lam_c[k]);
                    dst[k] = dst[k] + throughput[k] * em;
                }
                const double contribution = dst[k] * vignette; /* :615 */
                /* film update, src/daily_ray_trace.c:732-743 */
                f_sum[k] = f_sum[k] + contribution;
                if (!XYZ)
                {
                    double t0 = contribution - f_avg[k];
                    double t1 = t0;
                    t0 = t0 / denom;
                    f_avg[k] = f_avg[k] + t0;
                    t0 = contribution - f_avg[k];
                    t0 = t1 * t0;
                    f_var[k] = f_var[k] + t0;
                }
            }
        }
      }
        if (XYZ)
        {
            /* spectrum_to_xyz's sums (src/spectrum.c:58-66) over this batch's spectral sum: per lane, then across the wave */
            const double *table = SPDS_IN_LDS ? (const double *)lds : sc.spds;
            double X = 0.0, Y = 0.0, Z = 0.0;
#pragma unroll
            for (int k = 0; k < NSETS; k += 1)
            {
                const uint32_t lam = 64u * k + lane;
                if (lam < S_main)
                {
                    const double rw = spd_at(table, S, sp.cmf_rw, lam);
                    X += (spd_at(table, S, sp.cmf_x, lam) * f_sum[k] * rw);
                    Y += (spd_at(table, S, sp.cmf_y, lam) * f_sum[k] * rw);
                    Z += (spd_at(table, S, sp.cmf_z, lam) * f_sum[k] * rw);
                }
            }
            for (int off = 32; off > 0; off >>= 1)
            {
                X += __shfl_xor(X, off);
                Y += __shfl_xor(Y, off);
                Z += __shfl_xor(Z, off);
            }
            if (lane == 0)
            {
                px[0] += X;
                px[1] += Y;
                px[2] += Z;
                px[3] += (double)sp.n_samples * 1.0;
            }
        }
        else
        {
#pragma unroll
            for (int k = 0; k < NSETS; k += 1)
            {
                const uint32_t lam = 64u * k + lane;
                if (lam < S_main)
                {
                    px[lam] = f_sum[k];
                    pa[lam] = f_avg[k];
                    pv[lam] = f_var[k];
                }
            }
            if (lane == 0) px[S] += (double)sp.n_samples * 1.0; /* filter sum: += 1.0 per sample, :733 */
        }
      }

        if (tail_item && sp.mode != 1u) shade_tail_group<SPDS_IN_LDS, XYZ, SIMPLE>(sc, sp, (const double *)lds, rec_lds, records, headers, film_pixels, film_avgs, film_vars, chunk_base, chunk_end, lane);
    }
}

/* spectrum_to_xyz over the film (src/daily_ray_trace.c:15-23, src/spectrum.c:49-70): one thread per
 * pixel so the sums run in the reference's order. */
__global__ void drt_film_xyz_kernel(DevScene sc, uint32_t cmf_rw, uint32_t cmf_x, uint32_t cmf_y, uint32_t cmf_z, double interval,
                                    uint64_t n_pix, const double *__restrict__ film_pixels, double *__restrict__ xyz)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pix) return;
    const uint32_t S = sc.S;
    const double *rw = sc.spds + (size_t)cmf_rw * S, *cx = sc.spds + (size_t)cmf_x * S;
    const double *cy = sc.spds + (size_t)cmf_y * S, *cz = sc.spds + (size_t)cmf_z * S;
    const double *px = film_pixels + p * (uint64_t)(S + 1);
    double f = px[S];
    double n = 0.0;
    for (uint32_t i = 0; i < S; i += 1) n += (cy[i] * rw[i]);
    n *= interval;
    double X = 0.0, Y = 0.0, Z = 0.0;
    for (uint32_t i = 0; i < S; i += 1)
    {
        double v = px[i] / f;
        X += (cx[i] * v * rw[i]);
        Y += (cy[i] * v * rw[i]);
        Z += (cz[i] * v * rw[i]);
    }
    xyz[3 * p + 0] = X * (interval / n);
    xyz[3 * p + 1] = Y * (interval / n);
    xyz[3 * p + 2] = Z * (interval / n);
}

/*
 * Film -> the pixel bytes of the reference's .bmp outputs, on the device: spd_file_to_rgb_f64_pixels
 * (src/daily_ray_trace.c:1-28: divide by the filter sum when the buffer carries one), spectrum_to_xyz and the fixed XYZ -> linear RGB
 * matrix (src/spectrum.c:49-82), rgb_f64_to_rgb_u8 (src/win32_platform.c:136-147: clamp to [0,1], x 255, truncate; no gamma), stored
 * B, G, R, A as write_pixels_to_bmp lays them out (the alpha byte the reference leaves unset is 255). `which`: 0 = sum (/ filter),
 * 1 = running mean, 2 = variance divided by its largest sample (spectrum_normalise, src/spectrum.c:182-187, which render_image
 * applies before writing the variance file: src/daily_ray_trace.c:766-768). Same expressions in the same order as
 * host/drt_bmp.c, no contraction: the bytes are equal to the host path's.
 */
__global__ void drt_film_bgra_kernel(DevScene sc, uint32_t cmf_rw, uint32_t cmf_x, uint32_t cmf_y, uint32_t cmf_z, double interval,
                                     uint64_t n_pix, const double *__restrict__ film, int which, uint8_t *__restrict__ bgra)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pix) return;
    const uint32_t S = sc.S;
    const double *rw = sc.spds + (size_t)cmf_rw * S, *cx = sc.spds + (size_t)cmf_x * S;
    const double *cy = sc.spds + (size_t)cmf_y * S, *cz = sc.spds + (size_t)cmf_z * S;
    const double *px = film + p * (uint64_t)(which == 0 ? S + 1 : S);
    double div = 1.0;
    bool divide = false;
    if (which == 0) { div = px[S]; divide = true; }
    if (which == 2)
    {
        double highest = 0.0;
        for (uint32_t i = 0; i < S; i += 1) if (px[i] > highest) highest = px[i];
        div = highest;
        divide = true;
    }
    double n = 0.0;
    for (uint32_t i = 0; i < S; i += 1) n += (cy[i] * rw[i]);
    n *= interval;
    double X = 0.0, Y = 0.0, Z = 0.0;
    for (uint32_t i = 0; i < S; i += 1)
    {
        double v = divide ? px[i] / div : px[i];
        X += (cx[i] * v * rw[i]);
        Y += (cy[i] * v * rw[i]);
        Z += (cz[i] * v * rw[i]);
    }
    X *= (interval / n);
    Y *= (interval / n);
    Z *= (interval / n);
    double rgb[3];
    rgb[0] = (2.3706743 * X) - (0.9000405 * Y) - (0.4706338 * Z);
    rgb[1] = (-0.5138850 * X) + (1.4253036 * Y) + (0.0885814 * Z);
    rgb[2] = (0.0052982 * X) - (0.0146949 * Y) + (1.0093968 * Z);
    uint8_t out[3];
    for (int c = 0; c < 3; c += 1)
    {
        double f = rgb[c];
        f = (f < 0.0) ? 0.0 : f;
        f = (f > 1.0) ? 1.0 : f;
        out[c] = (f == f) ? (uint8_t)(f * 255.0) : (uint8_t)0; /* a NaN (0/0 in a pixel no path reached) converts to 0 on x86 too */
    }
    uchar4 o;
    o.x = out[2]; o.y = out[1]; o.z = out[0]; o.w = 255;
    ((uchar4 *)bgra)[p] = o;
}

/* XYZ film mode: accumulators -> per-pixel XYZ, the same normalisation as above (src/spectrum.c:52-69) */
__global__ void drt_xyz_finish_kernel(DevScene sc, uint32_t cmf_rw, uint32_t cmf_y, double interval, uint64_t n_pix,
                                      const double *__restrict__ film, double *__restrict__ xyz)
{
    uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pix) return;
    const uint32_t S = sc.S;
    const double *rw = sc.spds + (size_t)cmf_rw * S, *cy = sc.spds + (size_t)cmf_y * S;
    double n = 0.0;
    for (uint32_t i = 0; i < S; i += 1) n += (cy[i] * rw[i]);
    n *= interval;
    const double *f = film + p * (uint64_t)XYZ_FILM_WORDS;
    for (int c = 0; c < 3; c += 1) xyz[3 * p + c] = ((f[c] + f[4 + c]) / f[3]) * (interval / n);
}

/* arithmetic self-test (see drt_selftest_arith in include/drt_hip.h) */
__global__ void drt_selftest_kernel(int op, const double *a, const double *b, double *out, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (op)
    {
        case 0: out[i] = __builtin_sqrt(a[i]); break;
        case 1: out[i] = a[i] / b[i]; break;
        case 2:
        {
            double s, c;
            drt_sincos(a[i], s, c);
            out[2 * i] = s;
            out[2 * i + 1] = c;
            break;
        }
        case 3: out[i] = pow(a[i], b[i]); break;
        case 7: out[i] = drt_pow_shininess(a[i], b[i]); break; /* the glossy lobe's power (csrc/drt_device.h) */
        case 4:
        {
            uint64_t key = (uint64_t)__double_as_longlong(a[i]);
            uint64_t rs = drt_splitmix64(key);
            uint32_t draws = 0;
            double v = 0.0;
            for (int k = 0; k < 4; k += 1) v = drt_rng(rs, draws);
            out[i] = v;
            break;
        }
        case 5: /* issue-cost probes: a dependent f64 FMA chain on all lanes (5) or on lanes 0-4 of each wave only (6) */
        case 6:
        {
            double x = a[i], y = b[i];
            if (op == 5 || (threadIdx.x & 63u) < 5u)
            {
                for (int k = 0; k < 65536; k += 1) x = __builtin_fma(x, y, 1.0e-3);
            }
            out[i] = x;
            break;
        }
        default: out[i] = 0.0; break;
    }
}

/* Device-function self-test (see drt_selftest_unit in include/drt_hip.h): one record per thread, `in_stride` doubles in,
 * `out_stride` doubles out, each function called exactly as the trace / shade kernels call it. */
enum
{
    DRT_UNIT_LINE_SPHERE = 0,   /* in: o[3] d[3] c[3] r                    out: t                         src/geometry.c:123-146 */
    DRT_UNIT_LINE_PLANE,        /* in: o[3] d[3] p[3] n[3] u[3] v[3]       out: t                         src/geometry.c:157-182 */
    DRT_UNIT_REFLECT,           /* in: v[3] n[3]                           out: r[3]                      src/geometry.c:85-90   */
    DRT_UNIT_TRANSMIT,          /* in: v[3] n[3] ir tr                     out: t[3]                      src/geometry.c:92-106  */
    DRT_UNIT_ROTATION_BETWEEN,  /* in: v[3] w[3]                           out: m[9] (columns)            src/geometry.c:263-295 */
    DRT_UNIT_SAMPLE_SPHERE,     /* in: rng state (u64 bits)                out: p[3], state after (bits)  src/rng.c:14-23        */
    DRT_UNIT_SAMPLE_DISC,       /* in: rng state (u64 bits)                out: p[3], state after (bits)  src/rng.c:25-51        */
    DRT_UNIT_GGX,               /* in: sn[3] mn[3] roughness               out: D                         src/bdsf.c:3-20        */
    DRT_UNIT_GGX_ATT,           /* in: v[3] sn[3] mn[3] roughness          out: D * G1                    src/bdsf.c:22-42       */
    DRT_UNIT_FS_DIELECTRIC,     /* in: ir tr cos                           out: R                         src/bdsf.c:44-67       */
    DRT_UNIT_FS_CONDUCTOR,      /* in: ir tr te cos                        out: R                         src/bdsf.c:78-101      */
    DRT_UNIT_SEED_AND_DRAW,     /* in: path key (u64 bits)                 out: state (bits), first rng() src/rng.c:1-12, SURVEY 8a-R */
    DRT_UNIT_BVH_BOX,           /* in: o[3] d[3] lo[3] hi[3] (box in f32 values) out: the f32 slab test's entry bound, or < 0 = box rejected (bvh_box_entry) */
    DRT_UNIT_COUNT
};

__global__ void drt_unit_kernel(int func, const double *__restrict__ in, uint32_t in_stride, double *__restrict__ out,
                                uint32_t out_stride, uint64_t n)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double *a = in + i * in_stride;
    double *o = out + i * out_stride;
    auto V = [&](int k) { return v3(a[k], a[k + 1], a[k + 2]); };
    auto put = [&](int k, V3 v) { o[k] = v.x; o[k + 1] = v.y; o[k + 2] = v.z; };
    switch (func)
    {
        case DRT_UNIT_LINE_SPHERE: o[0] = line_sphere(V(0), V(3), V(6), a[9]); break;
        case DRT_UNIT_LINE_PLANE:
        {
            /* the per-plane constants as build_device_scene() derives them from the edge vectors */
            V3 u = V(12), v = V(15);
            double ul = v_length(u), vl = v_length(v);
            o[0] = line_plane(V(0), V(3), V(6), V(9), v_div(u, ul), v_div(v, vl), ul, vl);
            break;
        }
        case DRT_UNIT_REFLECT: put(0, v_reflect(V(0), V(3))); break;
        case DRT_UNIT_TRANSMIT: put(0, v_transmit(V(0), V(3), a[6], a[7])); break;
        case DRT_UNIT_ROTATION_BETWEEN:
        {
            M33 m = rotation_between(V(0), V(3));
            put(0, m.c[0]); put(3, m.c[1]); put(6, m.c[2]);
            break;
        }
        case DRT_UNIT_SAMPLE_SPHERE:
        case DRT_UNIT_SAMPLE_DISC:
        {
            uint64_t rs = (uint64_t)__double_as_longlong(a[0]);
            uint32_t draws = 0;
            put(0, func == DRT_UNIT_SAMPLE_SPHERE ? uniform_sample_sphere(rs, draws) : uniform_sample_disc(rs, draws));
            o[3] = __longlong_as_double((long long)rs);
            break;
        }
        case DRT_UNIT_GGX: o[0] = ggx(V(0), V(3), a[6]); break;
        case DRT_UNIT_GGX_ATT: o[0] = ggx_att(V(0), V(3), V(6), a[9]); break;
        case DRT_UNIT_FS_DIELECTRIC: o[0] = dielectric_reflectance(a[0], a[1], a[2], 1.0 - a[2] * a[2]); break; /* as the shade kernel calls it */
        case DRT_UNIT_FS_CONDUCTOR:
        {
            double c2 = a[3] * a[3];
            o[0] = conductor_reflectance(a[0], a[1], a[2], a[3], c2, 1.0 - c2);
            break;
        }
        case DRT_UNIT_SEED_AND_DRAW:
        {
            uint64_t rs = drt_splitmix64((uint64_t)__double_as_longlong(a[0]));
            uint32_t draws = 0;
            o[0] = __longlong_as_double((long long)rs);
            o[1] = drt_rng(rs, draws);
            break;
        }
        case DRT_UNIT_BVH_BOX:
        {
            BvhNode n;
            for (int k = 0; k < 3; k += 1)
            {
                n.lo[0][k] = (float)a[6 + k];
                n.hi[0][k] = (float)a[9 + k];
            }
            o[0] = (double)bvh_box_entry(n, 0, bvh_ray32(V(0), V(3)));
            break;
        }
        default: break;
    }
}
