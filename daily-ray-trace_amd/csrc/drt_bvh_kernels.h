/*
 * drt_bvh_kernels.h -- the trace stage for scenes behind the bounding-volume hierarchy (BASELINE config 5).
 *
 * Same arithmetic, records and statistics as drt_trace_kernel (drt_kernels.h); what differs is how rays meet the tree:
 *
 *   drt_primary_kernel   one wave = 64 consecutive path ids. Ids run over a pixel's samples first, so the wave's 64 camera
 *                        rays are nearly the same ray: they walk the tree TOGETHER -- one node per step for the whole wave
 *                        (uniform loads), every lane tests its own ray against it, a ballot decides where the wave goes, one
 *                        stack per wave. No lane waits for another lane's longer walk. Paths that leave the scene or end on
 *                        a light are finished here (header written); the others are queued, compacted by ballot + prefix
 *                        count, with their closest hit (surface index, distance).
 *   drt_bounce_kernel    one path per lane from the queue, persistent, idle lanes refilled by ballot + prefix count as in
 *                        drt_trace_kernel. Shadow rays and continuation rays are incoherent, so every lane walks the tree on
 *                        its own -- but as ONE kind of job: a loop iteration gives each lane a single traversal, the shadow
 *                        ray of its vertex's next light or the closest hit of its next ray, whichever its path needs, so the
 *                        two populations share the traversal instead of taking turns at half occupancy. Stacks live in LDS
 *                        (one word per level and lane), none in scratch.
 *
 * The hierarchy only prunes (drt_kernels.h): hit indices, statistics and RNG draw counts stay bit-exact.
 */
#pragma once

#include "drt_kernels.h"

#define BVH_LDS_STACK BVH_STACK /* levels: the builder refuses deeper trees */

struct PrimaryHit
{
    int32_t index, pad;
    double  min_dist;
};

/* scene_point from a closest-hit result: the part of find_ray_intersection after the scan, src/daily_ray_trace.c:366-402.
 * `ro` is the ray origin AFTER the vis_fudge offset (:339). */
__device__ __forceinline__ void hit_point_from_scan(const SceneView &sv, const DevScene &sc, HitPoint &ip, V3 ro, V3 rd, double min_dist, int index)
{
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

/* the start of path `pid`: pixel, sample, RNG state, camera ray -- the `started` block of drt_trace_kernel */
struct PathStart
{
    uint64_t q, s_local, hit_row, slot;
    uint32_t x, y, sample;
};
__device__ __forceinline__ PathStart path_start(const TraceParams &tp, uint64_t pid)
{
    PathStart ps;
    ps.q = pid / tp.n_samples; /* consecutive ids: the samples of one pixel */
    ps.s_local = pid - ps.q * tp.n_samples;
    ps.hit_row = ps.s_local * tp.n_pix + ps.q;
    uint32_t j = (uint32_t)(ps.q / tp.tile_w);
    uint32_t i = (uint32_t)(ps.q - (uint64_t)j * tp.tile_w);
    ps.x = tp.x0 + i;
    ps.y = tp.y0 + j * tp.row_stride;
    ps.sample = tp.first_sample + (uint32_t)ps.s_local;
    ps.slot = ps.q * (uint64_t)tp.batch + ps.s_local;
    return ps;
}
__device__ __forceinline__ uint64_t path_key(const TraceParams &tp, const PathStart &ps)
{
    return tp.seed + (((uint64_t)ps.sample * (uint64_t)tp.height + (uint64_t)ps.y) * (uint64_t)tp.width + (uint64_t)ps.x);
}

/* ---------------------------------------------------------------------------------------------- */
/* Primary rays: the wave walks the tree as one                                                     */

#define PRIMARY_BLOCK 256

#ifndef DRT_PRIMARY_WAVES_PER_SIMD
#define DRT_PRIMARY_WAVES_PER_SIMD 6 /* 80 registers, 28 bytes of scratch: config 5 trace stage 506 -> 500 ms (5 and 8 waves: 501, 500) */
#endif
__global__ __launch_bounds__(PRIMARY_BLOCK, DRT_PRIMARY_WAVES_PER_SIMD) void drt_primary_kernel(DevScene sc, DevCamera cam, TraceParams tp, uint64_t *__restrict__ headers,
                                                                    int32_t *__restrict__ hits, unsigned long long *__restrict__ counters,
                                                                    PrimaryHit *__restrict__ primary, uint64_t *__restrict__ queue,
                                                                    unsigned long long *__restrict__ queue_count)
{
    __shared__ int s_stack[PRIMARY_BLOCK / 64][BVH_LDS_STACK];
    SceneView sv;
    sv.n_surf = sc.n_surf; sv.n_lights = sc.n_lights;
    sv.surf = sc.surf; sv.lights = sc.lights; sv.surf_type = sc.surf_type; sv.surf_mat = sc.surf_mat;
    sv.light_type = sc.light_type; sv.light_mat = sc.light_mat; sv.mats = sc.mats;
    sv.bvh_nodes = sc.bvh_nodes; sv.bvh_leaf = sc.bvh_leaf;
    const uint32_t lane = threadIdx.x & 63u;
    int *stack = s_stack[threadIdx.x >> 6];
    const uint64_t n_packets = (tp.n_paths + 63u) / 64u;
    const uint64_t wave0 = (uint64_t)blockIdx.x * (PRIMARY_BLOCK / 64) + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * (PRIMARY_BLOCK / 64);
    uint32_t n_scans = 0, n_draws = 0, n_paths = 0;
    if (*tp.overflow) return; /* an earlier launch ran out of record blocks: the host renders from there again */

    for (uint64_t packet = wave0; packet < n_packets; packet += n_waves)
    {
        const uint64_t pid = packet * 64u + lane;
        const bool valid = pid < tp.n_paths;
        V3 ro = v3(0, 0, 0), rd = v3(0, 0, 1);
        PathStart ps = path_start(tp, valid ? pid : 0);
        uint64_t rs = 1;
        if (valid)
        {
            rs = drt_splitmix64(path_key(tp, ps));
            camera_ray(cam, tp.pixel_scheme, ps.x, ps.y, rs, n_draws, ro, rd);
            n_paths += 1;
            n_scans += 1;
        }
        /* find_ray_intersection, src/daily_ray_trace.c:334-364, for 64 rays at once */
        const V3 o = v_sum(ro, v_mul(rd, DRT_VIS_FUDGE));
        double min_dist = DRT_INF;
        int index = -1;
        const Ray32 r32 = bvh_ray32(o, rd);
        float lim = bvh_limit32(min_dist);
        int sp = 0;
        int cur = 0; /* wave-uniform */
        for (;;)
        {
            cur = __builtin_amdgcn_readfirstlane(cur); /* the whole wave is at this node: its words arrive by scalar loads */
            if (cur >= 0)
            {
                const BvhNode &n = sv.bvh_nodes[cur];
                int ref[2];
                float t[2];
                bool hit[2];
                bvh_children(n, r32, lim, ref, t, hit);
                hit[0] = hit[0] && valid;
                hit[1] = hit[1] && valid;
                const unsigned long long m0 = __ballot(hit[0]), m1 = __ballot(hit[1]);
                if (m0 != 0ull && m1 != 0ull)
                {
                    /* the child most lanes would enter first goes first; the other one waits on the wave's stack */
                    const int first1 = __popcll(__ballot(hit[1] && (!hit[0] || t[1] < t[0])));
                    const int first0 = __popcll(__ballot(hit[0] && (!hit[1] || !(t[1] < t[0]))));
                    const int near = first1 > first0 ? 1 : 0;
                    if (lane == 0) stack[sp] = ref[1 - near];
                    sp += 1;
                    cur = ref[near];
                    continue;
                }
                if (m0 != 0ull) { cur = ref[0]; continue; }
                if (m1 != 0ull) { cur = ref[1]; continue; }
            }
            else if (bvh_is_leaf(cur))
            {
                const int packed = -2 - cur;
                const int first = packed >> 3, count = (packed & 7) + 1;
                for (int k = 0; k < count; k += 1)
                {
                    const BvhLeafPrim &lp = sv.bvh_leaf[first + k];
                    if (!__any(valid && !sphere_certainly_missed(lp, r32, lim))) continue; /* no lane's ray comes near it */
                    double dist = leaf_distance(sv, lp, o, rd);
                    if (valid && (dist < min_dist || (dist == min_dist && (int)lp.index < index)))
                    {
                        min_dist = dist;
                        index = (int)lp.index;
                        lim = bvh_limit32(min_dist);
                    }
                }
            }
            if (sp == 0) break;
            sp -= 1;
            cur = __builtin_amdgcn_readfirstlane(stack[sp]); /* written by lane 0 of this wave, in program order */
        }

        /* cast_ray's first look at the hit, src/daily_ray_trace.c:449-457: paths that end here are finished here */
        bool queued = false;
        if (valid)
        {
            const uint32_t smat = index >= 0 ? sv.surf_mat[index] : sc.escape_mat;
            const DevMaterial &mat = sv.mats[smat];
            uint64_t *hdr = headers + ps.slot * REC_HEADER_WORDS;
            hdr[1] = (uint64_t)__double_as_longlong(v_dot(rd, cam.forward) * 1.0); /* vignette, :614 */
            if (tp.record_hits)
            {
                int32_t *h = hits + ((uint64_t)tp.hits_sample_offset * tp.n_pix + ps.hit_row) * tp.max_depth;
                h[0] = index;
                for (uint32_t d = 1; d < tp.max_depth; d += 1) h[d] = -2;
            }
            if (mat.is_black_body)
            {
                const uint32_t term = mat.is_emissive ? 1u : 0u;
                const uint32_t term_spd = mat.is_emissive ? ((uint32_t)mat.emission_spd & 0xFFFFu) : 0u;
                hdr[0] = ((uint64_t)term << 16) | ((uint64_t)term_spd << 32);
            }
            else
            {
                queued = true;
                primary[ps.slot].index = index;
                primary[ps.slot].min_dist = min_dist;
            }
        }
        const unsigned long long qm = __ballot(queued);
        if (qm != 0ull)
        {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(queue_count, (unsigned long long)__popcll(qm));
            base = __shfl(base, 0);
            if (queued) queue[base + (uint32_t)__popcll(qm & ((1ull << lane) - 1ull))] = pid;
        }
    }
    uint64_t vals[3] = {n_paths, n_scans, n_draws};
    const int slot_of[3] = {0, 1, 4};
    for (int k = 0; k < 3; k += 1)
    {
        uint64_t v = vals[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && v) atomicAdd(&counters[slot_of[k]], (unsigned long long)v);
    }
}

/* ---------------------------------------------------------------------------------------------- */
/* Bounces: one traversal job per lane and iteration                                                */

#define BOUNCE_BLOCK 256
#ifndef DRT_BOUNCE_WAVES_PER_SIMD
#define DRT_BOUNCE_WAVES_PER_SIMD 4 /* 128 registers: the walk fits them without a spill; the shading sections between walks spill (320 bytes of scratch per lane), and four waves hide the node fetches better than three pay for that: config 5 trace stage 532 -> 508 ms */
#endif

enum { JOB_NONE = 0, JOB_SHADOW = 1, JOB_CLOSEST = 2 };

/* One walk of the tree per lane: JOB_CLOSEST finds (min_dist, index) -- minimum distance, lowest surface index on ties, what
 * the reference's linear scan returns (src/daily_ray_trace.c:340-364); JOB_SHADOW answers "any surface nearer than limit?"
 * (:246-268). `stack` is this wave's LDS block, one word per level and lane. */
/* the next subtree off the lane's stack, or BVH_DONE. (Entries tagged with a lower bound of the subtree's entry distance, so that a
 * closest-hit walk drops what lies beyond a hit found since the push without fetching the node: 816 ms against 723 on config 5 --
 * nearer-child-first already leaves little to drop, and the tag's packing and the pop loop cost more.) */
__device__ __forceinline__ int stack_pop(const int *stack, uint32_t lane, int &sp)
{
    if (sp > 0) { sp -= 1; return stack[sp * 64 + lane]; }
    return BVH_DONE;
}

#define DRT_BVH_POSTPONE 4 /* leaves a lane may put aside before it has to wait for the wave's leaf phase (0: the plain while-while walk) */
#ifndef DRT_BVH_POSTPONE_EXIT
#define DRT_BVH_POSTPONE_EXIT 16 /* the node phase ends when fewer lanes than this are still walking and one of the others waits with a full queue */
#endif
#define BVH_QUEUE_WORDS ((DRT_BVH_POSTPONE > 0 ? DRT_BVH_POSTPONE : 1) * 64)

__device__ __forceinline__ void bvh_node_step(const SceneView &sv, int *stack, uint32_t lane, const Ray32 &r32, float lim, int &cur, int &sp)
{
    const BvhNode &n = sv.bvh_nodes[cur];
    int ref[2];
    float t[2];
    bool hit[2];
    bvh_children(n, r32, lim, ref, t, hit);
    if (hit[0] && hit[1])
    {
        const int near = t[1] < t[0] ? 1 : 0;
        stack[sp * 64 + lane] = ref[1 - near];
        sp += 1;
        cur = ref[near];
    }
    else if (hit[0]) cur = ref[0];
    else if (hit[1]) cur = ref[1];
    else cur = stack_pop(stack, lane, sp);
}

__device__ __forceinline__ void bvh_walk(const SceneView &sv, int *stack, int *leaf_queue, uint32_t lane, int job, V3 o, V3 d, double &limit, int &index,
                                         bool &occluded)
{
    const Ray32 r32 = bvh_ray32(o, d);
    float lim = bvh_limit32(limit);
    int sp = 0;
    int cur = job == JOB_NONE ? BVH_DONE : 0;
    /* "While-while" with leaves PUT ASIDE. In the plain form every lane walks inner nodes until it holds a leaf, and the distance to
     * the next leaf is so uneven from lane to lane (a handful of nodes on average, fifteen for the unluckiest of 64) that three
     * lanes in four wait. Here a lane that reaches a leaf notes it in a small queue of its own (LDS, DRT_BVH_POSTPONE entries) and
     * walks on; the node phase ends when nobody walks any more, or when few lanes do and one of the others waits with a full queue;
     * then the queued leaves are tested together -- the f32 bound of every one first, the f64 intersector for those it leaves in
     * the running -- and the walk resumes with the limit they gave. A leaf tested later than the plain form would have tested it
     * only delays pruning: the result is the same minimum with the same tie-break (closest hit), the same answer to "anything
     * nearer than the limit?" (shadow ray). */
    int nq = 0;
    for (;;)
    {
        for (;;)
        {
            if (bvh_is_leaf(cur) && nq < DRT_BVH_POSTPONE)
            {
                leaf_queue[nq * 64 + lane] = cur;
                nq += 1;
                cur = stack_pop(stack, lane, sp);
            }
            const bool walking = cur >= 0;
            const int n_walk = __popcll(__ballot(walking));
            if (n_walk == 0) break;
            if (n_walk < DRT_BVH_POSTPONE_EXIT && __any(bvh_is_leaf(cur) && nq == DRT_BVH_POSTPONE)) break;
            if (walking) bvh_node_step(sv, stack, lane, r32, lim, cur, sp);
        }
        if (!__any(nq > 0 || cur != BVH_DONE)) break;
        /* leaf phase, first half: the f32 bound of every queued leaf; those it does not rule out move to the front of the queue */
        int ns = 0;
        for (int k = 0; __any(k < nq); k += 1)
            if (k < nq)
            {
                const int leaf = leaf_queue[k * 64 + lane];
                const BvhLeafPrim *src = &sv.bvh_leaf[(-2 - leaf) >> 3];
                BvhLeafPrim b;
                b.c32[0] = src->c32[0]; b.c32[1] = src->c32[1]; b.c32[2] = src->c32[2]; b.reach32 = src->reach32;
                if (!sphere_certainly_missed(b, r32, lim))
                {
                    leaf_queue[ns * 64 + lane] = leaf;
                    ns += 1;
                }
            }
        nq = 0;
        /* second half: the f64 intersector (src/geometry.c:123-182) */
        for (int k = 0; __any(k < ns); k += 1)
            if (k < ns)
            {
                const BvhLeafPrim *src = &sv.bvh_leaf[(-2 - leaf_queue[k * 64 + lane]) >> 3];
                BvhLeafPrim lp;
                lp.index = src->index; lp.type = src->type;
                lp.f[0] = src->f[0]; lp.f[1] = src->f[1]; lp.f[2] = src->f[2]; lp.f[3] = src->f[3];
                const double dist = leaf_distance(sv, lp, o, d);
                if (job == JOB_CLOSEST)
                {
                    if (dist < limit || (dist == limit && (int)lp.index < index))
                    {
                        limit = dist;
                        index = (int)lp.index;
                        lim = bvh_limit32(limit);
                    }
                }
                else if (dist < limit) /* the reference breaks at the first occluder; which one does not matter */
                {
                    occluded = true;
                    ns = 0;
                    sp = 0;
                    cur = BVH_DONE;
                }
            }
    }
}

__global__ __launch_bounds__(BOUNCE_BLOCK, DRT_BOUNCE_WAVES_PER_SIMD) void drt_bounce_kernel(
    DevScene sc, DevCamera cam, TraceParams tp, uint64_t *__restrict__ records, uint64_t *__restrict__ headers, int32_t *__restrict__ hits,
    unsigned long long *__restrict__ counters, unsigned long long *__restrict__ work_counter, const PrimaryHit *__restrict__ primary,
    const uint64_t *__restrict__ queue, const unsigned long long *__restrict__ queue_count)
{
    __shared__ int s_stack[BOUNCE_BLOCK / 64][BVH_LDS_STACK * 64];
    __shared__ int s_leaf_queue[BOUNCE_BLOCK / 64][BVH_QUEUE_WORDS];
    SceneView sv;
    sv.n_surf = sc.n_surf; sv.n_lights = sc.n_lights;
    sv.surf = sc.surf; sv.lights = sc.lights; sv.surf_type = sc.surf_type; sv.surf_mat = sc.surf_mat;
    sv.light_type = sc.light_type; sv.light_mat = sc.light_mat; sv.mats = sc.mats;
    sv.bvh_nodes = sc.bvh_nodes; sv.bvh_leaf = sc.bvh_leaf;
    const uint32_t lane = threadIdx.x & 63u;
    int *stack = s_stack[threadIdx.x >> 6];
    int *leaf_queue = s_leaf_queue[threadIdx.x >> 6];
    if (*tp.overflow) return;
    const uint64_t n_work = *queue_count;
    const uint64_t CHUNK = tp.chunk;
    uint64_t chunk_next = 0, chunk_end = 0; /* wave-uniform */
    bool exhausted = false;

    uint32_t n_scans = 0, n_shaded = 0, n_shadow = 0, n_draws = 0;

    /* per-lane path state. A path is either AT A VERTEX (ip valid; lights [0, light) done) or ON A RAY with no vertex yet -- and then
     * the ray lives in ip's registers, origin in ip.position (it IS the vertex the ray leaves, src/daily_ray_trace.c:471) and
     * direction in ip.out: the two states never need both, and the kernel is short of registers. */
    bool alive = false, at_vertex = false;
    uint64_t rs = 1, hit_row = 0;
    uint32_t depth = 0, shaded = 0, light = 0;
    uint32_t masks = 0; /* bits 0-15: shaded vertex v has the two-lobe plastic list; bits 16-23: light 0 is visible from vertex v (header bits 48-63, 24-31) */
    uint64_t *hdr = nullptr;
    uint32_t blk = 0, tbl = 0; /* the pool block of the current pair of vertices; the path's table block (deep paths) */
    uint32_t spare = ~0u, spare_tbl = ~0u; /* a block (and, for deep paths, a table block) held ready: path_spare_block */
    bool new_vertex = false;   /* the path has arrived at a vertex and nothing of it is written yet: its first record opens it */
    WavePool wp = {0u, 0u, 0u};
    HitPoint ip;
    ip.position = ip.normal = ip.out = v3(0, 0, 0);
    ip.on_dot = 0.0;
    ip.surface_mat = ip.incident_mat = ip.transmit_mat = 0;
    ip.index = -1;

    /* A path at a vertex takes one iteration per light (the shadow job) and one more for its next ray (the closest-hit job), so a
     * wave whose lanes all started together stays IN STEP -- every lane on the same kind of job -- as long as new paths only join
     * at the start of that cycle. Then an iteration runs the sections of ONE kind of job with every live lane in them and skips the
     * others (their branches see no lane), instead of all of them half empty; a lane whose path ends waits for the cycle's start,
     * at most one iteration with one light. */
    const uint32_t cycle = sv.n_lights + 1u;
    uint32_t phase = 0;
    for (;;)
    {
        /* ---- refill idle lanes from the queue of paths whose first vertex is known ---- */
        unsigned long long idle_mask = __ballot(!alive);
        if (!exhausted && idle_mask == ~0ull) phase = 0; /* nobody left to stay in step with */
        if (idle_mask != 0ull && !exhausted && phase == 0u)
        {
            const uint32_t want = (uint32_t)__popcll(idle_mask);
            const uint32_t rank = (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
            uint64_t my = ~0ull; /* queue position this lane takes */
            uint64_t avail = chunk_end - chunk_next;
            uint32_t taken = 0;
            if (avail < want)
            {
                if (!alive && rank < avail) my = chunk_next + rank;
                taken = (uint32_t)avail;
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)CHUNK);
                base = __shfl(base, 0);
                if (base >= n_work)
                {
                    exhausted = true;
                    chunk_next = chunk_end = 0;
                }
                else
                {
                    chunk_next = base;
                    chunk_end = (base + CHUNK < n_work) ? base + CHUNK : n_work;
                    const uint64_t avail2 = chunk_end - chunk_next;
                    if (!alive && rank >= taken && (uint64_t)(rank - taken) < avail2) my = chunk_next + (rank - taken);
                    const uint32_t used = (want - taken < avail2) ? (want - taken) : (uint32_t)avail2;
                    chunk_next += used;
                }
            }
            else
            {
                if (!alive) my = chunk_next + rank;
                chunk_next += want;
            }
            if (my != ~0ull)
            {
                /* the path's start, recomputed (same arithmetic as the primary kernel), and its first vertex from the queue */
                const uint64_t pid = queue[my];
                const PathStart ps = path_start(tp, pid);
                hit_row = ps.hit_row;
                rs = drt_splitmix64(path_key(tp, ps));
                uint32_t camera_draws = 0; /* counted by the primary kernel */
                V3 ro, rd;
                camera_ray(cam, tp.pixel_scheme, ps.x, ps.y, rs, camera_draws, ro, rd);
                hdr = headers + ps.slot * REC_HEADER_WORDS;
                const PrimaryHit ph = primary[ps.slot];
                hit_point_from_scan(sv, sc, ip, v_sum(ro, v_mul(rd, DRT_VIS_FUDGE)), rd, ph.min_dist, ph.index);
                depth = 0;
                shaded = 0;
                light = 0;
                masks = 0;
                alive = true;
                at_vertex = true;
                new_vertex = true;
                n_shaded += 1; /* direct_light_contribution is entered for this vertex */
            }
        }
        if (!__any(alive)) break;
        phase = phase + 1u == cycle ? 0u : phase + 1u;
        /* a spare record block for the lanes whose path may open one at its next vertex (the whole wave takes part) */
        if (!path_spare_block(tp, wp, alive, shaded, spare, spare_tbl, lane))
        {
            hdr[0] = (uint64_t)HDR_TERM_NOT_DONE << 16;
            alive = false;
        }

        /* ---- one traversal job per lane ---- */
        int job = JOB_NONE;
        V3 jo = v3(0, 0, 0), jd = v3(0, 0, 1);
        double limit = DRT_INF;
        double light_c = 1.0; /* attenuation x area of the light sample, :326 */
        if (alive && at_vertex && light < sv.n_lights)
        {
            /* direct_light_contribution, src/daily_ray_trace.c:272-332: the sample of light `light` (drawn before the shadow test) */
            const uint32_t l = light;
            const uint32_t ltype = sv.light_type[l];
            const V3 lpos = v3(sv.lights[LF_PX * sv.n_lights + l], sv.lights[LF_PY * sv.n_lights + l], sv.lights[LF_PZ * sv.n_lights + l]);
            const double light_pdf = sv.lights[LF_PDF * sv.n_lights + l];
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
            /* points_mutually_visible, :238-270 */
            jd = v_normalise(v_sub(light_position, ip.position));
            jo = v_sum(ip.position, v_mul(jd, DRT_VIS_FUDGE));
            limit = v_length(v_sub(light_position, jo)) - DRT_VIS_FUDGE;
            light_c = attenuation * (light_pdf);
            job = JOB_SHADOW;
        }
        else if (alive && !at_vertex)
        {
            jd = ip.out; /* on a ray: its direction, and its origin in ip.position */
            jo = v_sum(ip.position, v_mul(jd, DRT_VIS_FUDGE)); /* :339 */
            job = JOB_CLOSEST;
        }
        int index = -1;
        bool occluded = false;
        bvh_walk(sv, stack, leaf_queue, lane, job, jo, jd, limit, index, occluded);

        /* ---- what the job was for ---- */
        if (job == JOB_CLOSEST)
        {
            /* cast_ray's loop head, :448-457 */
            hit_point_from_scan(sv, sc, ip, jo, jd, limit, index);
            n_scans += 1;
            if (tp.record_hits) hits[((uint64_t)tp.hits_sample_offset * tp.n_pix + hit_row) * tp.max_depth + depth] = ip.index;
            const DevMaterial &mat = sv.mats[ip.surface_mat];
            if (mat.is_black_body)
            {
                const uint32_t term = mat.is_emissive ? 1u : 0u;
                const uint32_t term_spd = mat.is_emissive ? ((uint32_t)mat.emission_spd & 0xFFFFu) : 0u;
                hdr[0] = (uint64_t)shaded | ((uint64_t)term << 16) | ((uint64_t)term_spd << 32) | ((uint64_t)(masks & 0xFFFFu) << 48) | ((uint64_t)(masks >> 16) << 24);
                alive = false;
            }
            else
            {
                at_vertex = true;
                new_vertex = true;
                light = 0;
                n_shaded += 1;
            }
        }
        else if (job == JOB_SHADOW)
        {
            const uint32_t l = light;
            if (new_vertex)
            {
                path_open_vertex(tp, wp, records, shaded, hdr, blk, tbl, spare, spare_tbl);
                new_vertex = false;
            }
            uint64_t *vrec = records + (uint64_t)blk * tp.block_words + (uint64_t)(shaded & (REC_BLOCK_VERTICES - 1u)) * tp.vertex_words;
            uint64_t *lrec = vrec + REC_VERTEX_WORDS + (uint64_t)l * REC_LIGHT_WORDS;
            uint32_t lflags = 0;
            if (!occluded)
            {
                /* incoming = normalise(light_position - position), :320: the very expression (same operands, same operations) that gave
                 * the shadow ray its direction, :241 -- so it is jd, bit for bit, and the light's position need not stay live */
                EvalCoef e = eval_coefficients(sc, sv, ip, jd);
                lflags = e.flags | FLAG_VISIBLE;
                if (l == 0 && shaded < 8u) masks |= 0x10000u << shaded;
                lrec[1] = (uint64_t)__double_as_longlong(light_c);
                store_coef(lrec + 2, e);
            }
            uint32_t em_spd = (uint32_t)sv.mats[sv.light_mat[l]].emission_spd & 0xFFFFu;
            lrec[0] = (uint64_t)em_spd | ((uint64_t)lflags << 16);
            light = l + 1;
        }
        if (alive && at_vertex && light >= sv.n_lights)
        {
            /* every light is done (or there is none): the sampled continuation, :464-472 */
            if (new_vertex)
            {
                path_open_vertex(tp, wp, records, shaded, hdr, blk, tbl, spare, spare_tbl);
                new_vertex = false;
            }
            const DevMaterial &mat = sv.mats[ip.surface_mat];
            uint64_t *vrec = records + (uint64_t)blk * tp.block_words + (uint64_t)(shaded & (REC_BLOCK_VERTICES - 1u)) * tp.vertex_words;
            V3 in;
            double dir_pdf;
            sample_direction(sc, sv, ip, rs, n_draws, in, dir_pdf);
            EvalCoef e = eval_coefficients(sc, sv, ip, in);
            vrec[0] = mat.bdsf_packed;
            vrec[1] = (uint64_t)mat.num_bdsfs | ((uint64_t)(e.flags | mat.vertex_flags) << 8) | ((uint64_t)((uint32_t)mat.diffuse_spd & 0xFFFFu) << 16) |
                      ((uint64_t)((uint32_t)mat.glossy_spd & 0xFFFFu) << 32) | ((uint64_t)((uint32_t)mat.mirror_spd & 0xFFFFu) << 48);
            vrec[2] = record_media_word(sc, sv, ip);
            vrec[3] = (uint64_t)__double_as_longlong(ip.on_dot);
            vrec[4] = (uint64_t)__double_as_longlong(dir_pdf);
            store_coef(vrec + 5, e);
            if ((mat.vertex_flags & FLAG_PLASTIC) && shaded < 16u) masks |= 1u << shaded;
            shaded += 1;
            ip.out = in; /* the path is on a ray again: from ip.position along `in` */
            depth += 1;
            at_vertex = false;
            if (depth >= tp.max_depth)
            {
                hdr[0] = (uint64_t)shaded | ((uint64_t)(masks & 0xFFFFu) << 48) | ((uint64_t)(masks >> 16) << 24);
                alive = false;
            }
        }
    }

    uint64_t vals[6] = {0, n_scans, n_shaded, n_shadow, n_draws, wp.taken};
    for (int k = 1; k < 6; k += 1)
    {
        uint64_t v = vals[k];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0 && v) atomicAdd(&counters[k], (unsigned long long)v);
    }
}
