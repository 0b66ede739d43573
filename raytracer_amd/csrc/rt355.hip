// rt355.hip -- kernels and C ABI of the MI355X renderer (gfx950 only).
//
// Frame pipeline (all on one HIP stream; kernels that serve several lights take the light's slot
// from blockIdx.y):
//   scenes of a few objects, and mode BRUTE:
//   k_render_direct   the whole pixel in one thread, every primitive tested in fp64
//   everything else:
//   k_gen_primary     camera rays (CalcView, main.cpp:269-276) -> fp32 directions
//   k_table_*         per bundle: fp32 bounding cones / circles and edge rows of the primitives, sorted by
//                     direction as seen from the bundle, chunk and superchunk bounds (kept while the
//                     bundle does not change)
//   k_search_packet / k_search   THE SEARCH: rays of a bundle against the table, fp32, conservative
//                     (no false negatives) -> per-ray candidate lists
//   k_resolve_primary fp64 reference formulas on the candidates -> closest hit, texture lookup,
//                     hit record, and the shadow rays of the first group of lights
//   k_scan_bins, k_bin_scatter   counting sort of each light's shadow rays by direction
//   k_shade_group     per hit and light of the group: fp64 closest occluder on the candidates,
//                     diffuse / specular / border / CalcShadow, running mean; then the next
//                     group's shadow rays or the final RGB8 store
//   k_average         alias > 0: mean of a pixel's sub-samples
//
// A "bundle" is a set of rays with a common origin: the camera rays (origin = eye) or the
// shadow rays of one point/spot/area light traced backwards (origin = light position).  For
// a common origin, "ray hits sphere" is "direction inside a cone" and "ray hits triangle"
// is "direction inside a pyramid", so the all-pairs search costs 3 FMA + 1 compare per pair.
// The search only proposes; every accepted intersection, every t and every colour comes
// from rt_exact.h in double, so frames equal the CPU reference byte for byte.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/rt355.h"
#include "rt_exact.h"

#define RT_ABI_VERSION 7
#define SEARCH_THREADS 256
#ifndef RAYS_PER_THREAD
#define RAYS_PER_THREAD 4   // rays per lane of the search kernel (2, 4 or 8)
#endif
#define TILE_W 32                       // tile width in pixels; one workgroup searches one tile
#define TILE_H (8 * RAYS_PER_THREAD)    // tile height in pixels
#define RAYS_PER_TILE (SEARCH_THREADS * RAYS_PER_THREAD)
// The packet search is compiled for 4 rays per lane (256-ray waves: fewer table walks per ray) and for 1
// (64-ray waves: narrower beams and four times as many waves to share a region where beams graze a
// surface and meet thousands of primitives).  Which one is faster depends on the scene and the bundle;
// every scene handle times both on its first frames and keeps the faster one per kind of launch.
#define PACKET_RAYS_MAX 4
#ifndef PACKET_WAVES
#define PACKET_WAVES 4 // waves per workgroup of the packet search
#endif
#define PACKET_THREADS (64 * PACKET_WAVES)
#ifndef PRIM_TILE
#define PRIM_TILE 512       // primitives staged in LDS per step (8 KiB, double buffered)
#endif
#define CAND_CAP 32         // candidate slots per ray; overflowing rays fall back to brute force
// a candidate is the primitive index plus the kind of exact test it needs (the search copies the
// kind bits from .w of the primitive's first L2 row)
// A ray with more than CAND_CAP candidates (a ray skimming a finely tessellated surface) spills the rest into
// one pool per bundle as a chain of its own: an entry is (id, index of the ray's previous entry), the ray's
// newest entry is in head[rid], and the exact tests of such a ray follow its chain.  `cnt` is the ray's
// candidate count before this one: at CAND_CAP the chain starts.  Only if the pool itself is full does the ray
// fall back to the ordered fp64 walk over every primitive.
#define POOL_CAP (1u << 22)
#define POOL_END 0xffffffffu
// A bundle's pool is POOL_CAP entries followed by head[n_rays] (so that a pool is one pointer in the kernels).
#define POOL_STRIDE(n_rays) ((size_t)POOL_CAP + ((size_t)(n_rays) + 1) / 2) /* in entries of 8 bytes */
struct Spill { uint2 *pool; uint32_t *count; };
__device__ __forceinline__ uint32_t *spill_head(const Spill &sp) { return (uint32_t *)(sp.pool + POOL_CAP); }
__device__ __forceinline__ void spill_candidate(const Spill &sp, uint32_t rid, uint32_t id, uint32_t cnt) {
    const uint32_t at = atomicAdd(sp.count, 1u);
    if (at < POOL_CAP) {
        uint32_t *head = spill_head(sp);
        sp.pool[at] = make_uint2(id, cnt == CAND_CAP ? POOL_END : head[rid]);
        head[rid] = at;
    }
}
// Candidates appended by SEVERAL waves to one ray's list (the packet search hands the rest of a wave's walk to many
// waves once the wave has met HEAVY_AFTER primitives, see k_search_packet): the slot comes from the ray's count in
// memory, and a spilled entry takes the ray's previous chain head in exchange for itself (the head of a ray without
// a chain has been set to POOL_END by the wave that handed over).
__device__ __forceinline__ void append_id_shared(uint32_t id, uint32_t rid, uint32_t n_rays, uint32_t *cand,
                                                 uint32_t *cand_cnt, const Spill &sp) {
    const uint32_t slot = atomicAdd(&cand_cnt[rid], 1u);
    if (slot < CAND_CAP) { cand[(size_t)slot * n_rays + rid] = id; return; }
    const uint32_t at = atomicAdd(sp.count, 1u);
    if (at < POOL_CAP) {
        const uint32_t prev = atomicExch(&spill_head(sp)[rid], at);
        sp.pool[at] = make_uint2(id, prev);
    }
}
// The waves of a packet search that handed the rest of their walk over: (workgroup << 2 | wave, ordinal of the first
// surviving chunk left), per workspace slot, and how many there are.
#ifndef HEAVY_CAP
#define HEAVY_CAP 32768u    // entries per slot; a wave that finds the list full walks on by itself
#endif
#define HEAVY_MIN_PRIMS 300000 // scenes below this are searched in one launch (the second costs 0.02 ms and gains nothing there)
#ifndef HEAVY_WORKERS
#define HEAVY_WORKERS 16u   // waves that share what one such wave left
// A launch lasts as long as its longest wave, and the fewer waves it has the more that wave weighs: a call that renders
// a share of a frame (one of several GPUs' tiles) hands over earlier, and does so in smaller scenes too.  Measured on
// tile shares, ms per frame with / without: 100 000 triangles at 3840x2160, half 0.71 / 0.79, quarter 0.47 / 0.55,
// eighth 0.38 / 0.47 (the whole frame: no difference); a million triangles at 7680x4320, eighth 1.08 / 1.14 (quarter
// and half: no gain, so the limit is six million rays); 10 000 spheres: no gain at any size.
#define HEAVY_SMALL_RAYS (6u << 20)
#define HEAVY_SMALL_PRIMS 32768
#define HEAVY_SMALL_AFTER 64u
#endif
#ifndef HEAVY_AFTER
#define HEAVY_AFTER 256u    // primitives met (64 pair tests each) after which a wave hands over
#endif
struct Heavy { uint2 *list; uint32_t *count; uint32_t after, workers; }; // after: HEAVY_AFTER, workers: HEAVY_WORKERS (a multiple of PACKET_WAVES), or what the launch's size asks for
#define CAND_SPHERE 0x40000000u
#define CAND_TRI 0x80000000u
#define CAND_ENVIRONMENT (CAND_SPHERE | CAND_TRI) // both bits: an Environment (its record is a sphere's)
#define REC_QUADRIC (-2) // neither bit: a plane or a quadric, told apart by the second int of the record (-1 | -2)
#define CAND_INDEX 0x3fffffffu
// A ray that starts far from the scene (a mirror ray's hit on an unbounded quadric or plane can lie 1e19 away)
// is not searched: Sphere::CalcIntersect's B * B - C cancels to noise there -- from a distance D it cannot tell
// a line that touches the sphere from one that passes ~2e-8 D beside it -- so the reference "hits" spheres no
// geometric bound predicts, and beyond 1e15 fp32 squares overflow besides.  Such a ray is tested against
// everything, as the reference tests it.  The limit is FAR_SCALE times the size of the scene (bounded
// primitives), where that noise reaches a thousandth of the scene's size; capped at FAR_ORIGIN.
#define FAR_ORIGIN 1e15
#define FAR_SCALE 1e5
#define CAND_UNSEARCHED 0xffffffffu // candidate count of a ray the search left to the fp64 walk over everything
#ifndef CAND_BATCH
#define CAND_BATCH 2        // candidates whose ids and records are loaded together by the exact tests
#endif
#ifndef SHADE_SKIP
#define SHADE_SKIP 0        // 1: the shading kernel leaves out the highlight and the shadow mix where they change nothing (measured: no gain)
#endif
#ifndef SHADE_PREFETCH
#define SHADE_PREFETCH 2    // the shading kernel asks for the next light's candidate count (1), first ids (2) and their records (3) ahead of time
#endif
#define SEARCH_MIN_PRIMS 64 // below this the fp64 brute-force walk is cheaper than a search

// ---- rid <-> pixel -------------------------------------------------------------------------------
// Rays are numbered tile-major ("rid"): rid = tile*RAYS_PER_TILE + r*256 + t, where search thread
// t of the tile's workgroup owns rays r = 0..RAYS_PER_THREAD-1.  Each lane owns a quad 2 pixels
// wide and RAYS_PER_THREAD/2 high, each wave an 8x8 block of quads, and the four waves a 2x2
// block of those, so the rays of a wave are neighbours on screen and all per-ray buffers are
// read and written fully coalesced in every kernel.
#define QUAD_H (RAYS_PER_THREAD / 2)
__device__ __forceinline__ void rid_to_xy(uint32_t rid, int tiles_x, int *x, int *y) {
    uint32_t tile = rid / RAYS_PER_TILE, local = rid % RAYS_PER_TILE;
    uint32_t r = local >> 8, t = local & 255u, wave = t >> 6, lane = t & 63u;
    int px = 16 * (int)(wave & 1u) + 2 * (int)(lane & 7u) + (int)(r & 1u);
    int py = 8 * QUAD_H * (int)(wave >> 1) + QUAD_H * (int)(lane >> 3) + (int)(r >> 1);
    *x = (int)(tile % (uint32_t)tiles_x) * TILE_W + px;
    *y = (int)(tile / (uint32_t)tiles_x) * TILE_H + py;
}

struct Frame {
    int32_t width, height, row_begin, rows; // rows of this slab
    int32_t tiles_x, tiles_y;
    uint32_t n_rays;                        // padded: tiles * 1024
    // The grid the rays are numbered over: the pixels of the slab, or, with alias > 0, their sub-samples
    // (sub-sample (p, q) of pixel (x, y) is grid point (x * alias + p, y * alias + q)).
    int32_t grid_w, grid_h;
    int32_t alias;
    uint32_t seed;
    // Reflection on searched scenes renders a frame level by level (camera rays, their mirror rays, ...): the
    // shading of level `level` >= 0 goes to Work::lvl_clr and k_fold writes the pixels; -1 = no levels.
    int32_t level;
    // Interleaved row tiles (the multi-GPU partition for frames whose cost is uneven from top to bottom): the slab
    // is cut into tiles of tile_rows rows and this call renders tiles tile_phase, tile_phase + tile_count, ...;
    // `rows` then counts the rows of those tiles only.  tile_count <= 1: the slab as a whole.
    int32_t tile_rows, tile_count, tile_phase;
    // Searched frames keep every ray's colour in Work::rec_sum and k_average writes the pixmap in whole rows
    // (the rays of a wave are a 16 x 16 block of every other pixel: their own byte stores leave every line of the
    // pixmap half written by workgroups that run on other XCDs).
    int32_t deferred;
    int32_t soft; // RT355_SHADING_SOFT: Shader::shading calls CalcSoftShadow (shader.h:91) in place of CalcShadow (:90)
    // The last stage of a searched frame (last light group's shading, the fold of the levels, k_average) can be
    // launched part by part, whole rows of tiles each, so that the readback of a part runs while the next is shaded:
    // first ray of this launch of a per-ray kernel, first local row of this launch of k_average.
    uint32_t rid_base;
    int32_t avg_row0;
};
// row of the frame that local row `y` of the call is (local rows are the rows rendered, in order)
__device__ __forceinline__ int frame_row(const Frame &f, int y) {
    if (f.tile_count <= 1) return y + f.row_begin;
    return f.row_begin + ((y / f.tile_rows) * f.tile_count + f.tile_phase) * f.tile_rows + y % f.tile_rows;
}
// inverse of rid_to_xy
__device__ __forceinline__ uint32_t xy_to_rid(int x, int y, int tiles_x) {
    const uint32_t tile = (uint32_t)(y / TILE_H) * (uint32_t)tiles_x + (uint32_t)(x / TILE_W);
    const uint32_t px = (uint32_t)(x % TILE_W), py = (uint32_t)(y % TILE_H);
    const uint32_t wave = (px >> 4) | ((py / (8 * QUAD_H)) << 1);
    const uint32_t lane = ((px >> 1) & 7u) | (((py % (8 * QUAD_H)) / QUAD_H) << 3);
    const uint32_t r = (px & 1u) | ((py % QUAD_H) << 1);
    return tile * RAYS_PER_TILE + r * 256u + wave * 64u + lane;
}

#define LIGHT_GROUP 8 // lights whose shadow rays and candidates are in flight together
struct Work { // per-ray buffers, all indexed by rid (SoA)
    double *rec_d;      // [10][n]: ph.xyz, nh.xyz, reh.xyz (view direction mirrored at the hit), border weight
    int32_t *rec_obj;   // object index of the hit, -1 miss, -2 outside the frame
    uint32_t *rec_cm0;  // material colour | no_border << 24
    uint32_t *rec_cm1;  // dark colour
    uint32_t *rec_sum;  // clr_sum of Shader::shading, carried from one group of lights to the next
    unsigned long long *counters; // 0 shadow rays, 1 candidates evaluated, 2 overflow rays, 3 packet survivors, 4 hits
    // One slot per bundle in flight: slot g serves light k0 + g of the current group (slot 0 first
    // serves the camera rays).  Every array below is [slots][its stride].
    uint32_t n;           // rays per slot (= Frame::n_rays)
    float4 *ray32;        // [n] fp32 search record of the slot's rays (NaN = inactive)
    uint32_t *cand_cnt;   // [n] candidates found
    uint32_t *cand;       // [CAND_CAP][n]
    Spill spill;          // [POOL_CAP] spilled candidates and [1] their count, per slot
    Heavy heavy;          // [HEAVY_CAP] waves of a packet search that handed their walk over and [1] their count, per slot
    uint32_t *tile_count; // [tile_stride] active rays per tile, then exclusive offsets; entry [tiles] = total
    uint32_t *active_rid; // [list_stride] rids of the slot's active rays, tile-major or direction-sorted
    uint32_t *bin_hist;   // [HIST_STRIDE] packet search: rays per (direction bin, counter shard), then offsets
    uint32_t *bin_id;     // [n] packet search: counter (direction bin and shard) of each ray that is cast
    uint32_t *bin_rank;   // [n] packet search: rank of the ray among those of its counter
    uint32_t *far_list;   // [n] the hits of the level being shaded that lie far outside the scene (see FAR_SCALE), and
    uint32_t *far_n;      // [1] their number
    uint8_t *face;        // [n] packet search: bit g = the hit casts a shadow ray to light g of the group whose rays are being sorted
    uint32_t tile_stride, list_stride;
    // reflection on searched scenes (allocated when such a frame is rendered), all [..][n]
    uint32_t *lvl_clr;    // [4] colour Shader::shading gave the ray of each level
    double *lvl_ks;       // [4] ks of the object the ray of each level hit (objs[0]'s on a miss, main.cpp:94)
    double *ktotal;       // reflection_ktotal after the deepest level so far
    uint32_t *deepest;    // deepest level the chain of this sub-sample reached
    double *ray_d;        // [6] the ray of the level being traced: pe.xyz, npe.xyz
    // The rays of a level after the first are numbered densely: dense index d < lvl_live[level] stands for ray
    // dense_rid[d] of the frame, and every per-ray array of the level (hit record, fp32 records of the shadow rays,
    // their candidates, bins) is indexed by d, so that the level's kernels run over its live chains only; the arrays
    // that outlive a level (ray_d, ktotal, deepest, lvl_clr, lvl_ks) stay indexed by the frame's rid.  The mirror rays'
    // own candidates are kept under their place in the level's compacted list; k_resolve_primary then numbers every
    // LEVEL_GROUP consecutive places (a workgroup's) anew, the rays that hit something from the first place up and the others from the last place
    // down (mirror rays that miss lie scattered among those that hit, and the shading kernel, which has nothing to do
    // for them, would run with the lanes of every wave half empty: now at most one wave of a group is mixed).
    uint32_t *lvl_live;   // [4] live chains of each level
    uint32_t *dense_rid;  // [n] the frame's rid of dense index d of the level being traced
    // the gather shaders on searched scenes (allocated when such a frame is rendered), see k_gather_init
    unsigned long long *g_live; // [ceil(dirs / 64)][n] bit i: the hit sends a ray in direction i
};

// Kernels that work on the rays of several lights at once take the workspace slot from blockIdx.y
// (or blockIdx.x for the one-workgroup scans) through this map.
struct Slots {
    int32_t count;
    int32_t slot[LIGHT_GROUP];  // workspace slot (rays, lists, candidates)
    int32_t table[LIGHT_GROUP]; // table slot of the bundle
};

// Statistics are summed per wave and added by one lane to one of COUNTER_SHARDS copies of the
// counter block (each on its own 128-byte line); the host adds the shards up.
#define COUNTER_SHARDS 64
#define COUNTER_STRIDE 16 // unsigned long longs per shard
__device__ __forceinline__ void wave_count(unsigned long long *counters, int which, uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63u) == 0 && v)
        atomicAdd(&counters[(blockIdx.x % COUNTER_SHARDS) * COUNTER_STRIDE + which], (unsigned long long)v);
}

// The same for a kernel that counts once, at its end, and whose every thread gets there: the waves' sums meet in
// LDS and one atomic per counter and WORKGROUP goes out.  (A device-wide atomic is a transaction with memory of its
// own on this chip -- the XCDs' L2s share no lines -- and the counters see it as 32 bytes written: three per wave
// were half as many bytes again as the pixels of the default frame.)
template <int N>
__device__ __forceinline__ void block_count(unsigned long long *counters, const int (&which)[N], const uint32_t (&value)[N]) {
    __shared__ uint32_t tally[N];
    if (threadIdx.x < N) tally[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        uint32_t v = value[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&tally[k], v);
    }
    __syncthreads();
    if (threadIdx.x < N && tally[threadIdx.x])
        atomicAdd(&counters[(blockIdx.x % COUNTER_SHARDS) * COUNTER_STRIDE + which[threadIdx.x]], (unsigned long long)tally[threadIdx.x]);
}

__device__ __forceinline__ float4 nan4() {
    float q = __builtin_nanf("");
    return make_float4(q, q, q, q);
}

// ---- bundle tables ---------------------------------------------------------------------------------
// what the search stores for a candidate: the primitive index plus the kind of exact test it needs
__device__ __forceinline__ uint32_t cand_id(const SceneView &s, int prim) {
    if (prim < 0) return 0u;
    if (s.prim_tri[prim] >= 0) return (uint32_t)prim | CAND_TRI;
    const int type = s.objects[s.prim_obj[prim]].type;
    return (uint32_t)prim | (type == RT_OBJ_SPHERE ? CAND_SPHERE : (type == RT_OBJ_ENVIRONMENT ? CAND_ENVIRONMENT : 0u));
}
// Triangle::CalcIntersect takes u and v from ONE component of the cross products, the first of AA = (p1 - p0) x
// (p2 - p0) that is not zero (anyobject.h:512-513).  Where that component is rounding noise beside the others -- a
// slanted plane that contains the x direction, say -- u and v are noise too, and the reference accepts or rejects
// points anywhere on the triangle's plane.  No bound of the triangle then holds its hits: such a triangle is a
// candidate of every ray, like a plane, in every table.  The limit: with coordinates of size P and a triangle of
// size S the computed u is off by about 8 * 2^-53 * P * S / |component|, so the reference takes points up to that
// times S outside an edge; at a component of 1e-7 of the largest (|AA| ~ S^2) that is 1e-8 * P, a tenth of what the
// tightest margin of the searches allows at that size (1e-6 rad in the bundle tables, 2^-18 relative in the scene
// table), and every such triangle costs every ray an fp64 test -- hence not larger.
__host__ __device__ inline bool triangle_is_unbounded(const double *p0, const double *p1, const double *p2) {
#ifdef RT355_IGNORE_UNBOUNDED_TRIANGLES // (to see the tests that need this fail)
    return false;
#endif
    const double ax = p1[0] - p0[0], ay = p1[1] - p0[1], az = p1[2] - p0[2];
    const double bx = p2[0] - p0[0], by = p2[1] - p0[1], bz = p2[2] - p0[2];
    const double cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx; // cross(), rt_exact.h
    const double sel = (cx != 0) ? cx : ((cy != 0) ? cy : cz);
    const double big = fmax(fabs(cx), fmax(fabs(cy), fabs(cz)));
    return fabs(sel) < 1e-7 * big;
}
// L1[i] = (ox, oy, oz, -thr): unit axis and negated cosine threshold of a cone, seen from the
// bundle origin, that contains primitive i.  A ray with unit direction d can only hit i if
// d . o - thr >= 0.  -thr = +2 marks "always a candidate" (planes, origin inside the sphere),
// -thr = -2 "never" (triangles facing the wrong way for the whole bundle, padding).
// L2[3i..3i+2] = inward unit normals of the three planes through the origin and a triangle
// edge; d can only hit the triangle if d . n >= -eps for all three.  Zero rows always pass.
struct TableRow { float4 l1, a, b, c; float sin_t; };
__device__ __forceinline__ TableRow bundle_origin_row(const SceneView &s, d3 O, int is_eye, int i) { // i = the primitive
    float4 never = make_float4(0.f, 0.f, 0.f, -2.f), always = make_float4(0.f, 0.f, 0.f, 2.f);
    float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 l1 = never, a = z4, b = z4, c = z4;
    // L1s[i] = sine of the cone's (inflated) half angle, used by the packet search's beam test:
    // 0 for "always", -inf for "never" (which turns cos(theta + beta) into +inf or NaN: no survivor)
    float sin_t = -__builtin_inff();
    {
        int obj = s.prim_obj[i], tri = s.prim_tri[i];
        d3 ctr; double rad; bool sphere_like = false;
        if (tri < 0) {
            const DObject &o = s.objects[obj];
            if (o.type == RT_OBJ_PLANE || o.type == RT_OBJ_ENVIRONMENT || o.type == RT_OBJ_QUADRIC) { l1 = always; sin_t = 0.f; } // (an Environment surrounds the scene; a quadric is unbounded)
            else { ctr = ld3(o.p); rad = fabs(o.radius); sphere_like = true; }
        } else {
            const DTriangle &t = s.tris[tri];
            d3 p0 = ld3(t.p0), p1 = ld3(t.p1), p2 = ld3(t.p2), nrm = ld3(t.normal), pc = ld3(t.p_center);
            bool cull;
            if (is_eye) {
                // Triangle::CalcIntersect with pe == O: a negative plane distance gives t < 0
                // for every direction that passes the back-face test (anyobject.h:479-501).
                cull = dot(nrm, vsub(O, pc)) < 0;
            } else {
                // Shadow rays travel toward O.  A triangle whose front side faces O can only be
                // hit beyond O, which never shadows (shader.h:413).
                d3 w = vsub(O, pc);
                cull = dot(nrm, w) > 1e-9 * (1.0 + vlen(w));
            }
            if (!cull && triangle_is_unbounded(t.p0, t.p1, t.p2)) {
                l1 = always; sin_t = 0.f;
            } else if (!cull) {
                ctr = pc;
                double r0 = vlen(vsub(p0, pc)), r1 = vlen(vsub(p1, pc)), r2 = vlen(vsub(p2, pc));
                rad = fmax(r0, fmax(r1, r2));
                sphere_like = true;
                d3 q0 = vsub(p0, O), q1 = vsub(p1, O), q2 = vsub(p2, O);
                d3 na = cross(q1, q2), nb = cross(q2, q0), nc = cross(q0, q1);
                double vol = dot(na, q0);
                double la = vlen(na), lb = vlen(nb), lc = vlen(nc);
                double scale = vlen(q0) * vlen(q1) * vlen(q2);
                if (fabs(vol) > 1e-7 * scale && la > 0 && lb > 0 && lc > 0) {
                    double sg = vol > 0 ? 1.0 : -1.0;
                    a = make_float4((float)(na.x * sg / la), (float)(na.y * sg / la), (float)(na.z * sg / la), 0.f);
                    b = make_float4((float)(nb.x * sg / lb), (float)(nb.y * sg / lb), (float)(nb.z * sg / lb), 0.f);
                    c = make_float4((float)(nc.x * sg / lc), (float)(nc.y * sg / lc), (float)(nc.z * sg / lc), 0.f);
                } // else: origin (nearly) in the triangle's plane -- leave the rows zero
            }
        }
        if (sphere_like) {
            d3 oc = vsub(ctr, O);
            double d2 = dot(oc, oc);
            if (d2 <= rad * rad * (1.0 + 1e-5) + 1e-300) {
                l1 = always; sin_t = 0.f;
            } else {
                double dist = sqrt(d2), sn = rad / dist;
                double thr = sqrt(1.0 - sn * sn) - 2e-6; // fp32 evaluation error of d.o is < 4e-7
                l1 = make_float4((float)(oc.x / dist), (float)(oc.y / dist), (float)(oc.z / dist), (float)(-thr));
                sin_t = (float)(sqrt(fmax(0.0, 1.0 - thr * thr)) * (1.0 + 1e-6));
            }
        }
    }
    a.w = __uint_as_float(cand_id(s, i));
    TableRow row;
    row.l1 = l1; row.a = a; row.b = b; row.c = c; row.sin_t = sin_t;
    return row;
}

// Bundle of parallel rays (the shadow rays of a directional light, shader.h:76-77): every ray
// runs along w = -direction.  With an orthonormal basis (e1, e2, w), "ray from ph hits sphere
// (c, r)" needs the 2-D distance between the projections of ph and c to be <= r, and "hits
// triangle" needs the projection of ph inside the projected triangle.
//   L1[i] = (u_c, v_c, R2, R): projected centre, inflated squared radius and radius of primitive
//           i's bounding circle; R2 = +inf "always" (planes), -inf "never" (culled, padding).
//   L2[3i..3i+2] = (A, B, C, 0): inward unit edge functions A*u + B*v + C of the projected
//           triangle, C already raised by the per-primitive rounding bound.  Zero rows pass.
// A ray carries (u_p, v_p, m_sq, m_lin): its projected origin and the two bounds that cover the
// fp32 rounding of its own coordinates (they grow with |u_p| + |v_p|; hit points on a plane can
// lie far from the scene).
struct Bundle {
    int kind;     // 0 = not searched (fp64 brute force), 1 = common origin, 2 = common direction
    d3 o;         // kind 1: the origin
    d3 e1, e2;    // kind 2: basis of the projection plane
    d3 dir;       // kind 2: the common ray direction (unit)
    double rmaxp; // kind 2: upper bound of (bounding radius + per-primitive rounding bound)
    int eye;      // kind 1: the bundle of the camera rays (its origin is every ray's exact origin)
    int binned;   // packet search: the rays of this bundle are also counted into direction bins
    float plane_extent; // kind 2: half-width of the binning grid on the projection plane
};
#define RT_EPS22 2.384185791015625e-07 /* 2^-22 */
struct Group { // lights k0 .. k0 + count - 1 and the bundles of their shadow rays (slot g = light k0 + g)
    int k0, count;
    Bundle b[LIGHT_GROUP];
};

__device__ __forceinline__ TableRow bundle_direction_row(const SceneView &s, const Bundle &bd, int i) {
    const d3 w = bd.dir;
    const float inf = __builtin_inff();
    // .w = the circle's radius for the packet search's beam test: NaN "never", +inf "always"
    float4 never = make_float4(0.f, 0.f, -inf, __builtin_nanf("")), always = make_float4(0.f, 0.f, inf, inf);
    float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 l1 = never, a = z4, b = z4, c = z4;
    {
        int obj = s.prim_obj[i], tri = s.prim_tri[i];
        d3 ctr; double rad = 0; bool round_like = false;
        if (tri < 0) {
            const DObject &o = s.objects[obj];
            if (o.type == RT_OBJ_ENVIRONMENT || o.type == RT_OBJ_QUADRIC) {
                l1 = always;
            } else if (o.type == RT_OBJ_PLANE) {
                // Plane::CalcIntersect misses when ni . npe >= 0 (anyobject.h:80); npe = w up to rounding
                d3 ni = ld3(o.n);
                l1 = (dot(ni, w) > 1e-12 * vlen(ni)) ? never : always;
            } else { ctr = ld3(o.p); rad = fabs(o.radius); round_like = true; }
        } else {
            const DTriangle &t = s.tris[tri];
            d3 p0 = ld3(t.p0), p1 = ld3(t.p1), p2 = ld3(t.p2), nrm = ld3(t.normal), pc = ld3(t.p_center);
            if (!(dot(nrm, w) > 1e-12) && triangle_is_unbounded(t.p0, t.p1, t.p2)) {
                l1 = always;
            } else if (!(dot(nrm, w) > 1e-12)) { // back-face test of anyobject.h:490 is the same for every ray
                ctr = pc;
                rad = fmax(vlen(vsub(p0, pc)), fmax(vlen(vsub(p1, pc)), vlen(vsub(p2, pc))));
                round_like = true;
                double u[3] = {dot(p0, bd.e1), dot(p1, bd.e1), dot(p2, bd.e1)};
                double v[3] = {dot(p0, bd.e2), dot(p1, bd.e2), dot(p2, bd.e2)};
                double area2 = (u[1] - u[0]) * (v[2] - v[0]) - (v[1] - v[0]) * (u[2] - u[0]);
                double big = fabs(u[0]) + fabs(v[0]) + fabs(u[1]) + fabs(v[1]) + fabs(u[2]) + fabs(v[2]);
                if (fabs(area2) > 1e-12 * (rad * rad + 1e-300)) {
                    float4 rows[3];
                    for (int k = 0; k < 3; k++) {
                        int q = (k + 1) % 3, r = (k + 2) % 3;
                        double A = -(v[q] - v[k]), B = (u[q] - u[k]);
                        double len = sqrt(A * A + B * B);
                        A /= len; B /= len;
                        double side = A * (u[r] - u[k]) + B * (v[r] - v[k]);
                        if (side < 0) { A = -A; B = -B; }
                        double C = -(A * u[k] + B * v[k]) + 4.0 * RT_EPS22 * big + 1e-6 * rad + 1e-30;
                        rows[k] = make_float4((float)A, (float)B, (float)C, 0.f);
                    }
                    a = rows[0]; b = rows[1]; c = rows[2];
                }
            }
        }
        if (round_like) {
            double uc = dot(ctr, bd.e1), vc = dot(ctr, bd.e2);
            double es = RT_EPS22 * (fabs(uc) + fabs(vc) + rad);
            double R = rad + es;
            l1 = make_float4((float)uc, (float)vc, (float)(R * R * (1.0 + 1e-6)), (float)(R * (1.0 + 1e-6)));
        }
    }
    a.w = __uint_as_float(cand_id(s, i));
    TableRow row;
    row.l1 = l1; row.a = a; row.b = b; row.c = c; row.sin_t = 0.f;
    return row;
}

// ---- bounds of bounds ----------------------------------------------------------------------------------
// One record bounding the 64 records held by the lanes of a wave, in the records' own format, so
// that the beam test of the packet search applies to every level alike.
//   kind 1: cone (unit axis, -cos, sin of the half angle) containing every member cone: axis = mean
//           of the member axes, half angle = max over members of (angle to the axis + own half angle).
//           Angles come from chord lengths (accurate for small angles) in double, plus 1e-5 rad.
//   kind 2: circle (centre, R^2, R) containing every member circle.
// "never" members are skipped, one "always" member makes the bound "always".
struct Bound { float4 p; float s; };
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ Bound bound_of_wave(int kind, float4 p, float st) {
    Bound out;
    const float inf = __builtin_inff();
    if (kind == 1) {
        const bool always = p.w >= 1.5f, never = p.w <= -1.5f, valid = !always && !never;
        const int n_valid = __popcll(__ballot(valid));
        out.p = make_float4(0.f, 0.f, 0.f, -2.f); out.s = -inf; // never
        if (__ballot(always)) { out.p.w = 2.f; out.s = 0.f; return out; }
        if (!n_valid) return out;
        double ax = wave_sum_d(valid ? (double)p.x : 0.0), ay = wave_sum_d(valid ? (double)p.y : 0.0),
               az = wave_sum_d(valid ? (double)p.z : 0.0);
        const double len = sqrt(ax * ax + ay * ay + az * az);
        double total = 0.0;
        if (len > 1e-3 * n_valid) {
            ax /= len; ay /= len; az /= len;
            if (valid) {
                const double dx = ax - p.x, dy = ay - p.y, dz = az - p.z;
                const double chord = sqrt(dx * dx + dy * dy + dz * dz);
                const double own = acos(fmin(1.0, fmax(-1.0, -(double)p.w)));
                total = 2.0 * asin(fmin(1.0, 0.5 * chord)) + own;
            }
            total = wave_max_d(total) + 1e-5;
        } else {
            total = 4.0; // the axes cancel out: no useful cone
        }
        if (total >= 1.5) { out.p.w = 2.f; out.s = 0.f; return out; }
        out.p = make_float4((float)ax, (float)ay, (float)az, (float)(-(cos(total) - 1e-6)));
        out.s = (float)(sin(total) * (1.0 + 1e-6) + 1e-6);
        return out;
    }
    const bool always = p.w == inf, never = !(p.w == p.w), valid = !always && !never;
    const int n_valid = __popcll(__ballot(valid));
    out.p = make_float4(0.f, 0.f, -inf, __builtin_nanf("")); out.s = 0.f; // never
    if (__ballot(always)) { out.p.z = inf; out.p.w = inf; return out; }
    if (!n_valid) return out;
    const double cx = wave_sum_d(valid ? (double)p.x : 0.0) / n_valid, cy = wave_sum_d(valid ? (double)p.y : 0.0) / n_valid;
    double reach = 0.0;
    if (valid) {
        const double dx = p.x - cx, dy = p.y - cy;
        reach = sqrt(dx * dx + dy * dy) + (double)p.w;
    }
    const double R = wave_max_d(reach) * (1.0 + 1e-6) + 4.0 * RT_EPS22 * (fabs(cx) + fabs(cy)) + 1e-30;
    if (!(R < 1e30)) { out.p.z = inf; out.p.w = inf; return out; }
    out.p = make_float4((float)cx, (float)cy, (float)(R * R * (1.0 + 1e-6)), (float)(R * (1.0 + 1e-6)));
    return out;
}

// fp32 search record of a shadow ray for the given bundle
__device__ __forceinline__ float4 bundle_ray(const Bundle &bd, d3 ph, d3 nhl) {
    if (bd.kind == 1) return make_float4((float)nhl.x, (float)nhl.y, (float)nhl.z, 0.f);
    double pu = dot(ph, bd.e1), pv = dot(ph, bd.e2);
    // ... and the reference's own rounding: from an origin far along the common direction (a mirror ray's hit on an
    // unbounded quadric or plane can lie 1e19 away) Sphere::CalcIntersect's B * B - C cancels to noise of
    // ~2e-8 |pe - c| in the distance it can tell from the radius, so spheres that far off the line may "hit"
    // (|ph|_1 bounds the distance along the direction; the direction itself would cost the callers 48 more scalars)
    double er = RT_EPS22 * (fabs(pu) + fabs(pv)) + 1.2e-7 * (fabs(ph.x) + fabs(ph.y) + fabs(ph.z));
    double msq = (2.0 * bd.rmaxp * er + er * er) * (1.0 + 1e-5);
    return make_float4((float)pu, (float)pv, (float)(msq * (1.0 + 1e-6)), (float)(4.0 * er));
}

// ---- camera rays -------------------------------------------------------------------------------------
// CalcPixel's ray for grid point (gx, gy): the pixel corner (main.cpp:327-329), or a jittered sub-sample
// (main.cpp:292-303)
__device__ __forceinline__ Ray grid_ray(const DCamera &cam, const Frame &f, int gx, int gy) {
    if (f.alias <= 0)
        return calc_view(cam, (double)gx / (double)f.width, (double)frame_row(f, gy) / (double)f.height);
    const int x = gx / f.alias, p = gx % f.alias, y = frame_row(f, gy / f.alias), q = gy % f.alias;
    const double pointX = (rt355x_jitter(f.seed, x, y, q * f.alias + p, 0) + p) / f.alias + x;
    const double pointY = (rt355x_jitter(f.seed, x, y, q * f.alias + p, 1) + q) / f.alias + y;
    return calc_view(cam, pointX / (double)f.width, pointY / (double)f.height);
}
__global__ void k_gen_primary(DCamera cam, Frame f, float4 *__restrict__ ray32) {
    uint32_t rid = blockIdx.x * blockDim.x + threadIdx.x;
    if (rid >= f.n_rays) return;
    int x, y;
    rid_to_xy(rid, f.tiles_x, &x, &y);
    float4 out = nan4();
    if (x < f.grid_w && y < f.grid_h) {
        // The search record is the fp32 rounding of the unit direction and only has to be within ~1e-9 of it (the
        // searches' margins start at 2^-25): the point on the image plane as CalcView computes it, then ONE
        // normalisation by a Newton-refined reciprocal square root instead of CalcView's two correctly rounded ones
        // (a square root and three divisions each), which the fp64 kernels make when they shade the ray.
        double x_per, y_per;
        if (f.alias <= 0) {
            x_per = (double)x / (double)f.width; y_per = (double)frame_row(f, y) / (double)f.height;
        } else {
            const int px = x / f.alias, p = x % f.alias, py = frame_row(f, y / f.alias), q = y % f.alias;
            x_per = ((rt355x_jitter(f.seed, px, py, q * f.alias + p, 0) + p) / f.alias + px) / (double)f.width;
            y_per = ((rt355x_jitter(f.seed, px, py, q * f.alias + p, 1) + q) / f.alias + py) / (double)f.height;
        }
        const d3 pp = vadd(vadd(ld3(cam.p00), vmul(ld3(cam.n0), x_per * cam.s_x)), vmul(ld3(cam.n1), y_per * cam.s_y));
        const d3 v = vsub(pp, ld3(cam.p_eye));
        const double len2 = v.x * v.x + v.y * v.y + v.z * v.z;
        if (len2 > 1e-200 && len2 < 1e200) {
            double r = __builtin_amdgcn_rsq(len2);
            r = fma(r * 0.5, fma(-len2 * r, r, 1.0), r);
            out = make_float4((float)(v.x * r), (float)(v.y * r), (float)(v.z * r), 0.f);
        } else { // (a degenerate camera: the reference's own operations)
            const Ray e = grid_ray(cam, f, x, y);
            out = make_float4((float)e.npe.x, (float)e.npe.y, (float)e.npe.z, 0.f);
        }
    }
    ray32[rid] = out;
}

// ---- compaction of a bundle's active rays -------------------------------------------------------------
// Shadow rays exist only for hits that face the light, so a bundle's active rays can be a small
// part of the frame (sky pixels, back-facing hits).  Three small kernels keep the search dense:
// count the active rays of each tile, scan the counts, scatter the active rids in tile-major
// order.  The search then runs over the compacted list; results stay indexed by rid.
__device__ __forceinline__ bool ray_active(const float4 *__restrict__ ray32, uint32_t rid) {
    float x = ray32[rid].x;
    return x == x; // inactive rays hold NaN
}

__global__ __launch_bounds__(SEARCH_THREADS) void k_count_active(const float4 *__restrict__ ray32_all,
                                                                  uint32_t *__restrict__ tile_count_all,
                                                                  uint32_t n_rays, uint32_t tile_stride, Slots sl) {
    const size_t g = (size_t)sl.slot[blockIdx.y];
    const float4 *__restrict__ ray32 = ray32_all + g * n_rays;
    uint32_t *__restrict__ tile_count = tile_count_all + g * tile_stride;
    __shared__ uint32_t wsum[SEARCH_THREADS / 64];
    const uint32_t t = threadIdx.x, base = blockIdx.x * RAYS_PER_TILE + t;
    uint32_t n = 0;
#pragma unroll
    for (int r = 0; r < RAYS_PER_THREAD; r++) n += ray_active(ray32, base + r * SEARCH_THREADS) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((t & 63u) == 0) wsum[t >> 6] = n;
    __syncthreads();
    if (t == 0) tile_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of n_tiles counts in place (single workgroup); tile_count[n_tiles] = total
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t *__restrict__ tile_count_all, int n_tiles,
                                                     uint32_t tile_stride, Slots sl) {
    uint32_t *__restrict__ tile_count = tile_count_all + (size_t)sl.slot[blockIdx.x] * tile_stride;
    __shared__ uint32_t part[1024];
    const int t = threadIdx.x;
    const int per = (n_tiles + 1023) / 1024;
    const int lo = t * per, hi = min(lo + per, n_tiles);
    uint32_t sum = 0;
    for (int i = lo; i < hi; i++) sum += tile_count[i];
    part[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan
        uint32_t v = (t >= off) ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - sum;
    for (int i = lo; i < hi; i++) { uint32_t c = tile_count[i]; tile_count[i] = run; run += c; }
    if (t == 1023) tile_count[n_tiles] = part[1023];
}

__global__ __launch_bounds__(SEARCH_THREADS) void k_scatter_active(const float4 *__restrict__ ray32_all,
                                                                    const uint32_t *__restrict__ tile_count_all,
                                                                    uint32_t *__restrict__ active_rid_all,
                                                                    uint32_t n_rays, uint32_t tile_stride,
                                                                    uint32_t list_stride, Slots sl) {
    const size_t g = (size_t)sl.slot[blockIdx.y];
    const float4 *__restrict__ ray32 = ray32_all + g * n_rays;
    const uint32_t *__restrict__ tile_count = tile_count_all + g * tile_stride;
    uint32_t *__restrict__ active_rid = active_rid_all + g * list_stride;
    __shared__ uint32_t wsum[RAYS_PER_THREAD][SEARCH_THREADS / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t base = blockIdx.x * RAYS_PER_TILE + t;
    bool act[RAYS_PER_THREAD];
    uint32_t below[RAYS_PER_THREAD];
#pragma unroll
    for (int r = 0; r < RAYS_PER_THREAD; r++) {
        act[r] = ray_active(ray32, base + r * SEARCH_THREADS);
        unsigned long long m = __ballot(act[r]);
        below[r] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[r][wave] = __popcll(m);
    }
    __syncthreads();
    // rank of local index r*256 + t among the tile's active rays
    uint32_t run = tile_count[blockIdx.x];
#pragma unroll
    for (int r = 0; r < RAYS_PER_THREAD; r++) {
#pragma unroll
        for (int w = 0; w < SEARCH_THREADS / 64; w++) {
            if ((uint32_t)w == wave && act[r]) active_rid[run + below[r]] = base + r * SEARCH_THREADS;
            run += wsum[r][w];
        }
    }
}

// ---- binning of a shadow bundle's rays by direction (packet search only) -----------------------------
// Neighbouring pixels can have hit points far apart (depth edges, grazing floors), so the shadow
// rays of a screen tile make a wide beam as seen from the light.  For the packet search the active
// rays are therefore ordered by WHERE THEY POINT instead of where they are on screen: a counting
// sort on a 256x256 grid over the sphere of directions (octahedral map) or over the projection
// plane, cells numbered in Morton order so that consecutive cells are neighbours.  Order inside a
// cell is whatever the atomics give; candidates are per ray, so frames do not depend on it.
#define RT_BINS 16384 /* 128 x 128 cells */
// Every cell has RT_SHARDS counters; a ray uses the one of its 256-ray block (block % RT_SHARDS).
// Neighbouring blocks run at the same time and point the same way, so with one counter per cell
// their atomics would queue up on one address.  Layout: [cell][shard], then the padded total.
#ifndef RT_SHARDS
#define RT_SHARDS 4
#endif
#define HIST_TOTAL (RT_BINS * RT_SHARDS + 1)
#define HIST_STRIDE (RT_BINS * RT_SHARDS + 2)
__device__ __forceinline__ uint32_t hist_index(uint32_t bin, uint32_t rid) {
    return bin * RT_SHARDS + (rid / 256u) % RT_SHARDS;
}
// index of cell (x, y) of a 128x128 grid along the Hilbert curve: consecutive indices are always
// neighbouring cells, so any run of consecutive rays covers one connected patch of directions
__device__ __forceinline__ uint32_t hilbert7(uint32_t x, uint32_t y) {
    uint32_t d = 0;
#pragma unroll
    for (uint32_t s = 64; s > 0; s >>= 1) {
        uint32_t rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
        d += s * s * ((3u * rx) ^ ry);
        if (ry == 0) {
            if (rx == 1) { x = 127u - x; y = 127u - y; }
            uint32_t tmp = x; x = y; y = tmp;
        }
    }
    return d;
}
__device__ __forceinline__ uint32_t ray_bin(int kind, const float4 &q, float plane_extent) {
    float u, v;
    if (kind == 1) { // octahedral map of the unit direction
        float inv = 1.0f / (fabsf(q.x) + fabsf(q.y) + fabsf(q.z));
        float px = q.x * inv, py = q.y * inv;
        if (q.z < 0.f) {
            float ox = (1.f - fabsf(py)) * (px >= 0.f ? 1.f : -1.f);
            float oy = (1.f - fabsf(px)) * (py >= 0.f ? 1.f : -1.f);
            px = ox; py = oy;
        }
        u = px * 0.5f + 0.5f; v = py * 0.5f + 0.5f;
    } else {         // grid over the projection plane; far-away origins share the border cells
        u = q.x / plane_extent * 0.5f + 0.5f; v = q.y / plane_extent * 0.5f + 0.5f;
    }
    int ix = (int)fminf(fmaxf(u * 128.f, 0.f), 127.f), iy = (int)fminf(fmaxf(v * 128.f, 0.f), 127.f);
    return hilbert7((uint32_t)ix, (uint32_t)iy);
}
// Rays of a wave mostly fall into a few cells.  The wave first groups its lanes by cell with
// ballots only (no memory traffic), then the first lane of every group adds the group's size to
// the cell counter in ONE atomic instruction for the whole wave, and each lane's rank inside its
// cell is the counter value returned to the group's first lane plus its rank inside the group.
// Must be called by all lanes of the wave (inactive lanes pass false).
__device__ __forceinline__ uint32_t wave_bin_count(uint32_t *__restrict__ hist, uint32_t bin, bool active,
                                                   uint32_t &rank, uint32_t &leader_lane) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t count = 0;
    leader_lane = lane; rank = 0;
    unsigned long long pending = __ballot(active);
    while (pending) {
        const int leader = __builtin_ctzll(pending);
        const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
        const unsigned long long same = __ballot(active && bin == b) & pending;
        if ((same >> lane) & 1ull) {
            leader_lane = (uint32_t)leader;
            rank = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
            count = (uint32_t)__popcll(same);
        }
        pending &= ~same;
    }
    uint32_t base = 0;
    if (active && lane == leader_lane) base = atomicAdd(&hist[bin], count);
    return base; // valid in the group's first lane; the lane's rank in its cell is that value + rank
}
// position of every ray that is cast in the direction-sorted list: start of its counter's run + rank in it
__global__ __launch_bounds__(256) void k_bin_scatter(const uint8_t *__restrict__ face, const uint32_t *__restrict__ bin_id_all,
                                                     const uint32_t *__restrict__ bin_rank_all,
                                                     const uint32_t *__restrict__ offsets_all,
                                                     uint32_t *__restrict__ active_rid_all, uint32_t n_rays,
                                                     uint32_t list_stride, Slots sl) {
    const size_t g = (size_t)sl.slot[blockIdx.y];
    const uint32_t rid = blockIdx.x * blockDim.x + threadIdx.x; // the ray count is a multiple of the block size
    if (!((face[rid] >> g) & 1u)) return;
    const size_t at = g * n_rays + rid;
    active_rid_all[g * list_stride + offsets_all[g * HIST_STRIDE + bin_id_all[at]] + bin_rank_all[at]] = rid;
}

// Offsets of the direction bins, with every run of 16 consecutive bins (a 4x4 patch of cells on
// the Hilbert curve) padded to a multiple of 256 rays (the padding entries of the sorted list are
// written here), so that no wave of the packet search
// mixes rays of different patches: a wave's beam is never wider than one patch, whatever the ray
// density.  One workgroup per slot; thread c owns patch c (all shards of its cells, contiguous).
// hist[HIST_TOTAL] receives the padded total.
#define RT_PATCH 16
__global__ __launch_bounds__(1024) void k_scan_bins(uint32_t *__restrict__ hist_all,
                                                    uint32_t *__restrict__ active_rid_all, uint32_t list_stride,
                                                    Slots sl) {
    const size_t g = (size_t)sl.slot[blockIdx.x];
    uint32_t *__restrict__ hist = hist_all + g * HIST_STRIDE;
    uint32_t *__restrict__ active_rid = active_rid_all + g * list_stride;
    // Patch p's counters are RT_PATCH * RT_SHARDS contiguous words, and thread p owns them: they stay in its registers
    // from the patch's sum, through the scan of the padded sums over the workgroup (shuffles inside a wave, the 16
    // wave sums through LDS), to the offsets written back in their place.
    constexpr int WORDS = RT_PATCH * RT_SHARDS;
    static_assert(WORDS % 2 == 0 && (HIST_STRIDE % 2) == 0, "a patch is read as 8-byte words");
    __shared__ uint32_t wave_sum[16];
    __shared__ uint32_t pad_from[1024], pad_to[1024];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint2 *__restrict__ mine = (uint2 *)(hist + (size_t)t * WORDS);
    uint2 v[WORDS / 2];
    uint32_t sum_c = 0;
#pragma unroll
    for (int k = 0; k < WORDS / 2; k++) v[k] = mine[k];
#pragma unroll
    for (int k = 0; k < WORDS / 2; k++) sum_c += v[k].x + v[k].y;
    const uint32_t padded = (sum_c + (64u * PACKET_RAYS_MAX - 1u)) / (64u * PACKET_RAYS_MAX) * (64u * PACKET_RAYS_MAX);
    uint32_t incl = padded; // inclusive scan of the padded sums
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(incl, off, 64);
        if (lane >= off) incl += x;
    }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint32_t ws = wave_sum[k]; total += ws; if (k < wave) before += ws; }
    const uint32_t start_c = before + incl - padded; // patch t starts here
    // padding = no ray: up to 255 entries behind every patch, written by the waves together (a wave takes every 16th
    // patch, its lanes neighbouring entries)
    pad_from[t] = start_c + sum_c; pad_to[t] = start_c + padded;
    __syncthreads();
    for (int p = wave; p < 1024; p += 16)
        for (uint32_t i = pad_from[p] + (uint32_t)lane; i < pad_to[p]; i += 64u) active_rid[i] = 0xffffffffu;
    uint32_t run = start_c;
#pragma unroll
    for (int k = 0; k < WORDS / 2; k++) {
        uint32_t c = v[k].x; v[k].x = run; run += c;
        c = v[k].y; v[k].y = run; run += c;
    }
#pragma unroll
    for (int k = 0; k < WORDS / 2; k++) mine[k] = v[k];
    if (t == 1023) hist[HIST_TOTAL] = total;
}

// ---- the tables of a bundle ----------------------------------------------------------------------------
// One set per table slot.  Rows are laid out by WHERE THE PRIMITIVE LIES AS SEEN FROM THE BUNDLE: a
// counting sort on the same 128x128 Hilbert-ordered grid the rays use (octahedral map of the cone
// axis, or the projection plane), cell 0 reserved for the "always" records; primitives that can never
// be hit by the bundle get no row at all.  So the 64 rows of a chunk, and the 64 chunks of a
// superchunk, are neighbours in direction and their common bound is tight, whatever their depth.
// Order inside a cell is whatever the atomics give; nothing downstream depends on row order.
// n_table = 4096 * superchunks >= the padded primitive count; rows past the live ones are "never".
#define TB_CELLS 17408 /* 1 + 16384 cells, rounded up to 1024 * 17 */
struct Tables {
    float4 *L1;  float *L1s;  // [n_table] primitive bounds (L1s: sine of the cone's half angle, kind 1)
    float4 *L2;               // [3 n_table] pyramid / edge rows; row 0 .w = candidate id
    float4 *C1;  float *C1s;  // [n_table / 64] chunk bounds
    float4 *S1;  float *S1s;  // [top_stride] superchunk bounds, padded to whole wave loads with "never"
    uint32_t *hist;           // [TB_CELLS + 1] rows per cell, then their offsets; [TB_CELLS] = live rows
    uint32_t *cell, *rank;    // [n_prims_pad] cell of every primitive (0xffffffff = no row) and its rank there
    int32_t n_table, top_stride, n_prims_pad;
};
__device__ __forceinline__ Tables slot_tables(const Tables &t, size_t g) {
    Tables o = t;
    o.L1 += g * t.n_table; o.L1s += g * t.n_table; o.L2 += 3 * g * t.n_table;
    o.C1 += g * (t.n_table / 64); o.C1s += g * (t.n_table / 64);
    o.S1 += g * t.top_stride; o.S1s += g * t.top_stride;
    o.hist += g * (TB_CELLS + 1); o.cell += g * t.n_prims_pad; o.rank += g * t.n_prims_pad;
    return o;
}
// which table slot each bundle of a launch uses (blockIdx.y -> table slot)
struct TableMap { int32_t slot[LIGHT_GROUP]; };

__device__ __forceinline__ TableRow bundle_row(const SceneView &s, const Bundle &bd, int prim) {
    return (bd.kind == 1) ? bundle_origin_row(s, bd.o, bd.eye, prim) : bundle_direction_row(s, bd, prim);
}
// step 1: cell and rank of every primitive; thread = primitive, bundle = blockIdx.y
__global__ __launch_bounds__(256) void k_table_bin(SceneView s, Group gr, TableMap tm, Tables t_all) {
    const Bundle &bd = gr.b[blockIdx.y];
    if (!bd.kind) return;
    const Tables t = slot_tables(t_all, tm.slot[blockIdx.y]);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cell = 0xffffffffu;
    if (i < s.n_prims) {
        const TableRow row = bundle_row(s, bd, i);
        const bool never = (bd.kind == 1) ? (row.l1.w <= -1.5f) : !(row.l1.w == row.l1.w);
        const bool always = (bd.kind == 1) ? (row.l1.w >= 1.5f) : (row.l1.w == __builtin_inff());
        if (always) cell = 0u;
        else if (!never) cell = 1u + ray_bin(bd.kind, row.l1, bd.plane_extent);
    }
    uint32_t rank, leader;
    const uint32_t before = wave_bin_count(t.hist, cell, cell != 0xffffffffu, rank, leader);
    const uint32_t base = (uint32_t)__shfl((int)before, (int)leader, 64);
    if (i < s.n_prims) { t.cell[i] = cell; t.rank[i] = base + rank; }
}
// step 2: counts -> offsets (one workgroup per bundle)
__global__ __launch_bounds__(1024) void k_table_scan(Group gr, TableMap tm, Tables t_all) {
    if (!gr.b[blockIdx.x].kind) return;
    const Tables t = slot_tables(t_all, tm.slot[blockIdx.x]);
    __shared__ uint32_t part[1024];
    constexpr int PER = TB_CELLS / 1024;
    const int c = threadIdx.x;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { v[k] = t.hist[c * PER + k]; sum += v[k]; }
    part[c] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t x = (c >= off) ? part[c - off] : 0u;
        __syncthreads();
        part[c] += x;
        __syncthreads();
    }
    uint32_t run = part[c] - sum;
#pragma unroll
    for (int k = 0; k < PER; k++) { t.hist[c * PER + k] = run; run += v[k]; }
    if (c == 1023) t.hist[TB_CELLS] = part[1023];
}
// step 3: the rows, each at its cell's offset + its rank; thread = primitive
__global__ __launch_bounds__(256) void k_table_rows(SceneView s, Group gr, TableMap tm, Tables t_all) {
    const Bundle &bd = gr.b[blockIdx.y];
    if (!bd.kind) return;
    const Tables t = slot_tables(t_all, tm.slot[blockIdx.y]);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n_prims) return;
    const uint32_t cell = t.cell[i];
    if (cell == 0xffffffffu) return;
    const TableRow row = bundle_row(s, bd, i);
    const size_t at = (size_t)t.hist[cell] + t.rank[i];
    t.L1[at] = row.l1;
    t.L1s[at] = row.sin_t;
    t.L2[3 * at] = row.a; t.L2[3 * at + 1] = row.b; t.L2[3 * at + 2] = row.c;
}
// step 4: chunk bounds; rows past the live ones become "never".  thread = row, wave = chunk
__global__ __launch_bounds__(256) void k_table_chunks(Group gr, TableMap tm, Tables t_all) {
    const int kind = gr.b[blockIdx.y].kind;
    if (!kind) return;
    const Tables t = slot_tables(t_all, tm.slot[blockIdx.y]);
    const int i = blockIdx.x * blockDim.x + threadIdx.x; // < n_table (the grid covers it exactly)
    float4 p; float st;
    if ((uint32_t)i < t.hist[TB_CELLS]) {
        p = t.L1[i]; st = t.L1s[i];
    } else {
        p = (kind == 1) ? make_float4(0.f, 0.f, 0.f, -2.f) : make_float4(0.f, 0.f, -__builtin_inff(), __builtin_nanf(""));
        st = (kind == 1) ? -__builtin_inff() : 0.f;
        t.L1[i] = p; t.L1s[i] = st;
    }
    const Bound bnd = bound_of_wave(kind, p, st);
    if ((threadIdx.x & 63) == 0) { t.C1[i / 64] = bnd.p; t.C1s[i / 64] = bnd.s; }
}
// step 5: superchunk bounds from the chunk bounds: thread = chunk, wave = superchunk.  The grid covers
// top_stride * 64 chunks; those past the table are "never".
__global__ __launch_bounds__(256) void k_table_supers(Group gr, TableMap tm, Tables t_all) {
    const int kind = gr.b[blockIdx.y].kind;
    if (!kind) return;
    const Tables t = slot_tables(t_all, tm.slot[blockIdx.y]);
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    float4 p = (kind == 1) ? make_float4(0.f, 0.f, 0.f, -2.f) : make_float4(0.f, 0.f, -__builtin_inff(), __builtin_nanf(""));
    float st = (kind == 1) ? -__builtin_inff() : 0.f;
    if (c < t.n_table / 64) { p = t.C1[c]; st = t.C1s[c]; }
    const Bound bnd = bound_of_wave(kind, p, st);
    if ((threadIdx.x & 63) == 0) { t.S1[c / 64] = bnd.p; t.S1s[c / 64] = bnd.s; }
}

// ---- the all-pairs candidate search ------------------------------------------------------------------
// One workgroup = one 32x32 pixel tile, 4 rays per lane.  The primitive stream is staged through
// LDS in tiles of 512 cones (double buffered, one barrier per tile); every lane reads the same
// cone (an LDS broadcast) and tests its four directions.  The cone record holds -thr in .w, so a
// test is three fused multiply-adds, m = d.o - thr, and the pair is a candidate iff m >= 0.
// Primitives are taken four at a time: the sixteen m values are folded with max, and ONE compare
// and branch per group decides whether anything in the group needs a closer look.  That branch
// is almost never taken (a wave's 16x16 pixels see a handful of the scene's cones), and when it
// is, the group is re-examined pair by pair, the pyramid rows of the hit triangles are fetched,
// and survivors are appended to the rays' candidate lists in primitive order.
#ifndef SEARCH_GROUP
#define SEARCH_GROUP 4
#endif

// margin >= 0 (kind 1) / margin >= -m_sq (kind 2)  <=>  the pair is a candidate
template <int KIND>
__device__ __forceinline__ float pair_margin(float x, float y, float z, const float4 &p) {
    if (KIND == 1) return __builtin_fmaf(z, p.z, __builtin_fmaf(y, p.y, __builtin_fmaf(x, p.x, p.w)));
    float du = p.x - x, dv = p.y - y;
    return __builtin_fmaf(-dv, dv, __builtin_fmaf(-du, du, p.z));
}

// the rows of a triangle: is the ray's direction inside the pyramid over it / its projected origin inside the projected triangle?
template <int KIND>
__device__ __forceinline__ bool rows_pass(const float4 &a, const float4 &b, const float4 &c, float x, float y, float z, float lin) {
    float ea, eb, ec, floor_;
    if (KIND == 1) {
        ea = __builtin_fmaf(z, a.z, __builtin_fmaf(y, a.y, x * a.x));
        eb = __builtin_fmaf(z, b.z, __builtin_fmaf(y, b.y, x * b.x));
        ec = __builtin_fmaf(z, c.z, __builtin_fmaf(y, c.y, x * c.x));
        floor_ = -1e-6f;
    } else {
        ea = __builtin_fmaf(y, a.y, __builtin_fmaf(x, a.x, a.z));
        eb = __builtin_fmaf(y, b.y, __builtin_fmaf(x, b.x, b.z));
        ec = __builtin_fmaf(y, c.y, __builtin_fmaf(x, c.x, c.z));
        floor_ = -lin;
    }
    return fminf(ea, fminf(eb, ec)) >= floor_;
}
template <int KIND>
__device__ __forceinline__ void append_rows(const float4 &a, const float4 &b, const float4 &c, float x, float y,
                                            float z, float lin, uint32_t rid, uint32_t n_rays, uint32_t *cand,
                                            uint32_t &cnt, const Spill &sp) {
    float ea, eb, ec, floor_;
    if (KIND == 1) { // direction inside the pyramid over the triangle
        ea = __builtin_fmaf(z, a.z, __builtin_fmaf(y, a.y, x * a.x));
        eb = __builtin_fmaf(z, b.z, __builtin_fmaf(y, b.y, x * b.x));
        ec = __builtin_fmaf(z, c.z, __builtin_fmaf(y, c.y, x * c.x));
        floor_ = -1e-6f;
    } else {         // projected origin inside the projected triangle
        ea = __builtin_fmaf(y, a.y, __builtin_fmaf(x, a.x, a.z));
        eb = __builtin_fmaf(y, b.y, __builtin_fmaf(x, b.x, b.z));
        ec = __builtin_fmaf(y, c.y, __builtin_fmaf(x, c.x, c.z));
        floor_ = -lin;
    }
    if (fminf(ea, fminf(eb, ec)) >= floor_) {
        if (cnt < CAND_CAP) cand[(size_t)cnt * n_rays + rid] = __float_as_uint(a.w); // row 0 carries the candidate id
        else spill_candidate(sp, rid, __float_as_uint(a.w), cnt);
        cnt++;
    }
}
// a primitive without rows (sphere, plane): inside the cone / circle is all there is to test
__device__ __forceinline__ void append_id(uint32_t id, uint32_t rid, uint32_t n_rays, uint32_t *cand, uint32_t &cnt,
                                          const Spill &sp) {
    if (cnt < CAND_CAP) cand[(size_t)cnt * n_rays + rid] = id;
    else spill_candidate(sp, rid, id, cnt);
    cnt++;
}
template <int KIND>
__device__ __forceinline__ void append_candidate(const float4 *__restrict__ L2, uint32_t row, float x, float y,
                                                 float z, float lin, uint32_t rid, uint32_t n_rays, uint32_t *cand,
                                                 uint32_t &cnt, const Spill &sp) {
    const float4 a = L2[3 * (size_t)row], b = L2[3 * (size_t)row + 1], c = L2[3 * (size_t)row + 2];
    append_rows<KIND>(a, b, c, x, y, z, lin, rid, n_rays, cand, cnt, sp);
}

template <int KIND>
__global__ __launch_bounds__(SEARCH_THREADS) void k_search(Tables t_all, int n_padded,
                                                           const float4 *__restrict__ ray32_all, uint32_t n_rays,
                                                           const uint32_t *__restrict__ active_rid_all,
                                                           const uint32_t *__restrict__ n_active_all,
                                                           uint32_t *__restrict__ cand_all,
                                                           uint32_t *__restrict__ cand_cnt_all,
                                                           unsigned long long *__restrict__ counters, Spill sp_all,
                                                           Slots sl, uint32_t list_stride, uint32_t n_active_stride) {
    // the slot of this launch row: its table, rays, list and candidate arrays
    const size_t g = (size_t)sl.slot[blockIdx.y];
    const Tables tb = slot_tables(t_all, (size_t)sl.table[blockIdx.y]);
    const float4 *__restrict__ L1 = tb.L1, *__restrict__ L2 = tb.L2;
    const float4 *__restrict__ ray32 = ray32_all + g * n_rays;
    const uint32_t *__restrict__ active_rid = active_rid_all ? active_rid_all + g * list_stride : nullptr;
    const uint32_t *__restrict__ n_active = active_rid_all ? n_active_all + g * n_active_stride : nullptr;
    uint32_t *__restrict__ cand = cand_all + g * CAND_CAP * n_rays, *__restrict__ cand_cnt = cand_cnt_all + g * n_rays;
    Spill sp; sp.pool = sp_all.pool + g * POOL_STRIDE(n_rays); sp.count = sp_all.count + g;
    __shared__ float4 lds[2][PRIM_TILE];
    const uint32_t t = threadIdx.x;
    const uint32_t base = blockIdx.x * RAYS_PER_TILE + t;
    // active_rid == nullptr: the bundle is the whole frame, ray = rid.  Otherwise workgroup b
    // takes entries [1024 b, 1024 b + 1024) of the compacted list and retires if there are none.
    const uint32_t total = active_rid ? *n_active : n_rays;
    if (blockIdx.x * RAYS_PER_TILE >= total) return;
    float rx[RAYS_PER_THREAD], ry[RAYS_PER_THREAD], rz[RAYS_PER_THREAD], rw[RAYS_PER_THREAD];
    uint32_t cnt[RAYS_PER_THREAD], rid[RAYS_PER_THREAD];
    float floor_lane = 0.0f; // kind 2: the loosest -m_sq of this lane's rays
#pragma unroll
    for (int r = 0; r < RAYS_PER_THREAD; r++) {
        const uint32_t slot = base + r * SEARCH_THREADS;
        const bool live = slot < total;
        rid[r] = active_rid ? (live ? active_rid[slot] : 0u) : slot;
        float4 q = live ? ray32[rid[r]] : nan4();
        rx[r] = q.x; ry[r] = q.y; rz[r] = q.z; rw[r] = q.w; cnt[r] = 0;
        if (KIND == 2) floor_lane = fminf(floor_lane, -q.z);
    }
    const int n_tiles = max(1, min(n_padded, (int)tb.hist[TB_CELLS] + PRIM_TILE - 1) / PRIM_TILE); // live rows only
    if (t == 0) // pairs this workgroup tests: its rays x the rows it streams
        atomicAdd(&counters[((blockIdx.x + blockIdx.y) % COUNTER_SHARDS) * COUNTER_STRIDE + 5],
                  (unsigned long long)min((uint32_t)RAYS_PER_TILE, total - blockIdx.x * RAYS_PER_TILE) * (unsigned long long)(n_tiles * PRIM_TILE));
    float4 s0 = L1[t], s1 = L1[t + SEARCH_THREADS];
    lds[0][t] = s0; lds[0][t + SEARCH_THREADS] = s1;
    __syncthreads();
    for (int tile = 0; tile < n_tiles; tile++) {
        const int cur = tile & 1;
        if (tile + 1 < n_tiles) {
            s0 = L1[(size_t)(tile + 1) * PRIM_TILE + t];
            s1 = L1[(size_t)(tile + 1) * PRIM_TILE + t + SEARCH_THREADS];
        }
        const float4 *tile_lds = lds[cur];
        float4 nxt[SEARCH_GROUP];
#pragma unroll
        for (int g = 0; g < SEARCH_GROUP; g++) nxt[g] = tile_lds[g];
#pragma unroll 2
        for (int j = 0; j < PRIM_TILE; j += SEARCH_GROUP) {
            float4 p[SEARCH_GROUP];
#pragma unroll
            for (int g = 0; g < SEARCH_GROUP; g++) p[g] = nxt[g];
            const int jn = (j + SEARCH_GROUP) & (PRIM_TILE - 1); // the last prefetch wraps; its values are unused
#pragma unroll
            for (int g = 0; g < SEARCH_GROUP; g++) nxt[g] = tile_lds[jn + g];
            float top = -__builtin_inff();
#pragma unroll
            for (int g = 0; g < SEARCH_GROUP; g++) {
#pragma unroll
                for (int r = 0; r < RAYS_PER_THREAD; r++) top = fmaxf(top, pair_margin<KIND>(rx[r], ry[r], rz[r], p[g]));
            }
            if (__builtin_expect(top >= floor_lane, 0)) {
#pragma unroll
                for (int g = 0; g < SEARCH_GROUP; g++) {
                    const uint32_t prim = (uint32_t)(tile * PRIM_TILE + j + g);
#pragma unroll
                    for (int r = 0; r < RAYS_PER_THREAD; r++)
                        if (pair_margin<KIND>(rx[r], ry[r], rz[r], p[g]) >= (KIND == 2 ? -rz[r] : 0.0f))
                            append_candidate<KIND>(L2, prim, rx[r], ry[r], rz[r], rw[r], rid[r], n_rays, cand, cnt[r], sp);
                }
            }
        }
        if (tile + 1 < n_tiles) {
            lds[cur ^ 1][t] = s0; lds[cur ^ 1][t + SEARCH_THREADS] = s1;
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < RAYS_PER_THREAD; r++)
        if (base + r * SEARCH_THREADS < total) cand_cnt[rid[r]] = cnt[r];
}

// ---- packet search: beams against a three-level table ---------------------------------------------------
// The 256 rays of a wave are neighbours (on screen for camera rays, in direction for the sorted
// shadow rays), so they fit in a narrow beam: a cone around their mean direction (common-origin
// bundles) or a disc around their mean projected origin (common-direction bundles).  A primitive
// whose bounding cone / circle misses the beam cannot be a candidate of any of those rays, and
// neither can any primitive of a chunk or superchunk whose bound misses it.  The wave walks the
// table top down, 64 records per step (one coalesced 1 KiB read, served from L2): superchunk
// bounds, the chunk bounds of the surviving superchunks, the primitive records of the surviving
// chunks; each step is one beam test per lane and a __ballot.  Only the surviving primitives (a
// handful per wave) get the per-ray test of k_search, with operands broadcast from registers.
// Waves do not cooperate, so there is no barrier and no LDS; candidates come out in table order,
// which the exact resolve does not depend on.
// Wave-wide reductions: four DPP row rotations leave every lane of a 16-lane row with the row's
// result, then the four rows are combined through scalar registers.  No LDS traffic (a __shfl_xor
// butterfly is six ds_bpermute round trips).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
#define DPP_ROW_ROR(n) (0x120 + (n))
__device__ __forceinline__ float row_get(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float wave_sum_f(float v) {
    v += dpp_f<DPP_ROW_ROR(8)>(v); v += dpp_f<DPP_ROW_ROR(4)>(v); v += dpp_f<DPP_ROW_ROR(2)>(v); v += dpp_f<DPP_ROW_ROR(1)>(v);
    return (row_get(v, 0) + row_get(v, 16)) + (row_get(v, 32) + row_get(v, 48));
}
__device__ __forceinline__ float wave_min_f(float v) {
    v = fminf(v, dpp_f<DPP_ROW_ROR(8)>(v)); v = fminf(v, dpp_f<DPP_ROW_ROR(4)>(v));
    v = fminf(v, dpp_f<DPP_ROW_ROR(2)>(v)); v = fminf(v, dpp_f<DPP_ROW_ROR(1)>(v));
    return fminf(fminf(row_get(v, 0), row_get(v, 16)), fminf(row_get(v, 32), row_get(v, 48)));
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = fmaxf(v, dpp_f<DPP_ROW_ROR(8)>(v)); v = fmaxf(v, dpp_f<DPP_ROW_ROR(4)>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_ROR(2)>(v)); v = fmaxf(v, dpp_f<DPP_ROW_ROR(1)>(v));
    return fmaxf(fmaxf(row_get(v, 0), row_get(v, 16)), fmaxf(row_get(v, 32), row_get(v, 48)));
}
__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// HEAVY: the second launch of a search.  A wave whose beam grazes a finely tessellated surface meets thousands of
// primitives where the others meet ten, and a launch lasts as long as its longest wave (the 100 000-triangle frame:
// 7.5 primitives a wave on average, 3 152 in the worst; a million triangles at 7680x4320: 38 and 25 800).  A wave of
// the first launch that has met HEAVY_AFTER (256) primitives therefore stops at the next chunk, leaves its rays' counts as
// they are and puts itself on a list; in the second launch HEAVY_WORKERS waves per entry take the same rays, walk the
// same bounds (the walk is deterministic: surviving chunks are counted as they come) and share the chunks from
// where the first wave stopped, appending through append_id_shared.
// MODE 0: one launch, no wave hands over (scenes below HEAVY_MIN_PRIMS run the kernel they always ran); 1: the first
// launch of two; 2: the second.
template <int KIND, int PR, int MODE>
__device__ __forceinline__ void packet_search_wave(uint32_t block_x, const Tables &t_all,
                                                                  const float4 *__restrict__ ray32_all, uint32_t n_rays,
                                                                  const uint32_t *__restrict__ active_rid_all,
                                                                  const uint32_t *__restrict__ n_active_all,
                                                                  uint32_t *__restrict__ cand_all,
                                                                  uint32_t *__restrict__ cand_cnt_all,
                                                                  unsigned long long *__restrict__ counters, Spill sp_all,
                                                                  const Slots &sl, uint32_t list_stride, uint32_t n_active_stride,
                                                                  const Heavy &hv_all) {
    // the slot of this launch row: its tables, rays, list and candidate arrays
    const size_t g = (size_t)sl.slot[blockIdx.y];
    const Tables tb = slot_tables(t_all, (size_t)sl.table[blockIdx.y]);
    constexpr bool HEAVY = MODE == 2;
    uint2 *__restrict__ hv_list = MODE ? hv_all.list + g * HEAVY_CAP : nullptr;
    uint32_t *__restrict__ hv_count = MODE ? hv_all.count + g : nullptr;
    const float4 *__restrict__ ray32 = ray32_all + g * n_rays;
    const uint32_t *__restrict__ active_rid = active_rid_all ? active_rid_all + g * list_stride : nullptr;
    const uint32_t *__restrict__ n_active = active_rid_all ? n_active_all + g * n_active_stride : nullptr;
    uint32_t *__restrict__ cand = cand_all + g * CAND_CAP * n_rays, *__restrict__ cand_cnt = cand_cnt_all + g * n_rays;
    Spill sp; sp.pool = sp_all.pool + g * POOL_STRIDE(n_rays); sp.count = sp_all.count + g;
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t total = active_rid ? *n_active : n_rays;
    // the wave whose rays these are: this one, or (HEAVY) the wave of the first launch that this one helps
    uint32_t bx = block_x, wave = t >> 6, worker = 0, k_stop = 0;
    if (HEAVY) {
        const uint32_t entry = block_x / (hv_all.workers / PACKET_WAVES);
        if (entry >= min(*hv_count, HEAVY_CAP)) return;
        const uint2 he = hv_list[entry];
        worker = (block_x % (hv_all.workers / PACKET_WAVES)) * PACKET_WAVES + wave;
        bx = he.x >> 2; wave = he.x & 3u; k_stop = he.y;
    }
    if (bx * (PACKET_THREADS * PR) >= total) return;
    float rx[PR], ry[PR], rz[PR], rw[PR];
    uint32_t cnt[PR], rid[PR];
    bool live[PR];
    float sx = 0.f, sy = 0.f, sz = 0.f, nact = 0.f, msq_max = 0.f;
#pragma unroll
    for (int r = 0; r < PR; r++) {
        uint32_t slot;
        if (active_rid) { // sorted list: a wave takes 64*PR consecutive entries
            slot = bx * (PACKET_THREADS * PR) + wave * (64 * PR) + r * 64 + lane;
        } else {
            // camera rays: ray r of the wave's lanes = one 8x8 pixel block (4x4 quads of a quarter tile, 16
            // lanes per quad pixel, so a load touches 64-byte runs); 16 blocks per tile, a wave takes
            // PR consecutive ones
            const uint32_t blk = (bx * PACKET_WAVES + wave) * PR + (uint32_t)r;
            const uint32_t tile = blk / 16u, quarter = (blk / 4u) % 4u, qb = blk % 4u;
            const uint32_t qi = lane & 15u, pix = lane >> 4;
            const uint32_t quad = ((qb & 1u) * 4u + (qi & 3u)) + 8u * ((qb >> 1) * 4u + (qi >> 2));
            slot = tile * RAYS_PER_TILE + pix * SEARCH_THREADS + quarter * 64u + quad;
        }
        live[r] = slot < total;
        rid[r] = active_rid ? (live[r] ? active_rid[slot] : 0xffffffffu) : slot;
        if (rid[r] == 0xffffffffu) live[r] = false;
        float4 q = live[r] ? ray32[rid[r]] : nan4();
        rx[r] = q.x; ry[r] = q.y; rz[r] = q.z; rw[r] = q.w; cnt[r] = 0;
        if (q.x == q.x) {
            sx += q.x; sy += q.y; nact += 1.f;
            if (KIND == 1) sz += q.z; else msq_max = fmaxf(msq_max, q.z);
        }
    }
    // ---- the wave's beam
    sx = wave_sum_f(sx); sy = wave_sum_f(sy); sz = wave_sum_f(sz); nact = wave_sum_f(nact);
    msq_max = wave_max_f(msq_max);
    if (!(nact > 0.f)) { // nothing to search for in this wave (uniform)
#pragma unroll
        for (int r = 0; r < PR; r++) if (live[r] && !HEAVY) cand_cnt[rid[r]] = 0;
        return;
    }
    float wx, wy, wz = 0.f, w_cos = -1.f, w_sin = 0.f, w_rho = 0.f;
    bool w_all = false; // degenerate beam: everything survives
    if (KIND == 1) {
        const float wl2 = sx * sx + sy * sy + sz * sz;
        const float wi = 1.0f / sqrtf(fmaxf(wl2, 1e-30f));
        wx = sx * wi; wy = sy * wi; wz = sz * wi;
        float wlo = 1.f;
#pragma unroll
        for (int r = 0; r < PR; r++)
            if (rx[r] == rx[r]) wlo = fminf(wlo, __builtin_fmaf(rz[r], wz, __builtin_fmaf(ry[r], wy, rx[r] * wx)));
        const float w_spread = wave_min_f(wlo);
        w_all = !(wl2 > 1e-12f);
        w_cos = fmaxf(w_spread - 2e-6f, -1.f); w_sin = sqrtf(fmaxf(0.f, 1.f - w_cos * w_cos));
    } else {
        const float wi = 1.0f / fmaxf(nact, 1.f);
        wx = sx * wi; wy = sy * wi;
        float wfar = 0.f;
#pragma unroll
        for (int r = 0; r < PR; r++)
            if (rx[r] == rx[r]) {
                const float du = rx[r] - wx, dv = ry[r] - wy;
                wfar = fmaxf(wfar, __builtin_fmaf(du, du, dv * dv));
            }
        w_rho = sqrtf(wave_max_f(wfar));
        w_rho = w_rho * (1.f + 1e-5f) + 4e-6f * (fabsf(wx) + fabsf(wy) + w_rho);
        w_all = !(w_rho == w_rho) || !(w_rho < 3.0e38f);
    }
    if (w_all) { w_cos = -1.f; w_sin = 0.f; w_rho = __builtin_inff(); }

    // does the bound held in a record (of any level) meet the beam?
    auto meets = [&](const float4 &p, float st) -> bool {
        if (KIND == 1) {
            // cone (axis o, half angle theta) meets beam (axis a, half angle beta) iff
            // a.o >= cos(theta + beta), or theta + beta >= pi.  p.w = -cos(theta), st = sin(theta).
            const float ca = __builtin_fmaf(wz, p.z, __builtin_fmaf(wy, p.y, wx * p.x));
            const float rhs = __builtin_fmaf(p.w, -w_cos, __builtin_fmaf(st, -w_sin, -4e-6f));
            return (ca >= rhs) || (p.w >= w_cos); // second term: theta + beta >= pi, and the "always" records
        }
        // circle (centre c, radius R) meets disc (centre a, radius rho) iff |c - a| <= R + rho
        const float du = p.x - wx, dv = p.y - wy;
        const float d2 = __builtin_fmaf(du, du, dv * dv);
        const float R = p.w + w_rho;
        return d2 <= __builtin_fmaf(R, R * (1.f + 1e-5f), msq_max);
    };

    uint32_t n_surv = 0;
    uint32_t k_ord = 0;         // surviving chunks so far, in the order of the walk
    bool may_hand_over = MODE == 1 && bx < (1u << 30);
#ifdef RT355_DEBUG_SURVIVORS
    uint32_t dbg_super = 0, dbg_chunk = 0;
#endif
    const int n_super = tb.n_table / 4096;
    for (int top = 0; top * 64 < n_super; top++) {
        // ---- 64 superchunk bounds
        const float4 e2 = tb.S1[top * 64 + lane];
        const float e2s = (KIND == 1) ? tb.S1s[top * 64 + lane] : 0.f;
        unsigned long long m_super = __ballot(meets(e2, e2s));
#ifdef RT355_DEBUG_SURVIVORS
        dbg_super += (uint32_t)__popcll(m_super);
#endif
        float4 nq; float nqs = 0.f; // chunk bounds of the next surviving superchunk, in flight
        auto fetch_super = [&](int sc) {
            nq = tb.C1[(size_t)sc * 64 + lane];
            nqs = (KIND == 1) ? tb.C1s[(size_t)sc * 64 + lane] : 0.f;
        };
        int s_next = m_super ? top * 64 + __builtin_ctzll(m_super) : 0;
        if (m_super) fetch_super(s_next);
        while (m_super) {
            const int sc = s_next;
            m_super &= m_super - 1ull;
            const float4 e1 = nq; const float e1s = nqs;
            if (m_super) { s_next = top * 64 + __builtin_ctzll(m_super); fetch_super(s_next); }
            // ---- the 64 chunk bounds of superchunk sc
            unsigned long long m_chunk = __ballot(meets(e1, e1s));
#ifdef RT355_DEBUG_SURVIVORS
            dbg_chunk += (uint32_t)__popcll(m_chunk);
#endif
            // one primitive of a chunk (lane i's records) against the rays of every lane
            auto test_primitive = [&](int i, const float4 &p, const float4 &ra, const float4 &rb, const float4 &rc) {
                float4 q;
                q.x = lane_bcast(p.x, i); q.y = lane_bcast(p.y, i); q.z = lane_bcast(p.z, i); q.w = lane_bcast(p.w, i);
                bool hit[PR];
                bool any_hit = false;
#pragma unroll
                for (int r = 0; r < PR; r++) {
                    hit[r] = pair_margin<KIND>(rx[r], ry[r], rz[r], q) >= (KIND == 2 ? -rz[r] : 0.0f);
                    any_hit |= hit[r];
                }
                if (!any_hit) return;
                // some ray is inside the cone / circle; triangles also have pyramid / edge-function rows
                const uint32_t id = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(ra.w), i); // the candidate id
                if (id & CAND_TRI) {
                    float4 ea, eb, ec;
                    ea.x = lane_bcast(ra.x, i); ea.y = lane_bcast(ra.y, i); ea.z = lane_bcast(ra.z, i);
                    ea.w = __uint_as_float(id);
                    eb.x = lane_bcast(rb.x, i); eb.y = lane_bcast(rb.y, i); eb.z = lane_bcast(rb.z, i); eb.w = 0.f;
                    ec.x = lane_bcast(rc.x, i); ec.y = lane_bcast(rc.y, i); ec.z = lane_bcast(rc.z, i); ec.w = 0.f;
#pragma unroll
                    for (int r = 0; r < PR; r++) {
                        if (!hit[r]) continue;
                        if (!HEAVY) append_rows<KIND>(ea, eb, ec, rx[r], ry[r], rz[r], rw[r], rid[r], n_rays, cand, cnt[r], sp);
                        else if (rows_pass<KIND>(ea, eb, ec, rx[r], ry[r], rz[r], rw[r])) append_id_shared(id, rid[r], n_rays, cand, cand_cnt, sp);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < PR; r++) {
                        if (!hit[r]) continue;
                        if (!HEAVY) append_id(id, rid[r], n_rays, cand, cnt[r], sp);
                        else append_id_shared(id, rid[r], n_rays, cand, cand_cnt, sp);
                    }
                }
            };
            if (HEAVY) { // this wave's share of the chunks the first wave left: no rows in flight, a chunk at a time
                while (m_chunk) {
                    const int c = __builtin_ctzll(m_chunk);
                    m_chunk &= m_chunk - 1ull;
                    const uint32_t ord = k_ord++;
                    if (ord < k_stop || (ord - k_stop) % hv_all.workers != worker) continue;
                    const size_t at = ((size_t)sc * 64 + c) * 64 + lane;
                    const float4 p = tb.L1[at], ra = tb.L2[3 * at], rb = tb.L2[3 * at + 1], rc = tb.L2[3 * at + 2];
                    const float st = (KIND == 1) ? tb.L1s[at] : 0.f;
                    unsigned long long mask = __ballot(meets(p, st));
                    n_surv += (uint32_t)__popcll(mask);
                    while (mask) {
                        const int i = __builtin_ctzll(mask);
                        mask &= mask - 1ull;
                        test_primitive(i, p, ra, rb, rc);
                    }
                }
                continue;
            }
            float4 np, na, nb, nc; float nst = 0.f; // rows of the next surviving chunk, in flight
            auto fetch_chunk = [&](int c) {
                const size_t at = ((size_t)sc * 64 + c) * 64 + lane;
                np = tb.L1[at]; nst = (KIND == 1) ? tb.L1s[at] : 0.f;
                na = tb.L2[3 * at]; nb = tb.L2[3 * at + 1]; nc = tb.L2[3 * at + 2];
            };
            int c_next = m_chunk ? __builtin_ctzll(m_chunk) : 0;
            if (m_chunk) fetch_chunk(c_next);
            while (m_chunk) {
                m_chunk &= m_chunk - 1ull;
                const float4 p = np, ra = na, rb = nb, rc = nc;
                const float st = nst;
                if (m_chunk) { c_next = __builtin_ctzll(m_chunk); fetch_chunk(c_next); }
                k_ord++;
                // ---- the 64 primitives of the chunk
                unsigned long long mask = __ballot(meets(p, st));
                n_surv += (uint32_t)__popcll(mask);
                while (mask) {
                    const int i = __builtin_ctzll(mask);
                    mask &= mask - 1ull;
                    test_primitive(i, p, ra, rb, rc);
                }
                // ---- met enough for one wave: the rest of the walk goes to the second launch (uniform)
                if (MODE == 1 && n_surv >= hv_all.after && may_hand_over && (m_chunk || m_super || (top + 1) * 64 < n_super)) {
                    uint32_t at = 0;
                    if (lane == 0) at = atomicAdd(hv_count, 1u);
                    at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
                    if (at < HEAVY_CAP) {
                        if (lane == 0) hv_list[at] = make_uint2((bx << 2) | wave, k_ord);
#pragma unroll
                        for (int r = 0; r < PR; r++)
                            if (live[r]) {
                                cand_cnt[rid[r]] = cnt[r];
                                if (cnt[r] <= CAND_CAP) spill_head(sp)[rid[r]] = POOL_END; // no chain yet: the helpers start one
                            }
                        if (lane == 0) atomicAdd(&counters[(bx % COUNTER_SHARDS) * COUNTER_STRIDE + 3], (unsigned long long)n_surv);
                        return;
                    }
                    may_hand_over = false; // the list is full: this wave walks on by itself
                }
            }
        }
    }
    if (HEAVY) {
        if (lane == 0 && n_surv) atomicAdd(&counters[(block_x % COUNTER_SHARDS) * COUNTER_STRIDE + 3], (unsigned long long)n_surv);
        return;
    }
#pragma unroll
    for (int r = 0; r < PR; r++)
        if (live[r]) cand_cnt[rid[r]] = cnt[r];
    if (lane == 0 && n_surv) {
        atomicAdd(&counters[(block_x % COUNTER_SHARDS) * COUNTER_STRIDE + 3], (unsigned long long)n_surv);
    }
#ifdef RT355_DEBUG_SURVIVORS
    if (lane == 0) {
        atomicMax(&counters[8], (unsigned long long)n_surv);
        atomicAdd(&counters[9], 1ull);
        atomicAdd(&counters[10], (unsigned long long)dbg_super);
        atomicAdd(&counters[11], (unsigned long long)dbg_chunk);
    }
#endif
}
// The first launch: a workgroup per PACKET_WAVES waves of rays.  The second (HEAVY): a fixed grid whose workgroups
// take the helpers' places in turn, HEAVY_WORKERS / PACKET_WAVES of them per entry of the list.
#ifndef HEAVY_GRID
#define HEAVY_GRID 4096u
#endif
template <int KIND, int PR, int MODE>
#ifndef PACKET_MIN_WAVES
#define PACKET_MIN_WAVES 1 // waves per SIMD asked of the compiler (1: its own choice -- four with 4 rays a lane, seven with 1)
#endif
__global__ __launch_bounds__(PACKET_THREADS, PACKET_MIN_WAVES) void k_search_packet(Tables t_all, const float4 *__restrict__ ray32_all, uint32_t n_rays,
                                                                  const uint32_t *__restrict__ active_rid_all,
                                                                  const uint32_t *__restrict__ n_active_all,
                                                                  uint32_t *__restrict__ cand_all, uint32_t *__restrict__ cand_cnt_all,
                                                                  unsigned long long *__restrict__ counters, Spill sp_all,
                                                                  Slots sl, uint32_t list_stride, uint32_t n_active_stride,
                                                                  Heavy hv_all) {
    if (MODE != 2) {
        packet_search_wave<KIND, PR, MODE>(blockIdx.x, t_all, ray32_all, n_rays, active_rid_all, n_active_all, cand_all, cand_cnt_all,
                                            counters, sp_all, sl, list_stride, n_active_stride, hv_all);
        return;
    }
    const uint32_t places = min(hv_all.count[sl.slot[blockIdx.y]], HEAVY_CAP) * (hv_all.workers / PACKET_WAVES);
    for (uint32_t b = blockIdx.x; b < places; b += gridDim.x)
        packet_search_wave<KIND, PR, 2>(b, t_all, ray32_all, n_rays, active_rid_all, n_active_all, cand_all, cand_cnt_all,
                                           counters, sp_all, sl, list_stride, n_active_stride, hv_all);
}

// ---- search for rays that share neither origin nor direction (mirror rays) -----------------------------
// The tables above are built around one origin or one direction.  Mirror rays have neither, so they are
// searched against a table of the SCENE instead, built once per scene on the host: the bounding spheres of
// all spheres and triangles in Morton order of their centres, a bounding sphere for every 64 of them (a
// chunk) and for every 64 chunks.  A wave bounds its 64 rays by a ball around their origins and a cone around
// their directions; a sphere (c, R) can only be hit by one of them if c lies within R + rho of the origin
// ball's centre C or within the cone widened by asin((R + rho) / |c - C|).  Superchunks, chunks and
// primitives are tested against that beam 64 at a time (one per lane, one ballot), and the primitives that
// pass are tested ray by ray: distance of the centre from the ray's line, and not behind the origin.  All
// in fp32 with margins that cover the rounding of the rays to fp32 and of the arithmetic, so that a true hit
// is never lost; what passes goes to the ray's candidate list for the fp64 tests.  Planes are candidates
// of every ray.
struct GeneralTable {
    const float4 *prim;      // [n_prims_pad] centre, radius (rounded up); NaN = padding
    const uint32_t *prim_id; // [n_prims_pad] candidate id (primitive index | kind bits)
    const float4 *prim_n;    // [n_prims_pad] a triangle's unit normal and w = 1; zero for a sphere
    const float4 *mini;      // [n_supers * 512] bounding sphere of prim[8 m .. 8 m + 7]
    const float4 *chunk;     // [n_supers * 64] bounding sphere of prim[64 j .. 64 j + 63]
    const float4 *super;     // [n_supers] bounding sphere of chunk[64 k .. 64 k + 63]
    const uint32_t *plane_id;// [n_planes] candidate ids of the planes
    const uint4 *nodes;      // the 8-ary tree of walk_ray over `prim`: levels 0 .. n_lvls - 1, node k of level L is
                             // child (level_offset(L) + k) % 8 of the 64-byte group (level_offset(L) + k) / 8
    int32_t n_supers, n_planes, n_lvls;
    float tight_rho, tight_theta; // a wave whose origin ball and cone are within these searches as one beam
    float far_origin;             // SceneView::far_origin
};
struct Beam { float cx, cy, cz, rho, ax, ay, az, theta; };
// may a ray of the beam hit the sphere q?  (NaN q: no)
__device__ __forceinline__ bool beam_meets(const Beam &b, const float4 &q) {
    const float vx = q.x - b.cx, vy = q.y - b.cy, vz = q.z - b.cz;
    const float d = sqrtf(vx * vx + vy * vy + vz * vz);
    const float reach = (q.w + b.rho) * 1.00001f + 1e-6f * (1.0f + fabsf(q.x) + fabsf(q.y) + fabsf(q.z));
    if (!(d == d) || !(reach == reach)) return false;
    if (d * 0.99f <= reach) return true;                 // the origin ball touches the sphere (or nearly: asin gets steep)
    const float inv = 1.0f / d;
    const float ex = vx * inv - b.ax, ey = vy * inv - b.ay, ez = vz * inv - b.az;
    const float chord = sqrtf(ex * ex + ey * ey + ez * ez); // between two unit vectors: angle = 2 asin(chord / 2)
    const float ang = 2.0f * asinf(fminf(1.0f, 0.5f * chord));
    return ang <= b.theta + asinf(reach * inv) + 2e-5f;
}
// may the ray (o, u) hit the sphere q?
// (t_limit: the ray is only of interest up to there; the sphere is entered no earlier than along - reach)
__device__ __forceinline__ bool ray_meets(float ox, float oy, float oz, float ux, float uy, float uz, float o_l1,
                                          const float4 &q, float t_limit = 3.0e38f) {
    const float vx = q.x - ox, vy = q.y - oy, vz = q.z - oz;
    const float along = vx * ux + vy * uy + vz * uz;
    const float wx = vx - along * ux, wy = vy - along * uy, wz = vz - along * uz; // centre minus its foot on the line
    const float dist2 = wx * wx + wy * wy + wz * wz;
    const float len = fabsf(vx) + fabsf(vy) + fabsf(vz);
    const float reach = q.w + 3.8146973e-06f * (len + o_l1 + fabsf(q.x) + fabsf(q.y) + fabsf(q.z) + 1.0f); // 2^-18
    return dist2 <= reach * reach * 1.00001f && along >= -reach && along - reach <= t_limit;
}
// For a triangle the bounding sphere is followed by its plane: Triangle::CalcIntersect misses when the normal
// does not face the ray (anyobject.h:487-492), and where the ray meets the plane must lie inside the bounding
// sphere (the centroid the sphere is centred on lies on the plane, so t = n . (c - o) / n . u).  A grazing ray
// gets no opinion.
__device__ __forceinline__ bool ray_meets_plane_of(float ox, float oy, float oz, float ux, float uy, float uz, float o_l1,
                                                   const float4 &q, const float4 &nq) {
    if (nq.w == 0.0f) return true;
    const float den = nq.x * ux + nq.y * uy + nq.z * uz;
    if (den > 1e-5f) return false;
    if (den > -1e-2f) return true;
    const float vx = q.x - ox, vy = q.y - oy, vz = q.z - oz;
    const float t = (nq.x * vx + nq.y * vy + nq.z * vz) / den;
    const float slack = 2.4414062e-04f * (fabsf(vx) + fabsf(vy) + fabsf(vz) + o_l1 + fabsf(q.x) + fabsf(q.y) + fabsf(q.z) + 1.0f); // 2^-12
    if (!(t >= -slack)) return false;
    const float px = t * ux - vx, py = t * uy - vy, pz = t * uz - vz;
    const float reach = q.w + slack;
    return px * px + py * py + pz * pz <= reach * reach;
}
// The same search ray by ray: a lane walks an 8-ary tree over the same order of primitives on its own.  Level L
// of `nodes` holds the bounding BOXES of runs of 8^(n_lvls - 1 - L) primitives (a flat patch of triangles fills
// its box and very little of its sphere, and a ray that skims the patch would enter every sphere on its way);
// the boxes of the last level are those of the primitives.  A turn of the loop loads the 8 children of one node
// -- 64 bytes, the boxes as bytes on a grid of the node's own: lanes that walk apart each fetch their own lines,
// and the fetch is what a turn costs -- tests them all, and keeps the answers as a byte of a mask per level.  For waves whose rays fan out (mirror rays off small spheres, gather rays) the shared
// beam holds most of the scene and the walk per ray is the cheaper one.
__device__ __forceinline__ uint32_t level_offset(int lvl) { return ((8u << (3 * lvl)) - 8u) / 7u; } // nodes above level lvl
// A ray prepared for the box tests: the origin moved out by the ray's share of the margin on either side (the
// boxes carry their own share), and the reciprocal of the direction.
struct BoxRay { float opx, opy, opz, omx, omy, omz, ix, iy, iz; };
__device__ __forceinline__ float recip_clamped(float u) { return 1.0f / (fabsf(u) < 1e-20f ? (u < 0.0f ? -1e-20f : 1e-20f) : u); }
__device__ __forceinline__ BoxRay box_ray(float ox, float oy, float oz, float ux, float uy, float uz, float o_l1) {
    const float e = 3.8146973e-06f * o_l1 * 1.0001f; // 2^-18
    BoxRay r;
    r.opx = ox + e; r.opy = oy + e; r.opz = oz + e;
    r.omx = ox - e; r.omy = oy - e; r.omz = oz - e;
    r.ix = recip_clamped(ux); r.iy = recip_clamped(uy); r.iz = recip_clamped(uz);
    return r;
}
// The 8 children of one node, 64 bytes: the group's low corner (3 floats), a word of three power-of-two
// exponents and a mask of the children that exist, then the children's boxes as bytes on the grid
// corner + q * 2^(exponent - 127): lo.x[8], lo.y[8], lo.z[8], hi.x[8], hi.y[8], hi.z[8] (lo rounded down, hi up).
// Slab test: the ray is inside a box between the latest entry and the earliest exit over the three axes; a box
// wholly behind the origin, or entered later than t_limit, is passed over.  The margins (box_ray, and
// build_general_table for the boxes) are 2^-18 of the coordinates involved on each side, 32 times the rounding
// of the ray to fp32 and of this arithmetic.
__device__ __forceinline__ float byte_of(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }
__device__ __forceinline__ uint32_t meets8_box(const uint4 *__restrict__ g, const BoxRay &r, float t_limit) {
    const uint4 h = g[0], qa = g[1], qb = g[2], qc = g[3];
    const float gx = __uint_as_float(h.x), gy = __uint_as_float(h.y), gz = __uint_as_float(h.z);
    const float kx = __uint_as_float((h.w & 0xffu) << 23) * r.ix, ky = __uint_as_float(((h.w >> 8) & 0xffu) << 23) * r.iy,
                kz = __uint_as_float(((h.w >> 16) & 0xffu) << 23) * r.iz;
    // along an axis the ray runs up, a box is entered through its lower face and left through its upper one, and
    // the other way round: which bytes are "in" and which "out" is settled once per node, not per box
    const bool ux = r.ix >= 0.0f, uy = r.iy >= 0.0f, uz = r.iz >= 0.0f;
    const uint32_t xi0 = ux ? qa.x : qb.z, xi1 = ux ? qa.y : qb.w, xo0 = ux ? qb.z : qa.x, xo1 = ux ? qb.w : qa.y;
    const uint32_t yi0 = uy ? qa.z : qc.x, yi1 = uy ? qa.w : qc.y, yo0 = uy ? qc.x : qa.z, yo1 = uy ? qc.y : qa.w;
    const uint32_t zi0 = uz ? qb.x : qc.z, zi1 = uz ? qb.y : qc.w, zo0 = uz ? qc.z : qb.x, zo1 = uz ? qc.w : qb.y;
    // (lower faces are taken from the origin moved up by the ray's margin, upper faces from the origin moved down)
    const float xl = (gx - r.opx) * r.ix, xh = (gx - r.omx) * r.ix;
    const float yl = (gy - r.opy) * r.iy, yh = (gy - r.omy) * r.iy;
    const float zl = (gz - r.opz) * r.iz, zh = (gz - r.omz) * r.iz;
    const float xin = ux ? xl : xh, xout = ux ? xh : xl;
    const float yin = uy ? yl : yh, yout = uy ? yh : yl;
    const float zin = uz ? zl : zh, zout = uz ? zh : zl;
    uint32_t m = 0u;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int k = j & 3;
        const float t_in = fmaxf(fmaxf(fmaf(byte_of(j < 4 ? xi0 : xi1, k), kx, xin), fmaf(byte_of(j < 4 ? yi0 : yi1, k), ky, yin)),
                                 fmaf(byte_of(j < 4 ? zi0 : zi1, k), kz, zin));
        const float t_out = fminf(fminf(fmaf(byte_of(j < 4 ? xo0 : xo1, k), kx, xout), fmaf(byte_of(j < 4 ? yo0 : yo1, k), ky, yout)),
                                  fmaf(byte_of(j < 4 ? zo0 : zo1, k), kz, zout));
        m |= (t_in <= t_out * 1.000001f && t_out >= 0.0f && t_in <= t_limit) ? (1u << j) : 0u;
    }
    return m & (h.w >> 24);
}
// The walk only has to deliver the candidates the CLOSEST hit is among: as soon as one candidate is known to be
// hit at t > 0 -- by the reference's own formula, evaluated here in fp64 on a plane before the walk and on every
// sphere or triangle as it becomes a candidate -- nothing the ray enters later than that can be the closest, and
// boxes and spheres beyond are passed over.  (A mesh that a triangle hit at exactly t = 0 removes could void such
// a bound; closest_hit<.., PRUNED> sends that ray to the walk over everything.)
__device__ __forceinline__ void walk_ray(const GeneralTable &t, const SceneView &s, const Ray &e, float ox, float oy, float oz,
                                         float ux, float uy, float uz, float o_l1, uint32_t rid, uint32_t n_rays,
                                         uint32_t *__restrict__ cand, uint32_t &cnt, const Spill &sp, uint32_t &n_pairs) {
    if (t.n_lvls <= 0) return;
    const int last = t.n_lvls - 1;
    const BoxRay br = box_ray(ox, oy, oz, ux, uy, uz, o_l1);
    double t_hit = RT_FLT_MAX;
    for (int p = 0; p < t.n_planes; p++) {
        const DObject &o = s.objects[s.prim_obj[t.plane_id[p] & CAND_INDEX]];
        if (o.type != RT_OBJ_PLANE) continue;
        const double tp = plane_t(o, e);
        if (tp > 0 && tp < t_hit) t_hit = tp;
    }
    float t_limit = t_hit < 1e38 ? __double2float_ru(t_hit) * 1.000001f : 3.0e38f;
    int lvl = 0;
    uint32_t base = 0u;            // first of the 8 siblings being visited at level lvl
    unsigned long long left = 0ull; // byte L: the siblings at level L still to visit
    uint32_t m = meets8_box(t.nodes, br, t_limit);
    n_pairs += 8u;
    while (true) {
        if (m == 0u) { // this group is done: straight back to the deepest level that has siblings left
            const unsigned long long below = lvl > 0 ? (left & ((1ull << (8 * lvl)) - 1ull)) : 0ull;
            if (below == 0ull) break;
            const int up = (63 - __builtin_clzll(below)) >> 3;
            m = (uint32_t)(below >> (8 * up)) & 0xffu;
            base = (base >> (3 * (lvl - up))) & ~7u;
            lvl = up;
        }
        const uint32_t node = base + (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        if (lvl == last) {
            const float4 q = t.prim[node];
            if (ray_meets(ox, oy, oz, ux, uy, uz, o_l1, q, t_limit) && // (again: the limit may have come down since)
                ray_meets_plane_of(ox, oy, oz, ux, uy, uz, o_l1, q, t.prim_n[node])) {
                const uint32_t id = t.prim_id[node];
                append_id(id, rid, n_rays, cand, cnt, sp);
                const double4 rec = s.prim_rec[id & CAND_INDEX];
                double tc = -1;
                if (id & CAND_SPHERE) {
                    tc = sphere_t_rec(rec, e);
                } else if (id & CAND_TRI) {
                    int outside; d3 ph; double u, v;
                    tc = tri_t(s.tris[__double2hiint(rec.x)], e, &outside, &ph, &u, &v);
                }
                if (tc > 0 && tc < t_hit) {
                    t_hit = tc;
                    t_limit = tc < 1e38 ? __double2float_ru(tc) * 1.000001f : 3.0e38f;
                }
            }
            continue;
        }
        left = (left & ~(0xffull << (8 * lvl))) | ((unsigned long long)m << (8 * lvl));
        lvl++;
        base = node * 8u;
        m = meets8_box(t.nodes + (size_t)(level_offset(lvl) + base) / 8 * 4, br, t_limit);
        n_pairs += 8u;
    }
}

__global__ __launch_bounds__(256) void k_search_general(GeneralTable t, SceneView s, const double *__restrict__ ray_d,
                                                        const uint32_t *__restrict__ list,
                                                        const uint32_t *__restrict__ n_listed, uint32_t n_rays,
                                                        uint32_t *__restrict__ cand, uint32_t *__restrict__ cand_cnt,
                                                        Spill sp, unsigned long long *__restrict__ counters,
                                                        uint32_t *__restrict__ live_out) {
    // the rays of this level, compacted in tile order (k_scatter_active): 64 neighbours on screen per wave; their
    // candidates are kept under the position in that list, the level's dense numbering (see Work::dense_rid)
    const uint32_t at = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const size_t n = n_rays;
    const bool active = at < *n_listed;
    if (at == 0) *live_out = *n_listed; // (the list's own count is overwritten when the lights' rays are compacted)
    const uint32_t src = active ? list[at] : 0u; // the frame's rid: where the ray lies
    const uint32_t rid = at;
    float ox = 0, oy = 0, oz = 0, ux = 0, uy = 0, uz = 0;
    bool odd = false; // a ray the beam cannot hold (no direction, not finite): left to the fp64 walk
    Ray e;
    e.pe = D3(0, 0, 0); e.npe = D3(0, 0, 0);
    if (active) {
        e.pe = D3(ray_d[0 * n + src], ray_d[1 * n + src], ray_d[2 * n + src]);
        e.npe = D3(ray_d[3 * n + src], ray_d[4 * n + src], ray_d[5 * n + src]);
        ox = (float)e.pe.x; oy = (float)e.pe.y; oz = (float)e.pe.z;
        ux = (float)e.npe.x; uy = (float)e.npe.y; uz = (float)e.npe.z;
        const float ul = ux * ux + uy * uy + uz * uz, ol = fabsf(ox) + fabsf(oy) + fabsf(oz);
        odd = !(ul > 0.99f && ul < 1.01f) || !(ol < t.far_origin);
    }
    const bool in_beam = active && !odd;
    const float n_in = wave_sum_f(in_beam ? 1.0f : 0.0f);
    uint32_t cnt = 0, n_pairs = 0;
    if (n_in > 0.0f) { // uniform
        Beam b;
        b.cx = wave_sum_f(in_beam ? ox : 0.0f) / n_in; b.cy = wave_sum_f(in_beam ? oy : 0.0f) / n_in;
        b.cz = wave_sum_f(in_beam ? oz : 0.0f) / n_in;
        const float dx = ox - b.cx, dy = oy - b.cy, dz = oz - b.cz;
        b.rho = wave_max_f(in_beam ? sqrtf(dx * dx + dy * dy + dz * dz) : 0.0f);
        b.rho = b.rho * 1.00001f + 2e-6f * (1.0f + fabsf(b.cx) + fabsf(b.cy) + fabsf(b.cz));
        float sx = wave_sum_f(in_beam ? ux : 0.0f), sy = wave_sum_f(in_beam ? uy : 0.0f), sz = wave_sum_f(in_beam ? uz : 0.0f);
        const float sl = sqrtf(sx * sx + sy * sy + sz * sz);
        if (sl > 1e-3f * n_in) {
            b.ax = sx / sl; b.ay = sy / sl; b.az = sz / sl;
            const float ex = ux - b.ax, ey = uy - b.ay, ez = uz - b.az;
            const float ang = 2.0f * asinf(fminf(1.0f, 0.5f * sqrtf(ex * ex + ey * ey + ez * ez)));
            b.theta = wave_max_f(in_beam ? ang : 0.0f) + 2e-5f;
        } else { // the directions all but cancel: no cone
            b.ax = 1.0f; b.ay = 0.0f; b.az = 0.0f; b.theta = 4.0f;
        }
        const float o_l1 = fabsf(ox) + fabsf(oy) + fabsf(oz);
        if (in_beam)
            for (int p = 0; p < t.n_planes; p++) append_id(t.plane_id[p], rid, n_rays, cand, cnt, sp);
        const bool tight = b.rho <= t.tight_rho && b.theta <= t.tight_theta; // uniform
        if (!tight && in_beam) walk_ray(t, s, e, ox, oy, oz, ux, uy, uz, o_l1, rid, n_rays, cand, cnt, sp, n_pairs);
        for (int s0 = 0; tight && s0 < t.n_supers; s0 += 64) {
            const bool meets_s = (s0 + lane < t.n_supers) && beam_meets(b, t.super[s0 + lane]);
            unsigned long long ms = __ballot(meets_s);
            while (ms) { // uniform
                const int k = s0 + __builtin_ctzll(ms);
                ms &= ms - 1;
                unsigned long long mc = __ballot(beam_meets(b, t.chunk[(size_t)k * 64 + lane]));
                while (mc) {
                    const size_t j = (size_t)k * 64 + (size_t)__builtin_ctzll(mc);
                    mc &= mc - 1;
                    const float4 q = t.prim[j * 64 + lane];
                    const float4 qn = t.prim_n[j * 64 + lane];
                    const uint32_t qid = t.prim_id[j * 64 + lane];
                    unsigned long long mp = __ballot(beam_meets(b, q));
                    while (mp) {
                        const int l = __builtin_ctzll(mp);
                        mp &= mp - 1;
                        float4 one;
                        one.x = lane_bcast(q.x, l); one.y = lane_bcast(q.y, l); one.z = lane_bcast(q.z, l); one.w = lane_bcast(q.w, l);
                        float4 one_n;
                        one_n.x = lane_bcast(qn.x, l); one_n.y = lane_bcast(qn.y, l); one_n.z = lane_bcast(qn.z, l); one_n.w = lane_bcast(qn.w, l);
                        const uint32_t id = (uint32_t)__builtin_amdgcn_readlane((int)qid, l);
                        n_pairs += in_beam ? 1u : 0u;
                        if (in_beam && ray_meets(ox, oy, oz, ux, uy, uz, o_l1, one) &&
                            ray_meets_plane_of(ox, oy, oz, ux, uy, uz, o_l1, one, one_n))
                            append_id(id, rid, n_rays, cand, cnt, sp);
                    }
                }
            }
        }
    }
    if (active) cand_cnt[rid] = odd ? CAND_UNSEARCHED : cnt;
    wave_count(counters, 5, n_pairs);
}

// ---- fp64 closest hit over a ray's candidates (or over everything) -----------------------------------
// The candidates of a ray come in table order, not list order, so the list-order rules of the
// reference are restated without order: the closest-hit loop (main.cpp:65-79) keeps the FIRST object
// with the smallest t >= 0, i.e. the smallest (t, index) pair; a Mesh reports the first triangle with
// its smallest t >= 0, but only if that t is > 0 (the same loop inside Mesh::CalcIntersect), so a
// triangle hit at exactly t = 0 removes the whole mesh.  Hence: smallest (t, primitive) over sphere and
// plane hits with t >= 0 and triangle hits with t > 0, leaving out the triangles of a mesh that has a
// hit at t = 0.  Such a mesh is only known once all candidates were seen, so if the winner belongs to it
// the candidates are walked once more without it; two such meshes send the ray to the ordered walk over
// everything (as does a ray whose candidates did not fit its own slots AND the bundle's spill list).
// POSITIVE: the gather loops' rule, an object's hit counts only with t > 0 (shader.h:130,176).
// PRUNED: the candidates come from walk_ray, which stops at the closest hit it knows of: should a mesh be removed,
// what lies beyond was never listed, and the ray goes to the ordered walk over everything.
// What a kernel loads ahead of time for a ray's exact tests: the candidate count, the first CAND_BATCH ids and (the
// shading kernel with SHADE_PREFETCH 3) their records.
struct Pre { uint32_t n; uint32_t id[CAND_BATCH]; double4 rec[CAND_BATCH]; };
template <bool POSITIVE = false, bool ENV = true, bool PRUNED = false>
__device__ __forceinline__ void closest_hit(const SceneView &s, const Ray &e, bool brute, uint32_t rid,
                                            uint32_t n_rays, const uint32_t *__restrict__ cand,
                                            const uint32_t *__restrict__ cand_cnt, const Spill &sp, Closest &c,
                                            uint32_t &evaluated, uint32_t &overflowed, const Pre *pre = nullptr) {
    c.init();
    uint32_t n = 0, n_spill = 0;
    if (!brute) {
        n = pre ? pre->n : cand_cnt[rid];
        if (n == CAND_UNSEARCHED) { brute = true; n = 0; }
        else if (n > CAND_CAP) { // the rest is in the bundle's spill list, unless that overflowed too
            overflowed += 1;
            n_spill = *sp.count;
            if (n_spill > POOL_CAP) brute = true;
        }
    }
    if (!brute) {
        const uint32_t *cp = cand + rid;
        const uint32_t n_own = min(n, (uint32_t)CAND_CAP);
        int zero_obj = -1, exclude = -1;
        bool two_zero = false;
        for (int pass = 0; pass < 2; pass++) {
            double bt = RT_FLT_MAX;
            int bprim = -1, bobj = -1;
            auto consider = [&](uint32_t id, const double4 &rec) {
                const int prim = (int)(id & CAND_INDEX);
                double t; int obj = -1;
                if (ENV && (id & CAND_ENVIRONMENT) == CAND_ENVIRONMENT) {
                    int inside;
                    t = environment_t_rec(rec, e, &inside);
                } else if (id & CAND_SPHERE) {
                    t = sphere_t_rec(rec, e);
                } else if (id & CAND_TRI) {
                    obj = __double2loint(rec.x);
                    if (obj == exclude) return;
                    int outside; d3 ph; double u, v;
                    t = tri_t(s.tris[__double2hiint(rec.x)], e, &outside, &ph, &u, &v);
                    if (t == 0) { // also -0.0
                        if (zero_obj >= 0 && zero_obj != obj) two_zero = true;
                        zero_obj = obj;
                        return;
                    }
                } else if (ENV && __double2hiint(rec.x) == REC_QUADRIC) {
                    int outside, branch;
                    t = quadric_t(s.quadrics[s.objects[__double2loint(rec.x)].tri_begin], e, &outside, &branch);
                } else {
                    t = plane_t(s.objects[__double2loint(rec.x)], e);
                }
                if ((POSITIVE ? t > 0 : t >= 0) && (t < bt || (t == bt && prim < bprim))) { bt = t; bprim = prim; bobj = obj; }
            };
            // CAND_BATCH candidates at a time: first their ids, then their 32-byte records (independent
            // loads, all in flight together), then the tests
            for (uint32_t k0 = 0; k0 < n_own; k0 += CAND_BATCH) {
                uint32_t id[CAND_BATCH];
                double4 rec[CAND_BATCH];
#pragma unroll
                for (int j = 0; j < CAND_BATCH; j++)
                    id[j] = (k0 + j < n_own) ? ((pre && k0 == 0) ? pre->id[j] : cp[(size_t)(k0 + j) * n_rays]) : 0u;
#pragma unroll
                for (int j = 0; j < CAND_BATCH; j++) rec[j] = s.prim_rec[id[j] & CAND_INDEX];
#pragma unroll
                for (int j = 0; j < CAND_BATCH; j++)
                    if (k0 + j < n_own) consider(id[j], rec[j]);
            }
            if (n > CAND_CAP) // the ray's own chain in the pool
                for (uint32_t i = spill_head(sp)[rid]; i != POOL_END;) {
                    const uint2 entry = sp.pool[i];
                    consider(entry.x, s.prim_rec[entry.x & CAND_INDEX]);
                    i = entry.y;
                }
            c.t_best = bt; c.prim_best = bprim;
            if (PRUNED && zero_obj >= 0) { two_zero = true; break; }
            if (pass == 0 && zero_obj >= 0 && bobj == zero_obj && !two_zero) exclude = zero_obj; // once more without it
            else break;
        }
        if (two_zero) brute = true;
    }
    if (brute) {
        c.init();
        closest_by_kind<POSITIVE, false, ENV>(s, e, c.t_best, c.prim_best);
        n = (uint32_t)s.n_prims;
    }
    evaluated += n;
}

// CalcShadow only uses the distance of the closest occluder (shader.h:403-415), so the shadow rays' walk
// over the candidates keeps a minimum and nothing else.  A triangle hit at exactly t = 0 (which removes its
// mesh, see above), an overflowed list or an unsearched bundle go through closest_hit.
// the rare cases of nearest_occluder (inlined: a call costs the common case more than the registers do)
template <bool ENV = true>
__device__ __forceinline__ double nearest_occluder_general(const SceneView &s, const Ray &e, bool brute, uint32_t rid,
                                                        uint32_t n_rays, const uint32_t *__restrict__ cand,
                                                        const uint32_t *__restrict__ cand_cnt, const Spill &sp,
                                                        uint32_t *evaluated, uint32_t *overflowed) {
    Closest c;
    uint32_t ev = 0, ov = 0;
    closest_hit<false, ENV>(s, e, brute, rid, n_rays, cand, cand_cnt, sp, c, ev, ov);
    *evaluated += ev; *overflowed += ov;
    return c.t_best;
}
template <bool ENV = true>
__device__ __forceinline__ double nearest_occluder(const SceneView &s, const Ray &e, bool brute, uint32_t rid,
                                                   uint32_t n_rays, const uint32_t *__restrict__ cand,
                                                   const uint32_t *__restrict__ cand_cnt, const Spill &sp,
                                                   uint32_t &evaluated, uint32_t &overflowed,
                                                   const Pre *pre = nullptr) {
    const uint32_t n = brute ? 0u : (pre ? pre->n : cand_cnt[rid]);
    bool general = brute || n > CAND_CAP;
    double bt = RT_FLT_MAX;
    if (!general) {
        const uint32_t *cp = cand + rid;
        for (uint32_t k0 = 0; k0 < n; k0 += CAND_BATCH) {
            uint32_t id[CAND_BATCH];
            double4 rec[CAND_BATCH];
#pragma unroll
            for (int j = 0; j < CAND_BATCH; j++)
                id[j] = (k0 + j < n) ? ((SHADE_PREFETCH >= 2 && pre && k0 == 0) ? pre->id[j] : cp[(size_t)(k0 + j) * n_rays]) : 0u;
#pragma unroll
            for (int j = 0; j < CAND_BATCH; j++)
                rec[j] = (SHADE_PREFETCH >= 3 && pre && k0 == 0) ? pre->rec[j] : s.prim_rec[id[j] & CAND_INDEX];
#pragma unroll
            for (int j = 0; j < CAND_BATCH; j++) {
                if (k0 + j >= n) break;
                double t;
                if (ENV && (id[j] & CAND_ENVIRONMENT) == CAND_ENVIRONMENT) {
                    int inside;
                    t = environment_t_rec(rec[j], e, &inside);
                } else if (id[j] & CAND_SPHERE) {
                    t = sphere_t_rec(rec[j], e);
                } else if (id[j] & CAND_TRI) {
                    int outside; d3 ph; double u, v;
                    t = tri_t(s.tris[__double2hiint(rec[j].x)], e, &outside, &ph, &u, &v);
                    if (t == 0) { general = true; t = -1; }
                } else if (ENV && __double2hiint(rec[j].x) == REC_QUADRIC) {
                    int outside, branch;
                    t = quadric_t(s.quadrics[s.objects[__double2loint(rec[j].x)].tri_begin], e, &outside, &branch);
                } else {
                    t = plane_t(s.objects[__double2loint(rec[j].x)], e);
                }
                if (t >= 0 && t < bt) bt = t;
            }
        }
        evaluated += n;
    }
    if (general) {
        bt = nearest_occluder_general<ENV>(s, e, brute, rid, n_rays, cand, cand_cnt, sp, &evaluated, &overflowed);
        if (!brute && n <= CAND_CAP) evaluated -= n; // walked twice, counted once
    }
    return bt;
}

// The closest occluder of one ray (the same in every lane) over EVERY primitive, the lanes of the wave sharing
// the list: the primitive that nearest_occluder's walk over everything takes its distance from, found 64
// primitives at a time; -1 when there is none.  Objects take hits with t >= 0, triangles with t > 0 (a mesh
// reports its smallest t only if that is > 0); a triangle hit at exactly t = 0 removes its whole mesh, which only
// the ordered walk knows how to do: -2 then.
__device__ __forceinline__ int nearest_occluder_wave(const SceneView &s, const Ray &e) {
    const int lane = threadIdx.x & 63;
    const int first_env = s.n_planes + s.n_spheres, first_quadric = first_env + s.n_envs, first_tri = first_quadric + s.n_quadrics;
    double bt = RT_FLT_MAX;
    int bprim = -1;
    bool zero_tri = false;
    for (int k = lane; k < s.n_prims; k += 64) {
        const int prim = s.kind_prims[k];
        double t;
        if (k >= first_tri) {
            int outside; d3 ph; double u, v;
            t = tri_t(s.tris[s.prim_tri[prim]], e, &outside, &ph, &u, &v);
            if (t == 0) { zero_tri = true; t = -1; }
        } else if (k < s.n_planes) {
            t = plane_t(s.objects[s.prim_obj[prim]], e);
        } else if (k < first_env) {
            t = sphere_t_rec(s.prim_rec[prim], e);
        } else if (k < first_quadric) {
            int inside;
            t = environment_t_rec(s.prim_rec[prim], e, &inside);
        } else {
            int outside, branch;
            t = quadric_t(s.quadrics[s.objects[s.prim_obj[prim]].tri_begin], e, &outside, &branch);
        }
        if (t >= 0 && t < bt) { bt = t; bprim = prim; }
    }
    if (__any(zero_tri)) return -2;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ot = __shfl_xor(bt, off, 64);
        const int op = __shfl_xor(bprim, off, 64);
        if (ot < bt || (ot == bt && op >= 0 && (bprim < 0 || op < bprim))) { bt = ot; bprim = op; }
    }
    return bprim;
}

// the colour of grid point (x, y): a pixel of the frame, or a sub-sample kept for k_average
__device__ __forceinline__ void store_pixel(unsigned char *__restrict__ pixmap, const Frame &f, const Work &w,
                                            uint32_t rid, int x, int y, uint32_t clr) {
    if (f.level >= 0) { w.lvl_clr[(size_t)f.level * w.n + rid] = clr; return; }
    if (f.alias > 0 || f.deferred) { w.rec_sum[rid] = clr; return; }
    size_t i = ((size_t)frame_row(f, y) * (size_t)f.width + (size_t)x) * 3;
    pixmap[i] = (unsigned char)cr(clr); pixmap[i + 1] = (unsigned char)cg(clr); pixmap[i + 2] = (unsigned char)cb(clr);
}
// SEG neighbouring pixels of one row, one per lane in SEG consecutive lanes (SEG = 32: each half of the wave has a
// row of its own; SEG = 64: the whole wave), stored as 3 SEG / 4 whole dwords: lane l of the segment builds dword l
// from the colours of pixels a = 4 l / 3 and a + 1, which it takes from their lanes.  A segment that is not whole
// (the frame's right edge, a row past the slab) or not dword-aligned (a width that is no multiple of 4) falls back
// to byte stores.  Called by all lanes of the wave; `valid`: this lane has a pixel.
template <int SEG>
__device__ __forceinline__ void store_row_segment(unsigned char *__restrict__ pixmap, const Frame &f, int x, int y,
                                                  bool valid, uint32_t clr) {
    const int lane = threadIdx.x & 63, l = lane & (SEG - 1), first = lane - l;
    const int a = (4 * l) / 3, shift = 8 * (4 * l - 3 * a);
    const uint32_t ca = (uint32_t)__shfl((int)clr, first + min(a, SEG - 1), 64);
    const uint32_t cb_ = (uint32_t)__shfl((int)clr, first + min(a + 1, SEG - 1), 64);
    const unsigned long long have = __ballot(valid);
    const unsigned long long mine = (SEG == 64) ? ~0ull : (0xffffffffull << first);
    if (!valid) return;
    unsigned char *at = pixmap + ((size_t)frame_row(f, y) * (size_t)f.width + (size_t)(x - l)) * 3; // the segment's first byte
    if ((have & mine) == mine && ((uintptr_t)at & 3u) == 0) {
        if (l < 3 * SEG / 4)
            ((uint32_t *)at)[l] = (uint32_t)(((unsigned long long)(ca & 0xffffffu) | ((unsigned long long)(cb_ & 0xffffffu) << 24)) >> shift);
    } else {
        at += 3 * l;
        at[0] = (unsigned char)cr(clr); at[1] = (unsigned char)cg(clr); at[2] = (unsigned char)cb(clr);
    }
}
// The one-kernel renderers (a thread does the whole pixel) number their grid points so that a workgroup owns whole
// lines of the pixmap: 256 neighbouring grid points of one row per workgroup (768 contiguous bytes), 64 per wave.
#define DIRECT_SEG 64
__device__ __forceinline__ void direct_xy(const Frame &f, int *x, int *y) {
    const uint32_t across = ((uint32_t)f.grid_w + blockDim.x - 1u) / blockDim.x;
    *x = (int)((blockIdx.x % across) * blockDim.x + threadIdx.x);
    *y = (int)(blockIdx.x / across);
}
// ... and what they do with the colour: a sub-sample goes to k_average's array, a pixel to the pixmap
__device__ __forceinline__ void direct_store(unsigned char *__restrict__ pixmap, const Frame &f, const Work &w, int x, int y,
                                             bool valid, uint32_t clr) {
    if (f.alias > 0) { // uniform
        if (valid) w.rec_sum[xy_to_rid(x, y, f.tiles_x)] = clr;
        return;
    }
    store_row_segment<DIRECT_SEG>(pixmap, f, x, y, valid, clr);
}
// CalcPixel's average over the alias x alias sub-samples of a pixel (main.cpp:315-325): sums of the
// sub-samples' 8-bit channels in double, divided by alias * alias, truncated by Color's constructor
// (alias == 0: the one colour of the pixel as a searched frame left it in the array.)  A wave takes 64 neighbouring
// pixels of a row and writes them as 48 dwords.
__global__ __launch_bounds__(256) void k_average(Frame f, const uint32_t *__restrict__ sub,
                                                 unsigned char *__restrict__ pixmap) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = f.avg_row0 + (int)blockIdx.y;
    const bool valid = x < f.width && y < f.rows;
    uint32_t out = 0u;
    if (valid && f.alias <= 0) {
        out = sub[xy_to_rid(x, y, f.tiles_x)];
    } else if (valid) {
        double red = 0, green = 0, blue = 0;
        for (int q = 0; q < f.alias; q++)
            for (int p = 0; p < f.alias; p++) {
                const uint32_t clr = sub[xy_to_rid(x * f.alias + p, y * f.alias + q, f.tiles_x)];
                red += (double)(int)cr(clr); green += (double)(int)cg(clr); blue += (double)(int)cb(clr);
            }
        const double n = (double)(f.alias * f.alias);
        out = RGB(q8(red / n), q8(green / n), q8(blue / n));
    }
    store_row_segment<64>(pixmap, f, x, y, valid, out);
}

// Shadow rays of one group of lights for a primary hit (shader.h:395-404): the fp32 search record of
// each ray goes to the light's slot, and rays of binned bundles are counted into direction bins.
// Called by all lanes of a wave; returns the number of shadow rays this lane's hit needs.
__device__ __forceinline__ uint32_t emit_shadow_rays(const ShaderView &sh, const Group &gr, const Work &w,
                                                     uint32_t rid, bool in_range, bool hit, d3 ph, d3 nh, bool soft) {
    if (soft) return 0u; // (the shadow rays of CalcSoftShadow share neither origin nor direction: none is searched, and the shading kernel counts them)
    uint32_t n_shadow = 0;
    // Rays of binned bundles are counted into the direction bins: the counter value every lane's cell had before this
    // wave arrived is still in flight (one returning atomic per wave, cell and light) while the other lights are
    // dealt with.  Which lights the hit faces goes into one byte per ray; counter and rank are written for the rays
    // that are cast only (for eight lights that is half the ray slots).
    uint32_t facing = 0;
    bool any_binned = false;
    uint32_t pending[LIGHT_GROUP], rank_in_wave[LIGHT_GROUP], leader[LIGHT_GROUP];
#pragma unroll
    for (int g = 0; g < LIGHT_GROUP; g++) {
        pending[g] = 0; rank_in_wave[g] = 0; leader[g] = 0;
        if (g >= gr.count) continue; // uniform
        const Bundle &bd = gr.b[g];
        const size_t at = (size_t)g * w.n + rid;
        uint32_t bin = 0xffffffffu;
        if (hit) {
            // shader.h:396 casts the ray unless nh . (-nhl) <= 0, nhl = Normalize(ph - position).  The sign of that
            // sum is the sign of -(nh . (ph - position)) whenever the latter is not within rounding of zero (the
            // normalised components carry < 5 units of roundoff each, the sum below < 3; 2^-47 of the terms'
            // magnitudes is 64 units), and the fp32 search record only needs the direction to ~1e-9: so the square
            // root and the three divisions of the normalisation are made only for a hit that is tangent to the
            // light within rounding (or degenerate), which then takes the reference's own operations.
            const DLight &L = sh.lights[gr.k0 + g];
            d3 nhl;
            bool needs;
            const d3 v = vsub(ph, ld3(L.position));
            const double px = nh.x * v.x, py = nh.y * v.y, pz = nh.z * v.z, len2 = v.x * v.x + v.y * v.y + v.z * v.z;
            const double sum = px + py + pz, mag = fabs(px) + fabs(py) + fabs(pz);
            if (L.type != 0 && fabs(sum) > mag * 0x1p-47 && len2 > 1e-200 && len2 < 1e200) {
                needs = sum < 0;
                double y = __builtin_amdgcn_rsq(len2);           // ~2^-26; one Newton step: ~2^-51
                y = fma(y * 0.5, fma(-len2 * y, y, 1.0), y);
                nhl = vmul(v, y);
            } else {
                nhl = light_nhl(L, ph);
                needs = !(dot(nh, vmul(nhl, -1)) <= 0);
            }
            n_shadow += needs ? 1u : 0u;
            if (bd.kind) {
                float4 q = needs ? bundle_ray(bd, ph, nhl) : nan4();
                if (needs || !bd.binned) w.ray32[at] = q;
                if (bd.binned && needs) bin = ray_bin(bd.kind, q, bd.plane_extent);
            }
        } else if (in_range && bd.kind && !bd.binned) {
            w.ray32[at] = nan4();
        }
        if (bd.binned) {
            const uint32_t counter = bin != 0xffffffffu ? hist_index(bin, rid) : bin;
            pending[g] = wave_bin_count(w.bin_hist + (size_t)g * HIST_STRIDE, counter, bin != 0xffffffffu, rank_in_wave[g], leader[g]);
            any_binned = true;
            if (bin != 0xffffffffu) { facing |= 1u << g; w.bin_id[at] = counter; }
        }
    }
    if (any_binned && in_range) w.face[rid] = (uint8_t)facing;
#pragma unroll
    for (int g = 0; g < LIGHT_GROUP; g++) {
        if (g >= gr.count || !gr.b[g].binned) continue; // uniform
        const uint32_t base = (uint32_t)__shfl((int)pending[g], (int)leader[g], 64);
        if ((facing >> g) & 1u) w.bin_rank[(size_t)g * w.n + rid] = base + rank_in_wave[g];
    }
    return n_shadow;
}

// The accepted hit as the object's CalcIntersect returns it (main.cpp:69-77): point, normal, the mesh's
// interpolated UV.
struct HitGeo {
    d3 ph, nh;
    double uvx, uvy;
    int obj;
};
template <bool ENV = true>
__device__ __forceinline__ HitGeo hit_geometry(const SceneView &s, const Ray &e, const Closest &c) {
    HitGeo h;
    h.obj = s.prim_obj[c.prim_best];
    h.uvx = 0; h.uvy = 0;
    const int tri = s.prim_tri[c.prim_best];
    const DObject &o = s.objects[h.obj];
    if (tri >= 0) {
        const DTriangle &T = s.tris[tri];
        int outside; double u, v;
        tri_t(T, e, &outside, &h.ph, &u, &v);
        h.nh = vmul(ld3(T.normal), (double)outside);                 // anyobject.h:517
        double wgt = 1 - u - v;                                      // Triangle::GetUV, anyobject.h:524-526
        h.uvx = T.uv0[0] * wgt + T.uv1[0] * u + T.uv2[0] * v;
        h.uvy = T.uv0[1] * wgt + T.uv1[1] * u + T.uv2[1] * v;
    } else if (o.type == RT_OBJ_PLANE) {
        d3 ni = ld3(o.n);
        int outside = (dot(ni, vsub(e.pe, ld3(o.p))) <= 0) ? -1 : 1; // anyobject.h:71-77
        h.ph = vadd(e.pe, vmul(e.npe, c.t_best));                    // :85
        h.nh = vmul(ni, (double)outside);                            // :86
    } else if (ENV && o.type == RT_OBJ_ENVIRONMENT) {
        int inside;
        environment_t_rec(make_double4(o.p[0], o.p[1], o.p[2], o.radius), e, &inside);
        h.ph = vadd(e.pe, vmul(e.npe, c.t_best));                    // anyobject.h:215
        h.nh = vmul(normalize(vsub(ld3(o.p), h.ph)), (double)inside); // :218
    } else if (ENV && o.type == RT_OBJ_QUADRIC) {
        const DQuadric &q = s.quadrics[o.tri_begin];
        int outside, branch;
        quadric_t(q, e, &outside, &branch);
        h.ph = vadd(e.pe, vmul(e.npe, c.t_best));                    // anyobject.h:314,328
        h.nh = vmul(quadric_normal(q, h.ph), (double)outside);       // :320,339
    } else {
        int outside;
        sphere_t(o, e, &outside);
        h.ph = vadd(e.pe, vmul(e.npe, c.t_best));                    // :155
        h.nh = vmul(normalize(vsub(h.ph, ld3(o.p))), (double)outside); // :156
    }
    return h;
}
// What Shader::shading needs of the accepted hit (main.cpp:69-77, shader.h:53-58), plus the two terms of
// its light loop that do not depend on the light: reh of CalcSpecular and the weight of CalcBorder.
struct HitRec {
    d3 ph, nh, reh;
    double border_w;
    uint32_t cm0, cm1;
    int obj;
    bool no_border;
    bool emissive; // an Environment: Shader::shading returns its (textured) colour before the lights (shader.h:70-72)
};
template <bool ENV = true>
__device__ __forceinline__ HitRec primary_record(const SceneView &s, const ShaderView &sh, const Ray &e, const Closest &c) {
    const HitGeo g = hit_geometry<ENV>(s, e, c);
    HitRec h;
    h.obj = g.obj; h.ph = g.ph; h.nh = g.nh;
    const DObject &o = s.objects[h.obj];
    h.cm0 = o.rgb; h.cm1 = sh.ambient;
    if (o.texture_type == 1) texture_mapping(s, o, h.ph, g.uvx, g.uvy, &h.cm0, &h.cm1); // shader.h:53-58
    else if (o.texture_type == 2) solid_texture(s, o, h.ph, sh.ambient, &h.cm0, &h.cm1); // :59-63
    else if (o.texture_type == 3) h.cm0 = function_3d(h.ph);                             // :64-68 (cm1 = ambient)
    h.no_border = (o.type == RT_OBJ_PLANE || o.type == RT_OBJ_MESH); // main.cpp:74
    h.emissive = ENV && (o.type == RT_OBJ_ENVIRONMENT);
    h.reh = mirror_dir(h.nh, e.npe);
    h.border_w = h.no_border ? 0.0 : border_weight(e.pe, h.ph, h.nh);
    return h;
}
// One turn of Shader::shading's light loop (shader.h:74-93): colour before the shadow, CalcShadow with the
// closest occluder's distance from `occluder(shadow ray)`, running mean.
// What CalcSubPixel receives from CalcPixel for grid point (gx, gy) under RT355_SHADING_SOFT (rt355x_soft_args in
// rt355.h): the object a ray that misses goes on with, and Shader::shading's m and n.
struct SoftArgs { int obj0; double m, n; };
__device__ __forceinline__ SoftArgs soft_args(const Frame &f, int gx, int gy) {
    SoftArgs a; a.obj0 = 0; a.m = 0; a.n = 0;
    if (f.soft && f.alias > 0) {
        int32_t o;
        rt355x_soft_args(f.seed, gx / f.alias, frame_row(f, gy / f.alias), gx % f.alias, gy % f.alias, f.alias, &o, &a.m, &a.n);
        a.obj0 = o;
    }
    return a;
}
// SOFT: CalcSoftShadow (shader.h:439-483) in place of CalcShadow -- the shadow ray aims at the point
// lights[0].position + n0 * sx * m + n1 * sy * n of lights[0]'s area whichever light is being shaded (:441-443), the
// facing test takes the vector to that point as it is (:445-446), L stays the distance to light k's position (:462).
// SKIP: leave out the highlight and the shadow mix where they change nothing (see calc_specular, shadow_mix).
// light_colour: what light k gives the hit, shadow included (shader.h:75-92); shade_light: that, taken into the
// running mean over the lights (shader.h:93).
template <bool SOFT = false, bool SKIP = false, class Occluder>
__device__ __forceinline__ uint32_t light_colour(const ShaderView &sh, int k, const HitRec &h,
                                                 Occluder occluder, uint32_t &n_shadow, double soft_m = 0, double soft_n = 0) {
    const DLight &L = sh.lights[k];
    // nhl of shader.h:75-82, and with it |ph - position|: CalcShadow's L = |position - ph| (shader.h:412) is the same
    // double (the components are negated exactly, so the squares and their sum are the same), one square root saved
    d3 nhl;
    double dist = 0;
    if (L.type == 0) {
        nhl = ld3(L.direction);
    } else {
        nhl = vsub(h.ph, ld3(L.position));
        dist = vlen(nhl);
        if (dist != 0) nhl = vdiv(nhl, dist); // Normalize, vector3D.h:79-88
    }
    uint32_t clr = calc_diffuse(L, h.cm0, h.cm1, h.nh, nhl);
    clr = calc_specular<SKIP>(sh, L, clr, h.reh, nhl);
    if (!h.no_border) clr = calc_border(sh, clr, h.border_w);
    double shd = 0.5; // shader.h:396-398
    if (SOFT) {
        const DLight &L0 = sh.lights[0];
        d3 pl = ld3(L0.position);
        pl = vadd(vadd(pl, vmul(vmul(ld3(L0.n0), L0.sx), soft_m)), vmul(vmul(ld3(L0.n1), L0.sy), soft_n)); // :443
        const d3 nlh = vsub(pl, h.ph);                                                                      // :445
        if (!(dot(h.nh, nlh) <= 0)) {
            const double t_min = occluder(mkview(h.ph, nlh));                                               // :454
            const double Ld = L.type != 0 ? dist : vlen(vsub(ld3(L.position), h.ph));                       // :462
            shd = (t_min > 0 && t_min < Ld) ? t_min / Ld : 1.0;
            n_shadow += 1;
        }
    } else if (!(dot(h.nh, vmul(nhl, -1)) <= 0)) {
        Ray sr; // View(ph, nhl * (-1)), shader.h:395,404
        if (L.type == 0) { sr.pe = h.ph; sr.npe = ld3(L.shadow_dir); }
        else sr = mkview(h.ph, vmul(nhl, -1));
        const double t_min = occluder(sr);
        const double Ld = L.type != 0 ? dist : vlen(vsub(ld3(L.position), h.ph)); // shader.h:412
        shd = (t_min > 0 && t_min < Ld) ? t_min / Ld : 1.0;
        n_shadow += 1;
    }
    return shadow_mix<SKIP>(sh, L, clr, nhl, shd);
}
template <bool SOFT = false, bool SKIP = false, class Occluder>
__device__ __forceinline__ uint32_t shade_light(const ShaderView &sh, int k, const HitRec &h, uint32_t sum,
                                                Occluder occluder, uint32_t &n_shadow, double soft_m = 0, double soft_n = 0) {
    return light_mean(sum, light_colour<SOFT, SKIP>(sh, k, h, occluder, n_shadow, soft_m, soft_n), k);
}

// The whole pixel in one thread, the reference's way (every primitive tested in fp64 for the camera ray and
// for every shadow ray): used when nothing is searched -- scenes of a few objects, where tables and
// candidate lists cost more than they save, and RT355_MODE_BRUTE.
// REFLECT: the reflection recursion of CalcSubPixel (main.cpp:93-112) as a loop over at most four rays per
// pixel (REFLECTION_TIMES_THRES 3, REFLECTION_KTOTAL_THRES 0.05, main.cpp:27-28), colours folded back from the
// deepest.  On a miss the reference goes on with objs[0]'s ks, a zero hit point and a zero normal; so does
// this.  Compiled apart so that scenes without mirrors do not carry the loop, and (ENV) scenes without an
// Environment not its tests.
// (REFLECT: three waves per SIMD -- left alone the compiler takes 207 registers, two waves, and 26 % longer)
template <bool REFLECT, bool ENV, bool SOFT = false>
__global__ __launch_bounds__(256, REFLECT ? 3 : 1) void k_render_direct(SceneView s, ShaderView sh, DCamera cam, Frame f, Work w,
                                                       unsigned char *__restrict__ pixmap) {
    const uint32_t rid = 0; // (no per-ray lists on this path)
    uint32_t n_eval = 0, n_over = 0, n_hit = 0, n_shadow = 0, n_secondary = 0;
    int x, y;
    direct_xy(f, &x, &y);
    const bool valid = x < f.grid_w && y < f.grid_h;
    uint32_t pixel = 0u;
    do {
    if (!valid) break;
    Ray e = grid_ray(cam, f, x, y);
    Spill none; none.pool = nullptr; none.count = nullptr;
    constexpr int LEVELS = REFLECT ? 4 : 1;
    const SoftArgs soft = SOFT ? soft_args(f, x, y) : SoftArgs{0, 0.0, 0.0};
    uint32_t level_clr[LEVELS];
    double level_ks[LEVELS];
    int deepest = 0;
    double ktotal = 1.0;
    bool alive = true;
#pragma unroll
    for (int level = 0; level < LEVELS; level++) {
        level_clr[level] = 0u; level_ks[level] = 0.0;
        if (!alive) continue;
        deepest = level;
        Closest c;
        closest_hit<false, ENV>(s, e, true, rid, f.n_rays, nullptr, nullptr, none, c, n_eval, n_over);
        uint32_t sum = 0u; // main.cpp:83: no hit -> Color(0,0,0); no lights -> clr_sum = (0,0,0), shader.h:49,95
        d3 ph = D3(0, 0, 0), nh = D3(0, 0, 0);
        int obj = (SOFT && level == 0 && soft.obj0 < s.n_objects) ? soft.obj0 : 0; // what a ray that misses goes on with (main.cpp:94)
        if (c.prim_best >= 0) {
            if (level == 0) n_hit = 1;
            const HitRec h = primary_record<ENV>(s, sh, e, c);
            ph = h.ph; nh = h.nh; obj = h.obj;
            if (h.emissive) sum = h.cm0;
            for (int k = 0; !h.emissive && k < sh.n_lights; k++)
                sum = shade_light<SOFT, true>(sh, k, h, sum, [&](const Ray &sr) {
                    Closest o;
                    closest_hit<false, ENV>(s, sr, true, rid, f.n_rays, nullptr, nullptr, none, o, n_eval, n_over);
                    return o.t_best;
                }, n_shadow, level == 0 ? soft.m : 0.0, level == 0 ? soft.n : 0.0);
        }
        level_clr[level] = sum;
        if (REFLECT) {
            const double ks = s.n_objects > 0 ? s.objects[obj].ks : 0.0;      // main.cpp:94
            level_ks[level] = ks;
            ktotal *= ks;                                                      // :97
            alive = (level + 1 <= 3) && (ktotal >= 0.05) && s.n_objects > 0;   // :98-103: reflection_times = level + 1
            if (alive && level + 1 < LEVELS) {
                e = mkview(ph, vsub(e.npe, vmul(vmul(nh, 2), dot(nh, e.npe)))); // :104-108
                n_secondary += 1;
            }
        }
    }
    uint32_t clr = 0u; // clr * (1 - ks) + CalcSubPixel(reflected ray) * ks, from the deepest ray back (main.cpp:109)
#pragma unroll
    for (int level = LEVELS - 1; level >= 0; level--) {
        if (level == deepest) clr = level_clr[level];
        else if (level < deepest) clr = cadd(cmul(level_clr[level], 1 - level_ks[level]), cmul(clr, level_ks[level]));
    }
    pixel = clr;
    } while (0);
    direct_store(pixmap, f, w, x, y, valid, pixel);
    block_count<5>(w.counters, {4, 0, 1, 2, 6}, {n_hit, n_shadow, n_eval, n_over, REFLECT ? n_secondary : 0u});
}

// ---- refraction -----------------------------------------------------------------------------------------
// CalcSubPixel with objects of fresnel < 1 (main.cpp:116-154).  A call with outside >= 1 shades its closest hit,
// then casts the mirror ray while the thresholds of main.cpp:103 hold and -- whatever they say -- the
// refracted ray INTO the object hit if that has fresnel < 1; its colour is
//     (clr * (1 - ks) + mirror * ks) * fresnel + refracted * (1 - fresnel)            main.cpp:109,127
// with 8-bit truncation after every product, so the terms cannot be reordered.  A call with outside <= -1
// intersects only the object it is inside of, with `outside` as given (main.cpp:55), and returns what its one
// child returns (main.cpp:150: the shading it evaluates at :88 is overwritten, so it is skipped here, shadow
// rays included): the refracted ray out when (IOR c)^2 - IOR^2 + 1 > 0, else the mirrored ray inside with
// outside counted down, and after -5 a call with outside = 1 from where the ray is (main.cpp:144-148).
// The recursion is run as a loop: a node that waits for a child leaves (its colour so far, its object, what
// to do with the child's colour) on a stack of REFRACT_OPS entries; a node with both children also leaves
// what its second child starts from (at most 3 such nodes wait at a time: each raises reflection_times).
// Nothing in the reference bounds the refracted chain -- a ray that leaves a glass object and hits nothing goes
// on from the origin with that object's coefficients for ever (main.cpp:94,116) and overflows its stack --
// so a pixel that fills the stack raises counter 14 and the frame is reported as failed.
#define REFRACT_OPS 256
// lanes that share the shadow walks of a listed ray's nodes (k_render_refract<.., CHILDREN, LG>).  Measured on 10 000
// spheres, a tenth of glass, 8 lights, ms of the second step: 1 lane 2.70, 2 lanes 1.65, 4 lanes 1.62, 8 lanes 1.93 (every
// lane of a group runs the ray's recursion, so wider groups mean more waves for the same trees)
#ifndef REFRACT_GROUP
#define REFRACT_GROUP 4
#endif
#define REFRACT_SPAWNS 4
#define OP_MIRROR 0u  // value = clr * (1 - ks) + child * ks
#define OP_REFRACT 1u // value = clr * fresnel + child * (1 - fresnel)
#define OP_BOTH 2u    // the mirror child first, then the refracted one
struct RefractSpawn { d3 ph, nh, npe; double ktotal; int times, obj; };
// objs[obj]->CalcIntersect(eye, k) for k <= -1: point and normal of the tuple returned (t is not used by the caller)
__device__ __forceinline__ void intersect_inside(const SceneView &s, int obj, const Ray &e, int k, d3 *ph_out, d3 *nh_out) {
    const DObject &o = s.objects[obj];
    d3 ph = D3(0, 0, 0), nh = D3(0, 0, 0);
    if (o.type == RT_OBJ_PLANE) {                                    // anyobject.h:69-87, `outside` as given
        const d3 ni = ld3(o.n);
        if (dot(ni, e.npe) >= 0) {
            nh = vmul(normalize(ni), (double)k);                     // :81
        } else {
            const double t = -dot(ni, vsub(e.pe, ld3(o.p))) / dot(ni, e.npe);
            ph = vadd(e.pe, vmul(e.npe, t));
            nh = vmul(ni, (double)k);                                // :86
        }
    } else if (o.type == RT_OBJ_SPHERE) {                            // :128-158, outside = -1 (:144)
        const d3 c = ld3(o.p);
        const d3 ec = vsub(e.pe, c);
        const double B = dot(e.npe, vsub(c, e.pe));
        const double Cc = dot(ec, ec) - o.radius * o.radius;
        const double delta = B * B - Cc;
        if (delta < 0) {
            nh = vmul(D3(0, 0, 0), -1.0);                            // :151
        } else {
            const double t = B - (-1.0) * sqrt(delta);
            ph = vadd(e.pe, vmul(e.npe, t));
            nh = vmul(normalize(vsub(ph, c)), -1.0);
        }
    } else if (o.type == RT_OBJ_ENVIRONMENT) {                       // :199-220, inside = k unless the origin is outside
        const d3 c = ld3(o.p);
        const d3 ec = vsub(e.pe, c);
        int inside = k;
        if (dot(ec, ec) - o.radius * o.radius >= 0.00001) inside = -1;
        const double B = dot(e.npe, vsub(c, e.pe));
        const double Cc = dot(ec, ec) - o.radius * o.radius;
        const double delta = B * B - Cc;
        if (delta < 0) {
            nh = vmul(D3(0, 0, 0), (double)inside);
        } else {
            const double t = B + sqrt(delta);
            ph = vadd(e.pe, vmul(e.npe, t));
            nh = vmul(normalize(vsub(c, ph)), (double)inside);
        }
    } else if (o.type == RT_OBJ_QUADRIC) {                           // :283-345, `outside` as given
        const DQuadric &q = s.quadrics[o.tri_begin];
        int outside, branch;
        const double t = quadric_t(q, e, &outside, &branch);
        if (branch) {
            ph = vadd(e.pe, vmul(e.npe, t));
            nh = vmul(quadric_normal(q, ph), (double)k);
        }
    }
    // a Mesh: with outside < 0 a triangle reports a hit only at t == 0 exactly (anyobject.h:499), which the
    // mesh then drops (t_min > 0, :659): always the miss tuple, point and normal zero (:664)
    *ph_out = ph; *nh_out = nh;
}
// The closest hit of one ray found by its thread alone through the scene table (the walk k_search_general's lanes
// do, see walk_ray): the candidates go to the thread's own slots of workspace slot 0 -- planes, environments and
// quadrics first, they are candidates of every ray -- and the fp64 tests take them from there.  A ray the table
// cannot answer for (no finite unit direction, an origin far from the scene) is tested against everything.
template <bool ENV>
__device__ __forceinline__ void closest_hit_walk(const SceneView &s, const GeneralTable &t, const Ray &e, uint32_t rid,
                                                 const Work &w, Closest &c, uint32_t &evaluated, uint32_t &overflowed,
                                                 uint32_t &n_pairs) {
    const float ox = (float)e.pe.x, oy = (float)e.pe.y, oz = (float)e.pe.z;
    const float ux = (float)e.npe.x, uy = (float)e.npe.y, uz = (float)e.npe.z;
    const float ul = ux * ux + uy * uy + uz * uz, o_l1 = fabsf(ox) + fabsf(oy) + fabsf(oz);
    if (!(ul > 0.99f && ul < 1.01f) || !(o_l1 < t.far_origin)) {
        closest_hit<false, ENV>(s, e, true, rid, w.n, w.cand, w.cand_cnt, w.spill, c, evaluated, overflowed);
        return;
    }
    uint32_t cnt = 0;
    for (int p = 0; p < t.n_planes; p++) append_id(t.plane_id[p], rid, w.n, w.cand, cnt, w.spill);
    walk_ray(t, s, e, ox, oy, oz, ux, uy, uz, o_l1, rid, w.n, w.cand, cnt, w.spill, n_pairs);
    w.cand_cnt[rid] = cnt;
    closest_hit<false, ENV, true>(s, e, false, rid, w.n, w.cand, w.cand_cnt, w.spill, c, evaluated, overflowed);
}
// (three waves per SIMD: at the two the compiler picks by itself the frame takes 30 % longer, at four the spills cost more)
// WALK: a scene of many primitives -- every ray of the pixel's tree, shadow rays included, finds its candidates
// through the scene table (closest_hit_walk) instead of testing every primitive.
// The rays of a frame whose first call of CalcSubPixel has children -- the object hit (objs[0] after a miss,
// main.cpp:94,116) has ks >= 0.05 (times = 1 and ktotal = ks after the first call, main.cpp:97-103) or is of glass --
// listed for k_render_refract<.., CHILDREN> (the list and its count share the far hits' arrays, free by then).
__global__ __launch_bounds__(256) void k_list_children(SceneView s, Frame f, Work w) {
    const uint32_t rid = blockIdx.x * blockDim.x + threadIdx.x;
    bool has = false;
    if (rid < f.n_rays && s.n_objects > 0) {
        const int obj = w.rec_obj[rid];
        if (obj != -2) {
            const DObject &o = s.objects[obj >= 0 ? obj : 0];
            has = (o.ks >= 0.05) || (o.fresnel < 1);
        }
    }
    const unsigned long long m = __ballot(has);
    if (!m) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(w.far_n, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (has) w.far_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = rid;
}
// CHILDREN: a searched scene with glass in it is rendered in two steps.  The frame's own pipeline (packet search,
// k_resolve_primary, the light groups) does what every pixel needs -- the camera ray's closest hit and its shading
// with searched shadow rays -- and leaves hit record and colour in the workspace; this kernel then runs over the rays
// of the frame (one thread per rid) and, for the few whose first call of CalcSubPixel has children (an object with
// ks or glass at the hit, or objs[0]'s after a miss: main.cpp:94,116), takes the recursion up from there: the first
// call's closest hit and shading are read, not computed, and every later ray of the pixel's tree walks the scene table
// as before.  The other rays keep the colour they have.
// LG > 1 (CHILDREN only): LG neighbouring lanes stand for one listed ray.  All of them run the ray's recursion, in
// step and on the same values; where a node is shaded, the walks of its shadow rays -- most of what a node costs, one
// walk per light that the hit faces -- and the lights' colours are dealt to the lanes of the group (light k to lane
// k mod LG, each lane with workspace slot `lane in group` for its candidates) and the colours handed round with
// shuffles, so that a node takes the time of two walks and two lights instead of one walk plus one walk and one light
// per light.  Lane 0 of a group counts the statistics that are the ray's, every lane those of the shadow rays it cast.
template <bool SOFT, bool WALK, bool CHILDREN = false, int LG = 1>
__global__ __launch_bounds__(256, 3) void k_render_refract(SceneView s, ShaderView sh, DCamera cam, Frame f, Work w_all,
                                                        GeneralTable table, unsigned char *__restrict__ pixmap) {
    uint32_t n_eval = 0, n_over = 0, n_hit = 0, n_shadow = 0, n_secondary = 0, n_deep = 0, n_pairs = 0;
    int x = 0, y = 0;
    uint32_t rid_c = 0u;
    bool listed = false;
    const uint32_t sub = LG > 1 ? (threadIdx.x % (uint32_t)LG) : 0u; // the lane's place in its group
    Work w = w_all;
    if (LG > 1) { // the lane's own candidate slots: workspace slot `sub`
        const size_t n = f.n_rays;
        w.cand = w_all.cand + (size_t)sub * CAND_CAP * n;
        w.cand_cnt = w_all.cand_cnt + (size_t)sub * n;
        w.spill.pool = w_all.spill.pool + (size_t)sub * POOL_STRIDE(n);
        w.spill.count = w_all.spill.count + sub;
    }
    if (CHILDREN) { // the rays k_list_children found, one per thread (or group): full waves, whatever share of the frame they are
        const uint32_t at = (blockIdx.x * blockDim.x + threadIdx.x) / (uint32_t)LG;
        listed = at < *w.far_n;
        if (listed) { rid_c = w.far_list[at]; rid_to_xy(rid_c, f.tiles_x, &x, &y); }
    } else {
        direct_xy(f, &x, &y);
    }
    const bool valid = (!CHILDREN || listed) && x < f.grid_w && y < f.grid_h;
    const uint32_t rid = CHILDREN ? rid_c : ((WALK && valid) ? xy_to_rid(x, y, f.tiles_x) : 0u); // the thread's slots of the per-ray lists
    uint32_t pixel = 0u;
    bool changed = !CHILDREN; // CHILDREN: the ray's colour is written only where the recursion went on
    do {
    if (!valid) break;
    Ray e = grid_ray(cam, f, x, y);
    Spill none; none.pool = nullptr; none.count = nullptr;
    auto closest = [&](const Ray &ray, Closest &out, bool mine = true) { // mine: this lane counts the walk (LG > 1: the ray's own walks are lane 0's)
        uint32_t ev = 0, ov = 0, np = 0;
        if (WALK) closest_hit_walk<true>(s, table, ray, rid, w, out, ev, ov, np);
        else closest_hit(s, ray, true, rid, f.n_rays, nullptr, nullptr, none, out, ev, ov);
        if (mine) { n_eval += ev; n_over += ov; n_pairs += np; }
    };
    uint32_t op_clr[REFRACT_OPS], op_what[REFRACT_OPS]; // what = object | kind << 30
    RefractSpawn spawn[REFRACT_SPAWNS];
    int n_ops = 0, n_spawn = 0;
    const SoftArgs soft = SOFT ? soft_args(f, x, y) : SoftArgs{0, 0.0, 0.0};
    int times = 0, outside = 1, oi = (SOFT && soft.obj0 < s.n_objects) ? soft.obj0 : 0;
    double ktotal = 1.0;
    uint32_t value = 0u;
    bool first = true;
    for (;;) {
        // ---- one call of CalcSubPixel, down to its first child
        d3 ph = D3(0, 0, 0), nh = D3(0, 0, 0);
        uint32_t own = 0u;
        if (s.n_objects == 0) { value = 0u; break; } // (the reference would read objs[0] of an empty vector)
        if (CHILDREN && first) { // the camera ray's call: its hit and its colour are there already
            const size_t n = f.n_rays;
            const int obj0 = w.rec_obj[rid];
            if (obj0 >= 0) {
                ph = D3(w.rec_d[0 * n + rid], w.rec_d[1 * n + rid], w.rec_d[2 * n + rid]);
                nh = D3(w.rec_d[3 * n + rid], w.rec_d[4 * n + rid], w.rec_d[5 * n + rid]);
                oi = obj0;
            }
            own = w.rec_sum[rid];
            changed = true;
        } else if (outside <= -1) {
            intersect_inside(s, oi, e, outside, &ph, &nh);
        } else {
            Closest c;
            closest(e, c, sub == 0);
            if (c.prim_best >= 0) {
                if (first) n_hit = 1;
                const HitRec h = primary_record(s, sh, e, c);
                ph = h.ph; nh = h.nh; oi = h.obj;
                if (h.emissive) own = h.cm0;
                if (LG > 1) {
                    // LG lights at a time: this lane makes light k0 + sub's colour, its shadow ray's walk included, then
                    // the group takes the LG colours into the running mean in the lights' order, each read from its lane
                    const uint32_t lane = threadIdx.x & 63u, group0 = lane - sub;
                    for (int k0 = 0; !h.emissive && k0 < sh.n_lights; k0 += LG) {
                        uint32_t my_clr = 0u;
                        const int km = k0 + (int)sub;
                        if (km < sh.n_lights)
                            my_clr = light_colour<SOFT, true>(sh, km, h, [&](const Ray &sr) {
                                Closest o;
                                closest(sr, o);
                                return o.t_best;
                            }, n_shadow);
                        for (int j = 0; j < LG && k0 + j < sh.n_lights; j++)
                            own = light_mean(own, (uint32_t)__shfl((int)my_clr, (int)(group0 + (uint32_t)j), 64), k0 + j);
                    }
                } else {
                    for (int k = 0; !h.emissive && k < sh.n_lights; k++)
                        own = shade_light<SOFT, true>(sh, k, h, own, [&](const Ray &sr) {
                            Closest o;
                            closest(sr, o);
                            return o.t_best;
                        }, n_shadow, first ? soft.m : 0.0, first ? soft.n : 0.0);
                }
            }
        }
        first = false;
        const DObject &o = s.objects[oi];
        const double ks = o.ks, fresnel = o.fresnel;                               // main.cpp:94,116
        ktotal *= ks;                                                              // :97
        bool to_child = false;
        if (outside >= 1) {
            times += 1;                                                            // :98-100
            const bool mirror = times <= 3 && ktotal >= 0.05;                      // :103
            const bool glass = fresnel < 1;                                        // :117
            if (mirror || glass) {
                if (n_ops == REFRACT_OPS || (mirror && glass && n_spawn == REFRACT_SPAWNS)) { n_deep = 1; value = 0u; n_ops = 0; break; }
                op_clr[n_ops] = own;
                op_what[n_ops] = (uint32_t)oi | ((mirror && glass) ? OP_BOTH : (mirror ? OP_MIRROR : OP_REFRACT)) << 30;
                n_ops += 1;
                n_secondary += 1;
                if (mirror) {
                    if (glass) {
                        RefractSpawn &sp = spawn[n_spawn++];
                        sp.ph = ph; sp.nh = nh; sp.npe = e.npe; sp.ktotal = ktotal; sp.times = times; sp.obj = oi;
                    }
                    e = mkview(ph, vsub(e.npe, vmul(vmul(nh, 2), dot(nh, e.npe)))); // :104-108
                    outside = 1; oi = 0;                                           // :109: object_index defaults to 0
                } else {
                    const d3 V = vmul(e.npe, -1);                                  // :119-126
                    const double c = dot(nh, V);
                    const double IOR = 1 / o.ior;
                    d3 T = vadd(vmul(V, -IOR), vmul(nh, IOR * c - sqrt((IOR * c) * (IOR * c) - IOR * IOR + 1)));
                    e = mkview(ph, normalize(T));
                    outside = -1;
                }
                to_child = true;
            } else {
                value = own;                                                       // :111,153
            }
        } else {
            if (!(fresnel < 1)) { n_deep = 1; value = 0u; n_ops = 0; break; }      // (a ray is only ever inside a glass object)
            const d3 V = vmul(e.npe, -1);                                          // :129-150
            const double c = dot(nh, V);
            const double IOR = o.ior;
            const double delta = (IOR * c) * (IOR * c) - IOR * IOR + 1;
            n_secondary += 1;
            if (delta > 0) {
                d3 T = vadd(vmul(V, -IOR), vmul(nh, IOR * c - sqrt(delta)));
                e = mkview(ph, normalize(T));
                outside = 1;
            } else {
                e = mkview(ph, vsub(e.npe, vmul(vmul(nh, 2), dot(nh, e.npe))));
                outside = (outside <= -5) ? 1 : outside - 1;                       // :144-146
            }
            to_child = true;
        }
        if (to_child) continue;
        // ---- a call has returned `value`: hand it to the callers that wait
        bool resumed = false;
        while (n_ops > 0) {
            const uint32_t what = op_what[n_ops - 1], kind = what >> 30;
            const int pobj = (int)(what & 0x3fffffffu);
            const DObject &po = s.objects[pobj];
            const uint32_t pclr = op_clr[n_ops - 1];
            if (kind == OP_REFRACT) {
                value = cadd(cmul(pclr, po.fresnel), cmul(value, 1 - po.fresnel)); // :127
                n_ops -= 1;
            } else {
                value = cadd(cmul(pclr, 1 - po.ks), cmul(value, po.ks));           // :109
                if (kind == OP_MIRROR) { n_ops -= 1; continue; }
                // the second child: the refracted ray, from what the node left behind
                op_clr[n_ops - 1] = value;
                op_what[n_ops - 1] = (uint32_t)pobj | OP_REFRACT << 30;
                const RefractSpawn &sp = spawn[--n_spawn];
                const d3 V = vmul(sp.npe, -1);
                const double c = dot(sp.nh, V);
                const double IOR = 1 / po.ior;
                d3 T = vadd(vmul(V, -IOR), vmul(sp.nh, IOR * c - sqrt((IOR * c) * (IOR * c) - IOR * IOR + 1)));
                e = mkview(sp.ph, normalize(T));
                outside = -1; oi = sp.obj; times = sp.times; ktotal = sp.ktotal;
                n_secondary += 1;
                resumed = true;
                break;
            }
        }
        if (!resumed) break;
    }
    pixel = value;
    } while (0);
    if (CHILDREN) { if (valid && changed && sub == 0) w.rec_sum[rid] = pixel; } // (k_average writes the pixmap)
    else direct_store(pixmap, f, w, x, y, valid, pixel);
    if (LG > 1 && sub != 0) { n_hit = 0; n_secondary = 0; n_deep = 0; } // (the ray's own: counted by lane 0 of its group)
    block_count<7>(w.counters, {4, 0, 1, 2, 6, 14, 5}, {n_hit, n_shadow, n_eval, n_over, n_secondary, n_deep, WALK ? n_pairs : 0u});
}

// CalcSubPixel2 (main.cpp:163-209), the renderer HEAD of the reference selects: the closest hit as in
// CalcSubPixel, coloured by one of the gather shaders -- shading3 (colour bleeding, shader.h:143-191) or
// shading2 (ambient occlusion, shader.h:100-140).  Both turn the scene's direction set by two random angles
// (about z, then about x), skip the directions with ray . nh <= 0 and cast the others from the hit point
// against every object, taking hits with t > 0 only.  cm0 is the object's own colour as CalcIntersect
// returns it: no texture lookup, lights, border or reflection on this path.  The angles come from
// rt355x_rotation in place of rand(), their sine and cosine from sincos_rounded.
struct Gather { const double *dirs; int32_t n; };
// FEW > 0: the scene has at most FEW planes, at most FEW spheres and no mesh (the reference's own scene is
// two planes and two spheres).  All gather rays of a hit share their origin, so the terms of
// Plane::CalcIntersect and Sphere::CalcIntersect that depend on the origin alone -- ni . (pe - pi), pe - c,
// |pe - c|^2 - r^2 and the inside / outside decision -- are evaluated once per hit and kept in registers,
// and the loops over the objects unroll; the values are the ones the reference recomputes for every ray.
#define GATHER_FEW 4
template <bool BLEED, int FEW>
__global__ __launch_bounds__(256) void k_render_gather(SceneView s, ShaderView sh, DCamera cam, Frame f, Work w,
                                                       Gather ga, unsigned char *__restrict__ pixmap) {
    const uint32_t rid = 0; // (no per-ray lists on this path)
    uint32_t n_eval = 0, n_over = 0, n_hit = 0, n_gather = 0;
    int x, y;
    direct_xy(f, &x, &y);
    const bool valid = x < f.grid_w && y < f.grid_h;
    uint32_t pixel = 0u;
    do {
    if (!valid) break;
    const Ray e = grid_ray(cam, f, x, y);
    Spill none; none.pool = nullptr; none.count = nullptr;
    Closest c;
    closest_hit(s, e, true, rid, f.n_rays, nullptr, nullptr, none, c, n_eval, n_over);
    uint32_t clr = 0u; // main.cpp:167,204: no hit -> Color(0,0,0)
    if (c.prim_best >= 0) {
        n_hit = 1;
        const HitGeo h = hit_geometry(s, e, c);
        const uint32_t cm0 = s.objects[h.obj].rgb;
        // the sub-sample this hit belongs to keys the two random numbers
        int px = x, py = frame_row(f, y), sample = 0;
        if (f.alias > 0) {
            px = x / f.alias; py = frame_row(f, y / f.alias);
            sample = (y % f.alias) * f.alias + x % f.alias;
        }
        const double theta1 = rt355x_rotation(f.seed, px, py, sample, 0) * 2 * RT_PI; // shader.h:106-109,149-152
        const double theta2 = rt355x_rotation(f.seed, px, py, sample, 1) * 2 * RT_PI;
        double s1, c1, s2, c2;
        sincos_rounded(theta1, &s1, &c1);
        sincos_rounded(theta2, &s2, &c2);
        // origin terms of the few objects (see above); index 0 of the arrays exists even when FEW == 0
        constexpr int NF = FEW > 0 ? FEW : 1;
        d3 plane_n[NF], sphere_cpe[NF];
        double plane_num[NF], sphere_cc[NF], sphere_out[NF];
        int plane_prim[NF], sphere_prim[NF];
        if (FEW > 0) {
#pragma unroll
            for (int k = 0; k < NF; k++) {
                plane_prim[k] = -1; plane_n[k] = D3(0, 0, 0); plane_num[k] = 0;
                if (k < s.n_planes) {
                    plane_prim[k] = s.kind_prims[k];
                    const DObject &o = s.objects[s.prim_obj[plane_prim[k]]];
                    plane_n[k] = ld3(o.n);
                    plane_num[k] = -dot(plane_n[k], vsub(h.ph, ld3(o.p)));                // anyobject.h:84
                }
                sphere_prim[k] = -1; sphere_cpe[k] = D3(0, 0, 0); sphere_cc[k] = 0; sphere_out[k] = 1;
                if (k < s.n_spheres) {
                    sphere_prim[k] = s.kind_prims[s.n_planes + k];
                    const double4 q = s.prim_rec[sphere_prim[k]];
                    const d3 ctr = D3(q.x, q.y, q.z);
                    const d3 ec = vsub(h.ph, ctr);
                    sphere_cc[k] = dot(ec, ec) - q.w * q.w;                               // anyobject.h:137,146
                    sphere_out[k] = (sphere_cc[k] <= -0.00001) ? -1.0 : 1.0;
                    sphere_cpe[k] = vsub(ctr, h.ph);
                }
            }
        }
        const int len = ga.n;
        uint32_t cnt = 0, sum_r = 0, sum_g = 0, sum_b = 0;
        // The normal test ray . nh <= 0 (shader.h:122,167) drops about half of the directions, a different
        // half in every lane.  So the directions are taken 128 at a time: first every lane marks the ones
        // that may pass -- g . m with m = the normal turned back by the two rotations is ray . nh up to
        // rounding, so anything above -eps is marked -- then the lanes work through their own marks together,
        // and the reference's own test on the reference's own rotated ray decides.  The order in which a
        // lane meets its directions does not matter: the results are counts and sums of 8-bit colours.
        const double rot_a = h.nh.y * c2 + h.nh.z * s2;
        const double mx = h.nh.x * c1 + rot_a * s1, my = rot_a * c1 - h.nh.x * s1, mz = h.nh.z * c2 - h.nh.y * s2;
        const double eps = 1e-13 * (fabs(h.nh.x) + fabs(h.nh.y) + fabs(h.nh.z));
        for (int base = 0; base < len; base += 128) { // uniform
            const int n_low = min(64, len - base), n_high = min(128, len - base) - 64;
            unsigned long long marks0 = 0ull, marks1 = 0ull;
            for (int j = 0; j < n_low; j++) {
                const double *g = ga.dirs + 3 * (size_t)(base + j);
                const double d = __builtin_fma(g[0], mx, __builtin_fma(g[1], my, g[2] * mz));
                marks0 |= (unsigned long long)(!(d <= -eps)) << j;
            }
            for (int j = 0; j < n_high; j++) {
                const double *g = ga.dirs + 3 * (size_t)(base + 64 + j);
                const double d = __builtin_fma(g[0], mx, __builtin_fma(g[1], my, g[2] * mz));
                marks1 |= (unsigned long long)(!(d <= -eps)) << j;
            }
            auto pop = [&]() -> int { // this lane's next marked direction, -1 when it has none left
                if (marks0) { const int b = __ffsll((long long)marks0) - 1; marks0 &= marks0 - 1; return base + b; }
                if (marks1) { const int b = __ffsll((long long)marks1) - 1; marks1 &= marks1 - 1; return base + 64 + b; }
                return -1;
            };
            auto fetch = [&](int i) -> d3 {
                if (i < 0) return D3(0, 0, 0);
                const double *g = ga.dirs + 3 * (size_t)i;
                return D3(g[0], g[1], g[2]);
            };
            int i = pop();
            d3 g = fetch(i);
            while (i >= 0) {
                const int i_next = pop();       // its direction is on the way while this one is worked on
                const d3 g_next = fetch(i_next);
                const double x1 = c1 * g.x - s1 * g.y, y1 = s1 * g.x + c1 * g.y;          // shader.h:115,160
                const d3 ray = D3(x1, c2 * y1 - s2 * g.z, s2 * y1 + c2 * g.z);            // :119,164
                if (!(dot(ray, h.nh) <= 0)) {                                             // :122,167
                    n_gather += 1;
                    const Ray vw = mkview(h.ph, ray);                                     // :127,173
                    double t_hit = RT_FLT_MAX; int prim_hit = -1;                         // t > 0, :130,176
                    if (FEW > 0) {
                        auto consider = [&](double t, int prim) {
                            if (t > 0 && (t < t_hit || (t == t_hit && prim < prim_hit))) { t_hit = t; prim_hit = prim; }
                        };
#pragma unroll
                        for (int k = 0; k < NF; k++) {
                            if (k < s.n_planes) {                                         // anyobject.h:79-84
                                const double dn = dot(plane_n[k], vw.npe);
                                consider((dn >= 0) ? -1.0 : plane_num[k] / dn, plane_prim[k]);
                            }
                        }
#pragma unroll
                        for (int k = 0; k < NF; k++) {
                            if (k < s.n_spheres) {                                        // anyobject.h:144-154
                                const double B = dot(vw.npe, sphere_cpe[k]);
                                const double delta = B * B - sphere_cc[k];
                                consider((delta < 0) ? -1.0 : B - sphere_out[k] * sqrt(delta), sphere_prim[k]);
                            }
                        }
                    } else {
                        closest_by_kind<true, !BLEED>(s, vw, t_hit, prim_hit);            // shading2: any hit will do, :130-133
                    }
                    n_eval += (uint32_t)s.n_prims;
                    if (BLEED) {
                        if (t_hit < 100) {                                                // :181-184
                            const uint32_t hc = s.objects[s.prim_obj[prim_hit]].rgb;
                            sum_r += cr(hc); sum_g += cg(hc); sum_b += cb(hc);
                            cnt += 2;
                        }
                    } else if (prim_hit >= 0) {
                        cnt += 2;
                    }
                }
                i = i_next; g = g_next;
            }
        }
        const double part = (double)(int)cnt * 1.0 / (double)len;                         // :137,188
        uint32_t other = sh.ambient;                                                      // cm1 of shading2, main.cpp:205
        if (BLEED) {
            const double n = (double)(int)cnt;                                            // :187; 0 / 0 -> colour 0
            other = RGB(q8((double)sum_r / n), q8((double)sum_g / n), q8((double)sum_b / n));
        }
        clr = cadd(cmul(cm0, 1 - part), cmul(other, part));
    }
    pixel = clr;
    } while (0);
    direct_store(pixmap, f, w, x, y, valid, pixel);
    block_count<3>(w.counters, {4, 1, 7}, {n_hit, n_eval, n_gather});
}

// ---- the gather shaders on searched scenes ----------------------------------------------------------------
// The camera rays' hits are found as usual (k_resolve_primary writes the hit records).  k_gather_init gives
// every hit its two angles and marks the directions of the set that, turned by them, pass the normal test;
// k_gather_walk sends those rays through the scene table and adds what they hit to the hit's counts and colour
// sums; k_gather_finish blends the pixel.  The per-hit state lives in the arrays the levels of mirror rays use:
// lvl_ks[0..3] = sin and cos of the two angles, lvl_clr[0] = cnt, lvl_clr[1..3] = the colour sums -- or, for a set
// of at most GATHER_PACKED_DIRS directions (each sum then fits 16 bits), all four in one 64-bit word per hit, the first
// 2 n words of lvl_clr read as n such words: cnt << 48 | red << 32 | green << 16 | blue, so that a gather ray adds its
// hit with ONE atomic instead of four (every device-wide atomic is a transaction with memory of its own).
#define GATHER_PACKED_DIRS 256
__device__ __forceinline__ d3 gather_ray(const Gather &ga, int i, double s1, double c1, double s2, double c2) {
    const double gx = ga.dirs[3 * i], gy = ga.dirs[3 * i + 1], gz = ga.dirs[3 * i + 2];
    const double x1 = c1 * gx - s1 * gy, y1 = s1 * gx + c1 * gy;                          // shader.h:115,160
    return D3(x1, c2 * y1 - s2 * gz, s2 * y1 + c2 * gz);                                  // :119,164
}
__global__ __launch_bounds__(256) void k_gather_init(Frame f, Work w, Gather ga) {
    const uint32_t rid = blockIdx.x * blockDim.x + threadIdx.x;
    if (rid >= f.n_rays) return;
    const size_t n = f.n_rays;
    const int words = (ga.n + 63) / 64;
    if (w.rec_obj[rid] < 0) {
        for (int k = 0; k < words; k++) w.g_live[(size_t)k * n + rid] = 0ull;
        return;
    }
    int x, y;
    rid_to_xy(rid, f.tiles_x, &x, &y);
    int px = x, py = frame_row(f, y), sample = 0; // the sub-sample this hit belongs to keys the two random numbers
    if (f.alias > 0) {
        px = x / f.alias; py = frame_row(f, y / f.alias);
        sample = (y % f.alias) * f.alias + x % f.alias;
    }
    const double theta1 = rt355x_rotation(f.seed, px, py, sample, 0) * 2 * RT_PI; // shader.h:106-109,149-152
    const double theta2 = rt355x_rotation(f.seed, px, py, sample, 1) * 2 * RT_PI;
    double s1, c1, s2, c2;
    sincos_rounded(theta1, &s1, &c1);
    sincos_rounded(theta2, &s2, &c2);
    w.lvl_ks[0 * n + rid] = s1; w.lvl_ks[1 * n + rid] = c1; w.lvl_ks[2 * n + rid] = s2; w.lvl_ks[3 * n + rid] = c2;
    if (ga.n <= GATHER_PACKED_DIRS) ((unsigned long long *)w.lvl_clr)[rid] = 0ull; // (shading2 counts in its low half)
    else { w.lvl_clr[0 * n + rid] = 0u; w.lvl_clr[1 * n + rid] = 0u; w.lvl_clr[2 * n + rid] = 0u; w.lvl_clr[3 * n + rid] = 0u; }
    const d3 nh = D3(w.rec_d[3 * n + rid], w.rec_d[4 * n + rid], w.rec_d[5 * n + rid]);
    unsigned long long live = 0ull;
    for (int i = 0; i < ga.n; i++) {
        if (!(dot(gather_ray(ga, i, s1, c1, s2, c2), nh) <= 0)) live |= 1ull << (i & 63);  // :122,167
        if ((i & 63) == 63 || i == ga.n - 1) {
            w.g_live[(size_t)(i >> 6) * n + rid] = live;
            live = 0ull;
        }
    }
}
// All rays in one launch, by waves that stay.  The (direction, hit) pairs are taken 64 consecutive hits of one
// direction at a time, chunk k by wave k mod (number of waves); a lane that is free looks at the next pair, and if
// the hit sends a ray in that direction builds the ray and walks the scene table with it (walk_ray's loop, one
// turn at a time), keeping the closest hit by the fp64 formulas as it goes instead of listing candidates.  Rays of
// one wave end their walks at very different times (a third of the lanes' turns would be used if every wave took
// 64 rays and waited for the last, and a launch per direction would wait for the longest walk of each), so free
// lanes are handed new pairs whenever WALK_REFILL of them are free.  A finished ray adds to its hit's counts with
// atomics: integer sums, in any order.  (Taking the rays by the part of the sphere they point into, so that the
// lanes of a wave walk the same part of the table, was measured and gained nothing: the walk is bound by
// instruction issue, not by its loads.)
// The closest hit is the smallest (t, primitive) over t > 0, as closest_hit<POSITIVE> has it; a triangle hit at
// exactly t = 0 (its mesh is removed, and the walk may have passed over what lies behind) sends the ray to the
// walk over everything.  !BLEED only asks whether there is a hit (shader.h:130-133): after the first one only
// what the ray's origin lies in is still looked at, for such a triangle.
#ifndef WALK_REFILL
#define WALK_REFILL 24
#endif
#ifndef WALK_TURNS
#define WALK_TURNS 3
#endif
#ifndef WALK_HOLD
#define WALK_HOLD 8u // leaves a lane can hold
#endif
#ifndef WALK_TEST_MIN
#define WALK_TEST_MIN 40u
#endif
#ifndef WALK_WAIT_MAX
#define WALK_WAIT_MAX 24u
#endif
#ifndef WALK_WAVES
#define WALK_WAVES 4 // waves per SIMD (128 registers): at 3 the frame takes a quarter longer, at 5 the spills cost more
#endif
template <bool BLEED>
__global__ __launch_bounds__(256, WALK_WAVES) void k_gather_walk(GeneralTable t, SceneView s, Frame f, Work w, Gather ga) {
    const int lane = threadIdx.x & 63;
    const size_t n = f.n_rays;
    const int last = t.n_lvls - 1;
    const uint32_t per_pass = (f.n_rays + 63u) / 64u, n_chunks = (uint32_t)ga.n * per_pass, n_waves = gridDim.x * 4u;
    // the wave's place in the pairs (uniform)
    uint32_t chunk = blockIdx.x * 4u + (threadIdx.x >> 6);
    unsigned long long avail = 0ull; // the pairs of the chunk that send a ray and no lane has taken yet
    bool avail_known = false;
    // a lane's ray and where its walk stands
    bool has = false;
    uint32_t rid = 0u, base = 0u, m = 0u, n_pairs = 0u, n_exact = 0u, n_sent = 0u;
    int lvl = 0, bprim = -1;
    bool zero_tri = false;
    __shared__ uint32_t held_leaf[WALK_HOLD][256]; // the leaves whose boxes a lane's ray enters, waiting for their tests
    uint32_t held = 0u;
    unsigned long long left = 0ull;
    Ray e;
    e.pe = D3(0, 0, 0); e.npe = D3(0, 0, 0);
    float ox = 0, oy = 0, oz = 0, ux = 0, uy = 0, uz = 0, o_l1 = 0, t_limit = 0;
    double bt = RT_FLT_MAX;
    BoxRay br = box_ray(0, 0, 0, 1, 0, 0, 0);
    auto consider = [&](double tc, int prim) {
        if (tc > 0 && (tc < bt || (tc == bt && prim < bprim))) {
            bt = tc; bprim = prim;
            t_limit = BLEED ? (tc < 1e38 ? __double2float_ru(tc) * 1.000001f : 3.0e38f) : 0.0f;
        }
    };
    auto finish = [&]() { // the ray's closest hit is known (or must be asked of everything): add it to the hit's sums
        if (zero_tri) {
            closest_by_kind<true, !BLEED>(s, e, bt, bprim);
            n_exact += (uint32_t)s.n_prims;
        }
        if (BLEED) {
            if (bt < 100) {                                                               // shader.h:181-184
                const uint32_t hc = s.objects[s.prim_obj[bprim]].rgb;
                if (ga.n <= GATHER_PACKED_DIRS) { // uniform
                    atomicAdd((unsigned long long *)w.lvl_clr + rid, (2ull << 48) | ((unsigned long long)cr(hc) << 32) | ((unsigned long long)cg(hc) << 16) | (unsigned long long)cb(hc));
                } else {
                    atomicAdd(&w.lvl_clr[1 * n + rid], cr(hc)); atomicAdd(&w.lvl_clr[2 * n + rid], cg(hc)); atomicAdd(&w.lvl_clr[3 * n + rid], cb(hc));
                    atomicAdd(&w.lvl_clr[0 * n + rid], 2u);
                }
            }
        } else if (bprim >= 0) {                                                          // :130-133
            atomicAdd(ga.n <= GATHER_PACKED_DIRS ? &w.lvl_clr[2 * (size_t)rid] : &w.lvl_clr[0 * n + rid], 2u);
        }
        has = false;
    };
    while (true) {
        unsigned long long idle = __ballot(!has);
        while (chunk < n_chunks && (uint32_t)__popcll(idle) >= WALK_REFILL) { // uniform
            // First every free lane is given a pair that sends a ray, as long as there are any: the k-th free lane
            // takes the k-th such pair of the chunk (the pairs' numbers are brought to lanes 0, 1, 2 ... by one
            // permutation of the wave, and fetched from there), and the next chunk is opened when this one has none
            // left.  Only then are the rays built, by all the lanes that got one at once: building one takes
            // hundreds of fp64 instructions, which the whole wave pays for however few of its lanes take part.
            unsigned long long want = idle;
            bool got = false;
            uint32_t got_rid = 0u, got_dir = 0u;
            while (want != 0ull && chunk < n_chunks) { // uniform
                const uint32_t pass = chunk / per_pass, first = (chunk - pass * per_pass) * 64u;
                if (!avail_known) {
                    const uint32_t r = first + (uint32_t)lane;
                    avail = __ballot(r < f.n_rays && ((w.g_live[(size_t)(pass >> 6) * n + r] >> (pass & 63u)) & 1ull));
                    avail_known = true;
                }
                if (avail != 0ull) {
                    const unsigned long long below = (1ull << lane) - 1ull;
                    const bool in_avail = (avail >> lane) & 1ull, wants = (want >> lane) & 1ull;
                    const uint32_t n_avail = (uint32_t)__popcll(avail), ra = (uint32_t)__popcll(avail & below);
                    const uint32_t rw = (uint32_t)__popcll(want & below);
                    // (a permutation of all 64 lanes: the pairs that send a ray first, in order, then the others)
                    const uint32_t dest = in_avail ? ra : n_avail + ((uint32_t)lane - ra);
                    const int packed = __builtin_amdgcn_ds_permute((int)(dest * 4u), lane);
                    const int src = __builtin_amdgcn_ds_bpermute((int)((rw & 63u) * 4u), packed);
                    const uint32_t n_take = min(n_avail, (uint32_t)__popcll(want));
                    if (wants && rw < n_take) { got = true; got_rid = first + (uint32_t)src; got_dir = pass; }
                    avail = __ballot(in_avail && ra >= n_take);
                    want = __ballot(wants && rw >= n_take);
                }
                if (avail == 0ull) { chunk += n_waves; avail_known = false; }
            }
            if (got) {
                const uint32_t r = got_rid;
                const int i = (int)got_dir;
                {
                    rid = r;
                    n_sent++;
                    const d3 ray = gather_ray(ga, i, w.lvl_ks[0 * n + rid], w.lvl_ks[1 * n + rid], w.lvl_ks[2 * n + rid], w.lvl_ks[3 * n + rid]);
                    e = mkview(D3(w.rec_d[0 * n + rid], w.rec_d[1 * n + rid], w.rec_d[2 * n + rid]), ray); // shader.h:127,173
                    ox = (float)e.pe.x; oy = (float)e.pe.y; oz = (float)e.pe.z;
                    ux = (float)e.npe.x; uy = (float)e.npe.y; uz = (float)e.npe.z;
                    const float ul = ux * ux + uy * uy + uz * uz;
                    o_l1 = fabsf(ox) + fabsf(oy) + fabsf(oz);
                    bt = RT_FLT_MAX; bprim = -1; t_limit = 3.0e38f;
                    held = 0u;
                    has = true;
                    if (!(ul > 0.99f && ul < 1.01f) || !(o_l1 < t.far_origin)) {
                        zero_tri = true; // a ray the table cannot answer for: everything is asked
                        finish();
                    } else {
                        zero_tri = false;
                        for (int p = 0; p < t.n_planes; p++) { // planes, environments, quadrics, unbounded triangles: candidates of every ray
                            const uint32_t id = t.plane_id[p];
                            const int prim = (int)(id & CAND_INDEX);
                            const double4 rec = s.prim_rec[prim];
                            double tc;
                            if ((id & CAND_ENVIRONMENT) == CAND_ENVIRONMENT) {
                                int inside;
                                tc = environment_t_rec(rec, e, &inside);
                            } else if (id & CAND_TRI) { // a triangle no bound holds (triangle_is_unbounded)
                                int outside; d3 ph; double u, v;
                                tc = tri_t(s.tris[__double2hiint(rec.x)], e, &outside, &ph, &u, &v);
                                if (tc == 0) zero_tri = true;
                            } else if (__double2hiint(rec.x) == REC_QUADRIC) {
                                int outside, branch;
                                tc = quadric_t(s.quadrics[s.objects[__double2loint(rec.x)].tri_begin], e, &outside, &branch);
                            } else {
                                tc = plane_t(s.objects[__double2loint(rec.x)], e);
                            }
                            consider(tc, prim);
                        }
                        n_exact += (uint32_t)t.n_planes;
                        if (last < 0) {
                            finish();
                        } else {
                            br = box_ray(ox, oy, oz, ux, uy, uz, o_l1);
                            lvl = 0; base = 0u; left = 0ull;
                            m = meets8_box(t.nodes, br, t_limit);
                            n_pairs += 8u;
                        }
                    }
                }
            }
            idle = __ballot(!has);
        }
        if (idle == ~0ull) { // uniform: no lane has a ray
            if (chunk >= n_chunks) break;
            continue;
        }
#pragma unroll 1
        for (int step = 0; step < WALK_TURNS; step++) {
            if (!has || held == WALK_HOLD) continue;
            // A turn is priced by its most expensive lane, the one that enters a node, so every lane should enter
            // one: leaves are put among the held ones and finished groups are left on the way, both in a few
            // instructions -- the way back goes straight to the deepest level that has siblings left.
            auto hold_leaves = [&]() {
                while (m != 0u && held < WALK_HOLD) {
                    held_leaf[held][threadIdx.x] = base + (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    held++;
                }
            };
            if (lvl == last) {
                hold_leaves();
                if (m != 0u) continue; // no place left: on after the tests below
            }
            if (m == 0u) {
                const unsigned long long below = lvl > 0 ? (left & ((1ull << (8 * lvl)) - 1ull)) : 0ull;
                if (below == 0ull) { // the walk is over (a lane that still holds leaves waits for their tests below)
                    lvl = 0;
                    if (held == 0u) finish();
                    continue;
                }
                const int up = (63 - __builtin_clzll(below)) >> 3;
                m = (uint32_t)(below >> (8 * up)) & 0xffu;
                base = (base >> (3 * (lvl - up))) & ~7u;
                lvl = up;
            }
            const uint32_t node = base + (uint32_t)__builtin_ctz(m);
            m &= m - 1u;
            left = (left & ~(0xffull << (8 * lvl))) | ((unsigned long long)m << (8 * lvl));
            lvl++;
            base = node * 8u;
            m = meets8_box(t.nodes + (size_t)(level_offset(lvl) + base) / 8 * 4, br, t_limit);
            n_pairs += 8u;
            if (lvl == last) hold_leaves();
        }
        // The tests of the leaves the lanes hold, for all lanes together: when enough of them hold one, or when
        // lanes that cannot go on without theirs (all places taken, or the walk over) have become many, or when no
        // lane can go on at all.  First the fp32 tests of all held leaves (the sphere, and a triangle's plane), then
        // the fp64 tests of those that passed: a test made by the few lanes that happen to have reached a leaf in
        // this turn would cost the whole wave its loads and its hundreds of fp64 instructions.
        {
            const bool holds = has && held > 0u;
            const bool waits = has && (held == WALK_HOLD || (held > 0u && m == 0u && lvl == 0));
            const uint32_t n_holds = (uint32_t)__popcll(__ballot(holds)), n_waits = (uint32_t)__popcll(__ballot(waits));
            const uint32_t n_moving = (uint32_t)__popcll(__ballot(has && !waits));
            if (n_holds >= WALK_TEST_MIN || n_waits >= WALK_WAIT_MAX || (n_waits > 0u && n_moving == 0u)) { // uniform
                uint32_t passed = 0u;
#pragma unroll 1
                for (uint32_t k = 0; k < WALK_HOLD; k++) {
                    if (k < held) {
                        const uint32_t node = held_leaf[k][threadIdx.x];
                        const float4 q = t.prim[node];
                        if (ray_meets(ox, oy, oz, ux, uy, uz, o_l1, q, t_limit) && // (again: the limit may have come down since)
                            ray_meets_plane_of(ox, oy, oz, ux, uy, uz, o_l1, q, t.prim_n[node])) {
                            held_leaf[passed][threadIdx.x] = node;
                            passed++;
                        }
                    }
                    if (!__any(k + 1 < held)) break; // uniform
                }
                held = passed;
                while (__any(holds && held > 0u)) {
                    if (held > 0u) {
                        held--;
                        const uint32_t id = t.prim_id[held_leaf[held][threadIdx.x]];
                        const double4 rec = s.prim_rec[id & CAND_INDEX];
                        n_exact++;
                        if (id & CAND_TRI) {
                            int outside; d3 ph; double u, v;
                            const double tc = tri_t(s.tris[__double2hiint(rec.x)], e, &outside, &ph, &u, &v);
                            if (tc == 0) zero_tri = true; // also -0.0
                            consider(tc, (int)(id & CAND_INDEX));
                        } else {
                            consider(sphere_t_rec(rec, e), (int)(id & CAND_INDEX));
                        }
                    }
                }
                if (has && m == 0u && lvl == 0) finish(); // the walk was over and only waited for these
            }
        }
    }
    wave_count(w.counters, 5, n_pairs);
    wave_count(w.counters, 1, n_exact);
    wave_count(w.counters, 7, n_sent);
}
template <bool BLEED>
__global__ __launch_bounds__(256) void k_gather_finish(SceneView s, ShaderView sh, Frame f, Work w, Gather ga,
                                                       unsigned char *__restrict__ pixmap) {
    const uint32_t rid = blockIdx.x * blockDim.x + threadIdx.x;
    if (rid >= f.n_rays) return;
    const int obj = w.rec_obj[rid];
    if (obj == -2) return;
    int x, y;
    rid_to_xy(rid, f.tiles_x, &x, &y);
    uint32_t clr = 0u; // main.cpp:167,204
    if (obj >= 0) {
        const size_t n = f.n_rays;
        uint32_t cnt, sr = 0u, sg = 0u, sb = 0u;
        if (BLEED && ga.n <= GATHER_PACKED_DIRS) { // the four sums in one word (see k_gather_init)
            const unsigned long long word = ((const unsigned long long *)w.lvl_clr)[rid];
            cnt = (uint32_t)(word >> 48); sr = (uint32_t)(word >> 32) & 0xffffu; sg = (uint32_t)(word >> 16) & 0xffffu; sb = (uint32_t)word & 0xffffu;
        } else if (ga.n <= GATHER_PACKED_DIRS) { // shading2: the count alone, in the word's low half
            cnt = w.lvl_clr[2 * (size_t)rid];
        } else {
            cnt = w.lvl_clr[0 * n + rid];
            if (BLEED) { sr = w.lvl_clr[1 * n + rid]; sg = w.lvl_clr[2 * n + rid]; sb = w.lvl_clr[3 * n + rid]; }
        }
        const double part = (double)(int)cnt * 1.0 / (double)ga.n;                         // shader.h:137,188
        uint32_t other = sh.ambient;
        if (BLEED) {
            const double m = (double)(int)cnt;                                             // :187
            other = RGB(q8((double)sr / m), q8((double)sg / m), q8((double)sb / m));
        }
        clr = cadd(cmul(s.objects[obj].rgb, 1 - part), cmul(other, part));
    }
    store_pixel(pixmap, f, w, rid, x, y, clr);
}

// LEVELS: the frame is rendered level by level (mirror rays on a searched scene); compiled apart so that frames
// without levels run the kernel they always ran.  ENV: the scene holds an Environment or a quadric (their tests
// are compiled out otherwise).
#ifndef LEVEL_GROUP
#define LEVEL_GROUP 256 // threads of a workgroup that resolves a level after the first: the places it numbers anew (512, 1024: the kernel and the shading after it both slower)
#endif
template <bool LEVELS, bool ENV>
__global__ __launch_bounds__(LEVELS ? LEVEL_GROUP : 256) void k_resolve_primary(SceneView s, ShaderView sh, DCamera cam, Frame f_in, Work w,
                                                         int brute, Group first,
                                                         unsigned char *__restrict__ pixmap) {
    Frame f = f_in;
    if (!LEVELS) f.level = -1;
    const uint32_t rid = blockIdx.x * blockDim.x + threadIdx.x; // (a level after the first: the place in the level's list of rays)
    uint32_t slot = rid;    // where the ray's records go: rid, or on a level after the first its dense index (see Work::dense_rid)
    uint32_t out_rid = rid; // ... and the frame's rid, under which the level's colour is kept
    uint32_t n_eval = 0, n_over = 0, n_hit = 0;
    bool got_hit = false;
    d3 ph = D3(0, 0, 0), nh = D3(0, 0, 0);
    const bool later = LEVELS && f.level > 0; // uniform
    int x = 0, y = 0;
    Ray e;
    // what is kept of a ray whose closest hit is known: the hit record under `slot`, or the miss
    auto record = [&](const Closest &c) {
        if (c.prim_best < 0) { // main.cpp:83: no hit -> Color(0,0,0)
            w.rec_obj[slot] = -1;
            if (sh.n_lights == 0) store_pixel(pixmap, f, w, out_rid, x, y, 0u);
            return;
        }
        n_hit = f.level > 0 ? 0 : 1; // rt355_stats.primary_hits counts camera rays
        const bool environment = ENV && s.objects[s.prim_obj[c.prim_best]].type == RT_OBJ_ENVIRONMENT;
        if (sh.n_lights == 0 && f.level < 0 && !environment) { // Shader::shading with no lights returns clr_sum = (0,0,0), shader.h:49,95
            w.rec_obj[slot] = s.prim_obj[c.prim_best];
            store_pixel(pixmap, f, w, out_rid, x, y, 0u);
            return;
        }
        const HitRec h = primary_record<ENV>(s, sh, e, c);
        if (sh.n_lights == 0) store_pixel(pixmap, f, w, out_rid, x, y, h.emissive ? h.cm0 : 0u); // (levels: the mirror ray still needs ph and nh)
        ph = h.ph; nh = h.nh;
        got_hit = !h.emissive; // an Environment is not lit: no shadow rays (shader.h:70-72)
        const size_t n = f.n_rays;
        w.rec_obj[slot] = h.obj;
        w.rec_d[0 * n + slot] = ph.x; w.rec_d[1 * n + slot] = ph.y; w.rec_d[2 * n + slot] = ph.z;
        w.rec_d[3 * n + slot] = nh.x; w.rec_d[4 * n + slot] = nh.y; w.rec_d[5 * n + slot] = nh.z;
        w.rec_d[6 * n + slot] = h.reh.x; w.rec_d[7 * n + slot] = h.reh.y; w.rec_d[8 * n + slot] = h.reh.z;
        if (!h.no_border) w.rec_d[9 * n + slot] = h.border_w;
        const bool far = !(fabs(ph.x) + fabs(ph.y) + fabs(ph.z) < s.far_origin); // see FAR_SCALE
        if (far && !h.emissive && sh.n_lights) { // (rare: two atomics per such hit)
            atomicAdd(&w.counters[15], 1ull);
            w.far_list[atomicAdd(w.far_n, 1u)] = slot;
        }
        w.rec_cm0[slot] = h.cm0 | (h.no_border ? (1u << 24) : 0u) | (h.emissive ? (1u << 25) : 0u) | (far ? (1u << 26) : 0u);
        w.rec_cm1[slot] = h.cm1;
    };
    if (!later) { // camera rays
        do {
        if (rid >= f.n_rays) break;
        rid_to_xy(rid, f.tiles_x, &x, &y);
        if (!(x < f.grid_w && y < f.grid_h)) {
            w.rec_obj[rid] = -2;
            break;
        }
        e = grid_ray(cam, f, x, y);
        Closest c;
        closest_hit<false, ENV>(s, e, brute != 0, rid, f.n_rays, w.cand, w.cand_cnt, w.spill, c, n_eval, n_over);
        record(c);
        } while (0);
    } else { // the mirror rays of level f.level (k_reflect_rays left them in ray_d), in the order of their compacted list
        const bool active = rid < f.n_rays && rid < w.lvl_live[f.level];
        Closest c;
        c.init();
        if (active) {
            out_rid = w.active_rid[rid];
            const size_t n = f.n_rays;
            e.pe = D3(w.ray_d[0 * n + out_rid], w.ray_d[1 * n + out_rid], w.ray_d[2 * n + out_rid]);
            e.npe = D3(w.ray_d[3 * n + out_rid], w.ray_d[4 * n + out_rid], w.ray_d[5 * n + out_rid]);
            // (their candidates come from walk_ray)
            closest_hit<false, ENV, true>(s, e, brute != 0, rid, f.n_rays, w.cand, w.cand_cnt, w.spill, c, n_eval, n_over);
        }
        // the level's dense numbering: within the workgroup's places, hits from the front and misses from the back
        __shared__ uint32_t wave_hits[LEVEL_GROUP / 64], wave_rays[LEVEL_GROUP / 64];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned long long hits = __ballot(active && c.prim_best >= 0), rays = __ballot(active);
        if (lane == 0) { wave_hits[wave] = (uint32_t)__popcll(hits); wave_rays[wave] = (uint32_t)__popcll(rays); }
        __syncthreads();
        uint32_t hits_before = 0u, misses_before = 0u, in_group = 0u;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) { // uniform
            if (k < wave) { hits_before += wave_hits[k]; misses_before += wave_rays[k] - wave_hits[k]; }
            in_group += wave_rays[k];
        }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (active) {
            const uint32_t group_first = blockIdx.x * blockDim.x;
            slot = c.prim_best >= 0 ? group_first + hits_before + (uint32_t)__popcll(hits & below)
                                    : group_first + in_group - 1u - (misses_before + (uint32_t)__popcll(rays & ~hits & below));
            w.dense_rid[slot] = out_rid;
            record(c);
        }
    }
    const uint32_t n_shadow = emit_shadow_rays(sh, first, w, slot, rid < f.n_rays, got_hit && sh.n_lights, ph, nh, f.soft != 0);
    wave_count(w.counters, 4, n_hit);
    wave_count(w.counters, 0, n_shadow);
    wave_count(w.counters, 1, n_eval);
    wave_count(w.counters, 2, n_over);
}

// The reflection step of CalcSubPixel (main.cpp:93-108) for the rays of level `level`, after their shading:
// ks of the object hit (objs[0]'s after a miss, whose ph and nh are zero: object_index is still 0 at
// main.cpp:94), reflection_ktotal *= ks, reflection_times = level + 1, and while both thresholds hold
// (main.cpp:27-28,103) the mirror ray View(ph, npe - nh * 2 * (nh . npe)), which becomes the ray of level + 1.
__global__ __launch_bounds__(256) void k_reflect_rays(SceneView s, DCamera cam, Frame f, Work w, int level) {
    // level 0: one thread per ray of the frame, every ray's slot-0 record written (NaN where no mirror ray is cast).
    // A later level: one thread per live chain of that level (dense index, see Work::dense_rid); the records were
    // cleared to NaN before the launch and only the chains that go on write theirs.
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t rid = idx; // the frame's rid
    uint32_t n_secondary = 0;
    float4 rec = nan4(); // slot 0's fp32 record: the direction of a mirror ray, NaN where there is none
    bool mine = false;   // this thread stands for a ray of the frame
    do {
    if (idx >= f.n_rays) break;
    const size_t n = f.n_rays;
    int x = 0, y = 0;
    if (level == 0) {
        mine = true;
        rid_to_xy(rid, f.tiles_x, &x, &y);
        if (!(x < f.grid_w && y < f.grid_h)) { w.deepest[rid] = 0xffffffffu; break; }
        w.deepest[rid] = 0u; w.ktotal[rid] = 1.0;
    } else {
        if (idx >= w.lvl_live[level]) break;
        mine = true;
        rid = w.dense_rid[idx];
    }
    Ray e;
    if (level == 0) e = grid_ray(cam, f, x, y);
    else {
        e.pe = D3(w.ray_d[0 * n + rid], w.ray_d[1 * n + rid], w.ray_d[2 * n + rid]);
        e.npe = D3(w.ray_d[3 * n + rid], w.ray_d[4 * n + rid], w.ray_d[5 * n + rid]);
    }
    const int obj = w.rec_obj[idx];
    int miss_obj = 0; // what a ray that misses goes on with: object_index as CalcSubPixel received it
    if (f.soft && level == 0) { const SoftArgs a = soft_args(f, x, y); if (a.obj0 < s.n_objects) miss_obj = a.obj0; }
    const double ks = s.n_objects > 0 ? s.objects[obj >= 0 ? obj : miss_obj].ks : 0.0; // main.cpp:94
    w.lvl_ks[(size_t)level * n + rid] = ks;
    const double ktotal = w.ktotal[rid] * ks;                                    // :97
    w.ktotal[rid] = ktotal;
    if (!((level + 1 <= 3) && (ktotal >= 0.05) && s.n_objects > 0)) break;       // :98-103
    d3 ph = D3(0, 0, 0), nh = D3(0, 0, 0);
    if (obj >= 0) {
        ph = D3(w.rec_d[0 * n + idx], w.rec_d[1 * n + idx], w.rec_d[2 * n + idx]);
        nh = D3(w.rec_d[3 * n + idx], w.rec_d[4 * n + idx], w.rec_d[5 * n + idx]);
    }
    const Ray r = mkview(ph, vsub(e.npe, vmul(vmul(nh, 2), dot(nh, e.npe))));   // :104-108
    w.ray_d[0 * n + rid] = r.pe.x; w.ray_d[1 * n + rid] = r.pe.y; w.ray_d[2 * n + rid] = r.pe.z;
    w.ray_d[3 * n + rid] = r.npe.x; w.ray_d[4 * n + rid] = r.npe.y; w.ray_d[5 * n + rid] = r.npe.z;
    w.deepest[rid] = (uint32_t)level + 1u;
    // (a direction that is not a number still marks the ray as there: the search hands it to the fp64 walk)
    rec = make_float4((float)r.npe.x, (float)r.npe.y, (float)r.npe.z, 0.f);
    if (!(rec.x == rec.x)) rec.x = 0.f;
    n_secondary = 1;
    } while (0);
    if (level == 0 ? idx < f.n_rays : (mine && n_secondary)) w.ray32[rid] = rec;
    wave_count(w.counters, 6, n_secondary);
}

// clr * (1 - ks) + CalcSubPixel(mirror ray) * ks from the deepest level back (main.cpp:109), then the pixel
__global__ __launch_bounds__(256) void k_fold_levels(Frame f, Work w, unsigned char *__restrict__ pixmap) {
    const uint32_t rid = f.rid_base + blockIdx.x * blockDim.x + threadIdx.x;
    if (rid >= f.n_rays) return;
    int x, y;
    rid_to_xy(rid, f.tiles_x, &x, &y);
    if (!(x < f.grid_w && y < f.grid_h)) return;
    const size_t n = f.n_rays;
    const int deepest = (int)w.deepest[rid];
    uint32_t clr = w.lvl_clr[(size_t)deepest * n + rid];
    for (int level = deepest - 1; level >= 0; level--) {
        const double ks = w.lvl_ks[(size_t)level * n + rid];
        clr = cadd(cmul(w.lvl_clr[(size_t)level * n + rid], 1 - ks), cmul(clr, ks));
    }
    store_pixel(pixmap, f, w, rid, x, y, clr); // f.level is -1 here: the frame, or the sub-sample sums of k_average
}

// The shadow rays of the hits far from the scene (bit 26 of their record, see FAR_SCALE) against everything.
// One lane doing that alone in k_shade_group holds its wave for n_prims tests per light, so it is done here,
// a wave taking one such hit at a time with its lanes sharing the primitive list; the closest occluder found
// becomes the ray's whole candidate list for that light, which the shading kernel then walks as any other.
// Frames without such hits leave at the first line.
__global__ __launch_bounds__(256) void k_shade_far(SceneView s, ShaderView sh, Frame f, Work w, Group cur) {
    const uint32_t n_far = *w.far_n;
    if (n_far == 0) return;
    // one (hit, light) pair per wave and turn, whichever wave comes next: far hits lie together on screen (a floor's
    // horizon), and taken by the waves that own them a few waves would run 64 x lights walks each
    const uint32_t n_waves = gridDim.x * (blockDim.x / 64u), n_tasks = n_far * (uint32_t)cur.count;
    for (uint32_t task = blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u; task < n_tasks; task += n_waves) { // uniform in a wave
        const uint32_t rid = w.far_list[task / (uint32_t)cur.count];
        const int g = (int)(task % (uint32_t)cur.count);
        const size_t n = f.n_rays;
        const d3 p = D3(w.rec_d[0 * n + rid], w.rec_d[1 * n + rid], w.rec_d[2 * n + rid]);
        const d3 nrm = D3(w.rec_d[3 * n + rid], w.rec_d[4 * n + rid], w.rec_d[5 * n + rid]);
        const DLight &L = sh.lights[cur.k0 + g];
        const d3 nhl = light_nhl(L, p);
        if (dot(nrm, vmul(nhl, -1)) <= 0) continue;  // no shadow ray (shader.h:396)
        Ray sr; // as shade_light builds it
        if (L.type == 0) { sr.pe = p; sr.npe = ld3(L.shadow_dir); }
        else sr = mkview(p, vmul(nhl, -1));
        const int prim = nearest_occluder_wave(s, sr);
        if ((threadIdx.x & 63u) == 0) { // the list the shading kernel will walk: that primitive, nothing, or everything
            w.cand_cnt[(size_t)g * f.n_rays + rid] = prim >= 0 ? 1u : (prim == -1 ? 0u : CAND_UNSEARCHED);
            if (prim >= 0) w.cand[(size_t)g * CAND_CAP * f.n_rays + rid] = cand_id(s, prim);
        }
    }
}

// Shader::shading's loop over the lights (shader.h:60-94) for the lights of one group: colour before
// the shadow (shader.h:75-89), CalcShadow on the light's candidates (shader.h:393-433), the running
// mean (shader.h:93); then either the next group's shadow rays or the pixel store (main.cpp:367-369).
#ifndef SHADE_MIN_WAVES
#define SHADE_MIN_WAVES 3 // waves per SIMD the shading kernel's register budget allows
#endif
// ENV: the scene holds an Environment or a quadric (their tests are compiled out otherwise).  SOFT: CalcSoftShadow
// (every shadow ray goes through the fp64 walk over everything: `cur` holds no searched bundle).
template <bool ENV, bool SOFT = false>
__global__ __launch_bounds__(256, SHADE_MIN_WAVES) void k_shade_group(SceneView s, ShaderView sh, Frame f, Work w, Group cur,
                                                     Group next, unsigned char *__restrict__ pixmap) {
    uint32_t rid = f.rid_base + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t n_eval = 0, n_over = 0, n_soft = 0;
#ifdef RT355_DEBUG_SURVIVORS
    uint32_t dbg_n[LIGHT_GROUP] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const bool last = (cur.k0 + cur.count == sh.n_lights);
    bool hit = false;
    d3 ph = D3(0, 0, 0), nh = D3(0, 0, 0);
    do {
    if (rid >= f.n_rays) break;
    uint32_t out_rid = rid; // a level after the first is numbered densely (see Work::dense_rid): its colour goes to the frame's rid
    if (f.level > 0) {
        if (rid >= w.lvl_live[f.level]) break;
        out_rid = w.dense_rid[rid];
    }
    int obj = w.rec_obj[rid];
    if (obj == -2) break;
    int x, y;
    rid_to_xy(rid, f.tiles_x, &x, &y); // (level <= 0: the grid point, for the pixel and the soft-shadow strata)
    if (obj < 0) {
        if (last) store_pixel(pixmap, f, w, out_rid, x, y, 0u);
        break;
    }
    const size_t n = f.n_rays;
    HitRec h;
    h.obj = obj;
    h.ph = ph = D3(w.rec_d[0 * n + rid], w.rec_d[1 * n + rid], w.rec_d[2 * n + rid]);
    h.nh = nh = D3(w.rec_d[3 * n + rid], w.rec_d[4 * n + rid], w.rec_d[5 * n + rid]);
    h.reh = D3(w.rec_d[6 * n + rid], w.rec_d[7 * n + rid], w.rec_d[8 * n + rid]);
    const uint32_t cm0w = w.rec_cm0[rid];
    h.cm0 = cm0w & 0xffffffu; h.cm1 = w.rec_cm1[rid];
    h.no_border = (cm0w >> 24) & 1u;
    if ((cm0w >> 25) & 1u) { // an Environment: its colour as it is (shader.h:70-72)
        if (last) store_pixel(pixmap, f, w, out_rid, x, y, h.cm0);
        break;
    }
    h.border_w = h.no_border ? 0.0 : w.rec_d[9 * n + rid];
    uint32_t sum = cur.k0 ? w.rec_sum[rid] : 0u, unused = 0;
    // (the rays of the levels after the camera rays' are cast by the recursion, which passes neither m nor n)
    const SoftArgs soft = (SOFT && f.level <= 0) ? soft_args(f, x, y) : SoftArgs{0, 0.0, 0.0};
    // The candidate count and the first ids of light g + 1 are asked for before light g is shaded: count -> ids ->
    // records -> test is a chain of three dependent loads per light otherwise, and with four waves on a SIMD there
    // is little else to run meanwhile.  (Slots past the count hold stale ids of the workspace: loaded, never used.)
    Pre pre_a, pre_b, pre; // the light after next (count, ids), the next light (+ records), this light
    auto fetch_ids = [&](int g, Pre &dst) {
        dst.n = 0u;
#pragma unroll
        for (int j = 0; j < CAND_BATCH; j++) dst.id[j] = 0u;
        if (SOFT || g >= cur.count || cur.b[g].kind == 0) return; // uniform
        dst.n = w.cand_cnt[(size_t)g * n + rid];
        if (SHADE_PREFETCH < 2) return;
        const uint32_t *cp = w.cand + (size_t)g * CAND_CAP * n + rid;
#pragma unroll
        for (int j = 0; j < CAND_BATCH; j++) dst.id[j] = cp[(size_t)j * n];
    };
    auto fetch_recs = [&](Pre &dst) { // (ids past the count are stale ids of the workspace, or 0: any of them names a record)
#pragma unroll
        for (int j = 0; j < CAND_BATCH; j++) dst.rec[j] = s.prim_rec[min(dst.id[j] & CAND_INDEX, (uint32_t)(s.n_prims - 1))];
    };
    if (SHADE_PREFETCH) fetch_ids(0, pre_b);
    if (SHADE_PREFETCH >= 3) { fetch_ids(1, pre_a); fetch_recs(pre_b); }
    for (int g = 0; g < cur.count; g++) {
        pre = pre_b;
        if (SHADE_PREFETCH >= 3) { pre_b = pre_a; fetch_recs(pre_b); fetch_ids(g + 2, pre_a); }
        else if (SHADE_PREFETCH) fetch_ids(g + 1, pre_b);
        sum = shade_light<SOFT, SHADE_SKIP != 0>(sh, cur.k0 + g, h, sum, [&](const Ray &sr) {
            Spill sp; sp.pool = w.spill.pool + (size_t)g * POOL_STRIDE(n); sp.count = w.spill.count + g;
#ifdef RT355_DEBUG_SURVIVORS
            dbg_n[g] = cur.b[g].kind ? w.cand_cnt[(size_t)g * n + rid] : 0u;
#endif
            const bool brute = SOFT || cur.b[g].kind == 0;
#ifdef RT355_TIMING_NO_OCCLUDER /* timing-only build: what the kernel costs without the candidates' tests (frames differ) */
            if (!brute) return (double)RT_FLT_MAX;
#endif
            return nearest_occluder<ENV>(s, sr, brute, rid, f.n_rays, w.cand + (size_t)g * CAND_CAP * n,
                                    w.cand_cnt + (size_t)g * n, sp, n_eval, n_over, (brute || !SHADE_PREFETCH) ? nullptr : &pre);
        }, SOFT ? n_soft : unused, soft.m, soft.n);
    }
    if (last) {
        store_pixel(pixmap, f, w, out_rid, x, y, sum);
        break;
    }
    w.rec_sum[rid] = sum;
    hit = true;
    } while (0);
    if (!last) {
        const uint32_t n_shadow = emit_shadow_rays(sh, next, w, rid, rid < f.n_rays, hit, ph, nh, SOFT);
        wave_count(w.counters, 0, n_shadow);
    }
    if (SOFT) wave_count(w.counters, 0, n_soft);
#ifdef RT355_DEBUG_SURVIVORS
    for (int g = 0; g < LIGHT_GROUP; g++) { // lane slots the candidate loops occupy vs candidates tested
        uint32_t mx = dbg_n[g];
        for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
        if ((threadIdx.x & 63u) == 0) atomicAdd(&w.counters[12], (unsigned long long)((mx + CAND_BATCH - 1) / CAND_BATCH * CAND_BATCH) * 64ull);
        wave_count(w.counters, 13, dbg_n[g]);
    }
#endif
    wave_count(w.counters, 1, n_eval);
    wave_count(w.counters, 2, n_over);
}

// objs[obj]->CalcIntersect(View(pe,npe)) for n rays: out = t, ph[3], nh[3], uv[2] per ray.
__global__ void k_probe_intersect(SceneView s, int obj, const double *__restrict__ rays, int n,
                                  double *__restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray e = mkview(D3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), D3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]));
    const DObject &o = s.objects[obj];
    double t = -1, uvx = 0, uvy = 0;
    d3 ph = D3(0, 0, 0), nh = D3(0, 0, 0);
    if (o.type == RT_OBJ_PLANE) {
        t = plane_t(o, e);
        if (t != -1) {
            int outside = (dot(ld3(o.n), vsub(e.pe, ld3(o.p))) <= 0) ? -1 : 1;
            ph = vadd(e.pe, vmul(e.npe, t)); nh = vmul(ld3(o.n), (double)outside);
        }
    } else if (o.type == RT_OBJ_SPHERE) {
        int outside;
        t = sphere_t(o, e, &outside);
        if (t != -1) { ph = vadd(e.pe, vmul(e.npe, t)); nh = vmul(normalize(vsub(ph, ld3(o.p))), (double)outside); }
    } else if (o.type == RT_OBJ_ENVIRONMENT) {
        int inside;
        t = environment_t_rec(make_double4(o.p[0], o.p[1], o.p[2], o.radius), e, &inside);
        if (t != -1) { ph = vadd(e.pe, vmul(e.npe, t)); nh = vmul(normalize(vsub(ld3(o.p), ph)), (double)inside); }
    } else if (o.type == RT_OBJ_QUADRIC) {
        const DQuadric &q = s.quadrics[o.tri_begin];
        int outside, branch;
        t = quadric_t(q, e, &outside, &branch);
        if (branch) { ph = vadd(e.pe, vmul(e.npe, t)); nh = vmul(quadric_normal(q, ph), (double)outside); }
    } else {
        double t_min = RT_FLT_MAX; int best = -1;
        for (int q = 0; q < o.tri_count; q++) {
            int outside; d3 p; double u, v;
            double tt = tri_t(s.tris[o.tri_begin + q], e, &outside, &p, &u, &v);
            if (tt >= 0 && tt < t_min) { t_min = tt; best = q; }
        }
        if (t_min > 0 && t_min != RT_FLT_MAX) {
            const DTriangle &T = s.tris[o.tri_begin + best];
            int outside; double u, v;
            t = tri_t(T, e, &outside, &ph, &u, &v);
            nh = vmul(ld3(T.normal), (double)outside);
            double wgt = 1 - u - v;
            uvx = T.uv0[0] * wgt + T.uv1[0] * u + T.uv2[0] * v;
            uvy = T.uv0[1] * wgt + T.uv1[1] * u + T.uv2[1] * v;
        }
    }
    double *q = out + 9 * (size_t)i;
    q[0] = t; q[1] = ph.x; q[2] = ph.y; q[3] = ph.z; q[4] = nh.x; q[5] = nh.y; q[6] = nh.z; q[7] = uvx; q[8] = uvy;
}

// =====================================================================================================
// host side
// =====================================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) { g_err = msg; return code; }
#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (void)hipGetLastError(); /* (reported here: a later call must not find it again) */ \
            return fail(RT355_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                     \
    } while (0)

namespace {
struct h3 { double x, y, z; };
inline h3 H3(const double *p) { return {p[0], p[1], p[2]}; }
inline h3 hsub(h3 a, h3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline h3 hadd(h3 a, h3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline h3 hmul(h3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline h3 hcross(h3 u, h3 v) { return {u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x}; }
inline h3 hnormalize(h3 v) { // vector3D.h:79-88
    double l = sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    if (l != 0) return {v.x / l, v.y / l, v.z / l};
    return v;
}
inline void put3(double *d, h3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline double4 index_rec(int32_t obj, int32_t tri) { // (obj, tri) in the first 8 bytes of a primitive record
    double4 r = make_double4(0, 0, 0, 0);
    int32_t v[2] = {obj, tri};
    memcpy(&r.x, v, sizeof v);
    return r;
}
inline uint32_t pack_rgb(const uint8_t *c) { return (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16); }
} // namespace

// Tuning options of a scene handle (rt355_scene_set_option).  Their initial values come from the environment, read
// once when the handle is made (RT355_<NAME IN CAPITALS>); nothing on the per-frame path reads the environment.
struct Knobs {
    int64_t heavy_min_prims = HEAVY_MIN_PRIMS; // packet search in two launches from this many primitives (0: always)
    int64_t heavy_after = HEAVY_AFTER;         // ... a wave hands over after this many primitives met
    int64_t heavy_workers = HEAVY_WORKERS;     // ... to this many waves
    int64_t heavy_small_rays = HEAVY_SMALL_RAYS; // a call of at most this many rays (a share of a frame) searches a scene of
                                               // HEAVY_SMALL_PRIMS primitives or more in two launches too, handing over after HEAVY_SMALL_AFTER (0: never)
    int64_t shade_block = 64;                  // threads of a workgroup of the shading kernel: 64, 128 or 256 (its waves share nothing, and one-wave
                                               // workgroups leave no place on a SIMD idle until the slowest of four waves has finished)
    int64_t direct_block = 64;                 // threads of a workgroup of the one-kernel renderers of refracted and mirror rays: 64, 128 or 256
    int64_t packet_rays = 0;                   // rays per lane of the packet search: 1 or 4 (0: timed on the first frames)
    int64_t refract_single = 0;                // 1: one lane per listed ray in the second step of such a frame (measurement aid)
    int64_t refract_whole = 0;                 // 1: a searched scene with glass is rendered by the one kernel that walks every ray's tree (the former way)
    int64_t gather_general = 0;                // 1: the gather shaders' general kernel also for scenes of a few planes and spheres
    int64_t walk_blocks = 8192;                // workgroups of the searched gather's walk
    int64_t readback = 0;                      // rt355_render into a page-locked pixmap: 0 timed per kind of frame, 1 kernels write it, 2 copied in parts
    int64_t render_parts = 6;                  // ... parts of a one-kernel frame that is copied
    int64_t tail_parts = 4;                    // ... parts of a searched frame's last stage
    int64_t no_tail_parts = 0;                 // ... 1: a searched frame is enqueued and copied as a whole
    int64_t tail_bottom_first = 0, tail_even = 0, copy_streams = 2, last_copied = 0, no_copy_prime = 0; // (measurement aids)
    int64_t debug_readback = 0;                // 1: the handle prints its rt355_render call times when it is destroyed
};
struct KnobName { const char *name; int64_t Knobs::*field; int64_t lo, hi; };
static const KnobName kKnobs[] = {
    {"heavy_min_prims", &Knobs::heavy_min_prims, 0, INT32_MAX}, {"heavy_after", &Knobs::heavy_after, 1, INT32_MAX},
    {"heavy_workers", &Knobs::heavy_workers, 1, 1024}, {"heavy_small_rays", &Knobs::heavy_small_rays, 0, INT32_MAX},
    {"packet_rays", &Knobs::packet_rays, 0, 4}, {"shade_block", &Knobs::shade_block, 64, 256}, {"direct_block", &Knobs::direct_block, 64, 256},
    {"gather_general", &Knobs::gather_general, 0, 1}, {"refract_whole", &Knobs::refract_whole, 0, 1}, {"refract_single", &Knobs::refract_single, 0, 1}, {"walk_blocks", &Knobs::walk_blocks, 1, 1 << 20},
    {"readback", &Knobs::readback, 0, 2}, {"render_parts", &Knobs::render_parts, 1, 64}, {"tail_parts", &Knobs::tail_parts, 1, 64},
    {"no_tail_parts", &Knobs::no_tail_parts, 0, 1}, {"tail_bottom_first", &Knobs::tail_bottom_first, 0, 1},
    {"tail_even", &Knobs::tail_even, 0, 1}, {"copy_streams", &Knobs::copy_streams, 1, 2}, {"last_copied", &Knobs::last_copied, 0, 1},
    {"no_copy_prime", &Knobs::no_copy_prime, 0, 1}, {"debug_readback", &Knobs::debug_readback, 0, 1},
};
static int set_knob(Knobs &k, const char *name, int64_t value) {
    for (const KnobName &kn : kKnobs)
        if (!strcmp(kn.name, name)) {
            if (value < kn.lo || value > kn.hi || (kn.field == &Knobs::packet_rays && !(value == 0 || value == 1 || value == 4)) ||
                ((kn.field == &Knobs::shade_block || kn.field == &Knobs::direct_block) && !(value == 64 || value == 128 || value == 256)))
                return fail(RT355_ERR_ARG, std::string("option ") + name + ": value out of range");
            k.*(kn.field) = value;
            return RT355_OK;
        }
    return fail(RT355_ERR_ARG, std::string("unknown option ") + name);
}
static Knobs knobs_from_environment() {
    Knobs k;
    for (const KnobName &kn : kKnobs) {
        std::string env = "RT355_";
        for (const char *c = kn.name; *c; c++) env += (char)toupper((unsigned char)*c);
        if (const char *e = getenv(env.c_str())) {
            // (the former switches RT355_READBACK_COPY / RT355_READBACK_DIRECT are readback = 2 / 1)
            const int64_t v = *e ? strtoll(e, nullptr, 0) : 1;
            (void)set_knob(k, kn.name, v); // a value out of range is ignored
        }
    }
    if (getenv("RT355_READBACK_COPY")) k.readback = 2;
    if (getenv("RT355_READBACK_DIRECT")) k.readback = 1;
    return k;
}

// What the host keeps per frame in flight: the timing events, the page-locked copy of the counters, and what
// rt355_scene_sync needs to turn them into the caller's statistics.
struct FrameBook {
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_done = nullptr; // ev_done: the counters have reached host_counters
    unsigned long long *host_counters = nullptr; // pinned copy of Work::counters, filled by the frame's last command
    std::vector<hipEvent_t> ev_search;  // begin/end pairs
    std::vector<hipEvent_t> ev_resolve; // begin/end pairs
    int resolve_launches = 0, search_launches = 0;
    std::vector<std::pair<int, int>> launch_choice; // (kind of launch, rays per lane) per search launch
    rt355_stats *pending_stats = nullptr;
    hipStream_t pending_stream = nullptr;
    uint64_t pending_primary = 0;
    int pending_nprims = 0;
    bool pending = false;
};
struct MultiScene;
struct rt355_scene {
    MultiScene *multi = nullptr; // rt355_scene_create_multi: this handle only deals calls to one scene per device
    int device = 0;
    Knobs knobs{};
    SceneView view{};
    int n_padded = 0;
    bool has_tris = false;
    bool reflective = false; // some object has ks != 0
    bool refractive = false; // some object has fresnel < 1
    double rmax = 0;   // largest sphere radius / triangle bounding radius
    double extent = 0; // largest |coordinate| of a sphere centre or triangle vertex
    // owned device allocations of the scene
    std::vector<void *> owned;
    // per-render bundle tables and per-ray workspace (grown on demand)
    Tables tables{};   // one set per table slot: the lights of a group, then the camera rays
    std::vector<Bundle> built; // the bundle each table slot currently holds (kind 0 = none)
    int n_table = 0;   // table rows per slot: whole superchunks (multiples of 4096) >= n_padded
    Work work{};
    uint32_t work_rays = 0;
    int work_slots = 0, table_slots = 0;
    DLight *d_lights = nullptr; int lights_cap = 0;
    double *d_dirs = nullptr; int n_dirs = 0; // direction set of the gather shaders
    size_t g_live_rows = 0; // capacity of Work::g_live (rows of work_rays)
    GeneralTable general{};                     // scene table for mirror rays (reflective scenes that are searched)
    bool has_general = false;
    double ks_abs_max = 0;                      // largest |ks| of the scene
    // timing / stats of the outstanding render: two sets, so that a sweep can enqueue frame i + 1 while frame i still
    // runs (it waits for the statistics of frame i - 1 only); every other caller uses set 0 alone
    FrameBook book[2];
    int cur = 0; // the set the next enqueue uses
    FrameBook &bk() { return book[cur]; }
    double resolve_ms = 0;
    double search_ms = 0;
    int last_search_launches = 0, last_resolve_launches = 0; // of the frame synced last
    std::vector<double> launch_ms, resolve_launch_ms;
    // packet size per kind of search launch (0 camera rays, 1 lights with a position, 2 directional lights):
    // best time seen with 4 and with 1 ray per lane (< 0 = not tried yet), and what each launch of the
    // outstanding frame used
    double tune_ms[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}};
    // rt355_render (host pixmap): the device framebuffer kept from call to call, the stream the readback runs on
    // beside the rendering of the next part of the frame, and page-locked staging for pixmaps that are pageable
    unsigned char *fb = nullptr; size_t fb_bytes = 0;
    unsigned char *fb2 = nullptr; size_t fb2_bytes = 0; // rt355_render_sweep: the framebuffer of every other frame
    unsigned char *stage = nullptr; size_t stage_bytes = 0;
    hipStream_t copy_stream = nullptr, copy_stream2 = nullptr;
    hipStream_t sweep_stream = nullptr; // rt355_render_sweep renders on a stream of the handle's own
    double last_ms = 0;            // device time of the last frame (rt355_scene_sync)
    uint64_t readback_key = 0;     // rt355_render: the kind of frame the choice below was made for (0: none yet)
    uint64_t readback_frames = 0;  // ... frames of that kind rendered so far
    double debug_ms[2] = {0, 0}; uint64_t debug_calls[2] = {0, 0}; // RT355_DEBUG_READBACK: call time summed per way
    std::vector<float> debug_series;                               // ... and every call's (negative: the direct way)
    double readback_ms[2][3] = {{0, 0, 0}, {0, 0, 0}}; // ... what the last three calls took each way: [0] direct, [1] copied in parts (ms of wall time)
    uint32_t readback_calls[2] = {0, 0};                // ... and how many calls went each way
    std::vector<hipEvent_t> ev_copy, ev_part; // per part of a frame: its rows have arrived / have been rendered
};

// One scene on several GPUs of this process (rt355_scene_create_multi, below): a scene and a host thread per device.
#define MULTI_TILE_ROWS 32
#define MULTI_MAX_DEVICES 64
struct MultiWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = RT355_OK;
    std::string err;
};
// Row cuts of contiguous slabs, one per device, moved from frame to frame so that the slabs take the same time.  A
// slab's time is read as a part every share pays whatever its size (half the fastest share's time: launches, fills,
// the searches' depth) plus the sum of a cost per run of `align` rows (the packets of the search are 16 pixels high);
// after a frame the costs of a slab's runs are scaled so that they add up to what the slab took -- the profile keeps
// what earlier cuts have shown of it, and sharpens where the cuts move -- and the new cuts divide the summed cost evenly.
struct SlabBalance {
    int y0 = 0, y1 = 0, n = 0, align = 16;
    std::vector<double> cost; // per run of rows
    std::vector<int> cuts;    // n + 1 rows, y0 first and y1 last
    bool fits(int a, int b, int parts) const { return a == y0 && b == y1 && parts == n; }
    void start(int a, int b, int parts) {
        y0 = a; y1 = b; n = parts;
        align = (b - a) >= parts * 16 ? 16 : 1;
        cost.assign((size_t)((b - a + align - 1) / align), 1.0);
        cut();
    }
    void cut() {
        const int runs = (int)cost.size();
        double total = 0;
        for (double c : cost) total += c;
        cuts.assign(1, y0);
        int run = 0;
        double done = 0; // the runs before `run` add up to `done`
        for (int j = 1; j < n; j++) {
            const double want = total * j / n;
            while (run < runs - 1 && done + cost[run] <= want) { done += cost[run]; run++; }
            const int at = run + (want - done > 0.5 * cost[run] ? 1 : 0); // the nearer end of the run the target falls in
            const int lo = (cuts.back() - y0) / align + 1, hi = runs - (n - j);
            cuts.push_back(y0 + std::max(lo, std::min(hi, at)) * align);
        }
        cuts.push_back(y1);
    }
    void update(const std::vector<double> &ms) {
        double fastest = ms[0];
        for (double t : ms) fastest = std::min(fastest, t);
        for (int k = 0; k < n; k++) {
            const int first = (cuts[k] - y0) / align, last = (cuts[k + 1] - y0 + align - 1) / align;
            double have = 0;
            for (int r = first; r < last; r++) have += cost[r];
            const double scale = pow(std::max(ms[k] - 0.5 * fastest, 1e-9) / std::max(have, 1e-30), 0.6); // (damped: a full correction swings to and fro)
            for (int r = first; r < last; r++) cost[r] *= scale;
        }
        cut();
    }
};
struct MultiScene {
    std::vector<rt355_scene *> scenes;
    std::vector<int> devices;
    std::vector<std::unique_ptr<MultiWorker>> workers;
    int gather = 0;  // 0: every GPU writes its rows into the host pixmap; 1: rows gathered on the first GPU, one readback
    int balance = 0; // 0: a frame is dealt in interleaved 32-row tiles; 1: in contiguous slabs whose cuts follow the times the shares took in the calls before
    SlabBalance slabs;
};
extern "C" const char *rt355_last_error(void) { return g_err.c_str(); }
extern "C" int rt355_abi_version(void) { return RT_ABI_VERSION; }
extern "C" double rt355_jitter(uint32_t seed, int32_t x, int32_t y, int32_t sample, int32_t dim) {
    return rt355x_jitter(seed, x, y, sample, dim);
}
extern "C" double rt355_rotation(uint32_t seed, int32_t x, int32_t y, int32_t sample, int32_t axis) {
    return rt355x_rotation(seed, x, y, sample, axis);
}
extern "C" double rt355_strata(uint32_t seed, int32_t x, int32_t y, int32_t which) {
    return rt355x_strata(seed, x, y, which);
}
extern "C" int rt355_sincos(double x, double *sin_out, double *cos_out) {
    if (!sin_out || !cos_out || !(x >= 0 && x < 8)) return fail(RT355_ERR_ARG, "rt355_sincos: 0 <= x < 8");
    sincos_rounded(x, sin_out, cos_out);
    return RT355_OK;
}

// ---- the array of rt355_render_device_noalloc ---------------------------------------------------------------
// Device memory for callers that cannot allocate: the statistics counters, the device copy of the light list
// and the sub-sample colours of a supersampled frame, in one array of the library's code object.  Each GPU
// holds its own copy at its own address; `scratch` below is only the host-side name of the symbol, so the
// address is asked for once per device (noalloc_bind, from rt355_scene_create with that device current) and
// everything -- kernel arguments, hipMemsetAsync, copies -- goes through the pointer kept here.
#define NOALLOC_COUNTER_BYTES ((size_t)COUNTER_SHARDS * COUNTER_STRIDE * sizeof(unsigned long long))
#define NOALLOC_LIGHT_BYTES ((size_t)RT355_NOALLOC_MAX_LIGHTS * sizeof(DLight))
#define NOALLOC_LIGHTS_AT ((NOALLOC_COUNTER_BYTES + 255) / 256 * 256)
#define NOALLOC_SUB_AT ((NOALLOC_LIGHTS_AT + NOALLOC_LIGHT_BYTES + 255) / 256 * 256)
#define RT355_NOALLOC_BYTES (NOALLOC_SUB_AT + (size_t)RT355_NOALLOC_MAX_SUBSAMPLES * sizeof(uint32_t))
static __device__ __align__(256) unsigned char scratch[RT355_NOALLOC_BYTES];
#define NOALLOC_MAX_DEVICES 64
static unsigned char *g_noalloc_ptr[NOALLOC_MAX_DEVICES]; // per device: where its copy of the array lies
static std::mutex g_noalloc_mutex;
static int noalloc_bind(int device) { // the current device is `device`
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return RT355_OK; // (no such machine; the entry point then refuses)
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_ptr[device]) return RT355_OK;
    void *ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return fail(RT355_ERR_DEVICE, "hipGetSymbolAddress gave no address for the no-alloc array");
    g_noalloc_ptr[device] = (unsigned char *)ptr;
    return RT355_OK;
}
// the light list travels as a kernel argument (a copy from pageable host memory may stage through memory the
// runtime allocates): up to LIGHT_PACK lights per launch
#define LIGHT_PACK 16
struct LightPack { DLight l[LIGHT_PACK]; };
__global__ void k_put_lights(LightPack pack, DLight *__restrict__ dst, int first, int count) {
    const int i = threadIdx.x;
    if (i < count) dst[first + i] = pack.l[i];
}

template <typename T>
static int upload(rt355_scene *sc, const std::vector<T> &v, const T **out) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    sc->owned.push_back(p);
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T *)p;
    return RT355_OK;
}

// The scene table of k_search_general: bounding spheres of the spheres and triangles in Morton order of their
// centres, of every 64 of them and of every 64 of those.  Radii are rounded up and carry what the rounding of
// the centres to fp32 moved them by, so every sphere of the table contains what it stands for.
namespace {
inline float round_up(double x) {
    float f = (float)x;
    return ((double)f < x) ? nextafterf(f, INFINITY) : f;
}
inline uint64_t spread21(uint64_t v) { // 21 bits -> every third bit
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full; v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
}
struct BoundSphere { double c[3]; double r; uint32_t id; double nrm[3]; bool tri; double lo[3], hi[3]; };
// smallest-effort enclosing sphere of a run of spheres: centre = mean of their centres
inline BoundSphere enclose(const BoundSphere *b, size_t count) {
    BoundSphere o{};
    for (size_t i = 0; i < count; i++) for (int a = 0; a < 3; a++) o.c[a] += b[i].c[a] / (double)count;
    for (size_t i = 0; i < count; i++) {
        double dx = b[i].c[0] - o.c[0], dy = b[i].c[1] - o.c[1], dz = b[i].c[2] - o.c[2];
        o.r = std::max(o.r, sqrt(dx * dx + dy * dy + dz * dz) * (1 + 1e-12) + b[i].r);
    }
    return o;
}
inline float4 to_row(const BoundSphere &b) {
    float4 q = make_float4((float)b.c[0], (float)b.c[1], (float)b.c[2], 0.f);
    double dx = b.c[0] - (double)q.x, dy = b.c[1] - (double)q.y, dz = b.c[2] - (double)q.z;
    q.w = round_up((b.r + sqrt(dx * dx + dy * dy + dz * dz)) * (1 + 1e-9) + 1e-300);
    return q;
}
} // namespace
static int build_general_table(rt355_scene *sc, const std::vector<DObject> &objs, const std::vector<DTriangle> &tris,
                               const std::vector<int32_t> &prim_obj, const std::vector<int32_t> &prim_tri) {
    std::vector<BoundSphere> prims;
    std::vector<uint32_t> planes;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (size_t p = 0; p < prim_obj.size(); p++) {
        const DObject &o = objs[prim_obj[p]];
        BoundSphere b{};
        if (prim_tri[p] >= 0 && triangle_is_unbounded(tris[prim_tri[p]].p0, tris[prim_tri[p]].p1, tris[prim_tri[p]].p2)) {
            planes.push_back((uint32_t)p | CAND_TRI); // no bound holds its hits: a candidate of every ray
            continue;
        } else if (prim_tri[p] >= 0) {
            const DTriangle &t = tris[prim_tri[p]];
            for (int a = 0; a < 3; a++) b.c[a] = (t.p0[a] + t.p1[a] + t.p2[a]) / 3.0;
            for (const double *v : {t.p0, t.p1, t.p2}) {
                double dx = v[0] - b.c[0], dy = v[1] - b.c[1], dz = v[2] - b.c[2];
                b.r = std::max(b.r, sqrt(dx * dx + dy * dy + dz * dz) * (1 + 1e-12));
            }
            for (int a = 0; a < 3; a++) {
                b.lo[a] = std::min(t.p0[a], std::min(t.p1[a], t.p2[a]));
                b.hi[a] = std::max(t.p0[a], std::max(t.p1[a], t.p2[a]));
            }
            b.id = (uint32_t)p | CAND_TRI;
            b.tri = true;
            for (int a = 0; a < 3; a++) b.nrm[a] = t.normal[a];
        } else if (o.type == RT_OBJ_SPHERE) {
            for (int a = 0; a < 3; a++) b.c[a] = o.p[a];
            b.r = fabs(o.radius);
            for (int a = 0; a < 3; a++) { b.lo[a] = b.c[a] - b.r; b.hi[a] = b.c[a] + b.r; }
            b.id = (uint32_t)p | CAND_SPHERE;
        } else { // planes, and environments (a sphere around everything would only bloat its group): candidates of every ray
            planes.push_back((uint32_t)p | (o.type == RT_OBJ_ENVIRONMENT ? CAND_ENVIRONMENT : 0u));
            continue;
        }
        if (!(std::isfinite(b.c[0]) && std::isfinite(b.c[1]) && std::isfinite(b.c[2]) && std::isfinite(b.r)))
            return fail(RT355_ERR_ARG, "primitive " + std::to_string(p) + ": coordinates are not finite");
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.c[a]); hi[a] = std::max(hi[a], b.c[a]); }
        prims.push_back(b);
    }
    // Order: the order of the 8-ary tree of walk_ray.  A node's run (8 S consecutive primitives, S a power of 8; the
    // last node of a level may hold fewer) is cut by position at 4 S, its halves at 2 S and 6 S, its quarters at S, 3 S,
    // 5 S and 7 S -- each piece along the axis its centres extend furthest in, the lower coordinates first.  Every
    // group of 8, 64, 512, 4096 ... consecutive primitives is then a box-shaped cell: the children of the tree's nodes,
    // and the groups of the beam search's three levels.
    // (RT355_GENERAL_ORDER=morton keeps the plain Morton order.)
    int n_lvls = prims.empty() ? 0 : 1;
    for (size_t span = 8; span < prims.size(); span *= 8) n_lvls++;
    std::vector<BoundSphere> sorted(prims.size());
    if (getenv("RT355_GENERAL_ORDER") && !strcmp(getenv("RT355_GENERAL_ORDER"), "morton")) {
        std::vector<std::pair<uint64_t, uint32_t>> order(prims.size());
        for (size_t i = 0; i < prims.size(); i++) {
            uint64_t code = 0;
            for (int a = 0; a < 3; a++) {
                const double span = hi[a] - lo[a];
                const double u = span > 0 ? (prims[i].c[a] - lo[a]) / span : 0.0;
                code |= spread21((uint64_t)std::min(2097151.0, std::max(0.0, u * 2097152.0))) << a;
            }
            order[i] = {code, (uint32_t)i};
        }
        std::sort(order.begin(), order.end());
        for (size_t i = 0; i < prims.size(); i++) sorted[i] = prims[order[i].second];
    } else {
        sorted = prims;
        struct Run { size_t b, e; int L; };
        std::vector<Run> todo;
        if (n_lvls) todo.push_back({0, sorted.size(), 0});
        while (!todo.empty()) {
            const Run r = todo.back();
            todo.pop_back();
            const size_t S = size_t(1) << (3 * (n_lvls - 1 - r.L)); // a child's run
            // three rounds of cuts: the run, its halves, its quarters (pieces past the run's end are empty, and a piece
            // that ends before its middle is not cut)
            for (int round = 0; round < 3; round++) {
                const size_t piece = (8 * S) >> round; // length of a whole piece
                for (size_t pb = r.b; pb < r.e; pb += piece) {
                    const size_t pe = std::min(r.e, pb + piece), mid = pb + piece / 2;
                    if (mid >= pe) continue;
                    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
                    for (size_t i = pb; i < pe; i++)
                        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], sorted[i].c[a]); mx[a] = std::max(mx[a], sorted[i].c[a]); }
                    int axis = 0;
                    for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
                    std::nth_element(sorted.begin() + pb, sorted.begin() + mid, sorted.begin() + pe,
                                     [axis](const BoundSphere &p, const BoundSphere &q) { return p.c[axis] < q.c[axis]; });
                }
            }
            if (S > 1)
                for (size_t cb = r.b; cb < r.e; cb += S) todo.push_back({cb, std::min(r.e, cb + S), r.L + 1});
        }
    }
    const size_t n_chunks = (sorted.size() + 63) / 64, n_supers = (n_chunks + 63) / 64;
    const float nanf_ = nanf("");
    std::vector<float4> prim_rows(n_supers * 4096, make_float4(nanf_, nanf_, nanf_, nanf_)), chunk_rows(n_supers * 64, make_float4(nanf_, nanf_, nanf_, nanf_)), super_rows(n_supers);
    std::vector<uint32_t> prim_ids(n_supers * 4096, 0u);
    std::vector<float4> prim_nrm(n_supers * 4096, make_float4(0.f, 0.f, 0.f, 0.f));
    std::vector<BoundSphere> chunks(n_chunks);
    std::vector<float4> mini_rows(n_supers * 512, make_float4(nanf_, nanf_, nanf_, nanf_));
    for (size_t i = 0; i < sorted.size(); i++) {
        prim_rows[i] = to_row(sorted[i]);
        prim_ids[i] = sorted[i].id;
        const double *nn = sorted[i].nrm;
        const bool unit = sorted[i].tri && fabs(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2] - 1.0) < 1e-9; // a degenerate triangle has none
        if (unit) prim_nrm[i] = make_float4((float)nn[0], (float)nn[1], (float)nn[2], 1.f);
    }
    for (size_t m = 0; m * 8 < sorted.size(); m++) mini_rows[m] = to_row(enclose(&sorted[m * 8], std::min<size_t>(8, sorted.size() - m * 8)));
    for (size_t j = 0; j < n_chunks; j++) {
        chunks[j] = enclose(&sorted[j * 64], std::min<size_t>(64, sorted.size() - j * 64));
        chunk_rows[j] = to_row(chunks[j]);
    }
    for (size_t k = 0; k < n_supers; k++) super_rows[k] = to_row(enclose(&chunks[k * 64], std::min<size_t>(64, n_chunks - k * 64)));
    // the 8-ary tree of walk_ray: every level's boxes straight from the primitives of the run, moved out by
    // 2^-18 of (the largest coordinate sum of the box + 1), then put on the byte grid of their group
    const size_t last_at = n_lvls ? ((size_t(8) << (3 * (n_lvls - 1))) - 8) / 7 : 0; // nodes above the last level
    const size_t n_groups_q = std::max<size_t>(1, last_at / 8 + (sorted.size() + 7) / 8);
    std::vector<uint32_t> nodes(n_groups_q * 16, 0u);
    for (int L = 0; L < n_lvls; L++) {
        const size_t span = size_t(1) << (3 * (n_lvls - 1 - L)), at = ((size_t(8) << (3 * L)) - 8) / 7;
        const size_t n_children = (sorted.size() + span - 1) / span;
        for (size_t gi = 0; gi * 8 < n_children; gi++) {
            double blo[8][3], bhi[8][3], glo[3] = {1e300, 1e300, 1e300}, ghi[3] = {-1e300, -1e300, -1e300};
            uint32_t valid = 0u;
            for (size_t c = 0; c < 8 && gi * 8 + c < n_children; c++) {
                const size_t i = gi * 8 + c;
                double lo3[3] = {1e300, 1e300, 1e300}, hi3[3] = {-1e300, -1e300, -1e300};
                for (size_t q = i * span; q < std::min(sorted.size(), (i + 1) * span); q++)
                    for (int a = 0; a < 3; a++) { lo3[a] = std::min(lo3[a], sorted[q].lo[a]); hi3[a] = std::max(hi3[a], sorted[q].hi[a]); }
                double l1 = 1.0;
                for (int a = 0; a < 3; a++) l1 += std::max(fabs(lo3[a]), fabs(hi3[a]));
                const double margin = l1 * (1.0 / 262144.0);
                for (int a = 0; a < 3; a++) {
                    blo[c][a] = lo3[a] - margin; bhi[c][a] = hi3[a] + margin;
                    glo[a] = std::min(glo[a], blo[c][a]); ghi[a] = std::max(ghi[a], bhi[c][a]);
                }
                valid |= 1u << c;
            }
            uint32_t *grp = &nodes[(at / 8 + gi) * 16];
            uint32_t expo[3];
            for (int a = 0; a < 3; a++) {
                const float corner = -round_up(-glo[a]);
                int e = 0;
                (void)frexp((ghi[a] - (double)corner) / 254.0, &e); // 2^e >= its argument
                e = std::min(254, std::max(1, e + 127));
                while (e < 254 && ldexp(1.0, e - 127) * 254.0 < ghi[a] - (double)corner) e++;
                const double step = ldexp(1.0, e - 127);
                memcpy(&grp[a], &corner, sizeof corner);
                expo[a] = (uint32_t)e;
                for (size_t c = 0; c < 8; c++) {
                    if (!((valid >> c) & 1u)) continue;
                    const uint32_t qlo = (uint32_t)std::min(255.0, std::max(0.0, floor((blo[c][a] - (double)corner) / step)));
                    const uint32_t qhi = (uint32_t)std::min(255.0, std::max(0.0, ceil((bhi[c][a] - (double)corner) / step)));
                    grp[4 + a * 2 + c / 4] |= qlo << (8 * (c % 4));
                    grp[10 + a * 2 + c / 4] |= qhi << (8 * (c % 4));
                }
            }
            grp[3] = expo[0] | (expo[1] << 8) | (expo[2] << 16) | (valid << 24);
        }
    }
    GeneralTable &g = sc->general;
    int rc;
    const uint32_t *d_nodes = nullptr;
    if ((rc = upload(sc, nodes, &d_nodes)) || (rc = upload(sc, prim_rows, &g.prim)) || (rc = upload(sc, prim_ids, &g.prim_id)) || (rc = upload(sc, prim_nrm, &g.prim_n)) ||
        (rc = upload(sc, chunk_rows, &g.chunk)) ||
        (rc = upload(sc, mini_rows, &g.mini)) ||
        (rc = upload(sc, super_rows, &g.super)) || (rc = upload(sc, planes, &g.plane_id)))
        return rc;
    g.nodes = (const uint4 *)d_nodes;
    g.n_supers = (int32_t)n_supers; g.n_planes = (int32_t)planes.size(); g.n_lvls = n_lvls;
    double mean_group = 0; // mean radius of the groups of 8: the scale below which a beam is "thin"
    size_t n_groups8 = 0;
    for (size_t m = 0; m * 8 < sorted.size(); m++, n_groups8++) mean_group += mini_rows[m].w;
    g.tight_rho = n_groups8 ? (float)(mean_group / (double)n_groups8) : 0.0f;
    g.tight_rho *= 2.0f;
    g.tight_theta = 0.15f;
    g.far_origin = (float)sc->view.far_origin;
    if (const char *e = getenv("RT355_GENERAL_TIGHT")) { // "rho_factor,theta": tuning aid
        double a = 1, b = 0.2;
        if (sscanf(e, "%lf,%lf", &a, &b) == 2) { g.tight_rho *= (float)a; g.tight_theta = (float)b; }
    }
    sc->has_general = true;
    return RT355_OK;
}

extern "C" int rt355_scene_create(const rt355_scene_desc *d, rt355_scene **out) {
    if (!d || !out) return fail(RT355_ERR_ARG, "rt355_scene_create: null argument");
    if (d->n_objects < 0 || d->n_triangles < 0 || d->n_textures < 0 || (d->n_objects && !d->objects) ||
        (d->n_triangles && !d->triangles) || (d->n_textures && !d->textures))
        return fail(RT355_ERR_ARG, "rt355_scene_create: inconsistent counts/pointers");
    std::vector<DObject> objs(d->n_objects);
    std::vector<DTriangle> tris(d->n_triangles);
    std::vector<int32_t> prim_obj, prim_tri, by_kind[5]; // planes, spheres, triangles, environments, quadrics
    std::vector<DQuadric> quadrics;
    std::vector<double4> prim_rec;
    bool any_ks = false, any_glass = false;
    double rmax = 0, extent = 0, ks_abs_max = 0;
    for (int i = 0; i < d->n_objects; i++) {
        const rt355_object &o = d->objects[i];
        if (o.type < RT355_OBJ_PLANE || o.type > RT355_OBJ_QUADRIC)
            return fail(RT355_ERR_ARG, "object " + std::to_string(i) + ": type is one of RT355_OBJ_*");
        if (o.fresnel < 1) any_glass = true; // (main.cpp:117: any other value, NaN included, never refracts)
        if (o.ks != 0) any_ks = true;
        ks_abs_max = std::max(ks_abs_max, std::isfinite(o.ks) ? fabs(o.ks) : 1e300);
        if (o.texture_type < 0 || o.texture_type > 3)
            return fail(RT355_ERR_ARG, "object " + std::to_string(i) + ": texture_type is 0, 1, 2 or 3 (anyobject.h:27-29)");
        if ((o.texture_type == 1 || o.texture_type == 2) && !(o.texture_type == 1 && o.type == RT355_OBJ_QUADRIC) && (o.tex0 < 0 || o.tex1 < 0 || o.tex0 >= d->n_textures || o.tex1 >= d->n_textures))
            return fail(RT355_ERR_ARG, "object " + std::to_string(i) + ": texture mapping needs textures[0] and textures[1]");
        if (o.texture_type == 3 && (o.tex0 < 0 || o.tex0 >= d->n_textures)) // shader.h:66 calls through textures[0]
            return fail(RT355_ERR_ARG, "object " + std::to_string(i) + ": texture_type 3 needs textures[0]");
        DObject &q = objs[i];
        memset(&q, 0, sizeof q);
        memcpy(q.p, o.p, sizeof q.p); memcpy(q.n, o.n, sizeof q.n);
        memcpy(q.n0, o.n0, sizeof q.n0); memcpy(q.n1, o.n1, sizeof q.n1); memcpy(q.n2, o.n2, sizeof q.n2);
        if (o.type == RT355_OBJ_PLANE) put3(q.n1, hcross(H3(o.n0), H3(o.n))); // Plane ctor, anyobject.h:57
        q.radius = o.radius;
        q.ks = o.ks; q.fresnel = o.fresnel; q.ior = o.ior;
        q.type = o.type; q.texture_type = o.texture_type; q.tex0 = o.tex0; q.tex1 = o.tex1;
        q.tri_begin = o.tri_begin; q.tri_count = o.tri_count;
        q.rgb = pack_rgb(o.rgb);
        if (o.type == RT355_OBJ_SPHERE) {
            rmax = std::max(rmax, fabs(o.radius));
            for (int a = 0; a < 3; a++) extent = std::max(extent, fabs(o.p[a]));
        }
        if (o.type == RT355_OBJ_MESH) {
            if (o.tri_begin < 0 || o.tri_count < 0 || (int64_t)o.tri_begin + o.tri_count > d->n_triangles)
                return fail(RT355_ERR_ARG, "object " + std::to_string(i) + ": triangle range out of bounds");
            for (int k = 0; k < o.tri_count; k++) {
                by_kind[2].push_back((int32_t)prim_obj.size());
                prim_obj.push_back(i); prim_tri.push_back(o.tri_begin + k);
                prim_rec.push_back(index_rec(i, o.tri_begin + k));
            }
        } else if (o.type == RT355_OBJ_QUADRIC) {
            DQuadric k;
            memcpy(k.pc, o.p, sizeof k.pc); memcpy(k.n0, o.n0, sizeof k.n0); memcpy(k.n1, o.n1, sizeof k.n1); memcpy(k.n2, o.n2, sizeof k.n2);
            for (int a = 0; a < 5; a++) {
                if (o.a[a] < -(1 << 30) || o.a[a] > (1 << 30)) // the formulas double them as ints (anyobject.h:304)
                    return fail(RT355_ERR_ARG, "object " + std::to_string(i) + ": quadric coefficient out of range");
                k.a[a] = (double)o.a[a];
            }
            for (int a = 0; a < 3; a++) {
                k.s[a] = o.s[a];
                k.c2[a] = (double)(2 * o.a[a]) / (o.s[a] * o.s[a]); // 2 * aCoor[i] / pow(sCoor[i], 2), :304,317,334
            }
            k.k = (double)o.a[3] / o.s[2];                           // aCoor[3] / sCoor[2], :285,300,301,315,331
            q.tri_begin = (int32_t)quadrics.size(); q.tri_count = 0;
            quadrics.push_back(k);
            by_kind[4].push_back((int32_t)prim_obj.size());
            prim_obj.push_back(i); prim_tri.push_back(-1);
            prim_rec.push_back(index_rec(i, REC_QUADRIC));
        } else {
            by_kind[o.type == RT355_OBJ_SPHERE ? 1 : (o.type == RT355_OBJ_ENVIRONMENT ? 3 : 0)].push_back((int32_t)prim_obj.size());
            prim_obj.push_back(i); prim_tri.push_back(-1);
            prim_rec.push_back(o.type == RT355_OBJ_PLANE ? index_rec(i, -1) : make_double4(o.p[0], o.p[1], o.p[2], o.radius));
        }
        if (prim_obj.size() > (size_t)CAND_INDEX)
            return fail(RT355_ERR_UNSUPPORTED, "more than 2^30 - 1 primitives");
    }
    for (int i = 0; i < d->n_triangles; i++) { // Triangle ctor, anyobject.h:456-461
        const rt355_triangle &t = d->triangles[i];
        DTriangle &q = tris[i];
        memcpy(q.p0, t.p0, sizeof q.p0); memcpy(q.p1, t.p1, sizeof q.p1); memcpy(q.p2, t.p2, sizeof q.p2);
        h3 p0 = H3(t.p0), p1 = H3(t.p1), p2 = H3(t.p2);
        put3(q.normal, hnormalize(hcross(hsub(p1, p0), hsub(p2, p0))));
        q.p_center[0] = (p0.x + p1.x + p2.x) / 3.0; q.p_center[1] = (p0.y + p1.y + p2.y) / 3.0;
        q.p_center[2] = (p0.z + p1.z + p2.z) / 3.0;
        memcpy(q.uv0, t.uv0, sizeof q.uv0); memcpy(q.uv1, t.uv1, sizeof q.uv1); memcpy(q.uv2, t.uv2, sizeof q.uv2);
        h3 pc = H3(q.p_center);
        for (h3 v : {p0, p1, p2}) {
            h3 e = hsub(v, pc);
            rmax = std::max(rmax, sqrt(e.x * e.x + e.y * e.y + e.z * e.z));
            extent = std::max(extent, std::max(fabs(v.x), std::max(fabs(v.y), fabs(v.z))));
        }
    }
    std::vector<DTexture> tex(d->n_textures);
    std::vector<uint8_t> pool;
    for (int i = 0; i < d->n_textures; i++) {
        const rt355_texture &t = d->textures[i];
        if (!t.rgb || t.width <= 0 || t.height <= 0) return fail(RT355_ERR_ARG, "texture " + std::to_string(i) + ": empty");
        tex[i].offset = pool.size(); tex[i].sx = t.sx; tex[i].sy = t.sy; tex[i].width = t.width; tex[i].height = t.height;
        memcpy(tex[i].n0, t.n0, sizeof t.n0); memcpy(tex[i].n1, t.n1, sizeof t.n1); memcpy(tex[i].n2, t.n2, sizeof t.n2);
        memcpy(tex[i].p_center, t.p_center, sizeof t.p_center);
        for (int a = 0; a < 3; a++) // p00 = p_center - n0 * (sx / 2) - n1 * (sy / 2), texture.h:45
            tex[i].p00[a] = t.p_center[a] - t.n0[a] * (t.sx / 2) - t.n1[a] * (t.sy / 2);
        pool.insert(pool.end(), t.rgb, t.rgb + (size_t)t.width * t.height * 3);
    }
    std::vector<int32_t> kind_prims(by_kind[0]);
    kind_prims.insert(kind_prims.end(), by_kind[1].begin(), by_kind[1].end());
    kind_prims.insert(kind_prims.end(), by_kind[3].begin(), by_kind[3].end());
    kind_prims.insert(kind_prims.end(), by_kind[4].begin(), by_kind[4].end());
    kind_prims.insert(kind_prims.end(), by_kind[2].begin(), by_kind[2].end());
    rt355_scene *sc = new rt355_scene();
    sc->knobs = knobs_from_environment();
    if (sc->knobs.debug_readback) sc->debug_series.reserve(4096);
    // everything after `new` runs inside one function, so that any failure takes the one cleanup path below
    auto init = [&]() -> int {
    HIP_TRY(hipGetDevice(&sc->device));
    int rc;
    if ((rc = noalloc_bind(sc->device))) return rc;
    if ((rc = upload(sc, kind_prims, &sc->view.kind_prims)) || (rc = upload(sc, objs, &sc->view.objects)) || (rc = upload(sc, tris, &sc->view.tris)) ||
        (rc = upload(sc, prim_obj, &sc->view.prim_obj)) || (rc = upload(sc, prim_tri, &sc->view.prim_tri)) ||
        (rc = upload(sc, prim_rec, &sc->view.prim_rec)) || (rc = upload(sc, quadrics, &sc->view.quadrics)) ||
        (rc = upload(sc, tex, &sc->view.textures)) || (rc = upload(sc, pool, &sc->view.texels)))
        return rc;
    sc->view.n_objects = d->n_objects;
    sc->view.n_prims = (int)prim_obj.size();
    sc->view.n_planes = (int)by_kind[0].size(); sc->view.n_spheres = (int)by_kind[1].size();
    sc->view.n_envs = (int)by_kind[3].size(); sc->view.n_quadrics = (int)by_kind[4].size();
    sc->n_padded = ((sc->view.n_prims + PRIM_TILE - 1) / PRIM_TILE) * PRIM_TILE;
    if (sc->n_padded == 0) sc->n_padded = PRIM_TILE;
    sc->n_table = (sc->n_padded + 4095) / 4096 * 4096;
    sc->tables.n_table = sc->n_table;
    sc->tables.top_stride = (sc->n_table / 4096 + 63) / 64 * 64;
    sc->tables.n_prims_pad = std::max(256, (sc->view.n_prims + 255) / 256 * 256);
    sc->has_tris = d->n_triangles > 0;
    sc->reflective = any_ks;
    sc->refractive = any_glass;
    sc->ks_abs_max = ks_abs_max;
    sc->rmax = rmax; sc->extent = extent;
    sc->view.far_origin = std::min(FAR_ORIGIN, FAR_SCALE * (1.0 + extent + rmax));
    if (sc->view.n_prims >= SEARCH_MIN_PRIMS && (rc = build_general_table(sc, objs, tris, prim_obj, prim_tri)))
        return rc;
    for (FrameBook &b : sc->book) {
        HIP_TRY(hipEventCreate(&b.ev_begin));
        HIP_TRY(hipEventCreate(&b.ev_end));
        HIP_TRY(hipEventCreate(&b.ev_done));
        HIP_TRY(hipHostMalloc((void **)&b.host_counters, COUNTER_SHARDS * COUNTER_STRIDE * sizeof(unsigned long long), hipHostMallocDefault));
    }
    void *ctr = nullptr;
    HIP_TRY(hipMalloc(&ctr, COUNTER_SHARDS * COUNTER_STRIDE * sizeof(unsigned long long)));
    sc->owned.push_back(ctr);
    sc->work.counters = (unsigned long long *)ctr;
    for (int i = 0; i < 2; i++) { // the event pairs every frame needs (frames with more launches add theirs in ensure_work)
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        sc->book[0].ev_search.push_back(e);
        HIP_TRY(hipEventCreate(&e));
        sc->book[0].ev_resolve.push_back(e);
    }
    return RT355_OK;
    };
    const int rc_init = init();
    if (rc_init != RT355_OK) {
        const std::string why = g_err; // (the clean-up's own HIP calls must not replace the message)
        rt355_scene_destroy(sc);
        g_err = why;
        return rc_init;
    }
    *out = sc;
    return RT355_OK;
}

static void free_tables(rt355_scene *sc) {
    Tables &t = sc->tables;
    void *ps[] = {t.L1, t.L1s, t.L2, t.C1, t.C1s, t.S1, t.S1s, t.hist, t.cell, t.rank};
    for (void *p : ps) if (p) (void)hipFree(p);
    const int n_table = t.n_table, top_stride = t.top_stride, n_prims_pad = t.n_prims_pad;
    t = Tables{};
    t.n_table = n_table; t.top_stride = top_stride; t.n_prims_pad = n_prims_pad;
    sc->table_slots = 0;
    sc->built.clear();
}

static void free_work(rt355_scene *sc) {
    Work &w = sc->work;
    void *ps[] = {w.rec_d, w.rec_obj, w.rec_cm0, w.rec_cm1, w.rec_sum, w.ray32, w.cand_cnt, w.cand, w.spill.pool, w.spill.count,
                  w.heavy.list, w.heavy.count,
                  w.tile_count, w.active_rid, w.bin_hist, w.bin_id, w.bin_rank, w.face, w.far_list, w.far_n, w.lvl_clr, w.lvl_ks, w.ktotal, w.deepest, w.ray_d, w.lvl_live, w.dense_rid,
                  w.g_live};
    for (void *p : ps) if (p) (void)hipFree(p);
    sc->g_live_rows = 0;
    unsigned long long *keep = w.counters;
    w = Work{};
    w.counters = keep;
    sc->work_rays = 0;
    sc->work_slots = 0;
}

static void multi_destroy(rt355_scene *sc);
static int multi_render(rt355_scene *sc, const rt355_camera *cam, const rt355_shader *shader, const rt355_opts *opts,
                        unsigned char *pixmap, rt355_stats *stats);
static int multi_sweep(rt355_scene *sc, const rt355_camera *cams, int32_t n_frames, const rt355_shader *shader,
                       const rt355_opts *opts, unsigned char *const *pixmaps, rt355_stats *stats);
extern "C" void rt355_scene_destroy(rt355_scene *sc) {
    if (!sc) return;
    if (sc->multi) { multi_destroy(sc); return; }
    if (sc->knobs.debug_readback && sc->debug_calls[0] + sc->debug_calls[1])
        fprintf(stderr, "[rt355] rt355_render calls: %llu direct, mean %.3f ms; %llu copied in parts, mean %.3f ms\n",
                (unsigned long long)sc->debug_calls[0], sc->debug_ms[0] / (double)std::max<uint64_t>(1, sc->debug_calls[0]),
                (unsigned long long)sc->debug_calls[1], sc->debug_ms[1] / (double)std::max<uint64_t>(1, sc->debug_calls[1]));
    if (!sc->debug_series.empty()) {
        fprintf(stderr, "[rt355] rt355_render call times (ms, negative: written directly):");
        for (float v : sc->debug_series) fprintf(stderr, " %.2f", v);
        fprintf(stderr, "\n");
    }
    (void)hipSetDevice(sc->device);
    (void)hipDeviceSynchronize();
    free_work(sc);
    for (void *p : sc->owned) (void)hipFree(p);
    free_tables(sc);
    if (sc->d_lights) (void)hipFree(sc->d_lights);
    if (sc->d_dirs) (void)hipFree(sc->d_dirs);
    for (FrameBook &b : sc->book) {
        if (b.ev_begin) (void)hipEventDestroy(b.ev_begin);
        if (b.ev_end) (void)hipEventDestroy(b.ev_end);
        if (b.ev_done) (void)hipEventDestroy(b.ev_done);
        if (b.host_counters) (void)hipHostFree(b.host_counters);
        for (hipEvent_t e : b.ev_search) (void)hipEventDestroy(e);
        for (hipEvent_t e : b.ev_resolve) (void)hipEventDestroy(e);
    }
    if (sc->copy_stream) { (void)hipStreamSynchronize(sc->copy_stream); (void)hipStreamDestroy(sc->copy_stream); }
    if (sc->copy_stream2) { (void)hipStreamSynchronize(sc->copy_stream2); (void)hipStreamDestroy(sc->copy_stream2); }
    if (sc->sweep_stream) { (void)hipStreamSynchronize(sc->sweep_stream); (void)hipStreamDestroy(sc->sweep_stream); }
    if (sc->fb) (void)hipFree(sc->fb);
    if (sc->fb2) (void)hipFree(sc->fb2);
    if (sc->stage) (void)hipHostFree(sc->stage);
    for (hipEvent_t e : sc->ev_copy) (void)hipEventDestroy(e);
    for (hipEvent_t e : sc->ev_part) (void)hipEventDestroy(e);
    delete sc;
}

// Bytes of workspace one slot needs per ray; the number of slots is cut down so that all of them stay
// below SLOT_BUDGET_BYTES (a 3840x2160 frame with 8 lights in flight takes about 11 GB).
#define SLOT_BYTES_PER_RAY (sizeof(float4) + (CAND_CAP + 4) * sizeof(uint32_t))
#define SLOT_BUDGET_BYTES (96ull << 30)
static int slots_for(uint32_t n_rays, int n_lights) {
    int slots = std::max(1, std::min(n_lights, LIGHT_GROUP));
    while (slots > 1 && (unsigned long long)slots * n_rays * SLOT_BYTES_PER_RAY > SLOT_BUDGET_BYTES) slots--;
    return slots;
}

static int sync_book(rt355_scene *sc, FrameBook &bk);
static bool any_pending(const rt355_scene *sc) { return sc->book[0].pending || sc->book[1].pending; }
// n_rays: the rays of the frame being enqueued (the strides follow it); cap_rays >= n_rays: what the workspace must
// hold without growing again (the largest part of a frame that is enqueued in parts).
static int ensure_work(rt355_scene *sc, uint32_t n_rays, int slots, int n_bundles, int n_groups, uint32_t cap_rays_wanted = 0) {
    if (slots + 1 > sc->table_slots) { // one set of bundle tables per light in flight + one for the camera rays
        if (any_pending(sc)) { int rc = rt355_scene_sync(sc); if (rc) return rc; } // (a sweep's frame before this one may still run)
        free_tables(sc);
        Tables &t = sc->tables;
        const size_t sets = (size_t)slots + 1;
        const size_t rows = sets * t.n_table, chunks = rows / 64, tops = sets * t.top_stride;
        HIP_TRY(hipMalloc((void **)&t.hist, sets * (TB_CELLS + 1) * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&t.cell, sets * t.n_prims_pad * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&t.rank, sets * t.n_prims_pad * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void **)&t.L1, rows * sizeof(float4)));
        HIP_TRY(hipMalloc((void **)&t.L1s, rows * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&t.L2, rows * 3 * sizeof(float4)));
        HIP_TRY(hipMalloc((void **)&t.C1, chunks * sizeof(float4)));
        HIP_TRY(hipMalloc((void **)&t.C1s, chunks * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&t.S1, tops * sizeof(float4)));
        HIP_TRY(hipMalloc((void **)&t.S1s, tops * sizeof(float)));
        sc->table_slots = slots + 1;
        Bundle none;
        memset(&none, 0, sizeof none);
        sc->built.assign(sets, none);
    }
    while ((int)sc->bk().ev_search.size() < 2 * n_bundles) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        sc->bk().ev_search.push_back(e);
    }
    while ((int)sc->bk().ev_resolve.size() < 2 * (1 + n_groups)) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        sc->bk().ev_resolve.push_back(e);
    }
    Work &w = sc->work;
    cap_rays_wanted = std::max(cap_rays_wanted, n_rays);
    if (cap_rays_wanted <= sc->work_rays && slots <= sc->work_slots) {
        // the slot strides follow the frame being rendered, not the capacity
        w.n = n_rays;
        w.tile_stride = n_rays / RAYS_PER_TILE + 1;
        w.list_stride = n_rays + RT_BINS / RT_PATCH * 64 * PACKET_RAYS_MAX;
        return RT355_OK;
    }
    const uint32_t cap_rays = std::max(cap_rays_wanted, sc->work_rays);
    const int cap_slots = std::max(slots, sc->work_slots);
    if (any_pending(sc)) { int rc = rt355_scene_sync(sc); if (rc) return rc; } // (a sweep's frame before this one may still run)
    free_work(sc);
    const size_t n = cap_rays, g = (size_t)cap_slots;
    HIP_TRY(hipMalloc((void **)&w.rec_d, 10 * n * sizeof(double)));
    HIP_TRY(hipMalloc((void **)&w.rec_obj, n * sizeof(int32_t)));
    HIP_TRY(hipMalloc((void **)&w.rec_cm0, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.rec_cm1, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.rec_sum, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.ray32, g * n * sizeof(float4)));
    HIP_TRY(hipMalloc((void **)&w.cand_cnt, g * n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.spill.pool, g * POOL_STRIDE(n) * sizeof(uint2)));
    HIP_TRY(hipMalloc((void **)&w.spill.count, LIGHT_GROUP * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.heavy.list, (size_t)LIGHT_GROUP * HEAVY_CAP * sizeof(uint2)));
    HIP_TRY(hipMalloc((void **)&w.heavy.count, LIGHT_GROUP * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.cand, g * CAND_CAP * n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.tile_count, g * (n / RAYS_PER_TILE + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.active_rid, g * (n + RT_BINS / RT_PATCH * 64 * PACKET_RAYS_MAX) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.bin_hist, g * HIST_STRIDE * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.bin_id, g * n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.bin_rank, g * n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.face, n * sizeof(uint8_t)));
    HIP_TRY(hipMalloc((void **)&w.far_list, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&w.far_n, sizeof(uint32_t)));
    sc->work_rays = cap_rays;
    sc->work_slots = cap_slots;
    w.n = n_rays;
    w.tile_stride = n_rays / RAYS_PER_TILE + 1;
    w.list_stride = n_rays + RT_BINS / RT_PATCH * 64 * PACKET_RAYS_MAX;
    return RT355_OK;
}

static DCamera set_camera(const rt355_camera *c) { // Scene::SetCamera, main.cpp:255-262
    DCamera o;
    h3 v_view = H3(c->v_view), v_up = H3(c->v_up), p_eye = H3(c->p_eye);
    h3 n_view = hnormalize(v_view);
    h3 p_center = hadd(p_eye, hmul(n_view, c->dist));
    h3 v0 = hcross(v_view, v_up);
    h3 n0 = hnormalize(v0);
    h3 n1 = hcross(n0, n_view);
    h3 p00 = hsub(hsub(p_center, hmul(n0, c->s_x / 2)), hmul(n1, c->s_y / 2));
    put3(o.p_eye, p_eye); put3(o.p00, p00); put3(o.n0, n0); put3(o.n1, n1);
    o.s_x = c->s_x; o.s_y = c->s_y;
    return o;
}

// The rows a call renders: the slab [row_begin, row_end), or its share of the slab's row tiles.
struct Rows { int y0, y1, count; };
static int rows_of(const rt355_opts *opts, Rows *out) {
    int y0 = opts->row_begin, y1 = opts->row_end;
    if (y0 == 0 && y1 == 0) y1 = opts->height;
    if (y0 < 0 || y1 > opts->height || y0 >= y1) return fail(RT355_ERR_ARG, "row slab out of range");
    int count = y1 - y0;
    if (opts->tile_count > 1) {
        if (opts->tile_rows <= 0 || opts->tile_phase < 0 || opts->tile_phase >= opts->tile_count)
            return fail(RT355_ERR_ARG, "row tiles: tile_rows > 0 and 0 <= tile_phase < tile_count");
        count = 0;
        for (int k = opts->tile_phase; k * opts->tile_rows < y1 - y0; k += opts->tile_count)
            count += std::min(opts->tile_rows, y1 - y0 - k * opts->tile_rows);
    }
    out->y0 = y0; out->y1 = y1; out->count = count;
    return RT355_OK;
}

// The level / gather arrays of the workspace: all five or none (a frame that needs them launches kernels on all).
static int ensure_levels(rt355_scene *sc) {
    Work &wk = sc->work;
    if (wk.lvl_clr && wk.lvl_ks && wk.ktotal && wk.deepest && wk.ray_d && wk.lvl_live && wk.dense_rid) return RT355_OK;
    const size_t cap = sc->work_rays;
    auto drop = [&]() {
        void *ps[] = {wk.lvl_clr, wk.lvl_ks, wk.ktotal, wk.deepest, wk.ray_d, wk.lvl_live, wk.dense_rid};
        for (void *q : ps) if (q) (void)hipFree(q);
        wk.lvl_clr = nullptr; wk.lvl_ks = nullptr; wk.ktotal = nullptr; wk.deepest = nullptr; wk.ray_d = nullptr;
        wk.lvl_live = nullptr; wk.dense_rid = nullptr;
    };
    drop();
    hipError_t e;
    if ((e = hipMalloc((void **)&wk.lvl_clr, 4 * cap * sizeof(uint32_t))) == hipSuccess &&
        (e = hipMalloc((void **)&wk.lvl_ks, 4 * cap * sizeof(double))) == hipSuccess &&
        (e = hipMalloc((void **)&wk.ktotal, cap * sizeof(double))) == hipSuccess &&
        (e = hipMalloc((void **)&wk.deepest, cap * sizeof(uint32_t))) == hipSuccess &&
        (e = hipMalloc((void **)&wk.ray_d, 6 * cap * sizeof(double))) == hipSuccess &&
        (e = hipMalloc((void **)&wk.lvl_live, 4 * sizeof(uint32_t))) == hipSuccess &&
        (e = hipMalloc((void **)&wk.dense_rid, cap * sizeof(uint32_t))) == hipSuccess)
        return RT355_OK;
    drop();
    return fail(RT355_ERR_DEVICE, std::string("level arrays: ") + hipGetErrorString(e));
}

// The array of the gather shaders on a searched scene, grown to what the direction set asks for.
static int ensure_gather(rt355_scene *sc, size_t live_rows) {
    Work &wk = sc->work;
    if (live_rows <= sc->g_live_rows) return RT355_OK;
    if (wk.g_live) (void)hipFree(wk.g_live);
    wk.g_live = nullptr; sc->g_live_rows = 0;
    const hipError_t e = hipMalloc((void **)&wk.g_live, live_rows * (size_t)sc->work_rays * sizeof(unsigned long long));
    if (e != hipSuccess) return fail(RT355_ERR_DEVICE, std::string("gather rays: ") + hipGetErrorString(e));
    sc->g_live_rows = live_rows;
    return RT355_OK;
}

// A frame that one kernel renders can be enqueued as several calls, one per group of rows, that count into the
// same statistics and are waited for as one: `first` clears the counters and uploads the lights, `last` closes
// the frame (rt355_render reads every part back while the next one renders).
// A scene with glass (Shader::shading's renderer) that is searched: rendered in two steps, see k_render_refract<.., CHILDREN>.
static bool searched_with_glass(const rt355_scene *sc, const rt355_opts *opts) {
    return sc->refractive && opts->shading == RT355_SHADING_LIGHTS && opts->mode != RT355_MODE_BRUTE && sc->has_general &&
           sc->view.n_prims >= SEARCH_MIN_PRIMS && !sc->knobs.refract_whole;
}
struct PartOfFrame { bool first, last; int max_rows = 0; }; // max_rows: the rows of the frame's largest part (0: this one)
// How rt355_render takes a searched frame's pixels back while the frame is still being shaded: the last stage of the
// frame (the shading of the last group of lights, the fold of the mirror levels, k_average) is launched in `n_parts`
// parts of whole tile rows, and after each part `copy_rows(ya, yb)` is called for the local rows it completed in the
// device framebuffer.  The last part's pixels go to `last_direct` instead (the caller's page-locked pixmap as the
// device sees it), when that is given: no later kernel would overlap its transfer.
struct TailParts {
    int n_parts;
    unsigned char *last_direct;
    std::function<int(int, int)> copy_rows;
};
// One frame (or slab) enqueued on `hip_stream`.  noalloc: the entry point that allocates nothing -- the counters,
// the lights and the sub-sample colours live in the static array, and frames that need the workspace are refused.
static int render_enqueue(rt355_scene *sc, const rt355_camera *cam, const rt355_shader *shader,
                          const rt355_opts *opts, void *d_pixmap, void *hip_stream, rt355_stats *stats, bool noalloc,
                          PartOfFrame part = PartOfFrame{true, true}, const TailParts *tail = nullptr) {
    if (!sc || !cam || !shader || !opts || !d_pixmap) return fail(RT355_ERR_ARG, "rt355_render_device: null argument");
    if (opts->width <= 0 || opts->height <= 0) return fail(RT355_ERR_ARG, "width and height must be positive");
    if (opts->alias < 0 || opts->alias > 64) return fail(RT355_ERR_ARG, "alias must be 0..64");
    if (shader->n_lights < 0 || (shader->n_lights && !shader->lights)) return fail(RT355_ERR_ARG, "bad light list");
    Rows rows;
    { int rc0 = rows_of(opts, &rows); if (rc0) return rc0; }
    const int y0 = rows.y0;
    if (opts->mode != RT355_MODE_ALLPAIRS && opts->mode != RT355_MODE_BRUTE && opts->mode != RT355_MODE_PACKET)
        return fail(RT355_ERR_ARG, "unknown mode");
    if (opts->shading != RT355_SHADING_LIGHTS && opts->shading != RT355_SHADING_OCCLUSION && opts->shading != RT355_SHADING_BLEED &&
        opts->shading != RT355_SHADING_SOFT)
        return fail(RT355_ERR_ARG, "unknown shading");
    const bool soft = opts->shading == RT355_SHADING_SOFT;
    const bool lit = soft || opts->shading == RT355_SHADING_LIGHTS; // CalcSubPixel with Shader::shading (not CalcSubPixel2)
    HIP_TRY(hipSetDevice(sc->device));
    // (the set of the bookkeeping this frame uses must be free; a sweep's other set may still hold the frame before)
    if (sc->bk().pending) { int rc = sync_book(sc, sc->bk()); if (rc) return rc; }
    hipStream_t st = (hipStream_t)hip_stream;
    if (rows.count == 0 && part.first && part.last) { // a share without tiles (more shares than the slab has tiles): nothing to render
        if (stats) memset(stats, 0, sizeof *stats);
        return RT355_OK;
    }

    Frame f;
    f.width = opts->width; f.height = opts->height; f.row_begin = y0; f.rows = rows.count;
    f.tile_rows = opts->tile_count > 1 ? opts->tile_rows : 1;
    f.tile_count = opts->tile_count > 1 ? opts->tile_count : 1;
    f.tile_phase = opts->tile_count > 1 ? opts->tile_phase : 0;
    f.alias = opts->alias; f.seed = opts->seed;
    const int64_t sub = opts->alias > 0 ? opts->alias : 1;
    const int64_t grid_w = f.width * sub, grid_h = f.rows * sub;
    uint64_t n_rays64 = (uint64_t)((grid_w + TILE_W - 1) / TILE_W) * (uint64_t)((grid_h + TILE_H - 1) / TILE_H) * RAYS_PER_TILE;
    if (n_rays64 > 0x7fffffffull) return fail(RT355_ERR_ARG, "slab too large: split it into fewer rows per call");
    f.grid_w = (int32_t)grid_w; f.grid_h = (int32_t)grid_h;
    f.tiles_x = (f.grid_w + TILE_W - 1) / TILE_W; f.tiles_y = (f.grid_h + TILE_H - 1) / TILE_H;
    f.n_rays = (uint32_t)n_rays64;
    f.level = -1;
    f.deferred = 0;
    f.rid_base = 0; f.avg_row0 = 0;
    const int nl = shader->n_lights;
    // (a frame enqueued in parts sizes the workspace for its largest part at once: growing it between the parts would
    // free memory that queued parts still use; all parts then group the lights alike)
    const uint32_t cap_rays = (uint32_t)std::min<uint64_t>(0x7fffffffull, (uint64_t)f.tiles_x *
                              (uint64_t)(((int64_t)std::max(part.max_rows, f.rows) * sub + TILE_H - 1) / TILE_H) * RAYS_PER_TILE);
    const int slots = slots_for(cap_rays, nl);
    const int n_groups = (nl + slots - 1) / slots;
    // Mirror rays on a searched scene: the frame is rendered level by level.  A chain goes on while the product
    // of the ks it met is >= 0.05 and it has at most three mirror rays (main.cpp:27-28,103), so the largest |ks|
    // of the scene bounds the number of levels.
    // refraction (Shader::shading's renderer only: CalcSubPixel2 does not recurse) is rendered unsearched, one
    // thread per pixel walking its whole tree of rays
    const bool refract = sc->refractive && lit;
    // ... unless the scene is searched: then the frame's own pipeline renders every camera ray's hit and shading, and
    // the kernel that walks the trees takes over only where a first call has children (k_render_refract<.., CHILDREN>)
    const bool refract_children = refract && searched_with_glass(sc, opts) && !noalloc;
    const bool can_search = (opts->mode != RT355_MODE_BRUTE) && sc->view.n_prims >= SEARCH_MIN_PRIMS && (!refract || refract_children);
    int extra_levels = 0;
    if (can_search && sc->reflective && lit && !refract) { // (with glass in the scene the trees' own mirror rays are the reflection)
        double k = 1.0;
        while (extra_levels < 3 && (k *= sc->ks_abs_max) >= 0.05) extra_levels++;
    }
    const bool levels = extra_levels > 0;
    const bool gather_searched = can_search && !lit;
    f.soft = soft ? 1 : 0;
    f.deferred = can_search ? 1 : 0;
    int rc = RT355_OK;
    unsigned char *na = nullptr; // the no-alloc array of this device
    if (noalloc) {
        if (can_search)
            return fail(RT355_ERR_UNSUPPORTED, "rt355_render_device_noalloc: this frame is searched (64 primitives or more) and needs "
                        "the per-ray workspace; use RT355_MODE_BRUTE or rt355_render_device");
        if (nl > RT355_NOALLOC_MAX_LIGHTS)
            return fail(RT355_ERR_UNSUPPORTED, "rt355_render_device_noalloc: more than " + std::to_string(RT355_NOALLOC_MAX_LIGHTS) + " lights");
        if (f.alias > 0 && f.n_rays > (uint32_t)RT355_NOALLOC_MAX_SUBSAMPLES)
            return fail(RT355_ERR_UNSUPPORTED, "rt355_render_device_noalloc: " + std::to_string(f.n_rays) + " sub-samples in one call, the limit is " +
                        std::to_string(RT355_NOALLOC_MAX_SUBSAMPLES) + ": render the frame in row slabs");
        na = (sc->device >= 0 && sc->device < NOALLOC_MAX_DEVICES) ? g_noalloc_ptr[sc->device] : nullptr;
        if (!na) return fail(RT355_ERR_DEVICE, "rt355_render_device_noalloc: the no-alloc array of this device was not bound");
    } else {
        // search launches: the camera rays and two per group of lights, per level; the gather passes record two pairs
        const int n_search = std::max((1 + 2 * n_groups) * (1 + extra_levels), gather_searched ? 2 : 1) + (refract_children ? 1 : 0);
        // (the trees of a scene with glass deal a node's shadow walks to REFRACT_GROUP lanes, each with a slot of its own)
        const int alloc_slots = refract_children ? std::max(slots, slots_for(cap_rays, REFRACT_GROUP)) : slots;
        rc = ensure_work(sc, f.n_rays, alloc_slots, n_search, (1 + n_groups) * (1 + extra_levels) - 1, cap_rays);
        if (rc) return rc;
        if ((levels || gather_searched) && (rc = ensure_levels(sc))) return rc;
    }

    // lights -> device
    DLight hl_fixed[RT355_NOALLOC_MAX_LIGHTS];
    std::vector<DLight> hl_grown;
    if (nl > RT355_NOALLOC_MAX_LIGHTS) hl_grown.resize(nl);
    DLight *hl = nl > RT355_NOALLOC_MAX_LIGHTS ? hl_grown.data() : hl_fixed;
    for (int i = 0; i < nl; i++) {
        const rt355_light &l = shader->lights[i];
        if (l.type < 0 || l.type > 3) return fail(RT355_ERR_ARG, "light type must be 0..3");
        memcpy(hl[i].position, l.position, sizeof hl[i].position);
        memcpy(hl[i].direction, l.direction, sizeof hl[i].direction);
        memcpy(hl[i].color, l.color, sizeof hl[i].color);
        hl[i].type = l.type; hl[i].pad_ = 0;
        put3(hl[i].shadow_dir, hnormalize(hmul(H3(l.direction), -1.0)));
        memcpy(hl[i].n0, l.n0, sizeof hl[i].n0); memcpy(hl[i].n1, l.n1, sizeof hl[i].n1);
        hl[i].sx = l.sx; hl[i].sy = l.sy;
    }
    DLight *d_lights = nullptr;
    if (noalloc) {
        d_lights = (DLight *)(na + NOALLOC_LIGHTS_AT);
        for (int first = 0; first < nl && part.first; first += LIGHT_PACK) {
            LightPack pack;
            const int count = std::min(LIGHT_PACK, nl - first);
            memcpy(pack.l, hl + first, (size_t)count * sizeof(DLight));
            hipLaunchKernelGGL(k_put_lights, dim3(1), dim3(64), 0, st, pack, d_lights, first, count);
        }
    } else {
        if (nl > sc->lights_cap) {
            if (sc->d_lights) { (void)hipFree(sc->d_lights); sc->d_lights = nullptr; sc->lights_cap = 0; }
            HIP_TRY(hipMalloc((void **)&sc->d_lights, (size_t)nl * sizeof(DLight)));
            sc->lights_cap = nl;
        }
        if (nl && part.first) HIP_TRY(hipMemcpyAsync(sc->d_lights, hl, (size_t)nl * sizeof(DLight), hipMemcpyHostToDevice, st));
        d_lights = sc->d_lights;
    }
    ShaderView sh;
    sh.lights = d_lights; sh.n_lights = nl;
    sh.ambient = pack_rgb(shader->ambient); sh.specular = pack_rgb(shader->specular); sh.border = pack_rgb(shader->border);
    DCamera dc = set_camera(cam);

    // bundle 0 = camera rays (common origin: the eye); bundle 1+k = shadow rays of light k (common
    // origin: the light position, or common direction for a directional light)
    if (gather_searched && !sc->has_general) return fail(RT355_ERR_ARG, "internal: the scene table for gather rays is missing");
    if (levels && !sc->has_general) return fail(RT355_ERR_ARG, "internal: the scene table for mirror rays is missing");
    std::vector<Bundle> bundles(can_search ? 2 + nl : 0); // (unsearched frames have none: no allocation on that path)
    for (auto &b : bundles) memset(&b, 0, sizeof b);
    if (can_search) {
        bundles[0].kind = 1;
        bundles[0].eye = 1;
        bundles[0].o.x = dc.p_eye[0]; bundles[0].o.y = dc.p_eye[1]; bundles[0].o.z = dc.p_eye[2];
        for (int k = 0; k < nl && !soft; k++) { // (soft shadows: no bundle, every shadow ray takes the fp64 walk)
            Bundle &b = bundles[1 + k];
            if (hl[k].type != 0) {
                b.kind = 1;
                b.o.x = hl[k].position[0]; b.o.y = hl[k].position[1]; b.o.z = hl[k].position[2];
            } else {
                h3 w = hnormalize(hmul(H3(hl[k].direction), -1.0)); // shadow rays run along -direction (shader.h:395)
                if (w.x == 0 && w.y == 0 && w.z == 0) continue;     // degenerate light: leave it to the fp64 walk
                h3 a = (fabs(w.x) <= fabs(w.y) && fabs(w.x) <= fabs(w.z)) ? h3{1, 0, 0}
                       : (fabs(w.y) <= fabs(w.z) ? h3{0, 1, 0} : h3{0, 0, 1});
                h3 e1 = hnormalize(hcross(w, a)), e2 = hcross(w, e1);
                b.kind = 2;
                b.e1.x = e1.x; b.e1.y = e1.y; b.e1.z = e1.z; b.e2.x = e2.x; b.e2.y = e2.y; b.e2.z = e2.z;
                b.rmaxp = sc->rmax * (1.0 + 1e-6) + 8.0 * 2.384185791015625e-07 * (sc->extent + sc->rmax) + 1e-30;
                b.plane_extent = (float)(2.0 * (sc->extent + sc->rmax) + 1.0);
                b.dir.x = w.x; b.dir.y = w.y; b.dir.z = w.z;
            }
        }
    }

    if (can_search && opts->mode == RT355_MODE_PACKET)
        for (int k = 0; k < nl; k++)
            if (bundles[1 + k].kind) bundles[1 + k].binned = 1;

    Work w_noalloc{};
    if (noalloc) {
        w_noalloc.counters = (unsigned long long *)na;
        w_noalloc.rec_sum = (uint32_t *)(na + NOALLOC_SUB_AT);
        w_noalloc.n = f.n_rays;
    }
    const Work &w = noalloc ? w_noalloc : sc->work;
    if (part.first) {
        HIP_TRY(hipEventRecord(sc->bk().ev_begin, st));
        HIP_TRY(hipMemsetAsync(w.counters, 0, COUNTER_SHARDS * COUNTER_STRIDE * sizeof(unsigned long long), st));
        sc->bk().pending_primary = 0;
    }
    if (!(part.first && part.last) && can_search)
        return fail(RT355_ERR_ARG, "internal: only a frame that one kernel renders is enqueued in parts");
    const unsigned per_ray_blocks = (f.n_rays + 255) / 256;
    const unsigned direct_blocks = (unsigned)((f.grid_w + 255) / 256) * (unsigned)f.grid_h; // see direct_xy
    const unsigned table_blocks = (unsigned)(sc->n_table / 256);            // one thread per table row
    const unsigned top_blocks = (unsigned)(sc->tables.top_stride * 64 / 256); // one thread per chunk, padding included
    const unsigned n_tiles = (unsigned)(f.tiles_x * f.tiles_y);
    unsigned char *pix = (unsigned char *)d_pixmap;
    int launches = 0;
    sc->bk().launch_choice.clear();
    // rays per lane for a kind of launch: try 4, then 1, then keep the faster
    auto packet_rays_for = [&](int cls) -> int {
        if (sc->knobs.packet_rays) return (int)sc->knobs.packet_rays;
        if (sc->tune_ms[cls][0] < 0) return 4;
        if (sc->tune_ms[cls][1] < 0) return 1;
        return sc->tune_ms[cls][0] <= sc->tune_ms[cls][1] ? 4 : 1;
    };
    // one search launch over the given slots (all of bundle kind `kind`); list / total = null for the camera rays
    const int heavy_min_prims = (int)sc->knobs.heavy_min_prims; // (tests: 0 = always)
    auto launch_search = [&](int kind, const Slots &sl, const uint32_t *list, const uint32_t *total,
                             uint32_t total_stride, bool spill_cleared = false) -> int {
        for (int i = 0; i < sl.count && !spill_cleared; i++) // empty spill lists for the bundles of this launch
            HIP_TRY(hipMemsetAsync(w.spill.count + sl.slot[i], 0, sizeof(uint32_t), st));
        HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches], st));
        if (opts->mode == RT355_MODE_PACKET) {
            const int cls = list ? kind : 0, pr = packet_rays_for(cls);
            sc->bk().launch_choice.push_back({cls, pr});
            const size_t n_entries = list ? (size_t)w.list_stride : (size_t)f.n_rays, per_block = (size_t)PACKET_THREADS * pr;
            const dim3 grid((unsigned)((n_entries + per_block - 1) / per_block), (unsigned)sl.count);
            // the search, then the waves that share what its longest waves handed over (blocks past the list's end leave at once)
            const dim3 grid_heavy(HEAVY_GRID, (unsigned)sl.count);
            const bool small_call = f.n_rays <= (uint64_t)sc->knobs.heavy_small_rays && sc->view.n_prims >= HEAVY_SMALL_PRIMS;
            const bool two_launches = sc->view.n_prims >= heavy_min_prims || small_call;
            Heavy hv = w.heavy;
            hv.after = (uint32_t)sc->knobs.heavy_after; // (tests: hand over early)
            if (small_call) hv.after = std::min(hv.after, HEAVY_SMALL_AFTER);
            hv.workers = (uint32_t)sc->knobs.heavy_workers;
            hv.workers = std::max<uint32_t>(PACKET_WAVES, hv.workers / PACKET_WAVES * PACKET_WAVES);
            if (two_launches) HIP_TRY(hipMemsetAsync(w.heavy.count, 0, LIGHT_GROUP * sizeof(uint32_t), st));
#define PACKET_ARGS sc->tables, w.ray32, f.n_rays, list, total, w.cand, w.cand_cnt, w.counters, w.spill, sl, w.list_stride, total_stride, hv
#define LAUNCH_PACKET(K, P) do { \
    if (!two_launches) hipLaunchKernelGGL((k_search_packet<K, P, 0>), grid, dim3(PACKET_THREADS), 0, st, PACKET_ARGS); \
    else { hipLaunchKernelGGL((k_search_packet<K, P, 1>), grid, dim3(PACKET_THREADS), 0, st, PACKET_ARGS); \
           hipLaunchKernelGGL((k_search_packet<K, P, 2>), grid_heavy, dim3(PACKET_THREADS), 0, st, PACKET_ARGS); } } while (0)
            if (kind == 1 && pr == 4) LAUNCH_PACKET(1, 4);
            else if (kind == 1) LAUNCH_PACKET(1, 1);
            else if (pr == 4) LAUNCH_PACKET(2, 4);
            else LAUNCH_PACKET(2, 1);
#undef LAUNCH_PACKET
        } else {
            sc->bk().launch_choice.push_back({-1, 0});
            const dim3 grid(n_tiles, (unsigned)sl.count);
            if (kind == 1)
                hipLaunchKernelGGL(k_search<1>, grid, dim3(SEARCH_THREADS), 0, st, sc->tables, sc->n_padded, w.ray32, f.n_rays, list, total, w.cand, w.cand_cnt, w.counters, w.spill, sl, w.list_stride, total_stride);
            else
                hipLaunchKernelGGL(k_search<2>, grid, dim3(SEARCH_THREADS), 0, st, sc->tables, sc->n_padded, w.ray32, f.n_rays, list, total, w.cand, w.cand_cnt, w.counters, w.spill, sl, w.list_stride, total_stride);
        }
        HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches + 1], st));
        launches++;
        return RT355_OK;
    };
    // (Re)build the tables of the bundles of `gr` in the table slots `tm` names; a slot that already holds
    // the same bundle (same light as in the last frame; the scene never changes) is left alone.
    const unsigned prim_blocks = (unsigned)std::max(1, (sc->view.n_prims + 255) / 256);
    auto build_tables = [&](const Group &gr, const TableMap &tm) -> int {
        Group todo = gr;
        bool any = false;
        for (int g = 0; g < gr.count; g++) {
            Bundle &have = sc->built[tm.slot[g]];
            if (!gr.b[g].kind || memcmp(&have, &gr.b[g], sizeof(Bundle)) == 0) { todo.b[g].kind = 0; continue; }
            have = gr.b[g];
            any = true;
            HIP_TRY(hipMemsetAsync(sc->tables.hist + (size_t)tm.slot[g] * (TB_CELLS + 1), 0, (TB_CELLS + 1) * sizeof(uint32_t), st));
        }
        if (!any) return RT355_OK;
        const unsigned ny = (unsigned)gr.count;
        hipLaunchKernelGGL(k_table_bin, dim3(prim_blocks, ny), dim3(256), 0, st, sc->view, todo, tm, sc->tables);
        hipLaunchKernelGGL(k_table_scan, dim3(ny), dim3(1024), 0, st, todo, tm, sc->tables);
        hipLaunchKernelGGL(k_table_rows, dim3(prim_blocks, ny), dim3(256), 0, st, sc->view, todo, tm, sc->tables);
        hipLaunchKernelGGL(k_table_chunks, dim3(table_blocks, ny), dim3(256), 0, st, todo, tm, sc->tables);
        hipLaunchKernelGGL(k_table_supers, dim3(top_blocks, ny), dim3(256), 0, st, todo, tm, sc->tables);
        return RT355_OK;
    };
    // tables, ray lists and searches of all the lights of a group: one launch per step, slot = blockIdx.y
    auto search_group = [&](const Group &gr) -> int {
        Slots searched{}, of_kind[3]{};
        TableMap tm{};
        for (int g = 0; g < gr.count; g++) {
            tm.slot[g] = g; // light k0 + g: workspace slot g, table slot g
            if (gr.b[g].kind) {
                searched.table[searched.count] = g;
                searched.slot[searched.count++] = g;
                Slots &k = of_kind[gr.b[g].kind];
                k.table[k.count] = g;
                k.slot[k.count++] = g;
            }
        }
        if (!searched.count) return RT355_OK;
        int rc1 = build_tables(gr, tm);
        if (rc1) return rc1;
        const uint32_t *total; uint32_t total_stride;
        if (opts->mode == RT355_MODE_PACKET) { // the producer of the rays counted them into direction bins: order them
            hipLaunchKernelGGL(k_scan_bins, dim3((unsigned)searched.count), dim3(1024), 0, st, w.bin_hist, w.active_rid, w.list_stride, searched);
            hipLaunchKernelGGL(k_bin_scatter, dim3(per_ray_blocks, (unsigned)searched.count), dim3(256), 0, st, w.face, w.bin_id, w.bin_rank, w.bin_hist, w.active_rid, f.n_rays, w.list_stride, searched);
            total = w.bin_hist + HIST_TOTAL; total_stride = HIST_STRIDE;
        } else { // shadow bundles are sparse: compact their active rays first
            hipLaunchKernelGGL(k_count_active, dim3(n_tiles, (unsigned)searched.count), dim3(SEARCH_THREADS), 0, st, w.ray32, w.tile_count, f.n_rays, w.tile_stride, searched);
            hipLaunchKernelGGL(k_scan_tiles, dim3((unsigned)searched.count), dim3(1024), 0, st, w.tile_count, (int)n_tiles, w.tile_stride, searched);
            hipLaunchKernelGGL(k_scatter_active, dim3(n_tiles, (unsigned)searched.count), dim3(SEARCH_THREADS), 0, st, w.ray32, w.tile_count, w.active_rid, f.n_rays, w.tile_stride, w.list_stride, searched);
            total = w.tile_count + n_tiles; total_stride = w.tile_stride;
        }
        // (the spill lists of all the group's bundles emptied by one command, not one each)
        HIP_TRY(hipMemsetAsync(w.spill.count, 0, LIGHT_GROUP * sizeof(uint32_t), st));
        for (int kind = 1; kind <= 2; kind++)
            if (of_kind[kind].count) {
                int rc2 = launch_search(kind, of_kind[kind], w.active_rid, total, total_stride, true);
                if (rc2) return rc2;
            }
        return RT355_OK;
    };
    // the lights, in groups of `slots`
    auto group_of = [&](int j) {
        Group gr;
        memset(&gr, 0, sizeof gr);
        gr.k0 = j * slots;
        gr.count = std::max(0, std::min(slots, nl - gr.k0));
        for (int g = 0; g < gr.count; g++) gr.b[g] = bundles[1 + gr.k0 + g];
        return gr;
    };
    auto clear_bins = [&](const Group &gr) -> int { // the next kernel counts the group's rays into the bins
        bool any = false;
        for (int g = 0; g < gr.count; g++) any = any || gr.b[g].binned;
        if (any) HIP_TRY(hipMemsetAsync(w.bin_hist, 0, (size_t)gr.count * HIST_STRIDE * sizeof(uint32_t), st));
        return RT355_OK;
    };
    if (!can_search) { // nothing to search: the whole pixel in one kernel
        if (part.first) HIP_TRY(hipEventRecord(sc->bk().ev_resolve[0], st));
        const Gather ga{sc->d_dirs, sc->n_dirs};
        const SceneView &v = sc->view;
        const bool few = v.n_planes <= GATHER_FEW && v.n_spheres <= GATHER_FEW && v.n_planes + v.n_spheres == v.n_prims &&
                         !sc->knobs.gather_general;
        // (the renderers whose pixels differ most in what they cost -- trees of refracted rays, chains of mirror rays -- go in
        // one-wave workgroups: a place on a SIMD is then free as soon as its wave is done, not when the slowest of four is)
        const unsigned db = (unsigned)sc->knobs.direct_block, db_blocks = (unsigned)((f.grid_w + (int)db - 1) / (int)db) * (unsigned)f.grid_h;
#define LAUNCH_GATHER(B, F) hipLaunchKernelGGL((k_render_gather<B, F>), dim3(direct_blocks), dim3(256), 0, st, sc->view, sh, dc, f, w, ga, pix)
        if (opts->shading == RT355_SHADING_BLEED) {
            if (few) LAUNCH_GATHER(true, GATHER_FEW); else LAUNCH_GATHER(true, 0);
        } else if (opts->shading == RT355_SHADING_OCCLUSION) {
            if (few) LAUNCH_GATHER(false, GATHER_FEW); else LAUNCH_GATHER(false, 0);
        }
#undef LAUNCH_GATHER
#define LAUNCH_DIRECT(R, E) hipLaunchKernelGGL((k_render_direct<R, E>), dim3(R ? db_blocks : direct_blocks), dim3(R ? db : 256u), 0, st, sc->view, sh, dc, f, w, pix)
        else if (refract) {
            // scenes of many primitives find their candidates through the scene table (not in mode BRUTE, and not
            // from the entry point that has no workspace)
            const bool walk = sc->has_general && sc->view.n_prims >= SEARCH_MIN_PRIMS && opts->mode != RT355_MODE_BRUTE && !noalloc;
            if (walk) HIP_TRY(hipMemsetAsync(w.spill.count, 0, sizeof(uint32_t), st));
#define LAUNCH_REFRACT(S, W) hipLaunchKernelGGL((k_render_refract<S, W>), dim3(db_blocks), dim3(db), 0, st, sc->view, sh, dc, f, w, sc->general, pix)
            if (soft) { if (walk) LAUNCH_REFRACT(true, true); else LAUNCH_REFRACT(true, false); }
            else { if (walk) LAUNCH_REFRACT(false, true); else LAUNCH_REFRACT(false, false); }
#undef LAUNCH_REFRACT
        } else if (soft) {
            hipLaunchKernelGGL((k_render_direct<true, true, true>), dim3(direct_blocks), dim3(256), 0, st, sc->view, sh, dc, f, w, pix);
        } else if (sc->reflective) {
            if (sc->view.n_envs + sc->view.n_quadrics > 0) LAUNCH_DIRECT(true, true); else LAUNCH_DIRECT(true, false);
        } else {
            if (sc->view.n_envs + sc->view.n_quadrics > 0) LAUNCH_DIRECT(false, true); else LAUNCH_DIRECT(false, false);
        }
#undef LAUNCH_DIRECT
        if (part.last) HIP_TRY(hipEventRecord(sc->bk().ev_resolve[1], st));
    }
    int resolves = 0;
    // The last stage in parts (see TailParts): part p covers tile rows [tiles_y p / n, tiles_y (p + 1) / n).  Not for
    // supersampled frames (a pixel's sub-samples span tile rows), shares of interleaved tiles, or the gather shaders
    // (whose last stage is a blend that costs nothing).
    const int tail_parts = (tail && can_search && !gather_searched && !refract_children && f.alias == 0 && f.tile_count <= 1)
                               ? std::max(1, std::min(tail->n_parts, f.tiles_y)) : 1;
    bool tail_done = false;
    // the part of the last turn, whose transfer no kernel overlaps, is half as high as the others
    const bool tail_top_first = !sc->knobs.tail_bottom_first;
    auto tail_tile_row = [&](int p) {
        if (tail_parts == 1 || sc->knobs.tail_even) return (int)((int64_t)f.tiles_y * p / tail_parts);
        const int64_t halves = 2 * tail_parts - 1;
        const int64_t before = tail_top_first ? std::max(0, 2 * p - 1) : std::min<int64_t>(2 * p, halves); // half-heights below part p
        return (int)(f.tiles_y * before / halves);
    };
    auto tail_first_rid = [&](int p) { return (uint32_t)tail_tile_row(p) * (uint32_t)f.tiles_x * (uint32_t)RAYS_PER_TILE; };
    // fold and pixels of part p, then its rows handed to the caller
    // The parts are taken from the top of the frame down: the transfers are the longer side of most frames' last
    // stage, so it ends soonest when they start soonest -- with the cheapest part first, and where a frame has sky it
    // has it at the top.
    auto tail_part_of_turn = [&](int turn) { return tail_top_first ? tail_parts - 1 - turn : turn; };
    auto tail_finish = [&](int p, bool last_turn) -> int {
        const int ya = tail_tile_row(p) * TILE_H, yb = std::min(f.rows, tail_tile_row(p + 1) * TILE_H);
        if (ya >= yb) return RT355_OK; // (a frame of fewer tile rows than parts)
        Frame fp = f;
        fp.rid_base = tail_first_rid(p);
        fp.avg_row0 = ya;
        if (levels)
            hipLaunchKernelGGL(k_fold_levels, dim3((tail_first_rid(p + 1) - fp.rid_base) / 256u), dim3(256), 0, st, fp, w, pix);
        const bool direct = last_turn && tail->last_direct;
        hipLaunchKernelGGL(k_average, dim3((unsigned)((f.width + 255) / 256), (unsigned)(yb - ya)), dim3(256), 0, st, fp, w.rec_sum,
                           direct ? tail->last_direct : pix);
        return direct ? RT355_OK : tail->copy_rows(ya, yb);
    };
    // one level of the frame: the search of its rays (camera rays: the eye's tables; mirror rays: the scene table),
    // the fp64 closest hit and hit record, then the lights group by group
    auto run_level = [&](int L) -> int {
        Frame fl = f;
        fl.level = levels ? L : -1;
        if (L == 0) {
            if (bundles[0].kind) {
                // camera rays: workspace slot 0, the last table slot (so that the lights' tables survive the frame)
                Slots eye{};
                eye.count = 1;
                eye.table[0] = sc->table_slots - 1;
                hipLaunchKernelGGL(k_gen_primary, dim3(per_ray_blocks), dim3(256), 0, st, dc, f, w.ray32);
                Group eye_group;
                memset(&eye_group, 0, sizeof eye_group);
                eye_group.count = 1; eye_group.b[0] = bundles[0];
                TableMap eye_map{};
                eye_map.slot[0] = sc->table_slots - 1;
                int rc1 = build_tables(eye_group, eye_map);
                if (rc1) return rc1;
                if ((rc1 = launch_search(1, eye, nullptr, nullptr, 0))) return rc1;
            }
        } else {
            HIP_TRY(hipMemsetAsync(w.spill.count, 0, sizeof(uint32_t), st));
            HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches], st));
            Slots one{}; // the mirror rays of this level, compacted (slot 0's lists)
            one.count = 1;
            hipLaunchKernelGGL(k_count_active, dim3(n_tiles, 1), dim3(SEARCH_THREADS), 0, st, w.ray32, w.tile_count, f.n_rays, w.tile_stride, one);
            hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, w.tile_count, (int)n_tiles, w.tile_stride, one);
            hipLaunchKernelGGL(k_scatter_active, dim3(n_tiles, 1), dim3(SEARCH_THREADS), 0, st, w.ray32, w.tile_count, w.active_rid, f.n_rays, w.tile_stride, w.list_stride, one);
            // (one-wave workgroups: the walks of a wave's rays differ a great deal in length, and a place on a SIMD is free
            // as soon as its wave is done -- the three searches of the mirror frame 2.00 -> 1.87 ms)
            hipLaunchKernelGGL(k_search_general, dim3(per_ray_blocks * 4u), dim3(64), 0, st, sc->general, sc->view, w.ray_d, w.active_rid,
                               w.tile_count + n_tiles, f.n_rays, w.cand, w.cand_cnt, w.spill, w.counters, w.lvl_live + L);
            HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches + 1], st));
            sc->bk().launch_choice.push_back({-1, 0});
            launches++;
        }
        Group cur = group_of(0);
        int rc1 = clear_bins(cur);
        if (rc1) return rc1;
        HIP_TRY(hipMemsetAsync(w.far_n, 0, sizeof(uint32_t), st)); // (the far hits of this level: listed by the kernel below)
        HIP_TRY(hipEventRecord(sc->bk().ev_resolve[2 * resolves], st));
        {
            const int brute_l = (L == 0 && !bundles[0].kind) ? 1 : 0;
            const bool env = sc->view.n_envs + sc->view.n_quadrics > 0;
            // (a level after the first: larger workgroups, see Work::dense_rid)
            const unsigned group = (levels && L > 0) ? LEVEL_GROUP : 256u;
#define LAUNCH_RESOLVE(LV, E) hipLaunchKernelGGL((k_resolve_primary<LV, E>), dim3((f.n_rays + group - 1) / group), dim3(group), 0, st, sc->view, sh, dc, fl, w, brute_l, cur, pix)
            if (levels) { if (env) LAUNCH_RESOLVE(true, true); else LAUNCH_RESOLVE(true, false); }
            else { if (env) LAUNCH_RESOLVE(false, true); else LAUNCH_RESOLVE(false, false); }
#undef LAUNCH_RESOLVE
        }
        HIP_TRY(hipEventRecord(sc->bk().ev_resolve[2 * resolves + 1], st));
        resolves++;
        for (int j = 0; j < n_groups; j++) {
            if ((rc1 = search_group(cur))) return rc1;
            Group next = group_of(j + 1);
            if ((rc1 = clear_bins(next))) return rc1;
            HIP_TRY(hipEventRecord(sc->bk().ev_resolve[2 * resolves], st));
            hipLaunchKernelGGL(k_shade_far, dim3(std::min(per_ray_blocks, 2048u)), dim3(256), 0, st, sc->view, sh, fl, w, cur);
            // the frame's last shading launch goes part by part when the caller reads the parts back meanwhile (the
            // time recorded for it then includes the parts' folds and k_average launches)
            const bool in_parts = tail_parts > 1 && L == extra_levels && j == n_groups - 1;
            // (the rays of a level after the first are numbered densely, not by where they are on screen: such a
            // level is shaded in one launch, and only the fold and the pixels come part by part)
            const bool shade_in_parts = in_parts && L == 0;
            for (int turn = 0; turn < (shade_in_parts ? tail_parts : 1); turn++) {
                const int p = tail_part_of_turn(turn);
                Frame fp = fl;
                unsigned blocks = per_ray_blocks;
                if (shade_in_parts) {
                    fp.rid_base = tail_first_rid(p);
                    blocks = (tail_first_rid(p + 1) - fp.rid_base) / 256u;
                    if (blocks == 0) continue; // (a frame of fewer tile rows than parts)
                }
                const unsigned sb = L == 0 ? (unsigned)sc->knobs.shade_block : 256u; // (the levels of mirror rays: 1 % slower in one-wave workgroups)
                blocks *= 256u / sb;
                if (soft)
                    hipLaunchKernelGGL((k_shade_group<true, true>), dim3(blocks), dim3(sb), 0, st, sc->view, sh, fp, w, cur, next, pix);
                else if (sc->view.n_envs + sc->view.n_quadrics > 0)
                    hipLaunchKernelGGL(k_shade_group<true>, dim3(blocks), dim3(sb), 0, st, sc->view, sh, fp, w, cur, next, pix);
                else
                    hipLaunchKernelGGL(k_shade_group<false>, dim3(blocks), dim3(sb), 0, st, sc->view, sh, fp, w, cur, next, pix);
                if (shade_in_parts && (rc1 = tail_finish(p, turn == tail_parts - 1))) return rc1;
            }
            for (int turn = 0; in_parts && !shade_in_parts && turn < tail_parts; turn++)
                if ((rc1 = tail_finish(tail_part_of_turn(turn), turn == tail_parts - 1))) return rc1;
            if (in_parts) tail_done = true;
            HIP_TRY(hipEventRecord(sc->bk().ev_resolve[2 * resolves + 1], st));
            resolves++;
            cur = next;
        }
        return RT355_OK;
    };
    if (gather_searched) {
        // hits of the camera rays (no lights on this path: hit records only), then one pass per gather direction
        ShaderView no_lights = sh;
        no_lights.n_lights = 0;
        Frame fl = f;
        fl.level = 0;
        if (bundles[0].kind) {
            Slots eye{};
            eye.count = 1;
            eye.table[0] = sc->table_slots - 1;
            hipLaunchKernelGGL(k_gen_primary, dim3(per_ray_blocks), dim3(256), 0, st, dc, f, w.ray32);
            Group eye_group;
            memset(&eye_group, 0, sizeof eye_group);
            eye_group.count = 1; eye_group.b[0] = bundles[0];
            TableMap eye_map{};
            eye_map.slot[0] = sc->table_slots - 1;
            if ((rc = build_tables(eye_group, eye_map))) return rc;
            if ((rc = launch_search(1, eye, nullptr, nullptr, 0))) return rc;
        }
        Group none;
        memset(&none, 0, sizeof none);
        HIP_TRY(hipEventRecord(sc->bk().ev_resolve[0], st));
        hipLaunchKernelGGL((k_resolve_primary<true, true>), dim3(per_ray_blocks), dim3(256), 0, st, sc->view, no_lights, dc, fl, w,
                           bundles[0].kind ? 0 : 1, none, pix);
        HIP_TRY(hipEventRecord(sc->bk().ev_resolve[1], st));
        resolves = 1;
        const Gather ga{sc->d_dirs, sc->n_dirs};
        const int walk_blocks = (int)sc->knobs.walk_blocks; // waves that stay, four to a workgroup: several times what the chip holds at once, so that the last ones even out
        if ((rc = ensure_gather(sc, (size_t)(sc->n_dirs + 63) / 64))) return rc; // (w refers to sc->work)
        hipLaunchKernelGGL(k_gather_init, dim3(per_ray_blocks), dim3(256), 0, st, f, w, ga);
        const bool bleed = opts->shading == RT355_SHADING_BLEED;
        HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches], st));
        if (bleed) hipLaunchKernelGGL(k_gather_walk<true>, dim3((unsigned)walk_blocks), dim3(256), 0, st, sc->general, sc->view, f, w, ga);
        else hipLaunchKernelGGL(k_gather_walk<false>, dim3((unsigned)walk_blocks), dim3(256), 0, st, sc->general, sc->view, f, w, ga);
        HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches + 1], st)); // all passes as one entry of the search times
        sc->bk().launch_choice.push_back({-1, 0});
        launches++;
        if (bleed) hipLaunchKernelGGL(k_gather_finish<true>, dim3(per_ray_blocks), dim3(256), 0, st, sc->view, sh, f, w, ga, pix);
        else hipLaunchKernelGGL(k_gather_finish<false>, dim3(per_ray_blocks), dim3(256), 0, st, sc->view, sh, f, w, ga, pix);
    } else if (can_search) {
        if ((rc = run_level(0))) return rc;
        for (int L = 0; L < extra_levels; L++) {
            // (after the first level only the chains that go on write their slot-0 record: the others' must read "no ray")
            if (L > 0) HIP_TRY(hipMemsetAsync(w.ray32, 0xff, (size_t)f.n_rays * sizeof(float4), st));
            hipLaunchKernelGGL(k_reflect_rays, dim3(per_ray_blocks), dim3(256), 0, st, sc->view, dc, f, w, L);
            if ((rc = run_level(L + 1))) return rc;
        }
        if (levels && !tail_done) hipLaunchKernelGGL(k_fold_levels, dim3(per_ray_blocks), dim3(256), 0, st, f, w, pix);
        if (refract_children) { // the rays whose first call of CalcSubPixel has children: their trees, from the hit on
            HIP_TRY(hipMemsetAsync(w.spill.count, 0, LIGHT_GROUP * sizeof(uint32_t), st));
            HIP_TRY(hipMemsetAsync(w.far_n, 0, sizeof(uint32_t), st));
            HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches], st));
            hipLaunchKernelGGL(k_list_children, dim3(per_ray_blocks), dim3(256), 0, st, sc->view, f, w);
            const unsigned cb = (unsigned)sc->knobs.direct_block;
            if (sc->work_slots >= REFRACT_GROUP && !sc->knobs.refract_single) // a group of lanes per listed ray (blocks past the list leave at once)
                hipLaunchKernelGGL((k_render_refract<false, true, true, REFRACT_GROUP>), dim3(per_ray_blocks * REFRACT_GROUP * (256u / cb)), dim3(cb), 0, st, sc->view, sh, dc, f, w, sc->general, pix);
            else
                hipLaunchKernelGGL((k_render_refract<false, true, true>), dim3(per_ray_blocks * (256u / cb)), dim3(cb), 0, st, sc->view, sh, dc, f, w, sc->general, pix);
            HIP_TRY(hipEventRecord(sc->bk().ev_search[2 * launches + 1], st));
            sc->bk().launch_choice.push_back({-1, 0});
            launches++;
        }
    }
    if ((f.alias > 0 || f.deferred) && !tail_done)
        hipLaunchKernelGGL(k_average, dim3((unsigned)((f.width + 255) / 256), (unsigned)f.rows), dim3(256), 0, st, f, w.rec_sum, pix);
    if (tail && !tail_done && (rc = tail->copy_rows(0, f.rows))) return rc; // (a frame whose last stage is one launch: all rows at once)
    HIP_TRY(hipGetLastError());
    sc->bk().pending_primary += (uint64_t)f.grid_w * (uint64_t)f.grid_h;
    if (!part.last) return RT355_OK;
    HIP_TRY(hipEventRecord(sc->bk().ev_end, st));
    HIP_TRY(hipMemcpyAsync(sc->bk().host_counters, w.counters, COUNTER_SHARDS * COUNTER_STRIDE * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(sc->bk().ev_done, st));
    sc->bk().pending = true;
    sc->bk().pending_stats = stats;
    sc->bk().pending_stream = st;
    sc->bk().pending_nprims = sc->view.n_prims;
    sc->bk().search_launches = launches;
    sc->bk().resolve_launches = can_search ? resolves : 1;
    return RT355_OK;
}

#define MULTI_HAS_NO(sc, what) \
    if ((sc) && (sc)->multi) return fail(RT355_ERR_UNSUPPORTED, what ": a device pixmap lies on one GPU; a multi-device scene renders into host memory (rt355_render, rt355_render_sweep)")
extern "C" int rt355_render_device(rt355_scene *sc, const rt355_camera *cam, const rt355_shader *shader,
                                   const rt355_opts *opts, void *d_pixmap, void *hip_stream, rt355_stats *stats) {
    MULTI_HAS_NO(sc, "rt355_render_device");
    return render_enqueue(sc, cam, shader, opts, d_pixmap, hip_stream, stats, false);
}
extern "C" int rt355_render_device_noalloc(rt355_scene *sc, const rt355_camera *cam, const rt355_shader *shader,
                                           const rt355_opts *opts, void *d_pixmap, void *hip_stream, rt355_stats *stats) {
    MULTI_HAS_NO(sc, "rt355_render_device_noalloc");
    return render_enqueue(sc, cam, shader, opts, d_pixmap, hip_stream, stats, true);
}

// Wait for the frame of one set of the bookkeeping and turn its counters and events into the caller's statistics.
static int sync_book(rt355_scene *sc, FrameBook &bk) {
    if (!bk.pending) return RT355_OK;
    HIP_TRY(hipSetDevice(sc->device));
    HIP_TRY(hipEventSynchronize(bk.ev_done));
    bk.pending = false;
    const unsigned long long *shards = bk.host_counters;
    unsigned long long ctr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < COUNTER_SHARDS; i++)
        for (int j = 0; j < 8; j++) ctr[j] += shards[i * COUNTER_STRIDE + j];
    unsigned long long too_deep = 0;
    for (int i = 0; i < COUNTER_SHARDS; i++) too_deep += shards[i * COUNTER_STRIDE + 14];
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, bk.ev_begin, bk.ev_end));
    sc->search_ms = 0;
    sc->launch_ms.assign(bk.search_launches, 0.0);
    for (int i = 0; i < bk.search_launches; i++) {
        float m = 0;
        HIP_TRY(hipEventElapsedTime(&m, bk.ev_search[2 * i], bk.ev_search[2 * i + 1]));
        sc->search_ms += m;
        sc->launch_ms[i] = m;
    }
    { // what this frame's packet launches cost, per kind of launch (a frame can have several groups of lights)
        double sum[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        bool used[3][2] = {{false, false}, {false, false}, {false, false}};
        for (int i = 0; i < bk.search_launches && i < (int)bk.launch_choice.size(); i++) {
            const int cls = bk.launch_choice[i].first, which = bk.launch_choice[i].second == 4 ? 0 : 1;
            if (cls < 0) continue;
            sum[cls][which] += sc->launch_ms[i];
            used[cls][which] = true;
        }
        for (int c = 0; c < 3; c++)
            for (int k = 0; k < 2; k++)
                if (used[c][k]) sc->tune_ms[c][k] = sum[c][k];
    }
    sc->resolve_ms = 0;
    sc->resolve_launch_ms.assign(bk.resolve_launches, 0.0);
    for (int i = 0; i < bk.resolve_launches; i++) {
        float m = 0;
        HIP_TRY(hipEventElapsedTime(&m, bk.ev_resolve[2 * i], bk.ev_resolve[2 * i + 1]));
        sc->resolve_ms += m;
        sc->resolve_launch_ms[i] = m;
    }
    sc->last_ms = ms;
    sc->last_search_launches = bk.search_launches;
    sc->last_resolve_launches = bk.resolve_launches;
    if (bk.pending_stats) {
        rt355_stats *s = bk.pending_stats;
        s->primary_rays = bk.pending_primary;
        s->shadow_rays = ctr[0];
        s->exact_tests = ctr[1];
        s->overflow_rays = ctr[2];
        s->pair_tests = ctr[5];
        s->ms_device = ms;
        s->packet_survivors = ctr[3];
        s->primary_hits = ctr[4];
        s->secondary_rays = ctr[6];
        s->gather_rays = ctr[7];
#ifdef RT355_DEBUG_SURVIVORS
        { unsigned long long used = 0; for (int i = 0; i < COUNTER_SHARDS; i++) used += shards[i * COUNTER_STRIDE + 13];
          fprintf(stderr, "[rt355] shadow candidate loops: %llu lane slots for %llu candidates (%.1f %% used)\n", shards[12], used, 100.0 * (double)used / (double)std::max(1ull, shards[12])); }
        fprintf(stderr, "[rt355] packet search: %llu waves, per wave %.2f superchunks, %.2f chunks, %.2f primitives met (max %llu)\n",
                shards[9], (double)shards[10] / (double)std::max(1ull, shards[9]), (double)shards[11] / (double)std::max(1ull, shards[9]),
                (double)ctr[3] / (double)std::max(1ull, shards[9]), shards[8]);
#endif
    }
    if (too_deep)
        return fail(RT355_ERR_UNSUPPORTED, std::to_string(too_deep) + " pixel(s) whose chain of refracted rays did not end within " +
                    std::to_string(REFRACT_OPS) + " waiting calls of CalcSubPixel (main.cpp:126,138,148 bound nothing: the reference's recursion would not end either)");
    return RT355_OK;
}

extern "C" int rt355_scene_sync(rt355_scene *sc) {
    if (!sc) return fail(RT355_ERR_ARG, "rt355_scene_sync: null scene");
    if (sc->multi) return RT355_OK; // (its calls return when the frame is in the caller's memory)
    // the older frame first: after a sweep the set the next enqueue would use holds it
    const int rc_old = sync_book(sc, sc->book[sc->cur]);
    const int rc_new = sync_book(sc, sc->book[sc->cur ^ 1]);
    return rc_old != RT355_OK ? rc_old : rc_new;
}

extern "C" int rt355_scene_search_kernel_ms(rt355_scene *sc, double *ms_sum, int32_t *launches) {
    if (sc && sc->multi) return rt355_scene_search_kernel_ms(sc->multi->scenes[0], ms_sum, launches); // (the first device's share)
    if (!sc || !ms_sum || !launches) return fail(RT355_ERR_ARG, "null argument");
    *ms_sum = sc->search_ms;
    *launches = sc->last_search_launches;
    return RT355_OK;
}


extern "C" int rt355_scene_resolve_kernel_ms(rt355_scene *sc, double *ms_sum, int32_t *launches) {
    if (sc && sc->multi) return rt355_scene_resolve_kernel_ms(sc->multi->scenes[0], ms_sum, launches); // (the first device's share)
    if (!sc || !ms_sum || !launches) return fail(RT355_ERR_ARG, "null argument");
    *ms_sum = sc->resolve_ms;
    *launches = sc->last_resolve_launches;
    return RT355_OK;
}

extern "C" int rt355_scene_resolve_launch_ms(rt355_scene *sc, int32_t index, double *ms) {
    if (sc && sc->multi) return rt355_scene_resolve_launch_ms(sc->multi->scenes[0], index, ms); // (the first device's share)
    if (!sc || !ms || index < 0 || index >= (int32_t)sc->resolve_launch_ms.size()) return fail(RT355_ERR_ARG, "bad launch index");
    *ms = sc->resolve_launch_ms[index];
    return RT355_OK;
}

extern "C" int rt355_scene_search_launch_ms(rt355_scene *sc, int32_t index, double *ms) {
    if (sc && sc->multi) return rt355_scene_search_launch_ms(sc->multi->scenes[0], index, ms); // (the first device's share)
    if (!sc || !ms || index < 0 || index >= (int32_t)sc->launch_ms.size()) return fail(RT355_ERR_ARG, "bad launch index");
    *ms = sc->launch_ms[index];
    return RT355_OK;
}

// Page-locked host memory for the pixmap of rt355_render.
extern "C" void *rt355_host_alloc(uint64_t bytes) {
    void *p = nullptr;
    // page-locked for every GPU of the process and mapped into their address spaces: the GPUs of a multi-device scene
    // all write their rows of a frame into the one pixmap
    const unsigned flags = hipHostMallocPortable | hipHostMallocMapped;
    if (hipHostMalloc(&p, (size_t)std::max<uint64_t>(bytes, 1), flags) != hipSuccess) {
        (void)hipGetLastError();
        g_err = "rt355_host_alloc: hipHostMalloc failed";
        return nullptr;
    }
    return p;
}
extern "C" void rt355_host_free(void *p) { if (p) (void)hipHostFree(p); }
extern "C" int rt355_host_register(void *p, uint64_t bytes) {
    if (!p || !bytes) return fail(RT355_ERR_ARG, "rt355_host_register: null pointer or no bytes");
    HIP_TRY(hipHostRegister(p, (size_t)bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return RT355_OK;
}
extern "C" int rt355_host_unregister(void *p) {
    if (!p) return fail(RT355_ERR_ARG, "rt355_host_unregister: null pointer");
    HIP_TRY(hipHostUnregister(p));
    return RT355_OK;
}
extern "C" void *rt355_host_device_pointer(void *p) {
    void *mapped = nullptr;
    if (!p || hipHostGetDevicePointer(&mapped, p, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return mapped;
}

static bool host_is_pinned(const void *p) { // page-locked memory the runtime knows (hipHostMalloc, hipHostRegister)
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeHost;
}

#define RENDER_PARTS_MAX 8
// The streams and events of the readback (rt355_render, rt355_render_sweep), made at a handle's first such call.
static int ensure_readback(rt355_scene *sc) {
    const bool first_call = !sc->copy_stream2;
    if (!sc->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&sc->copy_stream, hipStreamNonBlocking));
    if (!sc->copy_stream2) HIP_TRY(hipStreamCreateWithFlags(&sc->copy_stream2, hipStreamNonBlocking));
    while (sc->ev_copy.size() < RENDER_PARTS_MAX) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sc->ev_copy.push_back(e);
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        sc->ev_part.push_back(e);
    }
    if (first_call && !sc->knobs.no_copy_prime && sc->fb) {
        // The runtime grows what it keeps per transfer in flight the first time a process has had some dozens of them
        // (one call of 6 ms somewhere among a handle's first ten frames, measured); a burst of small transfers between
        // events, as a frame's parts make them, has that happen here, in the call that sets the handle up anyway.
        if (any_pending(sc)) { int rc = rt355_scene_sync(sc); if (rc) return rc; } // (host_counters is the burst's target)
        const int n = 96;
        for (int i = 0; i < n; i++) {
            hipStream_t cs = (i & 1) ? sc->copy_stream2 : sc->copy_stream;
            HIP_TRY(hipEventRecord(sc->ev_part[i % RENDER_PARTS_MAX], nullptr));
            HIP_TRY(hipStreamWaitEvent(cs, sc->ev_part[i % RENDER_PARTS_MAX], 0));
            HIP_TRY(hipMemcpyAsync(sc->bk().host_counters, sc->fb, std::min<size_t>(64, sc->fb_bytes), hipMemcpyDeviceToHost, cs));
            HIP_TRY(hipEventRecord(sc->ev_copy[i % RENDER_PARTS_MAX], cs));
        }
        HIP_TRY(hipStreamSynchronize(sc->copy_stream));
        HIP_TRY(hipStreamSynchronize(sc->copy_stream2));
    }
    return RT355_OK;
}

// setPixels' replacement (main.cpp:339-376): render into the handle's device framebuffer, read back into the
// caller's pixmap.  The frame is rendered in a few parts (whole rows), and the readback of a part runs on a
// second stream while the next part renders; a page-locked pixmap is the target of the DMA itself, any other
// is filled from page-locked staging memory, the host copy of a part overlapping the transfer of the next.
// Nothing is allocated per frame: framebuffer and staging grow when a larger frame comes and are kept.
#define READBACK_PROBES 6
#define READBACK_RETRY 32
extern "C" int rt355_render(rt355_scene *sc, const rt355_camera *cam, const rt355_shader *shader,
                            const rt355_opts *opts, unsigned char *pixmap, rt355_stats *stats) {
    if (!sc || !opts || !pixmap) return fail(RT355_ERR_ARG, "rt355_render: null argument");
    if (sc->multi) return multi_render(sc, cam, shader, opts, pixmap, stats);
    if (opts->width <= 0 || opts->height <= 0) return fail(RT355_ERR_ARG, "width and height must be positive");
    const auto t_call = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(sc->device));
    Rows rows;
    { int rc0 = rows_of(opts, &rows); if (rc0) return rc0; }
    const size_t row_bytes = (size_t)opts->width * 3, bytes = row_bytes * (size_t)opts->height;
    if (bytes > sc->fb_bytes) {
        if (sc->fb) { (void)hipFree(sc->fb); sc->fb = nullptr; sc->fb_bytes = 0; }
        HIP_TRY(hipMalloc((void **)&sc->fb, bytes));
        sc->fb_bytes = bytes;
    }
    { int rc0 = ensure_readback(sc); if (rc0) return rc0; }
    const bool pinned = host_is_pinned(pixmap);
    const bool refract = sc->refractive && (opts->shading == RT355_SHADING_LIGHTS || opts->shading == RT355_SHADING_SOFT);
    const bool searched = opts->mode != RT355_MODE_BRUTE && sc->view.n_prims >= SEARCH_MIN_PRIMS && (!refract || searched_with_glass(sc, opts));
    const bool one_kernel = opts->tile_count <= 1 && !searched;
    // a searched frame whose last stage can be launched in parts, each read back while the next is shaded (TailParts)
    const bool tail_way = searched && opts->tile_count <= 1 && opts->alias == 0 && !sc->knobs.no_tail_parts &&
                          (opts->shading == RT355_SHADING_LIGHTS || opts->shading == RT355_SHADING_SOFT);
    // A page-locked pixmap is mapped into the device's address space, and the kernels that write the pixels can
    // write them there, whole dwords in row order: the transfer over PCIe then runs while the frame is rendered
    // and no readback step is left.  Measured on MI355X at 3840x2160: k_average streams a searched frame's pixels
    // out at the 55 GB/s the copy engine reaches; a one-kernel frame's own stores reach 40 GB/s, which beats
    // rendering and copying in parts when the kernel runs longer than the transfer (glass room 0.97 against 1.14
    // ms, mirrors 1.11 against 1.18) and loses when it is shorter (the default scene 0.61 against 0.53).  A searched
    // frame's last stage in four parts, each copied while the next is shaded, beats k_average writing the whole frame
    // across at the end (10 000 spheres 3.44 against 3.67 ms, the 100 000-triangle mesh 1.49 against 1.75).  Which it
    // is the handle finds out by timing its calls on the host: the first READBACK_PROBES frames of a kind (size,
    // shader, lights) go direct, copied, copied, copied, direct, direct -- the first direct write is what makes the
    // device map the pixmap's pages (copies alone take eight calls to reach their speed: 20, 0.8, 5.8, 0.7 ... 0.53 ms
    // for the default scene; after one direct frame the second copied frame has it), and two direct frames in a row
    // show what those cost back to back --, the frames after them go the way whose shortest of its last three calls
    // was shorter, and every READBACK_RETRY-th frame the other way once more
    // (what the link to the host gives changes while a process runs).  The event the call waits on orders the
    // writes before the caller's reads.  (The handle's option "readback" forces one way: rt355_scene_set_option.)
    bool direct = pinned, timing = false;
    if (pinned && (one_kernel || tail_way)) {
        uint64_t key = 0x9e3779b97f4a7c15ull;
        for (uint64_t v : {(uint64_t)opts->width, (uint64_t)opts->height, (uint64_t)rows.y0, (uint64_t)rows.y1, (uint64_t)opts->alias,
                           (uint64_t)opts->shading, (uint64_t)opts->mode, (uint64_t)(shader ? shader->n_lights : 0), (uint64_t)sc->n_dirs})
            key = (key ^ v) * 0x100000001b3ull;
        key |= 1ull;
        if (sc->readback_key != key) { sc->readback_key = key; sc->readback_frames = 0; sc->readback_calls[0] = sc->readback_calls[1] = 0; }
        auto recent = [&](int way) { // the shortest of the way's last three calls: one slow call (a first use, a busy host) decides nothing
            double best = 1e30;
            for (uint32_t i = 0; i < std::min(sc->readback_calls[way], 3u); i++) best = std::min(best, sc->readback_ms[way][i]);
            return best;
        };
        const bool faster_direct = recent(0) <= recent(1);
        if (sc->readback_frames < READBACK_PROBES) direct = sc->readback_frames == 0 || sc->readback_frames >= READBACK_PROBES - 2;
        else direct = (sc->readback_frames % READBACK_RETRY == 0) ? !faster_direct : faster_direct;
        timing = true;
    }
    if (sc->knobs.readback == 2) { direct = false; timing = false; }
    if (sc->knobs.readback == 1) { direct = pinned; timing = false; }
    auto timed = [&](int rc) { // a call has ended
        if (!timing) return rc;
        if (rc != RT355_OK) { sc->readback_key = 0; return rc; }
        const int way = direct ? 0 : 1;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        sc->readback_ms[way][sc->readback_calls[way]++ % 3u] = ms;
        sc->readback_frames++;
        sc->debug_ms[way] += ms;
        sc->debug_calls[way]++;
        if (sc->debug_series.capacity() > sc->debug_series.size()) sc->debug_series.push_back((float)(direct ? -ms : ms)); // (reserved when asked for: no allocation here)
        return rc;
    };
    if (direct) {
        void *mapped = nullptr;
        if (hipHostGetDevicePointer(&mapped, pixmap, 0) == hipSuccess && mapped) {
            int rc = render_enqueue(sc, cam, shader, opts, mapped, nullptr, stats, false);
            if (rc == RT355_OK) rc = rt355_scene_sync(sc);
            return timed(rc);
        }
        (void)hipGetLastError();
    }
    if (!pinned && bytes > sc->stage_bytes) {
        if (sc->stage) { (void)hipHostFree(sc->stage); sc->stage = nullptr; sc->stage_bytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&sc->stage, bytes, hipHostMallocDefault));
        sc->stage_bytes = bytes;
    }
    // parts: whole multiples of 32 rows (the tile height of the ray numbering divides it), at least 256 rows each;
    // a call that renders a share of interleaved row tiles is one part
    // a searched frame is a chain of a dozen launches whose cost does not shrink with the rows: it is enqueued once,
    // and only its last stage comes in parts (tail_way below)
    int n_parts = 1;
    if (one_kernel) {
        n_parts = (int)sc->knobs.render_parts;
        n_parts = std::max(1, std::min(n_parts, std::min(RENDER_PARTS_MAX, (rows.y1 - rows.y0) / 128)));
    }
    int part_y0[RENDER_PARTS_MAX + 1];
    for (int i = 0; i <= n_parts; i++) {
        part_y0[i] = rows.y0 + (int)((int64_t)(rows.y1 - rows.y0) * i / n_parts) / 32 * 32;
        if (i == 0) part_y0[i] = rows.y0;
        if (i == n_parts) part_y0[i] = rows.y1;
    }
    unsigned char *host_dst = pinned ? pixmap : sc->stage;
    // the rows a part (or a share's tiles) covers go back as one transfer each, once the part has been rendered
    // the transfers alternate between two streams, so that one's set-up hides behind the other's data, and the last
    // part, which no later kernel would overlap, is written straight into a page-locked pixmap
    const bool two_streams = sc->knobs.copy_streams != 1;
    const bool last_direct = pinned && !sc->knobs.last_copied;
    auto read_back = [&](int ya, int yb, hipEvent_t rendered, hipEvent_t done, hipStream_t copy_stream) -> int {
        HIP_TRY(hipEventRecord(rendered, nullptr));
        HIP_TRY(hipStreamWaitEvent(copy_stream, rendered, 0));
        if (opts->tile_count > 1) {
            for (int k = opts->tile_phase; k * opts->tile_rows < yb - ya; k += opts->tile_count) {
                const int ta = ya + k * opts->tile_rows, tb = std::min(yb, ta + opts->tile_rows);
                HIP_TRY(hipMemcpyAsync(host_dst + ta * row_bytes, sc->fb + ta * row_bytes, (size_t)(tb - ta) * row_bytes, hipMemcpyDeviceToHost, copy_stream));
            }
        } else {
            HIP_TRY(hipMemcpyAsync(host_dst + ya * row_bytes, sc->fb + ya * row_bytes, (size_t)(yb - ya) * row_bytes, hipMemcpyDeviceToHost, copy_stream));
        }
        HIP_TRY(hipEventRecord(done, copy_stream));
        return RT355_OK;
    };
    auto to_pixmap = [&](int ya, int yb) { // staged rows -> the caller's pageable pixmap
        if (opts->tile_count > 1) {
            for (int k = opts->tile_phase; k * opts->tile_rows < yb - ya; k += opts->tile_count) {
                const int ta = ya + k * opts->tile_rows, tb = std::min(yb, ta + opts->tile_rows);
                memcpy(pixmap + ta * row_bytes, sc->stage + ta * row_bytes, (size_t)(tb - ta) * row_bytes);
            }
        } else {
            memcpy(pixmap + ya * row_bytes, sc->stage + ya * row_bytes, (size_t)(yb - ya) * row_bytes);
        }
    };
    // Every part is enqueued at once -- they count into the same statistics -- and each is read back, on the second
    // stream, as soon as its kernel has finished, while the later parts render.
    int rc = RT355_OK;
    int n_copies = 0; // parts whose transfer was enqueued
    int copy_ya[RENDER_PARTS_MAX], copy_yb[RENDER_PARTS_MAX]; // ... and their rows
    if (tail_way) { // one enqueue; its last stage comes in parts, and each part's rows are read back as it completes
        TailParts tp;
        tp.n_parts = (int)sc->knobs.tail_parts;
        tp.n_parts = std::max(1, std::min(tp.n_parts, RENDER_PARTS_MAX));
        void *mapped = nullptr;
        if (!(last_direct && tp.n_parts > 1 && hipHostGetDevicePointer(&mapped, pixmap, 0) == hipSuccess && mapped)) {
            (void)hipGetLastError();
            mapped = nullptr;
        }
        tp.last_direct = (unsigned char *)mapped;
        tp.copy_rows = [&](int ya, int yb) -> int {
            if (n_copies >= RENDER_PARTS_MAX) return fail(RT355_ERR_ARG, "internal: more readback parts than events");
            const int i = n_copies;
            copy_ya[i] = rows.y0 + ya; copy_yb[i] = rows.y0 + yb;
            const int rc1 = read_back(copy_ya[i], copy_yb[i], sc->ev_part[i], sc->ev_copy[i], (two_streams && (i & 1)) ? sc->copy_stream2 : sc->copy_stream);
            if (rc1 == RT355_OK) n_copies = i + 1;
            return rc1;
        };
        rc = render_enqueue(sc, cam, shader, opts, sc->fb, nullptr, stats, false, PartOfFrame{true, true}, &tp);
        n_parts = 0;
    }
    int max_part_rows = 0;
    for (int i = 0; i < n_parts; i++) max_part_rows = std::max(max_part_rows, part_y0[i + 1] - part_y0[i]);
    for (int i = 0; i < n_parts && rc == RT355_OK; i++) {
        rt355_opts part = *opts;
        part.row_begin = part_y0[i]; part.row_end = part_y0[i + 1];
        void *mapped = nullptr;
        if (last_direct && i == n_parts - 1 && n_parts > 1 && hipHostGetDevicePointer(&mapped, pixmap, 0) == hipSuccess && mapped) {
            rc = render_enqueue(sc, cam, shader, &part, mapped, nullptr, stats, false, PartOfFrame{false, true, max_part_rows});
            continue;
        }
        rc = render_enqueue(sc, cam, shader, &part, sc->fb, nullptr, stats, false, PartOfFrame{i == 0, i == n_parts - 1, max_part_rows});
        if (rc == RT355_OK) rc = read_back(part_y0[i], part_y0[i + 1], sc->ev_part[i], sc->ev_copy[i], (two_streams && (i & 1)) ? sc->copy_stream2 : sc->copy_stream);
        if (rc == RT355_OK) { copy_ya[i] = part_y0[i]; copy_yb[i] = part_y0[i + 1]; n_copies = i + 1; }
    }
    for (int i = 0; i < n_copies && rc == RT355_OK && !pinned; i++) { // staged parts -> the pixmap, each as it arrives
        hipError_t e = hipEventSynchronize(sc->ev_copy[i]);
        if (e != hipSuccess) rc = fail(RT355_ERR_DEVICE, std::string("readback: ") + hipGetErrorString(e));
        else to_pixmap(copy_ya[i], copy_yb[i]);
    }
    // (on an error too: no transfer into the caller's memory may outlive the call)
    hipError_t e = hipStreamSynchronize(sc->copy_stream);
    if (e != hipSuccess && rc == RT355_OK) rc = fail(RT355_ERR_DEVICE, std::string("readback: ") + hipGetErrorString(e));
    e = hipStreamSynchronize(sc->copy_stream2);
    if (e != hipSuccess && rc == RT355_OK) rc = fail(RT355_ERR_DEVICE, std::string("readback: ") + hipGetErrorString(e));
    if (any_pending(sc)) { // the frame's statistics, and what its kernels reported
        const int rc_sync = rt355_scene_sync(sc);
        if (rc == RT355_OK) rc = rc_sync;
    }
    return timed(rc);
}

// A camera sweep: setPixels once per camera (main.cpp:640 in a loop over cameras), frame i into pixmaps[i].  Frames
// alternate between two framebuffers in HBM; the rows of frame i travel to the host on the copy stream while frame
// i + 1 is rendered, so that of all the transfers only the last one is not hidden.  A pageable pixmap is filled from
// the page-locked staging memory once its transfer has arrived (the next transfer waits for that copy).
extern "C" int rt355_render_sweep(rt355_scene *sc, const rt355_camera *cams, int32_t n_frames, const rt355_shader *shader,
                                  const rt355_opts *opts, unsigned char *const *pixmaps, rt355_stats *stats) {
    if (!sc || !opts || n_frames < 0 || (n_frames && (!cams || !pixmaps))) return fail(RT355_ERR_ARG, "rt355_render_sweep: bad argument");
    if (sc->multi) return multi_sweep(sc, cams, n_frames, shader, opts, pixmaps, stats);
    if (opts->width <= 0 || opts->height <= 0) return fail(RT355_ERR_ARG, "width and height must be positive");
    for (int i = 0; i < n_frames; i++)
        if (!pixmaps[i]) return fail(RT355_ERR_ARG, "rt355_render_sweep: null pixmap");
    if (n_frames == 0) return RT355_OK;
    HIP_TRY(hipSetDevice(sc->device));
    Rows rows;
    { int rc0 = rows_of(opts, &rows); if (rc0) return rc0; }
    const size_t row_bytes = (size_t)opts->width * 3, bytes = row_bytes * (size_t)opts->height;
    if (any_pending(sc)) { int rc0 = rt355_scene_sync(sc); if (rc0) return rc0; }
    if (bytes > sc->fb_bytes) {
        if (sc->fb) { (void)hipFree(sc->fb); sc->fb = nullptr; sc->fb_bytes = 0; }
        HIP_TRY(hipMalloc((void **)&sc->fb, bytes));
        sc->fb_bytes = bytes;
    }
    if (bytes > sc->fb2_bytes) {
        if (sc->fb2) { (void)hipFree(sc->fb2); sc->fb2 = nullptr; sc->fb2_bytes = 0; }
        HIP_TRY(hipMalloc((void **)&sc->fb2, bytes));
        sc->fb2_bytes = bytes;
    }
    { int rc0 = ensure_readback(sc); if (rc0) return rc0; }
    if (!sc->sweep_stream) HIP_TRY(hipStreamCreateWithFlags(&sc->sweep_stream, hipStreamNonBlocking));
    const hipStream_t st = sc->sweep_stream; // (two handles sweeping from two host threads then overlap on the GPU)
    bool any_pageable = false;
    for (int i = 0; i < n_frames; i++) any_pageable = any_pageable || !host_is_pinned(pixmaps[i]);
    if (any_pageable && bytes > sc->stage_bytes) {
        if (sc->stage) { (void)hipHostFree(sc->stage); sc->stage = nullptr; sc->stage_bytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&sc->stage, bytes, hipHostMallocDefault));
        sc->stage_bytes = bytes;
    }
    // the rows this call renders, as ranges of frame rows (a share of interleaved tiles: one range per tile)
    std::vector<std::pair<int, int>> ranges;
    if (opts->tile_count > 1) {
        for (int k = opts->tile_phase; k * opts->tile_rows < rows.y1 - rows.y0; k += opts->tile_count)
            ranges.push_back({rows.y0 + k * opts->tile_rows, std::min(rows.y1, rows.y0 + (k + 1) * opts->tile_rows)});
    } else {
        ranges.push_back({rows.y0, rows.y1});
    }
    int staged = -1; // the frame whose rows lie in the staging memory, still to be copied into its pageable pixmap
    auto unstage = [&]() -> int {
        if (staged < 0) return RT355_OK;
        hipError_t e = hipEventSynchronize(sc->ev_copy[staged & 1]);
        if (e != hipSuccess) return fail(RT355_ERR_DEVICE, std::string("rt355_render_sweep: ") + hipGetErrorString(e));
        for (const auto &r : ranges)
            memcpy(pixmaps[staged] + r.first * row_bytes, sc->stage + r.first * row_bytes, (size_t)(r.second - r.first) * row_bytes);
        staged = -1;
        return RT355_OK;
    };
    int rc = RT355_OK;
    for (int i = 0; i < n_frames && rc == RT355_OK; i++) {
        unsigned char *fb = (i & 1) ? sc->fb2 : sc->fb;
        // frame i - 2 has left this framebuffer before frame i is rendered into it
        if (i >= 2) {
            hipError_t e = hipStreamWaitEvent(st, sc->ev_copy[i & 1], 0);
            if (e != hipSuccess) { rc = fail(RT355_ERR_DEVICE, std::string("rt355_render_sweep: ") + hipGetErrorString(e)); break; }
        }
        rc = render_enqueue(sc, &cams[i], shader, opts, fb, st, stats ? &stats[i] : nullptr, false); // (waits for frame i - 2's statistics first)
        if (rc != RT355_OK) break;
        sc->cur ^= 1; // frame i + 1 is enqueued while this one runs: it takes the other set of events and counters
        const bool pinned = host_is_pinned(pixmaps[i]);
        if (!pinned && (rc = unstage()) != RT355_OK) break; // the staging memory is free again
        unsigned char *dst = pinned ? pixmaps[i] : sc->stage;
        hipError_t e = hipEventRecord(sc->ev_part[i & 1], st);
        if (e == hipSuccess) e = hipStreamWaitEvent(sc->copy_stream, sc->ev_part[i & 1], 0);
        for (size_t k = 0; k < ranges.size() && e == hipSuccess; k++)
            e = hipMemcpyAsync(dst + ranges[k].first * row_bytes, fb + ranges[k].first * row_bytes,
                               (size_t)(ranges[k].second - ranges[k].first) * row_bytes, hipMemcpyDeviceToHost, sc->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(sc->ev_copy[i & 1], sc->copy_stream);
        if (e != hipSuccess) { rc = fail(RT355_ERR_DEVICE, std::string("rt355_render_sweep: ") + hipGetErrorString(e)); break; }
        if (!pinned) staged = i;
    }
    // (on an error too: no transfer into the caller's memory may outlive the call)
    if (any_pending(sc)) { const int rc_sync = rt355_scene_sync(sc); if (rc == RT355_OK) rc = rc_sync; }
    sc->cur = 0;
    if (rc == RT355_OK) rc = unstage();
    hipError_t e = hipStreamSynchronize(sc->copy_stream);
    if (e != hipSuccess && rc == RT355_OK) rc = fail(RT355_ERR_DEVICE, std::string("rt355_render_sweep: ") + hipGetErrorString(e));
    return rc;
}


// =====================================================================================================
// One scene on several GPUs of this process (rt355_scene_create_multi)
// =====================================================================================================
// The handle holds one ordinary scene per device and a host thread per scene (the render calls block, and a thread's
// current device is its own).  rt355_render deals the 32-row tiles of the caller's slab to the scenes, share i to
// scene i, every GPU writing its rows over its own link into the caller's one pixmap (kernels' own stores for a
// page-locked pixmap, each scene's staging memory and a host copy for a pageable one); rt355_render_sweep deals whole
// frames, frame f to scene f mod N.  With the option "gather" = 1 the shares are rendered into HBM, copied to the
// first device's framebuffer over the GPUs' own links (hipMemcpyPeerAsync) and read back from there in one transfer.
static void multi_worker_main(MultiWorker *w, int device) {
    (void)hipSetDevice(device);
    std::unique_lock<std::mutex> lock(w->m);
    for (;;) {
        w->cv.wait(lock, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lock.unlock();
        g_err.clear();
        const int rc = job();
        lock.lock();
        w->rc = rc;
        w->err = g_err; // (the message of a failed call is the worker thread's own: carried back to the caller's)
        w->done = true;
        w->cv.notify_all();
    }
}
// fn(i) on scene i's thread, all at once; the first failure's code and message
static int multi_run(MultiScene &m, const std::function<int(int)> &fn) {
    const int n = (int)m.scenes.size();
    for (int i = 0; i < n; i++) {
        MultiWorker &w = *m.workers[i];
        std::lock_guard<std::mutex> lock(w.m);
        w.job = [i, &fn] { return fn(i); };
        w.has_job = true; w.done = false;
        w.cv.notify_all();
    }
    int rc = RT355_OK;
    std::string why;
    for (int i = 0; i < n; i++) {
        MultiWorker &w = *m.workers[i];
        std::unique_lock<std::mutex> lock(w.m);
        w.cv.wait(lock, [&] { return w.done; });
        if (w.rc != RT355_OK && rc == RT355_OK) { rc = w.rc; why = "device " + std::to_string(m.devices[i]) + ": " + w.err; }
    }
    if (rc != RT355_OK) g_err = why;
    return rc;
}
static void multi_destroy(rt355_scene *sc) {
    MultiScene *m = sc->multi;
    for (auto &w : m->workers) {
        { std::lock_guard<std::mutex> lock(w->m); w->quit = true; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
    }
    for (rt355_scene *r : m->scenes) rt355_scene_destroy(r);
    delete m;
    delete sc;
}
static void multi_sum(const std::vector<rt355_stats> &st, rt355_stats *out) {
    if (!out) return;
    memset(out, 0, sizeof *out);
    for (const rt355_stats &s : st) {
        out->primary_rays += s.primary_rays; out->shadow_rays += s.shadow_rays; out->pair_tests += s.pair_tests;
        out->exact_tests += s.exact_tests; out->overflow_rays += s.overflow_rays; out->packet_survivors += s.packet_survivors;
        out->primary_hits += s.primary_hits; out->secondary_rays += s.secondary_rays; out->gather_rays += s.gather_rays;
        out->ms_device = std::max(out->ms_device, s.ms_device); // the GPUs work side by side: the longest
    }
}

extern "C" int rt355_scene_create_multi(const rt355_scene_desc *d, const int32_t *devices, int32_t n_devices, rt355_scene **out) {
    if (!d || !out || !devices || n_devices < 1 || n_devices > MULTI_MAX_DEVICES)
        return fail(RT355_ERR_ARG, "rt355_scene_create_multi: null argument, or not 1.." + std::to_string(MULTI_MAX_DEVICES) + " devices");
    int have = 0, before = 0;
    HIP_TRY(hipGetDeviceCount(&have));
    HIP_TRY(hipGetDevice(&before));
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= have)
            return fail(RT355_ERR_ARG, "rt355_scene_create_multi: device " + std::to_string(devices[i]) + " of " + std::to_string(have));
    rt355_scene *sc = new rt355_scene();
    sc->multi = new MultiScene();
    sc->device = devices[0];
    int rc = RT355_OK;
    { // the scene replicated: flattened and uploaded once per device, all devices at once (a thread each: the scene
      // table of a million triangles takes a second to build, and a thread's current device is its own)
        std::vector<rt355_scene *> made((size_t)n_devices, nullptr);
        std::vector<int> rcs((size_t)n_devices, RT355_OK);
        std::vector<std::string> errs((size_t)n_devices);
        std::vector<std::thread> threads;
        for (int i = 0; i < n_devices; i++)
            threads.emplace_back([&, i] {
                if (hipSetDevice(devices[i]) != hipSuccess) { (void)hipGetLastError(); rcs[i] = RT355_ERR_DEVICE; errs[i] = "hipSetDevice failed"; return; }
                rcs[i] = rt355_scene_create(d, &made[i]);
                if (rcs[i] != RT355_OK) errs[i] = g_err;
            });
        for (auto &t : threads) t.join();
        for (int i = 0; i < n_devices; i++) {
            if (made[i]) { sc->multi->scenes.push_back(made[i]); sc->multi->devices.push_back(devices[i]); }
            if (rcs[i] != RT355_OK && rc == RT355_OK) rc = fail(rcs[i], "device " + std::to_string(devices[i]) + ": " + errs[i]);
        }
    }
    (void)hipSetDevice(before);
    if (rc != RT355_OK) {
        const std::string why = g_err;
        multi_destroy(sc);
        g_err = why;
        return rc;
    }
    for (int i = 0; i < n_devices; i++) {
        sc->multi->workers.emplace_back(new MultiWorker());
        MultiWorker *w = sc->multi->workers.back().get();
        w->th = std::thread(multi_worker_main, w, devices[i]);
    }
    *out = sc;
    return RT355_OK;
}

static int ensure_fb(rt355_scene *r, size_t bytes) {
    if (bytes <= r->fb_bytes) return RT355_OK;
    if (r->fb) { (void)hipFree(r->fb); r->fb = nullptr; r->fb_bytes = 0; }
    HIP_TRY(hipMalloc((void **)&r->fb, bytes));
    r->fb_bytes = bytes;
    return RT355_OK;
}

static int multi_render(rt355_scene *sc, const rt355_camera *cam, const rt355_shader *shader, const rt355_opts *opts,
                        unsigned char *pixmap, rt355_stats *stats) {
    MultiScene &m = *sc->multi;
    const int n = (int)m.scenes.size();
    if (!cam || !shader || !opts || !pixmap) return fail(RT355_ERR_ARG, "rt355_render: null argument");
    if (opts->width <= 0 || opts->height <= 0) return fail(RT355_ERR_ARG, "width and height must be positive");
    if (opts->tile_count > 1)
        return fail(RT355_ERR_UNSUPPORTED, "rt355_render on a multi-device scene deals the row tiles itself: leave tile_count at 0");
    Rows rows;
    { int rc0 = rows_of(opts, &rows); if (rc0) return rc0; }
    std::vector<rt355_stats> st((size_t)n);
    for (auto &s : st) memset(&s, 0, sizeof s);
    // (slabs need a row each; a frame of fewer rows than devices goes in tiles, of which some devices get none)
    const bool in_slabs = m.balance != 0 && n > 1 && rows.y1 - rows.y0 >= n;
    if (in_slabs && !m.slabs.fits(rows.y0, rows.y1, n)) m.slabs.start(rows.y0, rows.y1, n);
    std::vector<double> share_ms((size_t)n, 0.0);
    auto share_of = [&](int i) {
        rt355_opts o = *opts;
        if (in_slabs) { o.row_begin = m.slabs.cuts[i]; o.row_end = m.slabs.cuts[i + 1]; }
        else { o.tile_rows = MULTI_TILE_ROWS; o.tile_count = n; o.tile_phase = i; }
        return o;
    };
    int rc;
    if (m.gather == 0 || n == 1) {
        rc = multi_run(m, [&](int i) {
            const rt355_opts o = share_of(i);
            const auto t0 = std::chrono::steady_clock::now();
            const int rc1 = rt355_render(m.scenes[i], cam, shader, n == 1 ? opts : &o, pixmap, &st[i]);
            share_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return rc1;
        });
    } else {
        const size_t row_bytes = (size_t)opts->width * 3, bytes = row_bytes * (size_t)opts->height;
        rc = multi_run(m, [&](int i) { // every share into its own GPU's framebuffer
            rt355_scene *r = m.scenes[i];
            HIP_TRY(hipSetDevice(r->device));
            int rc1 = ensure_fb(r, bytes);
            if (rc1) return rc1;
            const rt355_opts o = share_of(i);
            const auto t0 = std::chrono::steady_clock::now();
            if ((rc1 = render_enqueue(r, cam, shader, &o, r->fb, nullptr, &st[i], false))) return rc1;
            rc1 = rt355_scene_sync(r);
            share_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            return rc1;
        });
        if (rc == RT355_OK)
            rc = multi_run(m, [&](int i) { // ... its rows to the first GPU over the link between the two
                if (i == 0) return (int)RT355_OK;
                rt355_scene *r = m.scenes[i], *r0 = m.scenes[0];
                HIP_TRY(hipSetDevice(r->device));
                if (r->device != r0->device) { // (direct access where the devices allow it; the copy works without)
                    const hipError_t e = hipDeviceEnablePeerAccess(r0->device, 0);
                    if (e != hipSuccess) (void)hipGetLastError();
                }
                if (in_slabs) {
                    const int ta = m.slabs.cuts[i], tb = m.slabs.cuts[i + 1];
                    HIP_TRY(hipMemcpyPeerAsync(r0->fb + ta * row_bytes, r0->device, r->fb + ta * row_bytes, r->device,
                                               (size_t)(tb - ta) * row_bytes, nullptr));
                }
                for (int k = i; !in_slabs && k * MULTI_TILE_ROWS < rows.y1 - rows.y0; k += n) {
                    const int ta = rows.y0 + k * MULTI_TILE_ROWS, tb = std::min(rows.y1, ta + MULTI_TILE_ROWS);
                    HIP_TRY(hipMemcpyPeerAsync(r0->fb + ta * row_bytes, r0->device, r->fb + ta * row_bytes, r->device,
                                               (size_t)(tb - ta) * row_bytes, nullptr));
                }
                HIP_TRY(hipStreamSynchronize(nullptr));
                return (int)RT355_OK;
            });
        if (rc == RT355_OK)
            rc = multi_run(m, [&](int i) { // ... and the frame's rows from there to the caller, one transfer
                if (i != 0) return (int)RT355_OK;
                rt355_scene *r0 = m.scenes[0];
                HIP_TRY(hipSetDevice(r0->device));
                HIP_TRY(hipMemcpy(pixmap + rows.y0 * row_bytes, r0->fb + rows.y0 * row_bytes, (size_t)(rows.y1 - rows.y0) * row_bytes, hipMemcpyDeviceToHost));
                return (int)RT355_OK;
            });
    }
    if (rc == RT355_OK) multi_sum(st, stats);
    if (rc == RT355_OK && in_slabs) m.slabs.update(share_ms); // the next call's cuts
    return rc;
}

static int multi_sweep(rt355_scene *sc, const rt355_camera *cams, int32_t n_frames, const rt355_shader *shader,
                       const rt355_opts *opts, unsigned char *const *pixmaps, rt355_stats *stats) {
    MultiScene &m = *sc->multi;
    const int n = (int)m.scenes.size();
    if (!opts || n_frames < 0 || (n_frames && (!cams || !pixmaps))) return fail(RT355_ERR_ARG, "rt355_render_sweep: bad argument");
    // frame f on scene f mod N: each scene sweeps its frames through its own framebuffers and copy stream
    std::vector<std::vector<rt355_camera>> my_cams((size_t)n);
    std::vector<std::vector<unsigned char *>> my_pix((size_t)n);
    std::vector<std::vector<rt355_stats>> my_st((size_t)n);
    for (int f = 0; f < n_frames; f++) {
        my_cams[f % n].push_back(cams[f]);
        my_pix[f % n].push_back(pixmaps[f]);
    }
    for (int i = 0; i < n; i++) my_st[i].resize(my_cams[i].size());
    const int rc = multi_run(m, [&](int i) {
        if (my_cams[i].empty()) return (int)RT355_OK;
        return rt355_render_sweep(m.scenes[i], my_cams[i].data(), (int32_t)my_cams[i].size(), shader, opts, my_pix[i].data(), my_st[i].data());
    });
    for (int f = 0; f < n_frames && stats && rc == RT355_OK; f++) stats[f] = my_st[f % n][f / n];
    return rc;
}

extern "C" int rt355_scene_set_option(rt355_scene *sc, const char *name, int64_t value) {
    if (!sc || !name) return fail(RT355_ERR_ARG, "rt355_scene_set_option: null argument");
    if (sc->multi) {
        if (!strcmp(name, "gather")) {
            if (value != 0 && value != 1) return fail(RT355_ERR_ARG, "option gather: 0 (rows into the host pixmap) or 1 (rows to the first GPU)");
            sc->multi->gather = (int)value;
            return RT355_OK;
        }
        if (!strcmp(name, "balance")) {
            if (value != 0 && value != 1) return fail(RT355_ERR_ARG, "option balance: 0 (interleaved row tiles) or 1 (contiguous slabs cut by the shares' times)");
            sc->multi->balance = (int)value;
            sc->multi->slabs = SlabBalance();
            return RT355_OK;
        }
        for (rt355_scene *r : sc->multi->scenes) { const int rc = rt355_scene_set_option(r, name, value); if (rc) return rc; }
        return RT355_OK;
    }
    if (any_pending(sc)) { int rc = rt355_scene_sync(sc); if (rc) return rc; }
    const int rc = set_knob(sc->knobs, name, value);
    if (rc == RT355_OK && !strcmp(name, "readback")) sc->readback_key = 0; // (what was timed so far decides nothing any more)
    if (rc == RT355_OK && !strcmp(name, "debug_readback") && value) sc->debug_series.reserve(4096);
    return rc;
}

extern "C" int rt355_scene_set_directions(rt355_scene *sc, const double *xyz, int32_t n) {
    if (!sc || n < 0 || (n && !xyz)) return fail(RT355_ERR_ARG, "rt355_scene_set_directions: bad argument");
    if (sc->multi) {
        for (rt355_scene *r : sc->multi->scenes) { const int rc = rt355_scene_set_directions(r, xyz, n); if (rc) return rc; }
        return RT355_OK;
    }
    HIP_TRY(hipSetDevice(sc->device));
    if (any_pending(sc)) { int rc = rt355_scene_sync(sc); if (rc) return rc; }
    if (sc->d_dirs) { (void)hipFree(sc->d_dirs); sc->d_dirs = nullptr; }
    sc->n_dirs = 0;
    if (n) {
        HIP_TRY(hipMalloc((void **)&sc->d_dirs, (size_t)n * 3 * sizeof(double)));
        HIP_TRY(hipMemcpy(sc->d_dirs, xyz, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice));
        sc->n_dirs = n;
    }
    return RT355_OK;
}

// Debug probe: the device's sincos_rounded for n host arguments (0 <= x < 8) -> n x 2 doubles (sin, cos)
__global__ void k_probe_sincos(const double *__restrict__ x, int n, double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double sv, cv;
    sincos_rounded(x[i], &sv, &cv);
    out[2 * i] = sv; out[2 * i + 1] = cv;
}
extern "C" int rt355_probe_sincos(const double *x, int n, double *out) {
    if (!x || !out || n <= 0) return fail(RT355_ERR_ARG, "rt355_probe_sincos: bad argument");
    for (int i = 0; i < n; i++)
        if (!(x[i] >= 0 && x[i] < 8)) return fail(RT355_ERR_ARG, "rt355_probe_sincos: 0 <= x < 8");
    double *dx = nullptr, *dout = nullptr;
    auto run = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&dx, (size_t)n * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&dout, (size_t)n * 2 * sizeof(double)));
        HIP_TRY(hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe_sincos, dim3((n + 127) / 128), dim3(128), 0, 0, dx, n, dout);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out, dout, (size_t)n * 2 * sizeof(double), hipMemcpyDeviceToHost));
        return RT355_OK;
    };
    const int rc = run();
    if (dx) (void)hipFree(dx);
    if (dout) (void)hipFree(dout);
    return rc;
}

// Debug probe: objs[obj]->CalcIntersect for n host rays (n x 6 doubles) -> n x 9 doubles
// (t, ph, nh, uv).  Lets the fp64 device formulas be checked in isolation.
extern "C" int rt355_probe_intersect(rt355_scene *sc, int obj, const double *rays, int n, double *out) {
    if (sc && sc->multi) return rt355_probe_intersect(sc->multi->scenes[0], obj, rays, n, out);
    if (!sc || !rays || !out || n <= 0 || obj < 0 || obj >= sc->view.n_objects) return fail(RT355_ERR_ARG, "rt355_probe_intersect: bad argument");
    HIP_TRY(hipSetDevice(sc->device));
    double *dr = nullptr, *dout = nullptr;
    auto run = [&]() -> int {
        HIP_TRY(hipMalloc((void **)&dr, (size_t)n * 6 * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&dout, (size_t)n * 9 * sizeof(double)));
        HIP_TRY(hipMemcpy(dr, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_probe_intersect, dim3((n + 127) / 128), dim3(128), 0, 0, sc->view, obj, dr, n, dout);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(out, dout, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost));
        return RT355_OK;
    };
    const int rc = run();
    if (dr) (void)hipFree(dr);
    if (dout) (void)hipFree(dout);
    return rc;
}
