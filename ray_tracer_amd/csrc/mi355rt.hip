// mi355rt.hip -- HIP kernels (gfx950) and the C ABI of libmi355rt.so.
//
// Kernels
//   closest_hit_kernel  batched world.hit (camera.rs:258 / hittable.rs:49) for any scene: persistent waves take
//                       ranges of the batch (WaveWork), one ray per lane, LDS stack, full hit record.
//   bvh4_closest_hit_kernel (bvh_kernel.h)  the same query when the world is one BVH over primitives; followed by
//                       finalize_hits_kernel, which completes the records in one coalesced pass.
//   raygen_kernel       Camera::get_ray (camera.rs:207-240) for every (pixel, stratum) of a batch.
//   extend_kernel       closest hit for the live paths of the current bounce (queue of path ids).
//   shade_kernel        one bounce of Camera::ray_color (camera.rs:243-339) + wave-level
//                       ballot compaction of the surviving paths into the next queue.
//   tail_kernel         the last bounces of a batch in one launch (each lane follows its path to the end).
//   accumulate_kernel   pixel_color += ray_color(...) in stratum order (camera.rs:110-131).
// The product path has no CPU fallback: every entry point that computes needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "flatten.h"
#include "shade.h"
#include "bvh_kernel.h"

using namespace rt;

// Scratch of rt_closest_hit_device_noalloc: the transform context of every hit of a batch, kept between the
// traversal kernel and the record-completion kernel.  It lives in the code object, so every GPU the library
// runs on holds its own 64 MiB copy until the process ends.  In host code the name is only a handle for
// hipGetSymbolAddress: kernels and memsets take the per-device address kept in g_scratch_ptr.
constexpr uint64_t NOALLOC_MAX_RAYS = 1ull << 24;
static __device__ __align__(256) int32_t scratch[NOALLOC_MAX_RAYS];

// ============================================================================ device code
namespace {

constexpr int TAIL_PATHS = 1 << 16; // a batch with this many live paths or fewer finishes in one launch (tail_kernel)
constexpr unsigned long long RANGES_PER_WAVE = 8ull; // work ranges a wave of a persistent kernel gets on average
enum { C_WORK = 0, C_LIVE_IN = 1, C_LIVE_OUT = 2, C_NODES = 4, C_PRIMS = 5, C_RAYS = 6, C_WORDS = 32 };
// Work counters of the general persistent kernels: one per XCD, each on a 128-byte line of its own, behind the
// C_WORDS statistics words.
// (The primitive-BVH kernel is handed the first slot and addresses them as work_counter + (workgroup % 8) * 16.)
constexpr int WORK_SLOTS = 8, WORK_STRIDE = 16, C_WORK_SLOTS = 32, C_ALLOC = C_WORK_SLOTS + WORK_SLOTS * WORK_STRIDE;

struct CamD { // Camera after initialize() (camera.rs:30-42)
    rt_camera_derived d;
    double defocus_angle;
    double background[3];
    int32_t image_width;
    int32_t russian_roulette;
};

// One path = one 128-byte line in HBM.  The queues permute the paths after the first bounce, so every
// kernel gathers its paths by index: with one array per field that was seventeen 8-byte gathers (seventeen
// lines) per path and bounce, with one record it is a single line read and written once per kernel.
struct alignas(128) PathRec {
    double o[3], d[3];  // current ray (ray.rs:5-9)
    double time;
    double beta[3];     // throughput
    double L[3];        // accumulated radiance
    uint32_t rng_index; // draws taken so far from the path's stream
    uint32_t _pad;
    double hit_t;       // result of the last extend: the last 16 bytes of the line, written with one store
    uint32_t hit_ref;
    int32_t hit_ctx;
};
static_assert(sizeof(PathRec) == 128, "a path is one cache line");
static_assert(offsetof(PathRec, hit_t) == 112, "the hit block is one aligned 16-byte store");
constexpr uint64_t PATH_DOUBLES = sizeof(PathRec) / sizeof(double);

struct PathState {
    PathRec* rec;
    uint32_t* queue[2];
    unsigned long long* counters;
    double2* uv;   // per path: the u,v a constant-medium hit reports (null unless some phase function reads them: traverse.h UvTrack)
};

// The world ray of path p as the traversal and the record completion read it (strided view of the records).
__device__ __forceinline__ WorldRay path_ray(const PathState& ps, uint32_t p) {
    const double* b = reinterpret_cast<const double*>(ps.rec);
    return WorldRay{b + 0, b + 1, b + 2, b + 3, b + 4, b + 5, (uint64_t)p * PATH_DOUBLES};
}

struct BatchGeom { // which (pixel, stratum) a path slot stands for
    int32_t x0, y0, tw, th;        // tile origin and size
    int32_t sample0, nsamples;     // strata [sample0, sample0 + nsamples)
    uint32_t npix;                 // tw * th
    uint32_t npaths;               // npix * nsamples
};

__device__ __forceinline__ unsigned long long wave_fetch(unsigned long long* counter, unsigned long long amount) {
    unsigned long long base = 0;
    if ((threadIdx.x & 63u) == 0u) base = atomicAdd(counter, amount);
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Work distribution of the persistent kernels.  The chip serves same-address atomics at roughly 10 ns each:
// with one counter for all waves, one atomic per 64 rays caps a kernel near 6 G rays/s, and at launch, when
// every wave asks at the same moment, the last in line waits tens of microseconds.  So a range is up to 16
// wave-loads (fewer when the batch is small, so that every wave still gets several ranges); wave w starts on
// range w without asking anybody; after that the waves of XCD x (workgroups are dealt round-robin over the 8
// XCDs) take the ranges waves + 8k + x, k from a counter of their own.  Ranges are still handed out in queue
// order, so at any moment the whole chip works on one window of the queue, which keeps the path records it
// gathers close together in HBM (an even static split, every wave streaming through a region of its own,
// was 8 % slower; so was looking at the counter with a plain load before paying for the last atomic).
// The last quarter of the batch is handed out in ranges a quarter as long: a launch ends when its last range
// does, and with ~8 ranges per wave a full-length last range is an eighth of the launch.
struct WaveWork {
    unsigned long long n, len, short_len, n_long, waves, wave_id;
    uint32_t slot, slots;
    bool started;
    __device__ __forceinline__ WaveWork(unsigned long long n_) : n(n_), started(false) {
        waves = (unsigned long long)gridDim.x * (blockDim.x / 64u);
        wave_id = (unsigned long long)blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
        const unsigned long long k = n / (waves * RANGES_PER_WAVE * 64ull);
        len = 64ull * (k < 1ull ? 1ull : (k > 16ull ? 16ull : k));
        short_len = 64ull * (len / 256ull < 1ull ? 1ull : len / 256ull);
        n_long = (n - n / 4ull) / len; // ranges of full length; range g >= n_long starts at n_long*len + (g - n_long)*short_len
        slots = gridDim.x < (uint32_t)WORK_SLOTS ? gridDim.x : (uint32_t)WORK_SLOTS; // every residue class needs a taker
        slot = blockIdx.x % slots;
    }
    // Next range [base, base + length) of this wave; base >= n when the batch is finished.  `counters` is the
    // scene's counter block (statistics words, then the work slots).
    __device__ __forceinline__ unsigned long long next(unsigned long long* counters, unsigned long long& length) {
        unsigned long long g; // index of the range in queue order
        if (!started) { started = true; g = wave_id; }
        else {
            // cheap exit without an atomic when even this slot's first dynamic range lies beyond the batch
            if (n_long * len + (waves + slot >= n_long ? (waves + slot - n_long) * short_len : 0ull) >= n && waves + slot >= n_long) { length = 0; return n; }
            g = waves + slot + wave_fetch(counters + C_WORK_SLOTS + slot * WORK_STRIDE, 1ull) * (unsigned long long)slots;
        }
        if (g < n_long) { length = len; return g * len; }
        length = short_len;
        return n_long * len + (g - n_long) * short_len;
    }
};

// u,v of the record a medium hit was written into: those of the primitive hit that last wrote it, or the default 0, 0.
__device__ __forceinline__ void stale_uv(const DScene& S, const WorldRay& w, double time, const UvTrack& uvt, double& u, double& v) {
    u = 0.0; v = 0.0;
    if (uvt.ref == RT_REF_NONE) return;
    Best b;
    b.t = uvt.t; b.ref = uvt.ref; b.ctx = uvt.ctx;
    HitRec h;
    finalize(S, w, time, b, h, true);
    u = h.u; v = h.v;
}

// MEDIA = 0: the scene holds no constant medium; the nested boundary queries and the random draw are compiled out.
// UV: some phase function reads u,v, so the walk keeps track of whose u,v a medium hit inherits (traverse.h UvTrack).
template <bool COUNT, bool ORDERED, int MEDIA, bool UV = false>
__global__ __launch_bounds__(BLOCK) void closest_hit_kernel(DScene S, rt_ray_batch rays, unsigned long long n, rt_hit_batch hits,
                                                            int32_t* hit_ctx, unsigned long long seed, unsigned long long* counters, int rest_only,
                                                            int axis_order) {
    __shared__ uint32_t s_stack[STACK_MAX * BLOCK];
    uint32_t* stk = s_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t nn = 0, np = 0;
    WaveWork work(n);
    for (;;) {
        unsigned long long chunk;
        const unsigned long long base = work.next(counters, chunk);
        if (base >= n) break;
        for (unsigned long long off = 0; off < chunk; off += 64ull) {
        unsigned long long i = base + off + lane;
        if (i < n) {
            WorldRay w{rays.ox, rays.oy, rays.oz, rays.dx, rays.dy, rays.dz, i};
            RayD r;
            ray_in_context(S, w, 0, r);
            double time = rays.time ? rays.time[i] : 0.0;
            double tmin = rays.tmin ? rays.tmin[i] : 0.001;
            Best b;
            b.t = rays.tmax ? rays.tmax[i] : D_INF;
            b.ref = RT_REF_NONE; b.ctx = 0;
            Rng g{seed, 2u, (uint32_t)i, (uint32_t)(i >> 32), 0u};
            uint32_t root = S.world;
            if (rest_only) { // the primitive-BVH kernel has traced the world's first member into hits.t / hits.prim
                const uint32_t first_hit = hits.prim[i];
                if (first_hit != RT_REF_NONE) { b.t = hits.t[i]; b.ref = first_hit; }
                root = S.world_rest;
            }
            UvTrack uvt;
            uvt.ref = RT_REF_NONE; uvt.ctx = 0; uvt.t = 0.0; uvt.any = false;
            traverse<MEDIA, COUNT, ORDERED, true, true, UV>(S, root, r, 0, w, time, tmin, b, stk, 0, &g, nn, np, axis_order != 0, &uvt);
            if (UV && b.ref != RT_REF_NONE && (b.ref >> 28) == RT_REF_MEDIUM) { // finalize_hits_kernel leaves these two alone
                double u, v;
                stale_uv(S, w, time, uvt, u, v);
                if (hits.u) hits.u[i] = u;
                if (hits.v) hits.v[i] = v;
            }
            // (t, what was hit, in which transform context): finalize_hits_kernel completes the record in a pass of
            // its own, which keeps the record arithmetic (acos, atan2, the transform unwinding) out of this
            // kernel's register budget
            hits.t[i] = b.ref != RT_REF_NONE ? b.t : D_INF;
            hits.prim[i] = b.ref;
            hit_ctx[i] = b.ctx;
        }
        }
    }
    if (COUNT) {
        atomicAdd(&counters[C_NODES], (unsigned long long)nn);
        atomicAdd(&counters[C_PRIMS], (unsigned long long)np);
    }
}

__global__ void init_batch_kernel(unsigned long long* counters, unsigned long long npaths) {
    counters[C_WORK] = 0ull;
    for (int k = 0; k < WORK_SLOTS; ++k) counters[C_WORK_SLOTS + k * WORK_STRIDE] = 0ull;
    counters[C_LIVE_IN] = npaths;
    counters[C_LIVE_OUT] = 0ull;
}

// After a bounce: the compacted queue becomes the input of the next one.
// The counters are read and re-written in one kernel.  A uniform-address load goes through the
// scalar cache while the store goes through the vector path, and nothing orders the two: the
// plain form `in = c[OUT]; c[OUT] = 0` was observed to read the zero it stores.  Exchanges make
// every read-modify-write a single memory operation.
__global__ void advance_kernel(unsigned long long* counters) {
    unsigned long long out = atomicExch(&counters[C_LIVE_OUT], 0ull);
    unsigned long long in = atomicExch(&counters[C_LIVE_IN], out);
    atomicAdd(&counters[C_RAYS], in);
    atomicExch(&counters[C_WORK], 0ull);
    for (int k = 0; k < WORK_SLOTS; ++k) atomicExch(&counters[C_WORK_SLOTS + k * WORK_STRIDE], 0ull);
}

__global__ void reset_work_kernel(unsigned long long* counters) {
    counters[C_WORK] = 0ull;
    for (int k = 0; k < WORK_SLOTS; ++k) counters[C_WORK_SLOTS + k * WORK_STRIDE] = 0ull;
}

__global__ __launch_bounds__(BLOCK) void raygen_kernel(CamD cam, BatchGeom bg, PathState ps, unsigned long long seed) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= bg.npaths) return;
    uint32_t s_local = p / bg.npix, pix = p - s_local * bg.npix;
    int32_t i = bg.x0 + (int32_t)(pix % (uint32_t)bg.tw);
    int32_t j = bg.y0 + (int32_t)(pix / (uint32_t)bg.tw);
    int32_t sidx = bg.sample0 + (int32_t)s_local;
    int32_t s_j = sidx / cam.d.sqrt_spp, s_i = sidx % cam.d.sqrt_spp;
    Rng g{seed, 0u, (uint32_t)(j * cam.image_width + i), (uint32_t)sidx, 0u};
    // sample_square_stratified (camera.rs:227-231)
    double px = (((double)s_i + g.next()) * cam.d.recip_sqrt_spp) - 0.5;
    double py = (((double)s_j + g.next()) * cam.d.recip_sqrt_spp) - 0.5;
    D3 pixel_sample = d3p(cam.d.pixel00_loc) + (((double)i + px) * d3p(cam.d.pixel_delta_u)) + (((double)j + py) * d3p(cam.d.pixel_delta_v));
    D3 origin;
    if (cam.defocus_angle <= 0.0) origin = d3p(cam.d.center);
    else { // defocus_disk_sample (camera.rs:237-240)
        D3 dsk = random_in_unit_disk(g);
        origin = d3p(cam.d.center) + (dsk.x * d3p(cam.d.defocus_disk_u)) + (dsk.y * d3p(cam.d.defocus_disk_v));
    }
    D3 dir = pixel_sample - origin;
    double time = g.next();
    PathRec rec;
    rec.o[0] = origin.x; rec.o[1] = origin.y; rec.o[2] = origin.z;
    rec.d[0] = dir.x; rec.d[1] = dir.y; rec.d[2] = dir.z;
    rec.time = time;
    rec.beta[0] = 1.0; rec.beta[1] = 1.0; rec.beta[2] = 1.0;
    rec.L[0] = 0.0; rec.L[1] = 0.0; rec.L[2] = 0.0;
    rec.hit_t = D_INF; rec.hit_ref = RT_REF_NONE; rec.hit_ctx = 0;
    rec.rng_index = g.index; rec._pad = 0;
    ps.rec[p] = rec;
    ps.queue[0][p] = p;
}

// MEDIA = false: the scene holds no constant medium, so the nested boundary queries and the random draw inside
// traversal are compiled out (fewer registers, one more wave per SIMD).
// At least 3 waves per SIMD (168 registers): the variant with media sits within a few registers of that limit,
// and one wave less costs it more than 10 %.
template <int MEDIA, bool UV = false>
__global__ __launch_bounds__(BLOCK, 3) void extend_kernel(DScene S, PathState ps, int qsel, unsigned long long seed, BatchGeom bg, int32_t image_width,
                                                          int rest_only) {
    __shared__ uint32_t s_stack[STACK_MAX * BLOCK];
    uint32_t* stk = s_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long n = ps.counters[C_LIVE_IN];
    const uint32_t* queue = ps.queue[qsel];
    uint32_t nn = 0, np = 0;
#ifdef RT_PHASE_STATS
    if (threadIdx.x < 8) s_step_cycles[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long ph[4] = {0ull, 0ull, 0ull, 0ull}, pieces = 0ull; // wave cycles: work fetch, record load, traversal, write-back
#define RT_PHASE(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[k] += now_ - t_ph; t_ph = now_; } while (0)
    unsigned long long t_ph = __builtin_amdgcn_s_memtime();
#else
#define RT_PHASE(k) do { } while (0)
#endif
    WaveWork work(n);
    for (;;) {
        unsigned long long chunk;
        const unsigned long long base = work.next(ps.counters, chunk);
        RT_PHASE(0);
        if (base >= n) break;
        uint32_t p_next = base + lane < n ? queue[base + lane] : 0u; // the queue entry is read one wave-load ahead of its record
        for (unsigned long long off = 0; off < chunk; off += 64ull) {
        unsigned long long q = base + off + lane;
#ifdef RT_PHASE_STATS
        if (base + off < n) ++pieces;
#endif
        const uint32_t p = p_next;
        if (off + 64ull < chunk && q + 64ull < n) p_next = queue[q + 64ull];
        if (q < n) {
            PathRec* rec = ps.rec + p;
            const WorldRay w = path_ray(ps, p);
            RayD r;
            ray_in_context(S, w, 0, r);
            double time = rec->time;
            RT_PHASE(1);
            Best b;
            b.t = D_INF; b.ref = RT_REF_NONE; b.ctx = 0;
            uint32_t root = S.world;
            if (rest_only) { // the primitive-BVH kernel has traced the world's first member: go on from its closest hit
                b.t = rec->hit_t; b.ref = rec->hit_ref;
                root = S.world_rest;
            }
            uint32_t s_local = p / bg.npix, pix = p - s_local * bg.npix;
            int32_t i = bg.x0 + (int32_t)(pix % (uint32_t)bg.tw);
            int32_t j = bg.y0 + (int32_t)(pix / (uint32_t)bg.tw);
            Rng g{seed, 0u, (uint32_t)(j * image_width + i), (uint32_t)(bg.sample0 + (int32_t)s_local), rec->rng_index};
#ifdef RT_PHASE_STATS
            s_lane_iters[threadIdx.x] = 0u;
#endif
            UvTrack uvt;
            uvt.ref = RT_REF_NONE; uvt.ctx = 0; uvt.t = 0.0; uvt.any = false;
            traverse<MEDIA, false, true, true, MEDIA == 0, UV>(S, root, r, 0, w, time, 0.001, b, stk, 0, &g, nn, np, false, &uvt); // Interval::new(0.001, INFINITY) (camera.rs:258)
            if (UV && b.ref != RT_REF_NONE && (b.ref >> 28) == RT_REF_MEDIUM) {
                double2 uv;
                stale_uv(S, w, time, uvt, uv.x, uv.y);
                ps.uv[p] = uv;
            }
            RT_PHASE(2);
#ifdef RT_PHASE_STATS
            {
                uint32_t it = s_lane_iters[threadIdx.x], mx = it, sm = it;
                for (int o = 32; o > 0; o >>= 1) { uint32_t a = __shfl_xor(mx, o), c = __shfl_xor(sm, o); mx = a > mx ? a : mx; sm += c; }
                const unsigned long long act = __popcll(__ballot(true));
                if (lane == (uint32_t)__builtin_ctzll(__ballot(true))) { atomicAdd(&ps.counters[29], (unsigned long long)sm); atomicAdd(&ps.counters[30], (unsigned long long)mx * act); }
            }
#endif
            uint4 hit_block; // hit_t, hit_ref, hit_ctx
            hit_block.x = (uint32_t)__double2loint(b.t); hit_block.y = (uint32_t)__double2hiint(b.t);
            hit_block.z = b.ref; hit_block.w = (uint32_t)b.ctx;
            *reinterpret_cast<uint4*>(&rec->hit_t) = hit_block;
            if (MEDIA) rec->rng_index = g.index; // only a constant medium draws during traversal (constant_medium.rs:90)
            RT_PHASE(3);
        }
        }
    }
#ifdef RT_PHASE_STATS
    if (lane == 0u) {
        for (int k = 0; k < 4; ++k) atomicAdd(&ps.counters[16 + k], ph[k]);
        atomicAdd(&ps.counters[20], pieces);
    }
    __syncthreads();
    if (threadIdx.x < 8) atomicAdd(&ps.counters[8 + threadIdx.x], s_step_cycles[threadIdx.x]);
#endif
#undef RT_PHASE
}

constexpr int SHADE_ROUNDS = 16; // wave-loads of paths whose survivors share one queue reservation
constexpr unsigned long long SHADE_GROUP = 4ull; // consecutive wave-loads a wave shades before it moves on by a stride of all waves
// (two workgroups per CU: without the bound the textured variant needs 263 registers, seven over the limit for a
// second wave per SIMD, which it is worth a few spilled values to keep)
template <bool TEXTURES, bool RR>
__global__ __launch_bounds__(BLOCK, 2) void shade_kernel(DScene S, CamD cam, PathState ps, int qsel, unsigned long long seed, BatchGeom bg, int last_bounce) {
    const unsigned long long n = ps.counters[C_LIVE_IN];
    const uint32_t* qin = ps.queue[qsel];
    uint32_t* qout = ps.queue[qsel ^ 1];
    // Survivors of up to SHADE_ROUNDS wave-loads are collected in LDS and get their place in the next queue
    // with ONE atomic (all waves bump the same counter; one atomic per 64 paths capped this kernel).
    __shared__ uint32_t s_keep[(BLOCK / 64) * SHADE_ROUNDS * 64];
    uint32_t* keep = s_keep + (threadIdx.x >> 6) * (SHADE_ROUNDS * 64);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long waves = (unsigned long long)gridDim.x * (blockDim.x / 64u);
    const unsigned long long wave_id = (unsigned long long)blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
    // Wave w shades groups of SHADE_GROUP consecutive wave-loads, w, w + waves, w + 2*waves, ... in units of
    // groups: no wave gets more than one group above its share (handing out whole chunks of 16 wave-loads left a
    // quarter of the waves with 4 chunks and the rest with 3 at 12 M paths), at any moment the chip works on one
    // window of the queue, and neighbours in the queue stay neighbours in the next one (single wave-loads
    // interleaved across waves balance better still but cost the BVH scenes 3 %: the next bounce's waves
    // then mix paths from distant parts of the frame).
    const unsigned long long loads = (n + 63ull) / 64ull;
    const unsigned long long groups = (loads + SHADE_GROUP - 1ull) / SHADE_GROUP;
#ifdef RT_PHASE_STATS
    unsigned long long ph[5] = {0ull, 0ull, 0ull, 0ull, 0ull}; // wave cycles: record load, hit record, material, write-back, queue
#define RT_PHASE(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[k] += now_ - t_ph; t_ph = now_; } while (0)
    unsigned long long t_ph = __builtin_amdgcn_s_memtime();
#else
#define RT_PHASE(k) do { } while (0)
#endif
    {
        uint32_t kept = 0, rounds = 0; // wave-uniform
        uint32_t p_next = wave_id * SHADE_GROUP * 64ull + lane < n ? qin[wave_id * SHADE_GROUP * 64ull + lane] : 0u; // the queue entry is read one wave-load ahead of its record
        for (unsigned long long grp = wave_id; grp < groups; grp += waves)
        for (unsigned long long load = grp * SHADE_GROUP; load < (grp + 1ull) * SHADE_GROUP && load < loads; ++load) {
            const unsigned long long q = load * 64ull + lane;
            bool alive = false;
            const uint32_t p = p_next;
            {
                const unsigned long long q_next = load + 1ull < (grp + 1ull) * SHADE_GROUP ? q + 64ull : ((grp + waves) * SHADE_GROUP) * 64ull + lane;
                if (q_next < n) p_next = qin[q_next];
            }
            if (q < n) {
                PathRec* rec = ps.rec + p;
                D3 beta = d3p(rec->beta);
                D3 L = d3p(rec->L);
                Best b;
                b.t = rec->hit_t; b.ref = rec->hit_ref; b.ctx = rec->hit_ctx;
                RT_PHASE(0);
                if (b.ref == RT_REF_NONE) { // camera.rs:258-260
                    L = L + beta * d3p(cam.background);
                } else {
                    RayD wr; // the world ray, read together with the rest of the record
                    wr.ox = rec->o[0]; wr.oy = rec->o[1]; wr.oz = rec->o[2];
                    wr.dx = rec->d[0]; wr.dy = rec->d[1]; wr.dz = rec->d[2];
                    double time = rec->time;
                    HitRec h;
                    finalize(S, wr, time, b, h, TEXTURES && S.uses_uv != 0u); // only an image texture reads u,v (texture.rs:88-107)
                    if (TEXTURES && ps.uv && (b.ref >> 28) == RT_REF_MEDIUM) { // a medium hit reports the u,v its record held (extend_kernel left them here)
                        const double2 uv = ps.uv[p];
                        h.u = uv.x; h.v = uv.y;
                    }
                    RT_PHASE(1);
                    uint32_t s_local = p / bg.npix, pix = p - s_local * bg.npix;
                    int32_t i = bg.x0 + (int32_t)(pix % (uint32_t)bg.tw);
                    int32_t j = bg.y0 + (int32_t)(pix / (uint32_t)bg.tw);
                    Rng g{seed, 0u, (uint32_t)(j * cam.image_width + i), (uint32_t)(bg.sample0 + (int32_t)s_local), rec->rng_index};
                    D3 ray_dir = d3(wr.dx, wr.dy, wr.dz);
                    D3 nd;
                    alive = shade_hit<TEXTURES, RR>(S, h, ray_dir, g, beta, L, nd);
                    RT_PHASE(2);
                    if (alive) {
                        if (last_bounce) { // the next ray_color call has depth 0 and returns black (camera.rs:252-254)
                            L = L + beta * d3(0.0, 0.0, 0.0);
                            alive = false;
                        } else {
                            rec->o[0] = h.p.x; rec->o[1] = h.p.y; rec->o[2] = h.p.z;
                            rec->d[0] = nd.x; rec->d[1] = nd.y; rec->d[2] = nd.z;
                            rec->beta[0] = beta.x; rec->beta[1] = beta.y; rec->beta[2] = beta.z;
                            rec->rng_index = g.index;
                        }
                    }
                }
                rec->L[0] = L.x; rec->L[1] = L.y; rec->L[2] = L.z;
            }
            RT_PHASE(3);
            const unsigned long long mask = __ballot(alive);
            if (alive) keep[kept + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = p;
            kept += (uint32_t)__popcll(mask);
            ++rounds;
            const bool last_of_wave = grp + waves >= groups && (load + 1ull == (grp + 1ull) * SHADE_GROUP || load + 1ull == loads);
            if ((rounds == (uint32_t)SHADE_ROUNDS || last_of_wave) && kept) {
                // the survivors collected so far take consecutive slots of the next queue
                unsigned long long slot = 0;
                if (lane == 0u) slot = atomicAdd(&ps.counters[C_LIVE_OUT], (unsigned long long)kept);
                const uint32_t slot_lo = __builtin_amdgcn_readfirstlane((uint32_t)slot);
                __builtin_amdgcn_wave_barrier();
                for (uint32_t k = lane; k < kept; k += 64u) qout[slot_lo + k] = keep[k];
                __builtin_amdgcn_wave_barrier();
                kept = 0;
            }
            if (rounds == (uint32_t)SHADE_ROUNDS) rounds = 0;
            RT_PHASE(4);
        }
    }
#ifdef RT_PHASE_STATS
    if (lane == 0u) for (int k = 0; k < 5; ++k) atomicAdd(&ps.counters[24 + k], ph[k]);
#endif
#undef RT_PHASE
}

// The last bounces of a batch in one launch.  Once few paths are left, every bounce is a pair of launches that
// last as long as their slowest wave (170-200 us per bounce in a BVH scene, for tens of bounces), with the chip
// nearly empty.  Here a lane keeps its path and alternates traversal and shading until the path ends: the batch
// then takes as long as its slowest path, not the sum over bounces of each bounce's slowest wave.  Same device
// functions in the same order as extend_kernel + shade_kernel, so the same bits.
template <int MEDIA, bool TEXTURES, bool RR, bool UV = false>
__global__ __launch_bounds__(BLOCK) void tail_kernel(DScene S, CamD cam, PathState ps, int qsel, unsigned long long seed, BatchGeom bg,
                                                     int first_bounce, int max_depth) {
    __shared__ uint32_t s_stack[STACK_MAX * BLOCK];
    uint32_t* stk = s_stack + threadIdx.x;
    const unsigned long long n = ps.counters[C_LIVE_IN];
    const uint32_t* queue = ps.queue[qsel];
    const unsigned long long threads = (unsigned long long)gridDim.x * blockDim.x;
    unsigned long long traced = 0;
    uint32_t nn = 0, np = 0;
    for (unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += threads) {
        const uint32_t p = queue[q];
        PathRec* rec = ps.rec + p;
        const WorldRay w = path_ray(ps, p); // reads rec->o / rec->d: they are kept current in memory below
        RayD r;
        r.ox = rec->o[0]; r.oy = rec->o[1]; r.oz = rec->o[2];
        r.dx = rec->d[0]; r.dy = rec->d[1]; r.dz = rec->d[2];
        const double time = rec->time;
        D3 beta = d3p(rec->beta);
        D3 L = d3p(rec->L);
        uint32_t s_local = p / bg.npix, pix = p - s_local * bg.npix;
        int32_t i = bg.x0 + (int32_t)(pix % (uint32_t)bg.tw);
        int32_t j = bg.y0 + (int32_t)(pix / (uint32_t)bg.tw);
        Rng g{seed, 0u, (uint32_t)(j * cam.image_width + i), (uint32_t)(bg.sample0 + (int32_t)s_local), rec->rng_index};
        for (int bounce = first_bounce; bounce < max_depth; ++bounce) {
            set_inverse(r);
            Best b;
            b.t = D_INF; b.ref = RT_REF_NONE; b.ctx = 0;
            UvTrack uvt;
            uvt.ref = RT_REF_NONE; uvt.ctx = 0; uvt.t = 0.0; uvt.any = false;
            traverse<MEDIA, false, true, true, false, UV>(S, S.world, r, 0, w, time, 0.001, b, stk, 0, &g, nn, np, false, &uvt); // Interval::new(0.001, INFINITY) (camera.rs:258)
            ++traced;
            if (b.ref == RT_REF_NONE) { // camera.rs:258-260
                L = L + beta * d3p(cam.background);
                break;
            }
            HitRec h;
            finalize(S, r, time, b, h, TEXTURES && S.uses_uv != 0u);
            if (UV && (b.ref >> 28) == RT_REF_MEDIUM) stale_uv(S, w, time, uvt, h.u, h.v);
            D3 ray_dir = d3(r.dx, r.dy, r.dz);
            D3 nd;
            if (!shade_hit<TEXTURES, RR>(S, h, ray_dir, g, beta, L, nd)) break;
            if (bounce == max_depth - 1) { // the next ray_color call has depth 0 and returns black (camera.rs:252-254)
                L = L + beta * d3(0.0, 0.0, 0.0);
                break;
            }
            r.ox = h.p.x; r.oy = h.p.y; r.oz = h.p.z;
            r.dx = nd.x; r.dy = nd.y; r.dz = nd.z;
            rec->o[0] = h.p.x; rec->o[1] = h.p.y; rec->o[2] = h.p.z;
            rec->d[0] = nd.x; rec->d[1] = nd.y; rec->d[2] = nd.z;
        }
        rec->L[0] = L.x; rec->L[1] = L.y; rec->L[2] = L.z;
    }
    // extend queries of this launch, for the counter the separate launches keep in advance_kernel
    for (int o = 32; o > 0; o >>= 1) traced += (unsigned long long)__shfl_xor((long long)traced, o);
    if ((threadIdx.x & 63u) == 0u && traced) atomicAdd(&ps.counters[C_RAYS], traced);
}

__global__ __launch_bounds__(BLOCK) void accumulate_kernel(PathState ps, BatchGeom bg, double* accum, int32_t image_width) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= bg.npix) return;
    int32_t i = bg.x0 + (int32_t)(pix % (uint32_t)bg.tw);
    int32_t j = bg.y0 + (int32_t)(pix / (uint32_t)bg.tw);
    double* a = accum + ((size_t)j * (size_t)image_width + (size_t)i) * 3;
    double r = a[0], g = a[1], b = a[2];
    for (int32_t s = 0; s < bg.nsamples; ++s) { // pixel_color += ray_color(...) in stratum order
        size_t p = (size_t)s * bg.npix + pix;
        const PathRec* rec = ps.rec + p;
        r += rec->L[0]; g += rec->L[1]; b += rec->L[2];
    }
    a[0] = r; a[1] = g; a[2] = b;
}

} // namespace

// ============================================================================ host code
// Measurement switches, read from the environment ONCE, when a scene is created (never per launch).
struct Tuning {
    int refill = -1, vote = -1, chunk = 0; // primitive-BVH kernel scheduling (MI355RT_REFILL, MI355RT_VOTE, MI355RT_CHUNK; -1 / 0: the defaults of bvh_kernel.h)
    int prep_refill = PREP_REFILL_THRESHOLD, prep_vote = PREP_LEAF_VOTE;               // the same for reordered batches
    bool quant = true;           // MI355RT_QUANT=0: keep the f32 4-wide nodes
    int postpone = -1;           // MI355RT_POSTPONE=0/1: overrides the choice made from the tree (postponed primitive tests)
    bool packed = true;          // MI355RT_PACKED=0: keep the 64-byte 8-bit nodes where the 48-byte packed tree exists
    bool fast_extend = true;     // MI355RT_FAST_EXTEND=0: the renderer keeps the general traversal
    bool shade_textures = false; // MI355RT_SHADE_TEXTURES=1: always the full shading kernel
    int reorder = 1;             // MI355RT_REORDER: binning pass in front of large closest-hit batches: 0 never, 1 where it pays (below), 2 always
    bool debug = false;          // MI355RT_DEBUG: diagnostics on stderr
    uint64_t device_bvh_min = 1ull << 15; // MI355RT_DEVICE_BVH_MIN: BVHs over at least this many objects are built on the device (0 = never)
    int fast_blocks_per_cu = 0;  // MI355RT_FAST_BLOCKS_PER_CU: cap on the primitive-BVH kernel's workgroups per CU (occupancy experiments)
    uint64_t reorder_min = 1ull << 18; // smallest batch that is binned
    uint64_t tail_paths = 1ull << 16;
    uint64_t max_paths = 1ull << 27;
};

constexpr int HIT_RING = 32;

struct rt_scene {
    FlatScene flat;
    DScene dev{};
    bool host_only = false;
    int device = -1;
    std::vector<void*> allocs;
    uint64_t device_bytes = 0;
    unsigned long long* d_counters = nullptr;
    unsigned long long* h_counters = nullptr; // pinned
    // HIP events of the closest-hit calls since the counters were last read: a ring of HIT_RING launches, each with
    // {call start, traversal start, traversal end, call end}, harvested by rt_get_counters
    hipEvent_t ev_hit[HIT_RING][4] = {};
    bool hit_reordered_at[HIT_RING] = {};
    uint64_t hit_launches = 0, hit_harvested = 0;
    hipEvent_t ev_r0 = nullptr, ev_r1 = nullptr;
    bool render_timed = false;
    rt_counters counters{};
    int grid_blocks = 0;
    int fast_grid_blocks = 0;
    int fast_grid_blocks_prep = 0; // the same for launches on prepared rays (reorder.h), whose kernel variant fits more waves
    int extend_blocks = 0, shade_blocks = 0, tail_blocks = 0;
    size_t fast_lds = 0;
    uint2* fast_spill = nullptr; // HBM overflow of the per-lane stacks: [entry][thread]
    int32_t* hit_ctx = nullptr;  // transform context of every hit of the current general closest-hit batch
    uint64_t hit_ctx_cap = 0;
    int fast_prim = 0;           // leaf kind of the primitive BVH when uniform, else 0
    int fast_fmt = NODES_F32;    // node records the primitive-BVH kernel reads (bvh_kernel.h)
    uint32_t fast_entry = RT_REF_NONE; // its root in that format
    float fast_extent = 0.0f; // largest |coordinate| stored in the 4-wide nodes (error bound of the f32 slab test)
    // render workspace
    PathState ps{};
    uint64_t ws_paths = 0;
    std::vector<void*> ws_allocs;
    // reorder pass in front of the primitive-BVH kernel (reorder.h): prepared records in bin order, the bin of every
    // ray, histogram and bin cursors; grown to the largest batch seen
    PRay* prep = nullptr;
    uint16_t* prep_bins = nullptr;
    uint32_t* prep_hist = nullptr;   // REORDER_BINS words, zero between batches
    uint32_t* prep_cursor = nullptr; // REORDER_BINS words
    uint64_t prep_cap = 0;
    ReorderGrid grid{};
    Tuning tune;
};

namespace {

thread_local std::string g_err;

int set_err(int code, const std::string& m) { g_err = m; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_err(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// Device address of `scratch` on every GPU the library has set up a scene on.  Each GPU holds its own copy of the
// array at its own address, so the address is resolved once per device, with that device current.
constexpr int NOALLOC_MAX_DEVICES = 64;
int32_t* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
std::mutex g_scratch_mutex;

int resolve_scratch(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return RT_OK; // such a device simply has no allocation-free entry point
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RT_OK;
    int32_t* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = ptr;
    return RT_OK;
}

template <class T>
int upload(rt_scene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 256); // never a null pointer in the kernels
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    s->device_bytes += bytes;
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RT_OK;
}

using FastKernel = void (*)(DScene, uint32_t, rt_ray_batch, unsigned long long, RayQueue, uint32_t, double, float, double*, uint32_t*,
                            unsigned long long*, unsigned long long*, uint2*, int, int, int, int, const PRay*);
// fmt: NODES_F32 / NODES_8BIT / NODES_PACKED (bvh_kernel.h); the packed tree exists for triangle meshes only
template <bool COUNT, int FMT, int RS, bool PREP> FastKernel fast_kernel_for(int prim) {
    if (prim == (int)RT_REF_TRI) return bvh4_closest_hit_kernel<RT_REF_TRI, COUNT, FMT, RS, PREP>;
    if (FMT == NODES_PACKED) return nullptr;
    if (prim == (int)RT_REF_SPHERE) return bvh4_closest_hit_kernel<RT_REF_SPHERE, COUNT, FMT == NODES_PACKED ? NODES_8BIT : FMT, RS, PREP>;
    return bvh4_closest_hit_kernel<0, COUNT, FMT == NODES_PACKED ? NODES_8BIT : FMT, RS, PREP>;
}
template <int RS, bool PREP> FastKernel fast_kernel_pick(int prim, bool count, int fmt) {
    if (fmt == NODES_PACKED) return count ? fast_kernel_for<true, NODES_PACKED, RS, PREP>(prim) : fast_kernel_for<false, NODES_PACKED, RS, PREP>(prim);
    if (fmt == NODES_8BIT) return count ? fast_kernel_for<true, NODES_8BIT, RS, PREP>(prim) : fast_kernel_for<false, NODES_8BIT, RS, PREP>(prim);
    return count ? fast_kernel_for<true, NODES_F32, RS, PREP>(prim) : fast_kernel_for<false, NODES_F32, RS, PREP>(prim);
}
FastKernel fast_kernel(int prim, bool count, int fmt) { return fast_kernel_pick<1, false>(prim, count, fmt); }          // structure-of-arrays batches of the C ABI
FastKernel fast_kernel_prepared(int prim, bool count, int fmt) { return fast_kernel_pick<1, true>(prim, count, fmt); }  // batches binned and prepared by the reorder pass
FastKernel fast_kernel_paths(int prim, int fmt) { return fast_kernel_pick<(int)PATH_DOUBLES, false>(prim, false, fmt); } // strided views of the renderer's path records
int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
Tuning read_tuning() {
    Tuning t;
    t.refill = env_int("MI355RT_REFILL", t.refill);
    t.vote = env_int("MI355RT_VOTE", t.vote);
    t.chunk = env_int("MI355RT_CHUNK", t.chunk);
    t.packed = env_int("MI355RT_PACKED", 1) != 0;
    t.postpone = env_int("MI355RT_POSTPONE", -1);
    t.prep_refill = env_int("MI355RT_PREP_REFILL", t.prep_refill);
    t.prep_vote = env_int("MI355RT_PREP_VOTE", t.prep_vote);
    t.quant = env_int("MI355RT_QUANT", 1) != 0;
    t.fast_extend = env_int("MI355RT_FAST_EXTEND", 1) != 0;
    t.shade_textures = env_int("MI355RT_SHADE_TEXTURES", 0) != 0;
    t.reorder = env_int("MI355RT_REORDER", 1);
    t.debug = getenv("MI355RT_DEBUG") != nullptr;
    t.fast_blocks_per_cu = env_int("MI355RT_FAST_BLOCKS_PER_CU", 0);
    t.device_bvh_min = (uint64_t)std::max(0, env_int("MI355RT_DEVICE_BVH_MIN", (int)t.device_bvh_min));
    t.reorder_min = (uint64_t)std::max(1, env_int("MI355RT_REORDER_MIN", (int)t.reorder_min));
    t.tail_paths = (uint64_t)std::max(0, env_int("MI355RT_TAIL_PATHS", TAIL_PATHS));
    // Paths per render batch.  17 GB of path records and queues when a frame is that large (the workspace is sized
    // to the frame, up to this cap): every bounce is a launch whose tail runs at one ray's latency, so few large
    // batches beat many small ones (Cornell 600x600x100: 798 / 868 / 934 M samples/s at 2^24 / 2^25 / 2^26).
    // Path ids and queue slots are 32-bit, so the cap stays below 2^31.
    if (const char* e = getenv("MI355RT_MAX_PATHS")) {
        const long long v = atoll(e);
        if (v >= 1024) t.max_paths = (uint64_t)std::min<long long>(v, 1ll << 31);
    }
    return t;
}
int pick_grid(rt_scene* s) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, s->device));
    int per_cu = 0;
    if (s->flat.medium_depth > 1) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closest_hit_kernel<false, true, 2>, BLOCK, 0));
    else if (s->flat.has_medium) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closest_hit_kernel<false, true, 1>, BLOCK, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closest_hit_kernel<false, true, false>, BLOCK, 0));
    int per_cu2 = 0;
    if (s->flat.medium_depth > 1) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, extend_kernel<2>, BLOCK, 0));
    else if (s->flat.has_medium) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, extend_kernel<1>, BLOCK, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, extend_kernel<false>, BLOCK, 0));
    int per_cu3 = 0;
    // (grids are sized for the variants without the roulette rule; a render with the rule runs on the same grid)
    if (s->flat.has_textures) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu3, shade_kernel<true, false>, BLOCK, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu3, shade_kernel<false, false>, BLOCK, 0));
    s->grid_blocks = prop.multiProcessorCount * std::max(1, per_cu);      // batched closest hit (general kernel)
    s->extend_blocks = prop.multiProcessorCount * std::max(1, per_cu2);   // each kernel fills the chip at its own occupancy
    s->shade_blocks = prop.multiProcessorCount * std::max(1, per_cu3);
    int per_cu4 = 0;
    if (s->flat.medium_depth > 1) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, tail_kernel<2, true, false>, BLOCK, 0));
    else if (s->flat.has_medium) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, tail_kernel<1, true, false>, BLOCK, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu4, tail_kernel<false, true, false>, BLOCK, 0));
    s->tail_blocks = prop.multiProcessorCount * std::max(1, per_cu4);
    if (s->flat.fast_root != RT_REF_NONE) {
        s->fast_lds = 0;
        s->fast_extent = s->flat.q_extent;
        const bool tr = !s->flat.tris.empty(), sp = !s->flat.spheres.empty(), qd = !s->flat.quads.empty();
        s->fast_prim = (tr && !sp && !qd) ? (int)RT_REF_TRI : ((sp && !tr && !qd) ? (int)RT_REF_SPHERE : 0);
        int fast_per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fast_per_cu, fast_kernel(s->fast_prim, false, s->fast_fmt), FAST_BLOCK, 0));
        if (s->tune.debug)
            fprintf(stderr, "[mi355rt] primitive-BVH kernel: node format=%d (inflation %.3f, leaf overlap %.2f) blocks/CU=%d\n", s->fast_fmt, s->flat.quant_inflation,
                    s->flat.fast_leaf_overlap, fast_per_cu);
        // Workgroups per CU.  More waves hide more latency, but they also share the CU's vector L1 and the XCD's L2:
        // measured on 2^24 reordered rays, 4 / 5 workgroups take 7.56 / 7.06 ms on the 2^20-triangle soup and
        // 3.52 / 3.68 ms on a closed surface of 1.3 M triangles, whose rays of one cell walk the same few nodes.  Five
        // (the kernel has 81 registers), or four where rays cross few triangle boxes.
        if (s->fast_fmt == NODES_PACKED) fast_per_cu = std::min(fast_per_cu, s->flat.fast_leaf_overlap > POSTPONE_MIN_OVERLAP ? 5 : 4);
        else fast_per_cu = std::min(fast_per_cu, 5);
        if (s->tune.fast_blocks_per_cu > 0) fast_per_cu = std::min(fast_per_cu, s->tune.fast_blocks_per_cu);
        s->fast_grid_blocks = prop.multiProcessorCount * std::max(1, fast_per_cu);
        // The variant that reads prepared rays is built to 80 registers, six waves per SIMD (bvh_kernel.h: no spills;
        // the soup's traversal 6.72 -> 6.49 ms; 72 registers for seven waves spill ten and take 7.7 ms).
        int prep_per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&prep_per_cu, fast_kernel_prepared(s->fast_prim, false, s->fast_fmt), FAST_BLOCK, 0));
        if (s->fast_fmt == NODES_PACKED) prep_per_cu = std::min(prep_per_cu, s->flat.fast_leaf_overlap > POSTPONE_MIN_OVERLAP ? 6 : 4);
        else prep_per_cu = std::min(prep_per_cu, 5);
        if (s->tune.fast_blocks_per_cu > 0) prep_per_cu = std::min(prep_per_cu, s->tune.fast_blocks_per_cu);
        s->fast_grid_blocks_prep = prop.multiProcessorCount * std::max(1, prep_per_cu);
        {
            size_t entries = (size_t)std::max<uint32_t>(s->flat.fast_stack, 1u);
            size_t bytes = entries * (size_t)std::max(s->fast_grid_blocks, s->fast_grid_blocks_prep) * FAST_BLOCK * sizeof(uint2);
            void* p = nullptr;
            HIP_TRY(hipMalloc(&p, bytes));
            s->allocs.push_back(p);
            s->device_bytes += bytes;
            s->fast_spill = static_cast<uint2*>(p);
        }
    }
    return RT_OK;
}

void fill_host_dscene(rt_scene* s) {
    FlatScene& f = s->flat;
    DScene& d = s->dev;
    d.nodes = f.nodes.data(); d.wnodes = f.wnodes.data(); d.qnodes = f.qnodes.data(); d.cqnodes = f.cqnodes.data(); d.pnodes = f.pnodes.data(); d.ptris = f.ptris.data(); d.spheres = f.spheres.data(); d.quads = f.quads.data(); d.tris = f.tris.data();
    d.lists = f.lists.data(); d.list_items = f.list_items.data(); d.xforms = f.xforms.data();
    d.media = f.media.data(); d.materials = f.materials.data(); d.textures = f.textures.data(); d.images = f.images.data();
    d.image_bytes = f.image_bytes.data(); d.perlins = f.perlins.data(); d.lights = f.lights.data();
    d.world = f.world; d.n_lights = f.n_lights; d.lights_is_list = f.lights_is_list; d.q_extent = f.q_extent; d.uses_uv = f.uses_uv ? 1u : 0u; d.world_rest = f.world_rest;
}

int upload_scene(rt_scene* s) {
    FlatScene& f = s->flat;
    DScene& d = s->dev;
    int rc;
    if ((rc = upload(s, f.nodes, &d.nodes))) return rc;
    if ((rc = upload(s, f.wnodes, &d.wnodes))) return rc;
    // the f32 4-wide nodes always go to the device (the general kernel's ordered traversal enters them at every BVH
    // root); the 8-bit copy only when the primitive-BVH kernel will read it
    // (MI355RT_QUANT=0 keeps that kernel on the f32 nodes even when the 8-bit ones were accepted, for A/B measurements)
    // MI355RT_PACKED=0 keeps it on the 64-byte nodes when the packed tree exists
    s->fast_fmt = (!f.cqnodes.empty() && s->tune.quant) ? ((!f.pnodes.empty() && s->tune.packed) ? NODES_PACKED : NODES_8BIT) : NODES_F32;
    s->fast_entry = s->fast_fmt == NODES_PACKED ? make_ref(REF_QNODE, 0u) : f.fast_root;
    const std::vector<CQNode> no_cqnodes;
    const std::vector<PNode> no_pnodes;
    const std::vector<PTri> no_ptris;
    if ((rc = upload(s, f.qnodes, &d.qnodes))) return rc;
    // (a batch that brings its own tmin array may hold negative ones, which the packed nodes' integer sort keys do not
    // cover: such a batch is traced on the 64-byte nodes, so both copies go to the device)
    if ((rc = upload(s, s->fast_fmt != NODES_F32 ? f.cqnodes : no_cqnodes, &d.cqnodes))) return rc;
    if ((rc = upload(s, s->fast_fmt == NODES_PACKED ? f.pnodes : no_pnodes, &d.pnodes))) return rc;
    if ((rc = upload(s, s->fast_fmt == NODES_PACKED ? f.ptris : no_ptris, &d.ptris))) return rc;
    if ((rc = upload(s, f.spheres, &d.spheres))) return rc;
    if ((rc = upload(s, f.quads, &d.quads))) return rc;
    if ((rc = upload(s, f.tris, &d.tris))) return rc;
    if ((rc = upload(s, f.lists, &d.lists))) return rc;
    if ((rc = upload(s, f.list_items, &d.list_items))) return rc;
    if ((rc = upload(s, f.xforms, &d.xforms))) return rc;
    if ((rc = upload(s, f.media, &d.media))) return rc;
    if ((rc = upload(s, f.materials, &d.materials))) return rc;
    if ((rc = upload(s, f.textures, &d.textures))) return rc;
    if ((rc = upload(s, f.images, &d.images))) return rc;
    if ((rc = upload(s, f.image_bytes, &d.image_bytes))) return rc;
    if ((rc = upload(s, f.perlins, &d.perlins))) return rc;
    if ((rc = upload(s, f.lights, &d.lights))) return rc;
    d.world = f.world; d.n_lights = f.n_lights; d.lights_is_list = f.lights_is_list; d.q_extent = f.q_extent; d.uses_uv = f.uses_uv ? 1u : 0u; d.world_rest = f.world_rest;
    void* c = nullptr;
    HIP_TRY(hipMalloc(&c, C_ALLOC * sizeof(unsigned long long)));
    s->allocs.push_back(c);
    s->d_counters = static_cast<unsigned long long*>(c);
    HIP_TRY(hipMemset(c, 0, C_ALLOC * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_counters), C_WORDS * sizeof(unsigned long long)));
    for (int k = 0; k < HIT_RING; ++k)
        for (int e = 0; e < 4; ++e) HIP_TRY(hipEventCreate(&s->ev_hit[k][e]));
    HIP_TRY(hipEventCreate(&s->ev_r0));
    HIP_TRY(hipEventCreate(&s->ev_r1));
    {   // grid of the reorder pass: 16 cells per axis over the world's box (a degenerate or unbounded axis gets one cell)
        const double* wb = s->flat.world_bounds.b;
        for (int k = 0; k < 3; ++k) {
            const double lo = wb[2 * k], ext = wb[2 * k + 1] - wb[2 * k];
            const bool ok = std::isfinite(lo) && std::isfinite(ext) && ext > 0.0;
            s->grid.lo[k] = ok ? lo : 0.0;
            s->grid.scale[k] = ok ? (double)(1 << REORDER_BITS) / ext : 0.0;
        }
    }
    int rc2 = resolve_scratch(s->device); // the device is current here (rt_scene_create)
    if (rc2) return rc2;
    return pick_grid(s);
}

void free_workspace(rt_scene* s) {
    for (void* p : s->ws_allocs) (void)hipFree(p);
    s->ws_allocs.clear();
    s->ws_paths = 0;
}

int ensure_workspace(rt_scene* s, uint64_t paths) {
    if (s->ws_paths >= paths) return RT_OK;
    free_workspace(s);
    auto alloc = [&](void** out, size_t bytes) -> int {
        HIP_TRY(hipMalloc(out, bytes));
        s->ws_allocs.push_back(*out);
        return RT_OK;
    };
    PathState& ps = s->ps;
    int rc;
    if ((rc = alloc(reinterpret_cast<void**>(&ps.rec), paths * sizeof(PathRec)))) return rc;
    if ((rc = alloc(reinterpret_cast<void**>(&ps.queue[0]), paths * sizeof(uint32_t)))) return rc;
    if ((rc = alloc(reinterpret_cast<void**>(&ps.queue[1]), paths * sizeof(uint32_t)))) return rc;
    ps.uv = nullptr;
    if (s->flat.track_uv && (rc = alloc(reinterpret_cast<void**>(&ps.uv), paths * sizeof(double2)))) return rc;
    ps.counters = s->d_counters;
    s->ws_paths = paths;
    return RT_OK;
}

int ensure_reorder_scratch(rt_scene* s, uint64_t n) {
    if (s->prep_cap >= n) return RT_OK;
    if (s->prep) (void)hipFree(s->prep);
    if (s->prep_bins) (void)hipFree(s->prep_bins);
    s->prep = nullptr; s->prep_bins = nullptr; s->prep_cap = 0;
    void* p = nullptr;
    if (!s->prep_hist) {
        HIP_TRY(hipMalloc(&p, 2 * REORDER_BINS * sizeof(uint32_t)));
        s->prep_hist = static_cast<uint32_t*>(p);
        s->prep_cursor = s->prep_hist + REORDER_BINS;
        HIP_TRY(hipMemset(p, 0, 2 * REORDER_BINS * sizeof(uint32_t)));
    }
    HIP_TRY(hipMalloc(&p, n * sizeof(PRay)));
    s->prep = static_cast<PRay*>(p);
    HIP_TRY(hipMalloc(&p, n * sizeof(uint16_t)));
    s->prep_bins = static_cast<uint16_t*>(p);
    s->prep_cap = n;
    return RT_OK;
}

// ctx_scratch: where the general kernels keep the transform context of every hit between traversal and record
// completion (n words); nullptr = the scene's own buffer, grown with hipMalloc to the largest batch seen.
int launch_closest_hit(rt_scene* s, const rt_ray_batch* rays, uint64_t n, rt_hit_batch* hits, uint64_t seed, uint32_t flags, hipStream_t st,
                       int32_t* ctx_scratch = nullptr) {
    if (n == 0) return RT_OK;
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(unsigned long long), st)); // C_WORK
    HIP_TRY(hipMemsetAsync(s->d_counters + C_WORK_SLOTS, 0, WORK_SLOTS * WORK_STRIDE * sizeof(unsigned long long), st));
    uint64_t waves = (n + 63) / 64;
    int blocks = (int)std::min<uint64_t>((uint64_t)s->grid_blocks, (waves + (BLOCK / 64) - 1) / (BLOCK / 64));
    hipEvent_t* const ev = s->ev_hit[s->hit_launches % HIT_RING]; // start, traversal start, traversal end, end
    HIP_TRY(hipEventRecord(ev[0], st));
    // RT_HIT_ORDERED | RT_HIT_EXACT_CULL: the exact binary nodes, children ordered by the split axis (general kernel)
    const bool exact_cull = (flags & RT_HIT_ORDERED) && (flags & RT_HIT_EXACT_CULL);
    const bool count = flags & RT_HIT_COUNTERS, ordered = (flags & RT_HIT_ORDERED) && !exact_cull;
    const Tuning& tn = s->tune;
    if (ordered && s->flat.fast_root != RT_REF_NONE && !s->flat.fast_partial) {
        // the world is one BVH over primitives: dedicated traversal kernel + one pass that completes the records
        int fblocks = (int)std::min<uint64_t>((uint64_t)s->fast_grid_blocks, (waves + (FAST_BLOCK / 64) - 1) / (FAST_BLOCK / 64));
        RayQueue rq{nullptr, nullptr};
        const uint32_t opt = (rays->time ? 1u : 0u) | (rays->tmin ? 2u : 0u) | (rays->tmax ? 4u : 0u);
        // large batches are binned by origin cell and prepared first (reorder.h); the allocation-free entry point
        // (ctx_scratch given) never grows the pass's scratch and so goes without it
        // The pass costs 0.7 ms per 2^24 rays and buys locality and a cheaper refill.  That pays where a ray crosses many
        // leaf boxes, i.e. walks long: on the 2^20-triangle soup (leaf overlap 70) a step takes 7.1 instead of 9.4 ms; on a
        // closed surface of 1.3 M triangles (overlap 3.9) the traversal gains 0.1 ms and the step loses 0.6, and every
        // sphere scene measured loses too (486 spheres: half the rate).  So: packed trees whose leaf overlap is high.
        const bool pays = tn.reorder >= 2 || (s->fast_fmt == NODES_PACKED && s->flat.fast_leaf_overlap > POSTPONE_MIN_OVERLAP);
        const bool reorder = tn.reorder && pays && !ctx_scratch && n >= tn.reorder_min && n <= 0xFFFFFFFFull;
        if (reorder && !(flags & RT_HIT_COUNTERS)) fblocks = (int)std::min<uint64_t>((uint64_t)s->fast_grid_blocks_prep, // (the counting variant has more registers: it keeps the smaller grid)
                                                                                 (waves + (FAST_BLOCK / 64) - 1) / (FAST_BLOCK / 64));
        if (reorder) {
            int rc = ensure_reorder_scratch(s, n);
            if (rc) return rc;
            const int cblocks = (int)std::min<uint64_t>(2048, (n + REORDER_BLOCK - 1) / REORDER_BLOCK);
            hipLaunchKernelGGL(reorder_count_kernel, dim3(cblocks), dim3(REORDER_BLOCK), 0, st, *rays, (unsigned long long)n, s->grid, s->prep_bins, s->prep_hist);
            hipLaunchKernelGGL(reorder_scan_kernel, dim3(1), dim3(1024), 0, st, s->prep_hist, s->prep_cursor);
            // (the packed tree takes 64-byte records; it is not used when the batch brings lower ends of its own, see below)
            hipLaunchKernelGGL((s->fast_fmt == NODES_PACKED && !(opt & 2u)) ? reorder_scatter_kernel<true> : reorder_scatter_kernel<false>,
                               dim3((unsigned)((n + REORDER_TILE - 1) / REORDER_TILE)), dim3(SCATTER_BLOCK), 0, st, *rays,
                               (unsigned long long)n, opt, 0.001, s->fast_extent, (const uint16_t*)s->prep_bins, s->prep_cursor, s->prep);
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(ev[1], st));
        // the packed nodes order children by keys that need entry distances >= 0, i.e. tmin >= 0: certain only for the default tmin
        const int fmt = (s->fast_fmt == NODES_PACKED && (opt & 2u)) ? NODES_8BIT : s->fast_fmt;
        const uint32_t entry = fmt == s->fast_fmt ? s->fast_entry : s->flat.fast_root;
        // lanes go on past a primitive they have reached where a ray crosses many triangle boxes (bvh_kernel.h);
        // measured on reordered batches: the 2^20-triangle soup (overlap 70) gains 3 %, a closed surface of 1.3 M triangles
        // (overlap 3.9) loses 3 %; on batches traced as they come closed surfaces of 82 k and 1.3 M triangles gain 1-3 %
        const int postpone = tn.postpone >= 0 ? tn.postpone : (fmt == NODES_PACKED && (!reorder || s->flat.fast_leaf_overlap > POSTPONE_MIN_OVERLAP) ? 1 : 0);
        const int refill = reorder ? tn.prep_refill : (tn.refill >= 0 ? tn.refill : HIT_REFILL_THRESHOLD), vote = reorder ? tn.prep_vote : (tn.vote >= 0 ? tn.vote : HIT_LEAF_VOTE);
        const FastKernel kernel = reorder ? fast_kernel_prepared(s->fast_prim, count, fmt) : fast_kernel(s->fast_prim, count, fmt);
        if (count) { // the counting variants have more registers than the grids are sized for: every workgroup of a launch must be resident
            int per_cu = 0, cus = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, FAST_BLOCK, 0));
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->device));
            fblocks = std::min(fblocks, std::max(1, per_cu * cus));
        }
        hipLaunchKernelGGL(kernel,
                           dim3(fblocks), dim3(FAST_BLOCK), 0, st, s->dev, entry, *rays, (unsigned long long)n, rq, opt, 0.001,
                           s->fast_extent, hits->t, hits->prim, s->d_counters + C_WORK_SLOTS, s->d_counters + C_NODES, s->fast_spill,
                           refill, vote, tn.chunk, postpone, (const PRay*)s->prep);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[2], st));
        if (hits->u || hits->v || hits->front_face || hits->mat || hits->nx || hits->ny || hits->nz || hits->px || hits->py || hits->pz)
            hipLaunchKernelGGL(finalize_hits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->dev, *rays, (unsigned long long)n, *hits, (const int32_t*)nullptr, 0);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(ev[3], st)); // the whole call: reorder pass, traversal, record completion
        s->hit_reordered_at[s->hit_launches % HIT_RING] = reorder;
        ++s->hit_launches;
        s->counters.rays += n;
        return RT_OK;
    }
    if (!ctx_scratch && s->hit_ctx_cap < n) { // scratch between the traversal and the record-completion pass, grown to the largest batch seen
        if (s->hit_ctx) (void)hipFree(s->hit_ctx);
        s->hit_ctx = nullptr; s->hit_ctx_cap = 0;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, n * sizeof(int32_t)));
        s->hit_ctx = static_cast<int32_t*>(p);
        s->hit_ctx_cap = n;
    }
    int32_t* const hit_ctx = ctx_scratch ? ctx_scratch : s->hit_ctx;
    // ordered traversal of a world whose first member has a primitive BVH of its own: that member through the
    // primitive-BVH kernel, the rest through the general one, which starts from the first member's closest hit
    const bool split = ordered && !count && s->flat.fast_root != RT_REF_NONE && s->flat.fast_partial;
    if (split) {
        int fblocks = (int)std::min<uint64_t>((uint64_t)s->fast_grid_blocks, (waves + (FAST_BLOCK / 64) - 1) / (FAST_BLOCK / 64));
        RayQueue rq{nullptr, nullptr};
        const uint32_t opt = (rays->time ? 1u : 0u) | (rays->tmin ? 2u : 0u) | (rays->tmax ? 4u : 0u);
        hipLaunchKernelGGL(fast_kernel(s->fast_prim, false, s->fast_fmt), dim3(fblocks), dim3(FAST_BLOCK), 0, st, s->dev, s->fast_entry, *rays,
                           (unsigned long long)n, rq, opt, 0.001, s->fast_extent, hits->t, hits->prim, s->d_counters + C_WORK_SLOTS, s->d_counters + C_NODES,
                           s->fast_spill, tn.refill >= 0 ? tn.refill : HIT_REFILL_THRESHOLD, tn.vote >= 0 ? tn.vote : HIT_LEAF_VOTE, tn.chunk, 0, (const PRay*)nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(s->d_counters + C_WORK_SLOTS, 0, WORK_SLOTS * WORK_STRIDE * sizeof(unsigned long long), st));
    }
    HIP_TRY(hipEventRecord(ev[1], st));
    auto launch = [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(BLOCK), 0, st, s->dev, *rays, (unsigned long long)n, *hits, hit_ctx, (unsigned long long)seed, s->d_counters,
                           split ? 1 : 0, exact_cull ? 1 : 0);
    };
    if (s->flat.track_uv) { // record history for a phase function that reads u,v: the two-level media variant serves every such scene
        if (count && ordered) launch(closest_hit_kernel<true, true, 2, true>);
        else if (count) launch(closest_hit_kernel<true, false, 2, true>);
        else if (ordered) launch(closest_hit_kernel<false, true, 2, true>);
        else launch(closest_hit_kernel<false, false, 2, true>);
    } else if (s->flat.medium_depth > 1) {
        if (count && ordered) launch(closest_hit_kernel<true, true, 2>);
        else if (count) launch(closest_hit_kernel<true, false, 2>);
        else if (ordered) launch(closest_hit_kernel<false, true, 2>);
        else launch(closest_hit_kernel<false, false, 2>);
    } else if (s->flat.has_medium) {
        if (count && ordered) launch(closest_hit_kernel<true, true, 1>);
        else if (count) launch(closest_hit_kernel<true, false, 1>);
        else if (ordered) launch(closest_hit_kernel<false, true, 1>);
        else launch(closest_hit_kernel<false, false, 1>);
    } else {
        if (count && ordered) launch(closest_hit_kernel<true, true, 0>);
        else if (count) launch(closest_hit_kernel<true, false, 0>);
        else if (ordered) launch(closest_hit_kernel<false, true, 0>);
        else launch(closest_hit_kernel<false, false, 0>);
    }
    HIP_TRY(hipEventRecord(ev[2], st));
    hipLaunchKernelGGL(finalize_hits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->dev, *rays, (unsigned long long)n, *hits, (const int32_t*)hit_ctx, s->flat.track_uv ? 1 : 0);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev[3], st));
    s->hit_reordered_at[s->hit_launches % HIT_RING] = false;
    ++s->hit_launches;
    s->counters.rays += n;
    return RT_OK;
}

int render_impl(rt_scene* s, const rt_camera* cam, uint64_t seed, const rt_work* work, double* d_accum, hipStream_t st) {
    CamD cd{};
    std::string err;
    int rc = camera_initialize(cam, &cd.d, err);
    if (rc) return set_err(rc, err);
    if (cam->max_depth < 0) return set_err(RT_E_INVALID, "camera: negative max_depth");
    cd.defocus_angle = cam->defocus_angle;
    cd.background[0] = cam->background[0]; cd.background[1] = cam->background[1]; cd.background[2] = cam->background[2];
    cd.image_width = cam->image_width;
    cd.russian_roulette = cam->russian_roulette;
    rt_work w;
    const int32_t nstrata = cd.d.sqrt_spp * cd.d.sqrt_spp;
    if (work) w = *work;
    else { w.x0 = 0; w.y0 = 0; w.x1 = cam->image_width; w.y1 = cd.d.image_height; w.sample_begin = 0; w.sample_end = nstrata; }
    if (w.x0 < 0 || w.y0 < 0 || w.x1 > cam->image_width || w.y1 > cd.d.image_height || w.x0 > w.x1 || w.y0 > w.y1)
        return set_err(RT_E_INVALID, "render: work rectangle outside the image");
    if (w.sample_begin < 0 || w.sample_end > nstrata || w.sample_begin > w.sample_end)
        return set_err(RT_E_INVALID, "render: sample range outside [0, sqrt_spp^2]");
    const int32_t tw = w.x1 - w.x0;
    if (tw == 0 || w.y0 == w.y1 || w.sample_begin == w.sample_end) return RT_OK;
    const uint64_t cap = s->tune.max_paths;
    if ((uint64_t)tw > cap) return set_err(RT_E_TOO_LARGE, "render: one image row exceeds the path workspace");
    const int32_t rows_per_tile = (int32_t)std::min<uint64_t>((uint64_t)(w.y1 - w.y0), cap / (uint64_t)tw);
    const uint64_t tile_pix_max = (uint64_t)rows_per_tile * (uint64_t)tw;
    const int32_t samples_per_batch = (int32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(w.sample_end - w.sample_begin), cap / tile_pix_max));
    if ((rc = ensure_workspace(s, tile_pix_max * (uint64_t)samples_per_batch))) return rc;
    PathState ps = s->ps;
    HIP_TRY(hipEventRecord(s->ev_r0, st));
    for (int32_t y = w.y0; y < w.y1; y += rows_per_tile) {
        const int32_t th = std::min(rows_per_tile, w.y1 - y);
        for (int32_t s0 = w.sample_begin; s0 < w.sample_end; s0 += samples_per_batch) {
            BatchGeom bg;
            bg.x0 = w.x0; bg.y0 = y; bg.tw = tw; bg.th = th;
            bg.sample0 = s0; bg.nsamples = std::min(samples_per_batch, w.sample_end - s0);
            bg.npix = (uint32_t)tw * (uint32_t)th;
            bg.npaths = bg.npix * (uint32_t)bg.nsamples;
            hipLaunchKernelGGL(init_batch_kernel, dim3(1), dim3(1), 0, st, s->d_counters, (unsigned long long)bg.npaths);
            hipLaunchKernelGGL(raygen_kernel, dim3((bg.npaths + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, cd, bg, ps, (unsigned long long)seed);
            s->counters.paths += bg.npaths;
            int qsel = 0;
            uint64_t live = bg.npaths;
            // MI355RT_FAST_EXTEND=0 keeps the general traversal for BVH-only worlds (A/B measurements)
            const bool fast_extend = s->flat.fast_root != RT_REF_NONE && s->tune.fast_extend;
            const bool has_tex = s->flat.has_textures || s->tune.shade_textures; // the switch forces the full shading code (A/B tests)
            const bool rr = cam->russian_roulette != 0;
            const int media = s->flat.medium_depth > 1 ? 2 : (s->flat.has_medium ? 1 : 0);
            const uint64_t tail_paths = s->tune.tail_paths;
            const int k_refill = s->tune.refill >= 0 ? s->tune.refill : REFILL_THRESHOLD, k_vote = s->tune.vote >= 0 ? s->tune.vote : LEAF_VOTE, k_chunk = s->tune.chunk;
            bool live_exact = true; // `live` is the device's count (it is refreshed only now and then)
            for (int32_t bounce = 0; bounce < cam->max_depth && live > 0; ++bounce) {
                if (live_exact && bounce > 0 && live <= tail_paths) {
                    // few paths left: they finish in one launch, each lane following its path to the end
                    const int tblocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)s->tail_blocks, (live + BLOCK - 1) / BLOCK));
                    // (nested media: the full shading code whatever the textures, to keep the number of variants down)
                    auto tail = s->flat.track_uv ? (rr ? tail_kernel<2, true, true, true> : tail_kernel<2, true, false, true>)
                              : media == 2 ? (rr ? tail_kernel<2, true, true> : tail_kernel<2, true, false>)
                              : rr ? (media ? (has_tex ? tail_kernel<1, true, true> : tail_kernel<1, false, true>)
                                            : (has_tex ? tail_kernel<0, true, true> : tail_kernel<0, false, true>))
                                   : (media ? (has_tex ? tail_kernel<1, true, false> : tail_kernel<1, false, false>)
                                            : (has_tex ? tail_kernel<0, true, false> : tail_kernel<0, false, false>));
                    hipLaunchKernelGGL(tail, dim3(tblocks), dim3(BLOCK), 0, st, s->dev, cd, ps, qsel, (unsigned long long)seed, bg, (int)bounce, (int)cam->max_depth);
                    s->counters.extend_launches += 1;
                    break;
                }
                live_exact = false;
                uint64_t waves = (live + 63) / 64;
                const uint64_t want = (waves + (BLOCK / 64) - 1) / (BLOCK / 64);
                const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)s->extend_blocks, want));
                const int sblocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)s->shade_blocks, want));
                if (fast_extend) {
                    // the world is one BVH over primitives: the primitive-BVH kernel reads the rays straight out of
                    // the path records (strided views) and writes hit_t / hit_ref back into them; no transforms and
                    // no media on this path, so hit_ctx (0 since raygen) and the RNG index are left alone
                    const double* rb = reinterpret_cast<const double*>(ps.rec);
                    rt_ray_batch view{rb + 0, rb + 1, rb + 2, rb + 3, rb + 4, rb + 5, rb + 6, nullptr, nullptr};
                    RayQueue rq{ps.queue[qsel], s->d_counters + C_LIVE_IN};
                    double* out_t = reinterpret_cast<double*>(ps.rec) + offsetof(PathRec, hit_t) / sizeof(double);
                    uint32_t* out_prim = reinterpret_cast<uint32_t*>(ps.rec) + offsetof(PathRec, hit_ref) / sizeof(uint32_t);
                    int fblocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)s->fast_grid_blocks, (waves + (FAST_BLOCK / 64) - 1) / (FAST_BLOCK / 64)));
                    hipLaunchKernelGGL(fast_kernel_paths(s->fast_prim, s->fast_fmt), dim3(fblocks), dim3(FAST_BLOCK), 0, st, s->dev, s->fast_entry,
                                       view, (unsigned long long)live, rq, 1u, 0.001, s->fast_extent, out_t, out_prim, s->d_counters + C_WORK_SLOTS,
                                       s->d_counters + C_NODES, s->fast_spill, k_refill, k_vote, k_chunk, 0, (const PRay*)nullptr);
                }
                if (!fast_extend || s->flat.fast_partial) { // the whole world, or what follows its first member
                    if (fast_extend) hipLaunchKernelGGL(reset_work_kernel, dim3(1), dim3(1), 0, st, s->d_counters);
                    auto extend = media == 2 ? extend_kernel<2> : (media ? extend_kernel<1> : extend_kernel<0>);
                    if (s->flat.track_uv) extend = extend_kernel<2, true>;
                    hipLaunchKernelGGL(extend, dim3(blocks), dim3(BLOCK), 0, st, s->dev, ps, qsel,
                                       (unsigned long long)seed, bg, cam->image_width, fast_extend ? 1 : 0);
                }
                hipLaunchKernelGGL(rr ? (has_tex ? shade_kernel<true, true> : shade_kernel<false, true>) : (has_tex ? shade_kernel<true, false> : shade_kernel<false, false>), dim3(sblocks), dim3(BLOCK), 0, st, s->dev, cd, ps, qsel, (unsigned long long)seed, bg, bounce == cam->max_depth - 1 ? 1 : 0);
                hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, s->d_counters);
                s->counters.extend_launches += 1;
                qsel ^= 1;
                // `live` only sizes the next launches and ends a dead batch early (the kernels read the exact count
                // on the device).  Each refresh is a host round trip with the GPU idle, which costs more than a few
                // over-sized launches, so it happens at bounces 2, 5, 9, then every 3rd (by then the batch is close to the
                // size at which tail_kernel takes over, and the refresh is what lets it).
                if (bounce == 2 || bounce == 5 || bounce == 9 || (bounce > 9 && bounce % 3 == 0)) {
                    HIP_TRY(hipMemcpyAsync(s->h_counters, s->d_counters, C_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
                    HIP_TRY(hipStreamSynchronize(st));
                    live = s->h_counters[C_LIVE_IN];
                    live_exact = true;
                }
            }
            hipLaunchKernelGGL(accumulate_kernel, dim3((bg.npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, ps, bg, d_accum, cam->image_width);
            HIP_TRY(hipGetLastError());
        }
    }
    HIP_TRY(hipEventRecord(s->ev_r1, st));
    s->render_timed = true;
    return RT_OK;
}

int need_device(rt_scene* s) {
    if (!s) return set_err(RT_E_INVALID, "null scene");
    if (s->host_only) return set_err(RT_E_HIP, "scene was created with RT_SCENE_HOST_ONLY: no device data, nothing can be traced");
    HIP_TRY(hipSetDevice(s->device));
    return RT_OK;
}

} // namespace

namespace rt { int set_last_error(int code, const std::string& message) { return set_err(code, message); } } // for obj.cpp

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }
const char* rt_version(void) { return "mi355rt 0.1 (gfx950)"; }

int rt_scene_create(const rt_scene_desc* desc, uint32_t flags, rt_scene** out) {
    if (!out) return set_err(RT_E_INVALID, "rt_scene_create: null out pointer");
    *out = nullptr;
    rt_scene* s = new rt_scene();
    s->tune = read_tuning();
    std::string err;
    if (!(flags & RT_SCENE_HOST_ONLY)) {
        hipError_t e = hipGetDevice(&s->device);
        if (e != hipSuccess) { delete s; return set_err(RT_E_HIP, std::string("no HIP device: ") + hipGetErrorString(e)); }
    }
    // large BVHs are built on the device (bvh_build.hip), in the same split order as the host recursion
    int rc = flatten_scene(desc, s->flat, err, (flags & RT_SCENE_HOST_ONLY) ? 0 : s->tune.device_bvh_min);
    if (rc) { delete s; return set_err(rc, err); }
    if (flags & RT_SCENE_HOST_ONLY) {
        s->host_only = true;
        fill_host_dscene(s);
    } else {
        rc = upload_scene(s);
        if (rc) { rt_scene_destroy(s); return rc; }
    }
    *out = s;
    return RT_OK;
}

int rt_scene_destroy(rt_scene* s) {
    if (!s) return RT_OK;
    if (!s->host_only) {
        (void)hipSetDevice(s->device);
        free_workspace(s);
        for (void* p : s->allocs) (void)hipFree(p);
        if (s->hit_ctx) (void)hipFree(s->hit_ctx);
        if (s->prep) (void)hipFree(s->prep);
        if (s->prep_bins) (void)hipFree(s->prep_bins);
        if (s->prep_hist) (void)hipFree(s->prep_hist);
        if (s->h_counters) (void)hipHostFree(s->h_counters);
        for (int k = 0; k < HIT_RING; ++k)
            for (int e = 0; e < 4; ++e) if (s->ev_hit[k][e]) (void)hipEventDestroy(s->ev_hit[k][e]);
        if (s->ev_r0) (void)hipEventDestroy(s->ev_r0);
        if (s->ev_r1) (void)hipEventDestroy(s->ev_r1);
    }
    delete s;
    return RT_OK;
}

int rt_scene_get_info(const rt_scene* s, rt_scene_info* info) {
    if (!s || !info) return set_err(RT_E_INVALID, "rt_scene_get_info: null argument");
    info->n_nodes = s->flat.nodes.size();
    info->n_lists = s->flat.lists.size();
    info->n_xforms = s->flat.xforms.size();
    info->n_media = s->flat.media.size();
    info->n_lights = s->flat.n_lights < 0 ? 0 : (uint64_t)s->flat.n_lights;
    info->world_ref = s->flat.world;
    info->max_stack = s->flat.max_stack;
    info->device_bytes = s->device_bytes;
    for (int k = 0; k < 6; ++k) info->world_bounds[k] = s->flat.world_bounds.b[k];
    return RT_OK;
}

int rt_scene_get_nodes(const rt_scene* s, uint64_t first, uint64_t count, rt_bvh_node* out) {
    if (!s || !out) return set_err(RT_E_INVALID, "rt_scene_get_nodes: null argument");
    if (first + count > s->flat.nodes.size()) return set_err(RT_E_INVALID, "rt_scene_get_nodes: range outside the node array");
    std::memcpy(out, s->flat.nodes.data() + first, count * sizeof(rt_bvh_node));
    return RT_OK;
}

int rt_scene_wide_node_count(const rt_scene* s, uint64_t* count) {
    if (!s || !count) return set_err(RT_E_INVALID, "rt_scene_wide_node_count: null argument");
    *count = s->flat.qnodes.size();
    return RT_OK;
}

int rt_scene_get_wide_nodes(const rt_scene* s, uint64_t first, uint64_t count, rt_wide_node* out) {
    if (!s || !out) return set_err(RT_E_INVALID, "rt_scene_get_wide_nodes: null argument");
    if (first + count > s->flat.qnodes.size()) return set_err(RT_E_INVALID, "rt_scene_get_wide_nodes: range outside the node array");
    const bool has8 = !s->flat.cqnodes.empty();
    for (uint64_t i = 0; i < count; ++i) {
        const QNode& q = s->flat.qnodes[first + i];
        rt_wide_node& o = out[i];
        o.has_box8 = has8 ? 1 : 0;
        o._pad = 0;
        for (int j = 0; j < 4; ++j) {
            // inner children are reported as RT_REF_NODE | wide index (the internal kind stays internal)
            // and a run of quads as the list it stands for
            uint32_t c = q.child[j];
            if (RT_REF_KIND(c) == REF_QNODE) c = (RT_REF_NODE << 28) | RT_REF_INDEX(c);
            else if (RT_REF_KIND(c) == REF_QUADS)
                for (const auto& qr : s->flat.quad_runs) if (qr.first == c) { c = qr.second; break; }
            o.child[j] = c;
            for (int a = 0; a < 3; ++a) {
                o.box32[j][2 * a] = q.lo[a][j];
                o.box32[j][2 * a + 1] = q.hi[a][j];
                o.box8[j][2 * a] = o.box8[j][2 * a + 1] = 0.0;
            }
        }
        if (has8) {
            const CQNode& c = s->flat.cqnodes[first + i];
            for (int a = 0; a < 3; ++a) {
                const double step = std::ldexp(1.0, (int)c.exp[a] - 127);
                for (int j = 0; j < 4; ++j) {
                    o.box8[j][2 * a] = (double)c.origin[a] + (double)c.qlo[a][j] * step;
                    o.box8[j][2 * a + 1] = (double)c.origin[a] + (double)c.qhi[a][j] * step;
                }
            }
        }
    }
    return RT_OK;
}

int rt_scene_packed_node_count(const rt_scene* s, uint64_t* count, double* leaf_overlap) {
    if (!s || !count) return set_err(RT_E_INVALID, "rt_scene_packed_node_count: null argument");
    *count = s->flat.pnodes.size();
    if (leaf_overlap) *leaf_overlap = s->flat.fast_leaf_overlap;
    return RT_OK;
}

int rt_scene_get_packed_nodes(const rt_scene* s, uint64_t first, uint64_t count, rt_wide_node* out) {
    if (!s || !out) return set_err(RT_E_INVALID, "rt_scene_get_packed_nodes: null argument");
    if (first + count > s->flat.pnodes.size()) return set_err(RT_E_INVALID, "rt_scene_get_packed_nodes: range outside the node array");
    for (uint64_t i = 0; i < count; ++i) {
        const PNode& n = s->flat.pnodes[first + i];
        rt_wide_node& o = out[i];
        o.has_box8 = 1;
        o._pad = 0;
        const uint32_t n_inner = n.counts & 15u, n_all = n.counts >> 4;
        for (uint32_t j = 0; j < 4; ++j) {
            // the arithmetic of the kernel: slot j is node_base + j below n_inner, leaf_base + j from there on
            uint32_t c = RT_REF_NONE;
            if (j < n_inner) c = (RT_REF_NODE << 28) | RT_REF_INDEX(n.node_base + j);
            else if (j < n_all) c = make_ref(RT_REF_TRI, s->flat.ptris[RT_REF_INDEX(n.leaf_base + j)].orig);
            o.child[j] = c;
            for (int a = 0; a < 3; ++a) {
                const double step = std::ldexp(1.0, (int)n.exp[a] - 127);
                o.box32[j][2 * a] = o.box32[j][2 * a + 1] = 0.0;
                o.box8[j][2 * a] = (double)n.origin[a] + (double)n.qlo[a][j] * step;
                o.box8[j][2 * a + 1] = (double)n.origin[a] + (double)n.qhi[a][j] * step;
            }
        }
    }
    return RT_OK;
}

int rt_scene_object_bounds(const rt_scene* s, int32_t object, double out[6]) {
    if (!s || !out) return set_err(RT_E_INVALID, "rt_scene_object_bounds: null argument");
    if (object < 0 || (size_t)object >= s->flat.object_bounds.size()) return set_err(RT_E_INVALID, "rt_scene_object_bounds: object id out of range");
    for (int k = 0; k < 6; ++k) out[k] = s->flat.object_bounds[object].b[k];
    return RT_OK;
}

int rt_closest_hit_device(rt_scene* s, const rt_ray_batch* rays, uint64_t n, rt_hit_batch* hits, uint64_t seed, uint32_t flags, void* hip_stream) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!rays || !hits) return set_err(RT_E_INVALID, "rt_closest_hit_device: null batch");
    if (n && (!rays->ox || !rays->oy || !rays->oz || !rays->dx || !rays->dy || !rays->dz || !hits->t || !hits->prim))
        return set_err(RT_E_INVALID, "rt_closest_hit_device: origin, direction, t and prim arrays are required");
    return launch_closest_hit(s, rays, n, hits, seed, flags, static_cast<hipStream_t>(hip_stream));
}

int rt_closest_hit_device_noalloc(rt_scene* s, const rt_ray_batch* rays, uint64_t n, rt_hit_batch* hits, uint64_t seed, uint32_t flags, void* hip_stream) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!rays || !hits) return set_err(RT_E_INVALID, "rt_closest_hit_device_noalloc: null batch");
    if (n && (!rays->ox || !rays->oy || !rays->oz || !rays->dx || !rays->dy || !rays->dz || !hits->t || !hits->prim))
        return set_err(RT_E_INVALID, "rt_closest_hit_device_noalloc: origin, direction, t and prim arrays are required");
    if (n > NOALLOC_MAX_RAYS)
        return set_err(RT_E_TOO_LARGE, "rt_closest_hit_device_noalloc: at most 16777216 (2^24) rays per call; split the batch or use rt_closest_hit_device");
    int32_t* const ptr = (s->device >= 0 && s->device < NOALLOC_MAX_DEVICES) ? g_scratch_ptr[s->device] : nullptr;
    if (!ptr) return set_err(RT_E_HIP, "rt_closest_hit_device_noalloc: the scratch array of this device was not resolved at scene creation");
    return launch_closest_hit(s, rays, n, hits, seed, flags, static_cast<hipStream_t>(hip_stream), ptr);
}

int rt_closest_hit(rt_scene* s, const rt_ray_batch* rays, uint64_t n, rt_hit_batch* hits, uint64_t seed, uint32_t flags) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!rays || !hits) return set_err(RT_E_INVALID, "rt_closest_hit: null batch");
    if (n == 0) return RT_OK;
    if (!rays->ox || !rays->oy || !rays->oz || !rays->dx || !rays->dy || !rays->dz || !hits->t || !hits->prim)
        return set_err(RT_E_INVALID, "rt_closest_hit: origin, direction, t and prim arrays are required");
    std::vector<void*> tmp;
    auto cleanup = [&]() { for (void* p : tmp) (void)hipFree(p); };
    auto to_dev = [&](const void* h, size_t bytes, const void** d) -> int {
        *d = nullptr;
        if (!h) return RT_OK;
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes));
        tmp.push_back(p);
        HIP_TRY(hipMemcpy(p, h, bytes, hipMemcpyHostToDevice));
        *d = p;
        return RT_OK;
    };
    auto out_dev = [&](const void* h, size_t bytes, void** d) -> int {
        *d = nullptr;
        if (!h) return RT_OK;
        HIP_TRY(hipMalloc(d, bytes));
        tmp.push_back(*d);
        return RT_OK;
    };
    rt_ray_batch dr{};
    rt_hit_batch dh{};
    const double* const src[9] = {rays->ox, rays->oy, rays->oz, rays->dx, rays->dy, rays->dz, rays->time, rays->tmin, rays->tmax};
    const double** dst[9] = {&dr.ox, &dr.oy, &dr.oz, &dr.dx, &dr.dy, &dr.dz, &dr.time, &dr.tmin, &dr.tmax};
    for (int k = 0; k < 9; ++k)
        if ((rc = to_dev(src[k], n * sizeof(double), reinterpret_cast<const void**>(dst[k])))) { cleanup(); return rc; }
    double* const hsrc[9] = {hits->t, hits->u, hits->v, hits->nx, hits->ny, hits->nz, hits->px, hits->py, hits->pz};
    double** hdst[9] = {&dh.t, &dh.u, &dh.v, &dh.nx, &dh.ny, &dh.nz, &dh.px, &dh.py, &dh.pz};
    for (int k = 0; k < 9; ++k)
        if ((rc = out_dev(hsrc[k], n * sizeof(double), reinterpret_cast<void**>(hdst[k])))) { cleanup(); return rc; }
    if ((rc = out_dev(hits->prim, n * sizeof(uint32_t), reinterpret_cast<void**>(&dh.prim)))) { cleanup(); return rc; }
    if ((rc = out_dev(hits->front_face, n * sizeof(uint8_t), reinterpret_cast<void**>(&dh.front_face)))) { cleanup(); return rc; }
    if ((rc = out_dev(hits->mat, n * sizeof(int32_t), reinterpret_cast<void**>(&dh.mat)))) { cleanup(); return rc; }
    rc = launch_closest_hit(s, &dr, n, &dh, seed, flags, nullptr);
    if (rc) { cleanup(); return rc; }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { cleanup(); return set_err(RT_E_HIP, std::string("closest_hit kernel: ") + hipGetErrorString(e)); }
    auto back = [&](void* h, const void* d, size_t bytes) -> int {
        if (!h) return RT_OK;
        HIP_TRY(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        return RT_OK;
    };
    for (int k = 0; k < 9; ++k)
        if ((rc = back(hsrc[k], *hdst[k], n * sizeof(double)))) { cleanup(); return rc; }
    if ((rc = back(hits->prim, dh.prim, n * sizeof(uint32_t)))) { cleanup(); return rc; }
    if ((rc = back(hits->front_face, dh.front_face, n * sizeof(uint8_t)))) { cleanup(); return rc; }
    if ((rc = back(hits->mat, dh.mat, n * sizeof(int32_t)))) { cleanup(); return rc; }
    cleanup();
    return RT_OK;
}

int rt_camera_initialize(const rt_camera* cam, rt_camera_derived* out) {
    std::string err;
    int rc = camera_initialize(cam, out, err);
    return rc ? set_err(rc, err) : RT_OK;
}

int rt_render_device(rt_scene* s, const rt_camera* cam, uint64_t seed, const rt_work* work, double* accum, void* hip_stream) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!cam || !accum) return set_err(RT_E_INVALID, "rt_render_device: null argument");
    return render_impl(s, cam, seed, work, accum, static_cast<hipStream_t>(hip_stream));
}

int rt_render(rt_scene* s, const rt_camera* cam, uint64_t seed, const rt_work* work, double* accum) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!cam || !accum) return set_err(RT_E_INVALID, "rt_render: null argument");
    rt_camera_derived d;
    std::string err;
    if ((rc = camera_initialize(cam, &d, err))) return set_err(rc, err);
    size_t bytes = (size_t)cam->image_width * (size_t)d.image_height * 3 * sizeof(double);
    double* dacc = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dacc), bytes));
    hipError_t e = hipMemcpy(dacc, accum, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = render_impl(s, cam, seed, work, dacc, nullptr);
        if (rc == RT_OK) {
            e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(accum, dacc, bytes, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(dacc);
    if (rc) return rc;
    if (e != hipSuccess) return set_err(RT_E_HIP, std::string("rt_render: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_resolve_rgb8(const rt_camera* cam, const double* accum, uint8_t* rgb) { // camera.rs:133-135, color.rs:58-80
    if (!cam || !accum || !rgb) return set_err(RT_E_INVALID, "rt_resolve_rgb8: null argument");
    rt_camera_derived d;
    std::string err;
    int rc = camera_initialize(cam, &d, err);
    if (rc) return set_err(rc, err);
    size_t n = (size_t)cam->image_width * (size_t)d.image_height * 3;
    for (size_t k = 0; k < n; ++k) {
        double x = d.pixel_samples_scale * accum[k];
        x = x > 0.0 ? std::sqrt(x) : 0.0;              // linear_to_gamma (NaN -> 0)
        x = x < 0.000 ? 0.000 : (x > 0.999 ? 0.999 : x); // Interval::clamp
        double b = 256.0 * x;
        rgb[k] = (uint8_t)(int32_t)b;
    }
    return RT_OK;
}

int rt_write_ppm(const char* path, int32_t width, int32_t height, const uint8_t* rgb) { // camera.rs:86,151-154
    if (!path || !rgb || width < 0 || height < 0) return set_err(RT_E_INVALID, "rt_write_ppm: bad argument");
    FILE* fp = std::fopen(path, "w");
    if (!fp) return set_err(RT_E_INVALID, std::string("rt_write_ppm: cannot open ") + path);
    std::fprintf(fp, "P3\n%d %d\n255\n", width, height);
    size_t n = (size_t)width * (size_t)height;
    for (size_t p = 0; p < n; ++p) std::fprintf(fp, "%d %d %d\n", (int)rgb[3 * p], (int)rgb[3 * p + 1], (int)rgb[3 * p + 2]);
    std::fclose(fp);
    return RT_OK;
}

int rt_get_counters(rt_scene* s, rt_counters* out) {
    if (!s || !out) return set_err(RT_E_INVALID, "rt_get_counters: null argument");
    if (!s->host_only) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(s->h_counters, s->d_counters, C_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        s->counters.node_visits = s->h_counters[C_NODES];
        s->counters.prim_tests = s->h_counters[C_PRIMS];
        // closest-hit calls since the last harvest (at most the HIT_RING newest still have their events)
        if (s->hit_launches - s->hit_harvested > (uint64_t)HIT_RING) s->hit_harvested = s->hit_launches - HIT_RING;
        for (; s->hit_harvested < s->hit_launches; ++s->hit_harvested) {
            hipEvent_t* const ev = s->ev_hit[s->hit_harvested % HIT_RING];
            float call = 0, trav = 0, pre = 0;
            HIP_TRY(hipEventElapsedTime(&call, ev[0], ev[3]));
            HIP_TRY(hipEventElapsedTime(&trav, ev[1], ev[2]));
            HIP_TRY(hipEventElapsedTime(&pre, ev[0], ev[1]));
            if (!s->hit_reordered_at[s->hit_harvested % HIT_RING]) pre = 0;
            s->counters.ms_last_closest_hit = call; s->counters.ms_last_traversal = trav; s->counters.ms_last_reorder = pre;
            s->counters.ms_sum_closest_hit += call; s->counters.ms_sum_traversal += trav; s->counters.ms_sum_reorder += pre;
            s->counters.timed_calls += 1;
        }
        if (s->render_timed) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, s->ev_r0, s->ev_r1)); s->counters.ms_last_render = ms; }
        s->counters.render_rays = s->h_counters[C_RAYS];
        if (s->tune.debug) {
#ifdef RT_PHASE_STATS
            std::fprintf(stderr, "[mi355rt] shade wave cycles: record load %llu hit record %llu material %llu write-back %llu queue %llu\n", s->h_counters[24],
                         s->h_counters[25], s->h_counters[26], s->h_counters[27], s->h_counters[28]);
            std::fprintf(stderr, "[mi355rt] traversal loop iterations: sum over lanes %llu, wave maximum x active lanes %llu\n", s->h_counters[29], s->h_counters[30]);
            std::fprintf(stderr, "[mi355rt] traversal step cycles: wide %llu exact %llu prim %llu list %llu xform %llu - %llu pop %llu\n", s->h_counters[8],
                         s->h_counters[9], s->h_counters[10], s->h_counters[11], s->h_counters[12], s->h_counters[13], s->h_counters[14]);
#endif
            std::fprintf(stderr, "[mi355rt] wave diag: iters %llu node_lanes %llu tri_phases %llu tri_lanes %llu pop_iters %llu pop_lanes %llu rounds %llu run_lanes %llu\n",
                         s->h_counters[8], s->h_counters[9], s->h_counters[10], s->h_counters[11], s->h_counters[12], s->h_counters[13], s->h_counters[14], s->h_counters[15]);
            std::fprintf(stderr, "[mi355rt] wave cycles (s_memtime, summed over waves): refill %llu node %llu leaf %llu pop %llu (64-ray pieces %llu)\n", s->h_counters[16],
                         s->h_counters[17], s->h_counters[18], s->h_counters[19], s->h_counters[20]);
            if (s->h_counters[23]) { // primitive-BVH kernel, counting build: the launch's ends on the 100 MHz clock
                const double t0 = (double)((1ull << 62) - s->h_counters[21]), t1 = (double)((1ull << 62) - s->h_counters[22]), t2 = (double)s->h_counters[23];
                std::fprintf(stderr, "[mi355rt] launch timeline (us): first wave starts 0, last wave starts %.1f, first wave finds the batch used up %.1f, last wave ends %.1f; longest ray %llu iterations\n", ((double)s->h_counters[20] - t0) / 100.0, (t1 - t0) / 100.0, (t2 - t0) / 100.0, s->h_counters[7]);
            }
        }
    }
    *out = s->counters;
    return RT_OK;
}

int rt_reset_counters(rt_scene* s) {
    if (!s) return set_err(RT_E_INVALID, "rt_reset_counters: null scene");
    s->counters = rt_counters{};
    s->hit_harvested = s->hit_launches;
    if (!s->host_only) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemset(s->d_counters, 0, C_ALLOC * sizeof(unsigned long long)));
    }
    return RT_OK;
}

double rt_rng_uniform(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t index) {
    return philox_uniform(seed, stream, a, b, index);
}

} // extern "C"
