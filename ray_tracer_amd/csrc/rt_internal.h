// rt_internal.h -- what the translation units of libmi355rt.so share: the path state of the wavefront renderer, the
// work distribution of the persistent kernels, the scene object behind the C ABI's opaque handle, and the launch
// functions of the renderer's kernels.
//
// Three translation units, so that each is compiled with the options it needs and they build in parallel:
//   mi355rt.hip        the C ABI, scene upload, the closest-hit kernels (general and primitive-BVH), the binning pass
//   extend_kernels.hip raygen / extend / accumulate and the small bookkeeping kernels of a render
//   shade_kernels.hip  shade_kernel and tail_kernel.  Only this one is compiled without the backend's
//                      SIOptimizeVGPRLiveRange pass, which miscompiles shade_hit (a value returned through a
//                      reference is lost on one edge of a divergent branch); the pass is worth 40-50 vector
//                      registers to the traversal kernels, which keep it.
#ifndef RT_INTERNAL_H
#define RT_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>
#include <string>
#include <vector>

#include "flatten.h"
#include "traverse.h"
#include "reorder.h"

namespace rt {

constexpr int TAIL_PATHS = 1 << 16; // a batch with this many live paths or fewer finishes in one launch (tail_kernel)
constexpr unsigned long long RANGES_PER_WAVE = 8ull; // work ranges a wave of a persistent kernel gets on average
enum { C_WORK = 0, C_LIVE_IN = 1, C_LIVE_OUT = 2, C_NODES = 4, C_PRIMS = 5, C_RAYS = 6, C_WORDS = 32 };
// Work counters of the general persistent kernels: one per XCD, each on a 128-byte line of its own, behind the
// C_WORDS statistics words.
// (The primitive-BVH kernel is handed the first slot and addresses them as work_counter + (workgroup % 8) * 16.)
constexpr int WORK_SLOTS = 8, WORK_STRIDE = 16, C_WORK_SLOTS = 32, C_STATS = C_WORK_SLOTS + WORK_SLOTS * WORK_STRIDE, C_ALLOC = C_STATS + 96; // (C_STATS: section statistics of the stats build)

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


} // namespace rt

// Measurement switches, read from the environment ONCE, when a scene is created (never per launch).
struct Tuning {
    int refill = -1, vote = -1, chunk = 0; // primitive-BVH kernel scheduling (MI355RT_REFILL, MI355RT_VOTE, MI355RT_CHUNK; -1 / 0: the defaults of bvh_kernel.h)
    int prep_refill = -1, prep_vote = -1;               // the same for reordered batches (-1: PREP_REFILL_THRESHOLD / PREP_LEAF_VOTE of bvh_kernel.h)
    bool quant = true;           // MI355RT_QUANT=0: keep the f32 4-wide nodes
    int postpone = -1;           // MI355RT_POSTPONE=0/1: overrides the choice made from the tree (postponed primitive tests)
    bool exact_fast = true;      // MI355RT_EXACT_FAST=0: RT_HIT_EXACT_CULL keeps the general kernel on the binary nodes
    bool exact_sure = true;  // MI355RT_EXACT_SURE=0: the f64 test for every accepted child on the f32 nodes as well (the results must not change)
    bool exact_f32 = true;   // ... on the f32 nodes, where a child that passes with room to spare needs no f64 test (0: 8-bit nodes, f64 test for every accepted child)
    bool packed = true;          // MI355RT_PACKED=0: keep the 64-byte 8-bit nodes where the 48-byte packed tree exists
    bool fast_extend = true;     // MI355RT_FAST_EXTEND=0: the renderer keeps the general traversal
    bool shade_textures = false; // MI355RT_SHADE_TEXTURES=1: always the full shading kernel
    int reorder = 1;             // MI355RT_REORDER: binning pass in front of large closest-hit batches: 0 never, 1 where it pays (below), 2 always
    bool debug = false;          // MI355RT_DEBUG: diagnostics on stderr
    uint64_t device_bvh_min = 1ull << 15; // MI355RT_DEVICE_BVH_MIN: BVHs over at least this many objects are built on the device (0 = never)
    uint64_t fast_tree_min = 1024; // MI355RT_FAST_TREE_MIN: meshes of at least this many triangles get a traversal-only tree (0 = never)
    int fast_blocks_per_cu = 0;  // MI355RT_FAST_BLOCKS_PER_CU: cap on the primitive-BVH kernel's workgroups per CU (occupancy experiments)
    uint64_t reorder_min = 1ull << 18; // smallest batch that is binned
    uint64_t tail_paths = 1ull << 16;
    uint64_t max_paths = 1ull << 27;
};

constexpr int HIT_RING = 32;

struct rt_scene {
    rt::FlatScene flat;
    rt::DScene dev{};
    bool host_only = false;
    int device = -1;
    std::vector<void*> allocs;
    struct PendingUpload { void* dst; const void* src; size_t bytes; };
    std::vector<PendingUpload> pending_uploads; // large arrays whose copy to the device is still to be made (rt_scene_create only)
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
    int exact_grid_blocks = 0, exact_grid_blocks_prep = 0; // the same for the exact-cull variants of that kernel (more registers, fewer resident workgroups)
    int extend_blocks = 0, shade_blocks = 0, tail_blocks = 0;
    size_t fast_lds = 0;
    uint2* fast_spill = nullptr; // HBM overflow of the per-lane stacks: [entry][thread]
    int32_t* hit_ctx = nullptr;  // transform context of every hit of the current general closest-hit batch
    uint64_t hit_ctx_cap = 0;
    int fast_prim = 0;           // leaf kind of the primitive BVH when uniform, else 0
    int fast_fmt = 0;            // (NODES_F32) node records the primitive-BVH kernel reads (bvh_kernel.h)
    uint32_t fast_entry = RT_REF_NONE; // its root in that format
    float fast_extent = 0.0f; // largest |coordinate| stored in the 4-wide nodes (error bound of the f32 slab test)
    // render workspace
    rt::PathState ps{};
    uint64_t ws_paths = 0;
    std::vector<void*> ws_allocs;
    // reorder pass in front of the primitive-BVH kernel (reorder.h): prepared records in bin order, the bin of every
    // ray, histogram and bin cursors; grown to the largest batch seen
    rt::PRay* prep = nullptr;
    uint16_t* prep_bins = nullptr;
    uint32_t* prep_hist = nullptr;   // REORDER_BINS words, zero between batches
    uint32_t* prep_cursor = nullptr; // REORDER_BINS words
    uint64_t prep_cap = 0;
    rt::ReorderGrid grid{};
    Tuning tune;
    // Calls on one scene may come from several host threads and on several streams (the reference calls world.hit
    // from every rayon worker, camera.rs:99).  The scene owns one set of counters and scratch buffers, so the library
    // puts the calls in a row itself: `call_mutex` is held while a call enqueues (for the whole call where it waits
    // for its results), and every call's first launch waits for `ev_done`, recorded behind the previous call's last
    // launch, when that call went to another stream.
    std::mutex call_mutex;
    hipEvent_t ev_done = nullptr;
    bool have_done = false;
    hipStream_t done_stream = nullptr;
};

namespace rt {

// Launch functions of the renderer's kernels (extend_kernels.hip, shade_kernels.hip).  media: 0 / 1 / 2 levels of
// constant media in the scene; uv: traversal keeps the record history for media whose phase function reads u,v.
void launch_init_batch(hipStream_t st, unsigned long long* counters, unsigned long long npaths);
void launch_advance(hipStream_t st, unsigned long long* counters);
void launch_reset_work(hipStream_t st, unsigned long long* counters);
void launch_raygen(hipStream_t st, const CamD& cam, const BatchGeom& bg, const PathState& ps, unsigned long long seed);
void launch_extend(hipStream_t st, int blocks, int media, bool uv, const DScene& S, const PathState& ps, int qsel, unsigned long long seed,
                   const BatchGeom& bg, int32_t image_width, int rest_only);
void launch_accumulate(hipStream_t st, const PathState& ps, const BatchGeom& bg, double* accum, int32_t image_width);
void launch_shade(hipStream_t st, int blocks, bool textures, bool rr, const DScene& S, const CamD& cam, const PathState& ps, int qsel,
                  unsigned long long seed, const BatchGeom& bg, int last_bounce);
void launch_tail(hipStream_t st, int blocks, int media, bool textures, bool rr, bool uv, const DScene& S, const CamD& cam, const PathState& ps,
                 int qsel, unsigned long long seed, const BatchGeom& bg, int first_bounce, int max_depth);
// Resident workgroups per CU of the variants a scene uses (grids are sized for the variants without the roulette rule).
hipError_t occupancy_extend(int media, int* per_cu);
hipError_t occupancy_shade(bool textures, int* per_cu);
hipError_t occupancy_tail(int media, int* per_cu);

} // namespace rt
#endif
