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
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <mutex>
#include <string>
#include <vector>

#define RT_REORDER_KERNELS
#include "rt_internal.h"
#include "host_parallel.h"
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

} // namespace

// ============================================================================ host code
namespace {

thread_local std::string g_err;

int set_err(int code, const std::string& m) { g_err = m; return code; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return set_err(RT_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));           \
    } while (0)

// One call on a scene at a time (see rt_scene::call_mutex).  begin() makes the call's stream wait for the previous
// call's work, end() marks the end of this call's work.  `chain` = false for the allocation-free entry point, which
// may run under graph capture (an event of another stream cannot be waited for there) and documents that
// overlapping calls race.
struct SceneCall {
    rt_scene* s;
    hipStream_t st;
    bool chain;
    std::unique_lock<std::mutex> lock;
    SceneCall(rt_scene* scene, hipStream_t stream, bool chain_) : s(scene), st(stream), chain(chain_), lock(scene->call_mutex) {}
    int begin() {
        if (chain && s->have_done && s->done_stream != st) HIP_TRY(hipStreamWaitEvent(st, s->ev_done, 0));
        return RT_OK;
    }
    int end() {
        if (chain) { HIP_TRY(hipEventRecord(s->ev_done, st)); s->have_done = true; s->done_stream = st; }
        return RT_OK;
    }
};

// Device address of `scratch` on every GPU the library has set up a scene on.  Each GPU holds its own copy of the
// array at its own address, so the address is resolved once per device, with that device current.
constexpr int NOALLOC_MAX_DEVICES = 64;
int32_t* g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};
std::mutex g_scratch_mutex;
// Who used a device's array last: the allocation-free entry point refuses a call from another stream while that work
// is still running (the array would be overwritten under it).  The event is created with the address, at scene creation.
struct ScratchUse { std::mutex m; hipEvent_t done = nullptr; hipStream_t stream = nullptr; bool pending = false; };
ScratchUse g_scratch_use[NOALLOC_MAX_DEVICES];

int resolve_scratch(int device) {
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return RT_OK; // such a device simply has no allocation-free entry point
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RT_OK;
    int32_t* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
    HIP_TRY(hipEventCreateWithFlags(&g_scratch_use[device].done, hipEventDisableTiming));
    g_scratch_ptr[device] = ptr;
    return RT_OK;
}

template <class T, class A>
int upload(rt_scene* s, const std::vector<T, A>& v, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 256); // never a null pointer in the kernels
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    s->device_bytes += bytes;
    // the large arrays are copied later, side by side (finish_uploads): a copy from pageable memory runs at a fraction of the link's rate
    if (v.size() * sizeof(T) >= (8u << 20)) s->pending_uploads.push_back({p, v.data(), v.size() * sizeof(T)});
    else if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return RT_OK;
}

int finish_uploads(rt_scene* s) {
    std::vector<hipError_t> results(s->pending_uploads.size(), hipSuccess);
    for_parts((int)s->pending_uploads.size(), (int)s->pending_uploads.size() - 1, [&](int k) {
        const rt_scene::PendingUpload& u = s->pending_uploads[(size_t)k];
        hipError_t e = hipSetDevice(s->device); // (the device is a property of the thread)
        results[(size_t)k] = e != hipSuccess ? e : hipMemcpy(u.dst, u.src, u.bytes, hipMemcpyHostToDevice);
    });
    hipError_t first = hipSuccess;
    for (const hipError_t e : results) if (first == hipSuccess) first = e;
    s->pending_uploads.clear();
    if (first != hipSuccess) return set_err(RT_E_HIP, std::string("upload of the scene arrays: ") + hipGetErrorString(first));
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
template <int FMT, bool PREP> FastKernel fast_kernel_exact_for(int prim, bool count) {
    if (prim == (int)RT_REF_TRI) return count ? bvh4_closest_hit_kernel<RT_REF_TRI, true, FMT, 1, PREP, true> : bvh4_closest_hit_kernel<RT_REF_TRI, false, FMT, 1, PREP, true>;
    if (prim == (int)RT_REF_SPHERE) return count ? bvh4_closest_hit_kernel<RT_REF_SPHERE, true, FMT, 1, PREP, true> : bvh4_closest_hit_kernel<RT_REF_SPHERE, false, FMT, 1, PREP, true>;
    return count ? bvh4_closest_hit_kernel<0, true, FMT, 1, PREP, true> : bvh4_closest_hit_kernel<0, false, FMT, 1, PREP, true>;
}
// RT_HIT_EXACT_CULL: the collapse of the reference's tree (64-byte nodes with 8-bit boxes, or f32 nodes) + exact boxes
FastKernel fast_kernel_exact(int prim, bool count, int fmt, bool prepared) {
    if (fmt == NODES_8BIT) return prepared ? fast_kernel_exact_for<NODES_8BIT, true>(prim, count) : fast_kernel_exact_for<NODES_8BIT, false>(prim, count);
    return prepared ? fast_kernel_exact_for<NODES_F32, true>(prim, count) : fast_kernel_exact_for<NODES_F32, false>(prim, count);
}
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
    t.prep_refill = env_int("MI355RT_PREP_REFILL", PREP_REFILL_THRESHOLD);
    t.prep_vote = env_int("MI355RT_PREP_VOTE", PREP_LEAF_VOTE);
    t.quant = env_int("MI355RT_QUANT", 1) != 0;
    t.exact_fast = env_int("MI355RT_EXACT_FAST", 1) != 0;
    t.exact_f32 = env_int("MI355RT_EXACT_F32", 1) != 0;
    t.exact_sure = env_int("MI355RT_EXACT_SURE", 1) != 0;
    t.fast_extend = env_int("MI355RT_FAST_EXTEND", 1) != 0;
    t.shade_textures = env_int("MI355RT_SHADE_TEXTURES", 0) != 0;
    t.reorder = env_int("MI355RT_REORDER", 1);
    t.debug = getenv("MI355RT_DEBUG") != nullptr;
    t.fast_blocks_per_cu = env_int("MI355RT_FAST_BLOCKS_PER_CU", 0);
    t.device_bvh_min = (uint64_t)std::max(0, env_int("MI355RT_DEVICE_BVH_MIN", (int)t.device_bvh_min));
    t.fast_tree_min = (uint64_t)std::max(0, env_int("MI355RT_FAST_TREE_MIN", (int)t.fast_tree_min));
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
    const int media = s->flat.medium_depth > 1 ? 2 : (s->flat.has_medium ? 1 : 0);
    if (media > 1) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closest_hit_kernel<false, true, 2>, BLOCK, 0));
    else if (media) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closest_hit_kernel<false, true, 1>, BLOCK, 0));
    else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, closest_hit_kernel<false, true, false>, BLOCK, 0));
    int per_cu2 = 0, per_cu3 = 0, per_cu4 = 0;
    // (grids are sized for the variants without the roulette rule; a render with the rule runs on the same grid)
    HIP_TRY(occupancy_extend(media, &per_cu2));
    HIP_TRY(occupancy_shade(s->flat.has_textures, &per_cu3));
    HIP_TRY(occupancy_tail(media, &per_cu4));
    s->grid_blocks = prop.multiProcessorCount * std::max(1, per_cu);      // batched closest hit (general kernel)
    s->extend_blocks = prop.multiProcessorCount * std::max(1, per_cu2);   // each kernel fills the chip at its own occupancy
    s->shade_blocks = prop.multiProcessorCount * std::max(1, per_cu3);
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
        // The exact-cull variants need more registers (98: four workgroups per CU); a grid sized for the others would leave
        // workgroups waiting for a place and start their first ranges when the rest of the batch is done (soup: 11.0 -> 10.7 ms).
        s->exact_grid_blocks = s->exact_grid_blocks_prep = 0;
        if (s->flat.exact_root != RT_REF_NONE) {
            const int efmt = (s->fast_fmt == NODES_F32 || s->tune.exact_f32) ? NODES_F32 : NODES_8BIT;
            int plain_per_cu = 0, ordered_per_cu = 0;
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&plain_per_cu, fast_kernel_exact(s->fast_prim, false, efmt, false), FAST_BLOCK, 0));
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&ordered_per_cu, fast_kernel_exact(s->fast_prim, false, efmt, true), FAST_BLOCK, 0));
            if (s->tune.fast_blocks_per_cu > 0) { plain_per_cu = std::min(plain_per_cu, s->tune.fast_blocks_per_cu); ordered_per_cu = std::min(ordered_per_cu, s->tune.fast_blocks_per_cu); }
            s->exact_grid_blocks = std::min(s->fast_grid_blocks, prop.multiProcessorCount * std::max(1, plain_per_cu));
            s->exact_grid_blocks_prep = std::min(s->fast_grid_blocks_prep, prop.multiProcessorCount * std::max(1, ordered_per_cu));
        }
        {
            size_t entries = (size_t)std::max<uint32_t>(std::max(s->flat.fast_stack, s->flat.exact_stack), 1u);
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
    d.ebox = f.ebox.data(); d.exact_first = f.exact_first;
    d.n_xforms = (uint32_t)f.xforms.size(); d.n_perlins = (uint32_t)f.perlins.size();
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
    const RawVec<CQNode> no_cqnodes;
    const RawVec<PNode> no_pnodes;
    const RawVec<PTri> no_ptris;
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
    if ((rc = upload(s, f.ebox, &d.ebox))) return rc;
    if ((rc = finish_uploads(s))) return rc;
    d.exact_first = f.exact_first;
    d.n_xforms = (uint32_t)f.xforms.size(); d.n_perlins = (uint32_t)f.perlins.size();
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
    HIP_TRY(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
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

int ensure_hit_ctx(rt_scene* s, uint64_t n) { // scratch between the traversal and the record-completion pass, grown to the largest batch seen
    if (s->hit_ctx_cap >= n) return RT_OK;
    if (s->hit_ctx) (void)hipFree(s->hit_ctx);
    s->hit_ctx = nullptr; s->hit_ctx_cap = 0;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(int32_t)));
    s->hit_ctx = static_cast<int32_t*>(p);
    s->hit_ctx_cap = n;
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
    // The allocation-free entry point (ctx_scratch given) records no timing events and leaves the ring alone: on a
    // capturing stream an event record is a graph node, not a timestamp, and rt_get_counters could not read it back.
    const bool timed = ctx_scratch == nullptr;
#define RT_EVENT(e_) do { if (timed) HIP_TRY(hipEventRecord((e_), st)); } while (0)
    RT_EVENT(ev[0]);
    // RT_HIT_ORDERED | RT_HIT_EXACT_CULL: the exact binary nodes, children ordered by the split axis (general kernel)
    const bool exact_cull = (flags & RT_HIT_ORDERED) && (flags & RT_HIT_EXACT_CULL);
    const bool count = flags & RT_HIT_COUNTERS, ordered = (flags & RT_HIT_ORDERED) && !exact_cull;
    const Tuning& tn = s->tune;
    // ... of a world that is one BVH over primitives: the primitive-BVH kernel on the collapse of the reference's tree, every
    // child confirmed on its exact box (MI355RT_EXACT_FAST=0 keeps the general kernel, for A/B measurements)
    const bool exact_fast = exact_cull && s->flat.exact_root != RT_REF_NONE && s->flat.fast_root != RT_REF_NONE && !s->flat.fast_partial && tn.exact_fast;
    if ((ordered || exact_fast) && s->flat.fast_root != RT_REF_NONE && !s->flat.fast_partial) {
        // the world is one BVH over primitives: dedicated traversal kernel + one pass that completes the records
        int fblocks = (int)std::min<uint64_t>((uint64_t)s->fast_grid_blocks, (waves + (FAST_BLOCK / 64) - 1) / (FAST_BLOCK / 64));
        RayQueue rq{nullptr, nullptr};
        const uint32_t opt = (rays->time ? 1u : 0u) | (rays->tmin ? 2u : 0u) | (rays->tmax ? 4u : 0u) | (tn.exact_sure ? 0u : 8u);
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
        if (exact_fast) fblocks = std::min(fblocks, reorder ? s->exact_grid_blocks_prep : s->exact_grid_blocks);
        if (reorder) {
            int rc = ensure_reorder_scratch(s, n);
            if (rc) return rc;
            const int cblocks = (int)std::min<uint64_t>(2048, (n + REORDER_BLOCK - 1) / REORDER_BLOCK);
            // (the scan pass leaves the histogram cleared; clearing it here as well costs 16 KiB and makes a batch
            // independent of how the previous one ended)
            HIP_TRY(hipMemsetAsync(s->prep_hist, 0, REORDER_BINS * sizeof(uint32_t), st));
            hipLaunchKernelGGL(reorder_count_kernel, dim3(cblocks), dim3(REORDER_BLOCK), 0, st, *rays, (unsigned long long)n, s->grid, s->prep_bins, s->prep_hist);
            hipLaunchKernelGGL(reorder_scan_kernel, dim3(1), dim3(1024), 0, st, s->prep_hist, s->prep_cursor);
            // (the packed tree takes 64-byte records; it is not used when the batch brings lower ends of its own, see below)
            hipLaunchKernelGGL((s->fast_fmt == NODES_PACKED && !(opt & 2u) && !exact_fast) ? reorder_scatter_kernel<true> : reorder_scatter_kernel<false>,
                               dim3((unsigned)((n + REORDER_TILE - 1) / REORDER_TILE)), dim3(SCATTER_BLOCK), 0, st, *rays,
                               (unsigned long long)n, opt, 0.001, s->fast_extent, (const uint16_t*)s->prep_bins, s->prep_cursor, s->prep);
            HIP_TRY(hipGetLastError());
        }
        RT_EVENT(ev[1]);
        // the packed nodes order children by keys that need entry distances >= 0, i.e. tmin >= 0: certain only for the default tmin
        const int fmt = exact_fast ? ((s->fast_fmt == NODES_F32 || tn.exact_f32) ? NODES_F32 : NODES_8BIT) : ((s->fast_fmt == NODES_PACKED && (opt & 2u)) ? NODES_8BIT : s->fast_fmt);
        const uint32_t entry = exact_fast ? s->flat.exact_root : (fmt == s->fast_fmt ? s->fast_entry : s->flat.fast_root);
        // lanes go on past a primitive they have reached where a ray crosses many triangle boxes (bvh_kernel.h);
        // measured on reordered batches: the 2^20-triangle soup (overlap 70) gains 3 %, a closed surface of 1.3 M triangles
        // (overlap 3.9) loses 3 %; on batches traced as they come closed surfaces of 82 k and 1.3 M triangles gain 1-3 %
        const int postpone = tn.postpone >= 0 ? tn.postpone : (fmt == NODES_PACKED && (!reorder || s->flat.fast_leaf_overlap > POSTPONE_MIN_OVERLAP) ? 1 : 0);
        const int refill = reorder ? tn.prep_refill : (tn.refill >= 0 ? tn.refill : HIT_REFILL_THRESHOLD), vote = reorder ? tn.prep_vote : (tn.vote >= 0 ? tn.vote : HIT_LEAF_VOTE);
        const FastKernel kernel = exact_fast ? fast_kernel_exact(s->fast_prim, count, fmt, reorder) : (reorder ? fast_kernel_prepared(s->fast_prim, count, fmt) : fast_kernel(s->fast_prim, count, fmt));
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
        RT_EVENT(ev[2]);
        if (hits->u || hits->v || hits->front_face || hits->mat || hits->nx || hits->ny || hits->nz || hits->px || hits->py || hits->pz)
            hipLaunchKernelGGL(finalize_hits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->dev, *rays, (unsigned long long)n, *hits, (const int32_t*)nullptr, 0);
        HIP_TRY(hipGetLastError());
        RT_EVENT(ev[3]); // the whole call: reorder pass, traversal, record completion
        if (timed) { s->hit_reordered_at[s->hit_launches % HIT_RING] = reorder; ++s->hit_launches; }
        s->counters.rays += n;
        return RT_OK;
    }
    if (!ctx_scratch) { int rcc = ensure_hit_ctx(s, n); if (rcc) return rcc; }
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
    RT_EVENT(ev[1]);
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
    RT_EVENT(ev[2]);
    hipLaunchKernelGGL(finalize_hits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s->dev, *rays, (unsigned long long)n, *hits, (const int32_t*)hit_ctx, s->flat.track_uv ? 1 : 0);
    HIP_TRY(hipGetLastError());
    RT_EVENT(ev[3]);
#undef RT_EVENT
    if (timed) { s->hit_reordered_at[s->hit_launches % HIT_RING] = false; ++s->hit_launches; }
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
            launch_init_batch(st, s->d_counters, (unsigned long long)bg.npaths);
            launch_raygen(st, cd, bg, ps, (unsigned long long)seed);
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
                    launch_tail(st, tblocks, media, has_tex, rr, s->flat.track_uv, s->dev, cd, ps, qsel, (unsigned long long)seed, bg, (int)bounce, (int)cam->max_depth);
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
                    if (fast_extend) launch_reset_work(st, s->d_counters);
                    launch_extend(st, blocks, media, s->flat.track_uv, s->dev, ps, qsel, (unsigned long long)seed, bg, cam->image_width, fast_extend ? 1 : 0);
                }
                launch_shade(st, sblocks, has_tex, rr, s->dev, cd, ps, qsel, (unsigned long long)seed, bg, bounce == cam->max_depth - 1 ? 1 : 0);
                launch_advance(st, s->d_counters);
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
            launch_accumulate(st, ps, bg, d_accum, cam->image_width);
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
    const auto t_0 = std::chrono::steady_clock::now();
    int rc = flatten_scene(desc, s->flat, err, (flags & RT_SCENE_HOST_ONLY) ? 0 : s->tune.device_bvh_min, s->tune.fast_tree_min);
    if (rc) { delete s; return set_err(rc, err); }
    const auto t_1 = std::chrono::steady_clock::now();
    if (flags & RT_SCENE_HOST_ONLY) {
        s->host_only = true;
        fill_host_dscene(s);
    } else {
        rc = upload_scene(s);
        if (rc) { rt_scene_destroy(s); return rc; }
    }
    if (s->tune.debug)
        fprintf(stderr, "[mi355rt] rt_scene_create: flatten %.3f s, device arrays %.3f s (%.0f MB)\n", std::chrono::duration<double>(t_1 - t_0).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_1).count(), (double)s->device_bytes / 1e6);
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
        if (s->ev_done) (void)hipEventDestroy(s->ev_done);
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
        if (o.has_box8) {
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

int rt_scene_fast_tree_info(const rt_scene* s, uint64_t* wide_root, uint64_t* references, int32_t* traversal_only) {
    if (!s) return set_err(RT_E_INVALID, "rt_scene_fast_tree_info: null scene");
    const bool have = s->flat.fast_root != RT_REF_NONE;
    if (wide_root) *wide_root = have ? RT_REF_INDEX(s->flat.fast_root) : 0;
    if (references) *references = s->flat.ptris.size();
    if (traversal_only) *traversal_only = s->flat.fast_tree ? 1 : 0;
    return have ? RT_OK : set_err(RT_E_INVALID, "rt_scene_fast_tree_info: the world is not one BVH over primitives");
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

int rt_scene_reserve(rt_scene* s, uint64_t max_rays, uint64_t max_paths) {
    int rc = need_device(s);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(s->call_mutex);
    HIP_TRY(hipDeviceSynchronize()); // buffers in use by calls still in flight are about to be replaced
    if (max_rays) {
        const bool fast_whole = s->flat.fast_root != RT_REF_NONE && !s->flat.fast_partial;
        if (fast_whole && max_rays <= 0xFFFFFFFFull && (rc = ensure_reorder_scratch(s, max_rays))) return rc;
        if ((rc = ensure_hit_ctx(s, max_rays))) return rc;
    }
    if (max_paths) {
        if ((rc = ensure_workspace(s, std::min<uint64_t>(max_paths, s->tune.max_paths)))) return rc;
    }
    return RT_OK;
}

int rt_closest_hit_device(rt_scene* s, const rt_ray_batch* rays, uint64_t n, rt_hit_batch* hits, uint64_t seed, uint32_t flags, void* hip_stream) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!rays || !hits) return set_err(RT_E_INVALID, "rt_closest_hit_device: null batch");
    if (n && (!rays->ox || !rays->oy || !rays->oz || !rays->dx || !rays->dy || !rays->dz || !hits->t || !hits->prim))
        return set_err(RT_E_INVALID, "rt_closest_hit_device: origin, direction, t and prim arrays are required");
    SceneCall call(s, static_cast<hipStream_t>(hip_stream), true);
    if ((rc = call.begin())) return rc;
    if ((rc = launch_closest_hit(s, rays, n, hits, seed, flags, call.st))) return rc;
    return call.end();
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
    SceneCall call(s, static_cast<hipStream_t>(hip_stream), false); // host-side state only: see SceneCall
    ScratchUse& use = g_scratch_use[s->device];
    std::lock_guard<std::mutex> guard(use.m);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    HIP_TRY(hipStreamIsCapturing(call.st, &capturing));
    const bool watched = capturing == hipStreamCaptureStatusNone; // (an event recorded during capture is a graph node, not a time)
    if (watched && use.pending && use.stream != call.st) {
        const hipError_t q = hipEventQuery(use.done);
        if (q == hipErrorNotReady)
            return set_err(RT_E_BUSY, "rt_closest_hit_device_noalloc: the previous allocation-free call on this GPU, on another stream, has not finished; its scratch would be overwritten");
        if (q != hipSuccess) return set_err(RT_E_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        use.pending = false;
    }
    rc = launch_closest_hit(s, rays, n, hits, seed, flags, call.st, ptr);
    if (watched && n) {
        HIP_TRY(hipEventRecord(use.done, call.st));
        use.stream = call.st;
        use.pending = true;
    }
    return rc;
}

int rt_closest_hit(rt_scene* s, const rt_ray_batch* rays, uint64_t n, rt_hit_batch* hits, uint64_t seed, uint32_t flags) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!rays || !hits) return set_err(RT_E_INVALID, "rt_closest_hit: null batch");
    if (n == 0) return RT_OK;
    if (!rays->ox || !rays->oy || !rays->oz || !rays->dx || !rays->dy || !rays->dz || !hits->t || !hits->prim)
        return set_err(RT_E_INVALID, "rt_closest_hit: origin, direction, t and prim arrays are required");
    SceneCall call(s, nullptr, true); // held to the end: the call waits for its results
    if ((rc = call.begin())) return rc;
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
    return call.end();
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
    SceneCall call(s, static_cast<hipStream_t>(hip_stream), true);
    if ((rc = call.begin())) return rc;
    if ((rc = render_impl(s, cam, seed, work, accum, call.st))) return rc;
    return call.end();
}

int rt_render(rt_scene* s, const rt_camera* cam, uint64_t seed, const rt_work* work, double* accum) {
    int rc = need_device(s);
    if (rc) return rc;
    if (!cam || !accum) return set_err(RT_E_INVALID, "rt_render: null argument");
    rt_camera_derived d;
    std::string err;
    if ((rc = camera_initialize(cam, &d, err))) return set_err(rc, err);
    size_t bytes = (size_t)cam->image_width * (size_t)d.image_height * 3 * sizeof(double);
    SceneCall call(s, nullptr, true);
    if ((rc = call.begin())) return rc;
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
    return call.end();
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
    std::lock_guard<std::mutex> lock(s->call_mutex);
    if (!s->host_only) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(s->h_counters, s->d_counters, C_WORDS * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        s->counters.node_visits = s->h_counters[C_NODES];
        s->counters.prim_tests = s->h_counters[C_PRIMS];
        if (s->h_counters[C_NODES + 20]) s->counters.shader_khz = s->h_counters[C_NODES + 20];
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
            {
                unsigned long long sec[72];
                HIP_TRY(hipMemcpy(sec, s->d_counters + C_STATS, sizeof(sec), hipMemcpyDeviceToHost));
                static const char* const names[24] = {"wave-load of paths", "finalize: ray into context", "finalize: record by kind", "finalize: transform unwinding",
                    "material record", "metal", "dielectric", "lambert/iso: colour", "lambert/iso: mixture choice + light pick", "lambert/iso: two draws + sampler",
                    "lambert/iso: pdf values (light tests)", "path record load", "shade_hit", "", "", "", "", "", "", "", "", "", "", ""};
                for (int k = 0; k < 13; ++k)
                    if (sec[24 + k])
                        std::fprintf(stderr, "[mi355rt] shade section %-44s runs %12llu  lanes/run %5.1f  wave cycles %14llu\n", names[k], sec[24 + k],
                                     (double)sec[k] / (double)sec[24 + k], sec[48 + k]);
            }
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
    std::lock_guard<std::mutex> lock(s->call_mutex);
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
