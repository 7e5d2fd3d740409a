// bvh_kernel.h -- closest hit for a world that is ONE BVH whose leaves are all primitives
// (the reference's Mesh, mesh.rs:95-183, or a BvhNode over spheres/quads/triangles).
//
// Same results as the general traversal in ordered mode; built for throughput:
//   - 4-wide nodes: two levels of the binary tree in one record, boxes tested with a conservative f32
//     slab test; children are entered nearest first.  Half as many dependent fetches per ray as the
//     binary tree.  Two record formats: 128 bytes with f32 boxes (QNode), or 64 bytes with 8-bit
//     boxes on a per-node grid (CQNode, QUANT = true).  The kernel's time follows the number of
//     128-byte lines a ray pulls through L2 from HBM (measured with FETCH_SIZE over four layouts),
//     so the small record, sibling nodes sharing lines and one line per triangle are what made it
//     faster; the 64-byte record also needs four 16-byte requests per visit instead of seven, which
//     matters to the vector L1 (about two cycles per divergent request, profiles/gather_probe.hip);
//   - stack entries carry the entry distance of the deferred child, so a subtree that the closest
//     hit has overtaken since it was pushed is dropped at pop time without fetching its node;
//     the newest 8 entries per lane sit in LDS, deeper ones spill to HBM (rare);
//   - persistent waves with per-lane refill: a wave owns a private range of the batch (one global
//     atomic per range) and a lane whose ray is finished takes the next ray of that range as soon
//     as enough lanes of its wave are idle, instead of waiting for the slowest ray of a 64-ray chunk;
//   - only (t, primitive) are produced here; the hit record is completed by finalize_hits_kernel,
//     one coalesced pass over the batch.
#ifndef RT_BVH_KERNEL_H
#define RT_BVH_KERNEL_H

#include <type_traits>
#include "traverse.h"
#include "reorder.h"

namespace rt {

constexpr int FAST_BLOCK = 256;
constexpr int NODES_F32 = 0, NODES_8BIT = 1, NODES_PACKED = 2; // QNode / CQNode / PNode + PTri (device_types.h)
constexpr int REFILL_THRESHOLD = 40; // refill when fewer lanes than this are still traversing
// The same two for batches of the closest-hit entry points that are traced as they come (no reorder pass): a wave that
// reads 64 consecutive rays of a structure-of-arrays batch refills cheaply, so it tops up earlier, and the intersector
// runs at a smaller quorum.  Measured over the scene matrix: closed meshes and the sphere fields gain 4-9 %, the
// 486-sphere scene loses 6 % on incoherent rays; the renderer's launches (rays gathered out of path records) lose 2-4 %
// and keep the values above.
constexpr int HIT_REFILL_THRESHOLD = 48;
constexpr int HIT_LEAF_VOTE = 16;
constexpr int POP_TRIPS = 2;         // stack entries a lane may take (and discard) per iteration
constexpr int LEAF_VOTE = 24;        // lanes that must stand on a primitive before the wave runs the intersector
// Reordered batches (reorder.h): a refill is eight loads from one line, so the wave tops itself up much earlier; and
// with the fabric no longer the bound, running the intersector with fewer lanes costs less than holding lanes back.
constexpr int PREP_REFILL_THRESHOLD = 56;
constexpr int PREP_LEAF_VOTE = 16;
constexpr double POSTPONE_MIN_OVERLAP = 16.0; // FlatScene::fast_leaf_overlap above which primitive tests are postponed

struct RayQueue {              // optional indirection: ray i of the launch is path queue[i]
    const uint32_t* queue;     // nullptr: identity
    const unsigned long long* count; // nullptr: use n
};

constexpr int LDS_STACK = 8;   // stack entries per lane held in LDS (power of two); deeper ones spill to HBM
constexpr int FAST_WAVES = FAST_BLOCK / 64;

// Per-lane traversal stack: the newest LDS_STACK entries live in LDS ([slot][thread], 8 bytes each),
// older ones in a per-thread column of `spill` (HBM, [entry][thread], coalesced across the wave).
// The LDS array is indexed directly (never through a pointer kept in a struct) so every access
// is compiled as a ds_ instruction rather than a flat one.
struct LaneStack {
    uint2* spill;              // the launch's spill area: this lane's column starts at spill[column()], entry k at + k * spill_stride
    size_t spill_stride;
    int sp;                    // entries on the stack
    int base;                  // oldest entry still in LDS
};
// The column index is rebuilt where a spill happens (rarely) instead of living in two registers for the whole kernel;
// the empty asm keeps the compiler from hoisting the address out of the branch again.
#define RT_STACK_COLUMN(out) do { uint32_t t_ = threadIdx.x; asm volatile("" : "+v"(t_)); (out) = (size_t)blockIdx.x * FAST_BLOCK + t_; } while (0)
#define RT_STACK_SLOT(k) s_fast_stack[((k) & (LDS_STACK - 1)) * FAST_BLOCK + threadIdx.x]
#define RT_STACK_RESET(st) do { (st).sp = 0; (st).base = 0; } while (0)
// Room for k more entries (k <= 3): the oldest LDS entries move to the lane's HBM column.  Checked once per node
// step, so that the pushes themselves are plain LDS writes.
#define RT_STACK_RESERVE(st, k)                                                               \
    do {                                                                                      \
        while ((st).sp - (st).base + (k) > LDS_STACK) {                                       \
            size_t col_;                                                                      \
            RT_STACK_COLUMN(col_);                                                            \
            (st).spill[col_ + (size_t)(st).base * (st).spill_stride] = RT_STACK_SLOT((st).base); \
            ++(st).base;                                                                      \
        }                                                                                     \
    } while (0)
#define RT_STACK_PUSH(st, ref, near_t)                                                        \
    do {                                                                                      \
        uint2 e_;                                                                             \
        e_.x = (ref); e_.y = __float_as_uint(near_t);                                         \
        RT_STACK_SLOT((st).sp) = e_;                                                          \
        ++(st).sp;                                                                            \
    } while (0)
// The LDS slot is read unconditionally and overridden by the spilled entry when there is one.  The
// empty asm pins the LDS value in registers before the branch: without it hipcc (ROCm 7.2) merges the
// two loads into ONE flat load through a selected generic pointer, and a flat load of an LDS address
// takes the long way through the vector memory path (measured: 39 % of the kernel's wave time sat in
// the pop loop waiting for it).
#define RT_STACK_POP(st, out)                                                                 \
    do {                                                                                      \
        --(st).sp;                                                                            \
        (out) = RT_STACK_SLOT((st).sp);                                                       \
        asm volatile("" : "+v"((out).x), "+v"((out).y));                                      \
        if ((st).sp < (st).base) {                                                            \
            size_t col_;                                                                      \
            RT_STACK_COLUMN(col_);                                                            \
            (st).base = (st).sp;                                                              \
            (out) = (st).spill[col_ + (size_t)(st).sp * (st).spill_stride];                   \
        }                                                                                     \
    } while (0)


// Node fetch: each lane reads its own node.  A cooperative variant (the eight lanes of a group read
// the eight 16-byte pieces of one node in a single instruction, nodes handed to their owners through
// LDS) reaches three times the gather rate in isolation (profiles/gather_probe.hip) but was slower
// here: it issues the loads of all 64 lanes whether or not they stand on a node, and its 8 KB of
// staging per wave costs occupancy.

// Ballot of a boolean: the lane mask the compiler already keeps for it.  (HIP's __ballot takes an int, which costs a
// select and a compare per call to turn the mask into a register and back; the loop below votes five times per step.)
#define RT_BALLOT(pred) ((unsigned long long)__builtin_amdgcn_ballot_w64(pred))

RT_DEV float ubyte_f32(uint32_t w, int byte) { return (float)((w >> (8 * byte)) & 0xFFu); } // v_cvt_f32_ubyteN

// PRIM: RT_REF_TRI / RT_REF_SPHERE / RT_REF_QUAD when every leaf has that kind, 0 for mixed leaves.
// RS = element stride of the batch arrays: 1 for the structure-of-arrays batches of the C ABI.  The
// renderer hands in strided views of its 128-byte path records instead of copying rays out of them
// (RS = doubles per record; out_t shares that stride, out_prim counts 32-bit words: 2 * RS).
// PREP: the batch is the array `prep` of prepared records in bin order (reorder.h) instead of `rays`; a record
// carries the ray's index in the caller's batch, where the result goes.
// Launch bound: the packed variant on prepared rays is held to 80 vector registers (it needs 81 unbounded), which
// lets a sixth wave per SIMD in without spilling; the other variants would spill and stay as they compile.
// EXACT: RT_HIT_EXACT_CULL.  The tree is the collapse of the reference's own, and a child whose conservative
// box the ray enters is only taken when Aabb::hit (aabb.rs:82-107, f64, `max <= min` culls) also passes on the box
// the reference would test last on its way to that child, with the interval [tmin, closest hit so far]: a box the
// ray merely touches is culled as the reference culls it.
template <int PRIM, bool COUNT, int FMT, int RS, bool PREP, bool EXACT = false>
__global__ __launch_bounds__(FAST_BLOCK, (FMT == NODES_PACKED && PREP && !COUNT) ? 6 : 1) void bvh4_closest_hit_kernel(DScene S, uint32_t root, rt_ray_batch rays,
                                                                       unsigned long long n_in, RayQueue rq, uint32_t opt, double tmin_default,
                                                                       float scene_extent, double* out_t, uint32_t* out_prim,
                                                                       unsigned long long* work_counter,
                                                                       unsigned long long* stat_counters, uint2* spill, int refill_threshold,
                                                                       int leaf_vote, int chunk_in, int postpone_leaves, const PRay* prep) {
    constexpr bool QUANT = FMT != NODES_F32, PACKED = FMT == NODES_PACKED;
    static_assert(!PACKED || PRIM == RT_REF_TRI, "the packed tree holds triangles only");
    static_assert(!EXACT || FMT != NODES_PACKED, "the exact-cull tree is walked on its 64-byte or f32 nodes");
    constexpr unsigned long long OUT_T_STRIDE = RS, OUT_PRIM_STRIDE = RS == 1 ? 1 : 2 * RS;
    __shared__ uint2 s_fast_stack[LDS_STACK * FAST_BLOCK];
    LaneStack st;
    st.spill_stride = (size_t)gridDim.x * FAST_BLOCK;
    st.spill = spill;
    RT_STACK_RESET(st);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long n = rq.count ? *rq.count : n_in;
    // rays a wave takes from the global counter at a time: large enough to make the counter cheap, small
    // enough that every launched wave gets several chunks (chunk_in > 0 overrides, for measurements)
    int chunk = chunk_in;
    if (chunk <= 0) {
        const unsigned long long per = n / ((unsigned long long)gridDim.x * FAST_WAVES * 8ull);
        chunk = per < 16ull ? 16 : (per > 128ull ? 128 : (int)per);
    }
    const char* const qbase = PACKED ? reinterpret_cast<const char*>(S.pnodes) : (QUANT ? reinterpret_cast<const char*>(S.cqnodes) : reinterpret_cast<const char*>(S.qnodes));

    bool active = false;
    bool pop = false;  // the lane has to take its next node or primitive from the stack
    // Wave w starts on chunk w of the batch without asking anybody (at launch every wave would hit the counter
    // at the same moment; a wave of an over-sized grid whose first chunk lies beyond the batch leaves at once).
    // Later chunks: the waves of XCD x (workgroups are dealt round-robin over the 8 XCDs) take the chunks
    // waves + 8k + x, k from a counter of their own, 128 bytes from the next one (fewer counters when the grid
    // has fewer than 8 workgroups).
    const unsigned long long waves_total = (unsigned long long)gridDim.x * FAST_WAVES;
    const uint32_t work_slots = gridDim.x < 8u ? gridDim.x : 8u; // a small grid uses fewer counters: every residue class needs a taker
    const uint32_t work_slot = blockIdx.x % work_slots;
    unsigned long long* const my_counter = work_counter + work_slot * 16u;
    // (the wave's index goes through readfirstlane so that the compiler keeps these in scalar registers)
    const unsigned long long wave_index = (unsigned long long)blockIdx.x * FAST_WAVES + (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long wave_next = wave_index * (unsigned long long)chunk;                                                     // wave-uniform:
    unsigned long long wave_end = wave_next + (unsigned long long)chunk;                                                        // the part of the batch this wave still owns
    if (wave_end > n) wave_end = n;
    bool more = wave_next < n; // wave-uniform: the batch may still hold rays for this wave
    if (!more) wave_next = wave_end = n;
    RayD r;
    Ray32 q;
    float tmin32 = 0.0f, best32 = 0.0f;
    // EXACT on f32 nodes: a child is taken without the f64 test when the f32 slabs pass with room to spare.  The f32
    // parameter of a plane lies within 1.75 E of the exact one (E as in Ray32: the deliberate shift, a quarter of it for the
    // arithmetic, a quarter for the node's outward-rounded f32 planes, a quarter for rounding c -+ E), so with
    // sure_m = 1.5 * max(cf - cn) >= 2.25 E an interval that is still open after both ends moved inwards by sure_m, and
    // against tmin rounded up and the closest hit rounded down, is open for Aabb::hit as well.  Infinite 1/d (an
    // axis-parallel ray: the f32 form drops that axis, Aabb::hit does not) makes sure_m infinite: always the f64 test.
    float sure_m = 0.0f, tmin32_up = 0.0f, best32_dn = 0.0f;
    constexpr bool SURE = EXACT && FMT == NODES_F32;
#define RT_EXACT_SETUP() do { if (SURE) { \
        const float w_ = fmaxf(fmaxf(q.cf[0] - q.cn[0], q.cf[1] - q.cn[1]), q.cf[2] - q.cn[2]); \
        const bool fin_ = __builtin_isfinite(q.inv[0]) && __builtin_isfinite(q.inv[1]) && __builtin_isfinite(q.inv[2]) && \
                          __builtin_isfinite(q.cf[0] - q.cn[0]) && __builtin_isfinite(q.cf[1] - q.cn[1]) && __builtin_isfinite(q.cf[2] - q.cn[2]); \
        sure_m = (fin_ && !(opt & 8u)) ? 1.5f * w_ : __builtin_inff(); \
        tmin32_up = f32_round_up(tmin); best32_dn = f32_round_down(best_t); } } while (0)
    // The packed tree is only used with the default lower end of the interval (the host sees to it: its sort keys need
    // tmin >= 0), the same for every ray: it stays in scalar registers.
    const float tmin32_default = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(f32_round_down(tmin_default))));
    double time = 0.0, tmin = 0.0, best_t = 0.0;
    uint32_t best_ref = RT_REF_NONE, cur = 0, pend = RT_REF_NONE;
    // (where the result goes: an index of the caller's batch; 32 bits unless the batch is addressed directly, RS == 1 without
    // a prepared copy, and may hold more than 2^32 rays)
    typedef typename std::conditional<PREP || RS != 1, uint32_t, unsigned long long>::type slot_t;
    slot_t ray_slot = 0;
    // a 32-bit slot waits in LDS from the refill to the end of its ray: written once, read once
    constexpr bool SLOT_IN_LDS = sizeof(slot_t) == 4;
    __shared__ uint32_t s_slot[SLOT_IN_LDS ? FAST_BLOCK : 1];
    uint32_t nn = 0, np = 0;
    // wave-level occupancy diagnostics (COUNT builds only): how many lanes take part in each phase
    unsigned long long c_refill = 0, c_node = 0, c_leaf = 0, c_pop = 0, c_mark = 0; // wave cycles per phase (s_memtime)
#define RT_TICK(acc) do { if (COUNT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); (acc) += now_ - c_mark; c_mark = now_; } } while (0)
    if (COUNT) c_mark = __builtin_amdgcn_s_memtime();
    // (COUNT builds: when the launch's first wave started, when the first wave found the batch used up, when the last
    // wave ended, on the 100 MHz clock all CUs share; stored as maxima of a zero-initialised word)
    constexpr unsigned long long T_BIG = 1ull << 62;
    bool d_end_noted = false;
    uint32_t d_ray_iters = 0, d_ray_iters_max = 0;
    const unsigned long long d_t_start = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
    const unsigned long long d_c_start = COUNT ? __builtin_amdgcn_s_memtime() : 0ull; // shader-clock cycles, against the 100 MHz clock above
    if (COUNT && lane == 0) atomicMax(&stat_counters[17], T_BIG - __builtin_amdgcn_s_memrealtime());
    uint32_t d_iters = 0, d_node_lanes = 0, d_tri_phases = 0, d_tri_lanes = 0, d_pop_iters = 0, d_pop_lanes = 0, d_rounds = 0, d_run_lanes = 0;

    for (;;) {
        if (COUNT) ++d_rounds;
        if (more) {
            const unsigned long long idle = RT_BALLOT(!active);
            if (idle) {
                // The wave owns a private range [wave_next, wave_end) of the batch and takes a new one of
                // `chunk` rays only when it is used up: one atomic per chunk instead of one per refill
                // (MI355X serves same-address atomics at roughly 10 ns each).
                if (wave_next >= wave_end) {
                    const int leader = __ffsll((long long)idle) - 1;
                    unsigned long long base = 0;
                    if ((int)lane == leader) base = atomicAdd(my_counter, 1ull);
                    const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)base, leader); // into scalar registers: the range is the wave's
                    const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(base >> 32), leader);
                    wave_next = (waves_total + work_slot + (unsigned long long)work_slots * (((unsigned long long)bhi << 32) | blo)) * (unsigned long long)chunk;
                    wave_end = wave_next + (unsigned long long)chunk;
                    if (wave_end > n) wave_end = n;
                    if (wave_next >= n) { wave_next = wave_end = n; more = false; }
                    if (COUNT && !more && !d_end_noted) { d_end_noted = true; if ((int)lane == leader) atomicMax(&stat_counters[18], T_BIG - __builtin_amdgcn_s_memrealtime()); }
                }
                if (!active) {
                    const unsigned long long qi = wave_next + (unsigned long long)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u)); // idle lanes below this one
                    if (PREP && PACKED) {
                        if (qi < wave_end) {
                            const PRayC p = reinterpret_cast<const PRayC*>(prep)[qi]; // half a line, four 16-byte loads
                            r.ox = p.o[0]; r.oy = p.o[1]; r.oz = p.o[2];
                            r.dx = p.d[0]; r.dy = p.d[1]; r.dz = p.d[2];
                            ray32_from_inv(r, p.inv, scene_extent, q);
                            best_t = (opt & 4u) ? rays.tmax[p.idx] : D_INF;
                            best32 = f32_round_up(best_t);
                            best_ref = RT_REF_NONE;
                            pend = RT_REF_NONE;
                            if (SLOT_IN_LDS) s_slot[threadIdx.x] = p.idx; else ray_slot = (slot_t)p.idx;
                            cur = root;
                            RT_STACK_RESET(st);
                            active = true;
                            pop = false;
                        }
                    } else if (PREP) {
                        if (qi < wave_end) {
                            const PRay p = prep[qi]; // one line, eight 16-byte loads
                            r.ox = p.o[0]; r.oy = p.o[1]; r.oz = p.o[2];
                            r.dx = p.d[0]; r.dy = p.d[1]; r.dz = p.d[2];
#pragma unroll
                            for (int k = 0; k < 3; ++k) { q.inv[k] = p.inv[k]; q.cn[k] = p.cn[k]; q.cf[k] = p.cf[k]; q.neg[k] = (p.neg >> k) & 1u; }
                            if (EXACT) set_inverse(r); // Aabb::hit's 1 / d (aabb.rs:88)
                            time = p.time; tmin = p.tmin; best_t = p.tmax;
                            tmin32 = p.tmin32; best32 = p.tmax32;
                            RT_EXACT_SETUP();
                            best_ref = RT_REF_NONE;
                            pend = RT_REF_NONE;
                            if (SLOT_IN_LDS) s_slot[threadIdx.x] = p.idx; else ray_slot = (slot_t)p.idx;
                            cur = root;
                            RT_STACK_RESET(st);
                            active = true;
                            pop = false;
                        }
                    } else if (qi < wave_end) {
                        const unsigned long long i = rq.queue ? (unsigned long long)rq.queue[qi] : qi;
                        const unsigned long long ri = i * (unsigned long long)RS;
                        r.ox = rays.ox[ri]; r.oy = rays.oy[ri]; r.oz = rays.oz[ri];
                        r.dx = rays.dx[ri]; r.dy = rays.dy[ri]; r.dz = rays.dz[ri];
                        set_inverse(r);
                        make_ray32(r, scene_extent, q);
                        // which optional arrays exist is decided on the host (opt bits 0/1/2 = time/tmin/tmax)
                        time = (opt & 1u) ? rays.time[ri] : 0.0;
                        tmin = (opt & 2u) ? rays.tmin[ri] : tmin_default;
                        best_t = (opt & 4u) ? rays.tmax[ri] : D_INF;
                        tmin32 = f32_round_down(tmin);
                        best32 = f32_round_up(best_t);
                        RT_EXACT_SETUP();
                        best_ref = RT_REF_NONE;
                        pend = RT_REF_NONE;
                        if (SLOT_IN_LDS) s_slot[threadIdx.x] = (uint32_t)i; else ray_slot = (slot_t)i;
                        cur = root;
                        RT_STACK_RESET(st);
                        active = true;
                        pop = false;
                    }
                }
                const unsigned long long left = wave_end - wave_next, want = (unsigned long long)__popcll(idle);
                wave_next += want < left ? want : left;
            }
        }
        RT_TICK(c_refill);
        if (!RT_BALLOT(active)) break;

        // Wave-uniform loop: runs until the wave is empty or enough lanes are idle to refill.
        for (;;) {
            // One iteration = (1) lanes standing on a node take a node step, (2) lanes standing on a
            // primitive test it once enough of the wave stands on one (LEAF_VOTE lanes, or every
            // lane still running), (3) lanes whose node missed or whose primitive was tested pop.
            // Holding primitive tests back until a quorum is reached keeps the f64 intersector from
            // running with a handful of lanes while the rest of the wave waits.
            const bool at_node = active && !pop && (cur >> 28) == REF_QNODE;
            if (COUNT && active) ++d_ray_iters;
            if (COUNT) { ++d_iters; d_run_lanes += __popcll(RT_BALLOT(active)); d_node_lanes += __popcll(RT_BALLOT(at_node)); }
            if (at_node) {
                uint4 ch = make_uint4(0u, 0u, 0u, 0u);
                uint32_t n_inner = 0u, n_all = 0u, inner_base = 0u, leaf_base = 0u; // packed nodes: where the children are
                float nx[4], fx[4], ny[4], fy[4], nz[4], fz[4]; // entry / exit plane per axis and child
                float ax, ay, az, bnx, bfx, bny, bfy, bnz, bfz; // t(plane) = plane * a + b
                if (QUANT) {
                    uint4 h0, qa, qb;
                    if (PACKED) {
                        // (a 32-bit byte offset from a base in scalar registers: the packed arrays stay below 4 GB, flatten.cpp)
                        const uint4* row = reinterpret_cast<const uint4*>(qbase + (uint32_t)((cur & 0x0FFFFFFFu) * (uint32_t)sizeof(PNode)));
                        h0 = row[0];                 // origin x y z | exp x, exp y, exp z, child counts
                        qa = row[1];                 // qlo x | qlo y | qlo z | qhi x   (one byte per child)
                        qb = row[2];                 // qhi y | qhi z | first inner child | first triangle - inner children
                        n_inner = (h0.w >> 24) & 15u; n_all = h0.w >> 28;
                        inner_base = qb.z; leaf_base = qb.w;
                    } else {
                        const uint4* row = reinterpret_cast<const uint4*>(qbase + (size_t)(cur & 0x0FFFFFFFu) * sizeof(CQNode));
                        h0 = row[0];                 // origin x y z | exp x, exp y, exp z
                        ch = row[1];
                        qa = row[2];                 // qlo x | qlo y | qlo z | qhi x   (one byte per child)
                        qb = row[3];                 // qhi y | qhi z
                    }
                    // t(plane) = (origin + g * step - o) / d = g * (step * inv) + (origin * inv + c): with
                    // |origin + g * step| <= M the rounding errors of this form stay below the margin E
                    // (at most 7 * 2^-24 * |inv| * (M + |o|)), see Ray32
                    ax = __uint_as_float((h0.w & 0xFFu) << 23) * q.inv[0];
                    ay = __uint_as_float(((h0.w >> 8) & 0xFFu) << 23) * q.inv[1];
                    az = __uint_as_float(((h0.w >> 16) & 0xFFu) << 23) * q.inv[2];
                    const float ox = __uint_as_float(h0.x), oy = __uint_as_float(h0.y), oz = __uint_as_float(h0.z);
                    bnx = __builtin_fmaf(ox, q.inv[0], q.cn[0]); bfx = __builtin_fmaf(ox, q.inv[0], q.cf[0]);
                    bny = __builtin_fmaf(oy, q.inv[1], q.cn[1]); bfy = __builtin_fmaf(oy, q.inv[1], q.cf[1]);
                    bnz = __builtin_fmaf(oz, q.inv[2], q.cn[2]); bfz = __builtin_fmaf(oz, q.inv[2], q.cf[2]);
                    const uint32_t nxw = q.neg[0] ? qa.w : qa.x, fxw = q.neg[0] ? qa.x : qa.w;
                    const uint32_t nyw = q.neg[1] ? qb.x : qa.y, fyw = q.neg[1] ? qa.y : qb.x;
                    const uint32_t nzw = q.neg[2] ? qb.y : qa.z, fzw = q.neg[2] ? qa.z : qb.y;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        nx[j] = ubyte_f32(nxw, j); fx[j] = ubyte_f32(fxw, j);
                        ny[j] = ubyte_f32(nyw, j); fy[j] = ubyte_f32(fyw, j);
                        nz[j] = ubyte_f32(nzw, j); fz[j] = ubyte_f32(fzw, j);
                    }
                } else {
                    const float4* row = reinterpret_cast<const float4*>(qbase + (size_t)(cur & 0x0FFFFFFFu) * sizeof(QNode));
                    const float4 lox = row[0], loy = row[1], loz = row[2];
                    const float4 hix = row[3], hiy = row[4], hiz = row[5];
                    const float4 chf = row[6];
                    ch = make_uint4(__float_as_uint(chf.x), __float_as_uint(chf.y), __float_as_uint(chf.z), __float_as_uint(chf.w));
                    const float lx[4] = {lox.x, lox.y, lox.z, lox.w}, hx[4] = {hix.x, hix.y, hix.z, hix.w};
                    const float ly[4] = {loy.x, loy.y, loy.z, loy.w}, hy[4] = {hiy.x, hiy.y, hiy.z, hiy.w};
                    const float lz[4] = {loz.x, loz.y, loz.z, loz.w}, hz[4] = {hiz.x, hiz.y, hiz.z, hiz.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        nx[j] = q.neg[0] ? hx[j] : lx[j]; fx[j] = q.neg[0] ? lx[j] : hx[j];
                        ny[j] = q.neg[1] ? hy[j] : ly[j]; fy[j] = q.neg[1] ? ly[j] : hy[j];
                        nz[j] = q.neg[2] ? hz[j] : lz[j]; fz[j] = q.neg[2] ? lz[j] : hz[j];
                    }
                    ax = q.inv[0]; ay = q.inv[1]; az = q.inv[2];
                    bnx = q.cn[0]; bfx = q.cf[0]; bny = q.cn[1]; bfy = q.cf[1]; bnz = q.cn[2]; bfz = q.cf[2];
                }
                if (PACKED) {
                    // The child in slot j is inner_base + j (j < n_inner) or leaf_base + j, so only the slot number has
                    // to travel through the sort: it rides in the two lowest bits of the entry distance (cleared first:
                    // the distance is >= 0 on this path, so that rounds it down and it stays a lower bound), and as
                    // unsigned integers such keys order like the distances.  An exchange is a min and a max.
                    uint32_t k[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float tn = fmaxf(fmaxf(__builtin_fmaf(nx[j], ax, bnx), __builtin_fmaf(ny[j], ay, bny)), fmaxf(__builtin_fmaf(nz[j], az, bnz), tmin32_default));
                        const float tf = fminf(fminf(__builtin_fmaf(fx[j], ax, bfx), __builtin_fmaf(fy[j], ay, bfy)), fminf(__builtin_fmaf(fz[j], az, bfz), best32));
                        const bool hit = (tn <= tf) && ((uint32_t)j < n_all);
                        k[j] = hit ? ((__float_as_uint(tn) & ~3u) | (uint32_t)j) : 0xFFFFFFFFu;
                        if (COUNT) nn += (uint32_t)j < n_all ? 1u : 0u;
                    }
#define RT_USORT2(a, b) do { const uint32_t lo_ = (a) < (b) ? (a) : (b), hi_ = (a) < (b) ? (b) : (a); (a) = lo_; (b) = hi_; } while (0)
                    RT_USORT2(k[0], k[1]);
                    RT_USORT2(k[2], k[3]);
                    RT_USORT2(k[0], k[2]);
                    RT_USORT2(k[1], k[3]);
                    RT_USORT2(k[1], k[2]);
#undef RT_USORT2
#define RT_SLOT_REF(key_) ((((key_) & 3u) < n_inner ? inner_base : leaf_base) + ((key_) & 3u))
                    if (k[0] != 0xFFFFFFFFu) {
                        const bool h1 = k[1] != 0xFFFFFFFFu, h2 = k[2] != 0xFFFFFFFFu, h3 = k[3] != 0xFFFFFFFFu;
                        RT_STACK_RESERVE(st, (h1 ? 1 : 0) + (h2 ? 1 : 0) + (h3 ? 1 : 0));
                        // (the slot bits are cleared again before the key becomes the deferred child's entry distance: with
                        // them the stored value could lie up to three units in the last place ABOVE the computed distance,
                        // more than the slab test's margin covers, and the pop-time cull could drop a subtree whose triangle
                        // ties with, or lies a hair in front of, the closest hit so far)
                        if (h3) RT_STACK_PUSH(st, RT_SLOT_REF(k[3]), __uint_as_float(k[3] & ~3u));
                        if (h2) RT_STACK_PUSH(st, RT_SLOT_REF(k[2]), __uint_as_float(k[2] & ~3u));
                        if (h1) RT_STACK_PUSH(st, RT_SLOT_REF(k[1]), __uint_as_float(k[1] & ~3u));
                        cur = RT_SLOT_REF(k[0]);
                    } else {
                        pop = true;
                    }
#undef RT_SLOT_REF
                } else {
                uint32_t rf[4] = {ch.x, ch.y, ch.z, ch.w};
                float key[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float tn_ax = fmaxf(fmaxf(__builtin_fmaf(nx[j], ax, bnx), __builtin_fmaf(ny[j], ay, bny)), __builtin_fmaf(nz[j], az, bnz));
                    const float tf_ax = fminf(fminf(__builtin_fmaf(fx[j], ax, bfx), __builtin_fmaf(fy[j], ay, bfy)), __builtin_fmaf(fz[j], az, bfz));
                    const float tn = fmaxf(tn_ax, tmin32);
                    const float tf = fminf(tf_ax, best32);
                    bool hit = (tn <= tf) && (rf[j] != RT_REF_NONE);
                    const bool sure = SURE && fmaxf(tn_ax + sure_m, tmin32_up) < fminf(tf_ax - sure_m, best32_dn);
                    if (EXACT && hit && !sure) hit = slab_test(S.ebox + ((size_t)((cur & 0x0FFFFFFFu) - S.exact_first) * 4u + (size_t)j) * 6u, r, tmin, best_t);
                    key[j] = hit ? tn : __builtin_inff();
                    if (COUNT) nn += rf[j] != RT_REF_NONE ? 1u : 0u;
                }
                // ascending by entry distance; misses (+inf) sink to the end.  (Deferring the other children without
                // sorting them completely saves ten instructions per step and costs 6 % more node visits.)
                sort2(key[0], rf[0], key[1], rf[1]);
                sort2(key[2], rf[2], key[3], rf[3]);
                sort2(key[0], rf[0], key[2], rf[2]);
                sort2(key[1], rf[1], key[3], rf[3]);
                sort2(key[1], rf[1], key[2], rf[2]);
                if (key[0] < __builtin_inff()) {
                    const bool h1 = key[1] < __builtin_inff(), h2 = key[2] < __builtin_inff(), h3 = key[3] < __builtin_inff();
                    RT_STACK_RESERVE(st, (h1 ? 1 : 0) + (h2 ? 1 : 0) + (h3 ? 1 : 0));
                    if (h3) RT_STACK_PUSH(st, rf[3], key[3]);
                    if (h2) RT_STACK_PUSH(st, rf[2], key[2]);
                    if (h1) RT_STACK_PUSH(st, rf[1], key[1]);
                    cur = rf[0];
                } else {
                    pop = true;
                }
                }
            }
            RT_TICK(c_node);
            {
                // A lane that reaches a primitive notes it as pending.  The f64 intersector runs for all pending
                // primitives of the wave once leaf_vote of them have come together, or when no lane can go on without
                // it.  Until then the lane waits -- or, with postpone_leaves, goes on with its stack and only waits
                // when it reaches a second primitive or runs out of stack.  Going on pays where most primitives
                // reached are missed (a soup of overlapping triangles: -3 %) and costs where the first one reached is
                // usually the hit that would have cut the rest of the stack (a closed surface: +3 %); the host decides
                // from the tree (FlatScene::fast_leaf_overlap).
                bool blocked = false;
                if (active && !pop && (cur >> 28) != REF_QNODE) {
                    if (pend != RT_REF_NONE) blocked = true;
                    else { pend = cur; if (postpone_leaves) pop = true; else blocked = true; }
                }
                const bool has_pend = active && pend != RT_REF_NONE;
                const unsigned long long pend_mask = RT_BALLOT(has_pend);
                const bool can_go = active && !blocked && (!pop || st.sp > 0);
                const bool quorum = __popcll(pend_mask) >= leaf_vote || RT_BALLOT(can_go) == 0ull;
                if (COUNT && quorum && pend_mask) { ++d_tri_phases; d_tri_lanes += __popcll(pend_mask); }
                if (has_pend && quorum) {
                    const uint32_t kind = pend >> 28, idx = pend & 0x0FFFFFFFu;
                    if (PACKED) {
                        const PTri* tp = reinterpret_cast<const PTri*>(reinterpret_cast<const char*>(S.ptris) + (uint32_t)(idx * (uint32_t)sizeof(PTri)));
                        if (COUNT) ++np;
                        const uint32_t orig = tp->orig; // rides in the request that brings v2.z
                        double t, u, v;
                        if (tri_test(d3p(tp->v0), d3p(tp->v1), d3p(tp->v2), r, tmin_default, best_t, t, u, v)) { best_t = t; best_ref = make_ref(RT_REF_TRI, orig); best32 = f32_round_up(t); }
                    } else if (PRIM == RT_REF_TRI || (PRIM == 0 && kind == RT_REF_TRI)) {
                        const DTri* tp = S.tris + idx;
                        if (COUNT) ++np;
                        double t, u, v;
                        if (tri_test(d3p(tp->v0), d3p(tp->v1), d3p(tp->v2), r, tmin, best_t, t, u, v)) { best_t = t; best_ref = pend; best32 = f32_round_up(t); if (SURE) best32_dn = f32_round_down(t); }
                    } else if (PRIM == RT_REF_SPHERE || (PRIM == 0 && kind == RT_REF_SPHERE)) {
                        const DSphere* sph = S.spheres + idx;
                        if (COUNT) ++np;
                        double t;
                        if (sphere_test(*sph, r, time, tmin, best_t, t)) { best_t = t; best_ref = pend; best32 = f32_round_up(t); if (SURE) best32_dn = f32_round_down(t); }
                    } else { // RT_REF_QUAD
                        const DQuad* qd = S.quads + idx;
                        if (COUNT) ++np;
                        double t, al, be;
                        if (quad_test(*qd, r, tmin, best_t, t, al, be)) { best_t = t; best_ref = pend; best32 = f32_round_up(t); if (SURE) best32_dn = f32_round_down(t); }
                    }
                    pend = RT_REF_NONE;
                    if (blocked) { if (postpone_leaves) pend = cur; pop = true; }
                }
            }
            RT_TICK(c_leaf);
            if (COUNT) { const unsigned long long pm = RT_BALLOT(pop); if (pm) { ++d_pop_iters; d_pop_lanes += __popcll(pm); } }
            if (pop) {
                // At most pop_trips entries are taken per iteration.  After a hit most of a lane's stack is
                // stale (farther than the hit); discarding it all at once would keep the whole wave in this
                // loop with one or two lanes working, so the rest is left for the following iterations.
#pragma unroll
                for (int trip = 0; trip < POP_TRIPS; ++trip) {
                    if (st.sp == 0) {
                        if (pend != RT_REF_NONE) break; // the ray's last primitive is still to be tested
                        slot_t slot = SLOT_IN_LDS ? (slot_t)s_slot[threadIdx.x] : ray_slot;
                        asm volatile("" : "+v"(slot)); // the two addresses are formed here, once per ray, not kept in registers from the refill on
                        out_t[(unsigned long long)slot * OUT_T_STRIDE] = best_ref == RT_REF_NONE ? D_INF : best_t;
                        out_prim[(unsigned long long)slot * OUT_PRIM_STRIDE] = best_ref;
                        if (COUNT) { d_ray_iters_max = d_ray_iters > d_ray_iters_max ? d_ray_iters : d_ray_iters_max; d_ray_iters = 0; }
                        active = false;
                        pop = false;
                        break;
                    }
                    uint2 e;
                    RT_STACK_POP(st, e);
                    // e.y is a lower bound of the distance at which the deferred subtree is entered;
                    // closed-interval primitives may still tie at exactly best_t, so only strictly
                    // farther subtrees are dropped
                    if (!(__uint_as_float(e.y) > best32)) { cur = e.x; pop = false; break; }
                }
            }
            RT_TICK(c_pop);
            const int running = __popcll(RT_BALLOT(active));
            if (running == 0 || (more && running < refill_threshold)) break;
        }
    }
    const unsigned long long d_t_end = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull; // (before the statistics below, whose atomics queue up)
    if (COUNT) {
        atomicMax(&stat_counters[3], (unsigned long long)d_ray_iters_max);
        atomicAdd(&stat_counters[0], (unsigned long long)nn);
        atomicAdd(&stat_counters[1], (unsigned long long)np);
        if (lane == 0) { // per-wave values: the ballots above are wave-uniform
            atomicAdd(&stat_counters[4], (unsigned long long)d_iters);
            atomicAdd(&stat_counters[5], (unsigned long long)d_node_lanes);
            atomicAdd(&stat_counters[6], (unsigned long long)d_tri_phases);
            atomicAdd(&stat_counters[7], (unsigned long long)d_tri_lanes);
            atomicAdd(&stat_counters[8], (unsigned long long)d_pop_iters);
            atomicAdd(&stat_counters[9], (unsigned long long)d_pop_lanes);
            atomicAdd(&stat_counters[10], (unsigned long long)d_rounds);
            atomicAdd(&stat_counters[11], (unsigned long long)d_run_lanes);
            atomicAdd(&stat_counters[12], c_refill);
            atomicAdd(&stat_counters[13], c_node);
            atomicAdd(&stat_counters[14], c_leaf);
            atomicAdd(&stat_counters[15], c_pop);
            atomicMax(&stat_counters[19], d_t_end);
            atomicMax(&stat_counters[16], d_t_start);
            // the shader clock this wave ran at, in kHz: its cycle counter against the constant 100 MHz counter over its lifetime
            const unsigned long long d_cycles = __builtin_amdgcn_s_memtime() - d_c_start, d_ticks = __builtin_amdgcn_s_memrealtime() - d_t_start;
            if (d_ticks > 1000ull) atomicMax(&stat_counters[20], d_cycles * 100000ull / d_ticks);
        }
    }
}

// Completes HitRecord fields for a batch whose (t, prim) are already written; `ctx`: transform context of every
// hit (null: all in the world's own context).  prim is rewritten, because a constant medium is reported by its
// object id, not by the internal reference the traversal leaves.
// keep_medium_uv: the traversal kernel has already written the u,v of constant-medium hits (scenes whose phase functions
// read them: traverse.h UvTrack); otherwise a medium hit reports 0, 0.
__global__ __launch_bounds__(256) void finalize_hits_kernel(DScene S, rt_ray_batch rays, unsigned long long n, rt_hit_batch hits, const int32_t* ctx,
                                                            int keep_medium_uv) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t prim = hits.prim[i];
    if (prim == RT_REF_NONE) {
        if (hits.u) hits.u[i] = 0.0;
        if (hits.v) hits.v[i] = 0.0;
        if (hits.front_face) hits.front_face[i] = 0;
        if (hits.mat) hits.mat[i] = -1;
        if (hits.nx) hits.nx[i] = 0.0;
        if (hits.ny) hits.ny[i] = 0.0;
        if (hits.nz) hits.nz[i] = 0.0;
        if (hits.px) hits.px[i] = 0.0;
        if (hits.py) hits.py[i] = 0.0;
        if (hits.pz) hits.pz[i] = 0.0;
        return;
    }
    WorldRay w{rays.ox, rays.oy, rays.oz, rays.dx, rays.dy, rays.dz, i};
    Best b;
    b.t = hits.t[i]; b.ref = prim; b.ctx = ctx ? ctx[i] : 0;
    HitRec h;
    finalize(S, w, rays.time ? rays.time[i] : 0.0, b, h);
    hits.prim[i] = h.prim;
    const bool keep_uv = keep_medium_uv && (prim >> 28) == RT_REF_MEDIUM;
    if (hits.u && !keep_uv) hits.u[i] = h.u;
    if (hits.v && !keep_uv) hits.v[i] = h.v;
    if (hits.front_face) hits.front_face[i] = h.front_face ? 1 : 0;
    if (hits.mat) hits.mat[i] = h.mat;
    if (hits.nx) hits.nx[i] = h.normal.x;
    if (hits.ny) hits.ny[i] = h.normal.y;
    if (hits.nz) hits.nz[i] = h.normal.z;
    if (hits.px) hits.px[i] = h.p.x;
    if (hits.py) hits.py[i] = h.p.y;
    if (hits.pz) hits.pz[i] = h.p.z;
}

} // namespace rt
#endif
