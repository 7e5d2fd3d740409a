// bvh8_kernel.h -- closest hit for a world that is ONE BVH over triangles (the reference's Mesh,
// mesh.rs:95-183), traversed on an 8-wide compressed copy of the tree.
//
// Why another layout: on MI355X a divergent per-lane load costs the vector L1 about two cycles per
// (lane, 16-byte request) whether or not the requests of a lane fall into one line
// (profiles/gather_probe.hip), so what a node visit costs is the number of 16-byte requests it
// issues, not the bytes.  The 4-wide f32 node tests 4 boxes with 7 requests; the 80-byte CNode tests
// 8 boxes with 5, and an 8-wide tree needs about half as many visits per ray.
//
// Results are those of bvh4_closest_hit_kernel: the quantised child boxes contain the f32 boxes,
// the slab test is conservative, and the triangle intersector is the same f64 code, so the closest
// t is bit-identical; only which of several primitives at exactly equal t is reported may differ.
#ifndef RT_BVH8_KERNEL_H
#define RT_BVH8_KERNEL_H

#include "bvh_kernel.h"

namespace rt {

constexpr int LDS_STACK8 = 16; // stack entries per lane in LDS (ring, power of two); an 8-wide node can defer 7 children
#define RT8_SLOT(k) s_stack8[((k) & (LDS_STACK8 - 1)) * FAST_BLOCK + threadIdx.x]
#define RT8_PUSH(st, ref, key)                                                                \
    do {                                                                                      \
        if ((st).sp - (st).base == LDS_STACK8) {                                              \
            (st).spill[(size_t)(st).base * (st).spill_stride] = RT8_SLOT((st).base);          \
            ++(st).base;                                                                      \
        }                                                                                     \
        uint2 e_;                                                                             \
        e_.x = (ref); e_.y = (key);                                                           \
        RT8_SLOT((st).sp) = e_;                                                               \
        ++(st).sp;                                                                            \
    } while (0)
#define RT8_POP(st, out)                                                                      \
    do {                                                                                      \
        --(st).sp;                                                                            \
        (out) = RT8_SLOT((st).sp);                                                            \
        if ((st).sp < (st).base) { (st).base = (st).sp; (out) = (st).spill[(size_t)(st).sp * (st).spill_stride]; } \
    } while (0)

// compare-exchange on packed keys (entry distance bits with the slot number in the low 3 bits)
#define RT8_CE(a, b) do { const uint32_t lo_ = min(k[a], k[b]); k[b] = max(k[a], k[b]); k[a] = lo_; } while (0)

RT_DEV float ubyte_f32(uint32_t w, int byte) { return (float)((w >> (8 * byte)) & 0xFFu); }

template <bool COUNT>
__global__ __launch_bounds__(FAST_BLOCK) void bvh8_closest_hit_kernel(DScene S, rt_ray_batch rays, unsigned long long n_in, RayQueue rq,
                                                                       uint32_t opt, double tmin_default, float scene_extent, double* out_t,
                                                                       uint32_t* out_prim, unsigned long long* work_counter,
                                                                       unsigned long long* stat_counters, uint2* spill) {
    __shared__ uint2 s_stack8[LDS_STACK8 * FAST_BLOCK];
    LaneStack st;
    st.spill_stride = (size_t)gridDim.x * FAST_BLOCK;
    st.spill = spill + (size_t)blockIdx.x * FAST_BLOCK + threadIdx.x;
    RT_STACK_RESET(st);
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lane_below = (1ull << lane) - 1ull;
    const unsigned long long n = rq.count ? *rq.count : n_in;
    const char* const cbase = reinterpret_cast<const char*>(S.cnodes);
    const char* const tbase = reinterpret_cast<const char*>(S.ctris);
    constexpr uint32_t TRI_TAG = RT_REF_TRI << 28;

    bool active = false;
    bool more = true; // wave-uniform: the counter may still hold rays
    RayD r;
    Ray32 q;
    float tmin32 = 0.0f, best32 = 0.0f;
    double tmin = 0.0, best_t = 0.0;
    uint32_t best_ref = RT_REF_NONE, cur = 0;
    unsigned long long ray_slot = 0;
    uint32_t nn = 0, np = 0;
    bool signed_t = false;

    for (;;) {
        if (more) {
            const unsigned long long idle = __ballot(!active);
            if (idle) {
                const int nidle = __popcll(idle);
                const int leader = __ffsll((long long)idle) - 1;
                unsigned long long base = 0;
                if ((int)lane == leader) base = atomicAdd(work_counter, (unsigned long long)nidle);
                const uint32_t blo = (uint32_t)__shfl((int)(uint32_t)base, leader);
                const uint32_t bhi = (uint32_t)__shfl((int)(uint32_t)(base >> 32), leader);
                base = ((unsigned long long)bhi << 32) | blo;
                if (!active) {
                    const unsigned long long qi = base + (unsigned long long)__popcll(idle & lane_below);
                    if (qi < n) {
                        const unsigned long long i = rq.queue ? (unsigned long long)rq.queue[qi] : qi;
                        r.ox = rays.ox[i]; r.oy = rays.oy[i]; r.oz = rays.oz[i];
                        r.dx = rays.dx[i]; r.dy = rays.dy[i]; r.dz = rays.dz[i];
                        make_ray32(r, scene_extent, q);
                        tmin = (opt & 2u) ? rays.tmin[i] : tmin_default;
                        best_t = (opt & 4u) ? rays.tmax[i] : D_INF;
                        tmin32 = f32_round_down(tmin);
                        // entry distances are sorted and kept as integer keys, which needs them non-negative;
                        // a ray with a negative tmin gives that up (children in slot order, nothing dropped at pop)
                        signed_t = tmin < 0.0;
                        best32 = f32_round_up(best_t);
                        best_ref = RT_REF_NONE;
                        ray_slot = i;
                        cur = 0; // cnodes[0] is the root
                        RT_STACK_RESET(st);
                        active = true;
                    }
                }
                if (base + (unsigned long long)nidle >= n) more = false;
            }
        }
        if (!__any(active)) break;

        for (;;) {
            const bool at_node = active && (cur >> 28) == RT_REF_NODE;
            bool pop = false;
            if (at_node) {
                const uint4* row = reinterpret_cast<const uint4*>(cbase + (size_t)cur * sizeof(CNode));
                const uint4 h0 = row[0]; // origin x y z | exp x, exp y, exp z, valid
                const uint4 h1 = row[1]; // child_base | prim_base | meta 0-3 | meta 4-7
                const uint4 q0 = row[2]; // qlo x (slots 0-3, 4-7) | qlo y
                const uint4 q1 = row[3]; // qlo z | qhi x
                const uint4 q2 = row[4]; // qhi y | qhi z
                // t(plane) = (origin + g * step - o) / d = g * (step * inv) + (origin * inv + c)
                const float ax = __uint_as_float((h0.w & 0xFFu) << 23) * q.inv[0];
                const float ay = __uint_as_float(((h0.w >> 8) & 0xFFu) << 23) * q.inv[1];
                const float az = __uint_as_float(((h0.w >> 16) & 0xFFu) << 23) * q.inv[2];
                const float ox = __uint_as_float(h0.x), oy = __uint_as_float(h0.y), oz = __uint_as_float(h0.z);
                const float bnx = __builtin_fmaf(ox, q.inv[0], q.cn[0]), bfx = __builtin_fmaf(ox, q.inv[0], q.cf[0]);
                const float bny = __builtin_fmaf(oy, q.inv[1], q.cn[1]), bfy = __builtin_fmaf(oy, q.inv[1], q.cf[1]);
                const float bnz = __builtin_fmaf(oz, q.inv[2], q.cn[2]), bfz = __builtin_fmaf(oz, q.inv[2], q.cf[2]);
                // entry / exit grid coordinates per axis: the box maximum is the entry plane when d < 0
                const uint32_t nxw[2] = {q.neg[0] ? q1.z : q0.x, q.neg[0] ? q1.w : q0.y};
                const uint32_t fxw[2] = {q.neg[0] ? q0.x : q1.z, q.neg[0] ? q0.y : q1.w};
                const uint32_t nyw[2] = {q.neg[1] ? q2.x : q0.z, q.neg[1] ? q2.y : q0.w};
                const uint32_t fyw[2] = {q.neg[1] ? q0.z : q2.x, q.neg[1] ? q0.w : q2.y};
                const uint32_t nzw[2] = {q.neg[2] ? q2.z : q1.x, q.neg[2] ? q2.w : q1.y};
                const uint32_t fzw[2] = {q.neg[2] ? q1.x : q2.z, q.neg[2] ? q1.y : q2.w};
                const uint32_t valid = h0.w >> 24;
                uint32_t k[8];
                int nh = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int w = j >> 2, b = j & 3;
                    const float tn = fmaxf(fmaxf(__builtin_fmaf(ubyte_f32(nxw[w], b), ax, bnx), __builtin_fmaf(ubyte_f32(nyw[w], b), ay, bny)),
                                           fmaxf(__builtin_fmaf(ubyte_f32(nzw[w], b), az, bnz), tmin32));
                    const float tf = fminf(fminf(__builtin_fmaf(ubyte_f32(fxw[w], b), ax, bfx), __builtin_fmaf(ubyte_f32(fyw[w], b), ay, bfy)),
                                           fminf(__builtin_fmaf(ubyte_f32(fzw[w], b), az, bfz), best32));
                    const bool hit = (tn <= tf) && ((valid >> j) & 1u);
                    // tn >= tmin32 >= 0, so its bit pattern orders like the value; the low 3 bits carry the slot
                    k[j] = hit ? (((signed_t ? 0u : __float_as_uint(tn)) & ~7u) | (uint32_t)j) : (0xFFFFFFF8u | (uint32_t)j);
                    nh += hit ? 1 : 0;
                    if (COUNT) nn += (valid >> j) & 1u;
                }
                if (nh > 0) {
                    // 19-exchange sorting network, ascending: hits first (nearest first), misses last
                    RT8_CE(0, 1); RT8_CE(2, 3); RT8_CE(4, 5); RT8_CE(6, 7);
                    RT8_CE(0, 2); RT8_CE(1, 3); RT8_CE(4, 6); RT8_CE(5, 7);
                    RT8_CE(1, 2); RT8_CE(5, 6); RT8_CE(0, 4); RT8_CE(3, 7);
                    RT8_CE(1, 5); RT8_CE(2, 6);
                    RT8_CE(1, 4); RT8_CE(3, 6);
                    RT8_CE(2, 4); RT8_CE(3, 5);
                    RT8_CE(3, 4);
                    // reference of the child in slot s: inner children count from child_base, leaves from prim_base
                    auto child_ref = [&](uint32_t key) -> uint32_t {
                        const uint32_t s = key & 7u;
                        const uint32_t m = (((s & 4u) ? h1.w : h1.z) >> (8u * (s & 3u))) & 0xFFu;
                        return ((m & 0x80u) ? (TRI_TAG | h1.y) : h1.x) + (m & 0x3Fu);
                    };
#pragma unroll
                    for (int j = 7; j >= 1; --j)
                        if (j < nh) RT8_PUSH(st, child_ref(k[j]), signed_t ? 0xFF800000u : (k[j] & ~7u)); // -inf: never dropped
                    cur = child_ref(k[0]);
                } else {
                    pop = true;
                }
            }
            {
                const bool at_leaf = active && !pop && (cur >> 28) != RT_REF_NODE;
                const unsigned long long leaf_mask = __ballot(at_leaf);
                const unsigned long long run_mask = __ballot(active);
                const bool quorum = __popcll(leaf_mask) >= LEAF_VOTE || leaf_mask == run_mask;
                if (at_leaf && quorum) {
                    const CTri* tp = reinterpret_cast<const CTri*>(tbase + (size_t)(cur & 0x0FFFFFFFu) * sizeof(CTri));
                    if (COUNT) ++np;
                    double t, u, v;
                    if (tri_test(d3p(tp->v0), d3p(tp->v1), d3p(tp->v2), r, tmin, best_t, t, u, v)) { best_t = t; best_ref = tp->ref; best32 = f32_round_up(t); }
                    pop = true;
                }
            }
            if (pop) {
                for (;;) {
                    if (st.sp == 0) {
                        out_t[ray_slot] = best_ref == RT_REF_NONE ? D_INF : best_t;
                        out_prim[ray_slot] = best_ref;
                        active = false;
                        break;
                    }
                    uint2 e;
                    RT8_POP(st, e);
                    // e.y is a lower bound of the distance at which the deferred subtree is entered; triangles
                    // may tie at exactly best_t (closed interval), so only strictly farther subtrees are dropped
                    if (!(__uint_as_float(e.y) > best32)) { cur = e.x; break; }
                }
            }
            const int running = __popcll(__ballot(active));
            if (running == 0 || (more && running < REFILL_THRESHOLD)) break;
        }
    }
    if (COUNT) {
        atomicAdd(&stat_counters[0], (unsigned long long)nn);
        atomicAdd(&stat_counters[1], (unsigned long long)np);
    }
}

} // namespace rt
#endif
