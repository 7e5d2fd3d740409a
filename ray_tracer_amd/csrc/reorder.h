// reorder.h -- pass in front of the primitive-BVH kernel for large batches: bins the rays of a batch by the cell
// of a 16 x 16 x 16 grid over the scene that holds their origin (Morton order of the cells) and writes one
// 128-byte prepared record per ray, in bin order.
//
// Why: the kernel's waves take consecutive pieces of the batch, so after the pass the rays in flight at any moment
// start in the same few cells and walk the same part of the tree: of the 128-byte lines a ray of an incoherent
// batch pulls through the fabric two thirds are then found in L2 (measured on the 2^20-triangle soup: 56.6 GB ->
// 19 GB per 2^24 rays).  And the per-ray set-up that the kernel used to run at refill time with a handful of lanes
// (three f64 divisions, the f32 form of the ray with its error margins) runs here with every lane busy; a refill
// becomes eight 16-byte loads from one line, so the kernel can refill as soon as a few lanes are idle.
//
// The order of the rays inside a bin is not defined (positions come from atomic counters); results are written
// back through the record's original ray index, so the batch's output does not depend on it.
#ifndef RT_REORDER_H
#define RT_REORDER_H

#include "traverse.h"

namespace rt {

constexpr int REORDER_BITS = 4;                        // cells per axis = 2^4
constexpr int REORDER_BINS = 1 << (3 * REORDER_BITS);  // 4096
constexpr int REORDER_BLOCK = 256;
constexpr int REORDER_TILE = 16384;                    // rays per workgroup of the scatter pass
constexpr int SCATTER_BLOCK = 256;

struct alignas(128) PRay {      // one line: everything the kernel needs to start a ray
    double o[3], d[3];          // ray.rs:6-7
    double time, tmin, tmax;    // ray.rs:8 and the query interval
    float inv[3], cn[3], cf[3]; // Ray32 (device_math.h): 1/d and the two offset terms of the conservative slab test
    uint32_t idx;               // index of the ray in the caller's batch
    uint32_t neg;               // bit k: d[k] < 0
    float tmin32, tmax32;       // tmin rounded down, tmax rounded up
    uint32_t _pad;
};
static_assert(sizeof(PRay) == 128, "a prepared ray is one cache line");

// The packed tree (triangles only, default lower end of the interval) needs less: origin, direction, 1/d and the
// ray's index, half a line.  The kernel derives the offset terms of the slab test from 1/d when it starts the ray
// (ray32_from_inv), and reads an upper end of the interval, if the caller gave one, from the caller's batch.
struct alignas(64) PRayC {
    double o[3], d[3];
    float inv[3];               // (float)(1 / d)
    uint32_t idx;
};
static_assert(sizeof(PRayC) == 64, "a compact prepared ray is half a cache line");

// Ray32 from the f32 inverse alone: c = -(o * inv32) rounded to f32.  Both terms of t = b * inv32 + c then carry
// the same rounding of the inverse, a relative error of 2^-24 in t, i.e. at most 2^-24 * |inv| * (|b| + |o|) more
// than the form make_ray32 builds; the margin is widened by a quarter (10 instead of 8 units of 2^-24) to cover it.
RT_DEV void ray32_from_inv(const RayD& r, const float inv32[3], float scene_extent, Ray32& q) {
    const double o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float c = (float)(-(o[k] * (double)inv32[k]));
        const float e = 5.9604644775390625e-07f * fabsf(inv32[k]) * (scene_extent + fabsf((float)o[k])); // 1.25 * 2^-21
        q.inv[k] = inv32[k];
        q.cn[k] = c - e;
        q.cf[k] = c + e;
        q.neg[k] = d[k] < 0.0;
    }
}

struct ReorderGrid {            // cell = clamp((o - lo) * scale, 0, 15) per axis
    double lo[3];
    double scale[3];
};

RT_DEV uint32_t morton_part(uint32_t v) { // 4 bits -> every third bit
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6);
}

RT_DEV uint32_t ray_bin(const ReorderGrid& g, double ox, double oy, double oz) {
    const double c[3] = {(ox - g.lo[0]) * g.scale[0], (oy - g.lo[1]) * g.scale[1], (oz - g.lo[2]) * g.scale[2]};
    uint32_t key = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // NaN and out-of-range origins land in a border cell: the bin only steers locality
        const double v = fmin(fmax(c[k], 0.0), (double)((1 << REORDER_BITS) - 1));
        key |= morton_part((uint32_t)(int)v) << k;
    }
    return key;
}

#ifdef RT_REORDER_KERNELS // the kernels are compiled into the translation unit that launches them (mi355rt.hip)
// Pass 1: bin of every ray (kept for pass 2) and the batch's histogram.
__global__ __launch_bounds__(REORDER_BLOCK) void reorder_count_kernel(rt_ray_batch rays, unsigned long long n, ReorderGrid g, uint16_t* bins, uint32_t* hist) {
    __shared__ uint32_t s_hist[REORDER_BINS];
    for (int k = threadIdx.x; k < REORDER_BINS; k += REORDER_BLOCK) s_hist[k] = 0u;
    __syncthreads();
    const unsigned long long stride = (unsigned long long)gridDim.x * REORDER_BLOCK;
    for (unsigned long long i = (unsigned long long)blockIdx.x * REORDER_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t b = ray_bin(g, rays.ox[i], rays.oy[i], rays.oz[i]);
        bins[i] = (uint16_t)b;
        atomicAdd(&s_hist[b], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < REORDER_BINS; k += REORDER_BLOCK)
        if (s_hist[k]) atomicAdd(&hist[k], s_hist[k]);
}

// Pass 2: exclusive prefix sum of the histogram (one workgroup): cursor[b] = first slot of bin b.  The
// histogram is cleared for the next batch.
__global__ __launch_bounds__(1024) void reorder_scan_kernel(uint32_t* hist, uint32_t* cursor) {
    __shared__ uint32_t s_part[1024];
    constexpr int PER = REORDER_BINS / 1024;
    uint32_t v[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { v[k] = hist[threadIdx.x * PER + k]; sum += v[k]; }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) { // Hillis-Steele inclusive scan of the per-thread sums
        const uint32_t add = threadIdx.x >= (uint32_t)off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += add;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) { cursor[threadIdx.x * PER + k] = run; run += v[k]; hist[threadIdx.x * PER + k] = 0u; }
}

// Pass 3: every workgroup counts its tile per bin, reserves that many consecutive slots of each bin with one
// atomic, and writes the prepared records of its rays into them.
// A record is a whole 128-byte line, and it has to leave the wave as one: a lane that stores its own record issues
// eight 16-byte stores into a line no other lane of the instruction touches, and each of them went to memory as a
// transaction of its own (2^24 records took 1.1 ms, the time of eight times their bytes).  So the wave stages its 64
// records in LDS and stores them with eight lanes per record: every store instruction writes eight complete lines.
// COMPACT: 64-byte records (PRayC), four lanes per record.
template <bool COMPACT>
__global__ __launch_bounds__(SCATTER_BLOCK) void reorder_scatter_kernel(rt_ray_batch rays, unsigned long long n, uint32_t opt, double tmin_default,
                                                                          float scene_extent, const uint16_t* bins, uint32_t* cursor, PRay* out) {
    constexpr int PIECES = COMPACT ? 4 : 8;                                                   // 16-byte pieces per record
    constexpr int WAVES = SCATTER_BLOCK / 64, PER = REORDER_TILE / SCATTER_BLOCK, STRIDE = PIECES + 1; // odd number of 16-byte pieces per staged record: conflict-free
    __shared__ uint32_t s_count[REORDER_BINS];
    __shared__ uint32_t s_base[REORDER_BINS];
    __shared__ uint4 s_stage[WAVES * 64 * STRIDE];
    __shared__ uint32_t s_pos[WAVES * 64];
    const unsigned long long tile0 = (unsigned long long)blockIdx.x * REORDER_TILE;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < REORDER_BINS; k += SCATTER_BLOCK) s_count[k] = 0u;
    __syncthreads();
    for (int j = 0; j < PER; ++j) {
        const unsigned long long i = tile0 + threadIdx.x + (unsigned long long)j * SCATTER_BLOCK;
        if (i < n) atomicAdd(&s_count[bins[i]], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < REORDER_BINS; k += SCATTER_BLOCK) {
        const uint32_t c = s_count[k];
        s_base[k] = c ? atomicAdd(&cursor[k], c) : 0u;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < REORDER_BINS; k += SCATTER_BLOCK) s_count[k] = 0u;
    __syncthreads();
    uint4* const stage = s_stage + wave * 64 * STRIDE;
    uint32_t* const spos = s_pos + wave * 64;
    for (int j = 0; j < PER; ++j) { // wave-uniform trip count: the stores below are cooperative
        const unsigned long long i = tile0 + threadIdx.x + (unsigned long long)j * SCATTER_BLOCK;
        uint32_t pos = 0xFFFFFFFFu;
        if (i < n) {
            const uint32_t b = bins[i];
            pos = s_base[b] + atomicAdd(&s_count[b], 1u);
            RayD r;
            r.ox = rays.ox[i]; r.oy = rays.oy[i]; r.oz = rays.oz[i];
            r.dx = rays.dx[i]; r.dy = rays.dy[i]; r.dz = rays.dz[i];
            set_inverse(r);
            if (COMPACT) {
                PRayC p;
                p.o[0] = r.ox; p.o[1] = r.oy; p.o[2] = r.oz;
                p.d[0] = r.dx; p.d[1] = r.dy; p.d[2] = r.dz;
                p.inv[0] = (float)r.ix; p.inv[1] = (float)r.iy; p.inv[2] = (float)r.iz;
                p.idx = (uint32_t)i;
                const uint4* src = reinterpret_cast<const uint4*>(&p);
#pragma unroll
                for (int k = 0; k < PIECES; ++k) stage[lane * STRIDE + k] = src[k];
            } else {
                Ray32 q;
                make_ray32(r, scene_extent, q);
                PRay p;
                p.o[0] = r.ox; p.o[1] = r.oy; p.o[2] = r.oz;
                p.d[0] = r.dx; p.d[1] = r.dy; p.d[2] = r.dz;
                p.time = (opt & 1u) ? rays.time[i] : 0.0;
                p.tmin = (opt & 2u) ? rays.tmin[i] : tmin_default;
                p.tmax = (opt & 4u) ? rays.tmax[i] : D_INF;
#pragma unroll
                for (int k = 0; k < 3; ++k) { p.inv[k] = q.inv[k]; p.cn[k] = q.cn[k]; p.cf[k] = q.cf[k]; }
                p.idx = (uint32_t)i;
                p.neg = (q.neg[0] ? 1u : 0u) | (q.neg[1] ? 2u : 0u) | (q.neg[2] ? 4u : 0u);
                p.tmin32 = f32_round_down(p.tmin);
                p.tmax32 = f32_round_up(p.tmax);
                p._pad = 0u;
                const uint4* src = reinterpret_cast<const uint4*>(&p);
#pragma unroll
                for (int k = 0; k < PIECES; ++k) stage[lane * STRIDE + k] = src[k];
            }
        }
        spos[lane] = pos;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < PIECES; ++k) { // 64 / PIECES records per store instruction, PIECES lanes each
            const uint32_t rec = (uint32_t)k * (64u / PIECES) + lane / PIECES, piece = lane % PIECES;
            const uint32_t rpos = spos[rec];
            const uint4 v = stage[rec * STRIDE + piece];
            if (rpos != 0xFFFFFFFFu) reinterpret_cast<uint4*>(out)[(size_t)rpos * PIECES + piece] = v;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

#endif // RT_REORDER_KERNELS

} // namespace rt
#endif
