// shade_kernels.hip -- one bounce of Camera::ray_color (camera.rs:243-339) for the live paths of a batch, and the
// kernel that finishes a batch's last paths in one launch.  Compiled without the backend's SIOptimizeVGPRLiveRange
// pass (Makefile): with it, shade_hit loses a value on one edge of a divergent branch.
#include "rt_internal.h"
#include "shade.h"

namespace rt {
namespace {

constexpr int SHADE_ROUNDS = 16; // wave-loads of paths whose survivors share one queue reservation
constexpr unsigned long long SHADE_GROUP = 4ull; // consecutive wave-loads a wave shades before it moves on by a stride of all waves
// (two workgroups per CU: without the bound the textured variant needs 263 registers, seven over the limit for a
// second wave per SIMD, which it is worth a few spilled values to keep)
template <bool TEXTURES, bool RR>
__global__ __launch_bounds__(BLOCK, 2) void shade_kernel(DScene S, CamD cam, PathState ps, int qsel, unsigned long long seed, BatchGeom bg, int last_bounce) {
    __shared__ DXform s_xforms[XF_LDS];
    const XfTable xt = xf_fill_lds(S, s_xforms);
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
    if (threadIdx.x < RT_SECTIONS) { s_sec_lanes[threadIdx.x] = 0ull; s_sec_runs[threadIdx.x] = 0ull; s_sec_cycles[threadIdx.x] = 0ull; }
    __syncthreads();
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
                RT_SEC_BEGIN(0);
                RT_SEC_BEGIN(11);
                PathRec* rec = ps.rec + p;
                D3 beta = d3p(rec->beta);
                D3 L = d3p(rec->L);
                Best b;
                b.t = rec->hit_t; b.ref = rec->hit_ref; b.ctx = rec->hit_ctx;
                RT_PHASE(0);
                RT_SEC_END(11);
                if (b.ref == RT_REF_NONE) { // camera.rs:258-260
                    L = L + beta * d3p(cam.background);
                } else {
                    RayD wr; // the world ray, read together with the rest of the record
                    wr.ox = rec->o[0]; wr.oy = rec->o[1]; wr.oz = rec->o[2];
                    wr.dx = rec->d[0]; wr.dy = rec->d[1]; wr.dz = rec->d[2];
                    double time = rec->time;
                    HitRec h;
                    finalize(S, xt, wr, time, b, h, TEXTURES && S.uses_uv != 0u); // only an image texture reads u,v (texture.rs:88-107)
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
                    RT_SEC_BEGIN(12);
                    alive = shade_hit<TEXTURES, RR>(S, h, ray_dir, g, beta, L, nd);
                    RT_SEC_END(12);
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
                RT_SEC_END(0);
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
    __syncthreads();
    if (threadIdx.x < RT_SECTIONS) { // behind the work slots: lanes, runs, cycles per section
        atomicAdd(&ps.counters[C_STATS + threadIdx.x], s_sec_lanes[threadIdx.x]);
        atomicAdd(&ps.counters[C_STATS + RT_SECTIONS + threadIdx.x], s_sec_runs[threadIdx.x]);
        atomicAdd(&ps.counters[C_STATS + 2 * RT_SECTIONS + threadIdx.x], s_sec_cycles[threadIdx.x]);
    }
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
    __shared__ DXform s_xforms[XF_LDS];
    const XfTable xt = xf_fill_lds(S, s_xforms);
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
            finalize(S, xt, r, time, b, h, TEXTURES && S.uses_uv != 0u);
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

} // namespace

void launch_shade(hipStream_t st, int blocks, bool textures, bool rr, const DScene& S, const CamD& cam, const PathState& ps, int qsel,
                  unsigned long long seed, const BatchGeom& bg, int last_bounce) {
    auto k = rr ? (textures ? shade_kernel<true, true> : shade_kernel<false, true>) : (textures ? shade_kernel<true, false> : shade_kernel<false, false>);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(BLOCK), 0, st, S, cam, ps, qsel, seed, bg, last_bounce);
}
void launch_tail(hipStream_t st, int blocks, int media, bool textures, bool rr, bool uv, const DScene& S, const CamD& cam, const PathState& ps,
                 int qsel, unsigned long long seed, const BatchGeom& bg, int first_bounce, int max_depth) {
    // (nested media: the full shading code whatever the textures, to keep the number of variants down)
    auto tail = uv ? (rr ? tail_kernel<2, true, true, true> : tail_kernel<2, true, false, true>)
              : media == 2 ? (rr ? tail_kernel<2, true, true> : tail_kernel<2, true, false>)
              : rr ? (media ? (textures ? tail_kernel<1, true, true> : tail_kernel<1, false, true>)
                            : (textures ? tail_kernel<0, true, true> : tail_kernel<0, false, true>))
                   : (media ? (textures ? tail_kernel<1, true, false> : tail_kernel<1, false, false>)
                            : (textures ? tail_kernel<0, true, false> : tail_kernel<0, false, false>));
    hipLaunchKernelGGL(tail, dim3(blocks), dim3(BLOCK), 0, st, S, cam, ps, qsel, seed, bg, first_bounce, max_depth);
}
hipError_t occupancy_shade(bool textures, int* per_cu) {
    if (textures) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, shade_kernel<true, false>, BLOCK, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, shade_kernel<false, false>, BLOCK, 0);
}
hipError_t occupancy_tail(int media, int* per_cu) {
    if (media > 1) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, tail_kernel<2, true, false>, BLOCK, 0);
    if (media) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, tail_kernel<1, true, false>, BLOCK, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, tail_kernel<0, true, false>, BLOCK, 0);
}

} // namespace rt
