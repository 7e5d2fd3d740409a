// extend_kernels.hip -- the renderer's kernels around shading: Camera::get_ray for a batch (camera.rs:207-240), closest
// hit for the live paths of a bounce (camera.rs:258), the stratum sum (camera.rs:110-131) and the queue bookkeeping.
#include "rt_internal.h"

namespace rt {
namespace {

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

void launch_init_batch(hipStream_t st, unsigned long long* counters, unsigned long long npaths) {
    hipLaunchKernelGGL(init_batch_kernel, dim3(1), dim3(1), 0, st, counters, npaths);
}
void launch_advance(hipStream_t st, unsigned long long* counters) { hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, counters); }
void launch_reset_work(hipStream_t st, unsigned long long* counters) { hipLaunchKernelGGL(reset_work_kernel, dim3(1), dim3(1), 0, st, counters); }
void launch_raygen(hipStream_t st, const CamD& cam, const BatchGeom& bg, const PathState& ps, unsigned long long seed) {
    hipLaunchKernelGGL(raygen_kernel, dim3((bg.npaths + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, cam, bg, ps, seed);
}
void launch_extend(hipStream_t st, int blocks, int media, bool uv, const DScene& S, const PathState& ps, int qsel, unsigned long long seed,
                   const BatchGeom& bg, int32_t image_width, int rest_only) {
    auto extend = media == 2 ? extend_kernel<2> : (media ? extend_kernel<1> : extend_kernel<0>);
    if (uv) extend = extend_kernel<2, true>;
    hipLaunchKernelGGL(extend, dim3(blocks), dim3(BLOCK), 0, st, S, ps, qsel, seed, bg, image_width, rest_only);
}
void launch_accumulate(hipStream_t st, const PathState& ps, const BatchGeom& bg, double* accum, int32_t image_width) {
    hipLaunchKernelGGL(accumulate_kernel, dim3((bg.npix + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, ps, bg, accum, image_width);
}
hipError_t occupancy_extend(int media, int* per_cu) {
    if (media > 1) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, extend_kernel<2>, BLOCK, 0);
    if (media) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, extend_kernel<1>, BLOCK, 0);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, extend_kernel<0>, BLOCK, 0);
}

} // namespace rt
