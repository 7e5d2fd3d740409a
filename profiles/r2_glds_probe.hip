// r2_glds_probe.hip -- per-lane pointer chasing through 64-byte records (the node fetch of the primitive-BVH
// kernel), two ways:
//   lane : every lane reads its own record with four 16-byte loads (four L1 look-ups per lane and step);
//   quad : the four lanes of a quad read the four 16-byte pieces of ONE record with one LDS-DMA instruction
//          (global_load_lds_dwordx4: one look-up per record), four instructions serve the quad's four records;
//          the records land lane-linear in a per-wave staging area and every lane reads its own back with
//          ds_read_b128.  A fraction of the lanes can be switched off (lanes that stand on a leaf in the real
//          kernel) to check that masked lanes neither load nor shift the LDS placement of the others.
// Both variants must produce the same checksums.
//   hipcc --offload-arch=gfx950 -O3 -o r2_glds_probe r2_glds_probe.hip && ./r2_glds_probe [table_MB] [active_percent]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int REC = 64;
constexpr int REGION = 1024 + 16;  // one LDS-DMA instruction writes 64 x 16 bytes; the 16-byte stagger keeps the read-back conflict-free

__device__ __forceinline__ bool lane_on(uint32_t tid, int it, uint32_t active_pct) {
    uint32_t h = (tid * 2654435761u) ^ ((uint32_t)it * 40503u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    return (h % 100u) < active_pct;
}

__global__ __launch_bounds__(256) void chase_lane(const uint4* __restrict__ table, uint32_t nrows, int iters, uint32_t active_pct, uint32_t* out) {
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    uint32_t idx = tid * 2654435761u % nrows, acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (lane_on(tid, it, active_pct)) {
            const uint4* row = table + (size_t)idx * (REC / 16);
            const uint4 a = row[0], b = row[1], c = row[2], d = row[3];
            acc += a.y ^ b.z ^ c.w ^ d.x ^ d.w;
            idx = a.x % nrows;
        }
    }
    out[tid] = idx + acc;
}

__global__ __launch_bounds__(256) void chase_quad(const uint4* __restrict__ table, uint32_t nrows, int iters, uint32_t active_pct, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char stage[4 * 4 * REGION];
    const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    char* const wstage = stage + wave * (4 * REGION);
    uint32_t idx = tid * 2654435761u % nrows, acc = 0;
    for (int it = 0; it < iters; ++it) {
        const bool on = lane_on(tid, it, active_pct);
        const uint32_t mine = on ? idx : 0xFFFFFFFFu;
#define QUAD_LOAD(i)                                                                                                   \
        {                                                                                                              \
            const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)mine, (i) * 0x55, 0xf, 0xf, false);      \
            if (r != 0xFFFFFFFFu)                                                                                      \
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(table + (size_t)r * (REC / 16) + (lane & 3u)), \
                                                 (__attribute__((address_space(3))) void*)(wstage + (i) * REGION), 16, 0, 0); \
        }
        QUAD_LOAD(0) QUAD_LOAD(1) QUAD_LOAD(2) QUAD_LOAD(3)
#undef QUAD_LOAD
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (on) {
            const uint4* row = reinterpret_cast<const uint4*>(wstage + (lane & 3u) * REGION + (lane >> 2) * REC);
            const uint4 a = row[0], b = row[1], c = row[2], d = row[3];
            acc += a.y ^ b.z ^ c.w ^ d.x ^ d.w;
            idx = a.x % nrows;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the read-back is done before the next round overwrites the staging area
    }
    out[tid] = idx + acc;
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 64;
    const uint32_t pct = argc > 2 ? atoi(argv[2]) : 100;
    const uint32_t nrows = (uint32_t)(mb * 1024 * 1024 / REC);
    std::vector<uint32_t> h((size_t)nrows * (REC / 4));
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
    uint4* table; uint32_t *out_a, *out_b;
    CK(hipMalloc(&table, h.size() * 4)); CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 5, iters = 256;
    const size_t nthreads = (size_t)blocks * 256;
    CK(hipMalloc(&out_a, nthreads * 4)); CK(hipMalloc(&out_b, nthreads * 4));
    printf("table %zu MB of %d-byte records, %d blocks x 256 lanes x %d steps, %u %% of the lanes active per step\n", mb, REC, blocks, iters, pct);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto kernel, uint32_t* out) {
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, nrows, iters, pct, out); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, table, nrows, iters, pct, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double recs = (double)nthreads * iters * pct / 100.0;
        printf("%-6s %8.3f ms  %7.2f G records/s  %6.2f TB/s of records\n", name, ms, recs / ms / 1e6, recs * REC / ms / 1e9);
    };
    time("lane", chase_lane, out_a);
    time("quad", chase_quad, out_b);
    std::vector<uint32_t> a(nthreads), b(nthreads);
    CK(hipMemcpy(a.data(), out_a, nthreads * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), out_b, nthreads * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < nthreads; ++i) bad += a[i] != b[i];
    printf("checksums differing between the two variants: %zu of %zu\n", bad, nthreads);
    return bad ? 1 : 0;
}
