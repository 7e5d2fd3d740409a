// gather_probe.hip -- how fast can MI355X serve per-lane pointer-chasing row gathers?
// Every lane walks its own chain through a table of 128-byte rows (next row = first word of the
// current row), the access pattern of a BVH traversal.  Variants differ in how much of the row a
// lane reads and how the reads are issued.  Prints rows/s and requested bytes/s per variant.
//   hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe [table_MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROW = 128;

// V: number of 16-byte loads per row (1..8), all issued back to back
template <int V>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ table, uint32_t nrows, int iters, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nrows;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float4* row = table + (size_t)idx * (ROW / 16);
        float4 v[V];
#pragma unroll
        for (int k = 0; k < V; ++k) v[k] = row[k];
#pragma unroll
        for (int k = 1; k < V; ++k) acc += v[k].y + v[k].w;
        idx = __float_as_uint(v[0].x) % nrows;
    }
    out[blockIdx.x * 256u + threadIdx.x] = idx + (uint32_t)acc;
}

// first 16 bytes, wait, then the rest (would hit in L1 if the line stayed there)
template <int V>
__global__ __launch_bounds__(256) void chase_warm(const float4* __restrict__ table, uint32_t nrows, int iters, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nrows;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const float4* row = table + (size_t)idx * (ROW / 16);
        float4 v0 = row[0];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t nxt = __float_as_uint(v0.x) % nrows;
        float4 v[V];
#pragma unroll
        for (int k = 1; k < V; ++k) v[k] = row[k];
#pragma unroll
        for (int k = 1; k < V; ++k) acc += v[k].y + v[k].w;
        idx = nxt;
    }
    out[blockIdx.x * 256u + threadIdx.x] = idx + (uint32_t)acc;
}

// cooperative: the 8 lanes of a group read the 8 x 16 bytes of ONE row in one instruction; 8
// instructions serve the 8 rows the group's lanes stand on.  Data stay in registers (each lane ends
// up with 1/8 of 8 rows): measures the memory side of a cooperative fetch without redistribution.
__global__ __launch_bounds__(256) void chase_coop(const float4* __restrict__ table, uint32_t nrows, int iters, uint32_t* out) {
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nrows;
    const int lane = threadIdx.x & 63, sub = lane & 7, grp = lane & ~7;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t r = (uint32_t)__shfl((int)idx, grp + k);
            v[k] = table[(size_t)r * (ROW / 16) + sub];
        }
        // lane sub==0 of slot k holds word 0 of row k: hand each lane its own next index
        uint32_t nxt = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t w = (uint32_t)__shfl((int)__float_as_uint(v[k].x), grp);
            if (sub == k) nxt = w;
            acc += v[k].y + v[k].w;
        }
        idx = nxt % nrows;
    }
    out[blockIdx.x * 256u + threadIdx.x] = idx + (uint32_t)acc;
}

// cooperative through LDS: same fetch, rows staged in LDS, then every lane reads V x 16 bytes of ITS row
template <int V>
__global__ __launch_bounds__(256) void chase_coop_lds(const float4* __restrict__ table, uint32_t nrows, int iters, uint32_t* out) {
    __shared__ float4 stage[4][64][9]; // per wave: 64 rows x (8 + 1 pad) float4
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % nrows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 7, grp = lane & ~7;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t r = (uint32_t)__shfl((int)idx, grp + k);
            v[k] = table[(size_t)r * (ROW / 16) + sub];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) stage[wave][grp + k][sub] = v[k];
        __builtin_amdgcn_wave_barrier();
        float4 w[V];
#pragma unroll
        for (int k = 0; k < V; ++k) w[k] = stage[wave][lane][k];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 1; k < V; ++k) acc += w[k].y + w[k].w;
        idx = __float_as_uint(w[0].x) % nrows;
    }
    out[blockIdx.x * 256u + threadIdx.x] = idx + (uint32_t)acc;
}

template <class F> static void run(const char* name, F launch, int blocks, int iters, int bytes_per_row) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double rows = (double)blocks * 256 * iters;
    printf("%-22s %8.3f ms  %7.2f Grows/s  %7.2f TB/s requested  %6.2f TB/s as 128-B lines\n", name, ms, rows / ms / 1e6,
           rows * bytes_per_row / ms / 1e9, rows * 128 / ms / 1e9);
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 128;
    const uint32_t nrows = (uint32_t)(mb * 1024 * 1024 / ROW);
    std::vector<uint32_t> h((size_t)nrows * (ROW / 4));
    uint64_t s = 88172645463325252ull;
    for (auto& w : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
    float4* table; uint32_t* out;
    CK(hipMalloc(&table, h.size() * 4)); CK(hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int blocks = 256 * 8, iters = 256;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    printf("table %zu MB, %u rows, %d blocks x 256 lanes x %d rows each\n", mb, nrows, blocks, iters);
#define RUN(name, kern, bytes) run(name, [&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, table, nrows, iters, out); }, blocks, iters, bytes)
    RUN("lane 1x16B", chase<1>, 16);
    RUN("lane 2x16B", chase<2>, 32);
    RUN("lane 4x16B (64-B node)", chase<4>, 64);
    RUN("lane 5x16B (80-B node)", chase<5>, 80);
    RUN("lane 7x16B (now)", chase<7>, 112);
    RUN("lane 8x16B", chase<8>, 128);
    RUN("warm 1+6", chase_warm<7>, 112);
    RUN("warm 1+3", chase_warm<4>, 64);
    RUN("coop regs 8x", chase_coop, 128);
    RUN("coop lds, read 7", chase_coop_lds<7>, 128);
    RUN("coop lds, read 4", chase_coop_lds<4>, 128);
    return 0;
}
