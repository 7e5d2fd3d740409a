// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE for the access pattern of the BVH kernel:
// every lane reads seven 16-byte pieces of its own random 128-byte row.  The rows are drawn without
// replacement from a table far larger than the 256 MiB Infinity Cache, so each row is fetched from
// HBM exactly once and the true traffic is known: rows * 128 B (whole lines) or rows * 112 B (bytes used).
// Measured on MI355X (rocprofv3 --pmc FETCH_SIZE / TCC_EA0_RDREQ_sum): both kernels issue ONE fabric read request
// per row (4.33 M requests for 4.19 M rows + the index array), whether a lane reads 112 bytes of the row or
// one 64-byte half of it, and FETCH_SIZE tallies each at 64 B: requests are per 128-byte line, so the true
// line traffic is FETCH_SIZE x 2 for the 128-byte and the 64-byte node format alike.
// Build: hipcc --offload-arch=gfx950 -O3 -o profiles/fetch_calib profiles/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

__global__ void gather_rows(const float4* table, const unsigned* rows, unsigned n, float* out) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* r = table + (size_t)rows[i] * 8;
    float4 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5], g = r[6];
    out[i] = a.x + b.y + c.z + d.w + e.x + f.y + g.z;
}

// The 64-byte node format: every lane reads four 16-byte pieces, i.e. one HALF of a random 128-byte line.
__global__ void gather_half_rows(const float4* table, const unsigned* rows, unsigned n, float* out) {
    unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4* r = table + (size_t)rows[i] * 8 + (rows[i] & 1u) * 4; // lower or upper half
    float4 a = r[0], b = r[1], c = r[2], d = r[3];
    out[i] = a.x + b.y + c.z + d.w;
}

int main() {
    const size_t table_rows = 1ull << 24;  // 2 GiB of 128-byte rows
    const unsigned n = 1u << 22;           // 4 Mi rows gathered = 512 MiB of lines
    float4* table; unsigned* rows; float* out;
    hipMalloc(&table, table_rows * 128); hipMalloc(&rows, n * 4ull); hipMalloc(&out, n * 4ull);
    hipMemset(table, 0, table_rows * 128);
    std::vector<unsigned> idx(table_rows);
    std::iota(idx.begin(), idx.end(), 0u);
    std::mt19937_64 g(7);
    for (unsigned i = 0; i < n; ++i) { size_t j = i + g() % (table_rows - i); std::swap(idx[i], idx[j]); }
    hipMemcpy(rows, idx.data(), n * 4ull, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        // evict: stream 1 GiB of other rows through the caches between repetitions
        hipMemset(table + (table_rows / 2) * 8, rep, (table_rows / 2) * 128);
        gather_rows<<<(n + 255) / 256, 256>>>(table, rows, n, out);
        hipDeviceSynchronize();
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(table + (table_rows / 2) * 8, rep, (table_rows / 2) * 128);
        gather_half_rows<<<(n + 255) / 256, 256>>>(table, rows, n, out);
        hipDeviceSynchronize();
    }
    printf("rows %u lines_bytes %llu used_bytes %llu (gather_rows); half rows: lines_bytes %llu used_bytes %llu (gather_half_rows)\n", n,
           (unsigned long long)n * 128ull, (unsigned long long)n * 112ull, (unsigned long long)n * 128ull, (unsigned long long)n * 64ull);
    return 0;
}
