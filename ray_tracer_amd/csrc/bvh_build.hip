// bvh_build.hip -- BvhNode::from_objects (bvh.rs:20-77) on the device, in the reference's split order.
//
// The reference builds top-down: bbox = union over the span, axis = Aabb::longest_axis (aabb.rs:109-123), STABLE sort
// of the span on bbox.axis(axis).min (bvh.rs:52, comparators bvh.rs:79-115), mid = start + span / 2, a span of one
// stores its object as both children, a span of two keeps its pair in incoming order.  The host flattener restates
// that recursion (flatten.cpp build_bvh); for a mesh of a million triangles its O(N log^2 N) sorts are what scene
// creation waits for.  Here the same tree is built level by level, every span of a level at once:
//   1. reduce: union box of every span (orderable-integer min / max, workgroup-local first);
//   2. nodes:  one thread per span writes its node in pre-order position (bounds, flags = axis << 1) and, for a
//              span of three or more, the two child spans of the next level; spans of one or two become leaves;
//   3. keys:   every item takes the box minimum on its span's axis as a 64-bit orderable key;
//   4. sort:   a stable radix sort of the whole array by key, then a stable radix sort by span number: every span is
//              back in its place, ordered by key, equal keys in incoming order -- the reference's stable sort.
// The sorts are rocPRIM's device-wide radix sorts (LSD radix sorts, stable by construction).
// The result is the same node array, byte for byte, as the host recursion's (tests compare them).
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>
#include <string>
#include <vector>

#include "flatten.h"

namespace rt {
namespace {

#define BUILD_TRY(expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); rc = RT_E_HIP; goto done; } \
    } while (0)

constexpr int BB = 256;

struct Span { uint32_t start, end, node, axis; }; // items [start, end) of the current order; node index inside this BVH

__host__ __device__ inline uint64_t node_count(uint64_t n) { // N(1) = N(2) = 1, N(n) = 1 + N(n / 2) + N(n - n / 2), without recursion:
    // all spans of one depth have size k or k + 1; a of them have size k, b have size k + 1
    if (n <= 2) return 1;
    uint64_t k = n, a = 1, b = 0, total = 0;
    for (;;) {
        total += a + b;
        if (k + 1 <= 2 || (k <= 2 && b == 0)) break;       // every span is a leaf
        if (k <= 2) { total += 2 * b; break; }              // k == 2: the size-3 spans split into (1, 2), both leaves
        if (k % 2 == 0) { a = 2 * a + b; k = k / 2; }        // k -> (k/2, k/2);        k+1 -> (k/2, k/2 + 1)
        else { const uint64_t a2 = a; a = a2; b = a2 + 2 * b; k = (k - 1) / 2; } // k -> ((k-1)/2, (k+1)/2); k+1 -> ((k+1)/2, (k+1)/2)
    }
    return total;
}

__device__ inline uint64_t orderable_bits(double x) { // monotone map f64 -> u64 that keeps every bit pattern apart (x is never NaN here)
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ inline uint64_t orderable(double x) { // the sort key: -0.0 and +0.0 compare equal in the reference (partial_cmp)
    if (x == 0.0) x = 0.0;
    return orderable_bits(x);
}
__device__ inline double from_orderable(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)u);
}

// positions -> span number of the item standing there (packed: span << 32 | item)
__global__ void init_kernel(uint64_t* packed, uint32_t n) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i < n) packed[i] = (uint64_t)i; // span 0
}

// accumulators of the union boxes: minima start at the largest key, maxima at the smallest
__global__ void acc_init_kernel(uint64_t* acc, uint32_t count) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i < count) acc[i] = (i & 1u) ? 0ull : ~0ull;
}

// union box of every span: acc[span * 6 + k] holds orderable(min) for even k and orderable(max) for odd k
__global__ __launch_bounds__(BB) void reduce_kernel(const BuildItem* items, const uint64_t* packed, uint32_t n, uint64_t* acc) {
    __shared__ uint64_t s_acc[6];
    __shared__ uint32_t s_first, s_mixed;
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (threadIdx.x == 0) { s_mixed = 0u; s_first = 0xFFFFFFFFu; }
    if (threadIdx.x < 6) s_acc[threadIdx.x] = (threadIdx.x & 1) ? 0ull : ~0ull;
    __syncthreads();
    uint32_t span = 0;
    uint64_t v[6];
    if (i < n) {
        const uint64_t p = packed[i];
        span = (uint32_t)(p >> 32);
        const BuildItem& it = items[(uint32_t)p];
        for (int k = 0; k < 6; ++k) v[k] = orderable_bits(it.box.b[k]);
        if (threadIdx.x == 0 || i == blockIdx.x * BB) s_first = span;
    }
    __syncthreads();
    if (i < n && span != s_first) s_mixed = 1u;
    __syncthreads();
    if (i < n) {
        if (s_mixed) { // the workgroup straddles spans (small spans, deep levels): every item for itself
            for (int k = 0; k < 6; ++k) {
                if (k & 1) atomicMax((unsigned long long*)&acc[(size_t)span * 6 + k], (unsigned long long)v[k]);
                else atomicMin((unsigned long long*)&acc[(size_t)span * 6 + k], (unsigned long long)v[k]);
            }
        } else {
            for (int k = 0; k < 6; ++k) {
                if (k & 1) atomicMax((unsigned long long*)&s_acc[k], (unsigned long long)v[k]);
                else atomicMin((unsigned long long*)&s_acc[k], (unsigned long long)v[k]);
            }
        }
    }
    __syncthreads();
    if (!s_mixed && threadIdx.x < 6 && s_first != 0xFFFFFFFFu) {
        const int k = threadIdx.x;
        if (k & 1) atomicMax((unsigned long long*)&acc[(size_t)s_first * 6 + k], (unsigned long long)s_acc[k]);
        else atomicMin((unsigned long long*)&acc[(size_t)s_first * 6 + k], (unsigned long long)s_acc[k]);
    }
}

// one thread per span of this level: its node, and its two child spans for the next level
__global__ void node_kernel(Span* spans, uint32_t n_spans, const uint64_t* acc, const BuildItem* items, const uint64_t* packed, DNode* nodes,
                            uint32_t* leaf_pos, uint64_t base, Span* next) {
    const uint32_t s = blockIdx.x * BB + threadIdx.x;
    if (s >= n_spans) return;
    Span sp = spans[s];
    Span l{0, 0, 0, 0}, r{0, 0, 0, 0};
    const uint32_t span = sp.end - sp.start;
    if (span > 0) {
        DNode nd;
        for (int k = 0; k < 6; ++k) nd.bounds[k] = from_orderable(acc[(size_t)s * 6 + k]);
        const double sx = nd.bounds[1] - nd.bounds[0], sy = nd.bounds[3] - nd.bounds[2], sz = nd.bounds[5] - nd.bounds[4];
        const uint32_t axis = sx > sy ? (sx > sz ? 0u : 2u) : (sy > sz ? 1u : 2u); // aabb.rs:109-123
        nd.flags = axis << 1;
        nd._pad = 0;
        if (span == 1) {
            nd.left = nd.right = items[(uint32_t)packed[sp.start]].ref;
            leaf_pos[sp.node] = sp.start;
        } else if (span == 2) {
            nd.left = items[(uint32_t)packed[sp.start]].ref;
            nd.right = items[(uint32_t)packed[sp.start + 1]].ref;
            leaf_pos[sp.node] = sp.start;
        } else {
            const uint32_t mid = sp.start + span / 2;
            l.start = sp.start; l.end = mid; l.node = sp.node + 1;
            r.start = mid; r.end = sp.end; r.node = l.node + (uint32_t)node_count(mid - sp.start);
            nd.left = make_ref(RT_REF_NODE, (uint32_t)(base + l.node));
            nd.right = make_ref(RT_REF_NODE, (uint32_t)(base + r.node));
            leaf_pos[sp.node] = 0xFFFFFFFFu;
        }
        nodes[sp.node] = nd;
        sp.axis = axis;
        spans[s] = sp;
    }
    next[2 * s] = l;
    next[2 * s + 1] = r;
}

// sort key of every item: the box minimum along its span's axis (spans that do not split keep their order: key 0)
__global__ void key_kernel(const BuildItem* items, const uint64_t* packed, uint32_t n, const Span* spans, uint64_t* keys) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = packed[i];
    const Span sp = spans[(uint32_t)(p >> 32)];
    keys[i] = (sp.end - sp.start >= 3u) ? orderable(items[(uint32_t)p].box.b[2 * sp.axis]) : 0ull;
}

// after the sorts: the item at position i belongs to the left or right child span of its old span
__global__ void descend_kernel(uint64_t* packed, uint32_t n, const Span* spans) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const uint64_t p = packed[i];
    const uint32_t s = (uint32_t)(p >> 32);
    const Span sp = spans[s];
    const uint32_t span = sp.end - sp.start;
    const uint32_t mid = sp.start + span / 2;
    const uint32_t child = 2u * s + ((span >= 3u && i >= mid) ? 1u : 0u);
    packed[i] = ((uint64_t)child << 32) | (p & 0xFFFFFFFFull);
}

__device__ inline float f32_down_dev(double x) { float f = (float)x; if ((double)f > x) f = __uint_as_float(__float_as_uint(f) + (f > 0.0f ? 0xFFFFFFFFu : 1u)); return f; }
__device__ inline float f32_up_dev(double x) { float f = (float)x; if ((double)f < x) f = __uint_as_float(__float_as_uint(f) + (f >= 0.0f ? 1u : 0xFFFFFFFFu)); return f; }
__device__ inline void widen_dev(const double* b, float* o) {
    for (int k = 0; k < 3; ++k) { o[2 * k] = f32_down_dev(b[2 * k]); o[2 * k + 1] = f32_up_dev(b[2 * k + 1]); }
}

// the wide node of every node: both child boxes rounded outward to f32 (children: nodes of this BVH, or items)
__global__ void wide_kernel(const DNode* nodes, uint32_t n_nodes, const uint32_t* leaf_pos, const BuildItem* items, const uint64_t* packed, uint64_t base, WNode* wnodes) {
    const uint32_t i = blockIdx.x * BB + threadIdx.x;
    if (i >= n_nodes) return;
    const DNode nd = nodes[i];
    WNode w;
    const uint32_t lp = leaf_pos[i];
    if (lp == 0xFFFFFFFFu) {
        widen_dev(nodes[RT_REF_INDEX(nd.left) - base].bounds, w.lb);
        widen_dev(nodes[RT_REF_INDEX(nd.right) - base].bounds, w.rb);
    } else {
        widen_dev(items[(uint32_t)packed[lp]].box.b, w.lb);
        widen_dev(items[(uint32_t)packed[lp + (nd.left == nd.right ? 0u : 1u)]].box.b, w.rb);
    }
    w.left = nd.left; w.right = nd.right;
    w.meta = nd.left == nd.right ? 2u : 0u;
    w.qroot = 0;
    wnodes[i] = w;
}

} // namespace

// Builds the BVH over items[0, n) (none of them a constant medium, no NaN box) into nodes / wnodes (host arrays of
// bvh_node_count(n) entries) on the current HIP device.  Node references are offset by `base`.
int build_bvh_device(const BuildItem* items, uint64_t n64, uint64_t base, DNode* nodes, WNode* wnodes, std::string& err) {
    int rc = RT_OK;
    const uint32_t n = (uint32_t)n64;
    const uint32_t n_nodes = (uint32_t)node_count(n);
    uint32_t cap = 1;
    while (cap < n) cap <<= 1;           // span slots of the deepest level never exceed this
    BuildItem* d_items = nullptr;
    uint64_t *d_packed[2] = {nullptr, nullptr}, *d_keys[2] = {nullptr, nullptr}, *d_acc = nullptr;
    Span* d_spans[2] = {nullptr, nullptr};
    DNode* d_nodes = nullptr;
    WNode* d_wnodes = nullptr;
    uint32_t* d_leaf = nullptr;
    void* d_temp = nullptr;
    size_t temp_bytes = 0, tb = 0;
    const int grid_n = (int)((n + BB - 1) / BB);
    int cur = 0;

    BUILD_TRY(hipMalloc(&d_items, (size_t)n * sizeof(BuildItem)));
    BUILD_TRY(hipMemcpy(d_items, items, (size_t)n * sizeof(BuildItem), hipMemcpyHostToDevice));
    for (int k = 0; k < 2; ++k) {
        BUILD_TRY(hipMalloc(&d_packed[k], (size_t)n * sizeof(uint64_t)));
        BUILD_TRY(hipMalloc(&d_keys[k], (size_t)n * sizeof(uint64_t)));
        BUILD_TRY(hipMalloc(&d_spans[k], (size_t)2 * cap * sizeof(Span)));
    }
    BUILD_TRY(hipMalloc(&d_acc, (size_t)cap * 6 * sizeof(uint64_t)));
    BUILD_TRY(hipMalloc(&d_nodes, (size_t)n_nodes * sizeof(DNode)));
    BUILD_TRY(hipMalloc(&d_wnodes, (size_t)n_nodes * sizeof(WNode)));
    BUILD_TRY(hipMalloc(&d_leaf, (size_t)n_nodes * sizeof(uint32_t)));
    // temporary storage of the two sorts (sized for the larger request)
    BUILD_TRY(rocprim::radix_sort_pairs(nullptr, tb, d_keys[0], d_keys[1], d_packed[0], d_packed[1], (size_t)n, 0u, 64u, (hipStream_t)0));
    temp_bytes = tb;
    BUILD_TRY(rocprim::radix_sort_keys(nullptr, tb, d_packed[0], d_packed[1], (size_t)n, 32u, 64u, (hipStream_t)0));
    temp_bytes = tb > temp_bytes ? tb : temp_bytes;
    BUILD_TRY(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 256));

    hipLaunchKernelGGL(init_kernel, dim3(grid_n), dim3(BB), 0, 0, d_packed[0], n);
    {
        const Span root{0u, n, 0u, 0u};
        BUILD_TRY(hipMemcpy(d_spans[0], &root, sizeof(Span), hipMemcpyHostToDevice));
    }
    {
        uint32_t n_spans = 1;
        uint64_t largest = n; // the largest span of a level: ceil(n / 2^level)
        for (uint32_t level = 0;; ++level) {
            // 1. union boxes (accumulators start at +inf for minima, -inf for maxima: the orderable extremes)
            hipLaunchKernelGGL(acc_init_kernel, dim3((n_spans * 6 + BB - 1) / BB), dim3(BB), 0, 0, d_acc, n_spans * 6);
            hipLaunchKernelGGL(reduce_kernel, dim3(grid_n), dim3(BB), 0, 0, (const BuildItem*)d_items, (const uint64_t*)d_packed[cur], n, d_acc);
            // 2. nodes of this level and the spans of the next
            hipLaunchKernelGGL(node_kernel, dim3((n_spans + BB - 1) / BB), dim3(BB), 0, 0, d_spans[level & 1], n_spans, (const uint64_t*)d_acc,
                               (const BuildItem*)d_items, (const uint64_t*)d_packed[cur], d_nodes, d_leaf, base, d_spans[(level + 1) & 1]);
            if (largest <= 2) break; // every span of this level was a leaf
            // 3. keys, 4. the two stable sorts
            hipLaunchKernelGGL(key_kernel, dim3(grid_n), dim3(BB), 0, 0, (const BuildItem*)d_items, (const uint64_t*)d_packed[cur], n,
                               (const Span*)d_spans[level & 1], d_keys[0]);
            tb = temp_bytes;
            BUILD_TRY(rocprim::radix_sort_pairs(d_temp, tb, d_keys[0], d_keys[1], d_packed[cur], d_packed[cur ^ 1], (size_t)n, 0u, 64u, (hipStream_t)0));
            tb = temp_bytes;
            BUILD_TRY(rocprim::radix_sort_keys(d_temp, tb, d_packed[cur ^ 1], d_packed[cur], (size_t)n, 32u, 33u + level, (hipStream_t)0));
            hipLaunchKernelGGL(descend_kernel, dim3(grid_n), dim3(BB), 0, 0, d_packed[cur], n, (const Span*)d_spans[level & 1]);
            BUILD_TRY(hipGetLastError());
            n_spans *= 2;
            largest = (largest + 1) / 2;
        }
    }
    hipLaunchKernelGGL(wide_kernel, dim3((n_nodes + BB - 1) / BB), dim3(BB), 0, 0, (const DNode*)d_nodes, n_nodes, (const uint32_t*)d_leaf,
                       (const BuildItem*)d_items, (const uint64_t*)d_packed[cur], base, d_wnodes);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipMemcpy(nodes, d_nodes, (size_t)n_nodes * sizeof(DNode), hipMemcpyDeviceToHost));
    BUILD_TRY(hipMemcpy(wnodes, d_wnodes, (size_t)n_nodes * sizeof(WNode), hipMemcpyDeviceToHost));
done:
    (void)hipFree(d_items); (void)hipFree(d_packed[0]); (void)hipFree(d_packed[1]); (void)hipFree(d_keys[0]); (void)hipFree(d_keys[1]);
    (void)hipFree(d_spans[0]); (void)hipFree(d_spans[1]); (void)hipFree(d_acc); (void)hipFree(d_nodes); (void)hipFree(d_wnodes);
    (void)hipFree(d_leaf); (void)hipFree(d_temp);
    return rc;
}

} // namespace rt
