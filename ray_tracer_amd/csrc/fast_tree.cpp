// fast_tree.cpp -- traversal-only tree over the triangles of a mesh, for the nearest-first closest-hit kernel.
//
// The reference builds its BVH by sorting on the boxes' minima and cutting at the median (bvh.rs:20-77).  That tree
// is kept, node for node, for the traversals that must visit what the reference visits.  The nearest-first traversal
// only has to find the same closest hit, and the closest hit over a set of triangles does not depend on the tree over
// them, so it may walk a tree built for speed: surface-area heuristic over binned centroids, with spatial splits
// (Stich, Friedrich, Dietrich: "Spatial Splits in Bounding Volume Hierarchies", 2009) where the two halves of the best
// object split overlap.  A triangle that straddles a spatial split is referenced from both sides, each reference
// with the box of its part of the triangle; every point of a triangle lies in one of its references' boxes, and a
// reference is tested with the exact f64 intersector on the WHOLE triangle (triangle.rs:153-199), so a ray finds the
// triangle's hit through whichever reference box it enters first, at the same t, bit for bit.
//
// On the 2^20-triangle soup (triangles as large as their spacing) a ray makes 27.6 node steps and 5.7 triangle tests in
// this tree against 36.8 and 6.5 in the 4-wide collapse of the reference tree, at 1.30 references per triangle; on a
// closed surface, where boxes hardly overlap, the gain is the heuristic's alone (12.7 against 13.8 node steps).
//
// Three passes, each a recursion whose large subtrees run as tasks of their own: (1) the binary tree, partitioning the
// references in place (a spatial split copies its node's references once); (2) which binary nodes become 4-wide nodes
// and how many of those every subtree holds; (3) with those counts every 4-wide node knows where it and its children
// go, and writes its record in all the formats the kernels read (QNode, CQNode, PNode + PTri: device_types.h).
#include "fast_tree.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <memory>
#include <thread>

#include "host_parallel.h"
#include "quantise.h"

namespace rt {
namespace {

constexpr int BINS = 16;
constexpr size_t SPATIAL_MIN = 256;   // spatial splits are searched in nodes of at least this many references
constexpr size_t SWEEP_MAX = 8;       // nodes of at most this many references are split by an exact sweep along one axis
constexpr size_t TASK_MIN = 8192;     // subtrees of at least this many references may run as tasks of their own
constexpr size_t CHUNK_MIN = 1 << 17; // nodes of at least this many references bin their references in parallel chunks
constexpr float FINF = std::numeric_limits<float>::infinity();

struct FBox {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = FINF; hi[a] = -FINF; } }
    void grow(const FBox& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    float area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0.0f) || !(dy >= 0.0f) || !(dz >= 0.0f)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
};
inline FBox meet(const FBox& a, const FBox& b) {
    FBox r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = std::max(a.lo[k], b.lo[k]); r.hi[k] = std::min(a.hi[k], b.hi[k]); }
    return r;
}
// One cache line per reference: its box, its triangle, and the triangle's vertices in f32 for the search among
// candidate planes (which would otherwise fetch the 128-byte f64 triangle record, a cache miss per reference and level).
struct alignas(64) FRef { FBox b; uint32_t tri; float v[3][3]; };
static_assert(sizeof(FRef) == 64, "one line per reference");

inline float down32(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -FINF); return f; }
inline float up32(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, FINF); return f; }

struct BNode {
    FBox b;
    int32_t left, right;  // leaf: left < 0
    uint32_t tri;         // leaf: index into the builder's triangle list
    uint32_t leaves;      // references below this node
    uint32_t wide;        // pass 2: 4-wide nodes below this node when it is the root of one, itself included
};

struct ObjBins {
    FBox box[3][BINS];
    uint32_t cnt[3][BINS];
    void reset() { for (int a = 0; a < 3; ++a) for (int i = 0; i < BINS; ++i) { box[a][i].reset(); cnt[a][i] = 0; } }
    void merge(const ObjBins& o) { for (int a = 0; a < 3; ++a) for (int i = 0; i < BINS; ++i) { box[a][i].grow(o.box[a][i]); cnt[a][i] += o.cnt[a][i]; } }
};
struct SpaBins {
    FBox box[BINS];
    uint32_t en[BINS], ex[BINS];
    void reset() { for (int i = 0; i < BINS; ++i) { box[i].reset(); en[i] = ex[i] = 0; } }
    void merge(const SpaBins& o) { for (int i = 0; i < BINS; ++i) { box[i].grow(o.box[i]); en[i] += o.en[i]; ex[i] += o.ex[i]; } }
};

struct Builder {
    std::atomic<long long> ns_a{0}, ns_b{0}, ns_c{0}, ns_d{0}, ns_sweep{0}, ns_search{0};
    const DTri* tris = nullptr;
    const uint32_t* ids = nullptr; // triangle numbers in DScene::tris
    std::unique_ptr<BNode[]> pool;
    size_t pool_size = 0;
    std::atomic<uint32_t> next{0};
    std::atomic<int64_t> spare{0}; // references that spatial splits may still add
    std::atomic<int> tasks{0};
    int max_tasks = 1;
    float root_area = 1.0f;
    std::atomic<bool> overflow{false};

    // Runs fn(begin, end, part) over `parts` near-equal pieces of [0, n), the first on this thread.
    // The pieces are spread over this thread and as many more as the builder's budget of threads has left (`tasks`: the
    // builder never runs more than max_tasks + 1 threads at a time, however many nodes bin their references at once).
    template <class Fn>
    void chunks(size_t n, int parts, Fn fn) {
        const int helpers = take(parts - 1);
        for_parts(parts, helpers, [&](int p) { fn(n * (size_t)p / (size_t)parts, n * (size_t)(p + 1) / (size_t)parts, p); });
        tasks -= helpers;
    }
    // Reserves up to `want` threads of the budget; returns how many it got.
    int take(int want) {
        int cur = tasks.load(std::memory_order_relaxed);
        for (;;) {
            const int got = std::min(want, max_tasks - cur);
            if (got <= 0) return 0;
            if (tasks.compare_exchange_weak(cur, cur + got)) return got;
        }
    }
    int parts_for(size_t n) const { return n >= CHUNK_MIN ? std::max(1, std::min(max_tasks + 1, n >= 4 * CHUNK_MIN ? 16 : 8)) : 1; }

    // Box of the part of triangle t with lo <= x[axis] <= hi, rounded outward to f32 (the clipping arithmetic's own
    // rounding is covered by a relative pad far below the f32 step).
    FBox clip(uint32_t t, int axis, double lo, double hi) const {
        const DTri& T = tris[ids[t]];
        const double* v[3] = {T.v0, T.v1, T.v2};
        double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
        bool any = false;
        auto take = [&](const double* p) { any = true; for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], p[k]); mx[k] = std::max(mx[k], p[k]); } };
        for (int i = 0; i < 3; ++i) {
            const double* a = v[i];
            const double* b = v[(i + 1) % 3];
            if (a[axis] >= lo && a[axis] <= hi) take(a);
            for (int side = 0; side < 2; ++side) { // where the edge a-b crosses the two planes
                const double plane = side == 0 ? lo : hi;
                const double da = a[axis] - plane, db = b[axis] - plane;
                if ((da < 0.0 && db > 0.0) || (da > 0.0 && db < 0.0)) {
                    const double s = da / (da - db);
                    double p[3];
                    for (int k = 0; k < 3; ++k) p[k] = a[k] + s * (b[k] - a[k]);
                    p[axis] = plane;
                    take(p);
                }
            }
        }
        FBox r;
        if (!any) { r.reset(); return r; }
        for (int k = 0; k < 3; ++k) {
            const double pad = 1e-12 * (std::fabs(mn[k]) + std::fabs(mx[k]) + (mx[k] - mn[k])) + 1e-300;
            r.lo[k] = down32(mn[k] - pad);
            r.hi[k] = up32(mx[k] + pad);
        }
        return r;
    }

    // The same box in plain f32 arithmetic, for the search among candidate planes (where a box that is off by a rounding
    // only changes an estimate); the split that is finally made uses clip().
    static FBox clip_estimate(const FRef& ref, int axis, float lo, float hi) {
        const float (*v)[3] = ref.v;
        FBox r; r.reset();
        for (int i = 0; i < 3; ++i) {
            const float* a = v[i];
            const float* b = v[(i + 1) % 3];
            if (a[axis] >= lo && a[axis] <= hi)
                for (int k = 0; k < 3; ++k) { r.lo[k] = std::min(r.lo[k], a[k]); r.hi[k] = std::max(r.hi[k], a[k]); }
            const float da = a[axis], db = b[axis];
            if (da == db) continue;
            const float inv = 1.0f / (db - da);
            for (int side = 0; side < 2; ++side) {
                const float plane = side == 0 ? lo : hi;
                if ((da < plane && db > plane) || (da > plane && db < plane)) {
                    const float s = (plane - da) * inv;
                    for (int k = 0; k < 3; ++k) { const float p = k == axis ? plane : a[k] + s * (b[k] - a[k]); r.lo[k] = std::min(r.lo[k], p); r.hi[k] = std::max(r.hi[k], p); }
                }
            }
        }
        return r;
    }

    // Node numbers are handed out in blocks of NODE_BLOCK per thread: one shared counter incremented for each of three
    // million nodes by sixteen threads is a cache line that never rests.
    static constexpr uint32_t NODE_BLOCK = 512;
    static uint64_t next_serial() { static std::atomic<uint64_t> n{0}; return ++n; }
    const uint64_t serial = next_serial();
    uint32_t alloc() {
        thread_local uint64_t owner = 0; // (a builder's number, not its address: the next build may stand where this one stood)
        thread_local uint32_t at = 0, end = 0;
        if (owner != serial || at == end) {
            owner = serial;
            at = next.fetch_add(NODE_BLOCK);
            end = at + NODE_BLOCK;
        }
        const uint32_t i = at++;
        if (i >= pool_size) { overflow = true; return (uint32_t)pool_size - 1; }
        return i;
    }

    // Two subtrees, the first as a task of its own when it is large and a thread is free.
    template <class FnL, class FnR>
    void both(size_t n_left, FnL left, FnR right) {
        std::future<void> fut;
        if (n_left >= TASK_MIN && take(1) == 1) {
            if (try_async(fut, [&]() { left(); --tasks; })) {
                right();
                fut.get();
                return;
            }
            --tasks;
        }
        left();
        right();
    }

    // Up to four subtrees: every large one but the last as a task of its own while threads are free.
    template <class Size, class Fn>
    void fan(int n, Size size, Fn fn) {
        std::future<void> futs[3];
        int started = 0;
        for (int j = 0; j < n; ++j) {
            if (j + 1 < n && size(j) >= TASK_MIN && take(1) == 1) {
                if (try_async(futs[started], [this, &fn, j]() { fn(j); --tasks; })) { ++started; continue; }
                --tasks;
            }
            fn(j);
        }
        for (int k = 0; k < started; ++k) futs[k].get();
    }

    // ---------------------------------------------------------------------------------------------- pass 1: binary tree
    // refs[0, n) belong to this call and are permuted in place.
    uint32_t build(FRef* refs, size_t n, int depth) {
        static const bool prof = getenv("MI355RT_TREE_PROFILE") != nullptr;
        const auto t_node = std::chrono::steady_clock::now();
        const uint32_t me = alloc();
        BNode& node = pool[me];
        node.left = node.right = -1;
        node.tri = refs[0].tri;
        node.leaves = 1;
        node.wide = 0;
        if (n == 1 || overflow) { node.b = refs[0].b; return me; }
        const int parts = parts_for(n);
        FBox nb, cb; // box of the references, box of their centres
        nb.reset(); cb.reset();
        {
            std::vector<FBox> pn(parts), pc(parts);
            chunks(n, parts, [&](size_t lo, size_t hi, int p) {
                FBox a, c; a.reset(); c.reset();
                for (size_t i = lo; i < hi; ++i) {
                    a.grow(refs[i].b);
                    for (int k = 0; k < 3; ++k) { const float m = 0.5f * (refs[i].b.lo[k] + refs[i].b.hi[k]); c.lo[k] = std::min(c.lo[k], m); c.hi[k] = std::max(c.hi[k], m); }
                }
                pn[p] = a; pc[p] = c;
            });
            for (int p = 0; p < parts; ++p) { nb.grow(pn[p]); cb.grow(pc[p]); }
        }
        node.b = nb;
        const auto t_a = std::chrono::steady_clock::now();
        if (prof) ns_a += std::chrono::duration_cast<std::chrono::nanoseconds>(t_a - t_node).count();
        size_t n_left = 0;             // in-place result: refs[0, n_left) | refs[n_left, n)
        RawVec<FRef> copy;             // a spatial split's references: left part, then right part (written by the threads that fill it)
        size_t copy_left = 0;
        bool done = false;
        if (n > 2 && n <= SWEEP_MAX && depth < 60) {
            // few references: order them by centre along the longest axis of the centres' box and try every cut
            int a = 0;
            for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[a] - cb.lo[a]) a = k;
            if (cb.hi[a] - cb.lo[a] > 0.0f) {
                std::sort(refs, refs + n, [a](const FRef& x, const FRef& y) { return x.b.lo[a] + x.b.hi[a] < y.b.lo[a] + y.b.hi[a]; });
                FBox suffix[SWEEP_MAX];
                FBox acc; acc.reset();
                for (size_t i = n; i-- > 1;) { acc.grow(refs[i].b); suffix[i] = acc; }
                acc.reset();
                float best = FINF;
                for (size_t i = 1; i < n; ++i) {
                    acc.grow(refs[i - 1].b);
                    const float cost = acc.area() * (float)i + suffix[i].area() * (float)(n - i);
                    if (cost < best) { best = cost; n_left = i; }
                }
                done = n_left > 0;
            }
            if (prof) ns_sweep += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_a).count();
        } else if (n > SWEEP_MAX && depth < 60) {
            // object split: surface-area heuristic over BINS bins of the centres, all three axes in one pass
            float sc[3];
            for (int a = 0; a < 3; ++a) { const float ext = cb.hi[a] - cb.lo[a]; sc[a] = ext > 0.0f ? (float)BINS / ext : 0.0f; }
            auto bin_of = [&](const FRef& r, int a) {
                int b = (int)((0.5f * (r.b.lo[a] + r.b.hi[a]) - cb.lo[a]) * sc[a]);
                return b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
            };
            std::vector<ObjBins> ob(parts);
            chunks(n, parts, [&](size_t lo, size_t hi, int p) {
                ObjBins& o = ob[p];
                o.reset();
                for (size_t i = lo; i < hi; ++i)
                    for (int a = 0; a < 3; ++a) { const int b = bin_of(refs[i], a); o.box[a][b].grow(refs[i].b); ++o.cnt[a][b]; }
            });
            for (int p = 1; p < parts; ++p) ob[0].merge(ob[p]);
            const ObjBins& bins = ob[0];
            float best = FINF; int best_axis = -1, best_bin = 0; FBox best_l, best_r;
            for (int a = 0; a < 3; ++a) {
                if (!(sc[a] > 0.0f)) continue;
                FBox rb[BINS]; uint32_t rc[BINS];
                FBox acc; acc.reset(); uint32_t c = 0;
                for (int i = BINS - 1; i > 0; --i) { acc.grow(bins.box[a][i]); c += bins.cnt[a][i]; rb[i] = acc; rc[i] = c; }
                acc.reset(); c = 0;
                for (int i = 1; i < BINS; ++i) {
                    acc.grow(bins.box[a][i - 1]); c += bins.cnt[a][i - 1];
                    if (c == 0 || rc[i] == 0) continue;
                    const float cost = acc.area() * (float)c + rb[i].area() * (float)rc[i];
                    if (cost < best) { best = cost; best_axis = a; best_bin = i; best_l = acc; best_r = rb[i]; }
                }
            }
            const auto t_b = std::chrono::steady_clock::now();
            if (prof) ns_b += std::chrono::duration_cast<std::chrono::nanoseconds>(t_b - t_a).count();
            // Spatial split along the same axis, where the halves of the object split overlap.  Only in nodes of
            // SPATIAL_MIN references or more: measured on the 2^20-triangle soup, the splits that pay are those near the
            // top of the tree (with them in every node a ray makes 27.55 node steps, from 64 references on 27.60, from
            // 1024 on 28.2, without any 31.6), and near the leaves a reference spans many bins, which is where the
            // search's clipping time goes.
            if (n >= SPATIAL_MIN && best_axis >= 0 && spare.load(std::memory_order_relaxed) > 0 && meet(best_l, best_r).area() > 1e-6f * root_area) {
                const int a = best_axis;
                const float ext = nb.hi[a] - nb.lo[a];
                bool spatial = false; float s_pos = 0.0f;
                if (ext > 0.0f) {
                    const float ssc = (float)BINS / ext, w = ext / (float)BINS;
                    std::vector<SpaBins> sb(parts);
                    chunks(n, parts, [&](size_t lo, size_t hi, int p) {
                        SpaBins& s = sb[p];
                        s.reset();
                        for (size_t i = lo; i < hi; ++i) {
                            const FRef& r = refs[i];
                            int b0 = (int)((r.b.lo[a] - nb.lo[a]) * ssc), b1 = (int)((r.b.hi[a] - nb.lo[a]) * ssc);
                            b0 = b0 < 0 ? 0 : (b0 >= BINS ? BINS - 1 : b0);
                            b1 = b1 < b0 ? b0 : (b1 >= BINS ? BINS - 1 : b1);
                            ++s.en[b0]; ++s.ex[b1];
                            if (b0 == b1) { s.box[b0].grow(r.b); continue; }
                            for (int b = b0; b <= b1; ++b) {
                                const FBox c = meet(clip_estimate(r, a, nb.lo[a] + (float)b * w, nb.lo[a] + (float)(b + 1) * w), r.b);
                                if (c.valid()) s.box[b].grow(c);
                            }
                        }
                    });
                    for (int p = 1; p < parts; ++p) sb[0].merge(sb[p]);
                    const SpaBins& s = sb[0];
                    FBox rb[BINS]; uint32_t rc[BINS];
                    FBox acc; acc.reset(); uint32_t c = 0;
                    for (int i = BINS - 1; i > 0; --i) { acc.grow(s.box[i]); c += s.ex[i]; rb[i] = acc; rc[i] = c; }
                    acc.reset(); c = 0;
                    float s_cost = FINF;
                    for (int i = 1; i < BINS; ++i) {
                        acc.grow(s.box[i - 1]); c += s.en[i - 1];
                        if (c == 0 || rc[i] == 0) continue;
                        const float cost = acc.area() * (float)c + rb[i].area() * (float)rc[i];
                        if (cost < s_cost) { s_cost = cost; s_pos = nb.lo[a] + (float)i * w; }
                    }
                    spatial = s_cost < best;
                }
                if (prof) { const auto t_s = std::chrono::steady_clock::now(); ns_search += std::chrono::duration_cast<std::chrono::nanoseconds>(t_s - t_b).count(); }
                if (spatial) {
                    // every piece of the range sorts its references into a left and a right list; the lists are joined in
                    // the order of the pieces, i.e. in the order one thread would have produced
                    std::vector<std::vector<FRef>> lefts(parts), rights(parts);
                    std::vector<int64_t> addeds(parts, 0);
                    chunks(n, parts, [&](size_t lo, size_t hi, int p) {
                        std::vector<FRef>& left = lefts[p];
                        std::vector<FRef>& right = rights[p];
                        left.reserve((hi - lo) / 2 + (hi - lo) / 8); right.reserve((hi - lo) / 2 + (hi - lo) / 8);
                        int64_t added = 0;
                        for (size_t i = lo; i < hi; ++i) {
                            const FRef& r = refs[i];
                            if (r.b.hi[a] <= s_pos) left.push_back(r);
                            else if (r.b.lo[a] >= s_pos) right.push_back(r);
                            else {
                                FRef l = r, rr = r;
                                l.b = meet(clip(r.tri, a, -1e300, (double)s_pos), r.b);
                                rr.b = meet(clip(r.tri, a, (double)s_pos, 1e300), r.b);
                                const bool lv = l.b.valid(), rv = rr.b.valid();
                                if (lv) left.push_back(l);
                                if (rv) right.push_back(rr);
                                if (lv && rv) ++added;
                                if (!lv && !rv) left.push_back(r); // (cannot happen for a box that straddles the plane; never lose a reference)
                            }
                        }
                        addeds[p] = added;
                    });
                    size_t n_l = 0, n_r = 0;
                    int64_t added = 0;
                    for (int p = 0; p < parts; ++p) { n_l += lefts[p].size(); n_r += rights[p].size(); added += addeds[p]; }
                    if (n_l != 0 && n_r != 0 && n_l != n && n_r != n) {
                        copy.resize(n_l + n_r);
                        std::vector<size_t> at_l(parts), at_r(parts);
                        size_t al = 0, ar = n_l;
                        for (int p = 0; p < parts; ++p) { at_l[p] = al; at_r[p] = ar; al += lefts[p].size(); ar += rights[p].size(); }
                        chunks((size_t)parts, parts, [&](size_t lo, size_t hi, int) {
                            for (size_t p = lo; p < hi; ++p) {
                                std::copy(lefts[p].begin(), lefts[p].end(), copy.begin() + (std::ptrdiff_t)at_l[p]);
                                std::copy(rights[p].begin(), rights[p].end(), copy.begin() + (std::ptrdiff_t)at_r[p]);
                            }
                        });
                        copy_left = n_l;
                        spare.fetch_sub(added, std::memory_order_relaxed);
                        done = true;
                    }
                }
            }
            const auto t_c = std::chrono::steady_clock::now();
            if (prof) ns_c += std::chrono::duration_cast<std::chrono::nanoseconds>(t_c - t_b).count();
            if (!done && best_axis >= 0) {
                const int a = best_axis;
                n_left = (size_t)(std::partition(refs, refs + n, [&](const FRef& r) { return bin_of(r, a) < best_bin; }) - refs);
                done = n_left > 0 && n_left < n;
            }
            if (prof) ns_d += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_c).count();
        }
        if (!done) n_left = n / 2; // two references, all centres equal, or too deep: halve
        if (getenv("MI355RT_TREE_PROFILE") && n >= CHUNK_MIN) std::fprintf(stderr, "top node depth %d n %zu: %.4f s (%s)\n", depth, n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_node).count(), copy.empty() ? "object" : "spatial");
        FRef* lp = copy.empty() ? refs : copy.data();
        const size_t ln = copy.empty() ? n_left : copy_left, rn = copy.empty() ? n - n_left : copy.size() - copy_left;
        FRef* rp = lp + ln;
        uint32_t l = 0, r = 0;
        both(ln, [&]() { l = build(lp, ln, depth + 1); }, [&]() { r = build(rp, rn, depth + 1); });
        BNode& again = pool[me];
        again.left = (int32_t)l;
        again.right = (int32_t)r;
        again.leaves = pool[l].leaves + pool[r].leaves;
        return me;
    }

    // ------------------------------------------------------------------- pass 2: which nodes become 4-wide, and how many
    // The children of the 4-wide node rooted at binary node v: starting from v's two children, the inner child with the
    // largest box is replaced by its own two children until four are there.  Inner children first.
    int wide_children(uint32_t v, uint32_t kids[4]) const {
        int nk = 0;
        kids[nk++] = (uint32_t)pool[v].left;
        kids[nk++] = (uint32_t)pool[v].right;
        while (nk < 4) {
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < nk; ++i)
                if (pool[kids[i]].left >= 0 && pool[kids[i]].b.area() > ba) { ba = pool[kids[i]].b.area(); best = i; }
            if (best < 0) break;
            const uint32_t k = kids[best];
            kids[best] = (uint32_t)pool[k].left;
            kids[nk++] = (uint32_t)pool[k].right;
        }
        std::stable_partition(kids, kids + nk, [&](uint32_t k) { return pool[k].left >= 0; });
        return nk;
    }
    // Returns the stack entries a nearest-first traversal below v can need; sets pool[v].wide.
    int32_t count(uint32_t v) {
        uint32_t kids[4];
        const int nk = wide_children(v, kids);
        uint32_t wide = 1;
        int32_t deepest = 0;
        int32_t need[4] = {0, 0, 0, 0};
        int inner = 0;
        for (int j = 0; j < nk; ++j) inner += pool[kids[j]].left >= 0 ? 1 : 0;
        fan(inner, [&](int j) { return (size_t)pool[kids[j]].leaves; }, [&](int j) { need[j] = count(kids[j]); });
        for (int j = 0; j < inner; ++j) { wide += pool[kids[j]].wide; deepest = std::max(deepest, need[j]); }
        pool[v].wide = wide;
        return (nk - 1) + deepest; // up to nk - 1 siblings stay deferred while one child is walked
    }

    // -------------------------------------------------------------------------------------------- pass 3: the records
    QNode* qn = nullptr; CQNode* cq = nullptr; PNode* pn = nullptr; PTri* pt = nullptr; uint8_t* qd = nullptr;
    uint32_t q_first = 0;     // index of the tree's root in the scene's qnodes / cqnodes
    bool packed = false;
    // 4-wide node rooted at binary node v: record `at` (0 = root), its inner children at records [kids_at, ...), its
    // triangles at ptris [leaves_at, ...); everything below the children follows behind them, subtree by subtree.
    void emit(uint32_t v, uint32_t at, uint32_t kids_at, uint32_t leaves_at, int depth) {
        uint32_t kids[4];
        const int nk = wide_children(v, kids);
        int inner = 0;
        for (int j = 0; j < nk; ++j) inner += pool[kids[j]].left >= 0 ? 1 : 0;
        QNode q{};
        for (int j = 0; j < 4; ++j) {
            q._pad[j] = 0;
            if (j >= nk) {
                q.child[j] = RT_REF_NONE;
                for (int a = 0; a < 3; ++a) { q.lo[a][j] = FINF; q.hi[a][j] = -FINF; }
                continue;
            }
            const BNode& c = pool[kids[j]];
            for (int a = 0; a < 3; ++a) { q.lo[a][j] = c.b.lo[a]; q.hi[a][j] = c.b.hi[a]; }
            q.child[j] = j < inner ? make_ref(REF_QNODE, q_first + kids_at + (uint32_t)j) : make_ref(RT_REF_TRI, ids[c.tri]);
        }
        qn[at] = q;
        qd[at] = (uint8_t)std::min(depth, 255);
        CQNode c8{};
        quantise_node(q, c8);
        cq[at] = c8;
        if (packed) {
            PNode p{};
            for (int a = 0; a < 3; ++a) { p.origin[a] = c8.origin[a]; p.exp[a] = c8.exp[a]; }
            p.counts = (uint8_t)(inner | (nk << 4));
            for (int k = 0; k < 4; ++k)
                for (int a = 0; a < 3; ++a) { p.qlo[a][k] = c8.qlo[a][k]; p.qhi[a][k] = c8.qhi[a][k]; }
            p.node_base = make_ref(REF_QNODE, kids_at);
            p.leaf_base = make_ref(RT_REF_TRI, leaves_at) - (uint32_t)inner;
            pn[at] = p;
            for (int j = inner; j < nk; ++j) {
                const DTri& T = tris[ids[pool[kids[j]].tri]];
                PTri t{};
                for (int a = 0; a < 3; ++a) { t.v0[a] = T.v0[a]; t.v1[a] = T.v1[a]; t.v2[a] = T.v2[a]; }
                t.orig = ids[pool[kids[j]].tri];
                pt[leaves_at + (uint32_t)(j - inner)] = t;
            }
        }
        // the children's own children and triangles follow, subtree by subtree
        uint32_t below_at[4], below_leaves[4];
        uint32_t next_at = kids_at + (uint32_t)inner, next_leaves = leaves_at + (uint32_t)(nk - inner);
        for (int j = 0; j < inner; ++j) {
            below_at[j] = next_at; below_leaves[j] = next_leaves;
            next_at += pool[kids[j]].wide - 1u;
            next_leaves += pool[kids[j]].leaves;
        }
        fan(inner, [&](int j) { return (size_t)pool[kids[j]].leaves; },
            [&](int j) { emit(kids[j], kids_at + (uint32_t)j, below_at[j], below_leaves[j], depth + 1); });
    }
};

} // namespace

int build_fast_tree(const DTri* tris, FlatScene& f, std::vector<uint8_t>& qdepth, const uint32_t* tri_ids, uint64_t n, FastTreeResult& out, std::string& err) {
    if (n < 2 || n >= (1ull << 27)) { err = "fast tree: triangle count out of range"; return RT_E_TOO_LARGE; }
    const bool debug = getenv("MI355RT_DEBUG") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    Builder B;
    B.tris = tris;
    B.ids = tri_ids;
    const uint64_t extra = n;                       // spatial splits may add up to one more reference per triangle
    B.pool_size = 2 * (n + extra) + 64 + (size_t)Builder::NODE_BLOCK * 4096; // (blocks of node numbers that threads leave half used)
    B.pool.reset(new BNode[B.pool_size]);
    B.spare = (int64_t)extra;
    B.max_tasks = (int)std::min<unsigned>(15u, std::max(1u, std::thread::hardware_concurrency()) - 1u);
    if (const char* e = getenv("MI355RT_TREE_THREADS")) B.max_tasks = std::max(0, std::min(255, atoi(e) - 1)); // (measurements)
    std::unique_ptr<FRef[]> refs(new FRef[n]);
    {
        const int parts = B.parts_for(n);
        std::vector<FBox> roots(parts);
        std::atomic<bool> bad{false};
        B.chunks(n, parts, [&](size_t lo, size_t hi, int p) {
            FBox root; root.reset();
            for (size_t i = lo; i < hi; ++i) {
                const DTri& T = B.tris[tri_ids[i]];
                FRef& r = refs[i];
                r.tri = (uint32_t)i;
                for (int k = 0; k < 3; ++k) { r.v[0][k] = (float)T.v0[k]; r.v[1][k] = (float)T.v1[k]; r.v[2][k] = (float)T.v2[k]; }
                for (int k = 0; k < 3; ++k) {
                    double l = std::fmin(std::fmin(T.v0[k], T.v1[k]), T.v2[k]), h = std::fmax(std::fmax(T.v0[k], T.v1[k]), T.v2[k]);
                    if (h - l < 0.0001) { l -= 0.00005; h += 0.00005; } // as the reference pads thin boxes (aabb.rs:125-136)
                    r.b.lo[k] = down32(l); r.b.hi[k] = up32(h);
                }
                if (!r.b.valid() || !std::isfinite(r.b.area())) bad = true;
                root.grow(r.b);
            }
            roots[p] = root;
        });
        // (coordinates that are not finite, or whose boxes overflow f32: not this builder's to judge -- the scene keeps the
        // collapse of the reference's tree, and whatever the reference's own rules say about such a triangle applies there)
        if (bad) { err = "fast tree: a triangle's box is not finite in f32"; return RT_E_UNSUPPORTED; }
        FBox root; root.reset();
        for (const FBox& b : roots) root.grow(b);
        B.root_area = std::max(root.area(), 1e-30f);
    }
    const auto t_1 = std::chrono::steady_clock::now();
    const uint32_t broot = B.build(refs.get(), n, 0);
    if (B.overflow) { err = "fast tree: node pool exhausted"; return RT_E_TOO_LARGE; }
    refs.reset();
    const auto t_2 = std::chrono::steady_clock::now();
    out.stack_need = B.count(broot);
    const auto t_2a = std::chrono::steady_clock::now();
    const uint64_t wide = B.pool[broot].wide, leaves = B.pool[broot].leaves;
    const size_t first = f.qnodes.size();
    if (first + wide >= (1ull << 28)) { err = "fast tree: more than 2^28 wide nodes"; return RT_E_TOO_LARGE; }
    // (the kernel addresses the packed arrays by 32-bit byte offsets: they stay below 4 GB; larger meshes keep the 64-byte nodes)
    B.packed = (wide + 4u) * sizeof(PNode) < (1ull << 32) && (leaves + 4u) * sizeof(PTri) < (1ull << 32);
    f.qnodes.reserve(first + wide + n);             // room for the collapses of the reference's tree behind this one (address space only)
    f.qnodes.resize(first + wide);
    f.cqnodes.reserve(first + wide + n);
    f.cqnodes.resize(first + wide);
    qdepth.resize(first + wide);
    f.pnodes.clear(); f.ptris.clear();
    if (B.packed) { f.pnodes.resize(wide); f.ptris.resize(leaves); }
    B.qn = f.qnodes.data() + first; B.cq = f.cqnodes.data() + first; B.qd = qdepth.data() + first;
    B.pn = f.pnodes.data(); B.pt = f.ptris.data();
    B.q_first = (uint32_t)first;
    const auto t_2b = std::chrono::steady_clock::now();
    B.emit(broot, 0u, 1u, 0u, 0);
    const auto t_2c = std::chrono::steady_clock::now();
    out.root = (uint32_t)first;
    out.wide_nodes = wide;
    out.references = leaves;
    // the measures the flattener keeps: area of the leaf boxes, and how much the 8-bit grid inflates the boxes near the root
    out.leaf_area = 0.0; out.ratio_sum = 0.0; out.ratio_n = 0;
    for (uint64_t i = 0; i < wide; ++i) {
        const QNode& q = B.qn[i];
        for (int j = 0; j < 4; ++j) {
            if (q.child[j] == RT_REF_NONE) continue;
            if (RT_REF_KIND(q.child[j]) == RT_REF_TRI) {
                const double dx = (double)q.hi[0][j] - q.lo[0][j], dy = (double)q.hi[1][j] - q.lo[1][j], dz = (double)q.hi[2][j] - q.lo[2][j];
                out.leaf_area += dx * dy + dy * dz + dz * dx;
            }
            if (B.qd[i] < 4) { const double r = quantised_area_ratio(q, B.cq[i], j); if (r >= 0.0) { out.ratio_sum += r; ++out.ratio_n; } }
        }
    }
    if (debug) {
        const auto t_3 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mi355rt] traversal-only tree: %llu triangles, %llu references, %llu wide nodes; set-up %.3f s, binary tree %.3f s, records %.3f s (%d tasks)\n",
                     (unsigned long long)n, (unsigned long long)leaves, (unsigned long long)wide, std::chrono::duration<double>(t_1 - t_0).count(),
                     std::chrono::duration<double>(t_2 - t_1).count(), std::chrono::duration<double>(t_3 - t_2).count(), B.max_tasks);
        if (getenv("MI355RT_TREE_PROFILE")) std::fprintf(stderr, "[mi355rt]   binary tree, thread-seconds: bounds %.3f, object bins %.3f, spatial search+apply %.3f (search %.3f), partition %.3f, sweeps %.3f\n",
                                           B.ns_a * 1e-9, B.ns_b * 1e-9, B.ns_c * 1e-9, B.ns_search * 1e-9, B.ns_d * 1e-9, B.ns_sweep * 1e-9);
        std::fprintf(stderr, "[mi355rt]   records: count %.3f s, arrays %.3f s, emit %.3f s, measures %.3f s\n", std::chrono::duration<double>(t_2a - t_2).count(),
                     std::chrono::duration<double>(t_2b - t_2a).count(), std::chrono::duration<double>(t_2c - t_2b).count(), std::chrono::duration<double>(t_3 - t_2c).count());
    }
    return RT_OK;
}

} // namespace rt
