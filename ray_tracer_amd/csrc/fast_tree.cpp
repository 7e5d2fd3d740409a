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
// On the 2^20-triangle soup (triangles as large as their spacing) this tree costs a ray 22 % fewer node steps and 12 %
// fewer triangle tests than the 4-wide collapse of the reference tree, at 1.4 references per triangle; on a closed
// surface, where boxes hardly overlap, the gain is the heuristic's alone (8 % fewer node steps).
#include "fast_tree.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <future>
#include <limits>
#include <thread>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace rt {
namespace {

constexpr int BINS = 16;
constexpr size_t SPATIAL_MIN = 64;
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
struct FRef { FBox b; uint32_t tri; };

inline float down32(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -FINF); return f; }
inline float up32(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, FINF); return f; }

struct BNode { FBox b; int32_t left, right; uint32_t tri; }; // leaf: left < 0

struct Builder {
    const DTri* tris;
    const uint32_t* ids;        // triangle numbers in DScene::tris
    std::vector<BNode> pool;
    std::atomic<uint32_t> next{0};
    std::atomic<int64_t> spare; // references that spatial splits may still add
    std::atomic<int> tasks{0};
    int max_tasks = 1;
    float root_area = 1.0f;
    bool overflow = false;

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

    uint32_t alloc() {
        const uint32_t i = next.fetch_add(1);
        if (i >= pool.size()) { overflow = true; return (uint32_t)pool.size() - 1; }
        return i;
    }

    uint32_t build(std::vector<FRef>& refs, int depth) {
        const uint32_t me = alloc();
        FBox nb; nb.reset();
        for (const FRef& r : refs) nb.grow(r.b);
        pool[me].b = nb;
        pool[me].left = pool[me].right = -1;
        pool[me].tri = refs[0].tri;
        const size_t n = refs.size();
        if (n == 1 || overflow) return me;
        std::vector<FRef> L, R;
        bool done = false;
        if (n > 2 && depth < 60) {
            // object split: surface-area heuristic over BINS centroid bins, all three axes in one pass
            FBox cb; cb.reset();
            for (const FRef& r : refs)
                for (int k = 0; k < 3; ++k) { const float c = 0.5f * (r.b.lo[k] + r.b.hi[k]); cb.lo[k] = std::min(cb.lo[k], c); cb.hi[k] = std::max(cb.hi[k], c); }
            FBox bins[3][BINS];
            uint32_t cnt[3][BINS];
            float sc[3];
            for (int a = 0; a < 3; ++a) {
                const float ext = cb.hi[a] - cb.lo[a];
                sc[a] = ext > 0.0f ? (float)BINS / ext : 0.0f;
                for (int i = 0; i < BINS; ++i) { bins[a][i].reset(); cnt[a][i] = 0; }
            }
            for (const FRef& r : refs)
                for (int a = 0; a < 3; ++a) {
                    int b = (int)((0.5f * (r.b.lo[a] + r.b.hi[a]) - cb.lo[a]) * sc[a]);
                    b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    bins[a][b].grow(r.b);
                    ++cnt[a][b];
                }
            float best = FINF; int best_axis = -1, best_bin = 0; FBox best_l, best_r;
            for (int a = 0; a < 3; ++a) {
                if (!(sc[a] > 0.0f)) continue;
                FBox rb[BINS]; uint32_t rc[BINS];
                FBox acc; acc.reset(); uint32_t c = 0;
                for (int i = BINS - 1; i > 0; --i) { acc.grow(bins[a][i]); c += cnt[a][i]; rb[i] = acc; rc[i] = c; }
                acc.reset(); c = 0;
                for (int i = 1; i < BINS; ++i) {
                    acc.grow(bins[a][i - 1]); c += cnt[a][i - 1];
                    if (c == 0 || rc[i] == 0) continue;
                    const float cost = acc.area() * (float)c + rb[i].area() * (float)rc[i];
                    if (cost < best) { best = cost; best_axis = a; best_bin = i; best_l = acc; best_r = rb[i]; }
                }
            }
            // spatial split along the same axis, where the halves of the object split overlap
            bool spatial = false; float s_pos = 0.0f;
            // (only in nodes of SPATIAL_MIN references or more: measured on the 2^20-triangle soup, the splits that pay are
            // those near the top of the tree -- with them in every node a ray makes 27.55 node steps, from 64 references
            // on 27.60, from 1024 on 28.2, without any 31.6 -- and near the leaves a reference spans many bins, so the
            // search there is where most of its clipping time goes)
            if (n >= SPATIAL_MIN && best_axis >= 0 && spare.load(std::memory_order_relaxed) > 0 && meet(best_l, best_r).area() > 1e-6f * root_area) {
                const int a = best_axis;
                const float ext = nb.hi[a] - nb.lo[a];
                if (ext > 0.0f) {
                    FBox sb[BINS]; uint32_t en[BINS], ex[BINS];
                    for (int i = 0; i < BINS; ++i) { sb[i].reset(); en[i] = ex[i] = 0; }
                    const float ssc = (float)BINS / ext, w = ext / (float)BINS;
                    for (const FRef& r : refs) {
                        int b0 = (int)((r.b.lo[a] - nb.lo[a]) * ssc), b1 = (int)((r.b.hi[a] - nb.lo[a]) * ssc);
                        b0 = b0 < 0 ? 0 : (b0 >= BINS ? BINS - 1 : b0);
                        b1 = b1 < b0 ? b0 : (b1 >= BINS ? BINS - 1 : b1);
                        ++en[b0]; ++ex[b1];
                        if (b0 == b1) { sb[b0].grow(r.b); continue; }
                        for (int b = b0; b <= b1; ++b) {
                            const FBox c = meet(clip(r.tri, a, (double)(nb.lo[a] + (float)b * w), (double)(nb.lo[a] + (float)(b + 1) * w)), r.b);
                            if (c.valid()) sb[b].grow(c);
                        }
                    }
                    FBox rb[BINS]; uint32_t rc[BINS];
                    FBox acc; acc.reset(); uint32_t c = 0;
                    for (int i = BINS - 1; i > 0; --i) { acc.grow(sb[i]); c += ex[i]; rb[i] = acc; rc[i] = c; }
                    acc.reset(); c = 0;
                    float s_cost = FINF;
                    for (int i = 1; i < BINS; ++i) {
                        acc.grow(sb[i - 1]); c += en[i - 1];
                        if (c == 0 || rc[i] == 0) continue;
                        const float cost = acc.area() * (float)c + rb[i].area() * (float)rc[i];
                        if (cost < s_cost) { s_cost = cost; s_pos = nb.lo[a] + (float)i * w; }
                    }
                    spatial = s_cost < best;
                }
                if (spatial) {
                    L.reserve(n); R.reserve(n);
                    int64_t added = 0;
                    for (const FRef& r : refs) {
                        if (r.b.hi[a] <= s_pos) L.push_back(r);
                        else if (r.b.lo[a] >= s_pos) R.push_back(r);
                        else {
                            FRef l = r, rr = r;
                            l.b = meet(clip(r.tri, a, -1e300, (double)s_pos), r.b);
                            rr.b = meet(clip(r.tri, a, (double)s_pos, 1e300), r.b);
                            const bool lv = l.b.valid(), rv = rr.b.valid();
                            if (lv) L.push_back(l);
                            if (rv) R.push_back(rr);
                            if (lv && rv) ++added;
                            if (!lv && !rv) L.push_back(r); // (cannot happen for a box that straddles the plane; never lose a reference)
                        }
                    }
                    if (L.empty() || R.empty() || L.size() == n || R.size() == n) { L.clear(); R.clear(); spatial = false; }
                    else { spare.fetch_sub(added, std::memory_order_relaxed); done = true; }
                }
            }
            if (!done && best_axis >= 0) {
                L.reserve(n); R.reserve(n);
                for (const FRef& r : refs) {
                    int b = (int)((0.5f * (r.b.lo[best_axis] + r.b.hi[best_axis]) - cb.lo[best_axis]) * sc[best_axis]);
                    b = b < 0 ? 0 : (b >= BINS ? BINS - 1 : b);
                    (b < best_bin ? L : R).push_back(r);
                }
                done = !L.empty() && !R.empty();
                if (!done) { L.clear(); R.clear(); }
            }
        }
        if (!done) { // two references, all centroids equal, or too deep: halve
            L.assign(refs.begin(), refs.begin() + n / 2);
            R.assign(refs.begin() + n / 2, refs.end());
        }
        std::vector<FRef>().swap(refs);
        uint32_t l, r;
        if (n > 16384 && tasks.load() < max_tasks) {
            ++tasks;
            auto fut = std::async(std::launch::async, [&]() { const uint32_t x = build(L, depth + 1); --tasks; return x; });
            r = build(R, depth + 1);
            l = fut.get();
        } else {
            l = build(L, depth + 1);
            r = build(R, depth + 1);
        }
        pool[me].left = (int32_t)l;
        pool[me].right = (int32_t)r;
        return me;
    }
};

} // namespace

int build_fast_tree(const DTri* tris, const uint32_t* tri_ids, uint64_t n, std::vector<QNode>& qnodes, std::vector<uint8_t>& qdepth,
                    uint32_t& root_out, int32_t& stack_need, std::string& err) {
    if (n < 2 || n >= (1ull << 27)) { err = "fast tree: triangle count out of range"; return RT_E_TOO_LARGE; }
    Builder B;
    B.tris = tris;
    B.ids = tri_ids;
    const uint64_t extra = n;                       // spatial splits may add up to one more reference per triangle
    B.pool.resize(2 * (n + extra) + 64);
    B.spare = (int64_t)extra;
    B.max_tasks = (int)std::min<unsigned>(15u, std::max(1u, std::thread::hardware_concurrency()) - 1u);
    std::vector<FRef> refs(n);
    FBox root; root.reset();
    for (uint64_t i = 0; i < n; ++i) {
        const DTri& T = tris[tri_ids[i]];
        FRef& r = refs[i];
        r.tri = (uint32_t)i;
        for (int k = 0; k < 3; ++k) {
            double lo = std::fmin(std::fmin(T.v0[k], T.v1[k]), T.v2[k]), hi = std::fmax(std::fmax(T.v0[k], T.v1[k]), T.v2[k]);
            if (hi - lo < 0.0001) { lo -= 0.00005; hi += 0.00005; } // as the reference pads thin boxes (aabb.rs:125-136)
            r.b.lo[k] = down32(lo); r.b.hi[k] = up32(hi);
        }
        if (!r.b.valid() || !std::isfinite(r.b.area())) { err = "fast tree: non-finite triangle"; return RT_E_INVALID; }
        root.grow(r.b);
    }
    B.root_area = std::max(root.area(), 1e-30f);
    const bool debug = getenv("MI355RT_DEBUG") != nullptr;
    const auto t_0 = std::chrono::steady_clock::now();
    const uint32_t broot = B.build(refs, 0);
    const auto t_1 = std::chrono::steady_clock::now();
    if (B.overflow) { err = "fast tree: node pool exhausted"; return RT_E_TOO_LARGE; }

    // Collapse into 4-wide nodes: starting from a node's two children, the inner child with the largest box is
    // replaced by its own two children until four are there.  Inner children of a node get consecutive indices
    // starting at an even one (siblings share 128-byte lines in the 64-byte format).
    struct Job { uint32_t bnode, qnode; int depth; };
    auto add_empty = [&]() {
        QNode q{};
        for (int j = 0; j < 4; ++j) {
            q.child[j] = RT_REF_NONE;
            for (int a = 0; a < 3; ++a) { q.lo[a][j] = FINF; q.hi[a][j] = -FINF; }
        }
        qnodes.push_back(q);
        qdepth.push_back(255);
    };
    if (qnodes.size() & 1u) add_empty();
    root_out = (uint32_t)qnodes.size();
    add_empty();
    std::vector<Job> todo{Job{broot, root_out, 0}};
    std::vector<std::pair<uint32_t, uint32_t>> parent_of; // (qnode, parent qnode) in creation order, for the stack need
    parent_of.emplace_back(root_out, RT_REF_NONE);
    while (!todo.empty()) {
        const Job job = todo.back(); todo.pop_back();
        uint32_t kids[4]; int nk = 0;
        kids[nk++] = (uint32_t)B.pool[job.bnode].left;
        kids[nk++] = (uint32_t)B.pool[job.bnode].right;
        while (nk < 4) {
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < nk; ++i)
                if (B.pool[kids[i]].left >= 0 && B.pool[kids[i]].b.area() > ba) { ba = B.pool[kids[i]].b.area(); best = i; }
            if (best < 0) break;
            const uint32_t k = kids[best];
            kids[best] = (uint32_t)B.pool[k].left;
            kids[nk++] = (uint32_t)B.pool[k].right;
        }
        int n_inner = 0;
        for (int i = 0; i < nk; ++i) n_inner += B.pool[kids[i]].left >= 0 ? 1 : 0;
        uint32_t base = 0;
        if (n_inner) {
            if (qnodes.size() & 1u) add_empty();
            base = (uint32_t)qnodes.size();
            for (int i = 0; i < n_inner; ++i) add_empty();
        }
        if (qnodes.size() >= (1ull << 28)) { err = "fast tree: more than 2^28 wide nodes"; return RT_E_TOO_LARGE; }
        QNode q{};
        int k = 0;
        for (int j = 0; j < 4; ++j) {
            q._pad[j] = 0;
            if (j >= nk) {
                q.child[j] = RT_REF_NONE;
                for (int a = 0; a < 3; ++a) { q.lo[a][j] = FINF; q.hi[a][j] = -FINF; }
                continue;
            }
            const BNode& c = B.pool[kids[j]];
            for (int a = 0; a < 3; ++a) { q.lo[a][j] = c.b.lo[a]; q.hi[a][j] = c.b.hi[a]; }
            if (c.left >= 0) {
                const uint32_t child = base + (uint32_t)k++;
                q.child[j] = make_ref(REF_QNODE, child);
                todo.push_back(Job{kids[j], child, job.depth + 1});
                parent_of.emplace_back(child, job.qnode);
            } else {
                q.child[j] = make_ref(RT_REF_TRI, tri_ids[c.tri]);
            }
        }
        qnodes[job.qnode] = q;
        qdepth[job.qnode] = (uint8_t)std::min(job.depth, 255);
    }
    // stack entries a traversal below the root can need: up to (children - 1) siblings stay deferred per level
    std::vector<int32_t> need(qnodes.size() - root_out, 0);
    for (size_t i = parent_of.size(); i-- > 0;) { // children were created after their parents: walk backwards
        const uint32_t qn = parent_of[i].first, par = parent_of[i].second;
        int cnt = 0;
        for (int j = 0; j < 4; ++j) cnt += qnodes[qn].child[j] != RT_REF_NONE ? 1 : 0;
        const int32_t mine = std::max(0, cnt - 1) + need[qn - root_out]; // need[] holds the deepest child's need so far
        need[qn - root_out] = mine;
        if (par != RT_REF_NONE) need[par - root_out] = std::max(need[par - root_out], mine);
    }
    stack_need = need[0];
    if (debug)
        std::fprintf(stderr, "[mi355rt] traversal-only tree: %llu triangles, %u binary nodes, %zu wide nodes, build %.3f s, collapse %.3f s\n", (unsigned long long)n,
                     B.next.load(), qnodes.size() - root_out, std::chrono::duration<double>(t_1 - t_0).count(),
                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_1).count());
    return RT_OK;
}

} // namespace rt
