// tree_lab.cpp -- host-only experiment: how many wide-node steps and triangle tests per ray does an ordered
// (nearest child first, entry-distance culling) closest-hit traversal need on different trees over the same triangles?
//
//   g++ -O2 -std=c++17 -fopenmp profiles/tree_lab.cpp -o /tmp/lab/tree_lab
//   /tmp/lab/tree_lab tris.bin rays.bin <builder> [width] [alpha] [collapse]
// tris.bin: N x 9 doubles, rays.bin: M x 6 doubles (origin, direction).
// builder: median (bbox-min median split on the longest axis, the reference's BvhNode::from_objects, bvh.rs:20-77),
//          sah (binned surface-area heuristic over centroids), sbvh (the same with spatial splits).
// Exact f64 boxes; no quantisation.  Prints the mean number of wide-node visits and triangle tests per ray.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <limits>
#include <string>
#include <vector>

static const double INF = std::numeric_limits<double>::infinity();
struct Box {
    double lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = INF; hi[a] = -INF; } }
    void grow(const Box& b) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    void grow(const double* p) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    double area() const {
        const double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0;
        return dx * dy + dy * dz + dz * dx;
    }
    bool valid() const { return lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]; }
};
static Box isect(const Box& a, const Box& b) {
    Box r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = std::max(a.lo[k], b.lo[k]); r.hi[k] = std::min(a.hi[k], b.hi[k]); }
    return r;
}
struct Ref { Box b; uint32_t tri; };

static std::vector<double> g_tris; // 9 per triangle
static const double* tri(uint32_t i) { return &g_tris[(size_t)i * 9]; }

// bounding box of triangle `t` clipped to lo <= x[axis] <= hi
static Box clip_tri(uint32_t t, int axis, double lo, double hi) {
    double poly[2][8][3];
    int n = 3, cur = 0;
    for (int v = 0; v < 3; ++v) for (int k = 0; k < 3; ++k) poly[0][v][k] = tri(t)[3 * v + k];
    for (int side = 0; side < 2; ++side) {
        const double plane = side == 0 ? lo : hi, sgn = side == 0 ? 1.0 : -1.0;
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const double* a = poly[cur][i];
            const double* b = poly[cur][(i + 1) % n];
            const double da = sgn * (a[axis] - plane), db = sgn * (b[axis] - plane);
            if (da >= 0) { memcpy(poly[cur ^ 1][m++], a, 24); }
            if ((da > 0 && db < 0) || (da < 0 && db > 0)) {
                const double s = da / (da - db);
                for (int k = 0; k < 3; ++k) poly[cur ^ 1][m][k] = a[k] + s * (b[k] - a[k]);
                poly[cur ^ 1][m][axis] = plane;
                ++m;
            }
        }
        n = m; cur ^= 1;
        if (n == 0) break;
    }
    Box r; r.reset();
    for (int i = 0; i < n; ++i) r.grow(poly[cur][i]);
    return r;
}

struct BNode { Box b; int32_t left = -1, right = -1; uint32_t tri = 0; }; // leaf: left < 0
static std::vector<BNode> g_nodes;
static double g_alpha = 1e-5;
static size_t g_ref_budget = 0, g_refs_made = 0;
static double g_root_area = 1.0;
static int g_mode = 0; // 0 median, 1 sah, 2 sbvh
constexpr int BINS = 32;

static int build(std::vector<Ref>& refs, int depth) {
    const int me = (int)g_nodes.size();
    g_nodes.emplace_back();
    Box nb; nb.reset();
    for (auto& r : refs) nb.grow(r.b);
    g_nodes[me].b = nb;
    if (refs.size() == 1) { g_nodes[me].tri = refs[0].tri; return me; }
    std::vector<Ref> L, R;
    if (g_mode == 0) {
        const double sx = nb.hi[0] - nb.lo[0], sy = nb.hi[1] - nb.lo[1], sz = nb.hi[2] - nb.lo[2];
        const int axis = sx > sy ? (sx > sz ? 0 : 2) : (sy > sz ? 1 : 2);
        if (refs.size() > 2) std::stable_sort(refs.begin(), refs.end(), [axis](const Ref& a, const Ref& b) { return a.b.lo[axis] < b.b.lo[axis]; });
        const size_t mid = refs.size() / 2;
        L.assign(refs.begin(), refs.begin() + mid);
        R.assign(refs.begin() + mid, refs.end());
    } else {
        // object split: binned SAH over centroids
        Box cb; cb.reset();
        for (auto& r : refs) { double c[3]; for (int k = 0; k < 3; ++k) c[k] = 0.5 * (r.b.lo[k] + r.b.hi[k]); cb.grow(c); }
        double best_cost = INF; int best_axis = -1, best_bin = -1; Box best_lb, best_rb;
        for (int axis = 0; axis < 3; ++axis) {
            const double ext = cb.hi[axis] - cb.lo[axis];
            if (!(ext > 0)) continue;
            Box bb[BINS]; uint32_t cnt[BINS] = {};
            for (auto& b : bb) b.reset();
            const double sc = BINS / ext;
            for (auto& r : refs) {
                int b = (int)((0.5 * (r.b.lo[axis] + r.b.hi[axis]) - cb.lo[axis]) * sc);
                b = std::min(BINS - 1, std::max(0, b));
                bb[b].grow(r.b); ++cnt[b];
            }
            Box rb[BINS]; uint32_t rc[BINS];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int i = BINS - 1; i > 0; --i) { acc.grow(bb[i]); c += cnt[i]; rb[i] = acc; rc[i] = c; }
            acc.reset(); c = 0;
            for (int i = 1; i < BINS; ++i) {
                acc.grow(bb[i - 1]); c += cnt[i - 1];
                if (c == 0 || rc[i] == 0) continue;
                const double cost = acc.area() * c + rb[i].area() * rc[i];
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = i; best_lb = acc; best_rb = rb[i]; }
            }
        }
        bool spatial = false; int s_axis = -1; double s_pos = 0; double s_cost = INF;
        static const size_t spatial_max = getenv("SPATIAL_MAX_REFS") ? (size_t)atoll(getenv("SPATIAL_MAX_REFS")) : (size_t)-1;
        static const int one_axis = getenv("SPATIAL_ONE_AXIS") ? atoi(getenv("SPATIAL_ONE_AXIS")) : 0;
        static const size_t spatial_min = getenv("SPATIAL_MIN_REFS") ? (size_t)atoll(getenv("SPATIAL_MIN_REFS")) : 0;
        if (g_mode == 2 && best_axis >= 0 && g_refs_made < g_ref_budget && refs.size() <= spatial_max && refs.size() >= spatial_min) {
            const double ov = isect(best_lb, best_rb).area();
            if (ov / g_root_area > g_alpha) {
                for (int axis = 0; axis < 3; ++axis) {
                    const double ext = nb.hi[axis] - nb.lo[axis];
                    if (!(ext > 0)) continue;
                    if (one_axis && axis != best_axis) continue;
                    Box bb[BINS]; uint32_t en[BINS] = {}, ex[BINS] = {};
                    for (auto& b : bb) b.reset();
                    const double sc = BINS / ext, w = ext / BINS;
                    for (auto& r : refs) {
                        int b0 = std::min(BINS - 1, std::max(0, (int)((r.b.lo[axis] - nb.lo[axis]) * sc)));
                        int b1 = std::min(BINS - 1, std::max(0, (int)((r.b.hi[axis] - nb.lo[axis]) * sc)));
                        ++en[b0]; ++ex[b1];
                        if (b0 == b1) { bb[b0].grow(r.b); continue; }
                        for (int b = b0; b <= b1; ++b) {
                            Box c = isect(clip_tri(r.tri, axis, nb.lo[axis] + b * w, nb.lo[axis] + (b + 1) * w), r.b);
                            if (c.valid()) bb[b].grow(c);
                        }
                    }
                    Box rb[BINS]; uint32_t rc[BINS];
                    Box acc; acc.reset(); uint32_t c = 0;
                    for (int i = BINS - 1; i > 0; --i) { acc.grow(bb[i]); c += ex[i]; rb[i] = acc; rc[i] = c; }
                    acc.reset(); c = 0;
                    for (int i = 1; i < BINS; ++i) {
                        acc.grow(bb[i - 1]); c += en[i - 1];
                        if (c == 0 || rc[i] == 0) continue;
                        const double cost = acc.area() * c + rb[i].area() * rc[i];
                        if (cost < s_cost) { s_cost = cost; s_axis = axis; s_pos = nb.lo[axis] + i * w; }
                    }
                }
                if (s_cost < best_cost) spatial = true;
            }
        }
        if (spatial) {
            for (auto& r : refs) {
                if (r.b.hi[s_axis] <= s_pos) L.push_back(r);
                else if (r.b.lo[s_axis] >= s_pos) R.push_back(r);
                else {
                    Ref a = r, b = r;
                    a.b = isect(clip_tri(r.tri, s_axis, -INF, s_pos), r.b);
                    b.b = isect(clip_tri(r.tri, s_axis, s_pos, INF), r.b);
                    if (a.b.valid()) L.push_back(a);
                    if (b.b.valid()) R.push_back(b);
                    if (a.b.valid() && b.b.valid()) ++g_refs_made;
                }
            }
            if (L.empty() || R.empty() || L.size() == refs.size() || R.size() == refs.size()) { spatial = false; L.clear(); R.clear(); }
        }
        if (!spatial) {
            if (best_axis < 0) { // all centroids equal: halve
                const size_t mid = refs.size() / 2;
                L.assign(refs.begin(), refs.begin() + mid);
                R.assign(refs.begin() + mid, refs.end());
            } else {
                const double ext = cb.hi[best_axis] - cb.lo[best_axis], sc = BINS / ext;
                for (auto& r : refs) {
                    int b = (int)((0.5 * (r.b.lo[best_axis] + r.b.hi[best_axis]) - cb.lo[best_axis]) * sc);
                    b = std::min(BINS - 1, std::max(0, b));
                    (b < best_bin ? L : R).push_back(r);
                }
            }
        }
    }
    std::vector<Ref>().swap(refs);
    const int l = build(L, depth + 1);
    const int r = build(R, depth + 1);
    g_nodes[me].left = l; g_nodes[me].right = r;
    return me;
}

// wide tree
struct WChild { Box b; int32_t node; uint32_t tri; }; // node >= 0: inner wide node, else triangle
struct WNode { int n = 0; WChild c[8]; };
static std::vector<WNode> g_wide;
static int g_width = 4;
static int g_collapse = 0; // 0: fixed levels (two binary levels for width 4, three for 8), 1: open the largest child first

static int collapse(int bn) {
    const int me = (int)g_wide.size();
    g_wide.emplace_back();
    std::vector<int> kids{g_nodes[bn].left, g_nodes[bn].right};
    if (g_collapse == 0) {
        const int levels = g_width == 4 ? 1 : (g_width == 8 ? 2 : 0);
        for (int lv = 0; lv < levels; ++lv) {
            std::vector<int> next;
            for (int k : kids) { if (g_nodes[k].left >= 0) { next.push_back(g_nodes[k].left); next.push_back(g_nodes[k].right); } else next.push_back(k); }
            kids.swap(next);
        }
    } else {
        while ((int)kids.size() < g_width) {
            int best = -1; double ba = -1;
            for (int i = 0; i < (int)kids.size(); ++i) if (g_nodes[kids[i]].left >= 0 && g_nodes[kids[i]].b.area() > ba) { ba = g_nodes[kids[i]].b.area(); best = i; }
            if (best < 0) break;
            const int k = kids[best];
            kids[best] = g_nodes[k].left;
            kids.push_back(g_nodes[k].right);
        }
    }
    WNode w; w.n = (int)kids.size();
    for (int i = 0; i < w.n; ++i) {
        w.c[i].b = g_nodes[kids[i]].b;
        if (g_nodes[kids[i]].left >= 0) { w.c[i].node = collapse(kids[i]); w.c[i].tri = 0; }
        else { w.c[i].node = -1; w.c[i].tri = g_nodes[kids[i]].tri; }
    }
    g_wide[me] = w;
    return me;
}

static bool tri_test(const double* T, const double* o, const double* d, double tmin, double tmax, double& t_out) { // triangle.rs:153-199
    const double e1[3] = {T[3] - T[0], T[4] - T[1], T[5] - T[2]}, e2[3] = {T[6] - T[0], T[7] - T[1], T[8] - T[2]};
    const double h[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const double a = e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2];
    if (std::fabs(a) < 1e-8) return false;
    const double f = 1.0 / a;
    const double s[3] = {o[0] - T[0], o[1] - T[1], o[2] - T[2]};
    const double u = f * (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]);
    if (u < 0.0 || u > 1.0) return false;
    const double q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
    const double v = f * (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]);
    if (v < 0.0 || u + v > 1.0) return false;
    const double t = f * (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]);
    if (t < tmin || t > tmax) return false;
    t_out = t;
    return true;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: tree_lab tris.bin rays.bin median|sah|sbvh [width] [alpha] [collapse] [budget]\n"); return 2; }
    auto load = [](const char* p, std::vector<double>& v) {
        FILE* f = fopen(p, "rb"); if (!f) { perror(p); exit(1); }
        fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
        v.resize(sz / 8); if (fread(v.data(), 8, v.size(), f) != v.size()) exit(1); fclose(f);
    };
    std::vector<double> rays;
    load(argv[1], g_tris); load(argv[2], rays);
    const std::string mode = argv[3];
    g_mode = mode == "median" ? 0 : (mode == "sah" ? 1 : 2);
    g_width = argc > 4 ? atoi(argv[4]) : 4;
    g_alpha = argc > 5 ? atof(argv[5]) : 1e-5;
    g_collapse = argc > 6 ? atoi(argv[6]) : 0;
    const double budget = argc > 7 ? atof(argv[7]) : 3.0;
    const size_t N = g_tris.size() / 9, M = rays.size() / 6;
    g_ref_budget = (size_t)(budget * N);
    std::vector<Ref> refs(N);
    Box root; root.reset();
    for (size_t i = 0; i < N; ++i) {
        refs[i].tri = (uint32_t)i; refs[i].b.reset();
        for (int v = 0; v < 3; ++v) refs[i].b.grow(tri((uint32_t)i) + 3 * v);
        for (int a = 0; a < 3; ++a) if (refs[i].b.hi[a] - refs[i].b.lo[a] < 1e-4) { refs[i].b.lo[a] -= 5e-5; refs[i].b.hi[a] += 5e-5; } // aabb.rs:125-136
        root.grow(refs[i].b);
    }
    g_root_area = root.area();
    if (const char* e = getenv("PRESPLIT")) { // early split clipping: halve a reference along its longest axis while its box is larger than the threshold
        const double thr = atof(e);
        std::vector<Ref> out; out.reserve(refs.size() * 4);
        std::vector<Ref> work = refs;
        while (!work.empty()) {
            Ref r = work.back(); work.pop_back();
            const double ex[3] = {r.b.hi[0] - r.b.lo[0], r.b.hi[1] - r.b.lo[1], r.b.hi[2] - r.b.lo[2]};
            const int ax = ex[0] > ex[1] ? (ex[0] > ex[2] ? 0 : 2) : (ex[1] > ex[2] ? 1 : 2);
            if (ex[ax] <= thr) { out.push_back(r); continue; }
            const double mid = 0.5 * (r.b.lo[ax] + r.b.hi[ax]);
            Ref a = r, b = r;
            a.b = isect(clip_tri(r.tri, ax, -INF, mid), r.b);
            b.b = isect(clip_tri(r.tri, ax, mid, INF), r.b);
            if (!a.b.valid() || !b.b.valid()) { out.push_back(r); continue; }
            work.push_back(a); work.push_back(b);
        }
        refs.swap(out);
        fprintf(stderr, "presplit at %g: %zu references (x%.2f)\n", thr, refs.size(), (double)refs.size() / N);
    }
    g_nodes.reserve(8 * N);
    { const clock_t c0 = clock(); build(refs, 0); fprintf(stderr, "build: %.1f s (one thread)\n", (double)(clock() - c0) / CLOCKS_PER_SEC); }
    size_t leaves = 0; double leaf_area = 0;
    for (auto& n : g_nodes) if (n.left < 0) { ++leaves; leaf_area += n.b.area(); }
    collapse(0);
    // SAH cost of the wide tree: sum over wide nodes of area (node steps) and over leaves (tests), relative to the root
    fprintf(stderr, "%s width %d: %zu triangles, %zu leaves (x%.2f), %zu binary nodes, %zu wide nodes, leaf overlap %.1f\n", mode.c_str(), g_width, N, leaves,
            (double)leaves / N, g_nodes.size(), g_wide.size(), leaf_area / g_root_area);
    double sum_nodes = 0, sum_tris = 0, sum_boxes = 0; uint64_t max_nodes = 0; size_t hits = 0;
    std::vector<uint32_t> hist(64, 0);
#pragma omp parallel for reduction(+ : sum_nodes, sum_tris, sum_boxes, hits) reduction(max : max_nodes) schedule(dynamic, 256)
    for (long ri = 0; ri < (long)M; ++ri) {
        const double* o = &rays[ri * 6]; const double* d = o + 3;
        const double inv[3] = {1.0 / d[0], 1.0 / d[1], 1.0 / d[2]};
        struct E { int32_t node; uint32_t tri; double tn; };
        std::vector<E> st; st.reserve(128);
        double best = INF; bool hit = false;
        uint64_t nn = 0, nt = 0, nb = 0;
        st.push_back(E{0, 0, 0.0});
        while (!st.empty()) {
            const E e = st.back(); st.pop_back();
            if (e.tn > best) continue;
            if (e.node < 0) { ++nt; double t; if (tri_test(tri(e.tri), o, d, 0.001, best, t)) { best = t; hit = true; } continue; }
            ++nn;
            const WNode& w = g_wide[e.node];
            E kids[8]; int nk = 0;
            for (int i = 0; i < w.n; ++i) {
                ++nb;
                double tn = 0.001, tf = best; bool ok = true;
                for (int a = 0; a < 3; ++a) {
                    double t0 = (w.c[i].b.lo[a] - o[a]) * inv[a], t1 = (w.c[i].b.hi[a] - o[a]) * inv[a];
                    if (t0 > t1) std::swap(t0, t1);
                    if (t0 > tn) tn = t0;
                    if (t1 < tf) tf = t1;
                    if (tf < tn) { ok = false; break; }
                }
                if (ok) kids[nk++] = E{w.c[i].node, w.c[i].tri, tn};
            }
            std::sort(kids, kids + nk, [](const E& a, const E& b) { return a.tn > b.tn; }); // farthest first onto the stack
            for (int i = 0; i < nk; ++i) st.push_back(kids[i]);
        }
        sum_nodes += nn; sum_tris += nt; sum_boxes += nb; hits += hit ? 1 : 0; max_nodes = std::max<uint64_t>(max_nodes, nn + nt);
    }
    printf("%s w%d alpha %g collapse %d: node visits/ray %.2f, box tests/ray %.1f, triangle tests/ray %.2f, hit fraction %.3f, longest ray %llu steps, refs x%.2f\n",
           mode.c_str(), g_width, g_alpha, g_collapse, sum_nodes / M, sum_boxes / M, sum_tris / M, (double)hits / M, (unsigned long long)max_nodes, (double)leaves / N);
    return 0;
}
