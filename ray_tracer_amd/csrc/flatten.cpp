// flatten.cpp -- host flattener (no HIP in this file).
//
// Turns the declarative object graph of rt_scene_desc into the linear arrays the
// kernels traverse.  What it must reproduce from the reference so that closest-hit
// results can be compared one-to-one:
//   - bounding boxes of every Hittable: sphere.rs:25-45, quad.rs:51-61,
//     triangle.rs:127-139, hittable_list.rs:37-40, hittable.rs:67-76,104-143,
//     constant_medium.rs:106-108, obj_loader.rs:103-105, with the padding rule of
//     aabb.rs:125-136 applied exactly where the reference applies it;
//   - the BVH split order of BvhNode::from_objects (bvh.rs:20-77): bbox = union over
//     the span, axis = Aabb::longest_axis (aabb.rs:109-123), STABLE sort on
//     bbox.axis(axis).min, mid = start + span/2, span 1 -> both children are the same
//     object, span 2 -> the pair in incoming order;
//   - the precomputed quad / triangle constants (quad.rs:28-49, triangle.rs:122-125).
// Nodes of one BVH are laid out in pre-order (left child = index + 1).
#include "flatten.h"
#include "fast_tree.h"
#include "host_parallel.h"
#include "quantise.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <thread>
#include <limits>
#include <map>
#include <numeric>

namespace rt {
namespace {

const double INF = std::numeric_limits<double>::infinity();
const double PI = 3.1415926535897932385; // rtweekend.rs:7

struct V3 { double x, y, z; };
inline V3 v3(const double* p) { return V3{p[0], p[1], p[2]}; }
inline V3 add(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 sub(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 scale(double t, V3 v) { return V3{t * v.x, t * v.y, t * v.z}; }
inline double dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double len3(V3 a) { return std::sqrt(dot3(a, a)); }
inline V3 divs(V3 v, double t) { return scale(1.0 / t, v); } // v / t == v * (1/t)
inline V3 unit3(V3 v) { return divs(v, len3(v)); }
inline void put(double* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

inline Box box_empty() { return Box{{INF, -INF, INF, -INF, INF, -INF}}; } // interval.rs:12-15
inline void pad_axis(double& lo, double& hi) {                             // aabb.rs:125-136, interval.rs:75-81
    const double delta = 0.0001;
    if (hi - lo < delta) { double p = delta / 2.0; lo = lo - p; hi = hi + p; }
}
inline Box box_from_points(V3 a, V3 b) { // aabb.rs:43-63
    Box r;
    if (a.x <= b.x) { r.b[0] = a.x; r.b[1] = b.x; } else { r.b[0] = b.x; r.b[1] = a.x; }
    if (a.y <= b.y) { r.b[2] = a.y; r.b[3] = b.y; } else { r.b[2] = b.y; r.b[3] = a.y; }
    if (a.z <= b.z) { r.b[4] = a.z; r.b[5] = b.z; } else { r.b[4] = b.z; r.b[5] = a.z; }
    pad_axis(r.b[0], r.b[1]); pad_axis(r.b[2], r.b[3]); pad_axis(r.b[4], r.b[5]);
    return r;
}
inline Box box_union(const Box& a, const Box& b) { // aabb.rs:65-71, interval.rs:43-48
    Box r;
    for (int k = 0; k < 3; ++k) {
        r.b[2 * k] = std::fmin(a.b[2 * k], b.b[2 * k]);
        r.b[2 * k + 1] = std::fmax(a.b[2 * k + 1], b.b[2 * k + 1]);
    }
    return r;
}
inline int longest_axis(const Box& a) { // aabb.rs:109-123
    double sx = a.b[1] - a.b[0], sy = a.b[3] - a.b[2], sz = a.b[5] - a.b[4];
    if (sx > sy) return sx > sz ? 0 : 2;
    return sy > sz ? 1 : 2;
}
inline bool box_has_nan(const Box& a) {
    for (double v : a.b) if (v != v) return true;
    return false;
}

typedef BuildItem Item;

// f64 -> f32 rounded away from the box interior, so the f32 box contains the f64 box.
inline float f32_down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; }
inline float f32_up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; }
inline void widen(const double* b, float* o) {
    for (int k = 0; k < 3; ++k) { o[2 * k] = f32_down(b[2 * k]); o[2 * k + 1] = f32_up(b[2 * k + 1]); }
}

uint64_t bvh_node_count(uint64_t n) { // N(1)=N(2)=1, N(n)=1+N(n/2)+N(n-n/2)
    if (n <= 2) return 1;
    return 1 + bvh_node_count(n / 2) + bvh_node_count(n - n / 2);
}

// Builds the subtree for items[start,end) into nodes[at ...]; returns has-medium flag.
uint32_t build_bvh(std::vector<Item>& items, size_t start, size_t end, DNode* nodes, WNode* wnodes, uint64_t base, uint64_t at, int depth) {
    size_t span = end - start;
    Box bbox = box_empty();
    for (size_t i = start; i < end; ++i) bbox = box_union(bbox, items[i].box);
    int axis = longest_axis(bbox);
    DNode& n = nodes[at];
    for (int k = 0; k < 6; ++k) n.bounds[k] = bbox.b[k];
    n._pad = 0;
    uint32_t medium = 0;
    WNode& wn = wnodes[at];
    wn.qroot = 0;
    if (span == 1) {
        n.left = n.right = items[start].ref;
        medium = items[start].medium;
        widen(items[start].box.b, wn.lb);
        widen(items[start].box.b, wn.rb);
    } else if (span == 2) {
        n.left = items[start].ref;
        n.right = items[start + 1].ref;
        medium = items[start].medium | items[start + 1].medium;
        widen(items[start].box.b, wn.lb);
        widen(items[start + 1].box.b, wn.rb);
    } else {
        // Stable sort by the minimum of the node's longest axis (bvh.rs:47-58, comparators bvh.rs:79-115).  The keys
        // are pulled out next to the item numbers so the sort runs over 16-byte records, and a large span is sorted
        // in slices on several threads and merged: both keep equal keys in their incoming order.
        struct Key { double k; uint32_t i; };
        std::vector<Key> order(span);
        const Item* base_items = items.data() + start;
        for (size_t i = 0; i < span; ++i) order[i] = Key{base_items[i].box.b[2 * axis], (uint32_t)i};
        auto less = [](const Key& a, const Key& b) { return a.k < b.k; };
        const size_t slices = (depth < 3 && span > (1u << 16)) ? (size_t)(8 >> depth) : 1;
        if (slices > 1) {
            std::vector<size_t> cut(slices + 1);
            for (size_t s = 0; s <= slices; ++s) cut[s] = span * s / slices;
            for_parts((int)slices, (int)slices - 1, [&](int s) { std::stable_sort(order.begin() + cut[s], order.begin() + cut[s + 1], less); });
            for (size_t width = 1; width < slices; width *= 2)
                for (size_t s = 0; s + width < slices; s += 2 * width)
                    std::inplace_merge(order.begin() + cut[s], order.begin() + cut[s + width], order.begin() + cut[std::min(s + 2 * width, slices)], less);
        } else {
            std::stable_sort(order.begin(), order.end(), less);
        }
        {
            std::vector<Item> sorted(span);
            for (size_t i = 0; i < span; ++i) sorted[i] = base_items[order[i].i];
            std::copy(sorted.begin(), sorted.end(), items.begin() + start);
        }
        size_t mid = start + span / 2;
        uint64_t left_at = at + 1;
        uint64_t right_at = left_at + bvh_node_count(mid - start);
        n.left = make_ref(RT_REF_NODE, (uint32_t)(base + left_at));
        n.right = make_ref(RT_REF_NODE, (uint32_t)(base + right_at));
        uint32_t ml, mr;
        if (depth < 4 && span > 8192) { // the two halves are independent (bvh.rs:64-67 uses rayon::join)
            std::future<uint32_t> fut;
            const bool started = try_async(fut, [&]() { return build_bvh(items, start, mid, nodes, wnodes, base, left_at, depth + 1); });
            mr = build_bvh(items, mid, end, nodes, wnodes, base, right_at, depth + 1);
            ml = started ? fut.get() : build_bvh(items, start, mid, nodes, wnodes, base, left_at, depth + 1);
        } else {
            ml = build_bvh(items, start, mid, nodes, wnodes, base, left_at, depth + 1);
            mr = build_bvh(items, mid, end, nodes, wnodes, base, right_at, depth + 1);
        }
        medium = ml | mr;
        widen(nodes[left_at].bounds, wn.lb);
        widen(nodes[right_at].bounds, wn.rb);
    }
    n.flags = (medium ? 1u : 0u) | ((uint32_t)axis << 1); // bits 1-2: the axis this node's objects were sorted along (bvh.rs:32)
    wn.left = n.left; wn.right = n.right;
    wn.meta = (medium ? 1u : 0u) | (n.left == n.right ? 2u : 0u);
    return medium;
}

struct Flattener {
    const rt_scene_desc* d;
    FlatScene& f;
    uint64_t device_bvh_min = 0;
    uint64_t fast_tree_min = 0; // > 0: a primitive BVH over at least this many triangles gets a traversal-only tree (fast_tree.cpp)
    // BvhNode::new over `items` into nodes[base ...]: on the device when the BVH is large enough and holds no medium
    int build_nodes(std::vector<Item>& items, uint64_t base) {
        bool medium = false;
        for (const Item& it : items) medium = medium || it.medium;
        if (device_bvh_min && items.size() >= device_bvh_min && !medium) {
            std::string e;
            const int rc = build_bvh_device(items.data(), items.size(), base, f.nodes.data() + base, f.wnodes.data() + base, e);
            if (rc) return fail(rc, "device BVH build: " + e);
            return RT_OK;
        }
        build_bvh(items, 0, items.size(), f.nodes.data() + base, f.wnodes.data() + base, base, 0, 0);
        return RT_OK;
    }
    std::string& err;
    std::map<std::pair<int32_t, int32_t>, uint32_t> memo; // (object, ctx) -> ref
    std::vector<uint8_t> obj_medium;                     // object subtree contains a medium
    std::vector<int32_t> need_node;
    std::map<uint32_t, int32_t> need_q;                    // BVH root node index -> stack entries its 4-wide copy needs
    std::vector<uint32_t> bvh_roots;                       // node indices of all BVH roots, in creation order
    std::map<uint32_t, bool> bvh_prim_only;               // BVH root ref -> every leaf is a sphere/quad/triangle

    Flattener(const rt_scene_desc* desc, FlatScene& out, std::string& e) : d(desc), f(out), err(e) {}

    int fail(int code, const std::string& m) { err = m; return code; }

    int check_textures() {
        for (uint64_t i = 0; i < d->n_textures; ++i) {
            const rt_texture& t = d->textures[i];
            DTexture o{};
            o.kind = t.kind; o.even = t.even; o.odd = t.odd; o.image = t.image; o.perlin = t.perlin;
            o.color[0] = t.color[0]; o.color[1] = t.color[1]; o.color[2] = t.color[2];
            o.scale = t.scale;
            switch (t.kind) {
            case RT_TEX_SOLID: break;
            case RT_TEX_CHECKER:
                if (t.even < 0 || t.odd < 0 || (uint64_t)t.even >= d->n_textures || (uint64_t)t.odd >= d->n_textures)
                    return fail(RT_E_INVALID, "checker texture: child texture out of range");
                // A checker only nests textures created before it (Arc ownership makes cycles impossible in the reference).
                if ((uint64_t)t.even >= i || (uint64_t)t.odd >= i) return fail(RT_E_INVALID, "checker texture: children must precede it");
                o.scale = 1.0 / t.scale; // inv_scale (texture.rs:43)
                break;
            case RT_TEX_IMAGE:
                f.uses_uv = true;
                if (t.image >= 0) {
                    if ((uint64_t)t.image >= d->n_images) return fail(RT_E_INVALID, "image texture: image out of range");
                    const rt_image& im = d->images[t.image];
                    if (im.width < 0 || im.height < 0 || im.offset + (uint64_t)im.width * (uint64_t)im.height * 3 > d->n_image_bytes)
                        return fail(RT_E_INVALID, "image texture: pixel data out of range");
                }
                break;
            case RT_TEX_NOISE:
                if (t.perlin < 0 || (uint64_t)t.perlin >= d->n_perlins) return fail(RT_E_INVALID, "noise texture: perlin table out of range");
                break;
            default: return fail(RT_E_INVALID, "texture: unknown kind");
            }
            f.textures.push_back(o);
        }
        for (uint64_t i = 0; i < d->n_perlins; ++i)
            for (int k = 0; k < 256; ++k) {
                const rt_perlin& p = d->perlins[i];
                if ((p.perm_x[k] | p.perm_y[k] | p.perm_z[k]) & ~255) return fail(RT_E_INVALID, "perlin: permutation entry outside 0..255");
            }
        return RT_OK;
    }

    bool texture_uses_uv(int32_t tex) const {
        const rt_texture& t = d->textures[tex];
        if (t.kind == RT_TEX_IMAGE) return true;
        if (t.kind == RT_TEX_CHECKER) return texture_uses_uv(t.even) || texture_uses_uv(t.odd);
        return false;
    }

    int check_materials() {
        for (uint64_t i = 0; i < d->n_materials; ++i) {
            const rt_material& m = d->materials[i];
            DMaterial o{};
            o.kind = m.kind; o.tex = m.tex;
            o.albedo[0] = m.albedo[0]; o.albedo[1] = m.albedo[1]; o.albedo[2] = m.albedo[2];
            o.fuzz = std::fmin(m.fuzz, 1.0); // material.rs:127-132
            o.ior = m.ior;
            switch (m.kind) {
            case RT_MAT_LAMBERTIAN: case RT_MAT_DIFFUSE_LIGHT: case RT_MAT_ISOTROPIC:
                if (m.tex < 0 || (uint64_t)m.tex >= d->n_textures) return fail(RT_E_INVALID, "material: texture out of range");
                if (d->textures[m.tex].kind == RT_TEX_SOLID) { // the colour rides in the material record: one dependent fetch less per shaded hit
                    const rt_texture& t = d->textures[m.tex];
                    o.albedo[0] = t.color[0]; o.albedo[1] = t.color[1]; o.albedo[2] = t.color[2];
                    o.tex = -1;
                } else f.has_textures = true;
                break;
            case RT_MAT_METAL: case RT_MAT_DIELECTRIC: break;
            default: return fail(RT_E_INVALID, "material: unknown kind");
            }
            f.materials.push_back(o);
        }
        return RT_OK;
    }

    bool mat_ok(int32_t m) const { return m >= 0 && (uint64_t)m < d->n_materials; }

    int build_primitives() {
        f.spheres.resize(d->n_spheres);
        for (uint64_t i = 0; i < d->n_spheres; ++i) {
            const rt_sphere& s = d->spheres[i];
            if (!mat_ok(s.mat)) return fail(RT_E_INVALID, "sphere: material out of range");
            DSphere& o = f.spheres[i];
            V3 c1 = v3(s.center);
            put(o.c, c1);
            if (s.moving) put(o.dc, sub(v3(s.center2), c1)); // sphere.rs:40
            else put(o.dc, V3{0.0, 0.0, 0.0});               // sphere.rs:28
            o.radius = std::fmax(s.radius, 0.0);             // sphere.rs:29
            o.mat = s.mat; o._pad = 0;
        }
        f.quads.resize(d->n_quads);
        for (uint64_t i = 0; i < d->n_quads; ++i) { // quad.rs:28-49
            const rt_quad& s = d->quads[i];
            if (!mat_ok(s.mat)) return fail(RT_E_INVALID, "quad: material out of range");
            DQuad& o = f.quads[i];
            V3 q = v3(s.q), u = v3(s.u), v = v3(s.v);
            V3 n = cross3(u, v);
            V3 normal = unit3(n);
            put(o.q, q); put(o.u, u); put(o.v, v);
            put(o.n, normal);
            o.d = dot3(normal, q);
            put(o.w, divs(n, dot3(n, n)));
            o.area = len3(n);
            o.mat = s.mat; o._pad = 0;
        }
        f.tris.resize(d->n_triangles); // (left uninitialised: every record is written below, by the thread that owns its part)
        const uint64_t nt = d->n_triangles;
        const int parts = nt >= (1u << 16) ? (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())) : 1;
        std::atomic<bool> bad_mat{false};
        auto fill = [&](uint64_t lo, uint64_t hi) {
            for (uint64_t i = lo; i < hi; ++i) { // triangle.rs:122-125
                const rt_triangle& s = d->triangles[i];
                if (!mat_ok(s.mat)) { bad_mat = true; return; }
                DTri& o = f.tris[i];
                V3 v0 = v3(s.v0), v1 = v3(s.v1), v2 = v3(s.v2);
                put(o.v0, v0); put(o.v1, v1); put(o.v2, v2);
                put(o.n, unit3(cross3(sub(v1, v0), sub(v2, v0))));
                o.mat = s.mat; o._pad0 = 0;
                o._pad[0] = o._pad[1] = o._pad[2] = 0.0;
            }
        };
        for_parts(parts, parts - 1, [&](int p) { fill(nt * (uint64_t)p / (uint64_t)parts, nt * (uint64_t)(p + 1) / (uint64_t)parts); });
        if (bad_mat) return fail(RT_E_INVALID, "triangle: material out of range");
        return RT_OK;
    }

    Box sphere_box(uint64_t i) const { // sphere.rs:26-44
        const rt_sphere& s = d->spheres[i];
        V3 rvec{s.radius, s.radius, s.radius};
        V3 c1 = v3(s.center);
        Box b1 = box_from_points(sub(c1, rvec), add(c1, rvec));
        if (!s.moving) return b1;
        V3 c2 = v3(s.center2);
        return box_union(b1, box_from_points(sub(c2, rvec), add(c2, rvec)));
    }
    Box quad_box(uint64_t i) const { // quad.rs:51-61
        const rt_quad& s = d->quads[i];
        V3 q = v3(s.q), u = v3(s.u), v = v3(s.v);
        V3 p1 = q, p2 = add(q, u), p3 = add(q, v), p4 = add(add(q, u), v);
        return box_union(box_from_points(p1, p4), box_from_points(p2, p3));
    }
    Box tri_box(uint64_t i) const { // triangle.rs:127-139
        const rt_triangle& s = d->triangles[i];
        V3 mn{std::fmin(std::fmin(s.v0[0], s.v1[0]), s.v2[0]), std::fmin(std::fmin(s.v0[1], s.v1[1]), s.v2[1]), std::fmin(std::fmin(s.v0[2], s.v1[2]), s.v2[2])};
        V3 mx{std::fmax(std::fmax(s.v0[0], s.v1[0]), s.v2[0]), std::fmax(std::fmax(s.v0[1], s.v1[1]), s.v2[1]), std::fmax(std::fmax(s.v0[2], s.v1[2]), s.v2[2])};
        return box_from_points(mn, mx);
    }

    // bounding_box() of every object, in id order (children have smaller ids).
    int compute_bounds() {
        f.object_bounds.resize(d->n_objects);
        obj_medium.assign(d->n_objects, 0);
        for (uint64_t i = 0; i < d->n_objects; ++i) {
            const rt_object& o = d->objects[i];
            Box b = box_empty();
            auto child_ok = [&](int32_t c) { return c >= 0 && (uint64_t)c < i; };
            switch (o.kind) {
            case RT_OBJ_SPHERE: if (o.a < 0 || (uint64_t)o.a >= d->n_spheres) return fail(RT_E_INVALID, "object: sphere index out of range"); b = sphere_box(o.a); break;
            case RT_OBJ_QUAD: if (o.a < 0 || (uint64_t)o.a >= d->n_quads) return fail(RT_E_INVALID, "object: quad index out of range"); b = quad_box(o.a); break;
            case RT_OBJ_TRIANGLE: if (o.a < 0 || (uint64_t)o.a >= d->n_triangles) return fail(RT_E_INVALID, "object: triangle index out of range"); b = tri_box(o.a); break;
            case RT_OBJ_LIST: case RT_OBJ_BVH: {
                if (o.a < 0 || o.b < 0) return fail(RT_E_INVALID, "object: negative child range");
                uint64_t limit = 0;
                switch (o.child_kind) {
                case RT_CHILD_OBJECTS: limit = d->n_children; break;
                case RT_CHILD_SPHERES: limit = d->n_spheres; break;
                case RT_CHILD_QUADS: limit = d->n_quads; break;
                case RT_CHILD_TRIANGLES: limit = d->n_triangles; break;
                default: return fail(RT_E_INVALID, "object: unknown child kind");
                }
                if ((uint64_t)o.a + (uint64_t)o.b > limit) return fail(RT_E_INVALID, "object: child range out of bounds");
                if (o.kind == RT_OBJ_BVH && o.b == 0) return fail(RT_E_INVALID, "bvh: built over an empty list (the reference indexes objects[0] and panics)");
                for (int32_t k = 0; k < o.b; ++k) { // hittable_list.rs:37-40 / bvh.rs:27-30: plain union, no padding
                    uint64_t idx = (uint64_t)o.a + (uint64_t)k;
                    if (o.child_kind == RT_CHILD_OBJECTS) {
                        int32_t c = d->children[idx];
                        if (!child_ok(c)) return fail(RT_E_INVALID, "object: a child must have a smaller id than its parent");
                        b = box_union(b, f.object_bounds[c]);
                        obj_medium[i] = std::max(obj_medium[i], obj_medium[c]);
                    } else if (o.child_kind == RT_CHILD_SPHERES) b = box_union(b, sphere_box(idx));
                    else if (o.child_kind == RT_CHILD_QUADS) b = box_union(b, quad_box(idx));
                    else b = box_union(b, tri_box(idx));
                }
                break;
            }
            case RT_OBJ_TRANSLATE: { // hittable.rs:68-69, aabb.rs:138-145
                if (!child_ok(o.a)) return fail(RT_E_INVALID, "translate: bad child");
                const Box& c = f.object_bounds[o.a];
                for (int k = 0; k < 3; ++k) { b.b[2 * k] = c.b[2 * k] + o.param[k]; b.b[2 * k + 1] = c.b[2 * k + 1] + o.param[k]; }
                obj_medium[i] = obj_medium[o.a];
                break;
            }
            case RT_OBJ_ROTATE_Y: { // hittable.rs:105-134
                if (!child_ok(o.a)) return fail(RT_E_INVALID, "rotate_y: bad child");
                const Box& c = f.object_bounds[o.a];
                double radians = o.param[0] * (PI / 180.0); // f64::to_radians
                volatile double radians_again = radians; // two libm calls whatever the compiler (glibc's sincos differs from sin / cos in the last place now and then)
                double sin_theta = std::sin(radians), cos_theta = std::cos(radians_again);
                double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
                for (int ii = 0; ii < 2; ++ii)
                    for (int jj = 0; jj < 2; ++jj)
                        for (int kk = 0; kk < 2; ++kk) {
                            double x = ii == 0 ? c.b[0] : c.b[1];
                            double y = jj == 0 ? c.b[2] : c.b[3];
                            double z = kk == 0 ? c.b[4] : c.b[5];
                            double newx = cos_theta * x + sin_theta * z;
                            double newz = -sin_theta * x + cos_theta * z;
                            double t[3] = {newx, y, newz};
                            for (int cc = 0; cc < 3; ++cc) { mn[cc] = std::fmin(mn[cc], t[cc]); mx[cc] = std::fmax(mx[cc], t[cc]); }
                        }
                b = box_from_points(V3{mn[0], mn[1], mn[2]}, V3{mx[0], mx[1], mx[2]});
                obj_medium[i] = obj_medium[o.a];
                break;
            }
            case RT_OBJ_CONSTANT_MEDIUM: // constant_medium.rs:106-108
                if (!child_ok(o.a)) return fail(RT_E_INVALID, "constant_medium: bad boundary");
                if (!mat_ok(o.mat)) return fail(RT_E_INVALID, "constant_medium: material out of range");
                if (obj_medium[o.a] >= 2) return fail(RT_E_UNSUPPORTED, "constant_medium: media nested more than two deep are not supported");
                // the reference never writes u,v for a medium hit (constant_medium.rs:96-101): a phase function over an image
                // texture reads what the record held before, which the kernels then have to keep track of
                if (d->materials[o.mat].tex >= 0 && texture_uses_uv(d->materials[o.mat].tex)) f.track_uv = true;
                b = f.object_bounds[o.a];
                obj_medium[i] = (uint8_t)(1 + obj_medium[o.a]);
                f.medium_depth = std::max(f.medium_depth, (int)obj_medium[i]);
                break;
            case RT_OBJ_OBJMODEL: { // obj_loader.rs:38-39,70-71,103-105
                if (o.a < 0 || o.b < 0 || (uint64_t)o.a + (uint64_t)o.b > d->n_triangles) return fail(RT_E_INVALID, "objmodel: triangle range out of bounds");
                double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
                for (int32_t k = 0; k < o.b; ++k) {
                    const rt_triangle& t = d->triangles[o.a + k];
                    const double* vs[3] = {t.v0, t.v1, t.v2};
                    for (auto* vv : vs)
                        for (int cc = 0; cc < 3; ++cc) { mn[cc] = std::fmin(mn[cc], vv[cc]); mx[cc] = std::fmax(mx[cc], vv[cc]); }
                }
                b = box_from_points(V3{mn[0], mn[1], mn[2]}, V3{mx[0], mx[1], mx[2]});
                break;
            }
            default: return fail(RT_E_INVALID, "object: unknown kind");
            }
            f.object_bounds[i] = b;
        }
        return RT_OK;
    }

    // Children of a LIST / BVH object as (box, ref, medium) items, flattened under context `ctx`.
    int gather_items(const rt_object& o, int32_t ctx, std::vector<Item>& items) {
        items.resize((size_t)o.b);
        for (int32_t k = 0; k < o.b; ++k) {
            uint64_t idx = (uint64_t)o.a + (uint64_t)k;
            Item& it = items[(size_t)k];
            it.medium = 0;
            switch (o.child_kind) {
            case RT_CHILD_OBJECTS: {
                int32_t c = d->children[idx];
                uint32_t r;
                int rc = flatten_object(c, ctx, r);
                if (rc) return rc;
                it.ref = r; it.box = f.object_bounds[c]; it.medium = obj_medium[c] ? 1 : 0;
                break;
            }
            case RT_CHILD_SPHERES: it.ref = make_ref(RT_REF_SPHERE, (uint32_t)idx); it.box = sphere_box(idx); break;
            case RT_CHILD_QUADS: it.ref = make_ref(RT_REF_QUAD, (uint32_t)idx); it.box = quad_box(idx); break;
            default: it.ref = make_ref(RT_REF_TRI, (uint32_t)idx); it.box = tri_box(idx); break;
            }
        }
        return RT_OK;
    }

    int flatten_object(int32_t id, int32_t ctx, uint32_t& out) {
        auto key = std::make_pair(id, ctx);
        auto it = memo.find(key);
        if (it != memo.end()) { out = it->second; return RT_OK; }
        const rt_object& o = d->objects[id];
        uint32_t ref = RT_REF_NONE;
        switch (o.kind) {
        case RT_OBJ_SPHERE: ref = make_ref(RT_REF_SPHERE, (uint32_t)o.a); break;
        case RT_OBJ_QUAD: ref = make_ref(RT_REF_QUAD, (uint32_t)o.a); break;
        case RT_OBJ_TRIANGLE: ref = make_ref(RT_REF_TRI, (uint32_t)o.a); break;
        case RT_OBJ_LIST: {
            std::vector<Item> items;
            int rc = gather_items(o, ctx, items);
            if (rc) return rc;
            DList l{(uint32_t)f.list_items.size(), (uint32_t)items.size()};
            for (auto& i2 : items) f.list_items.push_back(i2.ref);
            ref = make_ref(RT_REF_LIST, (uint32_t)f.lists.size());
            f.lists.push_back(l);
            break;
        }
        case RT_OBJ_OBJMODEL: {
            DList l{(uint32_t)f.list_items.size(), (uint32_t)o.b};
            for (int32_t k = 0; k < o.b; ++k) f.list_items.push_back(make_ref(RT_REF_TRI, (uint32_t)(o.a + k)));
            ref = make_ref(RT_REF_LIST, (uint32_t)f.lists.size());
            f.lists.push_back(l);
            break;
        }
        case RT_OBJ_BVH: {
            std::vector<Item> items;
            int rc = gather_items(o, ctx, items);
            if (rc) return rc;
            for (auto& i2 : items)
                if (box_has_nan(i2.box)) return fail(RT_E_INVALID, "bvh: NaN in a bounding box (the reference panics in partial_cmp().unwrap(), bvh.rs:90-93)");
            uint64_t count = bvh_node_count(items.size());
            uint64_t base = f.nodes.size();
            if (base + count >= (1ull << 28)) return fail(RT_E_TOO_LARGE, "bvh: more than 2^28 nodes");
            f.nodes.resize(base + count);
            f.wnodes.resize(base + count);
            if (int rcb = build_nodes(items, base)) return rcb;
            ref = make_ref(RT_REF_NODE, (uint32_t)base);
            {
                bool prim_only = true;
                for (auto& i2 : items) {
                    uint32_t k = RT_REF_KIND(i2.ref);
                    if (k != RT_REF_SPHERE && k != RT_REF_QUAD && k != RT_REF_TRI) { prim_only = false; break; }
                }
                bvh_prim_only[ref] = prim_only;
                bvh_roots.push_back((uint32_t)base);
            }
            break;
        }
        case RT_OBJ_TRANSLATE: case RT_OBJ_ROTATE_Y: {
            DXform x{};
            x.parent = ctx;
            x.depth = (ctx == 0 ? 0 : f.xforms[(size_t)ctx - 1].depth) + 1;
            if (x.depth > XFORM_DEPTH_MAX) return fail(RT_E_TOO_LARGE, "transform nesting deeper than 8 levels");
            if (o.kind == RT_OBJ_TRANSLATE) { x.kind = 0; x.a[0] = o.param[0]; x.a[1] = o.param[1]; x.a[2] = o.param[2]; }
            else {
                double radians = o.param[0] * (PI / 180.0); // hittable.rs:106
                volatile double radians_again = radians; // two libm calls, as the reference makes them (hittable.rs:106-108)
                x.kind = 1; x.a[0] = std::sin(radians); x.a[1] = std::cos(radians_again); x.a[2] = 0.0;
            }
            size_t xi = f.xforms.size();
            f.xforms.push_back(x);
            uint32_t child;
            int rc = flatten_object(o.a, (int32_t)xi + 1, child);
            if (rc) return rc;
            f.xforms[xi].child = child;
            ref = make_ref(RT_REF_XFORM, (uint32_t)xi);
            break;
        }
        case RT_OBJ_CONSTANT_MEDIUM: {
            uint32_t boundary;
            int rc = flatten_object(o.a, ctx, boundary);
            if (rc) return rc;
            DMedium m{};
            m.boundary = boundary; m.mat = o.mat; m.neg_inv_density = -1.0 / o.param[0]; // constant_medium.rs:29
            m.object = id;
            ref = make_ref(RT_REF_MEDIUM, (uint32_t)f.media.size());
            f.media.push_back(m);
            f.has_medium = true;
            break;
        }
        default: return fail(RT_E_INVALID, "object: unknown kind");
        }
        memo[key] = ref;
        out = ref;
        return RT_OK;
    }

    // Collapses two levels of the binary tree under wide node `idx` into the 4-wide node `me` (already
    // allocated).  Children keep the reference's left-to-right order; which of them is entered first is
    // decided per ray at run time.  The inner children of a node get consecutive indices starting at an
    // even one, so that siblings share 128-byte lines (two 64-byte nodes per line): a ray that enters
    // several children of a node finds the later ones in a line it has already pulled in.
    // Returns the stack entries a traversal below this node can need.
    std::vector<uint8_t> qdepth; // depth of every 4-wide node below the root of its tree
    // exact_boxes != nullptr: also record, for every child, the f64 box the reference tests last on its way to that
    // child -- the child's own node box for an inner child, the box of the binary node that holds it for a primitive
    // (the reference tests node boxes only, bvh.rs:118-140; a primitive is tested whenever its node's box passes).
    // Entry ((wide node - exact_base) * 4 + slot) * 6 of *exact_boxes, for the traversal that culls exactly like Aabb::hit.
    std::vector<double>* exact_boxes = nullptr;
    uint32_t exact_base = 0;
    int32_t fill_qnode(uint32_t me, uint32_t idx, int depth) {
        qdepth[me] = (uint8_t)std::min(depth, 255);
        uint32_t refs[4];
        float boxes[4][6];
        const double* path_box[4] = {nullptr, nullptr, nullptr, nullptr};
        int n = 0;
        auto add = [&](uint32_t ref, const float* b, uint32_t holder) {
            refs[n] = ref;
            for (int k = 0; k < 6; ++k) boxes[n][k] = b[k];
            path_box[n] = RT_REF_KIND(ref) == RT_REF_NODE ? f.nodes[RT_REF_INDEX(ref)].bounds : f.nodes[holder].bounds;
            ++n;
        };
        auto expand = [&](uint32_t ref, const float* b) {
            if (RT_REF_KIND(ref) == RT_REF_NODE) {
                const WNode& c = f.wnodes[RT_REF_INDEX(ref)];
                add(c.left, c.lb, RT_REF_INDEX(ref));
                if (!(c.meta & 2u)) add(c.right, c.rb, RT_REF_INDEX(ref));
            } else add(ref, b, idx);
        };
        {
            const WNode w = f.wnodes[idx];
            expand(w.left, w.lb);
            if (!(w.meta & 2u)) expand(w.right, w.rb);
        }
        int n_inner = 0;
        for (int j = 0; j < n; ++j) n_inner += RT_REF_KIND(refs[j]) == RT_REF_NODE ? 1 : 0;
        uint32_t base = 0;
        if (n_inner) {
            if (f.qnodes.size() & 1u) add_empty_qnode(); // pad: sibling groups start on a line boundary
            base = (uint32_t)f.qnodes.size();
            for (int k = 0; k < n_inner; ++k) add_empty_qnode();
        }
        if (exact_boxes) {
            const size_t at = (size_t)(me - exact_base) * 24;
            if (exact_boxes->size() < at + 24) exact_boxes->resize(std::max(at + 24, (size_t)(f.qnodes.size() - exact_base) * 24), 0.0);
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < 6; ++k) (*exact_boxes)[at + (size_t)j * 6 + k] = path_box[j][k];
        }
        int32_t deepest = 0;
        uint32_t out_refs[4] = {RT_REF_NONE, RT_REF_NONE, RT_REF_NONE, RT_REF_NONE};
        int k = 0;
        for (int j = 0; j < n; ++j) {
            if (RT_REF_KIND(refs[j]) == RT_REF_NODE) {
                const uint32_t child = base + (uint32_t)k++;
                out_refs[j] = make_ref(REF_QNODE, child);
                deepest = std::max(deepest, fill_qnode(child, RT_REF_INDEX(refs[j]), depth + 1));
            } else out_refs[j] = quad_run(refs[j]);
        }
        QNode& q = f.qnodes[me];
        for (int j = 0; j < 4; ++j) {
            q.child[j] = out_refs[j];
            q._pad[j] = 0;
            for (int a = 0; a < 3; ++a) {
                // an empty slot gets an inverted box, which no ray enters
                q.lo[a][j] = j < n ? boxes[j][2 * a] : std::numeric_limits<float>::infinity();
                q.hi[a][j] = j < n ? boxes[j][2 * a + 1] : -std::numeric_limits<float>::infinity();
            }
        }
        return (n - 1) + deepest; // up to n-1 siblings stay on the stack while one child is traversed
    }
    // The tree of RT_HIT_ORDERED | RT_HIT_EXACT_CULL for a world that is one BVH over primitives: the 4-wide collapse of
    // the reference's tree once more, with the exact boxes beside it, under a root node of its own whose single child is
    // the BVH's root (so that the root's box is tested like every other).
    int exact_cull_tree(uint32_t root) {
        if (f.qnodes.size() & 1u) add_empty_qnode();
        f.exact_first = (uint32_t)f.qnodes.size();
        const uint32_t top = f.exact_first;
        add_empty_qnode();
        add_empty_qnode(); // (keeps the BVH root's own sibling group on an even index)
        const uint32_t me = (uint32_t)f.qnodes.size();
        add_empty_qnode();
        f.ebox.clear();
        exact_boxes = &f.ebox;
        exact_base = f.exact_first;
        const int32_t need = fill_qnode(me, root, 1);
        exact_boxes = nullptr;
        f.ebox.resize((size_t)(f.qnodes.size() - f.exact_first) * 24, 0.0);
        QNode& q = f.qnodes[top];
        q.child[0] = make_ref(REF_QNODE, me);
        widen_box(f.nodes[root].bounds, q, 0);
        for (int k = 0; k < 6; ++k) f.ebox[k] = f.nodes[root].bounds[k];
        qdepth[top] = 0;
        f.exact_root = make_ref(REF_QNODE, top);
        f.exact_stack = (uint32_t)need + 2;
        f.exact_count = (uint32_t)f.qnodes.size() - f.exact_first;
        return RT_OK;
    }
    void widen_box(const double* b, QNode& q, int j) {
        float o[6];
        widen(b, o);
        for (int a = 0; a < 3; ++a) { q.lo[a][j] = o[2 * a]; q.hi[a][j] = o[2 * a + 1]; }
    }
    // The renderer's primitive-BVH kernel is several times faster per ray than the general traversal, but it only
    // walks a BVH whose leaves are primitives.  When the world is a list whose FIRST member is a BVH without
    // transforms or media below it (Book-2: the field of boxes, each a list of six quads), the primitives below that
    // member are gathered into one more BVH, used for nothing but this: the kernel finds the member's closest hit,
    // and the general traversal then walks the remaining members with that hit as the interval's upper end, which
    // is the order HittableList::hit visits them in (hittable_list.rs:46-57).  The closest hit over a set of
    // primitives does not depend on the tree over them (ties aside, as everywhere in the ordered traversal).
    bool hoist_whole_world = false; // set by hoist_first_member: the extra BVH covers every member of the world list
    int hoist_first_member(int64_t& root_out) {
        root_out = -1;
        if (RT_REF_KIND(f.world) != RT_REF_LIST) return RT_OK;
        const DList w = f.lists[RT_REF_INDEX(f.world)];
        if (w.count < 2u) return RT_OK;
        // Primitives below `start`, or false when an instance or a medium lies below it.
        auto gather = [&](uint32_t start, std::vector<Item>& items) -> bool {
            std::vector<uint32_t> todo{start};
            while (!todo.empty()) {
                const uint32_t r = todo.back(); todo.pop_back();
                const uint32_t k = RT_REF_KIND(r), i = RT_REF_INDEX(r);
                if (k == RT_REF_NODE) {
                    const DNode& n = f.nodes[i];
                    if (n.flags & 1u) return false; // a medium below: keep the reference order
                    todo.push_back(n.left);
                    if (n.right != n.left) todo.push_back(n.right); // a one-object span stores the object twice (bvh.rs:41-44)
                } else if (k == RT_REF_SPHERE || k == RT_REF_QUAD || k == RT_REF_TRI) {
                    Item it; it.ref = r; it.medium = 0;
                    it.box = k == RT_REF_SPHERE ? sphere_box(i) : (k == RT_REF_QUAD ? quad_box(i) : tri_box(i));
                    items.push_back(it);
                } else if (k == RT_REF_LIST) {
                    const DList l = f.lists[i];
                    for (uint32_t j = 0; j < l.count; ++j) todo.push_back(f.list_items[l.first + j]);
                } else {
                    return false; // instances, media: not hoisted
                }
            }
            for (auto& it : items) {
                if (box_has_nan(it.box)) return false;
                for (double v : it.box.b) if (!std::isfinite(v)) return false;
            }
            return true;
        };
        auto build = [&](std::vector<Item>& items) -> int64_t {
            const uint64_t count = bvh_node_count(items.size());
            const uint64_t base = f.nodes.size();
            if (base + count >= (1ull << 28)) return -1;
            f.nodes.resize(base + count);
            f.wnodes.resize(base + count);
            if (build_nodes(items, base)) return -1;
            bvh_roots.push_back((uint32_t)base);
            return (int64_t)base;
        };
        // (1) A world without instances and media: the closest hit over all its primitives does not depend on the
        //     order of the list, so ONE BVH over all of them replaces the whole walk (mesh + spheres: two BVHs in a list).
        if (!f.has_medium) {
            std::vector<Item> all;
            if (gather(f.world, all) && all.size() >= 64u) {
                root_out = build(all);
                hoist_whole_world = root_out >= 0;
                return RT_OK;
            }
        }
        // (2) Otherwise the first member alone, when it is such a BVH.
        const uint32_t first = f.list_items[w.first];
        if (RT_REF_KIND(first) != RT_REF_NODE) return RT_OK;
        auto po = bvh_prim_only.find(first);
        const bool prim_only = po != bvh_prim_only.end() && po->second;
        std::vector<Item> items;
        if (!gather(first, items)) return RT_OK;
        // too small to pay for a second launch per bounce; a BVH whose leaves are primitives already is walked well
        // enough by the general traversal until it is large (mesh + spheres, first member 485 spheres: 3 % slower split)
        if (items.size() < (prim_only ? 4096u : 64u)) return RT_OK;
        if (prim_only) { // its own 4-wide copy serves
            root_out = (int64_t)RT_REF_INDEX(first);
            return RT_OK;
        }
        root_out = build(items);
        return RT_OK;
    }

    // When the BVH at binary node `root` holds nothing but triangles (a Mesh), the nearest-first kernel gets a tree of
    // its own over them (fast_tree.cpp); the collapse of the reference tree stays where it is for the traversals that
    // follow the reference's boxes.
    int traversal_only_tree(uint32_t root) {
        if (!fast_tree_min) return RT_OK;
        FastTreeResult res;
        if (early.started && (early.root != root || !f.qnodes.empty())) { // not the BVH it was started for after all: drop it
            (void)early.done.get();
            early.started = false;
            early.scene = FlatScene();
        }
        if (early.started) {
            // built beside the reference BVHs (start_early_fast_tree): its arrays become the first nodes of the scene's
            const int rc = early.done.get();
            early.started = false;
            if (rc == RT_E_UNSUPPORTED) { early.scene = FlatScene(); return RT_OK; } // (see build_fast_tree: not an error of the scene)
            if (rc) return fail(rc, early.err);
            f.qnodes = std::move(early.scene.qnodes);
            f.cqnodes = std::move(early.scene.cqnodes);
            f.pnodes = std::move(early.scene.pnodes);
            f.ptris = std::move(early.scene.ptris);
            qdepth = std::move(early.depth);
            res = early.res;
            fast_tree_first = 0;
        } else {
            std::vector<uint32_t> ids, todo{root};
            while (!todo.empty()) {
                const uint32_t i = todo.back(); todo.pop_back();
                const DNode& n = f.nodes[i];
                const uint32_t kids[2] = {n.left, n.right};
                for (int k = 0; k < (n.left == n.right ? 1 : 2); ++k) {
                    if (RT_REF_KIND(kids[k]) == RT_REF_NODE) todo.push_back(RT_REF_INDEX(kids[k]));
                    else if (RT_REF_KIND(kids[k]) == RT_REF_TRI) ids.push_back(RT_REF_INDEX(kids[k]));
                    else return RT_OK; // spheres or quads among the leaves: the collapse of the reference tree serves
                }
            }
            if (ids.size() < fast_tree_min) return RT_OK;
            std::sort(ids.begin(), ids.end()); // description order: the tree does not depend on the reference tree's leaf order
            std::string e;
            fast_tree_first = f.qnodes.size();
            const int rc = build_fast_tree(f.tris.data(), f, qdepth, ids.data(), ids.size(), res, e);
            if (rc == RT_E_UNSUPPORTED) return RT_OK;
            if (rc) return fail(rc, e);
        }
        f.fast_root = make_ref(REF_QNODE, res.root);
        f.fast_stack = (uint32_t)res.stack_need + 1;
        f.fast_tree = true;
        fast_tree_count = (size_t)res.wide_nodes;
        fast_ratio_sum = res.ratio_sum; fast_ratio_n = res.ratio_n;
        fast_leaf_area = res.leaf_area;
        // the general kernel's nearest-first traversal enters the same tree at this BVH's root
        need_q[root] = res.stack_need;
        f.wnodes[root].qroot = res.root + 1;
        // (the builder has counted it: no second walk over the tree's nodes when the scene's stack depth is worked out)
        if (need_wide_memo.size() < f.qnodes.size()) need_wide_memo.resize(f.qnodes.size(), -1);
        need_wide_memo[res.root] = res.stack_need;
        return RT_OK;
    }
    // The common case, a world that is a Mesh (a BVH declared over a range of triangles): the traversal-only tree needs
    // nothing but the triangles, so it is built on threads of its own while this one builds the reference's BVHs.
    struct EarlyFastTree {
        bool started = false;
        uint32_t root = 0;       // first binary node of the BVH it belongs to
        std::future<int> done;
        FlatScene scene;         // holds the tree's arrays until they are merged
        std::vector<uint8_t> depth;
        std::vector<uint32_t> ids;
        FastTreeResult res;
        std::string err;
    } early;
    void start_early_fast_tree() {
        if (!fast_tree_min || d->world < 0 || (uint64_t)d->world >= d->n_objects) return;
        const rt_object* o = &d->objects[d->world];
        if (o->kind == RT_OBJ_LIST && o->child_kind == RT_CHILD_OBJECTS && o->b == 1 && o->a >= 0 && (uint64_t)o->a < d->n_children) {
            const int32_t c = d->children[o->a];
            if (c < 0 || c >= d->world) return;
            o = &d->objects[c];
        }
        if (o->kind != RT_OBJ_BVH || o->child_kind != RT_CHILD_TRIANGLES || o->a < 0 || o->b < 0) return;
        if ((uint64_t)o->a + (uint64_t)o->b > d->n_triangles || (uint64_t)o->b < fast_tree_min) return;
        early.ids.resize((size_t)o->b);
        for (int32_t k = 0; k < o->b; ++k) early.ids[(size_t)k] = (uint32_t)(o->a + k);
        early.root = 0; // the world's BVH is the first one built: its nodes start the node array
        // (no thread to be had: the tree is built later, in its turn, by traversal_only_tree)
        early.started = try_async(early.done, [this]() {
            return build_fast_tree(f.tris.data(), early.scene, early.depth, early.ids.data(), early.ids.size(), early.res, early.err);
        });
    }
    bool need_memo_kept = false; // need_node already holds the binary nodes below the world's BVH root (filled while the traversal-only tree was being built)
    size_t quantised_first = 0, quantised_count = 0; // wide nodes whose 8-bit form exists already (the exact-cull tree, when it was built beside the traversal-only one)
    size_t fast_tree_first = 0, fast_tree_count = 0; // the traversal-only tree's nodes in qnodes (they come with their 8-bit and packed forms)
    double fast_ratio_sum = 0.0, fast_leaf_area = 0.0;
    uint64_t fast_ratio_n = 0;

    // A list of 2..16 quads stored one after the other (box_new: quad.rs:140-190 pushes the six sides in a row)
    // is entered through a reference that names the quads directly.
    uint32_t quad_run(uint32_t ref) {
        if (RT_REF_KIND(ref) != RT_REF_LIST) return ref;
        const DList& l = f.lists[RT_REF_INDEX(ref)];
        if (l.count < 2u || l.count > 16u) return ref;
        const uint32_t r0 = f.list_items[l.first];
        if (RT_REF_KIND(r0) != RT_REF_QUAD || RT_REF_INDEX(r0) + l.count > (1u << 24)) return ref;
        for (uint32_t k = 1; k < l.count; ++k)
            if (f.list_items[l.first + k] != make_ref(RT_REF_QUAD, RT_REF_INDEX(r0) + k)) return ref;
        const uint32_t run = make_ref(REF_QUADS, ((l.count - 1u) << 24) | RT_REF_INDEX(r0));
        f.quad_runs.emplace_back(run, ref);
        return run;
    }
    void add_empty_qnode() { // all four slots empty: never entered, keeps the index arithmetic simple
        QNode q{};
        for (int j = 0; j < 4; ++j) {
            q.child[j] = RT_REF_NONE;
            for (int a = 0; a < 3; ++a) { q.lo[a][j] = std::numeric_limits<float>::infinity(); q.hi[a][j] = -std::numeric_limits<float>::infinity(); }
        }
        f.qnodes.push_back(q);
        qdepth.push_back(255);
    }

    // 8-bit copy of every 4-wide node (same indices).  Returns the mean, over the child boxes of the top
    // four levels (the nodes nearly every ray passes), of (surface area quantised / surface area f32), each
    // ratio capped at 16: close to 1 when siblings are of similar size, large when one huge primitive forces
    // a coarse grid onto its small neighbours all the way up the tree (then the f32 nodes are the better choice).
    double quantise_qnodes() {
        f.cqnodes.resize(f.qnodes.size(), CQNode{}); // (the traversal-only tree has brought its 8-bit nodes along: fast_tree.cpp)
        double ratio_sum = fast_ratio_sum;
        uint64_t ratio_n = fast_ratio_n;
        const size_t fast_end = f.fast_tree ? fast_tree_first + fast_tree_count : 0;
        const size_t total = f.qnodes.size();
        // (the nodes near a root, whose ratios are summed, are few: they are done in order by this thread; the rest in parallel)
        auto quantise = [&](size_t lo, size_t hi, bool near_root) {
            for (size_t i = lo; i < hi; ++i) {
                if (f.fast_tree && i >= fast_tree_first && i < fast_end) continue;          // came with their 8-bit form
                if (!near_root && i >= quantised_first && i < quantised_first + quantised_count) continue; // done while waiting for that tree
                if ((qdepth[i] < 4) != near_root) continue;
                quantise_node(f.qnodes[i], f.cqnodes[i]);
                if (near_root)
                    for (int j = 0; j < 4; ++j) {
                        if (f.qnodes[i].child[j] == RT_REF_NONE) continue;
                        const double r = quantised_area_ratio(f.qnodes[i], f.cqnodes[i], j);
                        if (r >= 0.0) { ratio_sum += r; ++ratio_n; }
                    }
            }
        };
        const int parts = total >= (1u << 16) ? (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())) : 1;
        for_parts(parts, parts - 1, [&](int p) { quantise(total * (size_t)p / (size_t)parts, total * (size_t)(p + 1) / (size_t)parts, false); });
        quantise(0, total, true);
        return ratio_n ? ratio_sum / (double)ratio_n : 1.0;
    }

    // Packed copy (PNode / PTri) of the 8-bit tree at fast_root, when every leaf below it is a triangle.
    void pack_fast_tree() {
        auto half = [](const Box& b) {
            const double dx = b.b[1] - b.b[0], dy = b.b[3] - b.b[2], dz = b.b[5] - b.b[4];
            return dx * dy + dy * dz + dz * dx;
        };
        if (f.fast_tree) { // packed by fast_tree.cpp; the 8-bit boxes may have been rejected since
            if (f.cqnodes.empty() || f.fast_partial) { f.pnodes.clear(); f.ptris.clear(); }
            const double wa = half(f.world_bounds);
            f.fast_leaf_overlap = wa > 0.0 ? fast_leaf_area / wa : 0.0;
            return;
        }
        f.pnodes.clear();
        f.ptris.clear();
        if (f.cqnodes.empty() || f.fast_root == RT_REF_NONE || f.fast_partial) return;
        std::vector<std::pair<uint32_t, uint32_t>> todo; // (index in cqnodes, index in pnodes)
        auto half_area = [](const Box& b) {
            const double dx = b.b[1] - b.b[0], dy = b.b[3] - b.b[2], dz = b.b[5] - b.b[4];
            return dx * dy + dy * dz + dz * dx;
        };
        double leaf_area = 0.0;
        f.pnodes.push_back(PNode{});
        todo.emplace_back(RT_REF_INDEX(f.fast_root), 0u);
        while (!todo.empty()) {
            const uint32_t q = todo.back().first, me = todo.back().second;
            todo.pop_back();
            const CQNode& c = f.cqnodes[q];
            int order[4], n_inner = 0, n_all = 0;
            for (int j = 0; j < 4; ++j)
                if (c.child[j] != RT_REF_NONE && RT_REF_KIND(c.child[j]) == REF_QNODE) order[n_all++] = j;
            n_inner = n_all;
            for (int j = 0; j < 4; ++j) {
                if (c.child[j] == RT_REF_NONE || RT_REF_KIND(c.child[j]) == REF_QNODE) continue;
                if (RT_REF_KIND(c.child[j]) != RT_REF_TRI) { f.pnodes.clear(); f.ptris.clear(); return; }
                order[n_all++] = j;
            }
            PNode n{};
            for (int a = 0; a < 3; ++a) { n.origin[a] = c.origin[a]; n.exp[a] = c.exp[a]; }
            n.counts = (uint8_t)(n_inner | (n_all << 4));
            for (int k = 0; k < 4; ++k)
                for (int a = 0; a < 3; ++a) {
                    n.qlo[a][k] = k < n_all ? c.qlo[a][order[k]] : 255;
                    n.qhi[a][k] = k < n_all ? c.qhi[a][order[k]] : 0;
                }
            const uint32_t node_base = (uint32_t)f.pnodes.size(), leaf_base = (uint32_t)f.ptris.size();
            // (the kernel addresses both arrays by 32-bit byte offsets: they stay below 4 GB, i.e. 2^25 triangles; larger
            // meshes keep the 64-byte nodes)
            if (((uint64_t)node_base + 4u) * sizeof(PNode) >= (1ull << 32) || ((uint64_t)leaf_base + 4u) * sizeof(PTri) >= (1ull << 32)) { f.pnodes.clear(); f.ptris.clear(); return; }
            n.node_base = make_ref(REF_QNODE, node_base);
            n.leaf_base = make_ref(RT_REF_TRI, leaf_base) - (uint32_t)n_inner;
            for (int k = 0; k < n_inner; ++k) {
                f.pnodes.push_back(PNode{});
                todo.emplace_back(RT_REF_INDEX(c.child[order[k]]), node_base + (uint32_t)k);
            }
            for (int k = n_inner; k < n_all; ++k) {
                const uint32_t t = RT_REF_INDEX(c.child[order[k]]);
                {   // the box this leaf is entered through (of the whole triangle, or of its part on this side of a spatial split)
                    const QNode& w = f.qnodes[q];
                    Box lb;
                    for (int a = 0; a < 3; ++a) { lb.b[2 * a] = w.lo[a][order[k]]; lb.b[2 * a + 1] = w.hi[a][order[k]]; }
                    leaf_area += half_area(lb);
                }
                PTri pt{};
                for (int a = 0; a < 3; ++a) { pt.v0[a] = f.tris[t].v0[a]; pt.v1[a] = f.tris[t].v1[a]; pt.v2[a] = f.tris[t].v2[a]; }
                pt.orig = t;
                f.ptris.push_back(pt);
            }
            f.pnodes[me] = n;
        }
        const double ra = half_area(f.world_bounds);
        f.fast_leaf_overlap = ra > 0.0 ? leaf_area / ra : 0.0;
    }

    // Stack entries the ordered traversal can need below 4-wide node `q`: up to (children - 1) siblings stay deferred
    // while one child is walked, and that child is another 4-wide node or a leaf with a need of its own (a list pushes
    // its continuation, an instance its restore entry, and what is below them goes on top of that).
    std::vector<int32_t> need_wide_memo;
    int32_t need_wide(uint32_t q) {
        if (need_wide_memo.size() < f.qnodes.size()) need_wide_memo.resize(f.qnodes.size(), -1);
        if (need_wide_memo[q] >= 0) return need_wide_memo[q];
        int32_t n = 0, deepest = 0;
        for (int j = 0; j < 4; ++j) {
            const uint32_t c = f.qnodes[q].child[j];
            if (c == RT_REF_NONE) continue;
            ++n;
            if (RT_REF_KIND(c) == REF_QNODE) deepest = std::max(deepest, need_wide(RT_REF_INDEX(c)));
            else if (RT_REF_KIND(c) != REF_QUADS) deepest = std::max(deepest, need(c)); // a run of quads is tested in place
        }
        return need_wide_memo[q] = std::max(0, n - 1) + deepest;
    }

    // Stack entries a traversal starting at `ref` can need.
    int32_t need(uint32_t ref) {
        uint32_t kind = RT_REF_KIND(ref), idx = RT_REF_INDEX(ref);
        switch (kind) {
        case RT_REF_NODE: {
            if (need_node[idx] >= 0) return need_node[idx];
            const DNode& n = f.nodes[idx];
            int32_t l = need(n.left);
            int32_t r;
            if (n.left == n.right && !(n.flags & 1u)) r = l; // the duplicate child is not pushed
            else r = std::max(1 + l, need(n.right));
            if (f.wnodes[idx].qroot) r = std::max(r, need_wide(f.wnodes[idx].qroot - 1)); // ordered traversal enters the 4-wide copy here
            need_node[idx] = r;
            return r;
        }
        case RT_REF_LIST: {
            const DList& l = f.lists[idx];
            int32_t m = 0;
            // very long primitive-only lists (ObjModel) need no per-item recursion
            for (uint32_t k = 0; k < l.count; ++k) {
                uint32_t r = f.list_items[l.first + k];
                uint32_t rk = RT_REF_KIND(r);
                if (rk == RT_REF_SPHERE || rk == RT_REF_QUAD || rk == RT_REF_TRI) continue;
                m = std::max(m, need(r));
            }
            return (f.track_uv ? 8 : 2) + m; // continuation (2), and with record history the caller's u,v source and the end marker (6)
        }
        case RT_REF_XFORM: return 1 + need(f.xforms[idx].child);
        case RT_REF_MEDIUM: return need(f.media[idx].boundary);
        default: return 0;
        }
    }

    // MI355RT_DEBUG: where the time of a scene's set-up goes
    std::chrono::steady_clock::time_point t_mark = std::chrono::steady_clock::now();
    const bool debug = getenv("MI355RT_DEBUG") != nullptr;
    void stage(const char* name) {
        const auto now = std::chrono::steady_clock::now();
        if (debug) std::fprintf(stderr, "[mi355rt] flatten: %-28s %.3f s\n", name, std::chrono::duration<double>(now - t_mark).count());
        t_mark = now;
    }

    int run() {
        int rc;
        if (!d) return fail(RT_E_INVALID, "null scene description");
        if (d->n_spheres >= (1ull << 28) || d->n_quads >= (1ull << 28) || d->n_triangles >= (1ull << 28) || d->n_objects >= (1ull << 28))
            return fail(RT_E_TOO_LARGE, "more than 2^28 primitives or objects");
        if ((rc = check_textures())) return rc;
        if ((rc = check_materials())) return rc;
        f.images.assign(d->images, d->images + d->n_images);
        f.image_bytes.assign(d->image_bytes, d->image_bytes + d->n_image_bytes);
        f.perlins.assign(d->perlins, d->perlins + d->n_perlins);
        stage("tables");
        if ((rc = build_primitives())) return rc;
        stage("primitives");
        start_early_fast_tree();
        if ((rc = compute_bounds())) return rc;
        stage("bounds");
        if (d->world < 0 || (uint64_t)d->world >= d->n_objects) return fail(RT_E_INVALID, "world: object id out of range");
        if ((rc = flatten_object(d->world, 0, f.world))) return rc;
        f.world_bounds = f.object_bounds[d->world];
        stage("object graph + BVHs");
        // lights (camera.rs:81; hittable_list.rs:67-81, quad.rs:117-136, sphere.rs:117-141)
        if (d->lights >= 0) {
            if ((uint64_t)d->lights >= d->n_objects) return fail(RT_E_INVALID, "lights: object id out of range");
            auto light_of = [&](int32_t id) -> DLight {
                const rt_object& o = d->objects[id];
                if (o.kind == RT_OBJ_SPHERE) return DLight{RT_REF_SPHERE, (uint32_t)o.a};
                if (o.kind == RT_OBJ_QUAD) return DLight{RT_REF_QUAD, (uint32_t)o.a};
                return DLight{REF_OTHER, 0u}; // trait defaults: pdf_value 0, random (1,0,0)
            };
            const rt_object& lo = d->objects[d->lights];
            if (lo.kind == RT_OBJ_LIST) {
                f.lights_is_list = 1;
                for (int32_t k = 0; k < lo.b; ++k) {
                    uint64_t idx = (uint64_t)lo.a + (uint64_t)k;
                    if (lo.child_kind == RT_CHILD_OBJECTS) f.lights.push_back(light_of(d->children[idx]));
                    else if (lo.child_kind == RT_CHILD_SPHERES) f.lights.push_back(DLight{RT_REF_SPHERE, (uint32_t)idx});
                    else if (lo.child_kind == RT_CHILD_QUADS) f.lights.push_back(DLight{RT_REF_QUAD, (uint32_t)idx});
                    else f.lights.push_back(DLight{REF_OTHER, 0u});
                }
                if (f.lights.empty())
                    return fail(RT_E_INVALID, "lights: empty list (the reference panics in random_int(0,-1), hittable_list.rs:79)");
            } else {
                f.lights_is_list = 0;
                f.lights.push_back(light_of(d->lights));
            }
            f.n_lights = (int32_t)f.lights.size();
        } else {
            f.n_lights = -1;
        }
        int64_t hoisted_root = -1;
        // (with record history the world list is walked as declared: its first member's hit is part of that history)
        if (!f.track_uv)
            if (int rc2 = hoist_first_member(hoisted_root)) return rc2;
        // 4-wide copies of every BVH that holds no constant medium and has finite bounds (ordered traversal)
        f.qnodes.reserve(f.nodes.size() / 2 + 16);
        // the BVH the primitive-BVH kernel will walk, if any: the world itself, or the one gathered from all its members
        int64_t fast_candidate = -1;
        {
            uint32_t r = f.world;
            if (RT_REF_KIND(r) == RT_REF_LIST && f.lists[RT_REF_INDEX(r)].count == 1) r = f.list_items[f.lists[RT_REF_INDEX(r)].first];
            auto it = bvh_prim_only.find(r);
            if (RT_REF_KIND(r) == RT_REF_NODE && it != bvh_prim_only.end() && it->second) fast_candidate = (int64_t)RT_REF_INDEX(r);
            else if (hoisted_root >= 0 && hoist_whole_world) fast_candidate = hoisted_root;
        }
        for (uint32_t root : bvh_roots) {
            const DNode& n = f.nodes[root];
            bool ok = !(n.flags & 1u);
            for (double v : n.bounds) ok = ok && std::isfinite(v);
            if (!ok) continue;
            if ((int64_t)root == fast_candidate) {
                // a mesh gets a tree built for the nearest-first traversal instead of the collapse (first: a tree that was
                // built beside the reference BVHs brings its arrays along as the scene's first wide nodes)
                if (early.started && early.root == root && f.qnodes.empty() && !getenv("MI355RT_NO_BUILD_OVERLAP")) {
                    // That tree is still being built on its own threads, and the exact-cull tree needs nothing but the
                    // reference's nodes: collapse it meanwhile, from index 0, and move it behind the other afterwards.
                    if (int rce = exact_cull_tree(root)) return rce;
                    RawVec<QNode> held;
                    std::vector<uint8_t> held_depth;
                    held.swap(f.qnodes);
                    held_depth.swap(qdepth);
                    // (its 8-bit form too, with a few threads: the others are busy with the tree this one waits for)
                    RawVec<CQNode> held_cq(held.size());
                    for_parts(4, 3, [&](int p) { for (size_t i = held.size() * (size_t)p / 4; i < held.size() * (size_t)(p + 1) / 4; ++i) quantise_node(held[i], held_cq[i]); });
                    // (and the stack need of the reference-order traversal below this root, a walk over every binary node:
                    // nothing below the root depends on the trees still to come; the memo serves the final pass)
                    need_node.assign(f.nodes.size(), -1);
                    need_memo_kept = true;
                    (void)need(f.nodes[root].left);
                    if (f.nodes[root].right != f.nodes[root].left) (void)need(f.nodes[root].right);
                    if (int rcf = traversal_only_tree(root)) return rcf;
                    if (!f.fast_tree) {
                        // no such tree after all (coordinates its f32 boxes cannot hold): the collapse of the reference's
                        // tree serves, built below as for any other BVH, and the exact-cull tree behind it
                        f.ebox.clear();
                        f.exact_root = RT_REF_NONE; f.exact_first = f.exact_count = f.exact_stack = 0;
                    } else {
                    if (f.qnodes.size() & 1u) add_empty_qnode(); // sibling groups of the moved tree stay on even indices
                    const uint32_t shift = (uint32_t)f.qnodes.size();
                    for (QNode& q : held)
                        for (int j = 0; j < 4; ++j)
                            if (q.child[j] != RT_REF_NONE && RT_REF_KIND(q.child[j]) == REF_QNODE) q.child[j] = make_ref(REF_QNODE, RT_REF_INDEX(q.child[j]) + shift);
                    f.qnodes.insert(f.qnodes.end(), held.begin(), held.end());
                    qdepth.insert(qdepth.end(), held_depth.begin(), held_depth.end());
                    for (CQNode& c : held_cq) // the 8-bit nodes carry the same child references
                        for (int j = 0; j < 4; ++j)
                            if (c.child[j] != RT_REF_NONE && RT_REF_KIND(c.child[j]) == REF_QNODE) c.child[j] = make_ref(REF_QNODE, RT_REF_INDEX(c.child[j]) + shift);
                    f.cqnodes.resize(shift, CQNode{});
                    f.cqnodes.insert(f.cqnodes.end(), held_cq.begin(), held_cq.end());
                    quantised_first = shift; quantised_count = held.size();
                    f.exact_first += shift;
                    f.exact_root = make_ref(REF_QNODE, RT_REF_INDEX(f.exact_root) + shift);
                    continue;
                    }
                }
                if (int rcf = traversal_only_tree(root)) return rcf;
                if (f.fast_tree) {
                    if (int rce = exact_cull_tree(root)) return rce;
                    continue;
                }
            }
            if (f.qnodes.size() & 1u) add_empty_qnode();
            const uint32_t me = (uint32_t)f.qnodes.size();
            add_empty_qnode();
            need_q[root] = fill_qnode(me, root, 0);
            f.wnodes[root].qroot = me + 1;
            if ((int64_t)root == fast_candidate)
                if (int rce = exact_cull_tree(root)) return rce;
        }
        auto measure_extent = [&]() {
            const size_t total = f.qnodes.size();
            const int parts = total >= (1u << 16) ? (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())) : 1;
            std::vector<float> ext((size_t)parts, 0.0f);
            for_parts(parts, parts - 1, [&](int p) {
                float e = 0.0f;
                for (size_t i = total * (size_t)p / (size_t)parts; i < total * (size_t)(p + 1) / (size_t)parts; ++i) {
                    const QNode& q = f.qnodes[i];
                    for (int j = 0; j < 4; ++j)
                        if (q.child[j] != RT_REF_NONE)
                            for (int a = 0; a < 3; ++a) e = std::max(e, std::max(std::fabs(q.lo[a][j]), std::fabs(q.hi[a][j])));
                }
                ext[(size_t)p] = e;
            });
            f.q_extent = *std::max_element(ext.begin(), ext.end()) * 1.0000001f;
        };
        measure_extent();
        stage("4-wide trees");
        {
            uint32_t r = f.world;
            if (RT_REF_KIND(r) == RT_REF_LIST && f.lists[RT_REF_INDEX(r)].count == 1) r = f.list_items[f.lists[RT_REF_INDEX(r)].first];
            auto it = bvh_prim_only.find(r);
            if (RT_REF_KIND(r) == RT_REF_NODE && it != bvh_prim_only.end() && it->second && f.wnodes[RT_REF_INDEX(r)].qroot) {
                // the world is one BVH over primitives: the dedicated kernel traverses its 4-wide copy
                f.fast_root = make_ref(REF_QNODE, f.wnodes[RT_REF_INDEX(r)].qroot - 1);
                f.fast_stack = (uint32_t)need_q[RT_REF_INDEX(r)] + 1;
                f.quant_inflation = quantise_qnodes();
                stage("8-bit boxes");
                if (f.quant_inflation > 1.25) f.cqnodes.clear();
            }
        }
        if (hoisted_root >= 0 && f.fast_root == RT_REF_NONE && f.wnodes[(size_t)hoisted_root].qroot) {
            f.fast_root = make_ref(REF_QNODE, f.wnodes[(size_t)hoisted_root].qroot - 1);
            f.fast_stack = (uint32_t)need_q[(uint32_t)hoisted_root] + 1;
            if (hoist_whole_world) { // as if the world were that one BVH
                f.quant_inflation = quantise_qnodes();
                if (f.quant_inflation > 1.25) f.cqnodes.clear();
            } else {
                f.fast_partial = true;
                const DList w = f.lists[RT_REF_INDEX(f.world)];
                f.world_rest = make_ref(RT_REF_LIST, (uint32_t)f.lists.size());
                f.lists.push_back(DList{w.first + 1u, w.count - 1u}); // shares the world's items
            }
        }
        pack_fast_tree();
        stage("packed tree");
        if (!need_memo_kept) need_node.assign(f.nodes.size(), -1);
        f.max_stack = (uint32_t)need(f.world) + 1;
        if (f.max_stack > (uint32_t)STACK_MAX) {
            // too deep for the general kernel's stack with 4-wide nodes (up to three deferred children per two
            // binary levels): it keeps the binary nodes; the dedicated kernel has its own spilling stack
            for (uint32_t root : bvh_roots) f.wnodes[root].qroot = 0;
            need_node.assign(f.nodes.size(), -1);
            need_wide_memo.assign(f.qnodes.size(), -1);
            f.max_stack = (uint32_t)need(f.world) + 1;
        }
        if (f.max_stack > (uint32_t)STACK_MAX) return fail(RT_E_TOO_LARGE, "scene needs a deeper traversal stack than the kernels carry (32 entries)");
        stage("stack depth");
        return RT_OK;
    }
};

} // namespace

int flatten_scene(const rt_scene_desc* desc, FlatScene& out, std::string& err, uint64_t device_bvh_min, uint64_t fast_tree_min) {
    Flattener fl(desc, out, err);
    fl.device_bvh_min = device_bvh_min;
    fl.fast_tree_min = fast_tree_min;
    return fl.run();
}

int camera_initialize(const rt_camera* c, rt_camera_derived* d, std::string& err) { // camera.rs:161-205
    if (!c || !d) { err = "null camera"; return RT_E_INVALID; }
    if (c->image_width < 1 || c->samples_per_pixel < 1) { err = "camera: image_width and samples_per_pixel must be >= 1"; return RT_E_INVALID; }
    double hgt = (double)c->image_width / c->aspect_ratio;
    int32_t image_height = (hgt != hgt) ? 0 : (hgt >= 2147483647.0 ? INT32_MAX : (hgt <= -2147483648.0 ? INT32_MIN : (int32_t)hgt));
    d->image_height = image_height < 1 ? 1 : image_height;
    d->sqrt_spp = (int32_t)std::sqrt((double)c->samples_per_pixel);
    d->pixel_samples_scale = 1.0 / (double)(d->sqrt_spp * d->sqrt_spp);
    d->recip_sqrt_spp = 1.0 / (double)d->sqrt_spp;
    V3 lookfrom = v3(c->lookfrom), lookat = v3(c->lookat), vup = v3(c->vup);
    V3 center = lookfrom;
    double theta = c->vfov * PI / 180.0; // rtweekend.rs:11-13
    double h = std::tan(theta / 2.0);
    double viewport_height = 2.0 * h * c->focus_dist;
    double viewport_width = viewport_height * ((double)c->image_width / (double)d->image_height);
    V3 w = unit3(sub(lookfrom, lookat));
    V3 u = unit3(cross3(vup, w));
    V3 v = cross3(w, u);
    V3 viewport_u = scale(viewport_width, u);
    V3 viewport_v = scale(-viewport_height, v);
    V3 pdu = divs(viewport_u, (double)c->image_width);
    V3 pdv = divs(viewport_v, (double)d->image_height);
    V3 upper_left = sub(sub(sub(center, scale(c->focus_dist, w)), divs(viewport_u, 2.0)), divs(viewport_v, 2.0));
    V3 pixel00 = add(upper_left, scale(0.5, add(pdu, pdv)));
    double defocus_radius = c->focus_dist * std::tan((c->defocus_angle / 2.0) * PI / 180.0);
    put(d->center, center); put(d->pixel00_loc, pixel00); put(d->pixel_delta_u, pdu); put(d->pixel_delta_v, pdv);
    put(d->defocus_disk_u, scale(defocus_radius, u)); put(d->defocus_disk_v, scale(defocus_radius, v));
    return RT_OK;
}

} // namespace rt
