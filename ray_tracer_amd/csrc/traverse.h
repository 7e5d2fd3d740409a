// traverse.h -- per-lane stack traversal of the flattened scene (closest hit) and
// reconstruction of the full hit record.
//
// One ray per lane.  The traversal stack lives in LDS, laid out [entry][lane] so
// the 64 lanes of a wave touch 64 consecutive banks.  The visit order is the
// reference's: BvhNode::hit tests its box with the interval it is handed, then
// left, then right with the interval shortened to the closest hit so far
// (bvh.rs:118-140); HittableList::hit scans its members in order
// (hittable_list.rs:45-61).  Because the interval every callee receives is always
// [tmin, closest-so-far], an explicit stack with "push right, descend left"
// reproduces the recursion exactly.
//
// During traversal only (t, primitive, transform context) of the closest hit are
// kept; p, normal, front_face, u, v are recomputed afterwards by `finalize` with
// the same arithmetic the reference's hit() used, which yields the same bits.
#ifndef RT_TRAVERSE_H
#define RT_TRAVERSE_H

#include "device_math.h"

namespace rt {

constexpr int BLOCK = 256; // threads per workgroup of every traversal kernel

// Where the world-space ray of this lane lives (SoA arrays + index): transforms
// re-derive the context ray from it when a Translate/RotateY subtree is left.
struct WorldRay {
    const double* ox; const double* oy; const double* oz;
    const double* dx; const double* dy; const double* dz;
    uint64_t i;
};

struct Best {
    double t;      // closest hit so far == upper end of the interval handed to every callee
    uint32_t ref;  // primitive (or medium) that produced it, RT_REF_NONE if none
    int32_t ctx;   // transform context it was found in
};

// Translate::hit / RotateY::hit ray transform (hittable.rs:80-81, 147-157).
RT_DEV void apply_xform(const DXform& x, RayD& r) {
    if (x.kind == 0) {
        r.ox = r.ox - x.a[0]; r.oy = r.oy - x.a[1]; r.oz = r.oz - x.a[2];
    } else {
        double s = x.a[0], c = x.a[1];
        double nox = c * r.ox - s * r.oz, noz = s * r.ox + c * r.oz;
        double ndx = c * r.dx - s * r.dz, ndz = s * r.dx + c * r.dz;
        r.ox = nox; r.oz = noz; r.dx = ndx; r.dz = ndz;
    }
}

// Ray of this lane in context `ctx`: the world ray pushed through the chain of
// transforms from the outermost to `ctx`, one rounding per reference operation.
// `r` holds the world ray (origin and direction) on entry.
// The transform records of a scene, with the first ones copied into LDS by kernels that read them per lane and per hit
// (shading: a record fetched from memory by a handful of lanes costs the whole wave a memory round trip, and
// completing the record of a hit inside Translate(RotateY(..)) takes seven of them one after the other; measured
// on the Cornell box: a quarter of the shading kernel's wave time).  n_lds = 0: no copy, every record from memory.
constexpr int XF_LDS = 64;
typedef const __attribute__((address_space(3))) DXform* LdsXforms;
struct XfTable {
    const DXform* mem;
    LdsXforms lds;
    int32_t n_lds;
};
RT_DEV XfTable xf_from_memory(const DScene& S) { return XfTable{S.xforms, nullptr, 0}; }
RT_DEV DXform xf_get(const XfTable& t, int32_t i) {
    if (i < t.n_lds) {
        DXform x;
        x.kind = t.lds[i].kind; x.parent = t.lds[i].parent; x.child = t.lds[i].child; x.depth = t.lds[i].depth;
        x.a[0] = t.lds[i].a[0]; x.a[1] = t.lds[i].a[1]; x.a[2] = t.lds[i].a[2];
        return x;
    }
    return t.mem[i];
}
// Fills a workgroup's copy (every thread calls it; ends with a barrier).
RT_DEV XfTable xf_fill_lds(const DScene& S, DXform* table) {
    const int32_t n = (int32_t)(S.n_xforms < (uint32_t)XF_LDS ? S.n_xforms : (uint32_t)XF_LDS);
    for (int32_t i = (int32_t)threadIdx.x; i < n; i += (int32_t)blockDim.x) table[i] = S.xforms[i];
    __syncthreads();
    return XfTable{S.xforms, (LdsXforms)table, n};
}

RT_DEV void world_to_context(const XfTable& xt, int32_t ctx, RayD& r) {
    if (ctx != 0) {
        int depth = xf_get(xt, ctx - 1).depth;
        for (int level = 1; level <= depth; ++level) {
            int32_t c = ctx;
            for (int k = depth; k > level; --k) c = xf_get(xt, c - 1).parent;
            apply_xform(xf_get(xt, c - 1), r);
        }
    }
    set_inverse(r);
}
RT_DEV void world_to_context(const DScene& S, int32_t ctx, RayD& r) { world_to_context(xf_from_memory(S), ctx, r); }
RT_DEV void ray_in_context(const DScene& S, const WorldRay& w, int32_t ctx, RayD& r) {
    r.ox = w.ox[w.i]; r.oy = w.oy[w.i]; r.oz = w.oz[w.i];
    r.dx = w.dx[w.i]; r.dy = w.dy[w.i]; r.dz = w.dz[w.i];
    world_to_context(S, ctx, r);
}

struct NodeLoad { double b[6]; uint32_t left, right, flags, pad; };
RT_DEV void load_node(const DNode* nodes, uint32_t idx, NodeLoad& n) {
    // 64-byte record, four 16-byte loads per lane
    const double2* p = reinterpret_cast<const double2*>(nodes + idx);
    double2 a = p[0], b = p[1], c = p[2];
    const uint4 d = *reinterpret_cast<const uint4*>(p + 3);
    n.b[0] = a.x; n.b[1] = a.y; n.b[2] = b.x; n.b[3] = b.y; n.b[4] = c.x; n.b[5] = c.y;
    n.left = d.x; n.right = d.y; n.flags = d.z; n.pad = d.w;
}

struct WNodeLoad { float lb[6]; float rb[6]; uint32_t left, right, meta, pad; };
RT_DEV void load_wnode(const WNode* wnodes, uint32_t idx, WNodeLoad& n) {
    const float4* p = reinterpret_cast<const float4*>(wnodes + idx);
    float4 a = p[0], b = p[1], c = p[2];
    const uint4 d = *reinterpret_cast<const uint4*>(p + 3);
    n.lb[0] = a.x; n.lb[1] = a.y; n.lb[2] = a.z; n.lb[3] = a.w; n.lb[4] = b.x; n.lb[5] = b.y;
    n.rb[0] = b.z; n.rb[1] = b.w; n.rb[2] = c.x; n.rb[3] = c.y; n.rb[4] = c.z; n.rb[5] = c.w;
    n.left = d.x; n.right = d.y; n.meta = d.z; n.pad = d.w;
}

RT_DEV void sort2(float& ka, uint32_t& ra, float& kb, uint32_t& rb) { // ascending by key
    const bool sw = kb < ka;
    const float k0 = sw ? kb : ka, k1 = sw ? ka : kb;
    const uint32_t r0 = sw ? rb : ra, r1 = sw ? ra : rb;
    ka = k0; kb = k1; ra = r0; rb = r1;
}

// One primitive against the lane's ray over (tmin, best.t]; a closer hit replaces `best`.
RT_DEV void test_primitive(const DScene& S, uint32_t ref, const RayD& ray, int32_t ctx, double time, double tmin, Best& best) {
    const uint32_t kind = ref >> 28, idx = ref & 0x0FFFFFFFu;
    if (kind == RT_REF_TRI) {
        const DTri* tp = S.tris + idx;
        double t, u, v;
        if (tri_test(d3p(tp->v0), d3p(tp->v1), d3p(tp->v2), ray, tmin, best.t, t, u, v)) { best.t = t; best.ref = ref; best.ctx = ctx; }
    } else if (kind == RT_REF_SPHERE) {
        const DSphere s = S.spheres[idx];
        double t;
        if (sphere_test(s, ray, time, tmin, best.t, t)) { best.t = t; best.ref = ref; best.ctx = ctx; }
    } else { // RT_REF_QUAD
        const DQuad* q = S.quads + idx;
        double t, al, be;
        if (quad_test(*q, ray, tmin, best.t, t, al, be)) { best.t = t; best.ref = ref; best.ctx = ctx; }
    }
}

// ---- wave-uniform records --------------------------------------------------------------------------
// When every active lane of a wave wants the same record (a flat world such as the Cornell box: every ray
// tests every wall, in the same order), the record goes through the scalar cache into scalar registers: one
// request instead of 64.  It matters more than the request count suggests.  A compute unit's vector L1
// returns data in request order, and the same waves stream path records from HBM, so a vector load of a
// perfectly cached wall was measured to wait ~1900 cycles behind other waves' misses; the scalar path does
// not share that queue.
RT_DEV bool wave_uniform(uint32_t x) {
    return __builtin_amdgcn_ballot_w64(x != (uint32_t)__builtin_amdgcn_readfirstlane((int)x)) == 0ull;
}
RT_DEV uint32_t first_lane(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
template <typename T> RT_DEV T uniform_load(const T* p) { // p must be wave-uniform
    static_assert(sizeof(T) % 4 == 0, "records are whole dwords");
    typedef const uint32_t __attribute__((address_space(4))) * ConstWords;
    ConstWords cp = (ConstWords)(uintptr_t)p;
    union { T v; uint32_t w[sizeof(T) / 4]; } u;
#pragma unroll
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) u.w[k] = cp[k];
    return u.v;
}
RT_DEV void test_uniform_primitive(const DScene& S, uint32_t item, const RayD& ray, int32_t ctx, double time, double tmin, Best& best) {
    const uint32_t k = item >> 28, i = item & 0x0FFFFFFFu; // wave-uniform
    if (k == RT_REF_QUAD) {
        const DQuad q = uniform_load(S.quads + i);
        double t;
        if (quad_eval(q, ray, t) && tmin <= t && t <= best.t) { best.t = t; best.ref = item; best.ctx = ctx; }
    } else if (k == RT_REF_SPHERE) {
        const DSphere sp = uniform_load(S.spheres + i);
        double t;
        if (sphere_test(sp, ray, time, tmin, best.t, t)) { best.t = t; best.ref = item; best.ctx = ctx; }
    } else { // RT_REF_TRI
        const DTri tr = uniform_load(S.tris + i);
        double t, u, v;
        if (tri_test(d3p(tr.v0), d3p(tr.v1), d3p(tr.v2), ray, tmin, best.t, t, u, v)) { best.t = t; best.ref = item; best.ctx = ctx; }
    }
}

#ifdef RT_PHASE_STATS
// wave cycles per kind of traversal step (stats build only): 0 wide/4-wide node, 1 exact node, 2 single primitive,
// 3 list walk, 4 transform enter/leave, 5 medium bookkeeping, 6 pop
__shared__ unsigned long long s_step_cycles[8];
__shared__ uint32_t s_lane_iters[256]; // loop iterations of each lane's current traversal
#define RT_STEP_BEGIN() const unsigned long long step_t0_ = __builtin_amdgcn_s_memtime()
#define RT_STEP_END(k) do { if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) atomicAdd(&s_step_cycles[k], (unsigned long long)__builtin_amdgcn_s_memtime() - step_t0_); } while (0)
// Sections of the shading kernel (stats build only): for every execution of a section by a wave, the lanes that took
// part, and the wave's cycles from its begin to its end.  A section must end with the lanes it began with.
constexpr int RT_SECTIONS = 24;
__shared__ unsigned long long s_sec_lanes[RT_SECTIONS], s_sec_runs[RT_SECTIONS], s_sec_cycles[RT_SECTIONS];
#define RT_SEC_BEGIN(k) const unsigned long long sec_m_##k = __builtin_amdgcn_ballot_w64(true); const unsigned long long sec_t_##k = __builtin_amdgcn_s_memtime()
#define RT_SEC_END(k) do { if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(sec_m_##k)) { atomicAdd(&s_sec_lanes[k], (unsigned long long)__popcll(sec_m_##k)); \
        atomicAdd(&s_sec_runs[k], 1ull); atomicAdd(&s_sec_cycles[k], (unsigned long long)__builtin_amdgcn_s_memtime() - sec_t_##k); } } while (0)
#else
#define RT_STEP_BEGIN() do { } while (0)
#define RT_STEP_END(k) do { } while (0)
#define RT_SEC_BEGIN(k) do { } while (0)
#define RT_SEC_END(k) do { } while (0)
#endif

// Whose surface coordinates the reference's HitRecord would hold.  ConstantMedium::hit writes t, p, normal,
// front_face and mat but never u,v (constant_medium.rs:96-101), so a medium hit reports the u,v of the last primitive
// that was written into the same record: HittableList::hit works on a record of its own (`temp_rec`, default u = v = 0)
// and copies it out on every hit (hittable_list.rs:46-57); BvhNode, Translate and RotateY hand their caller's record
// down.  Only a phase function over an image texture reads these values, so only scenes that have one pay for the
// book-keeping (UV = true): the record of the current frame is named by the primitive hit that last wrote it.
constexpr uint32_t UV_UNWRITTEN = 0xFFFFFFFEu; // stands in for Best::ref during a step: no primitive accepted a hit yet
struct UvTrack {
    uint32_t ref;  // RT_REF_NONE: the default record, u = v = 0
    int32_t ctx;
    double t;
    bool any;      // something was hit since the current list frame was entered (the list will return true)
};

// axis_order (exact nodes only): enter the child on the ray's near side of the node's split axis first.  Every box is
// still tested by slab_test exactly as the reference does; only the order of two siblings changes, which can change
// nothing but which of several primitives hit at exactly equal t is reported.
// MEDIA: levels of constant media the walk resolves: 0 none in the scene, 1 media whose boundaries hold none, 2 a
// medium inside the boundary of a medium (the boundary walks of level k are walks of level k - 1).
template <int MEDIA, bool COUNT, bool ORDERED, bool WAVE_LISTS = true, bool KEEP_ENTRY_RAY = WAVE_LISTS, bool UV = false>
__device__ __forceinline__ void traverse(const DScene& S, uint32_t root, RayD ray, int32_t ctx, const WorldRay& w,
                                         double time, double tmin, Best& best, uint32_t* stk, int sp0, Rng* rng,
                                         uint32_t& n_nodes, uint32_t& n_prims, bool axis_order = false, UvTrack* uvt = nullptr) {
    int sp = sp0;
    uint32_t cur = root;
    const RayD entry_ray = ray;
    const int32_t entry_ctx = ctx;
    for (;;) {
#ifdef RT_PHASE_STATS
        if (MEDIA) s_lane_iters[threadIdx.x] += 1u;
#endif
        const uint32_t kind = cur >> 28, idx = cur & 0x0FFFFFFFu;
        bool pop = true;
        bool wide_done = false;
        // (UV) a primitive that accepts a hit in this step writes the whole record and so becomes the source of the
        // frame's u,v -- also when it writes the values the record already holds (the same primitive accepted again at
        // the same t through a second frame, e.g. the duplicated child of a one-object span, bvh.rs:41-44), which a
        // comparison of the record before and after would miss: the step starts from a reference no test writes
        const uint32_t uv_ref0 = best.ref;
        if (UV && kind != RT_REF_MEDIUM && kind != REF_LISTEND) best.ref = UV_UNWRITTEN;
        if (ORDERED && kind == RT_REF_NODE) {
            RT_STEP_BEGIN();
            WNodeLoad wn;
            load_wnode(S.wnodes, idx, wn);
            if (wn.pad) { // root of a BVH without media: continue in its 4-wide copy
                wide_done = true;
                cur = make_ref(REF_QNODE, wn.pad - 1u);
                pop = false;
            } else if (!(wn.meta & 1u)) {
                wide_done = true;
                double nl = 0.0, nr = 0.0;
                const bool hl = slab_test_wide(wn.lb, ray, tmin, best.t, nl);
                const bool hr = (wn.meta & 2u) ? false : slab_test_wide(wn.rb, ray, tmin, best.t, nr);
                if (COUNT) n_nodes += (wn.meta & 2u) ? 1u : 2u;
                if (hl && hr) {
                    const bool right_first = nr < nl;
                    stk[sp * BLOCK] = right_first ? wn.left : wn.right; ++sp;
                    cur = right_first ? wn.right : wn.left;
                    pop = false;
                } else if (hl) { cur = wn.left; pop = false; }
                else if (hr) { cur = wn.right; pop = false; }
            }
            RT_STEP_END(0);
        }
        if (wide_done) {
        } else if (ORDERED && kind == REF_QNODE) {
            // 4-wide node: conservative f32 slab test of the four child boxes, nearest child first.  The
            // children may be anything (primitives, lists, instances, further nodes).
            RT_STEP_BEGIN();
            const float4* row = reinterpret_cast<const float4*>(S.qnodes + idx);
            const float4 lox = row[0], loy = row[1], loz = row[2];
            const float4 hix = row[3], hiy = row[4], hiz = row[5];
            const float4 chf = row[6];
            Ray32 q;
            make_ray32(ray, S.q_extent, q);
            const float tmin32 = f32_round_down(tmin), best32 = f32_round_up(best.t);
            const float lx[4] = {lox.x, lox.y, lox.z, lox.w}, hx[4] = {hix.x, hix.y, hix.z, hix.w};
            const float ly[4] = {loy.x, loy.y, loy.z, loy.w}, hy[4] = {hiy.x, hiy.y, hiy.z, hiy.w};
            const float lz[4] = {loz.x, loz.y, loz.z, loz.w}, hz[4] = {hiz.x, hiz.y, hiz.z, hiz.w};
            uint32_t rf[4] = {__float_as_uint(chf.x), __float_as_uint(chf.y), __float_as_uint(chf.z), __float_as_uint(chf.w)};
            float key[4];
            int nh = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float nx = q.neg[0] ? hx[j] : lx[j], fx = q.neg[0] ? lx[j] : hx[j];
                const float ny = q.neg[1] ? hy[j] : ly[j], fy = q.neg[1] ? ly[j] : hy[j];
                const float nz = q.neg[2] ? hz[j] : lz[j], fz = q.neg[2] ? lz[j] : hz[j];
                const float tn = fmaxf(fmaxf(__builtin_fmaf(nx, q.inv[0], q.cn[0]), __builtin_fmaf(ny, q.inv[1], q.cn[1])), fmaxf(__builtin_fmaf(nz, q.inv[2], q.cn[2]), tmin32));
                const float tf = fminf(fminf(__builtin_fmaf(fx, q.inv[0], q.cf[0]), __builtin_fmaf(fy, q.inv[1], q.cf[1])), fminf(__builtin_fmaf(fz, q.inv[2], q.cf[2]), best32));
                const bool hit = (tn <= tf) && (rf[j] != RT_REF_NONE);
                key[j] = hit ? tn : __builtin_inff();
                nh += hit ? 1 : 0;
                if (COUNT) n_nodes += rf[j] != RT_REF_NONE ? 1u : 0u;
            }
            if (nh > 0) {
                sort2(key[0], rf[0], key[1], rf[1]);
                sort2(key[2], rf[2], key[3], rf[3]);
                sort2(key[0], rf[0], key[2], rf[2]);
                sort2(key[1], rf[1], key[3], rf[3]);
                sort2(key[1], rf[1], key[2], rf[2]);
                if (nh > 3) { stk[sp * BLOCK] = rf[3]; ++sp; }
                if (nh > 2) { stk[sp * BLOCK] = rf[2]; ++sp; }
                if (nh > 1) { stk[sp * BLOCK] = rf[1]; ++sp; }
                cur = rf[0];
                pop = false;
            }
            RT_STEP_END(0);
        } else if (kind == RT_REF_NODE) {
            RT_STEP_BEGIN();
            NodeLoad n;
            load_node(S.nodes, idx, n);
            if (COUNT) ++n_nodes;
            if (slab_test(n.b, ray, tmin, best.t)) {
                // A one-object span stores the object as both children (bvh.rs:41-44).  Testing it
                // a second time cannot change the record unless it draws random numbers, so the
                // duplicate is visited only when the subtree holds a constant medium.
                uint32_t first = n.left, second = n.right;
                if (axis_order && !(n.flags & 1u)) { // a subtree with a medium keeps the reference order (its random draws)
                    const uint32_t axis = (n.flags >> 1) & 3u;
                    const double da = axis == 0u ? ray.dx : (axis == 1u ? ray.dy : ray.dz);
                    if (da < 0.0) { first = n.right; second = n.left; }
                }
                if (n.right != n.left || (n.flags & 1u)) { stk[sp * BLOCK] = second; ++sp; }
                cur = first;
                pop = false;
            }
            RT_STEP_END(1);
        } else if (kind == RT_REF_TRI || kind == RT_REF_SPHERE || kind == RT_REF_QUAD) {
            RT_STEP_BEGIN();
            if (COUNT) ++n_prims;
            test_primitive(S, cur, ray, ctx, time, tmin, best);
            RT_STEP_END(2);
        } else if (ORDERED && kind == REF_QUADS) {
            // 2..16 consecutive quads standing for the list that holds them: same tests in the same order
            // (hittable_list.rs:46-57), without reading the list
            RT_STEP_BEGIN();
            const uint32_t first = idx & 0x00FFFFFFu, count = (idx >> 24) + 1u;
            if (COUNT) n_prims += count;
            if (WAVE_LISTS && wave_uniform(cur)) {
                const uint32_t ufirst = first_lane(first), ucount = first_lane(count);
                for (uint32_t k = 0; k < ucount; ++k) test_uniform_primitive(S, make_ref(RT_REF_QUAD, ufirst + k), ray, ctx, time, tmin, best);
            } else {
                uint32_t k = 0;
                for (; k + 1u < count; k += 2u) { // pairs: both evaluated before either is accepted, acceptance in list order
                    double ta, tb;
                    const bool ia = quad_eval(S.quads[first + k], ray, ta);
                    const bool ib = quad_eval(S.quads[first + k + 1u], ray, tb);
                    if (ia && tmin <= ta && ta <= best.t) { best.t = ta; best.ref = make_ref(RT_REF_QUAD, first + k); best.ctx = ctx; }
                    if (ib && tmin <= tb && tb <= best.t) { best.t = tb; best.ref = make_ref(RT_REF_QUAD, first + k + 1u); best.ctx = ctx; }
                }
                if (k < count) test_primitive(S, make_ref(RT_REF_QUAD, first + k), ray, ctx, time, tmin, best);
            }
            RT_STEP_END(3);
        } else if (UV && kind == REF_LISTEND) {
            // HittableList::hit returns: its caller's record took a copy of temp_rec at every hit, so it now holds the
            // list's u,v if the list hit anything, and what it held before otherwise
            sp -= 5;
            if (!uvt->any) {
                uvt->ref = stk[sp * BLOCK];
                uvt->ctx = (int32_t)stk[(sp + 1) * BLOCK];
                uvt->t = __hiloint2double((int)stk[(sp + 3) * BLOCK], (int)stk[(sp + 2) * BLOCK]);
            }
            uvt->any = uvt->any || stk[(sp + 4) * BLOCK] != 0u;
        } else if (kind == RT_REF_LIST || kind == REF_LISTCONT) {
            RT_STEP_BEGIN();
            if (UV && kind == RT_REF_LIST) { // a fresh temp_rec: remember the caller's, to be settled when the list is done
                stk[sp * BLOCK] = uvt->ref;
                stk[(sp + 1) * BLOCK] = (uint32_t)uvt->ctx;
                stk[(sp + 2) * BLOCK] = (uint32_t)__double2loint(uvt->t);
                stk[(sp + 3) * BLOCK] = (uint32_t)__double2hiint(uvt->t);
                stk[(sp + 4) * BLOCK] = uvt->any ? 1u : 0u;
                stk[(sp + 5) * BLOCK] = make_ref(REF_LISTEND, 0u);
                sp += 6;
                uvt->ref = RT_REF_NONE; uvt->any = false;
            }
            uint32_t pos = 0;
            if (kind == REF_LISTCONT) { --sp; pos = stk[sp * BLOCK]; }
            bool walked = false;
            if (WAVE_LISTS && __builtin_amdgcn_ballot_w64(idx != first_lane(idx) || pos != first_lane(pos)) == 0ull) {
                // every active lane stands at the same place of the same list
                walked = true;
                const uint32_t uidx = first_lane(idx);
                uint32_t upos = first_lane(pos);
                const DList ul = uniform_load(S.lists + uidx);
                while (upos < ul.count) {
                    const uint32_t item = uniform_load(S.list_items + ul.first + upos);
                    const uint32_t ik = item >> 28;
                    if (ik == RT_REF_TRI || ik == RT_REF_SPHERE || ik == RT_REF_QUAD) {
                        if (COUNT) ++n_prims;
                        test_uniform_primitive(S, item, ray, ctx, time, tmin, best);
                        ++upos;
                        continue;
                    }
                    if (upos + 1u < ul.count) {
                        stk[sp * BLOCK] = upos + 1u; ++sp;
                        stk[sp * BLOCK] = make_ref(REF_LISTCONT, uidx); ++sp;
                    }
                    cur = item;
                    pop = false;
                    break;
                }
            }
            if (!walked) {
            const DList l = S.lists[idx];
            // Items are visited in list order (hittable_list.rs:46-57).  Primitives are tested right here;
            // the first item that is something else is entered through the stack, with the rest of the
            // list left behind as a continuation.
            while (pos < l.count) {
                const uint32_t item = S.list_items[l.first + pos];
                const uint32_t ik = item >> 28;
                if (ik == RT_REF_QUAD && pos + 1 < l.count) {
                    // two quads in a row (the walls of a room, the sides of a box): both are evaluated before
                    // either is accepted, so their f64 chains overlap; acceptance stays in list order
                    const uint32_t item2 = S.list_items[l.first + pos + 1];
                    if ((item2 >> 28) == RT_REF_QUAD) {
                        if (COUNT) n_prims += 2;
                        double ta, tb;
                        const bool ia = quad_eval(S.quads[item & 0x0FFFFFFFu], ray, ta);
                        const bool ib = quad_eval(S.quads[item2 & 0x0FFFFFFFu], ray, tb);
                        if (ia && tmin <= ta && ta <= best.t) { best.t = ta; best.ref = item; best.ctx = ctx; }
                        if (ib && tmin <= tb && tb <= best.t) { best.t = tb; best.ref = item2; best.ctx = ctx; }
                        pos += 2;
                        continue;
                    }
                }
                if (ik == RT_REF_TRI || ik == RT_REF_SPHERE || ik == RT_REF_QUAD) {
                    if (COUNT) ++n_prims;
                    test_primitive(S, item, ray, ctx, time, tmin, best);
                    ++pos;
                    continue;
                }
                if (pos + 1 < l.count) {
                    stk[sp * BLOCK] = pos + 1; ++sp;
                    stk[sp * BLOCK] = make_ref(REF_LISTCONT, idx); ++sp;
                }
                cur = item;
                pop = false;
                break;
            }
            }
            RT_STEP_END(3);
        } else if (kind == RT_REF_XFORM) {
            RT_STEP_BEGIN();
            DXform x;
            if (WAVE_LISTS && wave_uniform(idx)) x = uniform_load(S.xforms + first_lane(idx));
            else x = S.xforms[idx];
            // Leaving this transform restores the enclosing context.  When the entry below is already such a
            // restore (this transform is the last thing its parent had to do: Translate(RotateY(box))), that
            // one overrides ours, so ours is not pushed.
            const bool restore_pending = sp > sp0 && (stk[(sp - 1) * BLOCK] >> 28) == REF_POPX;
            if (!restore_pending) { stk[sp * BLOCK] = make_ref(REF_POPX, (uint32_t)ctx); ++sp; }
            apply_xform(x, ray);
            set_inverse(ray);
            ctx = (int32_t)idx + 1;
            cur = x.child;
            pop = false;
            RT_STEP_END(4);
        } else if (kind == REF_POPX) {
            RT_STEP_BEGIN();
            ctx = (int32_t)idx;
            if (KEEP_ENTRY_RAY && ctx == entry_ctx) ray = entry_ray; // the ray this call was entered with, kept in registers
            else ray_in_context(S, w, ctx, ray);
            RT_STEP_END(4);
        } else if (kind == RT_REF_MEDIUM) {
            if constexpr (MEDIA > 0) { // ConstantMedium::hit (constant_medium.rs:48-104)
                const DMedium m = S.media[idx];
                if (COUNT) ++n_prims;
                double t1 = 0.0, t2 = 0.0;
                bool ok = true;
                for (int pass = 0; pass < 2 && ok; ++pass) {
                    Best b; b.t = D_INF; b.ref = RT_REF_NONE; b.ctx = 0;
                    double lo = pass == 0 ? -D_INF : t1 + 0.0001; // Interval::UNIVERSE, then [rec1.t + 0.0001, inf)
                    // (a medium inside the boundary draws from the same stream, once per boundary walk, as boundary.hit does)
                    // (rec1 / rec2 are records of their own and are dropped: no history to keep in the boundary walks)
                    traverse<MEDIA - 1, COUNT, ORDERED, WAVE_LISTS, KEEP_ENTRY_RAY, false>(S, m.boundary, ray, ctx, w, time, lo, b, stk, sp, rng, n_nodes, n_prims, axis_order);
                    if (b.ref == RT_REF_NONE) ok = false;
                    else if (pass == 0) t1 = b.t; else t2 = b.t;
                }
                if (ok) {
                    double rec1_t = t1 < tmin ? tmin : t1;
                    double rec2_t = t2 > best.t ? best.t : t2;
                    if (!(rec1_t >= rec2_t)) {
                        if (rec1_t < 0.0) rec1_t = 0.0;
                        double ray_length = length(d3(ray.dx, ray.dy, ray.dz));
                        double distance_inside_boundary = (rec2_t - rec1_t) * ray_length;
                        double hit_distance = m.neg_inv_density * log(rng->next());
                        if (!(hit_distance > distance_inside_boundary)) {
                            best.t = rec1_t + hit_distance / ray_length;
                            best.ref = cur; best.ctx = ctx;
                            if (UV) uvt->any = true; // a hit, but u,v stay as they are
                        }
                    }
                }
            }
        }
        if (UV && kind != RT_REF_MEDIUM && kind != REF_LISTEND) {
            if (best.ref != UV_UNWRITTEN) { uvt->ref = best.ref; uvt->ctx = best.ctx; uvt->t = best.t; uvt->any = true; }
            else best.ref = uv_ref0;
        }
        if (pop) {
            RT_STEP_BEGIN();
            if (sp == sp0) break;
            --sp;
            cur = stk[sp * BLOCK];
            RT_STEP_END(6);
        }
    }
}

// Full HitRecord (hittable.rs:14-22) of the closest hit found by `traverse`.
struct HitRec {
    double t, u, v;
    D3 p, normal;
    int32_t mat;
    uint32_t prim;  // reported primitive id: (kind << 28) | index in the scene description
    bool front_face;
};

RT_DEV void set_face_normal(HitRec& h, D3 dir, D3 outward_normal) { // hittable.rs:25-33
    h.front_face = dot(dir, outward_normal) < 0.0;
    h.normal = h.front_face ? outward_normal : -outward_normal;
}

// want_uv = false skips the surface coordinates (sphere: acos + atan2, sphere.rs:47-55; quad and triangle: the fetch of
// everything but normal and material) for callers that know nothing reads them.
// `r` holds the world ray on entry (it is taken into the context of the hit here).
RT_DEV void finalize(const DScene& S, const XfTable& xt, RayD r, double time, const Best& b, HitRec& h, bool want_uv = true) {
    RT_SEC_BEGIN(1);
    world_to_context(xt, b.ctx, r);
    RT_SEC_END(1);
    RT_SEC_BEGIN(2);
    const D3 o = d3(r.ox, r.oy, r.oz), d = d3(r.dx, r.dy, r.dz);
    const uint32_t kind = b.ref >> 28, idx = b.ref & 0x0FFFFFFFu;
    h.t = b.t;
    h.p = o + b.t * d; // r.at(t) (ray.rs:54-56)
    h.u = 0.0; h.v = 0.0;
    h.prim = b.ref;
    if (kind == RT_REF_SPHERE) { // sphere.rs:101-108
        const DSphere s = S.spheres[idx];
        D3 center = d3(s.c[0], s.c[1], s.c[2]) + time * d3(s.dc[0], s.dc[1], s.dc[2]);
        D3 outward_normal = (h.p - center) / s.radius;
        set_face_normal(h, d, outward_normal);
        if (want_uv) {
            double theta = acos(-outward_normal.y); // sphere.rs:47-55
            double phi = atan2(-outward_normal.z, outward_normal.x) + D_PI;
            h.u = phi / (2.0 * D_PI);
            h.v = theta / D_PI;
        }
        h.mat = s.mat;
    } else if (kind == RT_REF_QUAD) { // quad.rs:95-108
        const DQuad* q = S.quads + idx;
        if (want_uv) {
            D3 hp = h.p - d3p(q->q);
            h.u = dot(d3p(q->w), cross(hp, d3p(q->v)));
            h.v = dot(d3p(q->w), cross(d3p(q->u), hp));
        }
        set_face_normal(h, d, d3p(q->n));
        h.mat = q->mat;
    } else if (kind == RT_REF_TRI) { // triangle.rs:155-196
        const DTri* tp = S.tris + idx;
        if (want_uv) {
            D3 v0 = d3p(tp->v0), v1 = d3p(tp->v1), v2 = d3p(tp->v2);
            D3 edge1 = v1 - v0, edge2 = v2 - v0;
            D3 hh = cross(d, edge2);
            double f = 1.0 / dot(edge1, hh);
            D3 s = o - v0;
            h.u = f * dot(s, hh);
            h.v = f * dot(d, cross(s, edge1));
        }
        set_face_normal(h, d, d3p(tp->n));
        h.mat = tp->mat;
    } else { // constant medium (constant_medium.rs:96-101); u,v are not written by the reference
        const DMedium m = S.media[idx];
        h.normal = d3(1.0, 0.0, 0.0);
        h.front_face = true;
        h.mat = m.mat;
        h.prim = make_ref(RT_REF_MEDIUM, (uint32_t)m.object);
    }
    RT_SEC_END(2);
    RT_SEC_BEGIN(3);
    // Unwind the transform chain innermost first: RotateY rotates p and normal back and keeps
    // front_face (hittable.rs:165-179), Translate adds its offset to p (hittable.rs:86).
    for (int32_t c = b.ctx; c != 0;) {
        const DXform x = xf_get(xt, c - 1);
        if (x.kind == 0) {
            h.p = d3(h.p.x + x.a[0], h.p.y + x.a[1], h.p.z + x.a[2]);
        } else {
            double s = x.a[0], cs = x.a[1];
            D3 p = d3(cs * h.p.x + s * h.p.z, h.p.y, -s * h.p.x + cs * h.p.z);
            D3 n = d3(cs * h.normal.x + s * h.normal.z, h.normal.y, -s * h.normal.x + cs * h.normal.z);
            h.p = p; h.normal = n;
        }
        c = x.parent;
    }
    RT_SEC_END(3);
}

RT_DEV void finalize(const DScene& S, RayD r, double time, const Best& b, HitRec& h, bool want_uv = true) {
    finalize(S, xf_from_memory(S), r, time, b, h, want_uv);
}

RT_DEV void finalize(const DScene& S, const WorldRay& w, double time, const Best& b, HitRec& h, bool want_uv = true) {
    RayD r;
    r.ox = w.ox[w.i]; r.oy = w.oy[w.i]; r.oz = w.oz[w.i];
    r.dx = w.dx[w.i]; r.dy = w.dy[w.i]; r.dz = w.dz[w.i];
    finalize(S, r, time, b, h, want_uv);
}

} // namespace rt
#endif
