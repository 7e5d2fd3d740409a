// device_math.h -- f64 vector helpers, the counter-based RNG and the per-primitive
// intersection tests used by the HIP kernels.
//
// Arithmetic contract: every expression below is written in the operand order of
// the reference formula it cites and the translation unit is compiled with
// -ffp-contract=off, so +,-,*,/ and sqrt produce the bits the reference's f64 code
// produces (rustc never fuses a*b+c).  Comparisons are written as the reference
// writes them (no fmin/fmax where it uses `<`), which keeps its NaN behaviour.
#ifndef RT_DEVICE_MATH_H
#define RT_DEVICE_MATH_H

#include <hip/hip_runtime.h>
#include "device_types.h"

namespace rt {

#define RT_DEV __device__ __forceinline__

constexpr double D_INF = __builtin_huge_val();
constexpr double D_PI = 3.1415926535897932385; // rtweekend.rs:7

struct D3 { double x, y, z; };
RT_DEV D3 d3(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV D3 d3p(const double* p) { return d3(p[0], p[1], p[2]); }
RT_DEV D3 operator+(D3 a, D3 b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV D3 operator-(D3 a, D3 b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV D3 operator-(D3 a) { return d3(-a.x, -a.y, -a.z); }
RT_DEV D3 operator*(D3 a, D3 b) { return d3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV D3 operator*(double t, D3 v) { return d3(t * v.x, t * v.y, t * v.z); }
RT_DEV D3 operator*(D3 v, double t) { return d3(t * v.x, t * v.y, t * v.z); }
RT_DEV D3 operator/(D3 v, double t) { return (1.0 / t) * v; } // v / t == v * (1/t)
RT_DEV double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV D3 cross(D3 a, D3 b) { return d3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV double length_squared(D3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RT_DEV double length(D3 a) { return sqrt(length_squared(a)); }
RT_DEV D3 unit_vector(D3 v) { return v / length(v); }

RT_DEV int32_t sat_i32(double x) { // Rust `as i32`
    if (x != x) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return (int32_t)x;
}
RT_DEV uint32_t sat_u32(double x) { // Rust `as u32`
    if (x != x || x <= 0.0) return 0u;
    if (x >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)x;
}

// ------------------------------------------------------------------------ rng
// Philox4x32-10; stream layout documented at rt_rng_uniform in mi355rt.h.
__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)M0 * c[0];
        uint64_t p1 = (uint64_t)M1 * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += W0; k1 += W1;
    }
}

__host__ __device__ inline double philox_uniform(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t index) {
    uint32_t c[4] = {(uint32_t)(index >> 1), stream, a, b};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t bits = (index & 1) ? (((uint64_t)c[2] << 32) | c[3]) : (((uint64_t)c[0] << 32) | c[1]);
    return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
}

struct Rng { // one path's stream; `index` is the number of draws made so far
    uint64_t seed;
    uint32_t stream, a, b;
    uint32_t index;
    RT_DEV double next() { return philox_uniform(seed, stream, a, b, index++); }                // rtweekend.rs:15-17
    RT_DEV double range(double lo, double hi) { return lo + (hi - lo) * next(); }               // rtweekend.rs:19-21
    RT_DEV int32_t random_int(int32_t lo, int32_t hi) { return sat_i32(range((double)lo, (double)(hi + 1))); } // rtweekend.rs:23-25
};

RT_DEV D3 random_unit_vector(Rng& g) { // material.rs:146, pdf.rs:37
    for (;;) {
        double a = g.range(-1.0, 1.0);
        double b = g.range(-1.0, 1.0);
        double c = g.range(-1.0, 1.0);
        D3 p = d3(a, b, c);
        double lensq = length_squared(p);
        if (1e-160 < lensq && lensq <= 1.0) return p / sqrt(lensq);
    }
}
RT_DEV D3 random_in_unit_disk(Rng& g) { // camera.rs:238
    for (;;) {
        double a = g.range(-1.0, 1.0);
        double b = g.range(-1.0, 1.0);
        D3 p = d3(a, b, 0.0);
        if (length_squared(p) < 1.0) return p;
    }
}
// sin and cos of phi = 2*pi*r for r in [0, 1): the angle of every azimuth sample (pdf.rs:58, sphere.rs:57-67).
// The quadrant is split off in r (exact), the remainder times 2*pi lies in [-pi/4, pi/4] and goes through
// the fdlibm kernel polynomials (< 1 ulp).  The general library routines carry a large-argument reduction
// the samplers never need, and cost several times the instructions and registers.  The result may differ
// from a correctly rounded sin(fl(2*pi*r)) in the last place, as any two libm implementations do.
RT_DEV void sincos_2pi(double r, double& sn, double& cs) {
    const double q = rint(4.0 * r);             // 0..4
    const double f = r - 0.25 * q;              // exact, |f| <= 1/8
    const double x = __builtin_fma(f, 2.449293598294706414e-16, f * 6.283185307179586477); // 2*pi = hi + lo
    const double z = x * x;
    // __kernel_sin / __kernel_cos (fdlibm k_sin.c, k_cos.c)
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double ks = x + (z * x) * (S1 + z * ps);
    const double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double kc = w + (((1.0 - w) - hz) + z * pc);
    const int qi = (int)q & 3;
    const double s0 = (qi & 1) ? kc : ks, c0 = (qi & 1) ? ks : kc;
    sn = (qi & 2) ? -s0 : s0;
    cs = ((qi + 1) & 2) ? -c0 : c0;
}

RT_DEV D3 random_cosine_direction(Rng& g) { // pdf.rs:58
    double r1 = g.next();
    double r2 = g.next();
    double sphi, cphi;
    sincos_2pi(r1, sphi, cphi);
    double x = cphi * sqrt(r2);
    double y = sphi * sqrt(r2);
    double z = sqrt(1.0 - r2);
    return d3(x, y, z);
}
RT_DEV D3 reflect(D3 v, D3 n) { return v - 2.0 * dot(v, n) * n; } // material.rs:145,209
RT_DEV D3 refract(D3 uv, D3 n, double etai_over_etat) {           // material.rs:211
    double cos_theta = fmin(dot(-uv, n), 1.0);
    D3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    D3 r_out_parallel = -sqrt(fabs(1.0 - length_squared(r_out_perp))) * n;
    return r_out_perp + r_out_parallel;
}

struct Onb { // onb.rs:3-39
    D3 a0, a1, a2;
    RT_DEV explicit Onb(D3 n) {
        a2 = unit_vector(n);
        D3 a = fabs(a2.x) > 0.9 ? d3(0.0, 1.0, 0.0) : d3(1.0, 0.0, 0.0);
        a1 = unit_vector(cross(a2, a));
        a0 = cross(a2, a1);
    }
    RT_DEV D3 transform(D3 v) const { return a0 * v.x + a1 * v.y + a2 * v.z; }
};

// ----------------------------------------------------------------------- rays
struct RayD { // a ray in the coordinates of the current transform context
    double ox, oy, oz, dx, dy, dz;
    double ix, iy, iz; // 1/d per axis: the value aabb.rs:88 recomputes at every node
};
RT_DEV void set_inverse(RayD& r) { r.ix = 1.0 / r.dx; r.iy = 1.0 / r.dy; r.iz = 1.0 / r.dz; }

// Conservative f32 slab test of the 4-wide nodes.  For one axis the entry / exit parameters of the
// slab [b0, b1] are  t = b * inv + c  with c = -(o * inv)  (one fused multiply-add per plane; the
// fusion only concerns this culling test, never a value that is reported).  With inv and c rounded
// to f32 the computed t is within 2^-23 * |inv| * (|b| + |o|) of the exact one, so the near
// parameter is lowered and the far one raised by  E = 2^-21 * |inv| * (M + |o|),  M = largest
// |coordinate| of the BVH: the f32 test accepts every box the exact f64 test accepts.  A NaN
// (0 * inf) is dropped by fmin/fmax, which leaves that axis unconstrained (accepting).
struct Ray32 {
    float inv[3];  // 1/d
    float cn[3];   // c - E
    float cf[3];   // c + E
    bool neg[3];   // d < 0: the box max is the entry plane
};

RT_DEV void make_ray32(const RayD& r, float scene_extent, Ray32& q) { // needs r.ix, r.iy, r.iz (set_inverse)
    const double o[3] = {r.ox, r.oy, r.oz}, d[3] = {r.dx, r.dy, r.dz}, iv[3] = {r.ix, r.iy, r.iz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double inv = iv[k];
        const float inv32 = (float)inv;
        const float c = (float)(-(o[k] * inv));
        const float e = 4.76837158203125e-07f * fabsf(inv32) * (scene_extent + fabsf((float)o[k])); // 2^-21
        q.inv[k] = inv32;
        q.cn[k] = c - e;
        q.cf[k] = c + e;
        q.neg[k] = d[k] < 0.0;
    }
}

RT_DEV float f32_round_up(double x) { // smallest float >= x
    float f = (float)x;
    return ((double)f < x) ? __uint_as_float(__float_as_uint(f) + (f >= 0.0f ? 1u : 0xFFFFFFFFu)) : f;
}
RT_DEV float f32_round_down(double x) { // largest float <= x
    float f = (float)x;
    return ((double)f > x) ? __uint_as_float(__float_as_uint(f) + (f > 0.0f ? 0xFFFFFFFFu : 1u)) : f;
}


// Aabb::hit (aabb.rs:82-107) on bounds = xmin xmax ymin ymax zmin zmax.
RT_DEV bool slab_test(const double* b, const RayD& r, double lo, double hi) {
    {
        double t0 = (b[0] - r.ox) * r.ix, t1 = (b[1] - r.ox) * r.ix;
        double tmn = t0 < t1 ? t0 : t1, tmx = t0 < t1 ? t1 : t0;
        if (tmn > lo) lo = tmn;
        if (tmx < hi) hi = tmx;
        if (hi <= lo) return false;
    }
    {
        double t0 = (b[2] - r.oy) * r.iy, t1 = (b[3] - r.oy) * r.iy;
        double tmn = t0 < t1 ? t0 : t1, tmx = t0 < t1 ? t1 : t0;
        if (tmn > lo) lo = tmn;
        if (tmx < hi) hi = tmx;
        if (hi <= lo) return false;
    }
    {
        double t0 = (b[4] - r.oz) * r.iz, t1 = (b[5] - r.oz) * r.iz;
        double tmn = t0 < t1 ? t0 : t1, tmx = t0 < t1 ? t1 : t0;
        if (tmn > lo) lo = tmn;
        if (tmx < hi) hi = tmx;
        if (hi <= lo) return false;
    }
    return true;
}

// The same test on an f32 box that contains the node's f64 box (WNode); `near_t` is the clipped
// entry distance.  Every operation is monotone in the box coordinates, so this test passes
// whenever slab_test passes on the exact box: the ordered traversal visits a superset.
RT_DEV bool slab_test_wide(const float* b, const RayD& r, double lo, double hi, double& near_t) {
    {
        double t0 = ((double)b[0] - r.ox) * r.ix, t1 = ((double)b[1] - r.ox) * r.ix;
        double tmn = t0 < t1 ? t0 : t1, tmx = t0 < t1 ? t1 : t0;
        if (tmn > lo) lo = tmn;
        if (tmx < hi) hi = tmx;
        if (hi <= lo) return false;
    }
    {
        double t0 = ((double)b[2] - r.oy) * r.iy, t1 = ((double)b[3] - r.oy) * r.iy;
        double tmn = t0 < t1 ? t0 : t1, tmx = t0 < t1 ? t1 : t0;
        if (tmn > lo) lo = tmn;
        if (tmx < hi) hi = tmx;
        if (hi <= lo) return false;
    }
    {
        double t0 = ((double)b[4] - r.oz) * r.iz, t1 = ((double)b[5] - r.oz) * r.iz;
        double tmn = t0 < t1 ? t0 : t1, tmx = t0 < t1 ? t1 : t0;
        if (tmn > lo) lo = tmn;
        if (tmx < hi) hi = tmx;
        if (hi <= lo) return false;
    }
    near_t = lo;
    return true;
}

// Sphere::hit (sphere.rs:72-98): the accepted root, open interval (tmin, tmax).
RT_DEV bool sphere_test(const DSphere& s, const RayD& r, double time, double tmin, double tmax, double& t_out) {
    D3 center = d3(s.c[0], s.c[1], s.c[2]) + time * d3(s.dc[0], s.dc[1], s.dc[2]);
    D3 o = d3(r.ox, r.oy, r.oz), d = d3(r.dx, r.dy, r.dz);
    D3 oc = center - o;
    double a = length_squared(d);
    double h = dot(d, oc);
    double c = length_squared(oc) - s.radius * s.radius;
    double discriminant = h * h - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = sqrt(discriminant);
    double root = (h - sqrtd) / a;
    if (!(tmin < root && root < tmax)) {
        root = (h + sqrtd) / a;
        if (!(tmin < root && root < tmax)) return false;
    }
    t_out = root;
    return true;
}

// Quad::hit (quad.rs:76-103): closed interval [tmin, tmax]; alpha/beta returned for rec.u/v.
RT_DEV bool quad_test(const DQuad& qd, const RayD& r, double tmin, double tmax, double& t_out, double& alpha, double& beta) {
    D3 n = d3(qd.n[0], qd.n[1], qd.n[2]);
    D3 o = d3(r.ox, r.oy, r.oz), d = d3(r.dx, r.dy, r.dz);
    double denom = dot(n, d);
    if (fabs(denom) < 1e-8) return false;
    double t = (qd.d - dot(n, o)) / denom;
    if (!(tmin <= t && t <= tmax)) return false;
    D3 intersection = o + t * d;
    D3 hp = intersection - d3(qd.q[0], qd.q[1], qd.q[2]);
    D3 w = d3(qd.w[0], qd.w[1], qd.w[2]);
    alpha = dot(w, cross(hp, d3(qd.v[0], qd.v[1], qd.v[2])));
    beta = dot(w, cross(d3(qd.u[0], qd.u[1], qd.u[2]), hp));
    if (!(0.0 <= alpha && alpha <= 1.0) || !(0.0 <= beta && beta <= 1.0)) return false;
    t_out = t;
    return true;
}

// The same test without early exits: `t` of the plane hit and whether the point lies inside the quad, the
// interval check left to the caller.  Every condition of quad_test is a pure predicate of the inputs, so
// (geometric conditions) && (tmin <= t <= tmax) is the same boolean; written branch-free so that two
// evaluations placed next to each other overlap their f64 dependency chains.
RT_DEV bool quad_eval(const DQuad& qd, const RayD& r, double& t_out) {
    D3 n = d3(qd.n[0], qd.n[1], qd.n[2]);
    D3 o = d3(r.ox, r.oy, r.oz), d = d3(r.dx, r.dy, r.dz);
    double denom = dot(n, d);
    double t = (qd.d - dot(n, o)) / denom;
    D3 intersection = o + t * d;
    D3 hp = intersection - d3(qd.q[0], qd.q[1], qd.q[2]);
    D3 w = d3(qd.w[0], qd.w[1], qd.w[2]);
    double alpha = dot(w, cross(hp, d3(qd.v[0], qd.v[1], qd.v[2])));
    double beta = dot(w, cross(d3(qd.u[0], qd.u[1], qd.u[2]), hp));
    t_out = t;
    return !(fabs(denom) < 1e-8) & (0.0 <= alpha) & (alpha <= 1.0) & (0.0 <= beta) & (beta <= 1.0);
}

// Triangle::hit (triangle.rs:153-188): Moeller-Trumbore with edges recomputed per call.
RT_DEV bool tri_test(D3 v0, D3 v1, D3 v2, const RayD& r, double tmin, double tmax, double& t_out, double& u_out, double& v_out) {
    D3 o = d3(r.ox, r.oy, r.oz), d = d3(r.dx, r.dy, r.dz);
    D3 edge1 = v1 - v0, edge2 = v2 - v0;
    D3 h = cross(d, edge2);
    double a = dot(edge1, h);
    if (fabs(a) < 1e-8) return false;
    double f = 1.0 / a;
    D3 s = o - v0;
    double u = f * dot(s, h);
    if (u < 0.0 || u > 1.0) return false;
    D3 q = cross(s, edge1);
    double v = f * dot(d, q);
    if (v < 0.0 || u + v > 1.0) return false;
    double t = f * dot(edge2, q);
    if (!(tmin <= t && t <= tmax)) return false;
    t_out = t; u_out = u; v_out = v;
    return true;
}

} // namespace rt
#endif
