// oracle.cpp -- CPU restatement (C++17, f64) of the reference path tracer's hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing the product ships may include, link, load or
// call this file: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg use it, and only as the checker / the timed CPU baseline.
//
// PARITY UNPINNED: the reference (/root/reference, a Rust crate) holds no tests,
// golden vectors or fixtures for this path, cannot be compiled in this image (no
// rustc/cargo) and its src/vec3.rs is a byte-identical copy of src/texture.rs, so
// the Vec3 API below is reconstructed from its call sites and the conventions of
// the "Ray Tracing in One Weekend" books.  What pins this oracle are hand-derived
// known-answer tests (tests/golden/kat.json) and a BVH-vs-linear-scan differential.
//
// Every function cites the reference file:line it follows.  The recursion, the
// traversal order, the open/closed interval tests and the order of random draws
// are kept exactly as in the reference; only `rand::thread_rng()`
// (rtweekend.rs:15-21) is replaced by a counter-based Philox4x32-10 stream.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (see oracle/Makefile).

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <future>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../include/mi355rt.h"

namespace oracle {

static thread_local std::string g_error;
static const double INF = std::numeric_limits<double>::infinity();
static const double PI = 3.1415926535897932385; // rtweekend.rs:7

// ---------------------------------------------------------------------- rng
// Philox4x32-10 (Salmon et al., SC'11).  Stream layout as documented in mi355rt.h.
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
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

static inline double rng_uniform(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t index) {
    uint32_t c[4] = {(uint32_t)(index >> 1), stream, a, b};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint64_t bits = (index & 1) ? (((uint64_t)c[2] << 32) | c[3]) : (((uint64_t)c[0] << 32) | c[1]);
    return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
}

struct Rng {
    uint64_t seed = 0;
    uint32_t stream = 0, a = 0, b = 0;
    uint64_t index = 0;
    double next() { return rng_uniform(seed, stream, a, b, index++); }
};

// The reference draws from a thread-local generator (rtweekend.rs:15-17); the
// oracle points that thread-local at the stream of the path being evaluated.
static thread_local Rng* tl_rng = nullptr;
static thread_local uint64_t tl_node_visits = 0, tl_prim_tests = 0, tl_world_hits = 0;

static inline double random_double() { return tl_rng->next(); }                                   // rtweekend.rs:15-17
static inline double random_double_range(double lo, double hi) { return lo + (hi - lo) * random_double(); } // rtweekend.rs:19-21
static inline int32_t sat_i32(double x) { // Rust `as i32`: saturating, NaN -> 0
    if (x != x) return 0;
    if (x <= -2147483648.0) return INT32_MIN;
    if (x >= 2147483647.0) return INT32_MAX;
    return (int32_t)x;
}
static inline uint32_t sat_u32(double x) { // Rust `as u32`
    if (x != x || x <= 0.0) return 0;
    if (x >= 4294967295.0) return UINT32_MAX;
    return (uint32_t)x;
}
static inline int32_t random_int(int32_t lo, int32_t hi) { // rtweekend.rs:23-25
    return sat_i32(random_double_range((double)lo, (double)(hi + 1)));
}

// --------------------------------------------------------------------- vec3
// Reconstructed API (the reference's vec3.rs is missing); semantics from call sites.
struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    double length_squared() const { return x * x + y * y + z * z; }
    double length() const { return std::sqrt(length_squared()); }
};
using Point3 = Vec3;
using Color = Vec3; // color.rs:5
static inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
static inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline Vec3 operator*(double t, Vec3 v) { return Vec3(t * v.x, t * v.y, t * v.z); }
static inline Vec3 operator*(Vec3 v, double t) { return t * v; }
static inline Vec3 operator/(Vec3 v, double t) { return (1.0 / t) * v; }
static inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline Vec3 cross(Vec3 a, Vec3 b) {
    return Vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline Vec3 unit_vector(Vec3 v) { return v / v.length(); }
static inline Vec3 vec3_random_range(double lo, double hi) {
    double a = random_double_range(lo, hi);
    double b = random_double_range(lo, hi);
    double c = random_double_range(lo, hi);
    return Vec3(a, b, c);
}
static inline Vec3 random_unit_vector() { // used by material.rs:146, pdf.rs:37
    for (;;) {
        Vec3 p = vec3_random_range(-1.0, 1.0);
        double lensq = p.length_squared();
        if (1e-160 < lensq && lensq <= 1.0) return p / std::sqrt(lensq);
    }
}
static inline Vec3 random_in_unit_disk() { // used by camera.rs:238
    for (;;) {
        double a = random_double_range(-1.0, 1.0);
        double b = random_double_range(-1.0, 1.0);
        Vec3 p(a, b, 0.0);
        if (p.length_squared() < 1.0) return p;
    }
}
static inline Vec3 random_cosine_direction() { // used by pdf.rs:58
    double r1 = random_double();
    double r2 = random_double();
    double phi = 2.0 * PI * r1;
    volatile double phi_again = phi; // keeps cos and sin two libm calls, as in the reference (see RotateY)
    double x = std::cos(phi) * std::sqrt(r2);
    double y = std::sin(phi_again) * std::sqrt(r2);
    double z = std::sqrt(1.0 - r2);
    return Vec3(x, y, z);
}
static inline Vec3 reflect(Vec3 v, Vec3 n) { return v - 2.0 * dot(v, n) * n; } // used by material.rs:145,209
static inline Vec3 refract(Vec3 uv, Vec3 n, double etai_over_etat) {            // used by material.rs:211
    double cos_theta = std::fmin(dot(-uv, n), 1.0);
    Vec3 r_out_perp = etai_over_etat * (uv + cos_theta * n);
    Vec3 r_out_parallel = -std::sqrt(std::fabs(1.0 - r_out_perp.length_squared())) * n;
    return r_out_perp + r_out_parallel;
}

// ----------------------------------------------------------------- interval
struct Interval { // interval.rs:5-89
    double min = INF, max = -INF;
    Interval() {}
    Interval(double a, double b) : min(a), max(b) {}
    static Interval from_intervals(const Interval& a, const Interval& b) { // interval.rs:43-48
        return Interval(std::fmin(a.min, b.min), std::fmax(a.max, b.max));
    }
    double size() const { return max - min; }                        // interval.rs:51-53
    bool contains(double x) const { return min <= x && x <= max; }   // interval.rs:56-58
    bool surrounds(double x) const { return min < x && x < max; }    // interval.rs:61-63
    double clamp(double x) const { if (x < min) return min; if (x > max) return max; return x; } // interval.rs:65-73
    Interval expand(double delta) const { double p = delta / 2.0; return Interval(min - p, max + p); } // interval.rs:75-81
    Interval add(double d) const { return Interval(min + d, max + d); } // interval.rs:83-88
};
static const Interval UNIVERSE(-INF, INF);

// ---------------------------------------------------------------------- ray
struct Ray { // ray.rs:5-57
    Point3 orig; Vec3 dir; double tm = 0.0;
    Ray() {}
    Ray(Point3 o, Vec3 d) : orig(o), dir(d), tm(0.0) {}
    Ray(Point3 o, Vec3 d, double t) : orig(o), dir(d), tm(t) {}
    Point3 at(double t) const { return orig + t * dir; } // ray.rs:54-56
};

// --------------------------------------------------------------------- aabb
struct Aabb { // aabb.rs:10-161
    Interval x, y, z; // default = EMPTY
    Aabb() {}
    Aabb(Interval a, Interval b, Interval c) : x(a), y(b), z(c) {}
    static Aabb from_points(Point3 a, Point3 b) { // aabb.rs:43-63
        Aabb r;
        r.x = a.x <= b.x ? Interval(a.x, b.x) : Interval(b.x, a.x);
        r.y = a.y <= b.y ? Interval(a.y, b.y) : Interval(b.y, a.y);
        r.z = a.z <= b.z ? Interval(a.z, b.z) : Interval(b.z, a.z);
        r.pad_to_minimums();
        return r;
    }
    static Aabb from_aabbs(const Aabb& a, const Aabb& b) { // aabb.rs:65-71 (no padding)
        return Aabb(Interval::from_intervals(a.x, b.x), Interval::from_intervals(a.y, b.y),
                    Interval::from_intervals(a.z, b.z));
    }
    const Interval& axis_interval(int n) const { return n == 1 ? y : (n == 2 ? z : x); } // aabb.rs:73-80
    bool hit(const Ray& r, Interval ray_t) const { // aabb.rs:82-107
        ++tl_node_visits;
        for (int axis = 0; axis < 3; ++axis) {
            const Interval& ax = axis_interval(axis);
            double adinv = 1.0 / r.dir[axis];
            double t0 = (ax.min - r.orig[axis]) * adinv;
            double t1 = (ax.max - r.orig[axis]) * adinv;
            double t_min, t_max;
            if (t0 < t1) { t_min = t0; t_max = t1; } else { t_min = t1; t_max = t0; }
            if (t_min > ray_t.min) ray_t.min = t_min;
            if (t_max < ray_t.max) ray_t.max = t_max;
            if (ray_t.max <= ray_t.min) return false;
        }
        return true;
    }
    int longest_axis() const { // aabb.rs:109-123
        if (x.size() > y.size()) return x.size() > z.size() ? 0 : 2;
        return y.size() > z.size() ? 1 : 2;
    }
    void pad_to_minimums() { // aabb.rs:125-136
        double delta = 0.0001;
        if (x.size() < delta) x = x.expand(delta);
        if (y.size() < delta) y = y.expand(delta);
        if (z.size() < delta) z = z.expand(delta);
    }
    Aabb add_offset(Vec3 o) const { return Aabb(x.add(o.x), y.add(o.y), z.add(o.z)); } // aabb.rs:138-145
};

// ----------------------------------------------------------------- textures
struct Scene;
struct Texture { // texture.rs:6-8
    virtual ~Texture() {}
    virtual Color value(double u, double v, const Point3& p) const = 0;
};
struct SolidColor : Texture { // texture.rs:11-31
    Color albedo;
    explicit SolidColor(Color c) : albedo(c) {}
    Color value(double, double, const Point3&) const override { return albedo; }
};
struct CheckerTexture : Texture { // texture.rs:34-72
    double inv_scale; const Texture* even; const Texture* odd;
    CheckerTexture(double scale, const Texture* e, const Texture* o) : inv_scale(1.0 / scale), even(e), odd(o) {}
    Color value(double u, double v, const Point3& p) const override {
        int32_t xi = sat_i32(std::floor(inv_scale * p.x));
        int32_t yi = sat_i32(std::floor(inv_scale * p.y));
        int32_t zi = sat_i32(std::floor(inv_scale * p.z));
        bool is_even = ((int64_t)xi + yi + zi) % 2 == 0;
        return is_even ? even->value(u, v, p) : odd->value(u, v, p);
    }
};
struct ImageTexture : Texture { // texture.rs:75-108, rtw_stb_image.rs:155-171
    int32_t width = 0, height = 0; const uint8_t* data = nullptr;
    Color value(double u, double v, const Point3&) const override {
        if (height == 0 || data == nullptr) return Color(0.0, 1.0, 1.0);
        u = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);              // f64::clamp
        double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        v = 1.0 - vc;
        uint32_t i = sat_u32(u * (double)width);
        uint32_t j = sat_u32(v * (double)height);
        uint32_t xx = std::min<uint32_t>(i, (uint32_t)width - 1);
        uint32_t yy = std::min<uint32_t>(j, (uint32_t)height - 1);
        const uint8_t* px = data + (size_t)yy * (size_t)width * 3 + (size_t)xx * 3;
        double color_scale = 1.0 / 255.0;
        return Color(color_scale * (double)px[0], color_scale * (double)px[1], color_scale * (double)px[2]);
    }
};
struct Perlin { // perlin.rs:7-143; tables come from the scene description
    const rt_perlin* t;
    static double fade(double x) { return x * x * (3.0 - 2.0 * x); } // perlin.rs:56-58
    static double perlin_interp(const Vec3 c[2][2][2], double u, double v, double w) { // perlin.rs:75-93
        double uu = fade(u), vv = fade(v), ww = fade(w);
        double accum = 0.0;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    Vec3 weight_v(u - (double)i, v - (double)j, w - (double)k);
                    double i_factor = (double)i * uu + (1.0 - (double)i) * (1.0 - uu);
                    double j_factor = (double)j * vv + (1.0 - (double)j) * (1.0 - vv);
                    double k_factor = (double)k * ww + (1.0 - (double)k) * (1.0 - ww);
                    accum += i_factor * j_factor * k_factor * dot(c[i][j][k], weight_v);
                }
        return accum;
    }
    double noise(const Point3& p) const { // perlin.rs:95-129
        double u = p.x - std::floor(p.x);
        double v = p.y - std::floor(p.y);
        double w = p.z - std::floor(p.z);
        int32_t i = sat_i32(std::floor(p.x)), j = sat_i32(std::floor(p.y)), k = sat_i32(std::floor(p.z));
        Vec3 c[2][2][2];
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj)
                for (int dk = 0; dk < 2; ++dk) {
                    uint32_t xi = ((uint32_t)i + (uint32_t)di) & 255u;
                    uint32_t yi = ((uint32_t)j + (uint32_t)dj) & 255u;
                    uint32_t zi = ((uint32_t)k + (uint32_t)dk) & 255u;
                    int32_t index = t->perm_x[xi] ^ t->perm_y[yi] ^ t->perm_z[zi];
                    c[di][dj][dk] = Vec3(t->randvec[index][0], t->randvec[index][1], t->randvec[index][2]);
                }
        return perlin_interp(c, u, v, w);
    }
    double turb(const Point3& p, int depth) const { // perlin.rs:131-143
        double accum = 0.0; Point3 temp_p = p; double weight = 1.0;
        for (int i = 0; i < depth; ++i) {
            accum += weight * noise(temp_p);
            weight *= 0.5;
            temp_p = temp_p * 2.0;
        }
        return std::fabs(accum);
    }
};
struct NoiseTexture : Texture { // texture.rs:111-130
    Perlin noise; double scale;
    Color value(double, double, const Point3& p) const override {
        double noise_value = noise.turb(p, 7);
        return Color(0.5, 0.5, 0.5) * (1.0 + std::sin(scale * p.z + 10.0 * noise_value));
    }
};

// ------------------------------------------------------------ hit record etc.
struct Material;
struct HitRecord { // hittable.rs:14-34
    Point3 p; Vec3 normal; const Material* mat = nullptr; double t = 0, u = 0, v = 0; bool front_face = false;
    uint32_t prim = RT_REF_NONE; int32_t mat_id = -1; // instrumentation: which primitive wrote the record
    void set_face_normal(const Ray& r, Vec3 outward_normal) { // hittable.rs:25-33
        front_face = dot(r.dir, outward_normal) < 0.0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

struct Hittable { // hittable.rs:37-57
    virtual ~Hittable() {}
    virtual bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const = 0;
    virtual Aabb bounding_box() const = 0;
    virtual double pdf_value(const Point3&, const Vec3&) const { return 0.0; }
    virtual Vec3 random(const Point3&) const { return Vec3(1.0, 0.0, 0.0); }
};
using HittablePtr = std::shared_ptr<Hittable>;

// ---------------------------------------------------------------------- onb
struct Onb { // onb.rs:3-39
    Vec3 axis[3];
    explicit Onb(Vec3 n) {
        Vec3 axis_2 = unit_vector(n);
        Vec3 a = std::fabs(axis_2.x) > 0.9 ? Vec3(0.0, 1.0, 0.0) : Vec3(1.0, 0.0, 0.0);
        Vec3 axis_1 = unit_vector(cross(axis_2, a));
        Vec3 axis_0 = cross(axis_2, axis_1);
        axis[0] = axis_0; axis[1] = axis_1; axis[2] = axis_2;
    }
    const Vec3& w() const { return axis[2]; }
    Vec3 transform(Vec3 v) const { return axis[0] * v.x + axis[1] * v.y + axis[2] * v.z; } // onb.rs:36-38
};

// --------------------------------------------------------------------- pdfs
struct Pdf { // pdf.rs:12-15
    virtual ~Pdf() {}
    virtual double value(const Vec3& direction) const = 0;
    virtual Vec3 generate() const = 0;
};
struct SpherePdf : Pdf { // pdf.rs:23-39
    double value(const Vec3&) const override { return 1.0 / (4.0 * PI); }
    Vec3 generate() const override { return random_unit_vector(); }
};
struct CosinePdf : Pdf { // pdf.rs:41-60
    Onb uvw;
    explicit CosinePdf(Vec3 w) : uvw(w) {}
    double value(const Vec3& direction) const override {
        double cosine_theta = dot(unit_vector(direction), uvw.w());
        return std::fmax(cosine_theta / PI, 0.0);
    }
    Vec3 generate() const override { return uvw.transform(random_cosine_direction()); }
};
struct HittablePdf : Pdf { // pdf.rs:83-102
    const Hittable* objects; Point3 origin;
    HittablePdf(const Hittable* o, Point3 p) : objects(o), origin(p) {}
    double value(const Vec3& direction) const override { return objects->pdf_value(origin, direction); }
    Vec3 generate() const override { return objects->random(origin); }
};
struct MixturePdf : Pdf { // pdf.rs:104-126
    const Pdf* p[2];
    MixturePdf(const Pdf* p0, const Pdf* p1) { p[0] = p0; p[1] = p1; }
    double value(const Vec3& direction) const override {
        return 0.5 * p[0]->value(direction) + 0.5 * p[1]->value(direction);
    }
    Vec3 generate() const override {
        if (random_double() < 0.5) return p[0]->generate();
        return p[1]->generate();
    }
};

// ---------------------------------------------------------------- materials
struct ScatterRecord { // material.rs:17-23
    Color attenuation; std::shared_ptr<Pdf> pdf_ptr; bool skip_pdf = false; Ray skip_pdf_ray;
};
struct Material { // material.rs:36-56
    virtual ~Material() {}
    virtual bool scatter(const Ray&, const HitRecord&, ScatterRecord&) const { return false; }
    virtual Color emitted(const Ray&, const HitRecord&, double, double, const Point3&) const { return Color(0, 0, 0); }
    virtual double scattering_pdf(const Ray&, const HitRecord&, const Ray&) const { return 0.0; }
};
struct Lambertian : Material { // material.rs:58-118
    const Texture* tex;
    bool scatter(const Ray&, const HitRecord& rec, ScatterRecord& srec) const override {
        srec.attenuation = tex->value(rec.u, rec.v, rec.p);
        srec.pdf_ptr = std::make_shared<CosinePdf>(rec.normal);
        srec.skip_pdf = false;
        return true;
    }
    double scattering_pdf(const Ray&, const HitRecord& rec, const Ray& scattered) const override {
        double cos_theta = dot(rec.normal, unit_vector(scattered.dir));
        return cos_theta < 0.0 ? 0.0 : cos_theta / PI;
    }
};
struct Metal : Material { // material.rs:120-159
    Color albedo; double fuzz;
    bool scatter(const Ray& r_in, const HitRecord& rec, ScatterRecord& srec) const override {
        Vec3 reflected = reflect(r_in.dir, rec.normal);
        reflected = unit_vector(reflected) + fuzz * random_unit_vector();
        srec.attenuation = albedo;
        srec.pdf_ptr = nullptr;
        srec.skip_pdf = true;
        srec.skip_pdf_ray = Ray(rec.p, reflected, r_in.tm);
        return true;
    }
};
struct Dielectric : Material { // material.rs:161-218
    double refraction_index;
    static double reflectance(double cosine, double ref_idx) { // material.rs:175-178
        double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
        r0 = r0 * r0;                                  // powi(2)
        double m = 1.0 - cosine;
        double m2 = m * m, m4 = m2 * m2;               // powi(5) = m * (m^2)^2
        return r0 + (1.0 - r0) * (m * m4);
    }
    bool scatter(const Ray& r_in, const HitRecord& rec, ScatterRecord& srec) const override {
        srec.attenuation = Color(1.0, 1.0, 1.0);
        srec.pdf_ptr = nullptr;
        srec.skip_pdf = true;
        double ri = rec.front_face ? 1.0 / refraction_index : refraction_index;
        Vec3 unit_direction = unit_vector(r_in.dir);
        double cos_theta = std::fmin(-dot(unit_direction, rec.normal), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0;
        Vec3 direction;
        if (cannot_refract || reflectance(cos_theta, ri) > random_double()) // short-circuit: no draw under TIR
            direction = reflect(unit_direction, rec.normal);
        else
            direction = refract(unit_direction, rec.normal, ri);
        srec.skip_pdf_ray = Ray(rec.p, direction, r_in.tm);
        return true;
    }
};
struct DiffuseLight : Material { // material.rs:220-257
    const Texture* tex;
    Color emitted(const Ray&, const HitRecord& rec, double u, double v, const Point3& p) const override {
        if (!rec.front_face) return Color(0, 0, 0);
        return tex->value(u, v, p);
    }
};
struct Isotropic : Material { // material.rs:261-301
    const Texture* tex;
    bool scatter(const Ray&, const HitRecord& rec, ScatterRecord& srec) const override {
        srec.attenuation = tex->value(rec.u, rec.v, rec.p);
        srec.pdf_ptr = std::make_shared<SpherePdf>();
        srec.skip_pdf = false;
        return true;
    }
    double scattering_pdf(const Ray&, const HitRecord&, const Ray&) const override { return 1.0 / (4.0 * PI); }
};

// ------------------------------------------------------------------- sphere
struct Sphere : Hittable { // sphere.rs:12-142
    Ray center; double radius; const Material* mat; int32_t mat_id; Aabb bbox; uint32_t tag;
    Sphere(const rt_sphere& s, const Material* m, uint32_t tg) : mat(m), mat_id(s.mat), tag(tg) {
        Point3 c1(s.center[0], s.center[1], s.center[2]);
        Vec3 rvec(s.radius, s.radius, s.radius);
        if (!s.moving) { // sphere.rs:25-33
            center = Ray(c1, Vec3(0.0, 0.0, 0.0));
            bbox = Aabb::from_points(c1 - rvec, c1 + rvec);
        } else {         // sphere.rs:35-45
            Point3 c2(s.center2[0], s.center2[1], s.center2[2]);
            Aabb box1 = Aabb::from_points(c1 - rvec, c1 + rvec);
            Aabb box2 = Aabb::from_points(c2 - rvec, c2 + rvec);
            center = Ray(c1, c2 - c1);
            bbox = Aabb::from_aabbs(box1, box2);
        }
        radius = std::fmax(s.radius, 0.0);
    }
    static void get_sphere_uv(const Point3& p, double& u, double& v) { // sphere.rs:47-55
        double theta = std::acos(-p.y);
        double phi = std::atan2(-p.z, p.x) + PI;
        u = phi / (2.0 * PI);
        v = theta / PI;
    }
    static Vec3 random_to_sphere(double radius, double distance_squared) { // sphere.rs:57-67
        double r1 = random_double();
        double r2 = random_double();
        double z = 1.0 + r2 * (std::sqrt(1.0 - radius * radius / distance_squared) - 1.0);
        double phi = 2.0 * PI * r1;
        volatile double phi_again = phi; // two libm calls, as in the reference (see RotateY)
        double x = std::cos(phi) * std::sqrt(1.0 - z * z);
        double y = std::sin(phi_again) * std::sqrt(1.0 - z * z);
        return Vec3(x, y, z);
    }
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // sphere.rs:72-111
        ++tl_prim_tests;
        Point3 current_center = center.at(r.tm);
        Vec3 oc = current_center - r.orig;
        double a = r.dir.length_squared();
        double h = dot(r.dir, oc);
        double c = oc.length_squared() - radius * radius;
        double discriminant = h * h - a * c;
        if (discriminant < 0.0) return false;
        double sqrtd = std::sqrt(discriminant);
        double root = (h - sqrtd) / a;
        if (!ray_t.surrounds(root)) {
            root = (h + sqrtd) / a;
            if (!ray_t.surrounds(root)) return false;
        }
        rec.t = root;
        rec.p = r.at(rec.t);
        Vec3 outward_normal = (rec.p - current_center) / radius;
        rec.set_face_normal(r, outward_normal);
        get_sphere_uv(outward_normal, rec.u, rec.v);
        rec.mat = mat; rec.mat_id = mat_id; rec.prim = tag;
        return true;
    }
    Aabb bounding_box() const override { return bbox; }
    double pdf_value(const Point3& origin, const Vec3& direction) const override { // sphere.rs:117-131
        HitRecord rec;
        Ray ray(origin, direction);
        if (!hit(ray, Interval(0.001, INF), rec)) return 0.0;
        Point3 c = center.at(0.0);
        double dist_squared = (c - origin).length_squared();
        double cos_theta_max = std::sqrt(1.0 - radius * radius / dist_squared);
        double solid_angle = 2.0 * PI * (1.0 - cos_theta_max);
        return 1.0 / solid_angle;
    }
    Vec3 random(const Point3& origin) const override { // sphere.rs:133-141
        Point3 c = center.at(0.0);
        Vec3 direction = c - origin;
        double distance_squared = direction.length_squared();
        Onb uvw(direction);
        return uvw.transform(random_to_sphere(radius, distance_squared));
    }
};

// --------------------------------------------------------------------- quad
struct Quad : Hittable { // quad.rs:14-137
    Point3 q; Vec3 u, v, w; const Material* mat; int32_t mat_id; Aabb bbox; Vec3 normal; double d, area; uint32_t tag;
    Quad(const rt_quad& s, const Material* m, uint32_t tg) : mat(m), mat_id(s.mat), tag(tg) { // quad.rs:28-49
        q = Point3(s.q[0], s.q[1], s.q[2]); u = Vec3(s.u[0], s.u[1], s.u[2]); v = Vec3(s.v[0], s.v[1], s.v[2]);
        Vec3 n = cross(u, v);
        normal = unit_vector(n);
        d = dot(normal, q);
        w = n / dot(n, n);
        area = n.length();
        Point3 p1 = q, p2 = q + u, p3 = q + v, p4 = q + u + v; // quad.rs:51-61
        bbox = Aabb::from_aabbs(Aabb::from_points(p1, p4), Aabb::from_points(p2, p3));
    }
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // quad.rs:76-111
        ++tl_prim_tests;
        double denom = dot(normal, r.dir);
        if (std::fabs(denom) < 1e-8) return false;
        double t = (d - dot(normal, r.orig)) / denom;
        if (!ray_t.contains(t)) return false;
        Point3 intersection = r.at(t);
        Vec3 planar_hitpt_vector = intersection - q;
        double alpha = dot(w, cross(planar_hitpt_vector, v));
        double beta = dot(w, cross(u, planar_hitpt_vector));
        Interval unit_interval(0.0, 1.0); // quad.rs:63-72
        if (!unit_interval.contains(alpha) || !unit_interval.contains(beta)) return false;
        rec.u = alpha; rec.v = beta;
        rec.t = t;
        rec.p = intersection;
        rec.mat = mat; rec.mat_id = mat_id; rec.prim = tag;
        rec.set_face_normal(r, normal);
        return true;
    }
    Aabb bounding_box() const override { return bbox; }
    double pdf_value(const Point3& origin, const Vec3& direction) const override { // quad.rs:117-131
        HitRecord rec;
        if (!hit(Ray(origin, direction), Interval(0.001, INF), rec)) return 0.0;
        double distance_squared = rec.t * rec.t * direction.length_squared();
        double cosine = std::fabs(dot(direction, rec.normal) / direction.length());
        return distance_squared / (cosine * area);
    }
    Vec3 random(const Point3& origin) const override { // quad.rs:133-136
        double r1 = random_double();
        Vec3 a = r1 * u;
        double r2 = random_double();
        Vec3 b = r2 * v;
        Point3 p = q + a + b;
        return p - origin;
    }
};

// ----------------------------------------------------------------- triangle
struct Triangle : Hittable { // triangle.rs:111-204
    Point3 v0, v1, v2; Vec3 normal; const Material* mat; int32_t mat_id; Aabb bbox; uint32_t tag;
    Triangle(const rt_triangle& s, const Material* m, uint32_t tg) : mat(m), mat_id(s.mat), tag(tg) { // triangle.rs:122-149
        v0 = Point3(s.v0[0], s.v0[1], s.v0[2]); v1 = Point3(s.v1[0], s.v1[1], s.v1[2]); v2 = Point3(s.v2[0], s.v2[1], s.v2[2]);
        Vec3 edge1 = v1 - v0, edge2 = v2 - v0;
        normal = unit_vector(cross(edge1, edge2));
        Point3 mn(std::fmin(std::fmin(v0.x, v1.x), v2.x), std::fmin(std::fmin(v0.y, v1.y), v2.y), std::fmin(std::fmin(v0.z, v1.z), v2.z));
        Point3 mx(std::fmax(std::fmax(v0.x, v1.x), v2.x), std::fmax(std::fmax(v0.y, v1.y), v2.y), std::fmax(std::fmax(v0.z, v1.z), v2.z));
        bbox = Aabb::from_points(mn, mx);
    }
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // triangle.rs:153-199
        ++tl_prim_tests;
        Vec3 edge1 = v1 - v0, edge2 = v2 - v0;
        Vec3 h = cross(r.dir, edge2);
        double a = dot(edge1, h);
        if (std::fabs(a) < 1e-8) return false;
        double f = 1.0 / a;
        Vec3 s = r.orig - v0;
        double u = f * dot(s, h);
        if (u < 0.0 || u > 1.0) return false;
        Vec3 qq = cross(s, edge1);
        double v = f * dot(r.dir, qq);
        if (v < 0.0 || u + v > 1.0) return false;
        double t = f * dot(edge2, qq);
        if (!ray_t.contains(t)) return false;
        rec.t = t;
        rec.p = r.at(t);
        rec.set_face_normal(r, normal);
        rec.mat = mat; rec.mat_id = mat_id; rec.prim = tag;
        rec.u = u; rec.v = v;
        return true;
    }
    Aabb bounding_box() const override { return bbox; }
};

// ------------------------------------------------------------- hittable list
struct HittableList : Hittable { // hittable_list.rs:9-82
    std::vector<HittablePtr> objects; Aabb bbox;
    void add(HittablePtr o) { objects.push_back(o); bbox = Aabb::from_aabbs(bbox, o->bounding_box()); } // :37-40
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // :45-61
        HitRecord temp_rec;
        bool hit_anything = false;
        double closest_so_far = ray_t.max;
        for (const auto& object : objects) {
            if (object->hit(r, Interval(ray_t.min, closest_so_far), temp_rec)) {
                hit_anything = true;
                closest_so_far = temp_rec.t;
                rec = temp_rec;
            }
        }
        return hit_anything;
    }
    Aabb bounding_box() const override { return bbox; }
    double pdf_value(const Point3& origin, const Vec3& direction) const override { // :67-76
        double weight = 1.0 / (double)objects.size();
        double sum = 0.0;
        for (const auto& object : objects) sum += weight * object->pdf_value(origin, direction);
        return sum;
    }
    Vec3 random(const Point3& origin) const override { // :78-81
        int32_t index = random_int(0, (int32_t)objects.size() - 1);
        return objects[(size_t)index]->random(origin);
    }
};

// obj_loader.rs:13-106: triangles scanned linearly, each with a fresh record.
struct ObjModel : Hittable {
    std::vector<HittablePtr> triangles; Point3 bbox_min, bbox_max;
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // :86-101
        bool hit_anything = false;
        double closest_so_far = ray_t.max;
        for (const auto& tri : triangles) {
            HitRecord temp_rec;
            if (tri->hit(r, Interval(ray_t.min, closest_so_far), temp_rec)) {
                hit_anything = true;
                closest_so_far = temp_rec.t;
                rec = temp_rec;
            }
        }
        return hit_anything;
    }
    Aabb bounding_box() const override { return Aabb::from_points(bbox_min, bbox_max); } // :103-105
};

// ---------------------------------------------------------------------- bvh
struct BvhNode : Hittable { // bvh.rs:6-145
    HittablePtr left, right; Aabb bbox;
    static std::shared_ptr<BvhNode> from_objects(std::vector<HittablePtr>& objects, size_t start, size_t end, int depth) { // bvh.rs:20-77
        size_t object_span = end - start;
        auto node = std::make_shared<BvhNode>();
        Aabb bbox;
        for (size_t i = start; i < end; ++i) bbox = Aabb::from_aabbs(bbox, objects[i]->bounding_box());
        int axis = bbox.longest_axis();
        if (object_span == 1) {
            node->left = node->right = objects[start];
        } else if (object_span == 2) {
            node->left = objects[start];
            node->right = objects[start + 1];
        } else {
            // bvh.rs:52,79-94: stable sort on bbox.axis(axis).min.  The key of every object is read
            // once and an index permutation is sorted; a stable sort's result is unique, so this
            // is the order `sort_by(partial_cmp)` produces.
            std::vector<double> key(object_span);
            std::vector<uint32_t> order(object_span);
            for (size_t i = 0; i < object_span; ++i) {
                key[i] = objects[start + i]->bounding_box().axis_interval(axis).min;
                order[i] = (uint32_t)i;
            }
            std::stable_sort(order.begin(), order.end(), [&key](uint32_t a, uint32_t b) { return key[a] < key[b]; });
            {
                std::vector<HittablePtr> sorted(object_span);
                for (size_t i = 0; i < object_span; ++i) sorted[i] = std::move(objects[start + order[i]]);
                for (size_t i = 0; i < object_span; ++i) objects[start + i] = std::move(sorted[i]);
            }
            size_t mid = start + object_span / 2;
            std::vector<HittablePtr> left_objects(objects.begin() + start, objects.begin() + mid);
            std::vector<HittablePtr> right_objects(objects.begin() + mid, objects.begin() + end);
            if (depth < 4 && object_span > 4096) { // rayon::join (bvh.rs:64-67): same result, two threads
                auto fut = std::async(std::launch::async, [&]() { return from_objects(left_objects, 0, left_objects.size(), depth + 1); });
                node->right = from_objects(right_objects, 0, right_objects.size(), depth + 1);
                node->left = fut.get();
            } else {
                node->left = from_objects(left_objects, 0, left_objects.size(), depth + 1);
                node->right = from_objects(right_objects, 0, right_objects.size(), depth + 1);
            }
        }
        node->bbox = bbox;
        return node;
    }
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // bvh.rs:118-140
        if (!bbox.hit(r, ray_t)) return false;
        bool hit_left = left->hit(r, ray_t, rec);
        Interval new_ray_t = hit_left ? Interval(ray_t.min, rec.t) : ray_t;
        bool hit_right = right->hit(r, new_ray_t, rec);
        return hit_left || hit_right;
    }
    Aabb bounding_box() const override { return bbox; }
};

// --------------------------------------------------------------- transforms
struct Translate : Hittable { // hittable.rs:60-94
    HittablePtr object; Vec3 offset; Aabb bbox;
    Translate(HittablePtr o, Vec3 off) : object(o), offset(off) { bbox = o->bounding_box().add_offset(off); }
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override {
        Ray offset_r(r.orig - offset, r.dir, r.tm);
        if (!object->hit(offset_r, ray_t, rec)) return false;
        rec.p = rec.p + offset;
        return true;
    }
    Aabb bounding_box() const override { return bbox; }
};
struct RotateY : Hittable { // hittable.rs:96-187
    HittablePtr object; double sin_theta, cos_theta; Aabb bbox;
    RotateY(HittablePtr o, double angle) : object(o) {
        double radians = angle * (PI / 180.0); // f64::to_radians
        // two separate libm calls, as the reference makes them (hittable.rs:106-108): g++ would otherwise merge the
        // pair into one sincos() call, and glibc's sincos differs from its sin / cos in the last place for about one
        // argument in 750
        volatile double radians_again = radians;
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians_again);
        Aabb b = o->bounding_box();
        Point3 mn(INF, INF, INF), mx(-INF, -INF, -INF);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double x = i == 0 ? b.x.min : b.x.max;
                    double y = j == 0 ? b.y.min : b.y.max;
                    double z = k == 0 ? b.z.min : b.z.max;
                    double newx = cos_theta * x + sin_theta * z;
                    double newz = -sin_theta * x + cos_theta * z;
                    Vec3 tester(newx, y, newz);
                    for (int c = 0; c < 3; ++c) {
                        mn[c] = std::fmin(mn[c], tester[c]);
                        mx[c] = std::fmax(mx[c], tester[c]);
                    }
                }
        bbox = Aabb::from_points(mn, mx);
    }
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // hittable.rs:145-182
        Point3 origin(cos_theta * r.orig.x - sin_theta * r.orig.z, r.orig.y, sin_theta * r.orig.x + cos_theta * r.orig.z);
        Vec3 direction(cos_theta * r.dir.x - sin_theta * r.dir.z, r.dir.y, sin_theta * r.dir.x + cos_theta * r.dir.z);
        Ray rotated_r(origin, direction, r.tm);
        if (!object->hit(rotated_r, ray_t, rec)) return false;
        Point3 p(cos_theta * rec.p.x + sin_theta * rec.p.z, rec.p.y, -sin_theta * rec.p.x + cos_theta * rec.p.z);
        Vec3 normal(cos_theta * rec.normal.x + sin_theta * rec.normal.z, rec.normal.y,
                    -sin_theta * rec.normal.x + cos_theta * rec.normal.z);
        rec.p = p;
        rec.normal = normal; // front_face is deliberately not recomputed (hittable.rs:179)
        return true;
    }
    Aabb bounding_box() const override { return bbox; }
};

// ---------------------------------------------------------- constant medium
struct ConstantMedium : Hittable { // constant_medium.rs:13-109
    HittablePtr boundary; double neg_inv_density; const Material* phase_function; int32_t mat_id; uint32_t tag;
    bool hit(const Ray& r, Interval ray_t, HitRecord& rec) const override { // :48-104
        ++tl_prim_tests;
        HitRecord rec1, rec2;
        if (!boundary->hit(r, UNIVERSE, rec1)) return false;
        if (!boundary->hit(r, Interval(rec1.t + 0.0001, INF), rec2)) return false;
        double rec1_t = rec1.t < ray_t.min ? ray_t.min : rec1.t;
        double rec2_t = rec2.t > ray_t.max ? ray_t.max : rec2.t;
        if (rec1_t >= rec2_t) return false;
        if (rec1_t < 0.0) rec1_t = 0.0;
        double ray_length = r.dir.length();
        double distance_inside_boundary = (rec2_t - rec1_t) * ray_length;
        double hit_distance = neg_inv_density * std::log(random_double());
        if (hit_distance > distance_inside_boundary) return false;
        rec.t = rec1_t + hit_distance / ray_length;
        rec.p = r.at(rec.t);
        rec.normal = Vec3(1.0, 0.0, 0.0);
        rec.front_face = true;
        rec.mat = phase_function; rec.mat_id = mat_id; rec.prim = tag;
        return true; // rec.u / rec.v keep whatever the record held (constant_medium.rs never writes them)
    }
    Aabb bounding_box() const override { return boundary->bounding_box(); }
};

// -------------------------------------------------------------------- scene
struct Scene {
    rt_scene_desc d;
    std::vector<uint8_t> image_bytes;
    std::vector<rt_perlin> perlins;
    std::vector<std::unique_ptr<Texture>> textures;
    std::vector<std::unique_ptr<Material>> materials;
    std::vector<HittablePtr> spheres, quads, triangles, objects;
    HittablePtr world, lights;
};

static bool fail(const std::string& m) { g_error = m; return false; }

static bool build_scene(const rt_scene_desc* d, Scene& s) {
    s.d = *d;
    s.image_bytes.assign(d->image_bytes, d->image_bytes + d->n_image_bytes);
    s.perlins.assign(d->perlins, d->perlins + d->n_perlins);
    // textures: children (checker) may reference any index, so create in two passes
    s.textures.resize(d->n_textures);
    for (uint64_t i = 0; i < d->n_textures; ++i) {
        const rt_texture& t = d->textures[i];
        switch (t.kind) {
        case RT_TEX_SOLID: s.textures[i].reset(new SolidColor(Color(t.color[0], t.color[1], t.color[2]))); break;
        case RT_TEX_CHECKER: s.textures[i].reset(new CheckerTexture(t.scale, nullptr, nullptr)); break;
        case RT_TEX_IMAGE: {
            auto* it = new ImageTexture();
            if (t.image >= 0) {
                if ((uint64_t)t.image >= d->n_images) return fail("texture: bad image index");
                const rt_image& im = d->images[t.image];
                it->width = im.width; it->height = im.height; it->data = s.image_bytes.data() + im.offset;
            }
            s.textures[i].reset(it);
            break;
        }
        case RT_TEX_NOISE: {
            if (t.perlin < 0 || (uint64_t)t.perlin >= d->n_perlins) return fail("texture: bad perlin index");
            auto* nt = new NoiseTexture(); nt->noise.t = &s.perlins[t.perlin]; nt->scale = t.scale;
            s.textures[i].reset(nt);
            break;
        }
        default: return fail("texture: bad kind");
        }
    }
    for (uint64_t i = 0; i < d->n_textures; ++i) {
        const rt_texture& t = d->textures[i];
        if (t.kind == RT_TEX_CHECKER) {
            if (t.even < 0 || t.odd < 0 || (uint64_t)t.even >= d->n_textures || (uint64_t)t.odd >= d->n_textures) return fail("checker: bad child");
            auto* c = static_cast<CheckerTexture*>(s.textures[i].get());
            c->even = s.textures[t.even].get(); c->odd = s.textures[t.odd].get();
        }
    }
    auto tex = [&](int32_t id) -> const Texture* { return (id >= 0 && (uint64_t)id < d->n_textures) ? s.textures[id].get() : nullptr; };
    s.materials.resize(d->n_materials);
    for (uint64_t i = 0; i < d->n_materials; ++i) {
        const rt_material& m = d->materials[i];
        switch (m.kind) {
        case RT_MAT_LAMBERTIAN: { auto* x = new Lambertian(); x->tex = tex(m.tex); if (!x->tex) return fail("material: bad texture"); s.materials[i].reset(x); break; }
        case RT_MAT_METAL: { auto* x = new Metal(); x->albedo = Color(m.albedo[0], m.albedo[1], m.albedo[2]); x->fuzz = std::fmin(m.fuzz, 1.0); s.materials[i].reset(x); break; } // material.rs:127-132
        case RT_MAT_DIELECTRIC: { auto* x = new Dielectric(); x->refraction_index = m.ior; s.materials[i].reset(x); break; }
        case RT_MAT_DIFFUSE_LIGHT: { auto* x = new DiffuseLight(); x->tex = tex(m.tex); if (!x->tex) return fail("material: bad texture"); s.materials[i].reset(x); break; }
        case RT_MAT_ISOTROPIC: { auto* x = new Isotropic(); x->tex = tex(m.tex); if (!x->tex) return fail("material: bad texture"); s.materials[i].reset(x); break; }
        default: return fail("material: bad kind");
        }
    }
    auto mat = [&](int32_t id) -> const Material* { return (id >= 0 && (uint64_t)id < d->n_materials) ? s.materials[id].get() : nullptr; };
    s.spheres.resize(d->n_spheres);
    for (uint64_t i = 0; i < d->n_spheres; ++i) {
        if (!mat(d->spheres[i].mat)) return fail("sphere: bad material");
        s.spheres[i] = std::make_shared<Sphere>(d->spheres[i], mat(d->spheres[i].mat), (RT_REF_SPHERE << 28) | (uint32_t)i);
    }
    s.quads.resize(d->n_quads);
    for (uint64_t i = 0; i < d->n_quads; ++i) {
        if (!mat(d->quads[i].mat)) return fail("quad: bad material");
        s.quads[i] = std::make_shared<Quad>(d->quads[i], mat(d->quads[i].mat), (RT_REF_QUAD << 28) | (uint32_t)i);
    }
    s.triangles.resize(d->n_triangles);
    for (uint64_t i = 0; i < d->n_triangles; ++i) {
        if (!mat(d->triangles[i].mat)) return fail("triangle: bad material");
        s.triangles[i] = std::make_shared<Triangle>(d->triangles[i], mat(d->triangles[i].mat), (RT_REF_TRI << 28) | (uint32_t)i);
    }
    // objects: children must have a smaller id than their parent (the host emits them bottom-up)
    s.objects.resize(d->n_objects);
    for (uint64_t i = 0; i < d->n_objects; ++i) {
        const rt_object& o = d->objects[i];
        auto child_obj = [&](int32_t id) -> HittablePtr { return (id >= 0 && (uint64_t)id < i) ? s.objects[id] : nullptr; };
        auto gather = [&](std::vector<HittablePtr>& out) -> bool {
            if (o.a < 0 || o.b < 0) return fail("object: negative range");
            for (int32_t k = 0; k < o.b; ++k) {
                uint64_t idx = (uint64_t)o.a + (uint64_t)k;
                HittablePtr h;
                switch (o.child_kind) {
                case RT_CHILD_OBJECTS: if (idx >= d->n_children) return fail("object: child range"); h = child_obj(d->children[idx]); break;
                case RT_CHILD_SPHERES: if (idx < d->n_spheres) h = s.spheres[idx]; break;
                case RT_CHILD_QUADS: if (idx < d->n_quads) h = s.quads[idx]; break;
                case RT_CHILD_TRIANGLES: if (idx < d->n_triangles) h = s.triangles[idx]; break;
                default: break;
                }
                if (!h) return fail("object: bad child");
                out.push_back(h);
            }
            return true;
        };
        switch (o.kind) {
        case RT_OBJ_SPHERE: if (o.a < 0 || (uint64_t)o.a >= d->n_spheres) return fail("object: sphere index"); s.objects[i] = s.spheres[o.a]; break;
        case RT_OBJ_QUAD: if (o.a < 0 || (uint64_t)o.a >= d->n_quads) return fail("object: quad index"); s.objects[i] = s.quads[o.a]; break;
        case RT_OBJ_TRIANGLE: if (o.a < 0 || (uint64_t)o.a >= d->n_triangles) return fail("object: triangle index"); s.objects[i] = s.triangles[o.a]; break;
        case RT_OBJ_LIST: {
            auto l = std::make_shared<HittableList>();
            std::vector<HittablePtr> ch; if (!gather(ch)) return false;
            for (auto& c : ch) l->add(c);
            s.objects[i] = l; break;
        }
        case RT_OBJ_BVH: { // BvhNode::new (bvh.rs:14-18)
            std::vector<HittablePtr> ch; if (!gather(ch)) return false;
            if (ch.empty()) return fail("bvh: empty list");
            for (auto& c : ch) { Aabb b = c->bounding_box(); if (b.x.min != b.x.min || b.y.min != b.y.min || b.z.min != b.z.min) return fail("bvh: NaN bounding box"); }
            s.objects[i] = BvhNode::from_objects(ch, 0, ch.size(), 0); break;
        }
        case RT_OBJ_TRANSLATE: { auto c = child_obj(o.a); if (!c) return fail("translate: bad child"); s.objects[i] = std::make_shared<Translate>(c, Vec3(o.param[0], o.param[1], o.param[2])); break; }
        case RT_OBJ_ROTATE_Y: { auto c = child_obj(o.a); if (!c) return fail("rotate_y: bad child"); s.objects[i] = std::make_shared<RotateY>(c, o.param[0]); break; }
        case RT_OBJ_CONSTANT_MEDIUM: {
            auto c = child_obj(o.a); if (!c) return fail("medium: bad boundary");
            if (!mat(o.mat)) return fail("medium: bad material");
            auto m = std::make_shared<ConstantMedium>();
            m->boundary = c; m->neg_inv_density = -1.0 / o.param[0]; m->phase_function = mat(o.mat); m->mat_id = o.mat;
            m->tag = (RT_REF_MEDIUM << 28) | (uint32_t)i;
            s.objects[i] = m; break;
        }
        case RT_OBJ_OBJMODEL: {
            auto m = std::make_shared<ObjModel>();
            m->bbox_min = Point3(INF, INF, INF); m->bbox_max = Point3(-INF, -INF, -INF);
            if (o.a < 0 || o.b < 0 || (uint64_t)o.a + (uint64_t)o.b > d->n_triangles) return fail("objmodel: triangle range");
            for (int32_t k = 0; k < o.b; ++k) {
                const rt_triangle& t = d->triangles[o.a + k];
                const double* vs[3] = {t.v0, t.v1, t.v2};
                for (auto* vv : vs) { // obj_loader.rs:70-71
                    m->bbox_min = Point3(std::fmin(m->bbox_min.x, vv[0]), std::fmin(m->bbox_min.y, vv[1]), std::fmin(m->bbox_min.z, vv[2]));
                    m->bbox_max = Point3(std::fmax(m->bbox_max.x, vv[0]), std::fmax(m->bbox_max.y, vv[1]), std::fmax(m->bbox_max.z, vv[2]));
                }
                m->triangles.push_back(s.triangles[o.a + k]);
            }
            s.objects[i] = m; break;
        }
        default: return fail("object: bad kind");
        }
    }
    if (d->world < 0 || (uint64_t)d->world >= d->n_objects) return fail("scene: bad world id");
    s.world = s.objects[d->world];
    if (d->lights >= 0) {
        if ((uint64_t)d->lights >= d->n_objects) return fail("scene: bad lights id");
        s.lights = s.objects[d->lights];
    }
    return true;
}

// ------------------------------------------------------------------- camera
struct Camera { // camera.rs:15-339
    rt_camera c; rt_camera_derived d;
    Vec3 center, pixel00_loc, pixel_delta_u, pixel_delta_v, defocus_disk_u, defocus_disk_v; Color background;
    void initialize() { // camera.rs:161-205
        d.image_height = sat_i32((double)c.image_width / c.aspect_ratio);
        d.image_height = d.image_height < 1 ? 1 : d.image_height;
        d.sqrt_spp = sat_i32(std::sqrt((double)c.samples_per_pixel));
        d.pixel_samples_scale = 1.0 / (double)(d.sqrt_spp * d.sqrt_spp);
        d.recip_sqrt_spp = 1.0 / (double)d.sqrt_spp;
        Point3 lookfrom(c.lookfrom[0], c.lookfrom[1], c.lookfrom[2]), lookat(c.lookat[0], c.lookat[1], c.lookat[2]);
        Vec3 vup(c.vup[0], c.vup[1], c.vup[2]);
        center = lookfrom;
        double theta = c.vfov * PI / 180.0; // degrees_to_radians (rtweekend.rs:11-13)
        double h = std::tan(theta / 2.0);
        double viewport_height = 2.0 * h * c.focus_dist;
        double viewport_width = viewport_height * ((double)c.image_width / (double)d.image_height);
        Vec3 w = unit_vector(lookfrom - lookat);
        Vec3 u = unit_vector(cross(vup, w));
        Vec3 v = cross(w, u);
        Vec3 viewport_u = viewport_width * u;
        Vec3 viewport_v = -viewport_height * v;
        pixel_delta_u = viewport_u / (double)c.image_width;
        pixel_delta_v = viewport_v / (double)d.image_height;
        Point3 viewport_upper_left = center - (c.focus_dist * w) - viewport_u / 2.0 - viewport_v / 2.0;
        pixel00_loc = viewport_upper_left + 0.5 * (pixel_delta_u + pixel_delta_v);
        double defocus_radius = c.focus_dist * std::tan((c.defocus_angle / 2.0) * PI / 180.0);
        defocus_disk_u = u * defocus_radius;
        defocus_disk_v = v * defocus_radius;
        background = Color(c.background[0], c.background[1], c.background[2]);
        auto put = [](double* o, Vec3 a) { o[0] = a.x; o[1] = a.y; o[2] = a.z; };
        put(d.center, center); put(d.pixel00_loc, pixel00_loc); put(d.pixel_delta_u, pixel_delta_u);
        put(d.pixel_delta_v, pixel_delta_v); put(d.defocus_disk_u, defocus_disk_u); put(d.defocus_disk_v, defocus_disk_v);
    }
    Ray get_ray(int i, int j, int s_i, int s_j) const { // camera.rs:207-225
        double px = (((double)s_i + random_double()) * d.recip_sqrt_spp) - 0.5; // camera.rs:227-231
        double py = (((double)s_j + random_double()) * d.recip_sqrt_spp) - 0.5;
        Point3 pixel_sample = pixel00_loc + (((double)i + px) * pixel_delta_u) + (((double)j + py) * pixel_delta_v);
        Point3 ray_origin;
        if (c.defocus_angle <= 0.0) ray_origin = center;
        else { // camera.rs:237-240
            Vec3 p = random_in_unit_disk();
            ray_origin = center + (p.x * defocus_disk_u) + (p.y * defocus_disk_v);
        }
        Vec3 ray_direction = pixel_sample - ray_origin;
        double ray_time = random_double();
        return Ray(ray_origin, ray_direction, ray_time);
    }
    Color ray_color(const Ray& r, int depth, const Hittable* world, const Hittable* lights) const { // camera.rs:243-339
        if (depth <= 0) return Color(0, 0, 0);
        HitRecord rec;
        ++tl_world_hits;
        if (!world->hit(r, Interval(0.001, INF), rec)) return background;
        ScatterRecord srec;
        Color color_from_emission = rec.mat->emitted(r, rec, rec.u, rec.v, rec.p);
        if (!rec.mat->scatter(r, rec, srec)) return color_from_emission;
        if (srec.skip_pdf) {
            return srec.attenuation * ray_color(srec.skip_pdf_ray, depth - 1, world, lights); // emission dropped (:293-294)
        }
        Vec3 scattered_dir; double pdf_value;
        if (lights) {
            HittablePdf light_pdf(lights, rec.p);
            MixturePdf mixed_pdf(&light_pdf, srec.pdf_ptr.get());
            scattered_dir = mixed_pdf.generate();
            pdf_value = mixed_pdf.value(scattered_dir);
        } else {
            // No light list: the material PDF alone (the reference cannot render such a scene at HEAD;
            // this is the estimator of its commented history, camera.rs:340-428).
            scattered_dir = srec.pdf_ptr->generate();
            pdf_value = srec.pdf_ptr->value(scattered_dir);
        }
        Ray scattered(rec.p, scattered_dir, r.tm);
        double scattering_pdf = rec.mat->scattering_pdf(r, rec, scattered);
        if (c.russian_roulette) { // the rule of the reference's history (camera.rs:407-416): no re-weighting of survivors
            double p = std::fmax(std::fmax(srec.attenuation.x, srec.attenuation.y), srec.attenuation.z);
            p = p < 0.1 ? 0.1 : (p > 1.0 ? 1.0 : p); // f64::clamp(0.1, 1.0)
            if (random_double() >= p) return color_from_emission;
        }
        Color color_from_scatter = (srec.attenuation * scattering_pdf * ray_color(scattered, depth - 1, world, lights)) / pdf_value;
        return color_from_emission + color_from_scatter;
    }
};

static inline double linear_to_gamma(double x) { return x > 0.0 ? std::sqrt(x) : 0.0; } // color.rs:7-12

template <class F>
static void parallel_for(int64_t n, int nthreads, F f) {
    if (nthreads <= 1 || n <= 1) { for (int64_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&]() { for (;;) { int64_t i = next.fetch_add(1); if (i >= n) break; f(i); } });
    for (auto& t : th) t.join();
}

} // namespace oracle

using namespace oracle;

extern "C" {

const char* oracle_last_error(void) { return g_error.c_str(); }

void* oracle_scene_create(const rt_scene_desc* desc) {
    auto* s = new Scene();
    if (!build_scene(desc, *s)) { delete s; return nullptr; }
    return s;
}
void oracle_scene_destroy(void* scene) { delete static_cast<Scene*>(scene); }

int oracle_object_bounds(void* scene, int32_t object, double out[6]) {
    auto* s = static_cast<Scene*>(scene);
    if (object < 0 || (size_t)object >= s->objects.size()) return -1;
    Aabb b = s->objects[object]->bounding_box();
    out[0] = b.x.min; out[1] = b.x.max; out[2] = b.y.min; out[3] = b.y.max; out[4] = b.z.min; out[5] = b.z.max;
    return 0;
}

// Pre-order dump of the BvhNode tree rooted at `object`: one rt_bvh_node per BvhNode; a child
// that is a BvhNode is tagged (RT_REF_NODE<<28 | pre-order index), a primitive by its prim tag,
// anything else by 0xE0000000 (opaque).  Returns the node count (or -1).
static int64_t dump_bvh(const BvhNode* n, std::vector<rt_bvh_node>& out) {
    int64_t me = (int64_t)out.size();
    out.emplace_back();
    auto tag_of = [&](const HittablePtr& h) -> uint32_t {
        if (auto* b = dynamic_cast<const BvhNode*>(h.get())) return (uint32_t)dump_bvh(b, out);
        if (auto* s = dynamic_cast<const Sphere*>(h.get())) return s->tag;
        if (auto* q = dynamic_cast<const Quad*>(h.get())) return q->tag;
        if (auto* t = dynamic_cast<const Triangle*>(h.get())) return t->tag;
        if (auto* m = dynamic_cast<const ConstantMedium*>(h.get())) return m->tag;
        return 0xE0000000u;
    };
    uint32_t l = tag_of(n->left);
    uint32_t r = (n->right.get() == n->left.get()) ? l : tag_of(n->right);
    rt_bvh_node& o = out[(size_t)me];
    o.bounds[0] = n->bbox.x.min; o.bounds[1] = n->bbox.x.max; o.bounds[2] = n->bbox.y.min; o.bounds[3] = n->bbox.y.max;
    o.bounds[4] = n->bbox.z.min; o.bounds[5] = n->bbox.z.max;
    o.left = l; o.right = r; o.flags = 0; o._pad = 0;
    return me;
}
int64_t oracle_bvh_dump(void* scene, int32_t object, rt_bvh_node* out, uint64_t cap) {
    auto* s = static_cast<Scene*>(scene);
    if (object < 0 || (size_t)object >= s->objects.size()) return -1;
    auto* b = dynamic_cast<const BvhNode*>(s->objects[object].get());
    if (!b) return -1;
    std::vector<rt_bvh_node> v;
    dump_bvh(b, v);
    for (size_t i = 0; i < v.size() && i < cap; ++i) out[i] = v[i];
    return (int64_t)v.size();
}

// world.hit(ray, [tmin,tmax], rec) for every ray of the batch (camera.rs:258 form).
// counters[0] += Aabb::hit evaluations, counters[1] += primitive hit evaluations.
int oracle_closest_hit(void* scene, const rt_ray_batch* rays, uint64_t n, rt_hit_batch* hits,
                       uint64_t seed, int nthreads, uint64_t* counters) {
    auto* s = static_cast<Scene*>(scene);
    const Hittable* world = s->world.get();
    std::atomic<uint64_t> nv(0), np(0);
    const int64_t chunk = 4096;
    int64_t nchunks = ((int64_t)n + chunk - 1) / chunk;
    parallel_for(nchunks, nthreads, [&](int64_t ci) {
        tl_node_visits = 0; tl_prim_tests = 0;
        uint64_t lo = (uint64_t)ci * chunk, hi = std::min<uint64_t>(n, lo + chunk);
        for (uint64_t i = lo; i < hi; ++i) {
            Rng rng; rng.seed = seed; rng.stream = 2; rng.a = (uint32_t)i; rng.b = (uint32_t)(i >> 32);
            tl_rng = &rng;
            Ray r(Point3(rays->ox[i], rays->oy[i], rays->oz[i]), Vec3(rays->dx[i], rays->dy[i], rays->dz[i]),
                  rays->time ? rays->time[i] : 0.0);
            Interval iv(rays->tmin ? rays->tmin[i] : 0.001, rays->tmax ? rays->tmax[i] : INF);
            HitRecord rec;
            bool ok = world->hit(r, iv, rec);
            hits->t[i] = ok ? rec.t : INF;
            hits->prim[i] = ok ? rec.prim : RT_REF_NONE;
            if (hits->u) hits->u[i] = ok ? rec.u : 0.0;
            if (hits->v) hits->v[i] = ok ? rec.v : 0.0;
            if (hits->front_face) hits->front_face[i] = ok ? (rec.front_face ? 1 : 0) : 0;
            if (hits->mat) hits->mat[i] = ok ? rec.mat_id : -1;
            if (hits->nx) hits->nx[i] = ok ? rec.normal.x : 0.0;
            if (hits->ny) hits->ny[i] = ok ? rec.normal.y : 0.0;
            if (hits->nz) hits->nz[i] = ok ? rec.normal.z : 0.0;
            if (hits->px) hits->px[i] = ok ? rec.p.x : 0.0;
            if (hits->py) hits->py[i] = ok ? rec.p.y : 0.0;
            if (hits->pz) hits->pz[i] = ok ? rec.p.z : 0.0;
            tl_rng = nullptr;
        }
        nv += tl_node_visits; np += tl_prim_tests;
    });
    if (counters) { counters[0] += nv.load(); counters[1] += np.load(); }
    return 0;
}

int oracle_camera_initialize(const rt_camera* cam, rt_camera_derived* out) {
    Camera c; c.c = *cam; c.initialize(); *out = c.d; return 0;
}

// The sampling loop of Camera::render (camera.rs:99-131) over one work unit, rows in parallel
// (camera.rs:99 uses rayon the same way).  accum layout as rt_render.  n_rays += world.hit calls.
int oracle_render(void* scene, const rt_camera* cam, uint64_t seed, const rt_work* work, double* accum,
                  int nthreads, uint64_t* n_rays) {
    auto* s = static_cast<Scene*>(scene);
    Camera c; c.c = *cam; c.initialize();
    rt_work w;
    if (work) w = *work; else { w.x0 = 0; w.y0 = 0; w.x1 = cam->image_width; w.y1 = c.d.image_height; w.sample_begin = 0; w.sample_end = c.d.sqrt_spp * c.d.sqrt_spp; }
    const Hittable* world = s->world.get();
    const Hittable* lights = s->lights.get();
    if (lights) { // an empty light list panics in the reference (hittable_list.rs:79)
        if (auto* l = dynamic_cast<const HittableList*>(lights)) if (l->objects.empty()) { g_error = "render: empty light list"; return -1; }
    }
    std::atomic<uint64_t> rays(0);
    parallel_for((int64_t)(w.y1 - w.y0), nthreads, [&](int64_t row) {
        int j = w.y0 + (int)row;
        tl_world_hits = 0;
        for (int i = w.x0; i < w.x1; ++i) {
            Color pixel_color(0, 0, 0);
            for (int sidx = w.sample_begin; sidx < w.sample_end; ++sidx) {
                int s_j = sidx / c.d.sqrt_spp, s_i = sidx % c.d.sqrt_spp;
                Rng rng; rng.seed = seed; rng.stream = 0; rng.a = (uint32_t)(j * cam->image_width + i); rng.b = (uint32_t)sidx;
                tl_rng = &rng;
                Ray r = c.get_ray(i, j, s_i, s_j);
                Color col = c.ray_color(r, cam->max_depth, world, lights);
                pixel_color = pixel_color + col;
                tl_rng = nullptr;
            }
            double* a = accum + ((size_t)j * (size_t)cam->image_width + (size_t)i) * 3;
            a[0] += pixel_color.x; a[1] += pixel_color.y; a[2] += pixel_color.z;
        }
        rays += tl_world_hits;
    });
    if (n_rays) *n_rays += rays.load();
    return 0;
}

// camera.rs:133-135 + color.rs:58-80.
int oracle_resolve_rgb8(const rt_camera* cam, const double* accum, uint8_t* rgb) {
    Camera c; c.c = *cam; c.initialize();
    size_t npix = (size_t)cam->image_width * (size_t)c.d.image_height;
    Interval intensity(0.000, 0.999);
    for (size_t p = 0; p < npix; ++p)
        for (int k = 0; k < 3; ++k) {
            double x = c.d.pixel_samples_scale * accum[p * 3 + k];
            x = linear_to_gamma(x);
            int32_t b = sat_i32(256.0 * intensity.clamp(x));
            rgb[p * 3 + k] = (uint8_t)b;
        }
    return 0;
}

double oracle_rng_uniform(uint64_t seed, uint32_t stream, uint32_t a, uint32_t b, uint64_t index) {
    return rng_uniform(seed, stream, a, b, index);
}
void oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = c[i];
}

// Small probes for known-answer tests of individual reference formulas.
double oracle_quad_pdf_value(void* scene, int32_t quad, const double o[3], const double d[3]) {
    auto* s = static_cast<Scene*>(scene);
    return s->quads[quad]->pdf_value(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2]));
}
double oracle_sphere_pdf_value(void* scene, int32_t sphere, const double o[3], const double d[3]) {
    auto* s = static_cast<Scene*>(scene);
    return s->spheres[sphere]->pdf_value(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2]));
}
void oracle_texture_value(void* scene, int32_t tex, double u, double v, const double p[3], double out[3]) {
    auto* s = static_cast<Scene*>(scene);
    Color c = s->textures[tex]->value(u, v, Point3(p[0], p[1], p[2]));
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
int oracle_aabb_hit(const double bounds[6], const double o[3], const double d[3], double tmin, double tmax) {
    Aabb b(Interval(bounds[0], bounds[1]), Interval(bounds[2], bounds[3]), Interval(bounds[4], bounds[5]));
    return b.hit(Ray(Point3(o[0], o[1], o[2]), Vec3(d[0], d[1], d[2])), Interval(tmin, tmax)) ? 1 : 0;
}

} // extern "C"
