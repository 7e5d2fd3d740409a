// mi355rt_host.hpp -- C++ host mirror of the reference's scene-facing API above the C ABI.
//
// The reference is a compiled (Rust) program whose hot path is entered through
//   trait Hittable (hittable.rs:37-57), BvhNode::new (bvh.rs:14-18), Camera::render (camera.rs:78-158).
// This header keeps those names, constructor arguments and error behaviour in C++17 so that a scene
// script reads like the reference's main.rs, and turns the object graph into the POD arrays of
// rt_scene_desc (include/mi355rt.h).  Nothing here intersects rays: `hit` is answered in batches by
// libmi355rt.so (Scene::closest_hit), the sampling loop by rt_render (Camera::render).
#ifndef MI355RT_HOST_HPP
#define MI355RT_HOST_HPP

#include <algorithm>
#include <array>
#include <fstream>
#include <sstream>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"

namespace rt_host {

// ------------------------------------------------------------------ rtweekend.rs
inline constexpr double PI = 3.1415926535897932385; // rtweekend.rs:7
struct SceneRng { uint64_t seed = 0, index = 0; };
inline SceneRng& scene_rng() { static SceneRng r; return r; }
inline void seed(uint64_t s) { scene_rng() = SceneRng{s, 0}; }
inline double random_double() { auto& r = scene_rng(); return rt_rng_uniform(r.seed, 1, 0, 0, r.index++); } // rtweekend.rs:15-17
inline double random_double_range(double lo, double hi) { return lo + (hi - lo) * random_double(); }         // rtweekend.rs:19-21
inline int random_int(int lo, int hi) { return (int)random_double_range((double)lo, (double)(hi + 1)); }      // rtweekend.rs:23-25

// ------------------------------------------------------------------ vec3 (reconstructed API)
struct Vec3 {
    double e[3] = {0, 0, 0};
    Vec3() {}
    Vec3(double x, double y, double z) : e{x, y, z} {}
    double x() const { return e[0]; }
    double y() const { return e[1]; }
    double z() const { return e[2]; }
    double length_squared() const { return e[0] * e[0] + e[1] * e[1] + e[2] * e[2]; }
    double length() const { return std::sqrt(length_squared()); }
    static Vec3 random() { double a = random_double(), b = random_double(), c = random_double(); return Vec3(a, b, c); }
    static Vec3 random_range(double lo, double hi) {
        double a = random_double_range(lo, hi), b = random_double_range(lo, hi), c = random_double_range(lo, hi);
        return Vec3(a, b, c);
    }
};
using Point3 = Vec3;
using Color = Vec3; // color.rs:5
inline Vec3 operator-(Vec3 a) { return Vec3(-a.e[0], -a.e[1], -a.e[2]); }
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
inline Vec3 operator*(double t, Vec3 v) { return Vec3(t * v.e[0], t * v.e[1], t * v.e[2]); }
inline Vec3 operator*(Vec3 v, double t) { return t * v; }
inline Vec3 operator/(Vec3 v, double t) { return (1.0 / t) * v; }
inline Vec3 unit_vector(Vec3 v) { return v / v.length(); }

// ------------------------------------------------------------------ texture.rs / perlin.rs
struct Texture { virtual ~Texture() {} };
using TexturePtr = std::shared_ptr<Texture>;
struct SolidColor : Texture { Color albedo; explicit SolidColor(Color c) : albedo(c) {} };
struct CheckerTexture : Texture {
    double scale; TexturePtr even, odd;
    CheckerTexture(double s, TexturePtr e, TexturePtr o) : scale(s), even(e), odd(o) {}
    static std::shared_ptr<CheckerTexture> from_colors(double s, Color c1, Color c2) {
        return std::make_shared<CheckerTexture>(s, std::make_shared<SolidColor>(c1), std::make_shared<SolidColor>(c2));
    }
};
struct ImageTexture : Texture { // RGB8 rows (rtw_stb_image.rs:110,155-171); width == 0: no image (cyan)
    int width = 0, height = 0; std::vector<uint8_t> pixels;
    ImageTexture() {}
    ImageTexture(int w, int h, std::vector<uint8_t> px) : width(w), height(h), pixels(std::move(px)) {}
};
struct Perlin { // perlin.rs:18-53
    rt_perlin t;
    Perlin() {
        for (int i = 0; i < 256; ++i) { Vec3 v = unit_vector(Vec3::random_range(-1.0, 1.0)); for (int k = 0; k < 3; ++k) t.randvec[i][k] = v.e[k]; }
        generate_perm(t.perm_x); generate_perm(t.perm_y); generate_perm(t.perm_z);
    }
    static void generate_perm(int32_t* p) {
        for (int i = 0; i < 256; ++i) p[i] = i;
        for (int i = 255; i > 0; --i) { int target = random_int(0, i); std::swap(p[i], p[target]); }
    }
};
struct NoiseTexture : Texture { Perlin noise; double scale; explicit NoiseTexture(double s) : scale(s) {} };

// ------------------------------------------------------------------ material.rs
struct Material { virtual ~Material() {} };
using MaterialPtr = std::shared_ptr<Material>;
struct Lambertian : Material {
    TexturePtr tex;
    explicit Lambertian(Color albedo) : tex(std::make_shared<SolidColor>(albedo)) {}
    explicit Lambertian(TexturePtr t) : tex(t) {} // from_texture
};
struct Metal : Material { Color albedo; double fuzz; Metal(Color a, double f) : albedo(a), fuzz(std::fmin(f, 1.0)) {} };
struct Dielectric : Material { double refraction_index; explicit Dielectric(double ri) : refraction_index(ri) {} };
struct DiffuseLight : Material {
    TexturePtr tex;
    explicit DiffuseLight(TexturePtr t) : tex(t) {}
    explicit DiffuseLight(Color emit) : tex(std::make_shared<SolidColor>(emit)) {} // from_color
};
struct Isotropic : Material {
    TexturePtr tex;
    explicit Isotropic(TexturePtr t) : tex(t) {}
    explicit Isotropic(Color albedo) : tex(std::make_shared<SolidColor>(albedo)) {}
};

// ------------------------------------------------------------------ hittables
struct Hittable { virtual ~Hittable() {} };
using HittablePtr = std::shared_ptr<Hittable>;
struct Sphere : Hittable { // sphere.rs:25-45
    Point3 center1, center2; double radius; MaterialPtr mat; bool moving;
    Sphere(Point3 c, double r, MaterialPtr m) : center1(c), center2(c), radius(r), mat(m), moving(false) {}
    Sphere(Point3 c1, Point3 c2, double r, MaterialPtr m) : center1(c1), center2(c2), radius(r), mat(m), moving(true) {} // new_moving
};
struct Quad : Hittable { Point3 q; Vec3 u, v; MaterialPtr mat; Quad(Point3 q_, Vec3 u_, Vec3 v_, MaterialPtr m) : q(q_), u(u_), v(v_), mat(m) {} };
struct Triangle : Hittable { Point3 v0, v1, v2; MaterialPtr mat; Triangle(Point3 a, Point3 b, Point3 c, MaterialPtr m) : v0(a), v1(b), v2(c), mat(m) {} };
struct HittableList : Hittable { // hittable_list.rs:15-41
    std::vector<HittablePtr> objects;
    HittableList() {}
    explicit HittableList(HittablePtr o) { add(o); }
    void clear() { objects.clear(); }
    void add(HittablePtr o) { objects.push_back(o); }
};
struct BvhNode : Hittable { // BvhNode::new(&list) (bvh.rs:14-18); the split itself happens in the library
    std::vector<HittablePtr> objects;
    explicit BvhNode(const HittableList& l) : objects(l.objects) {}
};
struct Translate : Hittable { HittablePtr object; Vec3 offset; Translate(HittablePtr o, Vec3 off) : object(o), offset(off) {} };
struct RotateY : Hittable { HittablePtr object; double angle; RotateY(HittablePtr o, double deg) : object(o), angle(deg) {} };
struct ConstantMedium : Hittable { // constant_medium.rs:20-46
    HittablePtr boundary; double density; MaterialPtr phase_function;
    ConstantMedium(HittablePtr b, double d, Color albedo) : boundary(b), density(d), phase_function(std::make_shared<Isotropic>(albedo)) {}
    ConstantMedium(HittablePtr b, double d, TexturePtr tex) : boundary(b), density(d), phase_function(std::make_shared<Isotropic>(tex)) {}
};
struct TriangleSet : Hittable { // Mesh (mesh.rs:95-183, bvh = true) / ObjModel (obj_loader.rs:13-106, bvh = false)
    std::vector<rt_triangle> triangles; MaterialPtr mat; bool bvh;
    TriangleSet(std::vector<rt_triangle> t, MaterialPtr m, bool b) : triangles(std::move(t)), mat(m), bvh(b) {}
};

// Minimal Wavefront OBJ reader for mesh.rs:101-167 / obj_loader.rs:21-83.  The reference goes through the
// third-party crate tobj (triangulate + single_index); what reaches the hot path is a triangle list whose
// positions went through f32 (mesh.rs:137-153 widens them to f64).  Kept here: `v` lines parsed to f32,
// faces fan-triangulated (v0, v_i, v_i+1), negative indices relative to the end.  Parity with tobj on real
// assets is unpinned: the reference ships no model file.
inline std::vector<rt_triangle> load_obj_triangles(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error(path + ": cannot open");
    std::vector<std::array<float, 3>> pos;
    std::vector<rt_triangle> tris;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v") {
            double x, y, z;
            if (!(ls >> x >> y >> z)) throw std::runtime_error(path + ": malformed vertex line: " + line);
            pos.push_back({(float)x, (float)y, (float)z});
        } else if (tag == "f") {
            std::vector<long> idx;
            std::string tok;
            while (ls >> tok) {
                const long i = std::stol(tok.substr(0, tok.find('/')));
                idx.push_back(i > 0 ? i - 1 : (long)pos.size() + i);
            }
            for (size_t k = 1; k + 1 < idx.size(); ++k) {
                const long f[3] = {idx[0], idx[k], idx[k + 1]};
                rt_triangle t{};
                for (int c = 0; c < 3; ++c) {
                    if (f[c] < 0 || f[c] >= (long)pos.size()) throw std::runtime_error(path + ": face index out of range");
                    double* dst = c == 0 ? t.v0 : (c == 1 ? t.v1 : t.v2);
                    for (int a = 0; a < 3; ++a) dst[a] = (double)pos[(size_t)f[c]][a];
                }
                t.mat = -1;
                tris.push_back(t);
            }
        }
    }
    return tris;
}
// Mesh::from_obj (mesh.rs:101-167): a BVH over the triangles of the file.
inline std::shared_ptr<TriangleSet> mesh_from_obj(const std::string& path, MaterialPtr mat) {
    return std::make_shared<TriangleSet>(load_obj_triangles(path), mat, true);
}
// ObjModel::load (obj_loader.rs:21-83): p * scale + offset per component, linear scan over the triangles.
inline std::shared_ptr<TriangleSet> obj_model_load(const std::string& path, MaterialPtr mat, double scale, Vec3 offset) {
    std::vector<rt_triangle> t = load_obj_triangles(path);
    for (rt_triangle& tr : t)
        for (int a = 0; a < 3; ++a) {
            tr.v0[a] = tr.v0[a] * scale + offset.e[a];
            tr.v1[a] = tr.v1[a] * scale + offset.e[a];
            tr.v2[a] = tr.v2[a] * scale + offset.e[a];
        }
    return std::make_shared<TriangleSet>(std::move(t), mat, false);
}

inline std::shared_ptr<HittableList> box_new(Point3 a, Point3 b, MaterialPtr mat) { // quad.rs:139-193
    auto sides = std::make_shared<HittableList>();
    Point3 mn(std::fmin(a.x(), b.x()), std::fmin(a.y(), b.y()), std::fmin(a.z(), b.z()));
    Point3 mx(std::fmax(a.x(), b.x()), std::fmax(a.y(), b.y()), std::fmax(a.z(), b.z()));
    Vec3 dx(mx.x() - mn.x(), 0.0, 0.0), dy(0.0, mx.y() - mn.y(), 0.0), dz(0.0, 0.0, mx.z() - mn.z());
    sides->add(std::make_shared<Quad>(Point3(mn.x(), mn.y(), mx.z()), dx, dy, mat));
    sides->add(std::make_shared<Quad>(Point3(mx.x(), mn.y(), mx.z()), -dz, dy, mat));
    sides->add(std::make_shared<Quad>(Point3(mx.x(), mn.y(), mn.z()), -dx, dy, mat));
    sides->add(std::make_shared<Quad>(Point3(mn.x(), mn.y(), mn.z()), dz, dy, mat));
    sides->add(std::make_shared<Quad>(Point3(mn.x(), mx.y(), mx.z()), dx, -dz, mat));
    sides->add(std::make_shared<Quad>(Point3(mn.x(), mn.y(), mn.z()), dx, dz, mat));
    return sides;
}

struct RtError : std::runtime_error {
    int code;
    RtError(int c, const std::string& m) : std::runtime_error("mi355rt error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int rc) { if (rc != RT_OK) throw RtError(rc, rt_last_error()); }

// ------------------------------------------------------------------ object graph -> rt_scene_desc
class SceneDesc {
  public:
    std::vector<rt_sphere> spheres; std::vector<rt_quad> quads; std::vector<rt_triangle> triangles;
    std::vector<rt_object> objects; std::vector<int32_t> children; std::vector<rt_material> materials;
    std::vector<rt_texture> textures; std::vector<rt_image> images; std::vector<uint8_t> image_bytes;
    std::vector<rt_perlin> perlins;
    rt_scene_desc d{};

    SceneDesc(HittablePtr world, HittablePtr lights) {
        d.world = object(world);
        d.lights = lights ? object(lights) : -1;
        d.spheres = spheres.data(); d.n_spheres = spheres.size();
        d.quads = quads.data(); d.n_quads = quads.size();
        d.triangles = triangles.data(); d.n_triangles = triangles.size();
        d.objects = objects.data(); d.n_objects = objects.size();
        d.children = children.data(); d.n_children = children.size();
        d.materials = materials.data(); d.n_materials = materials.size();
        d.textures = textures.data(); d.n_textures = textures.size();
        d.images = images.data(); d.n_images = images.size();
        d.image_bytes = image_bytes.data(); d.n_image_bytes = image_bytes.size();
        d.perlins = perlins.data(); d.n_perlins = perlins.size();
    }

  private:
    std::map<const void*, int32_t> memo_;
    std::vector<std::shared_ptr<void>> keep_;
    static void put(double* o, Vec3 v) { o[0] = v.e[0]; o[1] = v.e[1]; o[2] = v.e[2]; }

    int32_t texture(const TexturePtr& t) {
        auto it = memo_.find(t.get());
        if (it != memo_.end()) return it->second;
        rt_texture r{};
        r.even = r.odd = r.image = r.perlin = -1;
        if (auto* s = dynamic_cast<SolidColor*>(t.get())) { r.kind = RT_TEX_SOLID; put(r.color, s->albedo); }
        else if (auto* c = dynamic_cast<CheckerTexture*>(t.get())) { r.kind = RT_TEX_CHECKER; r.even = texture(c->even); r.odd = texture(c->odd); r.scale = c->scale; }
        else if (auto* im = dynamic_cast<ImageTexture*>(t.get())) {
            r.kind = RT_TEX_IMAGE;
            if (im->width > 0) {
                r.image = (int32_t)images.size();
                images.push_back(rt_image{im->width, im->height, (uint64_t)image_bytes.size()});
                image_bytes.insert(image_bytes.end(), im->pixels.begin(), im->pixels.end());
            }
        } else if (auto* n = dynamic_cast<NoiseTexture*>(t.get())) { r.kind = RT_TEX_NOISE; r.perlin = (int32_t)perlins.size(); perlins.push_back(n->noise.t); r.scale = n->scale; }
        else throw std::invalid_argument("unknown texture type");
        textures.push_back(r);
        keep_.push_back(t);
        return memo_[t.get()] = (int32_t)textures.size() - 1;
    }
    int32_t material(const MaterialPtr& m) {
        auto it = memo_.find(m.get());
        if (it != memo_.end()) return it->second;
        rt_material r{};
        r.tex = -1;
        if (auto* l = dynamic_cast<Lambertian*>(m.get())) { r.kind = RT_MAT_LAMBERTIAN; r.tex = texture(l->tex); }
        else if (auto* me = dynamic_cast<Metal*>(m.get())) { r.kind = RT_MAT_METAL; put(r.albedo, me->albedo); r.fuzz = me->fuzz; }
        else if (auto* di = dynamic_cast<Dielectric*>(m.get())) { r.kind = RT_MAT_DIELECTRIC; r.ior = di->refraction_index; }
        else if (auto* dl = dynamic_cast<DiffuseLight*>(m.get())) { r.kind = RT_MAT_DIFFUSE_LIGHT; r.tex = texture(dl->tex); }
        else if (auto* is = dynamic_cast<Isotropic*>(m.get())) { r.kind = RT_MAT_ISOTROPIC; r.tex = texture(is->tex); }
        else throw std::invalid_argument("unknown material type");
        materials.push_back(r);
        keep_.push_back(m);
        return memo_[m.get()] = (int32_t)materials.size() - 1;
    }
    int32_t emit(int32_t kind, int32_t a = 0, int32_t b = 0, int32_t child_kind = 0, int32_t mat = -1, Vec3 p = Vec3()) {
        rt_object o{};
        o.kind = kind; o.a = a; o.b = b; o.child_kind = child_kind; o.mat = mat; put(o.param, p);
        objects.push_back(o);
        return (int32_t)objects.size() - 1;
    }
    int32_t object(const HittablePtr& h) { // children are emitted before their parent
        auto it = memo_.find(h.get());
        if (it != memo_.end()) return it->second;
        int32_t id;
        if (auto* s = dynamic_cast<Sphere*>(h.get())) {
            rt_sphere r{};
            put(r.center, s->center1); put(r.center2, s->center2); r.radius = s->radius; r.mat = material(s->mat); r.moving = s->moving ? 1 : 0;
            spheres.push_back(r);
            id = emit(RT_OBJ_SPHERE, (int32_t)spheres.size() - 1);
        } else if (auto* q = dynamic_cast<Quad*>(h.get())) {
            rt_quad r{};
            put(r.q, q->q); put(r.u, q->u); put(r.v, q->v); r.mat = material(q->mat);
            quads.push_back(r);
            id = emit(RT_OBJ_QUAD, (int32_t)quads.size() - 1);
        } else if (auto* t = dynamic_cast<Triangle*>(h.get())) {
            rt_triangle r{};
            put(r.v0, t->v0); put(r.v1, t->v1); put(r.v2, t->v2); r.mat = material(t->mat);
            triangles.push_back(r);
            id = emit(RT_OBJ_TRIANGLE, (int32_t)triangles.size() - 1);
        } else if (auto* l = dynamic_cast<HittableList*>(h.get())) {
            id = group(RT_OBJ_LIST, l->objects);
        } else if (auto* b = dynamic_cast<BvhNode*>(h.get())) {
            id = group(RT_OBJ_BVH, b->objects);
        } else if (auto* tr = dynamic_cast<Translate*>(h.get())) {
            id = emit(RT_OBJ_TRANSLATE, object(tr->object), 0, 0, -1, tr->offset);
        } else if (auto* ro = dynamic_cast<RotateY*>(h.get())) {
            id = emit(RT_OBJ_ROTATE_Y, object(ro->object), 0, 0, -1, Vec3(ro->angle, 0, 0));
        } else if (auto* cm = dynamic_cast<ConstantMedium*>(h.get())) {
            int32_t b2 = object(cm->boundary);
            id = emit(RT_OBJ_CONSTANT_MEDIUM, b2, 0, 0, material(cm->phase_function), Vec3(cm->density, 0, 0));
        } else if (auto* ts = dynamic_cast<TriangleSet*>(h.get())) {
            int32_t first = (int32_t)triangles.size(), m = material(ts->mat);
            for (rt_triangle t2 : ts->triangles) { t2.mat = m; triangles.push_back(t2); }
            id = ts->bvh ? emit(RT_OBJ_BVH, first, (int32_t)ts->triangles.size(), RT_CHILD_TRIANGLES)
                         : emit(RT_OBJ_OBJMODEL, first, (int32_t)ts->triangles.size());
        } else throw std::invalid_argument("unknown hittable type");
        keep_.push_back(h);
        return memo_[h.get()] = id;
    }
    int32_t group(int32_t kind, const std::vector<HittablePtr>& objs) {
        std::vector<int32_t> ids;
        for (auto& o : objs) ids.push_back(object(o));
        int32_t first = (int32_t)children.size();
        children.insert(children.end(), ids.begin(), ids.end());
        return emit(kind, first, (int32_t)ids.size(), RT_CHILD_OBJECTS);
    }
};

class Scene { // a flattened scene resident on the current HIP device
  public:
    Scene(HittablePtr world, HittablePtr lights = nullptr, uint32_t flags = 0) : desc_(world, lights) { check(rt_scene_create(&desc_.d, flags, &h_)); }
    ~Scene() { rt_scene_destroy(h_); }
    Scene(const Scene&) = delete;
    Scene& operator=(const Scene&) = delete;
    rt_scene* handle() const { return h_; }
    const SceneDesc& desc() const { return desc_; }
  private:
    SceneDesc desc_;
    rt_scene* h_ = nullptr;
};

// ------------------------------------------------------------------ camera.rs
struct Camera {
    double aspect_ratio = 1.0; int image_width = 100; int samples_per_pixel = 10; int max_depth = 10; // camera.rs:47-74
    Color background = Color(0, 0, 0); double vfov = 90.0;
    Point3 lookfrom = Point3(0, 0, 0), lookat = Point3(0, 0, -1); Vec3 vup = Vec3(0, 1, 0);
    double defocus_angle = 0.0, focus_dist = 10.0;

    rt_camera to_struct() const {
        rt_camera c{};
        c.aspect_ratio = aspect_ratio; c.image_width = image_width; c.samples_per_pixel = samples_per_pixel; c.max_depth = max_depth;
        for (int k = 0; k < 3; ++k) { c.background[k] = background.e[k]; c.lookfrom[k] = lookfrom.e[k]; c.lookat[k] = lookat.e[k]; c.vup[k] = vup.e[k]; }
        c.vfov = vfov; c.defocus_angle = defocus_angle; c.focus_dist = focus_dist;
        return c;
    }
    // Camera::render (camera.rs:78-158): samples on the GPU, resolves, writes the P3 text to `out_ppm`
    // (stdout in the reference).  `lights == nullptr`: material PDF only.  Returns the 8-bit image.
    std::vector<uint8_t> render(HittablePtr world, HittablePtr lights, uint64_t seed = 0, const char* out_ppm = nullptr) const {
        Scene scene(world, lights);
        rt_camera c = to_struct();
        rt_camera_derived d;
        check(rt_camera_initialize(&c, &d));
        std::vector<double> accum((size_t)image_width * d.image_height * 3, 0.0);
        check(rt_render(scene.handle(), &c, seed, nullptr, accum.data()));
        std::vector<uint8_t> rgb(accum.size());
        check(rt_resolve_rgb8(&c, accum.data(), rgb.data()));
        if (out_ppm) check(rt_write_ppm(out_ppm, image_width, d.image_height, rgb.data()));
        std::fprintf(stderr, "\rDone.                 \n"); // camera.rs:156
        return rgb;
    }
};

} // namespace rt_host
#endif
