// scenes.hpp -- the scene scripts of the reference's main.rs written against the C++ host mirror.
// Each function fills (world, lights, cam); lights stays null where the scene predates the light list.
#ifndef MI355RT_SCENES_HPP
#define MI355RT_SCENES_HPP

#include "mi355rt_host.hpp"

namespace rt_host {

struct SceneTuple { HittablePtr world; HittablePtr lights; Camera cam; };

template <class T, class... A> std::shared_ptr<T> mk(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }

inline Camera sky_camera(double vfov, Point3 lookfrom, Point3 lookat, double aspect = 16.0 / 9.0) {
    Camera cam;
    cam.aspect_ratio = aspect; cam.image_width = 400; cam.samples_per_pixel = 100; cam.max_depth = 50;
    cam.background = Color(0.70, 0.80, 1.00);
    cam.vfov = vfov; cam.lookfrom = lookfrom; cam.lookat = lookat; cam.vup = Vec3(0.0, 1.0, 0.0);
    cam.defocus_angle = 0.0;
    return cam;
}

inline Camera cornell_camera(int width, int spp, int depth) {
    Camera cam;
    cam.aspect_ratio = 1.0; cam.image_width = width; cam.samples_per_pixel = spp; cam.max_depth = depth;
    cam.background = Color(0.0, 0.0, 0.0);
    cam.vfov = 40.0; cam.lookfrom = Point3(278.0, 278.0, -800.0); cam.lookat = Point3(278.0, 278.0, 0.0); cam.vup = Vec3(0.0, 1.0, 0.0);
    cam.defocus_angle = 0.0;
    return cam;
}

// last_picture_the_first_book (main.rs:99-202), static grey-ground variant (main.rs:102-107,143).
inline SceneTuple book1_final(uint64_t seed_, int width = 1200, int spp = 10, int depth = 50) {
    seed(seed_);
    auto world = mk<HittableList>();
    world->add(mk<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, mk<Lambertian>(Color(0.5, 0.5, 0.5))));
    for (int a = -11; a < 11; ++a)
        for (int b = -11; b < 11; ++b) {
            double choose_mat = random_double();
            double cx = a + 0.9 * random_double();
            double cz = b + 0.9 * random_double();
            Point3 center(cx, 0.2, cz);
            if ((center - Point3(4.0, 0.2, 0.0)).length() > 0.9) {
                if (choose_mat < 0.8) {
                    Color c1 = Color::random();
                    Color c2 = Color::random();
                    world->add(mk<Sphere>(center, 0.2, mk<Lambertian>(c1 * c2)));
                } else if (choose_mat < 0.95) {
                    Color albedo = Color::random_range(0.5, 1.0);
                    double fuzz = random_double_range(0.0, 0.5);
                    world->add(mk<Sphere>(center, 0.2, mk<Metal>(albedo, fuzz)));
                } else {
                    world->add(mk<Sphere>(center, 0.2, mk<Dielectric>(1.5)));
                }
            }
        }
    world->add(mk<Sphere>(Point3(0.0, 1.0, 0.0), 1.0, mk<Dielectric>(1.5)));
    world->add(mk<Sphere>(Point3(-4.0, 1.0, 0.0), 1.0, mk<Lambertian>(Color(0.4, 0.2, 0.1))));
    world->add(mk<Sphere>(Point3(4.0, 1.0, 0.0), 1.0, mk<Metal>(Color(0.7, 0.6, 0.5), 0.0)));
    auto world_bvh = mk<HittableList>(mk<BvhNode>(*world)); // main.rs:178-180
    Camera cam = sky_camera(20.0, Point3(13.0, 2.0, 3.0), Point3(0.0, 0.0, 0.0));
    cam.image_width = width; cam.samples_per_pixel = spp; cam.max_depth = depth;
    cam.defocus_angle = 0.6; cam.focus_dist = 10.0;
    return {world_bvh, nullptr, cam};
}

inline SceneTuple quads() { // main.rs:305-366
    auto world = mk<HittableList>();
    world->add(mk<Quad>(Point3(-3.0, -2.0, 5.0), Vec3(0.0, 0.0, -4.0), Vec3(0.0, 4.0, 0.0), mk<Lambertian>(Color(1.0, 0.2, 0.2))));
    world->add(mk<Quad>(Point3(-2.0, -2.0, 0.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 4.0, 0.0), mk<Lambertian>(Color(0.2, 1.0, 0.2))));
    world->add(mk<Quad>(Point3(3.0, -2.0, 1.0), Vec3(0.0, 0.0, 4.0), Vec3(0.0, 4.0, 0.0), mk<Lambertian>(Color(0.2, 0.2, 1.0))));
    world->add(mk<Quad>(Point3(-2.0, 3.0, 1.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 0.0, 4.0), mk<Lambertian>(Color(1.0, 0.5, 0.0))));
    world->add(mk<Quad>(Point3(-2.0, -3.0, 5.0), Vec3(4.0, 0.0, 0.0), Vec3(0.0, 0.0, -4.0), mk<Lambertian>(Color(0.2, 0.8, 0.8))));
    auto world_bvh = mk<HittableList>(mk<BvhNode>(*world));
    return {world_bvh, nullptr, sky_camera(80.0, Point3(0.0, 0.0, 9.0), Point3(0.0, 0.0, 0.0), 1.0)};
}

inline SceneTuple simple_light(uint64_t seed_) { // main.rs:368-419
    seed(seed_);
    auto world = mk<HittableList>();
    auto mat = mk<Lambertian>(TexturePtr(mk<NoiseTexture>(4.0)));
    world->add(mk<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, mat));
    world->add(mk<Sphere>(Point3(0.0, 2.0, 0.0), 2.0, mat));
    auto difflight = mk<DiffuseLight>(Color(4.0, 4.0, 4.0));
    world->add(mk<Sphere>(Point3(0.0, 7.0, 0.0), 2.0, difflight));
    world->add(mk<Quad>(Point3(3.0, 1.0, -2.0), Vec3(2.0, 0.0, 0.0), Vec3(0.0, 2.0, 0.0), difflight));
    Camera cam = sky_camera(20.0, Point3(26.0, 3.0, 6.0), Point3(0.0, 2.0, 0.0));
    cam.background = Color(0.0, 0.0, 0.0);
    return {world, nullptr, cam};
}

// cornell_box (main.rs:421-551): the scene live at HEAD; lights = {ceiling quad, glass sphere}.
inline SceneTuple cornell_box(int width = 600, int spp = 1000, int depth = 50) {
    auto world = mk<HittableList>();
    auto red = mk<Lambertian>(Color(0.65, 0.05, 0.05));
    auto white = mk<Lambertian>(Color(0.73, 0.73, 0.73));
    auto green = mk<Lambertian>(Color(0.12, 0.45, 0.15));
    auto light = mk<DiffuseLight>(Color(15.0, 15.0, 15.0));
    world->add(mk<Quad>(Point3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), green));
    world->add(mk<Quad>(Point3(0.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), red));
    world->add(mk<Quad>(Point3(0.0, 0.0, 0.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 0.0, 555.0), white));
    world->add(mk<Quad>(Point3(555.0, 555.0, 555.0), Vec3(-555.0, 0.0, 0.0), Vec3(0.0, 0.0, -555.0), white));
    world->add(mk<Quad>(Point3(0.0, 0.0, 555.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), white));
    world->add(mk<Quad>(Point3(343.0, 554.0, 332.0), Vec3(-130.0, 0.0, 0.0), Vec3(0.0, 0.0, -105.0), light));
    HittablePtr box1 = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), white);
    box1 = mk<RotateY>(box1, 15.0);
    box1 = mk<Translate>(box1, Vec3(265.0, 0.0, 295.0));
    world->add(box1);
    world->add(mk<Sphere>(Point3(190.0, 90.0, 190.0), 90.0, mk<Dielectric>(1.5)));
    auto empty_material = mk<Lambertian>(Color(0.73, 0.73, 0.73));
    auto lights = mk<HittableList>();
    lights->add(mk<Quad>(Point3(343.0, 554.0, 332.0), Vec3(-130.0, 0.0, 0.0), Vec3(0.0, 0.0, -105.0), empty_material));
    lights->add(mk<Sphere>(Point3(190.0, 90.0, 190.0), 90.0, empty_material));
    return {world, lights, cornell_camera(width, spp, depth)};
}

// cornell_smoke (main.rs:553-649): two constant media bounded by instanced boxes, inside a BVH.
inline SceneTuple cornell_smoke(int width = 600, int spp = 200, int depth = 50) {
    auto world = mk<HittableList>();
    auto red = mk<Lambertian>(Color(0.65, 0.05, 0.05));
    auto white = mk<Lambertian>(Color(0.73, 0.73, 0.73));
    auto green = mk<Lambertian>(Color(0.12, 0.45, 0.15));
    auto light = mk<DiffuseLight>(Color(7.0, 7.0, 7.0));
    world->add(mk<Quad>(Point3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), green));
    world->add(mk<Quad>(Point3(0.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), Vec3(0.0, 0.0, 555.0), red));
    world->add(mk<Quad>(Point3(113.0, 554.0, 127.0), Vec3(333.0, 0.0, 0.0), Vec3(0.0, 0.0, 305.0), light));
    world->add(mk<Quad>(Point3(0.0, 555.0, 0.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 0.0, 555.0), white));
    world->add(mk<Quad>(Point3(0.0, 0.0, 0.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 0.0, 555.0), white));
    world->add(mk<Quad>(Point3(0.0, 0.0, 555.0), Vec3(555.0, 0.0, 0.0), Vec3(0.0, 555.0, 0.0), white));
    HittablePtr box1 = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 330.0, 165.0), white);
    box1 = mk<Translate>(mk<RotateY>(box1, 15.0), Vec3(265.0, 0.0, 295.0));
    HittablePtr box2 = box_new(Point3(0.0, 0.0, 0.0), Point3(165.0, 165.0, 165.0), white);
    box2 = mk<Translate>(mk<RotateY>(box2, -18.0), Vec3(130.0, 0.0, 65.0));
    world->add(mk<ConstantMedium>(box1, 0.01, Color(0.0, 0.0, 0.0)));
    world->add(mk<ConstantMedium>(box2, 0.01, Color(1.0, 1.0, 1.0)));
    auto world_bvh = mk<HittableList>(mk<BvhNode>(*world));
    auto lights = mk<HittableList>();
    lights->add(mk<Quad>(Point3(113.0, 554.0, 127.0), Vec3(333.0, 0.0, 0.0), Vec3(0.0, 0.0, 305.0), white));
    return {world_bvh, lights, cornell_camera(width, spp, depth)};
}

// test_obj_model-style scene (main.rs:790-862): an OBJ mesh (BVH) or ObjModel (linear scan) over a ground sphere.
inline SceneTuple obj_scene(const std::string& path, bool as_model) {
    auto world = mk<HittableList>();
    world->add(mk<Sphere>(Point3(0.0, -1000.0, 0.0), 1000.0, mk<Lambertian>(Color(0.5, 0.5, 0.5))));
    auto mat = mk<Lambertian>(Color(0.7, 0.3, 0.2));
    if (as_model) world->add(obj_model_load(path, mat, 1.0, Vec3(0.0, 0.0, 0.0)));
    else world->add(mesh_from_obj(path, mat));
    Camera cam = sky_camera(30.0, Point3(3.0, 2.0, 5.0), Point3(0.0, 0.7, 0.0));
    return {world, nullptr, cam};
}

} // namespace rt_host
#endif
