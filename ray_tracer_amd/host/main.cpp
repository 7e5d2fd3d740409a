// rt_cli -- the reference's `main` (main.rs:898-912: pick a scene, Camera::render, PPM out) on top of
// libmi355rt.so.  The reference selects the scene by editing a match arm and prints to stdout; here the
// scene, image size and seed are arguments and the PPM goes to --out (or stdout with --out -).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "scenes.hpp"

using namespace rt_host;

static void usage() {
    std::fprintf(stderr,
                 "usage: rt_cli --scene book1_final|quads|simple_light|cornell_box|cornell_smoke\n"
                 "                     |obj_mesh|obj_model (with --obj file.obj)\n"
                 "              [--width W] [--spp N] [--depth D] [--seed S] [--out file.ppm|-] [--describe]\n");
}

int main(int argc, char** argv) {
    std::string scene = "cornell_box", out, obj;
    int width = 0, spp = 0, depth = 0;
    uint64_t seed_ = 0;
    bool describe = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next();
        else if (a == "--width") width = std::atoi(next());
        else if (a == "--spp") spp = std::atoi(next());
        else if (a == "--depth") depth = std::atoi(next());
        else if (a == "--seed") seed_ = std::strtoull(next(), nullptr, 10);
        else if (a == "--out") out = next();
        else if (a == "--obj") obj = next();
        else if (a == "--describe") describe = true;
        else { usage(); return 2; }
    }
    try {
        SceneTuple s;
        if (scene == "book1_final") s = book1_final(seed_);
        else if (scene == "quads") s = quads();
        else if (scene == "simple_light") s = simple_light(seed_);
        else if (scene == "cornell_box") s = cornell_box();
        else if (scene == "cornell_smoke") s = cornell_smoke();
        else if ((scene == "obj_mesh" || scene == "obj_model") && !obj.empty()) s = obj_scene(obj, scene == "obj_model");
        else { usage(); return 2; }
        if (width > 0) s.cam.image_width = width;
        if (spp > 0) s.cam.samples_per_pixel = spp;
        if (depth > 0) s.cam.max_depth = depth;
        if (describe) { // host-only: flatten + validate, print the scene summary; touches no device
            SceneDesc d(s.world, s.lights);
            rt_scene* h = nullptr;
            check(rt_scene_create(&d.d, RT_SCENE_HOST_ONLY, &h));
            rt_scene_info info;
            check(rt_scene_get_info(h, &info));
            std::printf("{\"scene\": \"%s\", \"objects\": %zu, \"spheres\": %zu, \"quads\": %zu, \"triangles\": %zu, "
                        "\"nodes\": %llu, \"lists\": %llu, \"xforms\": %llu, \"media\": %llu, \"lights\": %llu, \"max_stack\": %u}\n",
                        scene.c_str(), d.objects.size(), d.spheres.size(), d.quads.size(), d.triangles.size(),
                        (unsigned long long)info.n_nodes, (unsigned long long)info.n_lists, (unsigned long long)info.n_xforms,
                        (unsigned long long)info.n_media, (unsigned long long)info.n_lights, info.max_stack);
            rt_scene_destroy(h);
            return 0;
        }
        auto t0 = std::chrono::steady_clock::now();
        std::string path = out.empty() ? scene + ".ppm" : (out == "-" ? "/dev/stdout" : out);
        s.cam.render(s.world, s.lights, seed_, path.c_str());
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "%s: %d px wide, %d spp, %.3f s (scene build + render + PPM)\n", scene.c_str(), s.cam.image_width,
                     s.cam.samples_per_pixel, sec);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rt_cli: %s\n", e.what());
        return 1;
    }
    return 0;
}
