// Host-only driver for a ThreadSanitizer run of the flattener and the tree builder (no GPU, no HIP):
//   cd ray_tracer_amd/csrc && g++ -std=c++17 -O1 -g -fsanitize=thread -I. -I../../include flatten.cpp fast_tree.cpp ../../profiles/r3_tsan_driver.cpp -o /tmp/drv -pthread && /tmp/drv 300000
// Result when it was last run (the build that produced profiles/r3_bench.json): two builds of a 300,000-triangle mesh (chunked binning, subtree tasks, the exact-cull
// collapse beside them), no report.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>
#include "flatten.h"
namespace rt { int build_bvh_device(const BuildItem*, uint64_t, uint64_t, DNode*, WNode*, std::string& err) { err = "no device"; return RT_E_HIP; } }
int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 300000;
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> u(0.0, 1.0), e(-0.02, 0.02);
    std::vector<rt_triangle> tris(n);
    for (auto& t : tris) {
        double c[3] = {u(g), u(g), u(g)};
        for (int k = 0; k < 3; ++k) { t.v0[k] = c[k]; t.v1[k] = c[k] + e(g); t.v2[k] = c[k] + e(g); }
        t.mat = 0; t._pad = 0;
    }
    rt_texture tx{}; tx.kind = RT_TEX_SOLID; tx.color[0] = tx.color[1] = tx.color[2] = 0.5;
    rt_material m{}; m.kind = RT_MAT_LAMBERTIAN; m.tex = 0; m.albedo[0] = m.albedo[1] = m.albedo[2] = 0.5;
    rt_object o{}; o.kind = RT_OBJ_BVH; o.child_kind = RT_CHILD_TRIANGLES; o.a = 0; o.b = (int32_t)n; o.mat = -1;
    rt_scene_desc d{};
    d.triangles = tris.data(); d.n_triangles = n; d.materials = &m; d.n_materials = 1; d.textures = &tx; d.n_textures = 1; d.objects = &o; d.n_objects = 1; d.world = 0; d.lights = -1;
    for (int rep = 0; rep < 2; ++rep) {
        rt::FlatScene f; std::string err;
        int rc = rt::flatten_scene(&d, f, err, 0, 1024);
        std::printf("rc %d (%s) qnodes %zu ptris %zu fast_tree %d\n", rc, err.c_str(), f.qnodes.size(), f.ptris.size(), (int)f.fast_tree);
    }
    return 0;
}
