// obj.cpp -- Wavefront OBJ ingestion of the C ABI: rt_obj_load / rt_obj_free.
//
// Replaces `tobj::load_obj(path, {triangulate: true, single_index: true})` plus the vertex loops of
// Mesh::from_obj (mesh.rs:101-158) and ObjModel::load (obj_loader.rs:21-83).  What reaches the hot path from an
// OBJ file is a list of triangles whose positions went through f32 (tobj stores f32; mesh.rs:137-153 and
// obj_loader.rs:51-67 widen to f64, the latter then applies p * scale + offset per component in f64).
// This reader keeps exactly that: `v` lines parsed with one correctly rounded decimal -> f32 conversion,
// faces of any size fan-triangulated (v0, v_i, v_i+1), 1-based and negative (relative) indices, the
// v / v/vt / v//vn / v/vt/vn forms, models (o / g) concatenated in file order.  tobj itself is not vendored with
// the reference and no model file ships with it, so parity with tobj on real assets is unpinned.
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi355rt.h"

namespace rt { int set_last_error(int code, const std::string& message); }

extern "C" int rt_obj_load(const char* path, const double* scale_offset, int32_t mat, rt_triangle** triangles, uint64_t* n_triangles) {
    if (!path || !triangles || !n_triangles) return rt::set_last_error(RT_E_INVALID, "rt_obj_load: null argument");
    *triangles = nullptr;
    *n_triangles = 0;
    FILE* fp = std::fopen(path, "rb");
    if (!fp) return rt::set_last_error(RT_E_INVALID, std::string("rt_obj_load: cannot open ") + path + ": " + std::strerror(errno));
    std::vector<float> pos;        // x y z per `v` line, f32 as tobj keeps them
    std::vector<uint32_t> corners; // three vertex indices per triangle
    std::string line;
    std::vector<char> buf(1 << 16);
    std::vector<long long> face;
    bool bad = false;
    std::string why;
    auto handle = [&](char* s) {
        while (*s == ' ' || *s == '\t') ++s;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char* p = s + 1;
            float v[3];
            for (int k = 0; k < 3; ++k) {
                char* end = nullptr;
                v[k] = std::strtof(p, &end);
                if (end == p) { bad = true; why = "malformed vertex line"; return; }
                p = end;
            }
            pos.insert(pos.end(), v, v + 3);
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            face.clear();
            char* p = s + 1;
            for (;;) {
                while (*p == ' ' || *p == '\t') ++p;
                if (*p == '\0' || *p == '\r' || *p == '\n' || *p == '#') break;
                char* end = nullptr;
                const long long i = std::strtoll(p, &end, 10);
                if (end == p) { bad = true; why = "malformed face line"; return; }
                face.push_back(i);
                p = end;
                while (*p != '\0' && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') ++p; // skip /vt/vn
            }
            const long long nv = (long long)(pos.size() / 3);
            for (long long& i : face) {
                i = i > 0 ? i - 1 : nv + i; // 1-based, or relative to the vertices read so far
                if (i < 0 || i >= nv) { bad = true; why = "face index out of range"; return; }
            }
            for (size_t k = 1; k + 1 < face.size(); ++k) {
                corners.push_back((uint32_t)face[0]);
                corners.push_back((uint32_t)face[k]);
                corners.push_back((uint32_t)face[k + 1]);
            }
        }
    };
    // lines of any length: fgets pieces are joined until the newline
    while (std::fgets(buf.data(), (int)buf.size(), fp)) {
        line += buf.data();
        if (!line.empty() && line.back() != '\n' && !std::feof(fp)) continue;
        handle(&line[0]);
        line.clear();
        if (bad) break;
    }
    std::fclose(fp);
    if (bad) return rt::set_last_error(RT_E_INVALID, std::string("rt_obj_load: ") + path + ": " + why);
    const uint64_t n = corners.size() / 3;
    if (n >= (1ull << 28)) return rt::set_last_error(RT_E_TOO_LARGE, "rt_obj_load: more than 2^28 triangles");
    rt_triangle* out = static_cast<rt_triangle*>(std::malloc(sizeof(rt_triangle) * (n ? n : 1)));
    if (!out) return rt::set_last_error(RT_E_INVALID, "rt_obj_load: out of host memory");
    for (uint64_t t = 0; t < n; ++t) {
        double* dst[3] = {out[t].v0, out[t].v1, out[t].v2};
        for (int c = 0; c < 3; ++c) {
            const float* p = &pos[(size_t)corners[3 * t + c] * 3];
            for (int k = 0; k < 3; ++k) {
                const double w = (double)p[k];                                    // mesh.rs:137-153
                dst[c][k] = scale_offset ? w * scale_offset[0] + scale_offset[1 + k] : w; // obj_loader.rs:51-67
            }
        }
        out[t].mat = mat;
        out[t]._pad = 0;
    }
    *triangles = out;
    *n_triangles = n;
    return RT_OK;
}

extern "C" void rt_obj_free(rt_triangle* triangles) { std::free(triangles); }
