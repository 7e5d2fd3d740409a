// shade.h -- what one bounce of Camera::ray_color (camera.rs:243-339) does after
// world.hit returned: emission, scatter, mixture-PDF sampling, throughput update.
// Iterative restatement of the recursion: a path carries its throughput `beta` and
// its accumulated radiance `L`;
//   color = emission + (attenuation * scattering_pdf * ray_color(scattered)) / pdf_value
// becomes  L += beta * emission;  beta *= attenuation * scattering_pdf / pdf_value.
#ifndef RT_SHADE_H
#define RT_SHADE_H

#include "traverse.h"

namespace rt {

// ------------------------------------------------------------------ textures
RT_DEV double perlin_noise(const rt_perlin* t, D3 p) { // perlin.rs:95-129 + 75-93
    double u = p.x - floor(p.x);
    double v = p.y - floor(p.y);
    double w = p.z - floor(p.z);
    int32_t i = sat_i32(floor(p.x)), j = sat_i32(floor(p.y)), k = sat_i32(floor(p.z));
    double uu = u * u * (3.0 - 2.0 * u); // fade (perlin.rs:56-58)
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t xi = ((uint32_t)i + (uint32_t)di) & 255u;
                uint32_t yi = ((uint32_t)j + (uint32_t)dj) & 255u;
                uint32_t zi = ((uint32_t)k + (uint32_t)dk) & 255u;
                int32_t index = t->perm_x[xi] ^ t->perm_y[yi] ^ t->perm_z[zi];
                D3 c = d3(t->randvec[index][0], t->randvec[index][1], t->randvec[index][2]);
                D3 weight_v = d3(u - (double)di, v - (double)dj, w - (double)dk);
                double i_factor = (double)di * uu + (1.0 - (double)di) * (1.0 - uu);
                double j_factor = (double)dj * vv + (1.0 - (double)dj) * (1.0 - vv);
                double k_factor = (double)dk * ww + (1.0 - (double)dk) * (1.0 - ww);
                accum += i_factor * j_factor * k_factor * dot(c, weight_v);
            }
    return accum;
}

RT_DEV double perlin_turb(const rt_perlin* t, D3 p, int depth) { // perlin.rs:131-143
    double accum = 0.0;
    D3 temp_p = p;
    double weight = 1.0;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(t, temp_p);
        weight *= 0.5;
        temp_p = temp_p * 2.0;
    }
    return fabs(accum);
}

RT_DEV D3 texture_value(const DScene& S, int32_t tex, double u, double v, D3 p) { // texture.rs:6-130
    // A checker selects one of two textures created before it, so the walk terminates.
    for (;;) {
        const DTexture* t = S.textures + tex;
        if (t->kind == RT_TEX_SOLID) return d3(t->color[0], t->color[1], t->color[2]);
        if (t->kind == RT_TEX_CHECKER) { // texture.rs:59-71; t->scale holds inv_scale
            int32_t xi = sat_i32(floor(t->scale * p.x));
            int32_t yi = sat_i32(floor(t->scale * p.y));
            int32_t zi = sat_i32(floor(t->scale * p.z));
            bool is_even = ((int64_t)xi + yi + zi) % 2 == 0;
            tex = is_even ? t->even : t->odd;
            continue;
        }
        if (t->kind == RT_TEX_IMAGE) { // texture.rs:88-107, rtw_stb_image.rs:155-171
            if (t->image < 0) return d3(0.0, 1.0, 1.0);
            const rt_image im = S.images[t->image];
            if (im.height == 0) return d3(0.0, 1.0, 1.0);
            double uc = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
            double vc = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            vc = 1.0 - vc;
            uint32_t i = sat_u32(uc * (double)im.width);
            uint32_t j = sat_u32(vc * (double)im.height);
            uint32_t x = i < (uint32_t)im.width - 1u ? i : (uint32_t)im.width - 1u;
            uint32_t y = j < (uint32_t)im.height - 1u ? j : (uint32_t)im.height - 1u;
            const uint8_t* px = S.image_bytes + im.offset + (size_t)y * (size_t)im.width * 3 + (size_t)x * 3;
            double color_scale = 1.0 / 255.0;
            return d3(color_scale * (double)px[0], color_scale * (double)px[1], color_scale * (double)px[2]);
        }
        // RT_TEX_NOISE (texture.rs:126-129)
        double noise_value = perlin_turb(S.perlins + t->perlin, p, 7);
        return d3(0.5, 0.5, 0.5) * (1.0 + sin(t->scale * p.z + 10.0 * noise_value));
    }
}

// TEXTURES = false: every texture of the scene is a solid colour (they all ride in the material records), and the
// checker / image / noise code is compiled out.  That code alone costs the shading kernel about 100 vector
// registers, i.e. one wave per SIMD.
template <bool TEXTURES>
RT_DEV D3 material_color(const DScene& S, const DMaterial& m, double u, double v, D3 p) {
    if (!TEXTURES || m.tex < 0) return d3(m.albedo[0], m.albedo[1], m.albedo[2]);
    return texture_value(S, m.tex, u, v, p);
}

// -------------------------------------------------------------------- lights
// The light list is the same for every lane, so its entries and the primitives they name are read through
// the scalar cache (uniform_load): a vector load of the same address by all lanes queues behind the path
// records the compute unit is streaming from HBM.
// pdf_value of one light for the ray (origin, direction, time 0) (quad.rs:117-131, sphere.rs:117-131).
RT_DEV double light_pdf_one(const DScene& S, DLight l, D3 origin, D3 direction) { // `l` is wave-uniform
    RayD r;
    r.ox = origin.x; r.oy = origin.y; r.oz = origin.z;
    r.dx = direction.x; r.dy = direction.y; r.dz = direction.z;
    if (l.kind == RT_REF_QUAD) {
        const DQuad q = uniform_load(S.quads + l.index);
        double t, al, be;
        if (!quad_test(q, r, 0.001, D_INF, t, al, be)) return 0.0;
        double distance_squared = t * t * length_squared(direction);
        double cosine = fabs(dot(direction, d3p(q.n)) / length(direction)); // |.| makes the face flip irrelevant
        return distance_squared / (cosine * q.area);
    }
    if (l.kind == RT_REF_SPHERE) {
        const DSphere s = uniform_load(S.spheres + l.index);
        double t;
        if (!sphere_test(s, r, 0.0, 0.001, D_INF, t)) return 0.0;
        D3 center = d3(s.c[0], s.c[1], s.c[2]) + 0.0 * d3(s.dc[0], s.dc[1], s.dc[2]);
        double dist_squared = length_squared(center - origin);
        double cos_theta_max = sqrt(1.0 - s.radius * s.radius / dist_squared);
        double solid_angle = 2.0 * D_PI * (1.0 - cos_theta_max);
        return 1.0 / solid_angle;
    }
    return 0.0; // hittable.rs:51-53
}

RT_DEV double lights_pdf_value(const DScene& S, D3 origin, D3 direction) {
    if (!S.lights_is_list) return light_pdf_one(S, uniform_load(S.lights), origin, direction);
    double weight = 1.0 / (double)S.n_lights; // hittable_list.rs:67-76
    double sum = 0.0;
    for (int32_t i = 0; i < S.n_lights; ++i) sum += weight * light_pdf_one(S, uniform_load(S.lights + i), origin, direction);
    return sum;
}

// random() of one light (quad.rs:133-136; sphere.rs:133-141, 57-67): two draws either way.
RT_DEV D3 light_random_one(const DScene& S, DLight l, D3 origin, Rng& g) { // `l` is wave-uniform
    if (l.kind == RT_REF_QUAD) {
        const DQuad q = uniform_load(S.quads + l.index);
        double r1 = g.next();
        D3 a = r1 * d3p(q.u);
        double r2 = g.next();
        D3 b = r2 * d3p(q.v);
        D3 p = d3p(q.q) + a + b;
        return p - origin;
    }
    if (l.kind == RT_REF_SPHERE) {
        const DSphere s = uniform_load(S.spheres + l.index);
        D3 center = d3(s.c[0], s.c[1], s.c[2]) + 0.0 * d3(s.dc[0], s.dc[1], s.dc[2]);
        D3 direction = center - origin;
        double distance_squared = length_squared(direction);
        Onb uvw(direction);
        double r1 = g.next();
        double r2 = g.next();
        double z = 1.0 + r2 * (sqrt(1.0 - s.radius * s.radius / distance_squared) - 1.0);
        double sphi, cphi;
        sincos_2pi(r1, sphi, cphi);
        double x = cphi * sqrt(1.0 - z * z);
        double y = sphi * sqrt(1.0 - z * z);
        return uvw.transform(d3(x, y, z));
    }
    return d3(1.0, 0.0, 0.0); // hittable.rs:54-56
}

RT_DEV D3 lights_random(const DScene& S, D3 origin, Rng& g) {
    if (!S.lights_is_list) return light_random_one(S, uniform_load(S.lights), origin, g);
    const int32_t index = g.random_int(0, S.n_lights - 1); // hittable_list.rs:78-81
    D3 out = d3(1.0, 0.0, 0.0);
    // every lane picked its own light; the lights are visited one after the other (what the hardware does with a
    // divergent branch anyway), each as a wave-uniform record
    for (int32_t i = 0; i < S.n_lights; ++i) {
        if (__builtin_amdgcn_ballot_w64(index == i) == 0ull) continue;
        const DLight l = uniform_load(S.lights + i);
        if (index == i) out = light_random_one(S, l, origin, g);
    }
    return out;
}

// ----------------------------------------------------------------- one bounce
// Returns true when the path continues from the hit point h.p in direction next_d (every scattering material
// starts the scattered ray at rec.p: material.rs:100, 153, 214, 291); beta/L are updated in place.
template <bool TEXTURES>
RT_DEV bool shade_hit(const DScene& S, const HitRec& h, D3 ray_dir, Rng& g, D3& beta, D3& L, D3& next_d) {
    const DMaterial m = S.materials[h.mat];
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) { // material.rs:237-256: emits on the front face, never scatters
        if (h.front_face) L = L + beta * material_color<TEXTURES>(S, m, h.u, h.v, h.p);
        return false;
    }
    if (m.kind == RT_MAT_METAL) { // material.rs:135-158; skip_pdf branch camera.rs:276-295
        D3 reflected = reflect(ray_dir, h.normal);
        reflected = unit_vector(reflected) + m.fuzz * random_unit_vector(g);
        beta = beta * d3(m.albedo[0], m.albedo[1], m.albedo[2]);
        next_d = reflected;
        return true;
    }
    if (m.kind == RT_MAT_DIELECTRIC) { // material.rs:181-217
        double ri = h.front_face ? 1.0 / m.ior : m.ior;
        D3 unit_direction = unit_vector(ray_dir);
        double cos_theta = fmin(-dot(unit_direction, h.normal), 1.0);
        double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ri * sin_theta > 1.0;
        bool do_reflect = cannot_refract;
        if (!do_reflect) { // Schlick (material.rs:175-178); the draw is skipped under total internal reflection
            double r0 = (1.0 - ri) / (1.0 + ri);
            r0 = r0 * r0;
            double mm = 1.0 - cos_theta;
            double m2 = mm * mm, m4 = m2 * m2;
            double reflectance = r0 + (1.0 - r0) * (mm * m4);
            do_reflect = reflectance > g.next();
        }
        D3 direction = do_reflect ? reflect(unit_direction, h.normal) : refract(unit_direction, h.normal, ri);
        beta = beta * d3(1.0, 1.0, 1.0);
        next_d = direction;
        return true;
    }
    // Lambertian (material.rs:77-117) / Isotropic (material.rs:278-300): sampled through a PDF
    const bool lambert = m.kind == RT_MAT_LAMBERTIAN;
    D3 attenuation = material_color<TEXTURES>(S, m, h.u, h.v, h.p);
    Onb uvw(h.normal); // CosinePdf::new (pdf.rs:46-48); unused for Isotropic
    D3 dir;
    bool from_light = false;
    if (S.n_lights >= 0) from_light = g.next() < 0.5; // MixturePdf::generate (pdf.rs:119-125)
    if (from_light) dir = lights_random(S, h.p, g);
    else if (lambert) dir = uvw.transform(random_cosine_direction(g)); // pdf.rs:57-59
    else dir = random_unit_vector(g);                                  // pdf.rs:36-38
    double mat_pdf;
    if (lambert) { // CosinePdf::value (pdf.rs:52-55)
        double cosine_theta = dot(unit_vector(dir), uvw.a2);
        mat_pdf = fmax(cosine_theta / D_PI, 0.0);
    } else {
        mat_pdf = 1.0 / (4.0 * D_PI);
    }
    double pdf_value = mat_pdf;
    if (S.n_lights >= 0) pdf_value = 0.5 * lights_pdf_value(S, h.p, dir) + 0.5 * mat_pdf; // pdf.rs:115-117
    double scattering_pdf;
    if (lambert) { // material.rs:110-117
        double cos_theta = dot(h.normal, unit_vector(dir));
        scattering_pdf = cos_theta < 0.0 ? 0.0 : cos_theta / D_PI;
    } else {
        scattering_pdf = 1.0 / (4.0 * D_PI);
    }
    beta = (beta * (attenuation * scattering_pdf)) / pdf_value; // camera.rs:326-336
    next_d = dir;
    return true;
}

} // namespace rt
#endif
