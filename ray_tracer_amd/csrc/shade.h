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

// Direction sampling of the mixture PDF (pdf.rs:119-125).  Two of the samplers have the same shape: the cosine
// lobe about the normal (pdf.rs:57-59) and the cone towards a sphere light (sphere.rs:57-67, 133-141) both build
// an orthonormal basis about an axis, draw two numbers, and place a point at azimuth 2*pi*r1.  Lanes of either kind
// run that part together (`SAMPLE_LOBE`), each with its own axis and its own two radial expressions; a quad light
// (quad.rs:133-136) or a list member without a sampler (hittable.rs:54-56) yields its direction at once.
// All three take exactly two draws, r1 then r2 (quad.rs:133-136, sphere.rs:57-67, the cosine direction of pdf.rs:57-59), so
// the lanes of a wave make them together whichever sampler each of them needs, and only the few operations that turn
// (r1, r2) into a direction differ.
enum { SAMPLE_DONE = 0, SAMPLE_LOBE = 1, SAMPLE_QUAD = 2 };
struct LobeSample {
    D3 axis;       // the normal, or the direction to the sphere's centre
    double cone;   // sphere light: sqrt(1 - r^2 / distance^2) - 1; unused for the cosine lobe
    bool sphere;
    uint32_t quad; // SAMPLE_QUAD: the quad light to sample
};

RT_DEV int light_pick_one(const DScene& S, DLight l, D3 origin, Rng& g, D3& dir, LobeSample& lobe) { // `l` is wave-uniform
    if (l.kind == RT_REF_QUAD) {
        lobe.quad = l.index;
        return SAMPLE_QUAD;
    }
    if (l.kind == RT_REF_SPHERE) {
        const DSphere s = uniform_load(S.spheres + l.index);
        D3 center = d3(s.c[0], s.c[1], s.c[2]) + 0.0 * d3(s.dc[0], s.dc[1], s.dc[2]);
        D3 direction = center - origin;
        double distance_squared = length_squared(direction);
        lobe.axis = direction;
        lobe.cone = sqrt(1.0 - s.radius * s.radius / distance_squared) - 1.0;
        lobe.sphere = true;
        return SAMPLE_LOBE;
    }
    dir = d3(1.0, 0.0, 0.0); // hittable.rs:54-56
    return SAMPLE_DONE;
}

RT_DEV int lights_pick(const DScene& S, D3 origin, Rng& g, D3& dir, LobeSample& lobe) {
    if (!S.lights_is_list) return light_pick_one(S, uniform_load(S.lights), origin, g, dir, lobe);
    const int32_t index = g.random_int(0, S.n_lights - 1); // hittable_list.rs:78-81
    int mode = SAMPLE_DONE;
    dir = d3(1.0, 0.0, 0.0);
    // every lane picked its own light; the lights are visited one after the other (what the hardware does with a
    // divergent branch anyway), each as a wave-uniform record
    for (int32_t i = 0; i < S.n_lights; ++i) {
        if (__builtin_amdgcn_ballot_w64(index == i) == 0ull) continue;
        const DLight l = uniform_load(S.lights + i);
        if (index == i) mode = light_pick_one(S, l, origin, g, dir, lobe);
    }
    return mode;
}

// Quad::random (quad.rs:133-136): q + r1 * u + r2 * v - origin.  Lanes may have picked different quads: each distinct
// one is read once, as a wave-uniform record.
RT_DEV D3 sample_quad(const DScene& S, uint32_t quad, bool mine, D3 origin, double r1, double r2) {
    D3 dir = d3(1.0, 0.0, 0.0);
    unsigned long long todo = __builtin_amdgcn_ballot_w64(mine);
    while (todo) {
        const uint32_t qi = (uint32_t)__builtin_amdgcn_readlane((int)quad, __builtin_ctzll(todo));
        const DQuad q = uniform_load(S.quads + qi);
        const bool here = mine && quad == qi;
        if (here) {
            D3 a = r1 * d3p(q.u);
            D3 b = r2 * d3p(q.v);
            D3 p = d3p(q.q) + a + b;
            dir = p - origin;
        }
        todo &= ~__builtin_amdgcn_ballot_w64(here);
    }
    return dir;
}

RT_DEV D3 sample_lobe(const LobeSample& lobe, double r1, double r2) {
    Onb uvw(lobe.axis);
    double sphi, cphi;
    sincos_2pi(r1, sphi, cphi);
    // sphere.rs:62-66: z = 1 + r2*(sqrt(1 - r^2/d^2) - 1), x = cos(phi)*sqrt(1 - z*z); pdf.rs / vec3 random_cosine_direction:
    // z = sqrt(1 - r2), x = cos(phi)*sqrt(r2)
    double z = lobe.sphere ? 1.0 + r2 * lobe.cone : sqrt(1.0 - r2);
    double radial = lobe.sphere ? sqrt(1.0 - z * z) : sqrt(r2);
    double x = cphi * radial;
    double y = sphi * radial;
    return uvw.transform(d3(x, y, z));
}

// ----------------------------------------------------------------- one bounce
// Returns true when the path continues from the hit point h.p in direction next_d (every scattering material
// starts the scattered ray at rec.p: material.rs:100, 153, 214, 291); beta/L are updated in place.
// RR: the termination rule of the reference's history (camera.rs:407-416); a template parameter because the rule's
// code in the kernel costs the textured shading kernel a wave per SIMD (264 instead of 243 registers).
template <bool TEXTURES, bool RR>
RT_DEV bool shade_hit(const DScene& S, const HitRec& h, D3 ray_dir, Rng& g, D3& beta, D3& L, D3& next_d) {
    RT_SEC_BEGIN(4);
    const DMaterial m = S.materials[h.mat];
    RT_SEC_END(4);
    if (m.kind == RT_MAT_DIFFUSE_LIGHT) { // material.rs:237-256: emits on the front face, never scatters
        // The back face emits black, and black still goes through the product: a path whose throughput is NaN
        // or infinite (a light-pdf of 0/0) makes the sample NaN in the recursion, so it must here too.
        L = L + beta * (h.front_face ? material_color<TEXTURES>(S, m, h.u, h.v, h.p) : d3(0.0, 0.0, 0.0));
        return false;
    }
    if (m.kind == RT_MAT_METAL) { // material.rs:135-158; skip_pdf branch camera.rs:276-295
        RT_SEC_BEGIN(5);
        D3 reflected = reflect(ray_dir, h.normal);
        reflected = unit_vector(reflected) + m.fuzz * random_unit_vector(g);
        beta = beta * d3(m.albedo[0], m.albedo[1], m.albedo[2]);
        next_d = reflected;
        RT_SEC_END(5);
        return true;
    }
    if (m.kind == RT_MAT_DIELECTRIC) { // material.rs:181-217
        RT_SEC_BEGIN(6);
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
        RT_SEC_END(6);
        return true;
    }
    // Lambertian (material.rs:77-117) / Isotropic (material.rs:278-300): sampled through a PDF
    const bool lambert = m.kind == RT_MAT_LAMBERTIAN;
    RT_SEC_BEGIN(7);
    D3 attenuation = material_color<TEXTURES>(S, m, h.u, h.v, h.p);
    RT_SEC_END(7);
    RT_SEC_BEGIN(8);
    const D3 unit_normal = unit_vector(h.normal); // the w axis of CosinePdf's basis (pdf.rs:46-48, onb.rs:13); unused for Isotropic
    D3 dir = d3(1.0, 0.0, 0.0);
    bool from_light = false;
    if (S.n_lights >= 0) from_light = g.next() < 0.5; // MixturePdf::generate (pdf.rs:119-125)
    int mode = SAMPLE_DONE;
    LobeSample lobe;
    lobe.axis = h.normal; lobe.cone = 0.0; lobe.sphere = false; lobe.quad = 0u;
    if (from_light) mode = lights_pick(S, h.p, g, dir, lobe);
    else if (lambert) mode = SAMPLE_LOBE;                              // pdf.rs:57-59
    else dir = random_unit_vector(g);                                  // pdf.rs:36-38
    RT_SEC_END(8);
    RT_SEC_BEGIN(9);
    if (mode != SAMPLE_DONE) {
        const double r1 = g.next();
        const double r2 = g.next();
        const D3 towards_quad = sample_quad(S, lobe.quad, mode == SAMPLE_QUAD, h.p, r1, r2);
        dir = mode == SAMPLE_QUAD ? towards_quad : sample_lobe(lobe, r1, r2);
    }
    RT_SEC_END(9);
    RT_SEC_BEGIN(10);
    double mat_pdf;
    if (lambert) { // CosinePdf::value (pdf.rs:52-55)
        double cosine_theta = dot(unit_vector(dir), unit_normal);
        mat_pdf = fmax(cosine_theta / D_PI, 0.0);
    } else {
        mat_pdf = 1.0 / (4.0 * D_PI);
    }
    double pdf_value = mat_pdf;
    if (S.n_lights >= 0) pdf_value = 0.5 * lights_pdf_value(S, h.p, dir) + 0.5 * mat_pdf; // pdf.rs:115-117
    RT_SEC_END(10);
    double scattering_pdf;
    if (lambert) { // material.rs:110-117
        double cos_theta = dot(h.normal, unit_vector(dir));
        scattering_pdf = cos_theta < 0.0 ? 0.0 : cos_theta / D_PI;
    } else {
        scattering_pdf = 1.0 / (4.0 * D_PI);
    }
    if (RR) { // no re-weighting of survivors
        double p = fmax(fmax(attenuation.x, attenuation.y), attenuation.z);
        p = p < 0.1 ? 0.1 : (p > 1.0 ? 1.0 : p);
        if (g.next() >= p) {
            // the recursion returns this vertex's emission (black for these materials) and its callers still multiply
            // it by their factors: a NaN or infinite throughput must turn the sample into NaN here as it does there
            L = L + beta * d3(0.0, 0.0, 0.0);
            return false;
        }
    }
    beta = (beta * (attenuation * scattering_pdf)) / pdf_value; // camera.rs:326-336
    next_d = dir;
    return true;
}

} // namespace rt
#endif
