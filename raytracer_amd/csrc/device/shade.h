// Device versions of Texture::value (src/texture.rs), Perlin (src/perlin.rs),
// Material::scatter / emitted (src/material.rs), the sampling helpers of
// src/vec3.rs:40-75 and Camera::get_ray (src/camera.rs:61-71).
#pragma once
#include "hitrecord.h"

namespace rtd {

// ---- sampling helpers (src/vec3.rs:40-75) on one RNG stream -----------------------
RT_D V3 random_in_unit_sphere(rt::Rng& rng) {  // :40-48
    for (;;) {
        const double x = rng.uniform(-1.0, 1.0);
        const double y = rng.uniform(-1.0, 1.0);
        const double z = rng.uniform(-1.0, 1.0);
        const V3 p = v3(x, y, z);
        if (squared_length(p) >= 1.0) continue;
        return p;
    }
}
RT_D V3 random_unit_vector(rt::Rng& rng) {  // :50-55
    const double a = rng.uniform(0.0, 2.0 * D_PI);
    const double z = rng.uniform(-1.0, 1.0);
    const double r = sqrt(1.0 - z * z);
    double sa, ca;
    sincos(a, &sa, &ca);
    return v3(r * ca, r * sa, z);
}
RT_D V3 random_in_unit_disk(rt::Rng& rng) {  // :67-75
    for (;;) {
        const double x = rng.uniform(-1.0, 1.0);
        const double y = rng.uniform(-1.0, 1.0);
        const V3 p = v3(x, y, 0.0);
        if (squared_length(p) >= 1.0) continue;
        return p;
    }
}

// ---- Perlin (src/perlin.rs:44-103) -------------------------------------------------
RT_D int f64_to_i32_sat(double f) {  // Rust `as i32`: saturating, NaN -> 0
    if (f != f) return 0;
    if (f <= -2147483648.0) return (int)0x80000000;
    if (f >= 2147483647.0) return 0x7FFFFFFF;
    return (int)f;
}

template <class PP>
RT_D double perlin_noise(PP P, V3 p) {  // noise :66-89 + perlin_interp :44-63
    const double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    const double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const uint32_t i = (uint32_t)f64_to_i32_sat(fx);
    const uint32_t j = (uint32_t)f64_to_i32_sat(fy);
    const uint32_t k = (uint32_t)f64_to_i32_sat(fz);
    const double uu = u * u * (3.0 - 2.0 * u);
    const double vv = v * v * (3.0 - 2.0 * v);
    const double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
#pragma unroll
    for (uint32_t di = 0; di < 2; ++di)
#pragma unroll
        for (uint32_t dj = 0; dj < 2; ++dj)
#pragma unroll
            for (uint32_t dk = 0; dk < 2; ++dk) {
                const int g = P->perm_x[(i + di) & 255u] ^ P->perm_y[(j + dj) & 255u] ^ P->perm_z[(k + dk) & 255u];
                const V3 c = v3p(P->ran_vec[g]);
                const V3 weight_v = v3(u - (double)di, v - (double)dj, w - (double)dk);
                accum += ((double)di * uu + (1.0 - (double)di) * (1.0 - uu)) *
                         ((double)dj * vv + (1.0 - (double)dj) * (1.0 - vv)) *
                         ((double)dk * ww + (1.0 - (double)dk) * (1.0 - ww)) * dot(c, weight_v);
            }
    return accum;
}

template <class PP>
RT_D double perlin_turb(PP P, V3 p, int depth) {  // :91-103
    double accum = 0.0;
    V3 temp_p = p;
    double weight = 1.0;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(P, temp_p);
        weight *= 0.5;
        temp_p = temp_p * 2.0;
    }
    return fabs(accum);
}

// ---- Texture::value -----------------------------------------------------------------
RT_D double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }  // clamp(x, 0, 1), src/rtweekend.rs:22-30
RT_D uint32_t f64_to_u32_sat(double f) {  // Rust `as u32`
    if (!(f > 0.0)) return 0u;
    if (f >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

template <class SC>
RT_D V3 texture_value(const SC& S, uint32_t tex, double u, double v, V3 p) {
    const auto T = S.textures + tex;
    switch (T->kind) {
        case RT_T_SOLID: return v3p(T->a);  // src/texture.rs:22-26
        case RT_T_CHECKER: {                // :42-51
            const double sines = sin(p.x * 10.0) * sin(10.0 * p.y) * sin(10.0 * p.z);
            return sines < 0.0 ? v3p(T->b) : v3p(T->a);
        }
        case RT_T_NOISE: {                  // :67-73
            const double scale = T->a[0];
            const double s = 0.5 * (1.0 + sin(scale * p.z + 10.0 * perlin_turb(S.perlins + T->aux, p * scale, 7)));
            return v3(s, s, s);  // Color::ones() * 0.5 * (...) = (1*0.5) * (...) per component
        }
        default: {                          // RT_T_IMAGE :92-114
            const RtImage* im = S.images + T->aux;
            const double uc = clamp01(u);
            const double vc = 1.0 - clamp01(v);
            uint32_t i = f64_to_u32_sat(uc * (double)im->width);
            uint32_t j = f64_to_u32_sat(vc * (double)im->height);
            if (i >= im->width) i = im->width - 1u;
            if (j >= im->height) j = im->height - 1u;
            const double color_scale = 1.0 / 255.0;
            const uint8_t* px = S.texels + im->offset + ((uint64_t)j * im->width + i) * 3u;
            return v3(color_scale * (double)px[0], color_scale * (double)px[1], color_scale * (double)px[2]);
        }
    }
}

// ---- Material -----------------------------------------------------------------------
RT_D double schlick(double cosine, double ref_idx) {  // src/material.rs:144-148; powi(5) = x * ((x*x) * (x*x))
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    const double x = 1.0 - cosine;
    const double x2 = x * x;
    return r0 + (1.0 - r0) * (x * (x2 * x2));
}

// Texture::value of the material's texture, evaluated once per hit: Lambertian and Isotropic use it as the
// attenuation (src/material.rs:48, 208), DiffuseLight as the emitted radiance (:176-178).
RT_D bool material_has_texture(uint32_t kind) { return kind == RT_M_LAMBERTIAN || kind == RT_M_DIFFUSE_LIGHT || kind == RT_M_ISOTROPIC; }

// scatter (src/material.rs:35-50, 68-83, 97-141, 167-175, 196-210). `din` = r_in.dire, `texv` = the material's
// texture value at the hit. Returns false when the path ends here. A zero-length incoming direction (the
// reference would panic in Vec3::unit, src/vec3.rs:285-286) also ends the path.
template <class PMAT>
RT_D bool material_scatter(PMAT M, V3 din, const HitRec& rec, V3 texv, rt::Rng& rng, V3& attenuation, V3& dout) {
    switch (M->kind) {
        case RT_M_LAMBERTIAN: {
            dout = rec.normal + random_unit_vector(rng);
            attenuation = texv;
            return true;
        }
        case RT_M_METAL: {
            const double len = length(din);
            if (len == 0.0) return false;
            const V3 unit = v3(din.x / len, din.y / len, din.z / len);
            const V3 reflected = reflect(unit, rec.normal);
            dout = reflected + random_in_unit_sphere(rng) * M->param;
            attenuation = v3p(M->albedo);
            return dot(dout, rec.normal) > 0.0;
        }
        case RT_M_DIELECTRIC: {
            attenuation = v3(1.0, 1.0, 1.0);
            const double ref_idx = M->param;
            const double etai_over_etat = rec.front_face ? 1.0 / ref_idx : ref_idx;
            const double len = length(din);
            if (len == 0.0) return false;
            const V3 unit_direction = v3(din.x / len, din.y / len, din.z / len);
            const double cos_theta = fmin(dot(-unit_direction, rec.normal), 1.0);
            const double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
            if (etai_over_etat * sin_theta > 1.0) {
                dout = reflect(unit_direction, rec.normal);
                return true;
            }
            const double reflect_prob = schlick(cos_theta, etai_over_etat);
            if (rng.uniform(0.0, 1.0) < reflect_prob) {
                dout = reflect(unit_direction, rec.normal);
                return true;
            }
            dout = refract(unit_direction, rec.normal, etai_over_etat);
            return true;
        }
        case RT_M_ISOTROPIC: {
            dout = random_in_unit_sphere(rng);
            attenuation = texv;
            return true;
        }
        default: return false;  // RT_M_DIFFUSE_LIGHT
    }
}

// ---- camera (src/camera.rs:61-71 and the jitter of src/main.rs:206-208) ----------------
RT_D void camera_ray(const RtCamera& cam, uint64_t seed, uint32_t width, uint32_t height, uint32_t x, uint32_t y, uint32_t pixel,
                     uint32_t sample, V3& o, V3& d, double& tm) {
    rt::Rng rng;
    rng.set_path(seed, pixel, sample, 0u, rt::RNG_JITTER);
    const double ju = rng.uniform(0.0, 1.0);
    const double jv = rng.uniform(0.0, 1.0);
    const double s = ((double)x + ju) / (double)(width - 1u);
    const double t = ((double)(height - y) + jv) / (double)(height - 1u);  // (height - y): the reference's row offset
    rng.set_path(seed, pixel, sample, 0u, rt::RNG_LENS);
    const V3 rd = random_in_unit_disk(rng) * cam.lens_radius;
    const V3 offset = v3p(cam.u) * rd.x + v3p(cam.v) * rd.y;
    rng.set_path(seed, pixel, sample, 0u, rt::RNG_TIME);
    tm = rng.uniform(cam.time0, cam.time1);
    const V3 orig = v3p(cam.orig);
    o = orig + offset;
    d = v3p(cam.lower_left_corner) + v3p(cam.horizontal) * s + v3p(cam.vertical) * t - orig - offset;
}

}  // namespace rtd
