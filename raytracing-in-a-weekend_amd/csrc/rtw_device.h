// rtw_device.h -- device-side math, RNG and shading for the gfx950 render kernels.
//
// Arithmetic contract (DESIGN.md "Arithmetic"): f32, one rounding per written operation, NO fused
// multiply-add (the TU is compiled with -ffp-contract=off), IEEE-correct divide and sqrt (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt).  Under that contract every value on a path
// (hit t, normal, scatter direction, RNG accept/reject, Schlick branch) is bit-identical to the
// Rust reference's scalar f32 code for the same random stream, so the only places a GPU image can
// differ from the CPU one are the three libm calls: powf (gamma), atan2f / acosf (texture UV).
//
// Reference files restated here (same operation order):
//   Rust/src/vec3.rs:188-261               Vec3 ops, random_unit_vec, random_in_unit_disk, reflect
//   Rust/src/objects/sphere.rs:99-147      Sphere::collision_normal
//   Rust/src/objects/materials.rs:89-154   refract, reflectance, Material::on_hit (+ diffuse :213-228)
//   Rust/src/viewport/ray_color.rs:12-92   ray_color_gradient / ray_color_bg_color
//   Rust/src/texture.rs:259-267            ImageTexture::color_at
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rtw.h"

namespace rtw {

typedef float f4 __attribute__((ext_vector_type(4)));
// Wave-uniform, read-only scene data is read through the constant address space so that hipcc emits
// scalar loads (s_load_dwordx4 -> SGPR operands of the VALU ops) instead of 64 identical vector loads.
typedef const f4 __attribute__((address_space(4))) *cf4_ptr;

// Member-wise copy on purpose: an implicit (memcpy-style) aggregate copy makes SROA slice the path
// state into <1 x float> pieces that mem2reg then refuses to promote (24 B of scratch per lane).
struct v3 {
    float x, y, z;
    __device__ __forceinline__ v3() {}
    __device__ __forceinline__ v3(const v3 &o) : x(o.x), y(o.y), z(o.z) {}
    __device__ __forceinline__ v3 &operator=(const v3 &o) { x = o.x; y = o.y; z = o.z; return *this; }
};

__device__ __forceinline__ v3 mk(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ v3 operator+(v3 a, v3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 operator-(v3 a, v3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 operator*(v3 a, v3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ v3 operator*(v3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ v3 operator/(v3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ v3 operator-(v3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
__device__ __forceinline__ float len(v3 a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ v3 unit(v3 a) { return a / len(a); }
// vec3.rs:256  self - (n * 2.0) * self.dot(n)
__device__ __forceinline__ v3 reflect(v3 a, v3 n) { return a - (n * 2.0f) * dot(a, n); }
__device__ __forceinline__ bool close_to_zero(v3 a) {
    return __builtin_fabsf(a.x) < 1e-7f && __builtin_fabsf(a.y) < 1e-7f && __builtin_fabsf(a.z) < 1e-7f;
}

// ---- RNG: 32-bit PCG (RXS-M-XS) with a per-stream odd increment, seeded by a lowbias32 hash chain
// of (seed, pixel, sample).  Same definition as oracle/rtw_oracle.c (independently written there).
struct Rng { uint32_t state, inc; };

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// `base` = the (seed, pixel) prefix of the chain, hoisted out of the sample loop.
__device__ __forceinline__ uint32_t rng_pixel_base(uint32_t seed_lo, uint32_t seed_hi, uint32_t pixel) {
    uint32_t h = mix32(seed_lo + 0x9E3779B9U);
    h = mix32(h ^ seed_hi);
    return mix32(h ^ pixel);
}
__device__ __forceinline__ Rng rng_start(uint32_t base, uint32_t sample) {
    Rng r; r.state = mix32(base ^ sample); r.inc = mix32(r.state ^ 0x85EBCA6BU) | 1U;
    return r;
}
__device__ __forceinline__ float rng_f32(Rng &r) {
    uint32_t old = r.state;
    r.state = old * 747796405U + r.inc;
    uint32_t word = ((old >> ((old >> 28) + 4U)) ^ old) * 277803737U;
    word = (word >> 22) ^ word;
    return (float)(word >> 8) * (1.0f / 16777216.0f);
}
// vec3.rs:228-239
__device__ __forceinline__ v3 random_unit_vec(Rng &r) {
    v3 p;
    for (;;) {
        p.x = rng_f32(r) * 2.0f + -1.0f;   // xi * (max - min) + min
        p.y = rng_f32(r) * 2.0f + -1.0f;
        p.z = rng_f32(r) * 2.0f + -1.0f;
        if (p.x * p.x + p.y * p.y + p.z * p.z <= 1.0f) break;
    }
    return unit(p);
}
// vec3.rs:240-254
__device__ __forceinline__ void random_in_unit_disk(Rng &r, float &px, float &py) {
    for (;;) {
        px = rng_f32(r) * 2.0f - 1.0f;
        py = rng_f32(r) * 2.0f - 1.0f;
        if (px * px + py * py <= 1.0f) break;
    }
}

// ---- device scene ------------------------------------------------------------------------------
// Hot, wave-uniform stream (scalar loads):  geom[i] = {cx, cy, cz, r*r},  vel[i] = {vx, vy, vz, 0}.
// Cold, per-lane record fetched only for the sphere a lane hit:
struct DevMat {
    float cm[3];          // tex < 0: tex_color * col_mod (sphere.rs:145, hoisted); else col_mod
    float metallicness;
    float opacity;
    float ir;
    int32_t tex;
    uint32_t pad0;
    float emitted[3];
    uint32_t pad1;
};

struct DevScene {
    const f4 *geom;
    const f4 *vel;
    const DevMat *mat;
    const RtwTexture *tex;
    const float *texels;
    uint32_t n;
    uint32_t moving;
};

// Rust `f as usize` (saturating, NaN -> 0), then clamped to the image like the oracle
__device__ __forceinline__ uint32_t tex_index(float f, uint32_t last) {
    if (!(f > 0.0f)) return 0;
    if (f >= (float)last) return last;
    return (uint32_t)f;
}

// sphere.rs:129-146 + texture.rs:259-267
__device__ __forceinline__ v3 sphere_albedo(const DevScene &sc, const DevMat &m, v3 normal) {
    if (m.tex < 0) return ld3(m.cm);
    const RtwTexture t = sc.tex[m.tex];
    const float PI = 3.14159265358979323846f, FRAC_1_PI = 0.318309886183790671538f;
    float u = (atan2f(-normal.z, normal.x) + PI) * FRAC_1_PI * 0.5f;
    float v = 1.0f - (FRAC_1_PI * acosf(-normal.y));
    uint32_t tx = tex_index(floorf(u * (float)(t.row - 1)), t.row - 1);
    uint32_t ty = tex_index(floorf(v * (float)(t.col - 1)), t.col - 1);
    const float *px = sc.texels + 3 * (size_t)(t.texel_offset + ty * t.row + tx);
    return (ld3(px) * 1.0f) * ld3(m.cm);
}

// materials.rs:89-97
__device__ __forceinline__ v3 refract(v3 uv, v3 n, float etai_over_etat) {
    float cos_theta = dot(-uv, n);
    if (cos_theta > 1.0f) cos_theta = 1.0f;
    v3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    v3 r_out_parallel = n * -__builtin_sqrtf(__builtin_fabsf(1.0f - len2(r_out_perp)));
    return r_out_perp + r_out_parallel;
}
// materials.rs:98-103, powi(5) = x * ((x*x)*(x*x))
__device__ __forceinline__ float reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = x * (x2 * x2);
    return r0 + (1.0f - r0) * x5;
}

// Material::on_hit (materials.rs:105-154) followed by the degenerate-direction fix-up of
// ray_color_* (ray_color.rs:31-33).  Returns the next direction; cos_theta for bg_color.
__device__ __forceinline__ v3 on_hit(const DevMat &m, v3 normal, v3 dir, Rng &rng, float &cos_theta) {
    const bool front = !(dot(dir, normal) > 0.0f);
    // Shared by both branches: unit(dir), and its mirror direction.  reflect(ud, -n) == reflect(ud, n)
    // bit for bit ((-n*2) * dot(ud,-n) == (n*2) * dot(ud,n): negation is exact and commutes with the
    // rounded products and sums), so the dielectric's reflect about the ray-facing normal is this too.
    const v3 ud = unit(dir);
    const v3 refl = reflect(ud, normal);
    v3 next;
    if (m.opacity > 0.0f) {
        const v3 n = front ? normal : -normal;
        const float ratio = front ? 1.0f / m.ir : m.ir;
        float ct = dot(-ud, n);
        if (ct > 1.0f) ct = 1.0f;
        const float st = __builtin_sqrtf(1.0f - ct * ct);
        const bool cannot_refract = ratio * st > 1.0f;
        const float rfl = reflectance(ct, ratio);
        bool do_reflect = cannot_refract;
        if (!do_reflect) do_reflect = rfl > rng_f32(rng);   // xi drawn only when refraction is possible
        next = do_reflect ? refl : refract(ud, n, ratio);
        cos_theta = 0.0f;
    } else {
        const v3 target = normal + random_unit_vec(rng);      // drawn even for mirrors (materials.rs:142)
        const v3 sc = close_to_zero(target) ? normal : target;
        next = refl * m.metallicness + sc * (1.0f - m.metallicness);
        cos_theta = (m.metallicness != 1.0f) ? dot(sc, normal) : 0.0f;
    }
    if (close_to_zero(next)) next = front ? normal : normal * -1.0f;
    return next;
}

// Rust2's Material trait objects (Rust2/src/objects/material.rs): MirrorGlass :130-162 (the Rust
// dielectric's arithmetic), Mirror :75-83 (reflects the un-normalised direction), Lambertian :25-36
// (unit(n + random_unit_vec())).  No degenerate-direction fix-up in Rust2's ray_color.
__device__ __forceinline__ v3 on_hit_rust2(const DevMat &m, v3 normal, v3 dir, Rng &rng) {
    if (m.opacity > 0.0f) {
        const bool front = !(dot(dir, normal) > 0.0f);
        const v3 n = front ? normal : -normal;
        const float ratio = front ? 1.0f / m.ir : m.ir;
        const v3 ud = unit(dir);
        float ct = dot(-ud, n);
        if (ct > 1.0f) ct = 1.0f;
        const float st = __builtin_sqrtf(1.0f - ct * ct);
        const bool cannot_refract = ratio * st > 1.0f;
        const float rfl = reflectance(ct, ratio);
        bool do_reflect = cannot_refract;
        if (!do_reflect) do_reflect = rfl > rng_f32(rng);
        return do_reflect ? reflect(ud, n) : refract(ud, n, ratio);
    }
    if (m.metallicness == 1.0f) return reflect(dir, normal);
    return unit(normal + random_unit_vec(rng));
}

// ray_color.rs:38-40
__device__ __forceinline__ v3 sky_gradient(v3 dir) {
    v3 ud = unit(dir);
    float t = 0.5f * (ud.y + 1.0f);
    return mk((1.0f - t) + t * 0.5f, (1.0f - t) + t * 0.7f, 1.0f);
}

} // namespace rtw
