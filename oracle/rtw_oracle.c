/*
 * rtw_oracle.c -- CPU oracle for the path-tracing hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this.  librtw_hip.so never calls it.
 *
 * What it is: a plain-C restatement, in f32 with NO fused multiply-add (build with
 * -ffp-contract=off; Rust never contracts), of the reference's per-pixel path
 *
 *   Rust/src/viewport.rs:270-305   render_row            -> render_pixel(), SAMPLER_ROW
 *   Rust/src/viewport.rs:430-478   Viewport::render      -> SAMPLER_STRATIFIED
 *   Rust/src/viewport.rs:479-516   render_no_rand        -> SAMPLER_NO_RAND
 *   Rust2/src/viewport.rs:87-114   Rust2 render_row      -> SAMPLER_CENTRES
 *   Rust/src/viewport.rs:308-401   Viewport::new         -> rtw_oracle_viewport_new()
 *   Rust/src/viewport/ray_color.rs:12-41  ray_color_gradient -> ray_color_gradient_rec()/path_iterative()
 *   Rust/src/viewport/ray_color.rs:43-92  ray_color_bg_color -> ray_color_bg_rec()
 *   Rust/src/viewport/glass_tests.rs:8-54 test integrator    -> INTEGRATOR_FLAG
 *   Rust2/src/viewport/ray_color.rs:12-37 + Rust2/src/objects/material.rs:25-162 -> INTEGRATOR_RUST2
 *   Rust/src/objects/sphere.rs:99-147     Sphere::collision_normal -> sphere_hit()
 *   Rust/src/objects/materials.rs:89-154,213-228 Material::on_hit  -> on_hit()
 *   Rust/src/vec3.rs:188-261              Vec3 math + samplers     -> v3_*, random_*()
 *   Rust/src/texture.rs:259-267           ImageTexture::color_at   -> texel()
 *   Rust/src/objects/quad.rs:37-110       Quad::collision_normal, Quad::new   -> quad_prepare(), quad_hit()
 *   Rust/src/objects/instance.rs:250-310  Instance::collision_normal (+ const_density :24-26) -> instance_hit()
 *   Rust/src/vec3.rs:161-181              Vec3::rotated                       -> rot_make(), v3_rotated()
 *   Rust/src/viewport.rs:136-150          Scene::collision_normal             -> closest_hit()
 *
 * Parity status.  The reference draws every random number from rand 0.8.5's ThreadRng (OS-seeded
 * ChaCha12, crate NOT vendored under /root/reference, Rust/Cargo.lock) and no reference test pins a
 * value that depends on it, so RNG-dependent output is "parity unpinned" against the reference: this
 * file defines its own counter-based stream (rng_seed/rng_next below) consumed in exactly the
 * reference's draw order.  Everything RNG-free IS pinned: tests/test_oracle_golden.py checks this
 * file against the Rust/cerr trace (58 pixels), against the reference's own C++ objects compiled
 * in oracle/_ref (s_test images, per-hit vectors), and against analytic known answers.
 */
#define _POSIX_C_SOURCE 200809L
#include "rtw_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------
 * Vec3 (Rust/src/vec3.rs:11-15,61-150,188-261).  Every function is one rounding per operation, in
 * the written order: (x*x + y*y) + z*z.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_ld(const float *p) { return v3_make(p[0], p[1], p[2]); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_div(v3 a, float s) { return v3_make(a.x / s, a.y / s, a.z / s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* vec3.rs:203 */
static inline float v3_len2(v3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }        /* vec3.rs:197 */
static inline float v3_len(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }  /* vec3.rs:200 */
static inline v3 v3_unit(v3 a) { return v3_div(a, v3_len(a)); }                        /* vec3.rs:213 */
static inline v3 v3_cross(v3 a, v3 b) {                                                /* vec3.rs:206 */
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* vec3.rs:256  *self - n * 2.0 * self.dot(n)  ==  self - ((n*2.0) * dot) */
static inline v3 v3_reflect(v3 a, v3 n) { return v3_sub(a, v3_scale(v3_scale(n, 2.0f), v3_dot(a, n))); }
/* vec3.rs:259 */
static inline int v3_close_to_zero(v3 a) { return fabsf(a.x) < 1e-7f && fabsf(a.y) < 1e-7f && fabsf(a.z) < 1e-7f; }

/* ------------------------------------------------------------------------------------------------
 * RNG.  NOT the reference's (see header).  One short stream per (seed, pixel, sample): a 32-bit LCG
 * (x <- 747796405 x + inc mod 2^32, the PCG family's multiplier) whose start state AND odd increment
 * come from a lowbias32 hash chain, so the image does not depend on how pixels are partitioned over
 * threads / lanes / GPUs.  A draw is the TOP 24 bits of the state (the strong bits of a power-of-two
 * LCG), xi = (x >> 8) * 2^-24 in [0,1): the mapping rand 0.8.5 `Standard` uses for f32.  (Rounds 1-2a
 * ran PCG's RXS-M-XS output permutation on top; on the GPU those 7 extra integer instructions per
 * draw were 9 % of the frame -- DESIGN.md "RNG" -- and tests/test_rng_quality.py finds nothing they
 * bought for streams this short and this well separated.)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t state, inc; uint32_t permuted; } rng_t;   /* permuted: RTW_ORACLE_FLAG_PERMUTED_STREAM (test-only, see rng_u32) */

static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
static inline rng_t rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t h = mix32((uint32_t)seed + 0x9E3779B9U);
    h = mix32(h ^ (uint32_t)(seed >> 32));
    h = mix32(h ^ pixel);
    h = mix32(h ^ sample);
    rng_t r; r.state = h; r.inc = mix32(h ^ 0x85EBCA6BU) | 1U; r.permuted = 0;
    return r;
}
static inline uint32_t rng_u32(rng_t *r) {
    uint32_t old = r->state;
    r->state = old * 747796405U + r->inc;
    if (r->permuted) {       /* TEST ONLY: the RXS-M-XS output stage of PCG that ran on top of the same LCG until the middle of round 2 (DESIGN.md "RNG"),
                              * kept so that tests can compare IMAGES rendered with and without it (tests/test_round3_cpu.py); the device never had a
                              * switch for it and the product stream is the plain one */
        uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
        return (word >> 22u) ^ word;
    }
    return old;
}
static inline float rng_f32(rng_t *r) { return (float)(rng_u32(r) >> 8) * (1.0f / 16777216.0f); }

void rtw_oracle_rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t state[2]) {
    rng_t r = rng_seed(seed, pixel, sample);
    state[0] = r.state; state[1] = r.inc;
}
float rtw_oracle_rng_next(uint32_t state[2]) {
    rng_t r; r.state = state[0]; r.inc = state[1]; r.permuted = 0;
    float f = rng_f32(&r);
    state[0] = r.state;
    return f;
}

/* vec3.rs:221-227  random::<f32>() * (max - min) + min, x then y then z */
static inline v3 random_vec(rng_t *r, float min, float max) {
    v3 p;
    p.x = rng_f32(r) * (max - min) + min;
    p.y = rng_f32(r) * (max - min) + min;
    p.z = rng_f32(r) * (max - min) + min;
    return p;
}
/* vec3.rs:228-239  rejection in the cube, accept len2 <= 1, then unit().
 * strict (RTW_FLAG_CPP_DIFFUSE): the C++ twin rejects len2 >= 1 (C++/src/vec3.cpp:28-34) */
static inline v3 random_unit_vec_s(rng_t *r, int strict) {
    for (;;) {
        v3 p = random_vec(r, -1.0f, 1.0f);
        float l2 = p.x * p.x + p.y * p.y + p.z * p.z;
        if (strict ? l2 < 1.0f : l2 <= 1.0f) return v3_unit(p);
    }
}
static inline v3 random_unit_vec(rng_t *r) { return random_unit_vec_s(r, 0); }
/* vec3.rs:240-254  (2 xi - 1, 2 xi - 1, 0), accept len2 <= 1  (C++/headers/vec3.h:35-41: < 1) */
static inline v3 random_in_unit_disk_s(rng_t *r, int strict) {
    for (;;) {
        v3 p;
        p.x = rng_f32(r) * 2.0f - 1.0f;
        p.y = rng_f32(r) * 2.0f - 1.0f;
        p.z = 0.0f;
        float l2 = p.x * p.x + p.y * p.y;
        if (strict ? l2 < 1.0f : l2 <= 1.0f) return p;
    }
}

/* ------------------------------------------------------------------------------------------------
 * Scene access
 * ---------------------------------------------------------------------------------------------- */
typedef struct { v3 origin, dir; float time; } ray_t;
static inline v3 ray_at(ray_t r, float t) { return v3_add(r.origin, v3_scale(r.dir, t)); } /* vec3.rs:289 */

/* `Material` (materials.rs:15-20) */
typedef struct { float metallicness, opacity, ir; float emitted[3]; } mat_t;

typedef struct {
    float t;
    v3 normal, point, col_mod;
    mat_t mat;
    int sphere;     /* index of the top-level object that was hit: spheres, then quads, then instances */
} hit_t;

typedef struct {
    uint64_t segments, sphere_tests, quad_tests;
} counters_t;

/* Rust `f as usize`: saturating, NaN -> 0 */
static inline uint32_t f32_as_usize(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

/* sphere.rs:129-146 + texture.rs:259-267: nearest texel by spherical UV, times col_mod.
 * Indices are clamped to the image (the reference would panic on an out-of-range index). */
static v3 sphere_albedo(const RtwScene *sc, const RtwSphere *s, v3 normal) {
    v3 tex;
    if (s->tex < 0 || (uint32_t)s->tex >= sc->n_textures) {
        tex = v3_ld(s->tex_color); /* 1x1: floor(u*0)=0, floor(v*0)=0 */
    } else {
        const RtwTexture *t = &sc->textures[s->tex];
        const float PI = 3.14159265358979323846f, FRAC_1_PI = 0.318309886183790671538f;
        float u = (atan2f(-normal.z, normal.x) + PI) * FRAC_1_PI * 0.5f;
        float v = 1.0f - (FRAC_1_PI * acosf(-normal.y));
        uint32_t tx = f32_as_usize(floorf(u * (float)(t->row - 1)));
        uint32_t ty = f32_as_usize(floorf(v * (float)(t->col - 1)));
        if (tx > t->row - 1) tx = t->row - 1;
        if (ty > t->col - 1) ty = t->col - 1;
        tex = v3_ld(&sc->texels[3 * (size_t)(t->texel_offset + ty * t->row + tx)]);
    }
    tex = v3_scale(tex, 1.0f);                    /* texture.rs:265  img[..] * noise_mult, noise None */
    return v3_mul(tex, v3_ld(s->col_mod));        /* sphere.rs:145 */
}

/* TEST ONLY: the device's lean atan2 / acos for the spherical UV (csrc/rtw_device.h atan2_plain / acos_plain), restated operation for operation --
 * `/` and sqrtf are correctly rounded here, as div_plain / sqrt_plain are on the device for the arguments they are used with -- so that their
 * accuracy and their effect on the texel choice can be measured on the CPU (tests/test_round3_cpu.py).  The oracle's own renders use libm
 * (sphere_albedo below), like the reference (Rust's f32::atan2 / f32::acos). */
static float atan2_plain(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float k = fmaxf(ax, ay) < 0x1p-60f ? 0x1p80f : 1.0f;
    const float mx = fmaxf(ax, ay) * k, mn = fminf(ax, ay) * k;
    const float t = mx == 0.0f ? 0.0f : mn / mx;
    const float s = t * t;
    float p = 0.002974563976749778f;
    p = fmaf(p, s, -0.016581078991293907f); p = fmaf(p, s, 0.043553370982408524f); p = fmaf(p, s, -0.07580564171075821f);
    p = fmaf(p, s, 0.10678933560848236f);   p = fmaf(p, s, -0.14214207231998444f); p = fmaf(p, s, 0.19994136691093445f);
    p = fmaf(p, s, -0.3333316743373871f);
    float a = fmaf(t * s, p, t);
    a = ay > ax ? 1.57079632679489661923f - a : a;
    a = signbit(x) ? 3.14159265358979323846f - a : a;
    a = (x != x || y != y) ? NAN : a;
    return copysignf(a, y);
}
static float acos_plain(float x) {
    const float ax = fabsf(x);
    const int small = ax <= 0.5f;
    const float s = small ? x * x : (1.0f - ax) * 0.5f;
    const float q = small ? x : sqrtf(s);
    float p = 0.04221854731440544f;
    p = fmaf(p, s, 0.02414761111140251f); p = fmaf(p, s, 0.04547709599137306f); p = fmaf(p, s, 0.07495241612195969f);
    p = fmaf(p, s, 0.16666753590106964f);
    const float r = fmaf(q * s, p, q);
    const float two = r + r;
    return small ? 1.57079632679489661923f - r : (x < 0.0f ? 3.14159265358979323846f - two : two);
}
/* (u, v) of sphere.rs:132-133 for n normals; plain != 0: through the device's sequences, else through libm.  out: [n][4] = atan2, acos, u, v */
void rtw_oracle_sphere_uv(const float *normals, size_t n, int plain, float *out) {
    const float PI = 3.14159265358979323846f, FRAC_1_PI = 0.318309886183790671538f;
    for (size_t i = 0; i < n; i++) {
        const float y = -normals[3 * i + 2], x = normals[3 * i], c = -normals[3 * i + 1];
        const float at = plain ? atan2_plain(y, x) : atan2f(y, x), ac = plain ? acos_plain(c) : acosf(c);
        out[4 * i] = at; out[4 * i + 1] = ac;
        out[4 * i + 2] = (at + PI) * FRAC_1_PI * 0.5f;
        out[4 * i + 3] = 1.0f - (FRAC_1_PI * ac);
    }
}

/* Rust2's ImageTexture::color_at (Rust2/src/objects/texture.rs:94-105) as written: the sample of an image of `width` x `height` texels at
 * (u, v) is img[x * width + y] with x = (u * width) as usize, y = (v * height) as usize -- scaled by the size (not size - 1 as in Rust/) and
 * indexed TRANSPOSED (x * width + y, not y * width + x: most likely a bug, and part of the contract); the emission image
 * uses floor() before the cast (:95-96), which is the same index for the non-negative u, v a sphere produces.  The reference panics when the
 * index leaves the Vec (u == 1, or a tall image); the restatement and the device clamp it to the last texel. */
static inline size_t rust2_texel_index(float u, float v, uint32_t width, uint32_t height, int floor_first) {
    const float fx = u * (float)width, fy = v * (float)height;
    const uint64_t x = f32_as_usize(floor_first ? floorf(fx) : fx), y = f32_as_usize(floor_first ? floorf(fy) : fy);
    const uint64_t idx = x * (uint64_t)width + y, last = (uint64_t)width * height - 1u;
    return (size_t)(idx > last ? last : idx);
}
uint32_t rtw_oracle_rust2_texel_index(float u, float v, uint32_t width, uint32_t height, int emission) {
    return (uint32_t)rust2_texel_index(u, v, width, height, emission);
}
/* Rust2 Sphere::color (Rust2/src/objects/sphere.rs:92-107) for a sphere with an image texture: ColorResult{emmited, multiplied}.  `multiplied`
 * is the texel (times the POD's col_mod, which a Rust2 scene leaves at 1: x * 1.0 == x); `emmited` the texel of the texture's emission
 * image (RtwTexture.emit_tex), or the sphere's constant emission when it has none. */
static void rust2_sphere_color(const RtwScene *sc, const RtwSphere *s, v3 normal, v3 *mult, v3 *emit) {
    const RtwTexture *t = &sc->textures[s->tex];
    const float PI = 3.14159265358979323846f, FRAC_1_PI = 0.318309886183790671538f;
    const float u = (atan2f(-normal.z, normal.x) + PI) * FRAC_1_PI * 0.5f;
    const float v = 1.0f - (FRAC_1_PI * acosf(-normal.y));
    *mult = v3_mul(v3_ld(&sc->texels[3 * ((size_t)t->texel_offset + rust2_texel_index(u, v, t->row, t->col, 0))]), v3_ld(s->col_mod));
    if (t->emit_tex != 0 && t->emit_tex <= sc->n_textures) {
        const RtwTexture *e = &sc->textures[t->emit_tex - 1];
        *emit = v3_ld(&sc->texels[3 * ((size_t)e->texel_offset + rust2_texel_index(u, v, e->row, e->col, 1))]);
    } else *emit = v3_ld(s->emitted);
}

/* sphere.rs:99-147.  Returns 1 and fills *t (no normal yet) when the sphere reports Some(Hit). */
static inline int sphere_hit_t(const RtwSphere *s, ray_t r, float mint, float maxt, float *t_out, v3 *centre_out) {
    v3 origin = v3_add(v3_ld(s->center), v3_scale(v3_ld(s->velocity), r.time)); /* :100 */
    v3 oc = v3_sub(r.origin, origin);
    float a = v3_dot(r.dir, r.dir);
    float b = v3_dot(oc, r.dir);
    float c = v3_dot(oc, oc) - s->radius * s->radius;
    float d = b * b - a * c;
    if (d < 0.0f) return 0;
    float x = (-b - sqrtf(d)) / a;
    if (x < mint) x = (-b + sqrtf(d)) / a;
    if (x < mint || x > maxt) return 0;
    *t_out = x; *centre_out = origin;
    return 1;
}

static inline mat_t sphere_mat(const RtwSphere *s) {
    mat_t m = { s->metallicness, s->opacity, s->ir, { s->emitted[0], s->emitted[1], s->emitted[2] } };
    return m;
}

/* Closest sphere of a list, in list order: `min_hit == None || min_hit > i` (camera_tests.rs:19-33,
 * aabb/aabb.rs:140-152) -- strict, so the first of equal t wins; a NaN t can only enter first.
 * (The reference's AABB tree only prunes this loop; its bounds ignore velocity, which this restatement
 * does not reproduce: DESIGN.md 1.) */
static int closest_sphere(const RtwScene *sc, const RtwSphere *list, uint32_t n, ray_t r, float mint, float maxt,
                          hit_t *h, counters_t *cn) {
    int best = -1; float best_t = 0.0f; v3 best_c = { 0, 0, 0 };
    for (uint32_t i = 0; i < n; i++) {
        float t; v3 c;
        cn->sphere_tests++;
        if (!sphere_hit_t(&list[i], r, mint, maxt, &t, &c)) continue;
        if (best < 0 || best_t > t) { best = (int)i; best_t = t; best_c = c; }
    }
    if (best < 0) return 0;
    h->t = best_t;
    h->sphere = best;
    h->point = ray_at(r, best_t);
    h->normal = v3_unit(v3_sub(ray_at(r, best_t), best_c));   /* :127 */
    h->col_mod = sphere_albedo(sc, &list[best], h->normal);
    h->mat = sphere_mat(&list[best]);
    return 1;
}

/* ---- quads (objects/quad.rs) ------------------------------------------------------------------- */
typedef struct { v3 normal, w; float d; } quad_derived_t;

/* Quad::new (quad.rs:96-108): n = u x v; normal = unit(n); d = normal . origin; w = n / (n . n) */
static inline quad_derived_t quad_prepare(const RtwQuad *q) {
    quad_derived_t g;
    v3 n = v3_cross(v3_ld(q->u), v3_ld(q->v));
    g.normal = v3_unit(n);
    g.d = v3_dot(g.normal, v3_ld(q->origin));
    g.w = v3_div(n, v3_dot(n, n));
    return g;
}

/* Quad::collision_normal (quad.rs:37-81) */
static int quad_hit(const RtwScene *sc, const RtwQuad *q, ray_t r, float mint, float maxt, hit_t *h) {
    quad_derived_t g = quad_prepare(q);
    float denominator = v3_dot(g.normal, r.dir);
    if (fabsf(denominator) <= 1e-8f) return 0;
    float t = (g.d - v3_dot(g.normal, r.origin)) / denominator;
    if (t < mint || t > maxt) return 0;
    v3 point = ray_at(r, t);
    v3 planar = v3_sub(point, v3_ld(q->origin));
    float alfa = v3_dot(g.w, v3_cross(planar, v3_ld(q->v)));
    float beta = v3_dot(g.w, v3_cross(v3_ld(q->u), planar));
    if (alfa < 0.0f || alfa > 1.0f || beta < 0.0f || beta > 1.0f) return 0;
    v3 tex;
    if (q->tex < 0 || (uint32_t)q->tex >= sc->n_textures) {
        tex = v3_ld(q->tex_color);                 /* 1x1: both indices are 0 */
    } else {
        const RtwTexture *tx = &sc->textures[q->tex];
        uint32_t ix = alfa != 1.0f ? f32_as_usize(floorf(alfa * (float)tx->row)) : tx->row - 1;   /* :66-70 */
        uint32_t iy = beta != 1.0f ? f32_as_usize(floorf(beta * (float)tx->col)) : tx->col - 1;   /* :71-75 */
        if (ix > tx->row - 1) ix = tx->row - 1;
        if (iy > tx->col - 1) iy = tx->col - 1;
        tex = v3_ld(&sc->texels[3 * (size_t)(tx->texel_offset + iy * tx->row + ix)]);
    }
    h->t = t; h->normal = g.normal; h->point = point;
    h->col_mod = v3_scale(tex, 1.0f);              /* texture.rs:265, noise None */
    h->mat.metallicness = q->metallicness; h->mat.opacity = q->opacity; h->mat.ir = q->ir;
    h->mat.emitted[0] = q->emitted[0]; h->mat.emitted[1] = q->emitted[1]; h->mat.emitted[2] = q->emitted[2];
    return 1;
}

/* Closest quad of a list in list order (QuadAABB::collision_normal, aabb/qaabb.rs:196-258, prunes this loop). */
static int closest_quad(const RtwScene *sc, const RtwQuad *list, uint32_t n, ray_t r, float mint, float maxt,
                        hit_t *h, counters_t *cn) {
    int best = -1;
    for (uint32_t i = 0; i < n; i++) {
        hit_t q;
        cn->quad_tests++;
        if (!quad_hit(sc, &list[i], r, mint, maxt, &q)) continue;
        if (best < 0 || h->t > q.t) { *h = q; h->sphere = (int)i; best = (int)i; }
    }
    return best >= 0;
}

/* ---- instances (objects/instance.rs) ------------------------------------------------------------- */
/* Vec3::rotated (vec3.rs:161-181), as written: the y-coefficient of x reads `asin*bsin*ccos - asin*ccos`. */
typedef struct { float asin, acos, bsin, bcos, csin, ccos; } rot_t;
static inline rot_t rot_make(v3 rot) {
    rot_t q = { sinf(rot.x), cosf(rot.x), sinf(rot.y), cosf(rot.y), sinf(rot.z), cosf(rot.z) };
    return q;
}
static inline v3 v3_rotated(v3 a, rot_t q) {
    v3 o;
    o.x = a.x * q.bcos * q.ccos + a.y * (q.asin * q.bsin * q.ccos - q.asin * q.ccos) + a.z * (q.acos * q.bsin * q.ccos + q.asin * q.csin);
    o.y = a.x * q.bcos * q.csin + a.y * (q.asin * q.bsin * q.csin + q.acos * q.ccos) + a.z * (q.acos * q.bsin * q.csin - q.asin * q.ccos);
    o.z = a.x * -q.bsin + a.y * q.asin * q.bcos + a.z * q.acos * q.bcos;
    return o;
}

/* Vec3::rotated exposed for the reference's own known answers (vec3.rs:363-404 rotation_tests) */
void rtw_oracle_rotated(const float v[3], const float rot[3], float out[3]) {
    v3 r = v3_rotated(v3_ld(v), rot_make(v3_ld(rot)));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ln(x) for a positive normal f32, computed in f64 (atanh series of (m-1)/(m+1), 12 terms) and rounded once:
 * the correctly rounded result for every xi = k * 2^-24 the RNG can produce (checked exhaustively against logl
 * in tests/test_oracle_golden.py).  Written out instead of calling libm so that the device, which evaluates
 * the same f64 operations, returns the same bits.  Rust's f32::ln is the platform logf (<= 1 ulp from this). */
static inline float ln_f32(float xf) {
    if (xf == 0.0f) return -INFINITY;
    uint32_t bits; memcpy(&bits, &xf, 4);
    int e = (int)(bits >> 23) - 127;
    uint32_t mb = (bits & 0x007FFFFFu) | 0x3F800000u;
    float mf; memcpy(&mf, &mb, 4);
    double m = (double)mf;
    if (mf > 1.41421354f) { m = m * 0.5; e += 1; }
    double f = m - 1.0;
    double s = f / (2.0 + f);
    double z = s * s;
    double p = 1.0 / 25.0;
    p = 1.0 / 23.0 + z * p; p = 1.0 / 21.0 + z * p; p = 1.0 / 19.0 + z * p; p = 1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p; p = 1.0 / 13.0 + z * p; p = 1.0 / 11.0 + z * p; p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;  p = 1.0 / 5.0 + z * p;  p = 1.0 / 3.0 + z * p;
    double lm = 2.0 * s + (2.0 * s) * (z * p);
    return (float)((double)e * 0.6931471805599453094 + lm);
}
float rtw_oracle_ln(float x) { return ln_f32(x); }
void rtw_oracle_ln_bulk(const float *x, float *out, size_t n) { for (size_t i = 0; i < n; i++) out[i] = ln_f32(x[i]); }

/* closest member of an instance, spheres then quads (instance.rs:263-273: `for i in vec![s_hit, q_hit]`) */
static int instance_members(const RtwScene *sc, const RtwInstance *in, ray_t r, float mint, float maxt,
                            hit_t *h, counters_t *cn) {
    hit_t s_hit, q_hit; int found = 0;
    if (closest_sphere(sc, sc->inst_spheres + in->first_sphere, in->n_spheres, r, mint, maxt, &s_hit, cn)) { *h = s_hit; found = 1; }
    if (closest_quad(sc, sc->inst_quads + in->first_quad, in->n_quads, r, mint, maxt, &q_hit, cn)) {
        if (!found || h->t > q_hit.t) { *h = q_hit; found = 1; }
    }
    return found;
}

/* Instance::collision_normal (instance.rs:250-310) */
static int instance_hit(const RtwScene *sc, const RtwInstance *in, ray_t r0, float mint, float maxt,
                        hit_t *h, counters_t *cn, rng_t *rng) {
    v3 tr = v3_ld(in->translation), rot = v3_ld(in->rotation);
    rot_t back = rot_make(v3_neg(rot)), fwd = rot_make(rot);
    ray_t r;                                        /* change to local (:257-258) */
    r.origin = v3_rotated(v3_sub(r0.origin, tr), back);
    r.dir = v3_rotated(r0.dir, back);
    r.time = r0.time;
    if (!instance_members(sc, in, r, mint, maxt, h, cn)) return 0;
    if (in->medium == RTW_MEDIUM_CONST_DENSITY) {   /* const_density (:24-26): thread_rng().gen::<f32>().ln() / -d */
        float distance = ln_f32(rng_f32(rng)) / -in->density;
        if (distance >= 0.0f) {
            hit_t second;
            r.origin = v3_add(h->point, v3_scale(r.dir, distance));
            if (!instance_members(sc, in, r, mint, maxt, &second, cn)) return 0;   /* left the volume first (:292) */
            h->point = r.origin;
            h->normal = random_unit_vec(rng);
        }
    }
    h->point = v3_add(v3_rotated(h->point, fwd), tr);  /* :301-302 */
    h->normal = v3_rotated(h->normal, fwd);            /* :304 */
    return 1;
}

/* Scene::collision_normal (viewport.rs:136-150): spheres, quads, instances; a later hit replaces an earlier
 * one only when strictly closer.  The three AABB trees (aabb.rs, qaabb.rs, iaabb.rs) only prune the list
 * walks below; instances are visited in list order (the reference's IAABB order comes from an unseeded
 * random split axis, aabb/iaabb.rs:87, and matters only for which medium draws first). */
static int closest_hit(const RtwScene *sc, ray_t r, float mint, float maxt, hit_t *h, counters_t *cn, rng_t *rng) {
    int found = 0; hit_t c;
    cn->segments++;
    if (closest_sphere(sc, sc->spheres, sc->n_spheres, r, mint, maxt, &c, cn)) { *h = c; found = 1; }
    if (sc->n_quads && closest_quad(sc, sc->quads, sc->n_quads, r, mint, maxt, &c, cn)) {
        if (!found || h->t > c.t) { *h = c; h->sphere = (int)sc->n_spheres + c.sphere; found = 1; }
    }
    if (sc->n_instances) {
        int ibest = 0; hit_t ih; int ifound = 0;
        for (uint32_t i = 0; i < sc->n_instances; i++) {
            if (!instance_hit(sc, &sc->instances[i], r, mint, maxt, &c, cn, rng)) continue;
            if (!ifound || ih.t > c.t) { ih = c; ibest = (int)i; ifound = 1; }
        }
        if (ifound && (!found || h->t > ih.t)) { *h = ih; h->sphere = (int)(sc->n_spheres + sc->n_quads) + ibest; found = 1; }
    }
    return found;
}

/* materials.rs:89-97 */
static inline v3 refract(v3 uv, v3 n, float etai_over_etat) {
    float cos_theta = v3_dot(v3_neg(uv), n);
    if (cos_theta > 1.0f) cos_theta = 1.0f;
    v3 r_out_perp = v3_scale(v3_add(uv, v3_scale(n, cos_theta)), etai_over_etat);
    v3 r_out_parallel = v3_scale(n, -sqrtf(fabsf(1.0f - v3_len2(r_out_perp))));
    return v3_add(r_out_perp, r_out_parallel);
}
/* materials.rs:98-103; powi(5) == x * ((x*x)*(x*x)) (LLVM powi expansion / compiler-rt __powisf2) */
static inline float reflectance(float cosine, float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    r0 = r0 * r0;
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = x * (x2 * x2);
    return r0 + (1.0f - r0) * x5;
}

typedef struct { ray_t next; v3 emitted; float cos_theta; int front_face, cannot_refract; float ratio; int cpp_fixed; } scatter_t;

/* materials.rs:105-154 (+ diffuse :213-228) */
static scatter_t on_hit(const mat_t *s, const hit_t *h, ray_t r, rng_t *rng, uint32_t flags) {
    scatter_t o; memset(&o, 0, sizeof o);
    o.emitted = v3_ld(s->emitted);
    o.front_face = !(v3_dot(r.dir, h->normal) > 0.0f);
    if (s->opacity > 0.0f) {
        v3 n = o.front_face ? h->normal : v3_neg(h->normal);
        float refraction_ratio = o.front_face ? 1.0f / s->ir : s->ir;
        v3 unit_direction = v3_unit(r.dir);
        float cos_theta = v3_dot(v3_neg(unit_direction), n);
        if (cos_theta > 1.0f) cos_theta = 1.0f;
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        int cannot_refract = refraction_ratio * sin_theta > 1.0f;
        float refl = reflectance(cos_theta, refraction_ratio);
        v3 direction;
        /* `cannot_refract || reflectance > random()`: xi is drawn only when the first test is false */
        int do_reflect = cannot_refract;
        if (!do_reflect && !(flags & RTW_FLAG_CPP_DIELECTRIC)) do_reflect = refl > rng_f32(rng);
        if (do_reflect) direction = v3_reflect(unit_direction, n);
        else direction = refract(unit_direction, n, refraction_ratio);
        o.next.origin = h->point; o.next.dir = direction; o.next.time = r.time;
        o.cos_theta = 0.0f; o.cannot_refract = cannot_refract; o.ratio = refraction_ratio;
        return o;
    }
    if (flags & RTW_FLAG_CPP_DIFFUSE) {
        /* the C++ twin, in f32 (C++/headers/materials.h:113-117, C++/src/materials.cpp:4-13, C++/src/sphere.cpp:29-31):
         *   sc = unit((point + normal + random_unit_vec()) - point) * (1 - m);  reflect = unit(reflect(unit(d), n));
         *   direction = reflect * m + sc;  near_zero (1e-8) -> normal.  cpp_fixed tells fix_degenerate to keep its hands off. */
        v3 target = v3_add(v3_add(h->point, h->normal), random_unit_vec_s(rng, 1));
        v3 sdir = v3_unit(v3_sub(target, h->point));
        v3 mdir = v3_unit(v3_reflect(v3_unit(r.dir), h->normal));
        v3 dir = v3_add(v3_scale(mdir, s->metallicness), v3_scale(sdir, 1.0f - s->metallicness));
        if (fabsf(dir.x) < 1e-8f && fabsf(dir.y) < 1e-8f && fabsf(dir.z) < 1e-8f) dir = h->normal;
        o.next.origin = h->point; o.next.dir = dir; o.next.time = r.time;
        o.cos_theta = (s->metallicness != 1.0f) ? v3_dot(sdir, h->normal) : 0.0f;
        o.cpp_fixed = 1;
        return o;
    }
    /* diffuse(): drawn even for metallicness == 1 (materials.rs:142) */
    v3 target = v3_add(h->normal, random_unit_vec(rng));
    v3 sc = v3_close_to_zero(target) ? h->normal : target;
    v3 refl = v3_reflect(v3_unit(r.dir), h->normal);
    v3 dir = v3_add(v3_scale(refl, s->metallicness), v3_scale(sc, 1.0f - s->metallicness));
    o.next.origin = h->point; o.next.dir = dir; o.next.time = r.time;
    o.cos_theta = (s->metallicness != 1.0f) ? v3_dot(sc, h->normal) : 0.0f;
    return o;
}

/* ray_color.rs:31-33 */
static inline void fix_degenerate(scatter_t *s, const hit_t *h) {
    if (s->cpp_fixed) return;                       /* the C++ branch applied its own near_zero rule */
    if (v3_close_to_zero(s->next.dir)) s->next.dir = s->front_face ? h->normal : v3_scale(h->normal, -1.0f);
}

/* ray_color.rs:38-40 */
static inline v3 sky_gradient(v3 dir) {
    v3 ud = v3_unit(dir);
    float t = 0.5f * (ud.y + 1.0f);
    return v3_make((1.0f - t) + t * 0.5f, (1.0f - t) + t * 0.7f, 1.0f);
}

typedef struct {
    const RtwScene *sc; const RtwParams *p; counters_t *cn; rng_t *rng;
    RtwOracleBounce *trace; int trace_cap, trace_n;
} ctx_t;

static void trace_record(ctx_t *c, int hit, const hit_t *h, const scatter_t *s, ray_t r) {
    if (!c->trace || c->trace_n >= c->trace_cap) return;
    RtwOracleBounce *b = &c->trace[c->trace_n++];
    memset(b, 0, sizeof *b);
    b->hit = hit;
    v3 ud = v3_unit(r.dir);
    b->unit_dir[0] = ud.x; b->unit_dir[1] = ud.y; b->unit_dir[2] = ud.z;
    if (!hit) { b->sphere = -1; return; }
    b->sphere = h->sphere; b->t = h->t;
    b->normal[0] = h->normal.x; b->normal[1] = h->normal.y; b->normal[2] = h->normal.z;
    b->point[0] = h->point.x; b->point[1] = h->point.y; b->point[2] = h->point.z;
    if (s) {
        b->front_face = s->front_face; b->cannot_refract = s->cannot_refract; b->ratio = s->ratio;
        b->next_dir[0] = s->next.dir.x; b->next_dir[1] = s->next.dir.y; b->next_dir[2] = s->next.dir.z;
    }
}

/* ray_color_gradient, reference recursion order: (child * cm) on the way back (ray_color.rs:35). */
static v3 ray_color_gradient_rec(ctx_t *c, ray_t r, uint32_t depth) {
    if (depth < 1) return v3_make(0, 0, 0);
    hit_t h;
    if (closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
        scatter_t s = on_hit(&h.mat, &h, r, c->rng, c->p->flags);
        fix_degenerate(&s, &h);
        trace_record(c, 1, &h, &s, r);
        return v3_mul(ray_color_gradient_rec(c, s.next, depth - 1), h.col_mod);
    }
    trace_record(c, 0, NULL, NULL, r);
    return sky_gradient(r.dir);
}

/* The same path as ray_color_gradient, accumulated front-to-back: throughput = cm_1*cm_2*..., then
 * sky * throughput.  Identical geometry and RNG draws; the colour differs from the recursion order
 * only by the association of the f32 products (<= depth ulp).  This is the order the GPU uses. */
static v3 ray_color_gradient_iter(ctx_t *c, ray_t r, uint32_t depth) {
    v3 thr = v3_make(1.0f, 1.0f, 1.0f);
    for (uint32_t k = 0; k < depth; k++) {
        hit_t h;
        if (!closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
            trace_record(c, 0, NULL, NULL, r);
            return v3_mul(sky_gradient(r.dir), thr);
        }
        scatter_t s = on_hit(&h.mat, &h, r, c->rng, c->p->flags);
        fix_degenerate(&s, &h);
        trace_record(c, 1, &h, &s, r);
        thr = v3_mul(thr, h.col_mod);
        r = s.next;
    }
    return v3_make(0, 0, 0);
}

/* materials.rs:5-13 */
static inline float lambertian_scatter_pdf(float cos_theta) {
    return cos_theta > 0.0f ? cos_theta * 0.318309886183790671538f : 0.0f;
}

/* ray_color_bg_color (ray_color.rs:43-92), reference recursion order. */
static v3 ray_color_bg_rec(ctx_t *c, ray_t r, uint32_t depth) {
    if (depth < 1) return v3_make(0, 0, 0);
    hit_t h;
    if (closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
        const mat_t *sp = &h.mat;
        scatter_t s = on_hit(sp, &h, r, c->rng, c->p->flags);
        fix_degenerate(&s, &h);
        trace_record(c, 1, &h, &s, r);
        v3 color = v3_mul(ray_color_bg_rec(c, s.next, depth - 1), h.col_mod);
        v3 scattered;
        if (sp->metallicness != 1.0f) {
            float pdf = lambertian_scatter_pdf(s.cos_theta);
            scattered = v3_div(v3_scale(color, pdf), pdf);   /* color * scattering_pdf / pdf : NaN when pdf == 0 */
        } else scattered = v3_make(0, 0, 0);
        return v3_add(v3_add(v3_scale(color, sp->metallicness), v3_scale(scattered, 1.0f - sp->metallicness)), s.emitted);
    }
    trace_record(c, 0, NULL, NULL, r);
    return v3_ld(c->sc->background);
}

/* Front-to-back form of ray_color_bg_color used by the GPU: L = sum_k T_k * e_k + T_n * bg with
 * T_k = prod cm_j; a bounce with metallicness != 1 and pdf == 0 poisons the path to NaN exactly as
 * the reference's 0/0 does.  (color*pdf)/pdf is taken as color: <= 1 ulp per bounce. */
static v3 ray_color_bg_iter(ctx_t *c, ray_t r, uint32_t depth) {
    v3 thr = v3_make(1.0f, 1.0f, 1.0f), L = v3_make(0, 0, 0);
    int poison = 0;
    for (uint32_t k = 0; k < depth; k++) {
        hit_t h;
        if (!closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
            trace_record(c, 0, NULL, NULL, r);
            L = v3_add(L, v3_mul(v3_ld(c->sc->background), thr));
            goto done;
        }
        const mat_t *sp = &h.mat;
        scatter_t s = on_hit(sp, &h, r, c->rng, c->p->flags);
        fix_degenerate(&s, &h);
        trace_record(c, 1, &h, &s, r);
        if (sp->metallicness != 1.0f && !(lambertian_scatter_pdf(s.cos_theta) > 0.0f)) poison = 1;
        L = v3_add(L, v3_mul(s.emitted, thr));
        thr = v3_mul(thr, h.col_mod);
        r = s.next;
    }
done:
    if (poison) { float n = nanf(""); return v3_make(n, n, n); }
    return L;
}

/* C++/src/tests.cpp:76-97 ray_colorSc: (normal + 1) * 0.5 of the closest hit, else sky */
static v3 ray_color_normal(ctx_t *c, ray_t r) {
    hit_t h;
    if (closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
        trace_record(c, 1, &h, NULL, r);
        return v3_scale(v3_make(h.normal.x + 1.0f, h.normal.y + 1.0f, h.normal.z + 1.0f), 0.5f);
    }
    trace_record(c, 0, NULL, NULL, r);
    return sky_gradient(r.dir);
}

/* glass_tests.rs:8-54: non-mirror hit -> yellow, sky -> blue, mirror/dielectric -> recurse * cm */
static v3 ray_color_flag(ctx_t *c, ray_t r, uint32_t depth) {
    v3 thr = v3_make(1.0f, 1.0f, 1.0f);
    for (uint32_t k = 0; k < depth; k++) {
        hit_t h;
        if (!closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
            trace_record(c, 0, NULL, NULL, r);
            return v3_mul(v3_make(0.0f, 0.0f, 1.0f), thr);
        }
        const mat_t *sp = &h.mat;
        if (sp->metallicness != 1.0f) {
            trace_record(c, 1, &h, NULL, r);
            return v3_mul(v3_make(1.0f, 1.0f, 0.0f), thr);
        }
        scatter_t s = on_hit(sp, &h, r, c->rng, c->p->flags);
        fix_degenerate(&s, &h);
        trace_record(c, 1, &h, &s, r);
        thr = v3_mul(thr, h.col_mod);
        r = s.next;
    }
    return v3_make(0, 0, 0);
}

/* ---- Rust2 trait surface ------------------------------------------------------ */
/* Material::on_hit of Rust2/src/objects/material.rs: Lambertian :25-36, Mirror :75-83, MirrorGlass :130-162.
 * The hit's ray is the incoming ray (Hit.r), its normal the outward one (Rust2/src/objects/sphere.rs:56-84). */
static ray_t rust2_on_hit(const mat_t *s, const hit_t *h, ray_t r, rng_t *rng, uint32_t flags) {
    ray_t o; o.origin = h->point; o.time = r.time;
    if (s->opacity > 0.0f) {                                      /* MirrorGlass{ir}: same arithmetic as the Rust dielectric */
        int front_face = !(v3_dot(r.dir, h->normal) > 0.0f);
        v3 n = front_face ? h->normal : v3_neg(h->normal);
        float refraction_ratio = front_face ? 1.0f / s->ir : s->ir;
        v3 unit_direction = v3_unit(r.dir);
        float cos_theta = v3_dot(v3_neg(unit_direction), n);
        if (cos_theta > 1.0f) cos_theta = 1.0f;
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        int cannot_refract = refraction_ratio * sin_theta > 1.0f;
        float refl = reflectance(cos_theta, refraction_ratio);
        int do_reflect = cannot_refract;
        if (!do_reflect && !(flags & RTW_FLAG_CPP_DIELECTRIC)) do_reflect = refl > rng_f32(rng);
        o.dir = do_reflect ? v3_reflect(unit_direction, n) : refract(unit_direction, n, refraction_ratio);
    } else if (s->metallicness == 1.0f) {
        o.dir = v3_reflect(r.dir, h->normal);                     /* Mirror: h.r.direction.reflect(h.n), not normalised (:80) */
    } else {
        o.dir = v3_unit(v3_add(h->normal, random_unit_vec(rng))); /* Lambertian: (h.n + random_unit_vec()).unit() (:28) */
    }
    return o;
}

/* A top-level sphere with an image texture takes its ColorResult from Rust2's own lookup rule (instance members keep the Rust/ rule: instances
 * are the Rust/ tree's, Rust2 has none). */
static void rust2_color_override(const RtwScene *sc, const hit_t *h, v3 *multiplied, v3 *emmited) {
    if (h->sphere >= 0 && (uint32_t)h->sphere < sc->n_spheres) {
        const RtwSphere *s = &sc->spheres[h->sphere];
        if (s->tex >= 0 && (uint32_t)s->tex < sc->n_textures) rust2_sphere_color(sc, s, h->normal, multiplied, emmited);
    }
}

/* Rust2 ray_color, the reference's recursion (Rust2/src/viewport/ray_color.rs:12-37). */
static v3 ray_color_rust2_rec(ctx_t *c, ray_t r, uint32_t depth) {
    if (depth == 0) return v3_ld(c->sc->background);
    hit_t h;
    if (closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
        const mat_t *sp = &h.mat;
        ray_t next = rust2_on_hit(sp, &h, r, c->rng, c->p->flags);   /* o.color(&h) draws nothing; o.reflect(&h) does */
        trace_record(c, 1, &h, NULL, r);
        v3 emmited = v3_ld(sp->emitted), multiplied = h.col_mod;
        rust2_color_override(c->sc, &h, &multiplied, &emmited);
        v3 next_color = ray_color_rust2_rec(c, next, depth - 1);
        return v3_add(emmited, v3_mul(next_color, multiplied));      /* emmited + next.field_wise_mult(multiplied) */
    }
    trace_record(c, 0, NULL, NULL, r);
    return v3_ld(c->sc->background);
}

/* Front-to-back form used by the GPU (same paths and draws; sums/products associate differently). */
static v3 ray_color_rust2_iter(ctx_t *c, ray_t r, uint32_t depth) {
    v3 thr = v3_make(1.0f, 1.0f, 1.0f), L = v3_make(0, 0, 0);
    for (uint32_t k = 0; k < depth; k++) {
        hit_t h;
        if (!closest_hit(c->sc, r, c->p->mint, c->p->maxt, &h, c->cn, c->rng)) {
            trace_record(c, 0, NULL, NULL, r);
            return v3_add(L, v3_mul(v3_ld(c->sc->background), thr));
        }
        const mat_t *sp = &h.mat;
        ray_t next = rust2_on_hit(sp, &h, r, c->rng, c->p->flags);
        trace_record(c, 1, &h, NULL, r);
        v3 emmited = v3_ld(sp->emitted), multiplied = h.col_mod;
        rust2_color_override(c->sc, &h, &multiplied, &emmited);
        L = v3_add(L, v3_mul(emmited, thr));
        thr = v3_mul(thr, multiplied);
        r = next;
    }
    return v3_add(L, v3_mul(v3_ld(c->sc->background), thr));        /* depth == 0 returns the background */
}

static v3 ray_color(ctx_t *c, ray_t r) {
    const RtwParams *p = c->p;
    int rec = (p->flags & RTW_FLAG_RECURSIVE_ORDER) != 0;
    switch (p->integrator) {
    case RTW_INTEGRATOR_GRADIENT: return rec ? ray_color_gradient_rec(c, r, p->depth) : ray_color_gradient_iter(c, r, p->depth);
    case RTW_INTEGRATOR_BG_COLOR: return rec ? ray_color_bg_rec(c, r, p->depth) : ray_color_bg_iter(c, r, p->depth);
    case RTW_INTEGRATOR_NORMAL:   return ray_color_normal(c, r);
    case RTW_INTEGRATOR_RUST2:    return rec ? ray_color_rust2_rec(c, r, p->depth) : ray_color_rust2_iter(c, r, p->depth);
    default:                      return ray_color_flag(c, r, p->depth);
    }
}

/* viewport.rs:207-213 */
static inline v3 gamma_correct(v3 c, float g) { return v3_make(powf(c.x, g), powf(c.y, g), powf(c.z, g)); }

/* How many (pixel, sample) rays a sampler traces for `samples`. */
static uint32_t sampler_count(uint32_t sampler, uint32_t samples, uint32_t *root) {
    uint32_t s_root = 0, n = samples;
    if (sampler == RTW_SAMPLER_STRATIFIED) { s_root = (uint32_t)ceilf(sqrtf((float)samples)); n = s_root * s_root; }   /* viewport.rs:443 */
    else if (sampler == RTW_SAMPLER_CENTRES) { s_root = (uint32_t)floorf(sqrtf((float)samples)); n = s_root * s_root; } /* Rust2 viewport.rs:90 (f64 sqrt of an exact integer: same floor for samples < 2^24) */
    else if (sampler == RTW_SAMPLER_NO_RAND) n = 1;
    if (root) *root = s_root;
    return n;
}

static void render_pixel(const RtwCamera *cam, const RtwScene *sc, const RtwParams *p,
                         uint32_t i, uint32_t j, float out[3], counters_t *cn, uint64_t *camera_rays) {
    v3 origin = v3_ld(cam->origin), cu = v3_ld(cam->u), cv = v3_ld(cam->v);
    v3 p00 = v3_ld(cam->pixel00), du = v3_ld(cam->delta_u), dv = v3_ld(cam->delta_v);
    float inv_g = 1.0f / p->gamma;                                  /* viewport.rs:232,439 */
    uint32_t s_root, n = sampler_count(p->sampler, p->samples, &s_root);
    uint32_t pixel = j * p->width + i;
    v3 color = v3_make(0, 0, 0), part = v3_make(0, 0, 0);
    ctx_t c; memset(&c, 0, sizeof c); c.sc = sc; c.p = p; c.cn = cn;

    if (p->sampler == RTW_SAMPLER_NO_RAND) {                        /* viewport.rs:498-508 */
        rng_t rng = rng_seed(p->seed, pixel, 0); rng.permuted = (p->flags & RTW_ORACLE_FLAG_PERMUTED_STREAM) != 0; c.rng = &rng;
        ray_t r; r.origin = origin; r.time = 0.0f;
        r.dir = v3_add(v3_add(p00, v3_scale(du, (float)i)), v3_scale(dv, (float)j));
        v3 col = gamma_correct(ray_color(&c, r), inv_g);
        out[0] = col.x; out[1] = col.y; out[2] = col.z;
        *camera_rays += 1;
        return;
    }
    for (uint32_t s = 0; s < n; s++) {
        rng_t rng = rng_seed(p->seed, pixel, s); rng.permuted = (p->flags & RTW_ORACLE_FLAG_PERMUTED_STREAM) != 0; c.rng = &rng;
        ray_t r;
        if (p->sampler == RTW_SAMPLER_ROW) {                        /* viewport.rs:287-299 */
            v3 rp = random_in_unit_disk_s(&rng, (p->flags & RTW_FLAG_CPP_DIFFUSE) != 0);
            r.origin = v3_add(origin, v3_scale(v3_add(v3_scale(cu, rp.x), v3_scale(cv, rp.y)), cam->lens_radius));
            float jx = (float)i + rng_f32(&rng);
            float jy = (float)j + rng_f32(&rng);
            r.dir = v3_add(v3_add(p00, v3_scale(du, jx)), v3_scale(dv, jy));
            r.time = cam->time0 + cam->shutter * rng_f32(&rng);
        } else if (p->sampler == RTW_SAMPLER_STRATIFIED) {          /* viewport.rs:452-470: x outer, y inner */
            uint32_t x = s / s_root, y = s % s_root;
            v3 rp = random_in_unit_disk_s(&rng, (p->flags & RTW_FLAG_CPP_DIFFUSE) != 0);
            r.origin = v3_add(origin, v3_scale(v3_add(v3_scale(cu, rp.x), v3_scale(cv, rp.y)), cam->lens_radius));
            float jx = (float)i + (((float)x + rng_f32(&rng)) / (float)s_root);
            float jy = (float)j + (((float)y + rng_f32(&rng)) / (float)s_root);
            r.dir = v3_add(v3_add(p00, v3_scale(du, jx)), v3_scale(dv, jy));
            r.time = 0.0f;                                          /* Ray::new */
        } else {                                                    /* Rust2/src/viewport.rs:92-104: k outer (x), l inner (y) */
            uint32_t k = s / s_root, l = s % s_root;
            v3 rp = random_in_unit_disk_s(&rng, (p->flags & RTW_FLAG_CPP_DIFFUSE) != 0);
            /* Rust2 offsets the origin by the raw disk point * lens_radius (not in the u,v basis) */
            r.origin = v3_add(origin, v3_scale(rp, cam->lens_radius));
            /* here pixel00/delta_u/delta_v hold Rust2's left_top and FULL-viewport delta_x/delta_y
             * (Rust2/src/viewport/camera.rs:36-50), divided by width/height at use */
            float jx = ((float)i + ((float)k + 0.5f) / (float)s_root) / (float)p->width;
            float jy = ((float)j + ((float)l + 0.5f) / (float)s_root) / (float)p->height;
            r.dir = v3_add(v3_add(p00, v3_scale(du, jx)), v3_scale(dv, jy));
            r.time = 0.0f;
        }
        if (p->flags & RTW_FLAG_CHUNK_SUMS) {
            /* device-mode restated (rtw.h): the samples of a chunk of RTW_SUM_CHUNK are added left to right into a partial sum,
             * the partial sums are added in chunk order */
            v3 L = ray_color(&c, r);
            part = (s % RTW_SUM_CHUNK) == 0 ? L : v3_add(part, L);
            if ((s % RTW_SUM_CHUNK) == RTW_SUM_CHUNK - 1 || s + 1 == n) color = v3_add(color, part);
        } else
        color = v3_add(color, ray_color(&c, r));                    /* :299 color += */
    }
    *camera_rays += n;
    v3 col = gamma_correct(v3_div(color, (float)n), inv_g);         /* :301 / :472 */
    out[0] = col.x; out[1] = col.y; out[2] = col.z;
}

/* ------------------------------------------------------------------------------------------------
 * Driver: one task per row (tokio::spawn(render_row(..)) viewport.rs:236-240; rayon
 * into_par_iter over rows Rust2/src/viewport.rs:119-122).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const RtwCamera *cam; const RtwScene *sc; const RtwParams *p;
    float *out; const uint32_t *rows; uint32_t n_rows;
    atomic_uint next;
    pthread_mutex_t mu;
    uint64_t camera_rays, segments, sphere_tests, quad_tests; uint32_t nan_pixels;
} job_t;

static void *worker(void *arg) {
    job_t *jb = (job_t *)arg;
    counters_t cn = { 0, 0, 0 }; uint64_t rays = 0; uint32_t nans = 0;
    for (;;) {
        uint32_t k = atomic_fetch_add(&jb->next, 1);
        if (k >= jb->n_rows) break;
        uint32_t j = jb->rows[k];
        float *row = jb->out + (size_t)k * jb->p->width * 3;
        for (uint32_t i = 0; i < jb->p->width; i++) {
            render_pixel(jb->cam, jb->sc, jb->p, i, j, row + 3 * (size_t)i, &cn, &rays);
            if (isnan(row[3 * i]) || isnan(row[3 * i + 1]) || isnan(row[3 * i + 2])) nans++;
        }
    }
    pthread_mutex_lock(&jb->mu);
    jb->camera_rays += rays; jb->segments += cn.segments; jb->sphere_tests += cn.sphere_tests; jb->quad_tests += cn.quad_tests; jb->nan_pixels += nans;
    pthread_mutex_unlock(&jb->mu);
    return NULL;
}

static int params_ok(const RtwCamera *cam, const RtwScene *sc, const RtwParams *p) {
    if (!cam || !sc || !p) return 0;
    if (p->width == 0 || p->height == 0 || p->samples == 0) return 0;
    if (sc->n_spheres && !sc->spheres) return 0;
    if ((sc->n_quads && !sc->quads) || (sc->n_instances && !sc->instances)) return 0;
    if ((sc->n_inst_spheres && !sc->inst_spheres) || (sc->n_inst_quads && !sc->inst_quads)) return 0;
    for (uint32_t i = 0; i < sc->n_instances; i++) {
        const RtwInstance *in = &sc->instances[i];
        if ((uint64_t)in->first_sphere + in->n_spheres > sc->n_inst_spheres) return 0;
        if ((uint64_t)in->first_quad + in->n_quads > sc->n_inst_quads) return 0;
    }
    if (p->integrator > RTW_INTEGRATOR_RUST2 || p->sampler > RTW_SAMPLER_NO_RAND) return 0;
    if (p->part_count > 1 && (p->row_block == 0 || p->part_index >= p->part_count)) return 0;
    return 1;
}

int rtw_oracle_render(const RtwCamera *cam, const RtwScene *sc, const RtwParams *p,
                      float *out_rgb, RtwStats *stats, int threads) {
    if (!params_ok(cam, sc, p) || !out_rgb) return RTW_E_INVALID;
    struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
    uint32_t *rows = (uint32_t *)malloc(sizeof(uint32_t) * p->height);
    if (!rows) return RTW_E_NOMEM;
    uint32_t n_rows = 0;
    for (uint32_t r = 0; r < p->height; r++)
        if (p->part_count <= 1 || (r / p->row_block) % p->part_count == p->part_index) rows[n_rows++] = r;

    job_t jb; memset(&jb, 0, sizeof jb);
    jb.cam = cam; jb.sc = sc; jb.p = p; jb.out = out_rgb; jb.rows = rows; jb.n_rows = n_rows;
    atomic_init(&jb.next, 0); pthread_mutex_init(&jb.mu, NULL);
    if (threads <= 1) worker(&jb);
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
        if (!th) { free(rows); return RTW_E_NOMEM; }
        int started = 0;
        for (int k = 0; k < threads; k++) if (pthread_create(&th[k], NULL, worker, &jb) == 0) started++; else break;
        if (started == 0) worker(&jb);
        for (int k = 0; k < started; k++) pthread_join(th[k], NULL);
        free(th);
    }
    pthread_mutex_destroy(&jb.mu);
    free(rows);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->camera_rays = jb.camera_rays; stats->segments = jb.segments; stats->sphere_tests = jb.sphere_tests; stats->quad_tests = jb.quad_tests;
        stats->nan_pixels = jb.nan_pixels; stats->rows = n_rows;
        stats->total_ms = (float)((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6);
        stats->kernel_ms = stats->total_ms;
    }
    return RTW_OK;
}

int rtw_oracle_trace_ray(const float origin[3], const float dir[3], float time,
                         const RtwScene *sc, const RtwParams *p, uint32_t pixel, uint32_t sample,
                         RtwOracleBounce *out, int cap, float rgb[3]) {
    if (!origin || !dir || !sc || !p) return RTW_E_INVALID;
    counters_t cn = { 0, 0, 0 };
    rng_t rng = rng_seed(p->seed, pixel, sample); rng.permuted = (p->flags & RTW_ORACLE_FLAG_PERMUTED_STREAM) != 0;
    ctx_t c; memset(&c, 0, sizeof c); c.sc = sc; c.p = p; c.cn = &cn; c.rng = &rng;
    c.trace = out; c.trace_cap = cap;
    ray_t r; r.origin = v3_ld(origin); r.dir = v3_ld(dir); r.time = time;
    v3 col = ray_color(&c, r);
    if (rgb) { rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z; }
    return c.trace_n;
}

/* ------------------------------------------------------------------------------------------------
 * Viewport::new (viewport.rs:308-401)
 * ---------------------------------------------------------------------------------------------- */
int rtw_oracle_viewport_new(uint32_t width, float aspect_ratio, const float *vfov, const float *origin,
                            const float *direction, const float *vup, const float *lens_radius,
                            RtwCamera *cam, uint32_t *height_out) {
    if (!cam || width == 0) return RTW_E_INVALID;
    v3 c_origin = origin ? v3_ld(origin) : v3_make(0.0f, 0.0f, 0.0f);
    v3 c_dir = direction ? v3_ld(direction) : v3_make(0.0f, 0.0f, -1.0f);
    v3 c_vup = vup ? v3_ld(vup) : v3_make(0.0f, 1.0f, 0.0f);
    float c_vfov = vfov ? *vfov : 90.0f;

    v3 w = v3_neg(c_dir);                           /* not normalised (:342) */
    v3 u = v3_unit(v3_cross(c_vup, w));
    v3 v = v3_cross(w, u);

    float hf = (float)width / aspect_ratio;         /* (width as f32 / aspect_ratio) as u64 (:346) */
    uint32_t height = f32_as_usize(hf);

    float h = tanf(c_vfov * 3.14159265358979323846f / 360.0f);   /* :348 */
    float viewport_height = 2.0f * h;
    float viewport_width = aspect_ratio * viewport_height;

    v3 viewport_u = v3_scale(u, viewport_width);
    v3 viewport_v = v3_scale(v3_neg(v), viewport_height);

    v3 pixel_delta_u = v3_div(viewport_u, (float)width);
    v3 pixel_delta_v = v3_div(viewport_v, (float)height);

    /* -w - viewport_u / 2.0 - viewport_v / 2.0 (:359) */
    v3 upper_left = v3_sub(v3_sub(v3_neg(w), v3_div(viewport_u, 2.0f)), v3_div(viewport_v, 2.0f));
    v3 pixel00 = v3_add(upper_left, v3_scale(v3_add(pixel_delta_u, pixel_delta_v), 0.5f));

    memset(cam, 0, sizeof *cam);
    cam->origin[0] = c_origin.x; cam->origin[1] = c_origin.y; cam->origin[2] = c_origin.z;
    cam->u[0] = u.x; cam->u[1] = u.y; cam->u[2] = u.z;
    cam->v[0] = v.x; cam->v[1] = v.y; cam->v[2] = v.z;
    cam->pixel00[0] = pixel00.x; cam->pixel00[1] = pixel00.y; cam->pixel00[2] = pixel00.z;
    cam->delta_u[0] = pixel_delta_u.x; cam->delta_u[1] = pixel_delta_u.y; cam->delta_u[2] = pixel_delta_u.z;
    cam->delta_v[0] = pixel_delta_v.x; cam->delta_v[1] = pixel_delta_v.y; cam->delta_v[2] = pixel_delta_v.z;
    cam->lens_radius = lens_radius ? *lens_radius : 0.0f;
    cam->time0 = 0.0f / 30.0f;                      /* frame 0 / fps 30 (:395-399) */
    cam->shutter = 0.0f;
    if (height_out) *height_out = height;
    return RTW_OK;
}
