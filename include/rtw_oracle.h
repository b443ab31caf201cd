/*
 * rtw_oracle.h -- ABI of the CPU oracle (oracle/librtw_oracle.so).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product path (librtw_hip.so) never links,
 * loads or calls it, and has no CPU fallback.
 *
 * The oracle is a plain-C, f32, no-FMA restatement of the reference's hot path
 * (Rust/src/viewport.rs, viewport/ray_color.rs, objects/sphere.rs, objects/materials.rs, vec3.rs);
 * see oracle/rtw_oracle.c for the per-function file:line citations.
 */
#ifndef RTW_ORACLE_H
#define RTW_ORACLE_H

#include "rtw.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Oracle-only, test-only bit of RtwParams.flags: draw through PCG's RXS-M-XS output permutation (the stream of rounds 1 - 2a) instead of taking
 * the top 24 bits of the LCG state directly (the product's stream).  Exists for the image-level comparison of the two streams; the device
 * ignores the bit (it has no such stream). */
#define RTW_ORACLE_FLAG_PERMUTED_STREAM 0x40000000u

/* Same contract as rtw_ctx_render(); `threads` row-parallel workers (one task per row, like
 * tokio::spawn(render_row) viewport.rs:236-240), threads <= 1 runs serially. */
int rtw_oracle_render(const RtwCamera *cam, const RtwScene *scene, const RtwParams *params,
                      float *out_rgb, RtwStats *stats, int threads);

/* Viewport::new restated (viewport.rs:308-401); checked bit-for-bit against rtw_viewport_new(). */
int rtw_oracle_viewport_new(uint32_t width, float aspect_ratio, const float *vfov, const float *origin,
                            const float *direction, const float *vup, const float *lens_radius,
                            RtwCamera *cam, uint32_t *height);

/* One record per closest-hit query of a single path, for vector-level golden tests
 * (Rust/cerr trace). */
typedef struct RtwOracleBounce {
    int32_t hit;             /* 1 hit, 0 miss (sky) */
    int32_t sphere;          /* index of the top-level object hit: spheres, then quads, then instances */
    int32_t front_face;      /* !(dir . normal > 0)  */
    int32_t cannot_refract;  /* dielectric only      */
    float   t;
    float   ratio;           /* refraction_ratio, dielectric only */
    float   normal[3];       /* outward geometric normal */
    float   point[3];
    float   unit_dir[3];     /* unit(ray.direction) of the incoming ray */
    float   next_dir[3];
} RtwOracleBounce;

/* Trace the path of one explicit ray (RNG stream = (seed, pixel, sample)); returns the number of
 * records written (<= cap). */
int rtw_oracle_trace_ray(const float origin[3], const float dir[3], float time,
                         const RtwScene *scene, const RtwParams *params,
                         uint32_t pixel, uint32_t sample,
                         RtwOracleBounce *out, int cap, float rgb[3]);

/* The RNG, exposed so tests can pin it with known-answer vectors. */
void  rtw_oracle_rng_seed(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t state[2] /* state, inc */);
float rtw_oracle_rng_next(uint32_t state[2]);
/* The natural logarithm the constant-density medium uses (instance.rs:24-26 `gen::<f32>().ln()`): f64 series,
 * rounded once; bulk form for the exhaustive check over every xi = k * 2^-24. */
float rtw_oracle_ln(float x);
void  rtw_oracle_ln_bulk(const float *x, float *out, size_t n);

/* Rust2's ImageTexture::color_at index rule (Rust2/src/objects/texture.rs:94-105): x * width + y with x = (u * width) as usize, y = (v * height)
 * as usize (emission != 0: floor() first, as :95-96 write it), clamped to the last texel where the reference would panic. */
uint32_t rtw_oracle_rust2_texel_index(float u, float v, uint32_t width, uint32_t height, int emission);

/* TEST ONLY: the spherical UV of sphere.rs:132-133 for n unit normals, through libm (plain == 0: what the oracle renders with) or through a
 * restatement of the device's lean atan2 / acos sequences (plain != 0).  out: [n][4] = atan2, acos, u, v. */
void  rtw_oracle_sphere_uv(const float *normals, size_t n, int plain, float *out);

/* Vec3::rotated (Rust/src/vec3.rs:161-181), for the reference's rotation_tests known answers (vec3.rs:363-404). */
void  rtw_oracle_rotated(const float v[3], const float rot[3], float out[3]);

#ifdef __cplusplus
}
#endif
#endif
