/*
 * rtw.h -- C ABI of the MI355X-native path tracer (librtw_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of Terence-23/RayTracing-in-a-weekend:
 *
 *     Viewport::render / render_row  ->  ray_color_*  ->  Sphere::collision_normal  ->  Material::on_hit
 *
 * The reference has no FFI: its seam is the function-typed `ray_color` parameter of
 *   Rust/src/viewport.rs:430      Viewport::render(&self, ray_color: &dyn Fn(Ray,&Scene,usize)->Rgb<f32>, scene)
 *   Rust/src/viewport.rs:215-219  async_render(viewport: Box<Viewport>, ray_color, scene: Box<Scene>) -> Img
 *   Rust2/src/viewport.rs:116,128 Viewport::render_rows_async(self) / render(self)
 *   C++/headers/viewport.h:95     Img Viewport::Render(RGB_float (*ray_color)(...), const Scene&)
 * A host closure cannot cross to the GPU, so the integrator is selected by enum and the scene is
 * passed as flat PODs.  Everything here is plain C: pointers, sizes, PODs; no torch, no C++ types.
 *
 * Conventions
 *   - caller owns every buffer; the library allocates only device scratch inside an rtw_ctx
 *   - no exceptions / panics cross the ABI: 0 == RTW_OK, negative == RTW_E_*
 *   - rtw_ctx_render() is blocking (mirrors `rt.block_on(render_multi(..))`, Rust/src/main.rs:76-78)
 *   - one rtw_ctx == one GPU == one HIP stream; contexts are independent (one per process/rank)
 *   - all arithmetic is f32 (Rust/src/vec3.rs:11-15)
 */
#ifndef RTW_H
#define RTW_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTW_ABI_VERSION 4

/* ---- status codes ------------------------------------------------------------------------- */
#define RTW_OK              0
#define RTW_E_INVALID      -1   /* bad argument (NULL pointer, zero size, unknown enum value)        */
#define RTW_E_NO_DEVICE    -2   /* no HIP device / device index out of range                         */
#define RTW_E_HIP          -3   /* a HIP runtime call failed (rtw_last_hip_error() has the code)     */
#define RTW_E_NOMEM        -4   /* host or device allocation failed                                  */
#define RTW_E_UNSUPPORTED  -5   /* valid enum value that this build does not implement on the device */
#define RTW_E_NO_SCENE     -6   /* rtw_ctx_render() before rtw_ctx_set_scene()                       */
#define RTW_E_INTERNAL     -7   /* a render kernel gave up (safety valve of its persistent loop): the image is incomplete */
#define RTW_E_RUNTIME_CONFLICT -8 /* rtw_ctx_create: two copies of the HIP runtime are loaded in this process (PyTorch imported after the
                                     first rtw_* call): import torch first, see INTEGRATION.md.  v4 */

/* ---- integrators: which `ray_color` closure the reference would have passed ----------------- */
enum {
    RTW_INTEGRATOR_GRADIENT = 0, /* ray_color_gradient, Rust/src/viewport/ray_color.rs:12-41 (== ray_color_d main.rs:17-46) */
    RTW_INTEGRATOR_BG_COLOR = 1, /* ray_color_bg_color, Rust/src/viewport/ray_color.rs:43-92 (emission + background)        */
    RTW_INTEGRATOR_NORMAL   = 2, /* normal shading of the closest hit, C++/src/tests.cpp:76-97 (ray_colorSc)                */
    RTW_INTEGRATOR_FLAG     = 3, /* RNG-free yellow/blue test integrator, Rust/src/viewport/glass_tests.rs:8-54             */
    RTW_INTEGRATOR_RUST2    = 4  /* Rust2 `ray_color` (Rust2/src/viewport/ray_color.rs:12-37): emmited + next * multiplied,
                                    depth 0 and misses return the background, with Rust2's Material trait objects
                                    (Rust2/src/objects/material.rs): opacity > 0 -> MirrorGlass{ir}, metallicness == 1 ->
                                    Mirror (reflects the UN-normalised direction), else Lambertian (unit(n + rand));
                                    ColorResult{emmited, multiplied} = {emitted, tex * col_mod}                             */
};

/* ---- samplers: which driver loop generates the camera rays ---------------------------------- */
enum {
    RTW_SAMPLER_ROW        = 0, /* render_row: exactly `samples` unstratified, shutter time; Rust/src/viewport.rs:270-305 */
    RTW_SAMPLER_STRATIFIED = 1, /* Viewport::render: ceil(sqrt(samples))^2 strata, time 0; Rust/src/viewport.rs:430-478   */
    RTW_SAMPLER_CENTRES    = 2, /* Rust2 render_row: floor(sqrt(samples))^2 fixed centres; Rust2/src/viewport.rs:87-114   */
    RTW_SAMPLER_NO_RAND    = 3  /* render_no_rand: one un-jittered ray per pixel; Rust/src/viewport.rs:479-516            */
};

/* ---- closest-hit strategy (result-invariant) ------------------------------------------------ */
enum {
    RTW_ACCEL_BRUTE = 0, /* every sphere, list order (the reference's test integrators: camera_tests.rs:19-33)   */
    RTW_ACCEL_BVH   = 1  /* device BVH, time-expanded bounds (Scene::collision_normal -> AABB, viewport.rs:136-150) */
};

/* ---- PODs ----------------------------------------------------------------------------------- */

/* The render-time fields of `Viewport` (Rust/src/viewport.rs:49-77), produced by
 * rtw_viewport_new() == Viewport::new (viewport.rs:308-401).  `pixel00` is
 * `upper_left_corner`: a DIRECTION (the origin is not added, viewport.rs:377-378). */
typedef struct RtwCamera {
    float origin[3];
    float u[3];
    float v[3];
    float pixel00[3];
    float delta_u[3];     /* p_delta_u */
    float delta_v[3];     /* p_delta_v */
    float lens_radius;
    float time0;          /* frame as f32 / fps            (viewport.rs:279) */
    float shutter;        /* shutter_speed                 (viewport.rs:297) */
} RtwCamera;

/* `Sphere` (Rust/src/objects/sphere.rs:13-20) with its `Material` (materials.rs:15-20) inlined.
 * `tex < 0` means the sphere carries the 1x1 ImageTexture::from_color(tex_color) that
 * Sphere::new builds (sphere.rs:151-173); the albedo the hit reports is texel * col_mod
 * (sphere.rs:145), so Sphere::new(c) yields c*c -- that quirk is reproduced by passing
 * tex_color == col_mod == c. */
typedef struct RtwSphere {
    float center[3];      /* origin                                   */
    float radius;
    float velocity[3];    /* centre(t) = origin + velocity * ray.time  (sphere.rs:100) */
    float col_mod[3];
    float tex_color[3];
    float metallicness;
    float opacity;        /* > 0 selects the dielectric branch (materials.rs:106) */
    float ir;
    float emitted[3];     /* `emmited` */
    int32_t tex;          /* index into RtwScene.textures, or -1 */
} RtwSphere;

/* `ImageTexture{img,row,col}` (Rust/src/texture.rs:21-27): row == width, col == height,
 * texel (x,y) lives at texels[3*(texel_offset + y*row + x)] (texture.rs:265). Perlin noise is
 * out of scope. */
typedef struct RtwTexture {
    uint32_t row;
    uint32_t col;
    uint32_t texel_offset;
    uint32_t emit_tex;    /* RTW_INTEGRATOR_RUST2 only: 1 + index of the texture that holds Rust2's `emmit_img` for this image
                             (Rust2/src/objects/texture.rs:34-41), 0 = none (the sphere's `emitted` is used).  Was `reserved` (0) up to v3. */
} RtwTexture;

/* `Quad` (Rust/src/objects/quad.rs:8-20) with its `Material` inlined.  The derived fields of Quad::new
 * (normal = unit(u x v), d = normal . origin, w = n / n.n; quad.rs:96-108) are recomputed by the library.
 * A quad has no col_mod: the hit's albedo is the texel alone (quad.rs:64-79) -- tex < 0 means the 1x1
 * ImageTexture::from_color(tex_color), else texel (floor(alfa*row), floor(beta*col)) of textures[tex].
 * `velocity` is carried by the reference's struct but not used by its hit test (quad.rs:37-81). */
typedef struct RtwQuad {
    float origin[3];
    float u[3];
    float v[3];
    float velocity[3];
    float tex_color[3];
    float metallicness;
    float opacity;
    float ir;
    float emitted[3];
    int32_t tex;
} RtwQuad;

/* `Instance` (Rust/src/objects/instance.rs:27-38): a group of spheres and quads with a translation and an
 * Euler rotation (Vec3::rotated, vec3.rs:161-181 -- restated as written, including its non-orthogonal
 * terms for rotations about more than one axis), hit in local coordinates (instance.rs:250-310).
 * medium != 0 selects `dist_fn = const_density` (instance.rs:24-26): the hit becomes a scattering event
 * at distance ln(xi) / -density behind the surface with a random normal (constant-density smoke). */
enum { RTW_MEDIUM_SURFACE = 0, RTW_MEDIUM_CONST_DENSITY = 1 };
typedef struct RtwInstance {
    uint32_t first_sphere, n_spheres;   /* members: RtwScene.inst_spheres[first_sphere .. +n_spheres) */
    uint32_t first_quad, n_quads;       /*          RtwScene.inst_quads[first_quad .. +n_quads)       */
    float translation[3];
    float rotation[3];
    float density;
    uint32_t medium;                    /* RTW_MEDIUM_* */
} RtwInstance;

/* `Scene` (Rust/src/viewport.rs:79-151): spheres, quads, instances, background colour.  Sphere-only callers
 * (Scene::new_sphere) leave everything after `background` zero. */
typedef struct RtwScene {
    const RtwSphere  *spheres;
    const RtwTexture *textures;   /* may be NULL when n_textures == 0 */
    const float      *texels;     /* [n_texels][3], may be NULL       */
    uint32_t n_spheres;
    uint32_t n_textures;
    uint32_t n_texels;
    float    background[3];       /* Scene.background_color (BG_COLOR / RUST2 integrators) */
    const RtwQuad     *quads;         /* Scene.quads                                       */
    const RtwInstance *instances;     /* Scene.instances                                   */
    const RtwSphere   *inst_spheres;  /* member pools of the instances                     */
    const RtwQuad     *inst_quads;
    uint32_t n_quads;
    uint32_t n_instances;
    uint32_t n_inst_spheres;
    uint32_t n_inst_quads;
} RtwScene;

typedef struct RtwParams {
    uint32_t width, height;       /* full image; each at most 65535 (RTW_E_INVALID beyond) */
    uint32_t samples;             /* Viewport.samples (see sampler for the count actually traced) */
    uint32_t depth;               /* Viewport.depth: max closest-hit queries per camera ray       */
    float    gamma;               /* output = powf(mean, 1/gamma) (viewport.rs:207-213)           */
    float    mint, maxt;          /* 0.001 / 1e5 in ray_color_gradient, 1e4 bg_color, 1e3 ray_color_d */
    uint32_t integrator;          /* RTW_INTEGRATOR_* */
    uint32_t sampler;             /* RTW_SAMPLER_*    */
    uint32_t accel;               /* RTW_ACCEL_*      */
    uint32_t flags;               /* RTW_FLAG_*       */
    uint64_t seed;                /* render seed of the counter-based RNG (DESIGN.md "RNG") */
    /* Row partition for multi-GPU: this call renders the rows r with
     *   (r / row_block) % part_count == part_index,
     * written compactly in increasing r to out_rgb[rows][width][3].
     * part_count <= 1 renders every row. */
    uint32_t row_block;
    uint32_t part_index;
    uint32_t part_count;
    uint32_t reserved;
} RtwParams;

#define RTW_FLAG_NONE            0u
#define RTW_FLAG_RECURSIVE_ORDER 1u  /* oracle only: multiply col_mod in the reference's recursion order */
#define RTW_FLAG_CPP_DIELECTRIC  2u  /* the C++ twin's deterministic dielectric (Schlick term commented out,
                                        C++/headers/materials.h:106): refract whenever possible, no draw */
#define RTW_FLAG_GLOBAL_NODES    4u  /* device only: keep the BVH nodes in global memory (f32, 64 B) even when the
                                        f16 LDS-resident copy is available -- for A/B measurements and tests */
#define RTW_FLAG_CPP_DIFFUSE     8u  /* the C++ twin's non-dielectric branch (C++/headers/materials.h:113-117,
                                        C++/src/materials.cpp:4-13, C++/src/vec3.cpp:28-39, C++/src/sphere.cpp:29-31), in f32:
                                        rejection accepts |p|^2 < 1 (not <= 1); the diffuse direction is
                                        unit((point + normal + rand_unit) - point); the mirror direction is normalised again,
                                        unit(reflect(unit(d), n)); a near-zero (1e-8) result becomes the normal.
                                        RTW_FLAG_CPP_DIELECTRIC | RTW_FLAG_CPP_DIFFUSE is what Viewport::RenderGPU of the C++
                                        tree asks for. */
#define RTW_FLAG_CHUNK_SUMS     16u  /* keep ONE partial sum per pixel and RTW_SUM_CHUNK consecutive samples in the device's sample bank instead
                                        of every sample: the samples of a chunk are added left to right, the chunks' sums in chunk order,
                                        ((s0+s1+s2+s3) + (s4+..)) + ..  Deterministic and independent of the GPU split like the default, but
                                        not the reference's left-to-right association (viewport.rs:299): the image differs by f32 rounding
                                        only (<= 2e-6 relative, far inside BASELINE.json's 1e-3).  The bank shrinks 4x (12.4 GB -> 3.1 GB at
                                        1920x1080x500).  The oracle implements the same association under the same flag. */
#define RTW_SUM_CHUNK            4u

typedef struct RtwStats {
    uint64_t camera_rays;    /* (pixel, sample) primary rays traced                   */
    uint64_t segments;       /* closest-hit queries == BASELINE "rays x bounces"      */
    uint64_t sphere_tests;   /* exact ray/sphere quadratic evaluations                */
    uint64_t node_tests;     /* BVH node slab tests (0 for RTW_ACCEL_BRUTE)           */
    uint32_t nan_pixels;     /* output pixels with a NaN channel                      */
    uint32_t rows;           /* rows written by this call                             */
    float    kernel_ms;      /* device time of the render kernels (hipEvent)          */
    float    total_ms;       /* host wall time of the call                            */
    /* BVH kernel scheduler census: wave-level steps executed per phase (0 traverse, 1 leaf, 2 shade; 3..5 reserved for
     * experimental phases) and the lanes that were live in them; lanes / (64 * steps) is the SIMD efficiency of a phase. */
    uint64_t phase_steps[6];
    uint64_t phase_lanes[6];
    uint64_t quad_tests;     /* ray/quad plane tests (top-level quads and instance members) */
    /* v4: the call's timeline.  enqueue_ms: host time from the begin of the call (rtw_mgpu_render: of the whole call, the same instant for
     * every device) until this context's kernels had been issued; start_ms: device time from a marker recorded on this context's stream at
     * the begin of the call (before any device was given work) to the start of its first kernel.  A fork that is asynchronous shows
     * enqueue_ms far below kernel_ms for EVERY device, and on distinct GPUs start_ms near zero for every device. */
    float    enqueue_ms;
    float    start_ms;
} RtwStats;

typedef struct rtw_ctx rtw_ctx;

/* ---- device path (librtw_hip.so) ------------------------------------------------------------ */

int         rtw_abi_version(void);
int         rtw_device_count(void);
const char *rtw_strerror(int status);
int         rtw_last_hip_error(void);
int         rtw_hip_runtime_count(void);   /* copies of libamdhip64 mapped into this process (1 is healthy; v4) */

/* One context per GPU.  `device` is the HIP device ordinal. */
int  rtw_ctx_create(int device, rtw_ctx **out);
void rtw_ctx_destroy(rtw_ctx *ctx);
/* Launch on this HIP stream (a hipStream_t passed as void*), NULL = the context's own stream. */
int  rtw_ctx_set_stream(rtw_ctx *ctx, void *hip_stream);
/* == Scene::new_sphere(spheres) (viewport.rs:90-105): copies the scene to the GPU and builds the
 * acceleration structure.  [t_begin, t_end] is the ray.time range the bounds must cover
 * (time0 .. time0 + shutter); pass 0,0 for static scenes.  The range is remembered: a later render of a
 * MOVING scene whose [time0, time0 + shutter] is not inside it walks the list instead of the tree (same image,
 * slower) rather than pruning with bounds that do not cover the spheres. */
int  rtw_ctx_set_scene(rtw_ctx *ctx, const RtwScene *scene, float t_begin, float t_end);
/* == Viewport::render(ray_color, scene) -> Img.  out_rgb is [rows][width][3] f32, gamma-corrected,
 * unclamped (viewport.rs:301); it may be host memory or device memory of ctx's GPU. */
int  rtw_ctx_render(rtw_ctx *ctx, const RtwCamera *cam, const RtwParams *params,
                    float *out_rgb, RtwStats *stats);
/* == render_multi(viewport, ray_color, scene) -> Vec<Img> (Rust/src/viewport.rs:249-269): frames start_frame .. start_frame +
 * n_frames, frame i rendered like async_render with time0 = i as f32 / fps (viewport.rs:279; cam->time0 is ignored, cam->shutter
 * is the shutter_speed).  out_rgb holds n_frames images of [rows][width][3] f32 back to back (host or device memory); stats, if not
 * NULL, is an array of n_frames.  The scene must have been set for the whole clip:
 * rtw_ctx_set_scene(ctx, scene, start_frame / fps, (start_frame + n_frames - 1) / fps + shutter). */
int  rtw_ctx_render_multi(rtw_ctx *ctx, const RtwCamera *cam, const RtwParams *params, float fps,
                          uint32_t start_frame, uint32_t n_frames, float *out_rgb, RtwStats *stats);
/* One-shot convenience: create ctx on the current device, set scene, render, destroy. */
int  rtw_render(const RtwCamera *cam, const RtwScene *scene, const RtwParams *params,
                float *out_rgb, RtwStats *stats);

/* Tuning knobs of a context (they were process environment variables up to ABI v2).  None of them changes the image. */
enum {
    RTW_OPT_CHUNK_LEN        = 1, /* samples per work unit, 1..255; 0 = chosen from the size of the launch (default)           */
    RTW_OPT_SAMPLE_BANK_GB   = 2, /* budget of the per-sample radiance bank in GiB (default 48); larger frames are
                                     rendered in bands of tile rows, a budget below one tile row fails with RTW_E_NOMEM  */
    RTW_OPT_LDS_GEOM         = 3, /* sphere {centre, r^2} in LDS next to the f16 nodes: -1 auto (default), 0 off, 1 on   */
    RTW_OPT_BLOCKS_PER_CU    = 4, /* resident workgroups per CU of the persistent grid: 0 auto (default), 1..8           */
    RTW_OPT_LIST_WALK_MAX    = 5, /* RTW_ACCEL_BVH requests for scenes with at most this many spheres walk the list
                                     instead (result-invariant; the traversal scheduler only costs there).  Default:
                                     the measured crossover; 0 = always use the tree                     */
    RTW_OPT_TILE_ORDER       = 6, /* order in which the 8x8 tiles enter the work queue: 0 (default) raster; 1 groups of 8
                                     tiles scattered over the frame; 2 expensive tiles first by a cost estimated from each tile's centre
                                     ray (sphere field / ground only / sky, nearer first; raster for scenes with quads or instances);
                                     3 reverse raster; 4 as 2 in 32 coarse steps per class, raster inside a step; 5 raster inside each
                                     class, classes in the order sphere field, bare ground, sky (v4; measured: no gain)                    */
    RTW_OPT_GRAB_BLOCKS      = 7, /* 64-item blocks of the work queue a wave may take with one atomic while plenty of work is left (single
                                     blocks towards the end of a launch): default 2; 1 = always one; 0 = up to one tile's blocks      */
    RTW_OPT_TAIL_UNITS       = 9, /* guided unit length: the last tiles of the work queue are cut into units of ONE sample (k blocks of them per
                                     resident wave, k = this value) and the tiles before them into units of a third of the launch's length;
                                     0 (default) = one unit length for the whole launch.  Measured: no gain.  v4                 */
    RTW_OPT_SUB_QUEUES       = 8  /* 0 (default): the work queue is eight sub-queues with a counter each (a wave starts on the one of its
                                     XCD and helps out on the others when it is empty), a single one for tiny launches; 1: always single */
};
int  rtw_ctx_set_option(rtw_ctx *ctx, uint32_t key, double value);

/* ---- one frame over several GPUs of a node ------------------------------------------------------
 * The reference forks one task per image row and joins them in order (tokio: Rust/src/viewport.rs:236-244; rayon:
 * Rust2/src/viewport.rs:119-122).  Here the rows are dealt to the devices in interleaved blocks of `row_block` rows
 * (RtwParams.row_block, 8 when 0; device k renders the rows r with (r / row_block) % n_devices == k) and every device
 * copies its blocks STRAIGHT INTO their image rows of the caller's frame (one strided 2-D copy per device, no gather
 * buffer, no de-interleave pass).  The counter-based RNG makes the image independent of the split: the result is bit-identical
 * to rtw_ctx_render of the whole frame on one GPU.  One host thread drives all devices; the call blocks until the frame is
 * complete.  The fork is asynchronous by construction: first every device is prepared (arguments, first-use allocations), then
 * every device's kernels are issued, and only then the copies toward the caller's frame -- none of which waits for a GPU: a frame
 * in device memory or in pinned host memory (hipHostMalloc / hipHostRegister) receives the strided copies directly, a frame in
 * ordinary pageable host memory (where a device-to-host copy would return only when it is done) is staged through a pinned buffer
 * per device and finished by the host at the join.  RtwStats.enqueue_ms / start_ms of per_device[] show the timeline.
 * `devices` are HIP ordinals and may repeat (several contexts on one GPU).  NOT YET MEASURED on more than one physical GPU.
 * out_rgb: the full [height][width][3] f32 frame, host memory or device memory of any of the GPUs.
 * params->part_count must be <= 1.  per_device (may be NULL): n_devices RtwStats; total (may be NULL): counters summed,
 * kernel_ms = the slowest device, total_ms = host wall time of the call. */
typedef struct rtw_mgpu rtw_mgpu;
int  rtw_mgpu_create(const int *devices, uint32_t n_devices, rtw_mgpu **out);
void rtw_mgpu_destroy(rtw_mgpu *m);
int  rtw_mgpu_set_scene(rtw_mgpu *m, const RtwScene *scene, float t_begin, float t_end);
int  rtw_mgpu_set_option(rtw_mgpu *m, uint32_t key, double value);
int  rtw_mgpu_render(rtw_mgpu *m, const RtwCamera *cam, const RtwParams *params, float *out_rgb,
                     RtwStats *per_device, RtwStats *total);
/* One-shot: create, set scene for [cam->time0, cam->time0 + cam->shutter], render, destroy. */
int  rtw_render_multi_gpu(const int *devices, uint32_t n_devices, const RtwCamera *cam, const RtwScene *scene,
                          const RtwParams *params, float *out_rgb, RtwStats *per_device);

/* ---- host mirror of the reference constructors (same library, no GPU needed) ---------------- */

/* Viewport::new (viewport.rs:308-401).  Options the reference takes as Option<> are pointers
 * (NULL == None).  Writes the camera and the derived height `(width as f32 / aspect) as u64`. */
int rtw_viewport_new(uint32_t width, float aspect_ratio, const float *vfov, const float *origin,
                     const float *direction, const float *vup, const float *lens_radius,
                     RtwCamera *cam, uint32_t *height);
/* Viewport::new_from_res (viewport.rs:402-428): aspect = width as f32 / height as f32. */
int rtw_viewport_new_from_res(uint32_t width, uint32_t height, const float *vfov, const float *origin,
                              const float *direction, const float *vup, const float *lens_radius,
                              RtwCamera *cam, uint32_t *height_out);
/* Rust2 `Camera::new(aspect, origin, vup, dir, vfov, lens_radius)` (Rust2/src/viewport/camera.rs:19-53) for
 * RTW_SAMPLER_CENTRES: pixel00 = left_top, delta_u/delta_v = the FULL-viewport delta_x/delta_y (divided by
 * width/height at use, Rust2/src/viewport.rs:95-99); the lens offset is the raw disk point (viewport.rs:101). */
int rtw_camera2_new(float aspect, const float origin[3], const float vup[3], const float dir[3], float vfov,
                    float lens_radius, RtwCamera *cam);
/* Sphere::new / new_moving (sphere.rs:151-199): col_mod==NULL -> (1,1,1); mat==NULL -> EMPTY_M. */
int rtw_sphere_new(const float origin[3], float radius, const float *col_mod,
                   const float *mat3 /* metallicness, opacity, ir */, const float *velocity,
                   RtwSphere *out);
/* Sphere::new_with_texture (sphere.rs:200-224). */
int rtw_sphere_new_with_texture(const float origin[3], float radius, const float *col_mod,
                                const float *mat3, const float *velocity, int32_t tex, RtwSphere *out);
/* Quad::new (quad.rs:84-110) with ImageTexture::from_color(color): velocity == NULL -> 0, emitted == NULL -> 0. */
int rtw_quad_new(const float origin[3], const float u[3], const float v[3], const float *mat3,
                 const float *emitted, const float color[3], RtwQuad *out);
/* Instance::new_box(a, b, tex, mat) (instance.rs:83-176): the six quads of the axis-aligned box, in the
 * reference's order, written to quads6[0..6). */
int rtw_box_quads(const float a[3], const float b[3], const float *mat3, const float color[3], RtwQuad quads6[6]);
/* Vec3::rotated(rot) (Rust/src/vec3.rs:161-181) as the library applies it to instances: sin / cos of the three angles on the
 * host, the products in the reference's written order (including its non-orthogonal terms for rotations about more than
 * one axis). */
void rtw_vec3_rotated(const float v[3], const float rot[3], float out[3]);
/* The tile permutation RTW_OPT_TILE_ORDER = mode uses for a whole width x height frame of this camera (host only; for tests and
 * tools): order[q] = index (row-major over the ceil(width/8) x ceil(height/8) tiles) of the tile the queue hands out q-th. */
int rtw_tile_order(uint32_t mode, uint32_t width, uint32_t height, const RtwCamera *cam, const RtwScene *scene,
                   uint32_t *order, uint32_t cap);
/* Rows a partition owns (see RtwParams). */
uint32_t rtw_part_rows(uint32_t height, uint32_t row_block, uint32_t part_index, uint32_t part_count);
/* write_img_f32 quantisation: round(clamp(c*255, 0, 255)) (Rust/src/write_img.rs:11-15). */
void rtw_quantize_u8(const float *rgb, size_t n_values, uint8_t *out);
/* Rust2 Vec3::to_rgb_u8: round(clamp(c*255.99, 0, 255)) (Rust2/src/vec3.rs:240-246). */
void rtw_quantize_u8_rust2(const float *rgb, size_t n_values, uint8_t *out);

/* ---- scene wire format and image writers (rtw_io.cpp; host only) -------------------------------- */
/* The reference's `json` scene object {"spheres":[{origin,radius,col_mod,material{metallicness,opacity,ir},
 * velocity,texture{row,col,img[]}}]} (Rust/src/viewport.rs:174-205, objects/sphere.rs:44-90,
 * objects/materials.rs:21-54, texture.rs:28-59; the C++ dialect without velocity/texture is accepted).
 * rtw_scene_to_json returns the length needed; rtw_scene_from_json uses the count-query pattern
 * (spheres == NULL -> sizes only). */
size_t rtw_scene_to_json(const RtwScene *scene, char *buf, size_t cap);
int rtw_scene_from_json(const char *text, size_t len,
                        RtwSphere *spheres, uint32_t sphere_cap, uint32_t *n_spheres,
                        RtwTexture *textures, uint32_t texture_cap, uint32_t *n_textures,
                        float *texels, uint32_t texel_cap, uint32_t *n_texels);
/* write_img_f32 (Rust/src/write_img.rs:6-19): quantise and save an 8-bit RGB PNG. */
int rtw_write_png_f32(const char *path, const float *rgb, uint32_t width, uint32_t height);
/* write_ppm (C++/src/ppm_writer.cpp:12-27): P3 text with the C++ truncation int(255 c). */
int rtw_write_ppm_f32(const char *path, const float *rgb, uint32_t width, uint32_t height);

/* Host-side self-check of the acceleration structure rtw_ctx_set_scene would build (no GPU): every sphere is
 * reachable exactly once (tree leaf or big list), nested bounds, time-expanded for [t_begin, t_end], the f16 copy
 * contains the f32 boxes, depth within the device stack.  RTW_OK or RTW_E_INVALID; optional outputs describe the tree. */
int rtw_bvh_validate(const RtwScene *scene, float t_begin, float t_end,
                     uint32_t *n_nodes, uint32_t *depth, uint32_t *n_big, uint32_t *has_f16);

/* Scene generators for the BASELINE configs.  Each fills caller arrays; call with
 * spheres == NULL to query the counts.  Returns RTW_OK or RTW_E_INVALID if capacity is too small. */
enum {
    RTW_SCENE_C1_THREE_SPHERES = 1, /* ground + lambert + metal (material_tests.rs:105-167 trimmed) */
    RTW_SCENE_C2_BOOK1_FINAL   = 2, /* Book-1 final random spheres, ~485                          */
    RTW_SCENE_C4_DIELECTRIC    = 4, /* 9x9 hollow-glass grid + fuzzy metal                         */
    RTW_SCENE_C5_MOTION_CHECKER= 5, /* C2 with moving lambert spheres + 4x2 image-textured ground  */
    RTW_SCENE_METAL_TEST       = 6, /* 4-sphere metal_test (material_tests.rs:105-167)             */
    RTW_SCENE_QUAD_TEST        = 7, /* the five quads of quad_test (objects/quad.rs:152-299)       */
    RTW_SCENE_PRESENTATION     = 8, /* presentation_image (main.rs:89-419): sphere + 6 quads (one a light) +
                                       a rotated smoke box + a rotated glass pane, black background  */
    RTW_SCENE_FIRST_FRAME      = 9  /* main()'s seven spheres (main.rs:427-496), the scene of Rust/First frame.png */
};
int rtw_scene_generate(uint32_t which, uint64_t scene_seed,
                       RtwSphere *spheres, uint32_t sphere_cap, uint32_t *n_spheres,
                       RtwTexture *textures, uint32_t texture_cap, uint32_t *n_textures,
                       float *texels, uint32_t texel_cap, uint32_t *n_texels);
/* Same for scenes with quads / instances (7, 8; the sphere-only ids work too).  Any output array may be NULL
 * when only the counts are wanted: counts[0..5) = spheres, quads, instances, inst_spheres, inst_quads
 * (these scenes carry no image textures). */
int rtw_scene_generate_geom(uint32_t which, uint64_t scene_seed, RtwSphere *spheres, RtwQuad *quads,
                            RtwInstance *instances, RtwSphere *inst_spheres, RtwQuad *inst_quads,
                            const uint32_t caps[5], uint32_t counts[5], float background[3]);
/* The camera + params each config is quoted with (width/height/samples/depth/lens/vfov/shutter). */
int rtw_scene_default_view(uint32_t which, RtwCamera *cam, RtwParams *params);

#ifdef __cplusplus
}
#endif
#endif /* RTW_H */
