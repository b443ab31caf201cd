// rtw_host.cpp -- host mirror of the reference's constructors + scene generators + BVH build.
// Built with -ffp-contract=off: every f32 expression below rounds once per written operation,
// like the Rust it restates.
#include "rtw_host.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>

// The ABI's PODs have no implicit padding (the ctypes / Rust mirrors rely on these sizes).
static_assert(sizeof(RtwCamera) == 84 && sizeof(RtwSphere) == 80 && sizeof(RtwTexture) == 16 && sizeof(RtwQuad) == 88, "POD layout");
static_assert(sizeof(RtwInstance) == 48 && sizeof(RtwScene) == 96 && sizeof(RtwParams) == 72 && sizeof(RtwStats) == 160, "POD layout");

namespace rtw {

uint32_t sampler_count(uint32_t sampler, uint32_t samples, uint32_t *s_root) {
    uint32_t root = 0, n = samples;
    if (sampler == RTW_SAMPLER_STRATIFIED) { root = (uint32_t)std::ceil(std::sqrt((float)samples)); n = root * root; }      // viewport.rs:443
    else if (sampler == RTW_SAMPLER_CENTRES) { root = (uint32_t)std::floor(std::sqrt((float)samples)); n = root * root; }   // Rust2 viewport.rs:90
    else if (sampler == RTW_SAMPLER_NO_RAND) n = 1;
    if (s_root) *s_root = root;
    return n;
}

namespace {

struct V { float x, y, z; };
inline V mk(float x, float y, float z) { return V{ x, y, z }; }
inline V ld(const float *p) { return V{ p[0], p[1], p[2] }; }
inline V operator+(V a, V b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V operator-(V a, V b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V operator*(V a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V operator/(V a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
inline V operator-(V a) { return mk(-a.x, -a.y, -a.z); }
inline float length(V a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V unit(V a) { return a / length(a); }
inline V cross(V a, V b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline void st(float *p, V a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

// Rust `f as u64` for the values that occur here (saturating, NaN -> 0)
inline uint32_t f32_as_u(float f) {
    if (!(f > 0.0f)) return 0;
    if (f >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

// ---- scene RNG: PCG32 (XSH-RR 64/32), only used to lay scenes out --------------------------------
struct Pcg32 {
    uint64_t state, inc;
    explicit Pcg32(uint64_t seed, uint64_t seq = 54u) {
        state = 0u; inc = (seq << 1u) | 1u; next(); state += seed; next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t)(old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
    }
    float f() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
    float range(float lo, float hi) { return lo + (hi - lo) * f(); }
};

const float SCATTER_M[3] = { 0.0f, 0.0f, 1.0f };          // materials.rs:168-177
const float METALLIC_M[3] = { 1.0f, 0.0f, 1.0f };         // :157-166
const float FUZZY3_M[3] = { 0.7f, 0.0f, 1.0f };           // :179-188
const float GLASS_M[3] = { 1.0f, 1.0f, 1.5f };            // :190-199
const float GLASSR_M[3] = { 1.0f, 1.0f, 1.0f / 1.5f };    // :201-210

// Sphere::new(origin, r, Some(c), Some(mat)): col_mod AND the 1x1 texture are c (sphere.rs:151-173)
RtwSphere sphere_new(V o, float r, V c, const float *mat) {
    RtwSphere s; std::memset(&s, 0, sizeof s);
    float org[3] = { o.x, o.y, o.z }, col[3] = { c.x, c.y, c.z };
    rtw_sphere_new(org, r, col, mat, nullptr, &s);
    return s;
}
// A sphere whose reported albedo is exactly c: texture = c, col_mod = (1,1,1)
// (== Sphere::new_with_texture(o, r, None, mat, ImageTexture::from_color(c)))
RtwSphere sphere_albedo(V o, float r, V c, const float *mat) {
    RtwSphere s = sphere_new(o, r, mk(1, 1, 1), mat);
    st(s.tex_color, c);
    return s;
}

} // namespace
} // namespace rtw

using namespace rtw;

extern "C" {

// Vec3::rotated (vec3.rs:161-181): the host twin of rtw_device.h rotated() -- same sin / cos (std::sin(float), as rtw_ctx_set_scene
// evaluates them for the device), same products in the same order (this TU is built with -ffp-contract=off).
void rtw_vec3_rotated(const float v[3], const float rot[3], float out[3]) {
    if (!v || !rot || !out) return;
    const float as = std::sin(rot[0]), ac = std::cos(rot[0]), bs = std::sin(rot[1]), bc = std::cos(rot[1]), cs = std::sin(rot[2]), cc = std::cos(rot[2]);
    const float x = v[0], y = v[1], z = v[2];
    out[0] = x * bc * cc + y * (as * bs * cc - as * cc) + z * (ac * bs * cc + as * cs);
    out[1] = x * bc * cs + y * (as * bs * cs + ac * cc) + z * (ac * bs * cs - as * cc);
    out[2] = x * -bs + y * as * bc + z * ac * bc;
}

uint32_t rtw_part_rows(uint32_t height, uint32_t row_block, uint32_t part_index, uint32_t part_count) {
    if (part_count <= 1) return height;
    if (row_block == 0 || part_index >= part_count) return 0;
    uint32_t n = 0;
    for (uint32_t r = 0; r < height; r++) if ((r / row_block) % part_count == part_index) n++;
    return n;
}

void rtw_quantize_u8(const float *rgb, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        float v = rgb[i] * 255.0f;                      // write_img.rs:11-15
        v = v != v ? v : (v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));   // clamp (NaN propagates)
        v = std::round(v);                              // f32::round: half away from zero
        out[i] = v != v ? 0 : (uint8_t)v;               // `as u8`: NaN -> 0
    }
}

void rtw_quantize_u8_rust2(const float *rgb, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) {
        float v = rgb[i] * 255.99f;                     // Rust2/src/vec3.rs:240-246
        v = v != v ? v : (v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));
        v = std::round(v);
        out[i] = v != v ? 0 : (uint8_t)v;
    }
}

// Rust2 Camera::new (Rust2/src/viewport/camera.rs:19-53)
int rtw_camera2_new(float aspect, const float origin[3], const float vup[3], const float dir[3], float vfov,
                    float lens_radius, RtwCamera *cam) {
    if (!cam || !origin || !vup || !dir) return RTW_E_INVALID;
    V w = -ld(dir);
    V u = unit(cross(ld(vup), w));
    V v = cross(w, u);
    float h = std::tan(vfov * 3.14159265358979323846f / 360.0f);
    float viewport_height = 2.0f * h;
    float viewport_width = aspect * viewport_height;
    V viewport_u = u * viewport_width;
    V viewport_v = (-v) * viewport_height;
    std::memset(cam, 0, sizeof *cam);
    st(cam->origin, ld(origin)); st(cam->u, u); st(cam->v, v);
    st(cam->pixel00, ((-w) - viewport_u / 2.0f) - viewport_v / 2.0f);    // left_top
    st(cam->delta_u, viewport_u); st(cam->delta_v, viewport_v);          // delta_x, delta_y
    cam->lens_radius = lens_radius;
    return RTW_OK;
}

// Viewport::new (viewport.rs:308-401)
int rtw_viewport_new(uint32_t width, float aspect_ratio, const float *vfov, const float *origin,
                     const float *direction, const float *vup, const float *lens_radius,
                     RtwCamera *cam, uint32_t *height_out) {
    if (!cam || width == 0) return RTW_E_INVALID;
    V c_origin = origin ? ld(origin) : mk(0.0f, 0.0f, 0.0f);
    V c_dir = direction ? ld(direction) : mk(0.0f, 0.0f, -1.0f);
    V c_vup = vup ? ld(vup) : mk(0.0f, 1.0f, 0.0f);
    float c_vfov = vfov ? *vfov : 90.0f;

    V w = -c_dir;                                          // :342, NOT normalised
    V u = unit(cross(c_vup, w));
    V v = cross(w, u);

    uint32_t height = f32_as_u((float)width / aspect_ratio);   // :346

    float h = std::tan(c_vfov * 3.14159265358979323846f / 360.0f);   // :348
    float viewport_height = 2.0f * h;
    float viewport_width = aspect_ratio * viewport_height;  // focal length is not applied (:349-351)

    V viewport_u = u * viewport_width;
    V viewport_v = (-v) * viewport_height;
    V pixel_delta_u = viewport_u / (float)width;
    V pixel_delta_v = viewport_v / (float)height;
    V viewport_upper_left = ((-w) - viewport_u / 2.0f) - viewport_v / 2.0f;      // :359, a direction
    V pixel00_loc = viewport_upper_left + (pixel_delta_u + pixel_delta_v) * 0.5f;

    std::memset(cam, 0, sizeof *cam);
    st(cam->origin, c_origin); st(cam->u, u); st(cam->v, v);
    st(cam->pixel00, pixel00_loc); st(cam->delta_u, pixel_delta_u); st(cam->delta_v, pixel_delta_v);
    cam->lens_radius = lens_radius ? *lens_radius : 0.0f;
    cam->time0 = 0.0f / 30.0f;      // frame 0, fps 30 (:395-399)
    cam->shutter = 0.0f;
    if (height_out) *height_out = height;
    return RTW_OK;
}

int rtw_viewport_new_from_res(uint32_t width, uint32_t height, const float *vfov, const float *origin,
                              const float *direction, const float *vup, const float *lens_radius,
                              RtwCamera *cam, uint32_t *height_out) {
    if (height == 0) return RTW_E_INVALID;
    return rtw_viewport_new(width, (float)width / (float)height, vfov, origin, direction, vup, lens_radius, cam, height_out);
}

int rtw_sphere_new(const float origin[3], float radius, const float *col_mod, const float *mat3,
                   const float *velocity, RtwSphere *out) {
    if (!origin || !out) return RTW_E_INVALID;
    std::memset(out, 0, sizeof *out);
    const float one[3] = { 1.0f, 1.0f, 1.0f };
    const float *c = col_mod ? col_mod : one;
    const float *m = mat3 ? mat3 : SCATTER_M;           // EMPTY_M == SCATTER_M (materials.rs:212)
    for (int k = 0; k < 3; k++) {
        out->center[k] = origin[k];
        out->col_mod[k] = c[k];
        out->tex_color[k] = c[k];                        // ImageTexture::from_color(col_mod) (sphere.rs:168-171)
        out->velocity[k] = velocity ? velocity[k] : 0.0f;
    }
    out->radius = radius;
    out->metallicness = m[0]; out->opacity = m[1]; out->ir = m[2];
    out->tex = -1;
    return RTW_OK;
}

int rtw_sphere_new_with_texture(const float origin[3], float radius, const float *col_mod, const float *mat3,
                                const float *velocity, int32_t tex, RtwSphere *out) {
    int rc = rtw_sphere_new(origin, radius, col_mod, mat3, velocity, out);
    if (rc != RTW_OK) return rc;
    out->tex = tex;
    out->tex_color[0] = out->tex_color[1] = out->tex_color[2] = 1.0f;
    return RTW_OK;
}

int rtw_quad_new(const float origin[3], const float u[3], const float v[3], const float *mat3,
                 const float *emitted, const float color[3], RtwQuad *out) {
    if (!origin || !u || !v || !color || !out) return RTW_E_INVALID;
    std::memset(out, 0, sizeof *out);
    const float *m = mat3 ? mat3 : SCATTER_M;
    for (int k = 0; k < 3; k++) {
        out->origin[k] = origin[k]; out->u[k] = u[k]; out->v[k] = v[k];
        out->tex_color[k] = color[k];                    // ImageTexture::from_color
        out->emitted[k] = emitted ? emitted[k] : 0.0f;
    }
    out->metallicness = m[0]; out->opacity = m[1]; out->ir = m[2];
    out->tex = -1;
    return RTW_OK;
}

// Instance::new_box (instance.rs:83-176): front, right, back, left, top, bottom
int rtw_box_quads(const float a[3], const float b[3], const float *mat3, const float color[3], RtwQuad q[6]) {
    if (!a || !b || !color || !q) return RTW_E_INVALID;
    const V lo = mk(a[0] <= b[0] ? a[0] : b[0], a[1] <= b[1] ? a[1] : b[1], a[2] <= b[2] ? a[2] : b[2]);   // minf / maxf (objects.rs:81-86)
    const V hi = mk(a[0] >= b[0] ? a[0] : b[0], a[1] >= b[1] ? a[1] : b[1], a[2] >= b[2] ? a[2] : b[2]);
    const V dx = mk(hi.x - lo.x, 0, 0), dy = mk(0, hi.y - lo.y, 0), dz = mk(0, 0, hi.z - lo.z);
    const V org[6] = { mk(lo.x, lo.y, hi.z), mk(hi.x, lo.y, hi.z), mk(hi.x, lo.y, lo.z), mk(lo.x, lo.y, lo.z), mk(lo.x, hi.y, hi.z), mk(lo.x, lo.y, lo.z) };
    const V uu[6] = { dx, -dz, -dx, dz, dx, dx };
    const V vv[6] = { dy, dy, dy, dy, -dz, dz };
    for (int k = 0; k < 6; k++) {
        float o3[3], u3[3], v3[3]; st(o3, org[k]); st(u3, uu[k]); st(v3, vv[k]);
        int rc = rtw_quad_new(o3, u3, v3, mat3, nullptr, color, &q[k]);
        if (rc != RTW_OK) return rc;
    }
    return RTW_OK;
}

// ---- scene generators.  None of these scenes exists in the reference; they are
// expressed with the reference's Sphere::new / material presets. ------------------------------------
static void book1_small_spheres(Pcg32 &rng, std::vector<RtwSphere> &out, bool moving) {
    for (int a = -11; a < 11; a++) {
        for (int b = -11; b < 11; b++) {
            float choose = rng.f();
            V centre = mk((float)a + 0.9f * rng.f(), 0.2f, (float)b + 0.9f * rng.f());
            if (!(length(centre - mk(4.0f, 0.2f, 0.0f)) > 0.9f)) continue;
            if (choose < 0.8f) {
                V albedo = mk(rng.f() * rng.f(), rng.f() * rng.f(), rng.f() * rng.f());
                RtwSphere s = sphere_albedo(centre, 0.2f, albedo, SCATTER_M);
                if (moving) s.velocity[1] = rng.range(0.0f, 0.5f) * 30.0f;   // centre + (0, U[0,.5), 0) over one 1/30 s shutter
                out.push_back(s);
            } else if (choose < 0.95f) {
                V albedo = mk(rng.range(0.5f, 1.0f), rng.range(0.5f, 1.0f), rng.range(0.5f, 1.0f));
                float fuzz = rng.range(0.0f, 0.5f);
                const float m[3] = { 1.0f - fuzz, 0.0f, 1.0f };   // closest analogue of book fuzz: lerp toward lambert
                out.push_back(sphere_albedo(centre, 0.2f, albedo, m));
            } else {
                out.push_back(sphere_albedo(centre, 0.2f, mk(1, 1, 1), GLASS_M));
            }
        }
    }
    out.push_back(sphere_albedo(mk(0, 1, 0), 1.0f, mk(1, 1, 1), GLASS_M));
    out.push_back(sphere_albedo(mk(-4, 1, 0), 1.0f, mk(0.4f, 0.2f, 0.1f), SCATTER_M));
    out.push_back(sphere_albedo(mk(4, 1, 0), 1.0f, mk(0.7f, 0.6f, 0.5f), METALLIC_M));
}

int rtw_scene_generate(uint32_t which, uint64_t scene_seed,
                       RtwSphere *spheres, uint32_t sphere_cap, uint32_t *n_spheres,
                       RtwTexture *textures, uint32_t texture_cap, uint32_t *n_textures,
                       float *texels, uint32_t texel_cap, uint32_t *n_texels) {
    std::vector<RtwSphere> sp; std::vector<RtwTexture> tx; std::vector<float> tl;
    Pcg32 rng(scene_seed);
    switch (which) {
    case RTW_SCENE_C1_THREE_SPHERES:
        sp.push_back(sphere_new(mk(0, -100.5f, -1), 100.0f, mk(0.8f, 0.8f, 0.0f), SCATTER_M));
        sp.push_back(sphere_new(mk(0, 0, -1), 0.5f, mk(0.7f, 0.3f, 0.3f), SCATTER_M));
        sp.push_back(sphere_new(mk(1, 0, -1), 0.5f, mk(0.8f, 0.6f, 0.2f), METALLIC_M));
        break;
    case RTW_SCENE_METAL_TEST:    // material_tests.rs:105-167
        sp.push_back(sphere_new(mk(-1, 0, -1), 0.5f, mk(0.8f, 0.8f, 0.8f), FUZZY3_M));
        sp.push_back(sphere_new(mk(1, 0, -1), 0.5f, mk(0.8f, 0.6f, 0.2f), METALLIC_M));
        sp.push_back(sphere_new(mk(0, 0, -1), 0.5f, mk(0.7f, 0.3f, 0.3f), SCATTER_M));
        sp.push_back(sphere_new(mk(0, -100.5f, -1), 100.0f, mk(0.8f, 0.8f, 0.0f), SCATTER_M));
        break;
    case RTW_SCENE_FIRST_FRAME: { // main.rs:427-496, the scene of Rust/First frame.png
        sp.push_back(sphere_new(mk(-0.8f, 0, -1.0f), 0.4f, mk(1.0f, 0.6f, 0.6f), FUZZY3_M));
        sp.push_back(sphere_new(mk(0.6f, 0, -1.2f), 0.3f, mk(0.5f, 0.9f, 0.9f), METALLIC_M));
        RtwSphere mv = sphere_new(mk(2.4f, 0, -0.8f), 1.4f, mk(0.9f, 0.9f, 0.9f), METALLIC_M);   // Sphere::new_moving(.., (0, 60, 0))
        mv.velocity[1] = 60.0f;
        sp.push_back(mv);
        sp.push_back(sphere_new(mk(0, 0, -0.7f), 0.3f, mk(1, 1, 1), GLASS_M));
        sp.push_back(sphere_new(mk(0, 0, -0.7f), 0.2f, mk(1, 1, 1), GLASSR_M));
        sp.push_back(sphere_new(mk(0, 0, -2.0f), 1.0f, mk(0.5f, 1.0f, 0.0f), SCATTER_M));
        sp.push_back(sphere_new(mk(0, -1000.9f, -5.0f), 1000.0f, mk(0.8f, 0.5f, 1.0f), SCATTER_M));
        break;
    }
    case RTW_SCENE_C2_BOOK1_FINAL:
        sp.push_back(sphere_albedo(mk(0, -1000, 0), 1000.0f, mk(0.5f, 0.5f, 0.5f), SCATTER_M));
        book1_small_spheres(rng, sp, false);
        break;
    case RTW_SCENE_C5_MOTION_CHECKER: {
        // ground: 4x2 two-colour image through the sphere-UV lookup (squares.png-like, texture.rs:259-267)
        RtwTexture t; t.row = 4; t.col = 2; t.texel_offset = 0; t.emit_tex = 0; tx.push_back(t);
        for (int y = 0; y < 2; y++) for (int x = 0; x < 4; x++) {
            bool dark = ((x + y) & 1) != 0;
            const float c[3] = { dark ? 0.2f : 0.9f, dark ? 0.3f : 0.9f, dark ? 0.1f : 0.9f };
            tl.insert(tl.end(), c, c + 3);
        }
        RtwSphere g; const float o[3] = { 0, -1000, 0 };
        rtw_sphere_new_with_texture(o, 1000.0f, nullptr, SCATTER_M, nullptr, 0, &g);
        sp.push_back(g);
        book1_small_spheres(rng, sp, true);
        break;
    }
    case RTW_SCENE_C4_DIELECTRIC: {
        sp.push_back(sphere_albedo(mk(0, -1000, 0), 1000.0f, mk(0.5f, 0.5f, 0.5f), SCATTER_M));
        for (int a = -4; a <= 4; a++) for (int b = -4; b <= 4; b++) {   // material_tests.rs:215-250 hollow-glass pattern
            V c = mk(1.1f * (float)a, 0.45f, 1.1f * (float)b);
            sp.push_back(sphere_albedo(c, 0.45f, mk(1, 1, 1), GLASS_M));
            sp.push_back(sphere_albedo(c, 0.35f, mk(1, 1, 1), GLASSR_M));
        }
        for (int k = 0; k < 20; k++) {
            V c = mk(rng.range(-6.0f, 6.0f), 1.3f + 0.3f * rng.f(), rng.range(-6.0f, 6.0f));
            V albedo = mk(rng.range(0.5f, 1.0f), rng.range(0.5f, 1.0f), rng.range(0.5f, 1.0f));
            sp.push_back(sphere_albedo(c, 0.3f, albedo, FUZZY3_M));
        }
        break;
    }
    default: return RTW_E_INVALID;
    }
    if (n_spheres) *n_spheres = (uint32_t)sp.size();
    if (n_textures) *n_textures = (uint32_t)tx.size();
    if (n_texels) *n_texels = (uint32_t)(tl.size() / 3);
    if (!spheres) return RTW_OK;     // count query
    if (sphere_cap < sp.size() || texture_cap < tx.size() || texel_cap < tl.size() / 3) return RTW_E_INVALID;
    if ((!tx.empty() && !textures) || (!tl.empty() && !texels)) return RTW_E_INVALID;
    std::copy(sp.begin(), sp.end(), spheres);
    if (!tx.empty()) std::copy(tx.begin(), tx.end(), textures);
    if (!tl.empty()) std::copy(tl.begin(), tl.end(), texels);
    return RTW_OK;
}

static RtwQuad quad_of(V o, V u, V v, const float *mat3, V color, const float *emitted = nullptr) {
    RtwQuad q; float o3[3], u3[3], v3[3], c3[3]; st(o3, o); st(u3, u); st(v3, v); st(c3, color);
    rtw_quad_new(o3, u3, v3, mat3, emitted, c3, &q);
    return q;
}

int rtw_scene_generate_geom(uint32_t which, uint64_t scene_seed, RtwSphere *spheres, RtwQuad *quads,
                            RtwInstance *instances, RtwSphere *inst_spheres, RtwQuad *inst_quads,
                            const uint32_t caps[5], uint32_t counts[5], float background[3]) {
    std::vector<RtwSphere> sp, isp; std::vector<RtwQuad> qd, iqd; std::vector<RtwInstance> in;
    if (background) background[0] = background[1] = background[2] = 0.0f;        // Scene::new (viewport.rs:131-135)
    const float PI = 3.14159265358979323846f;
    switch (which) {
    case RTW_SCENE_QUAD_TEST:         // objects/quad.rs:152-299: red, green, blue, orange, teal
        qd.push_back(quad_of(mk(-3, -2, 5), mk(0, 0, -4), mk(0, 4, 0), SCATTER_M, mk(1.0f, 0.2f, 0.2f)));
        qd.push_back(quad_of(mk(-2, -2, 0), mk(4, 0, 0), mk(0, 4, 0), SCATTER_M, mk(0.2f, 1.0f, 0.2f)));
        qd.push_back(quad_of(mk(3, -2, 1), mk(0, 0, 4), mk(0, 4, 0), SCATTER_M, mk(0.2f, 0.2f, 1.0f)));
        qd.push_back(quad_of(mk(-2, 3, 1), mk(4, 0, 0), mk(0, 0, 4), SCATTER_M, mk(1.0f, 0.5f, 0.0f)));
        qd.push_back(quad_of(mk(-2, -3, 5), mk(4, 0, 0), mk(0, 0, -4), SCATTER_M, mk(0.2f, 0.8f, 0.8f)));
        break;
    case RTW_SCENE_PRESENTATION: {    // main.rs:89-419
        sp.push_back(sphere_new(mk(-1.6f, -1.6f, 3.0f), 0.4f, mk(1, 0, 0), SCATTER_M));
        const float light[3] = { 4, 4, 4 }, dark[3] = { 0, 0, 0 };
        const float emit_m[3] = { 0.0f, 0.0f, 1.0f };                                     // Material::new_emmiting(0, 0, 1, ..)
        qd.push_back(quad_of(mk(-2, -2, 5), mk(0, 0, -4), mk(0, 4, 0), METALLIC_M, mk(0.85f, 0.85f, 0.85f)));        // "Red": a mirror wall
        qd.push_back(quad_of(mk(-1.5f, -1.5f, 1.001f), mk(3, 0, 0), mk(0, 3, 0), emit_m, mk(1, 1, 1), light));   // Light
        qd.push_back(quad_of(mk(2, -2, 1), mk(0, 0, 4), mk(0, 4, 0), SCATTER_M, mk(0.2f, 0.2f, 1.0f)));             // Blue
        qd.push_back(quad_of(mk(-2, 2, 1), mk(4, 0, 0), mk(0, 0, 4), SCATTER_M, mk(1.0f, 0.5f, 0.0f)));             // Orange
        qd.push_back(quad_of(mk(-2, -2, 5), mk(4, 0, 0), mk(0, 0, -4), SCATTER_M, mk(0.2f, 0.8f, 0.8f)));           // Teal
        qd.push_back(quad_of(mk(-2, -2, 1), mk(4, 0, 0), mk(0, 4, 0), emit_m, mk(0.2f, 1.0f, 0.2f), dark));        // Green
        // smoke: Instance::new_box((-1,-.5,-.5), (1,.5,.5), 0.2 grey, SCATTER_M), translate (0,-1.5,2), rotate y pi/4, density 2
        RtwInstance smoke; std::memset(&smoke, 0, sizeof smoke);
        const float ba[3] = { -1.0f, -0.5f, -0.5f }, bb[3] = { 1.0f, 0.5f, 0.5f }, grey[3] = { 0.2f, 0.2f, 0.2f };
        RtwQuad box[6]; rtw_box_quads(ba, bb, SCATTER_M, grey, box);
        smoke.first_quad = 0; smoke.n_quads = 6; iqd.insert(iqd.end(), box, box + 6);
        smoke.translation[0] = -0.0f; smoke.translation[1] = -1.5f; smoke.translation[2] = 2.0f;
        smoke.rotation[1] = PI / 4.0f;
        smoke.medium = RTW_MEDIUM_CONST_DENSITY; smoke.density = 2.0f;
        in.push_back(smoke);
        // glass pane: two coincident-ish quads, ir 1.5 and 2/3 (opacity 0.8 only selects the dielectric branch)
        RtwInstance glass; std::memset(&glass, 0, sizeof glass);
        const float g1[3] = { 1.0f, 0.8f, 1.50f }, g2[3] = { 1.0f, 0.8f, 2.0f / 3.0f };
        glass.first_quad = 6; glass.n_quads = 2;
        iqd.push_back(quad_of(mk(0, 0, 0), mk(0, 4, 0), mk(2, 0, 0), g1, mk(0.8f, 0.8f, 0.8f)));
        iqd.push_back(quad_of(mk(0, 0, -0.001f), mk(0, 4, 0), mk(2, 0, 0), g2, mk(0.8f, 0.8f, 0.8f)));
        glass.translation[0] = 2.0f - std::sqrt(3.0f); glass.translation[1] = -2.0f; glass.translation[2] = 3.0f;
        glass.rotation[1] = -PI / 6.0f;
        in.push_back(glass);
        break;
    }
    default: {                        // the sphere-only scenes
        uint32_t n = 0, nt = 0, nl = 0;
        int rc = rtw_scene_generate(which, scene_seed, nullptr, 0, &n, nullptr, 0, &nt, nullptr, 0, &nl);
        if (rc != RTW_OK) return rc;
        if (nt) return RTW_E_INVALID;                 // textured scenes go through rtw_scene_generate
        sp.resize(n);
        rc = rtw_scene_generate(which, scene_seed, sp.data(), n, &n, nullptr, 0, &nt, nullptr, 0, &nl);
        if (rc != RTW_OK) return rc;
    }
    }
    const size_t n5[5] = { sp.size(), qd.size(), in.size(), isp.size(), iqd.size() };
    if (counts) for (int k = 0; k < 5; k++) counts[k] = (uint32_t)n5[k];
    void *dst[5] = { spheres, quads, instances, inst_spheres, inst_quads };
    const void *src[5] = { sp.data(), qd.data(), in.data(), isp.data(), iqd.data() };
    const size_t esz[5] = { sizeof(RtwSphere), sizeof(RtwQuad), sizeof(RtwInstance), sizeof(RtwSphere), sizeof(RtwQuad) };
    for (int k = 0; k < 5; k++) {
        if (!dst[k] || n5[k] == 0) continue;
        if (!caps || caps[k] < n5[k]) return RTW_E_INVALID;
        std::memcpy(dst[k], src[k], n5[k] * esz[k]);
    }
    return RTW_OK;
}

int rtw_scene_default_view(uint32_t which, RtwCamera *cam, RtwParams *p) {
    if (!cam || !p) return RTW_E_INVALID;
    std::memset(p, 0, sizeof *p);
    p->gamma = 2.0f; p->mint = 0.001f; p->maxt = 100000.0f;     // ray_color_gradient (ray_color.rs:17-18)
    p->integrator = RTW_INTEGRATOR_GRADIENT; p->sampler = RTW_SAMPLER_ROW; p->accel = RTW_ACCEL_BVH;
    p->seed = 1; p->row_block = 8; p->part_index = 0; p->part_count = 1;
    uint32_t h = 0;
    if (which == RTW_SCENE_C1_THREE_SPHERES || which == RTW_SCENE_METAL_TEST) {
        p->width = 400; p->height = 225; p->depth = 10;
        p->samples = which == RTW_SCENE_METAL_TEST ? 100 : 10;
        if (which == RTW_SCENE_METAL_TEST) { p->sampler = RTW_SAMPLER_STRATIFIED; p->maxt = 1000.0f; }   // viewport.render(&ray_color_d) material_tests.rs:165
        return rtw_viewport_new_from_res(p->width, p->height, nullptr, nullptr, nullptr, nullptr, nullptr, cam, &h);
    }
    if (which == RTW_SCENE_FIRST_FRAME) {        // main.rs:497-545: 400x400, 100 spp, depth 100, frame 0 at 15 fps, shutter 0, test_run(.., ray_color_d, ..)
        p->width = 400; p->height = 400; p->samples = 100; p->depth = 100; p->maxt = 1000.0f;
        return rtw_viewport_new_from_res(p->width, p->height, nullptr, nullptr, nullptr, nullptr, nullptr, cam, &h);
    }
    if (which == RTW_SCENE_QUAD_TEST) {          // quad.rs:277-298: 400x400, 100 spp, depth 10, vfov 80 from (0,0,9), viewport.render(&ray_color_d)
        const float vf = 80.0f, from9[3] = { 0, 0, 9 }, fwd[3] = { 0, 0, -1 };
        p->width = 400; p->height = 400; p->samples = 100; p->depth = 10;
        p->sampler = RTW_SAMPLER_STRATIFIED; p->maxt = 1000.0f;
        return rtw_viewport_new_from_res(p->width, p->height, &vf, from9, fwd, nullptr, nullptr, cam, &h);
    }
    if (which == RTW_SCENE_PRESENTATION) {       // main.rs:386-417: 400x400, 2500 spp, depth 20, vfov 90 from (0,0,7), async_render(&ray_color_bg_color)
        const float vf = 90.0f, from7[3] = { 0, -0.0f, 7 }, fwd[3] = { 0, 0, -1 };
        p->width = 400; p->height = 400; p->samples = 2500; p->depth = 20;
        p->integrator = RTW_INTEGRATOR_BG_COLOR; p->maxt = 10000.0f;     // ray_color.rs:48-49
        return rtw_viewport_new_from_res(p->width, p->height, &vf, from7, fwd, nullptr, nullptr, cam, &h);
    }
    // Book-1 framing: look from (13,2,3) at the origin, vfov 20, aperture 0.1 (lens_radius 0.05).
    // The reference does not normalise `direction`: w = -direction and v = w x u inherit its length
    // (viewport.rs:342-344), and the focal length is never applied (:349-351), so the only
    // undistorted view is a UNIT direction (focus plane at distance 1; lens rays of a pixel stay
    // parallel, i.e. a constant 0.05-unit blur rather than a focus distance of 10).
    const float from[3] = { 13.0f, 2.0f, 3.0f };
    const float len = std::sqrt(13.0f * 13.0f + 2.0f * 2.0f + 3.0f * 3.0f);
    const float dir[3] = { -13.0f / len, -2.0f / len, -3.0f / len };
    const float vfov = 20.0f;
    const float lens = 0.05f;
    p->depth = 50;
    switch (which) {
    case RTW_SCENE_C2_BOOK1_FINAL: p->width = 1200; p->height = 675; p->samples = 100; break;
    case RTW_SCENE_C4_DIELECTRIC: p->width = 1920; p->height = 1080; p->samples = 1000; break;
    case RTW_SCENE_C5_MOTION_CHECKER: p->width = 1920; p->height = 1080; p->samples = 500; break;
    default: return RTW_E_INVALID;
    }
    int rc = rtw_viewport_new_from_res(p->width, p->height, &vfov, from, dir, nullptr, &lens, cam, &h);
    if (rc != RTW_OK) return rc;
    if (which == RTW_SCENE_C5_MOTION_CHECKER) { cam->time0 = 0.0f / 30.0f; cam->shutter = 1.0f / 30.0f; }   // frame 0, fps 30, shutter 1/fps
    return RTW_OK;
}

} // extern "C"

// ---- BVH build ------------------------------------------------------------------------------------
namespace rtw {
namespace {

struct Box { float lo[3], hi[3]; };
inline void box_empty(Box &b) { for (int k = 0; k < 3; k++) { b.lo[k] = FLT_MAX; b.hi[k] = -FLT_MAX; } }
inline void box_grow(Box &b, const Box &o) { for (int k = 0; k < 3; k++) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
inline float box_area(const Box &b) {
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

struct Prim { Box box; float cen[3]; uint32_t sphere; };
inline uint32_t ceil_log2(uint32_t n) { uint32_t l = 0; while ((1ull << l) < n) l++; return l; }

struct Builder {
    std::vector<Prim> prims;
    std::vector<BvhNode> nodes;
    uint32_t max_depth = 0;

    // returns child reference (>= 0 node, < 0 ~sphere) and its box
    int32_t build(uint32_t first, uint32_t count, uint32_t depth, Box &out_box) {
        max_depth = std::max(max_depth, depth);
        Box bb; box_empty(bb);
        for (uint32_t i = first; i < first + count; i++) box_grow(bb, prims[i].box);
        out_box = bb;
        if (count == 1) return ~(int32_t)prims[first].sphere;

        // binned SAH over centroids
        Box cb; box_empty(cb);
        for (uint32_t i = first; i < first + count; i++)
            for (int k = 0; k < 3; k++) { cb.lo[k] = std::min(cb.lo[k], prims[i].cen[k]); cb.hi[k] = std::max(cb.hi[k], prims[i].cen[k]); }
        uint32_t mid = first + count / 2;
        int best_axis = -1; float best_cost = FLT_MAX; float best_split = 0.0f;
        const int NB = 16;
        // Depth bound by construction (rtw_host.h: depth <= RTW_BVH_STACK): a median split of `count` spheres needs
        // ceil(log2(count)) more levels, so the invariant is  depth + ceil(log2(count)) <= RTW_BVH_STACK.  It holds at the root
        // (count <= 2^24) and a median split keeps it; a SAH split may leave count - 1 spheres on one side, so SAH is only tried
        // when even that child keeps the invariant.  (Geometrically spaced spheres used to reach depth 27 under the old
        // "SAH until depth 20" rule and overflowed the device stack.)
        if (depth + 1 + ceil_log2(count - 1) <= RTW_BVH_STACK) {
            for (int ax = 0; ax < 3; ax++) {
                float ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0.0f)) continue;
                Box bins[NB]; uint32_t cnt[NB];
                for (int b = 0; b < NB; b++) { box_empty(bins[b]); cnt[b] = 0; }
                for (uint32_t i = first; i < first + count; i++) {
                    // (clamped BEFORE the conversion: non-finite centres make this NaN or +-inf, and float -> int of those is undefined)
                    const float fb = (prims[i].cen[ax] - cb.lo[ax]) / ext * NB;
                    const int b = fb >= 0.0f ? (fb < (float)NB ? (int)fb : NB - 1) : 0;
                    box_grow(bins[b], prims[i].box); cnt[b]++;
                }
                float right_area[NB]; uint32_t right_cnt[NB];
                Box acc; box_empty(acc); uint32_t c = 0;
                for (int b = NB - 1; b > 0; b--) { box_grow(acc, bins[b]); c += cnt[b]; right_area[b] = c ? box_area(acc) : 0.0f; right_cnt[b] = c; }
                box_empty(acc); c = 0;
                for (int b = 0; b < NB - 1; b++) {
                    box_grow(acc, bins[b]); c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    float cost = box_area(acc) * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_split = cb.lo[ax] + ext * (float)(b + 1) / NB; }
                }
            }
        }
        if (best_axis >= 0) {
            auto it = std::partition(prims.begin() + first, prims.begin() + first + count,
                                     [&](const Prim &p) { return p.cen[best_axis] < best_split; });
            mid = (uint32_t)(it - prims.begin());
        }
        if (best_axis < 0 || mid == first || mid == first + count) {
            // median split on the widest centroid axis (also the depth-limit fallback)
            int ax = 0;
            for (int k = 1; k < 3; k++) if (cb.hi[k] - cb.lo[k] > cb.hi[ax] - cb.lo[ax]) ax = k;
            mid = first + count / 2;
            std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + first + count,
                             [&](const Prim &a, const Prim &b) { return a.cen[ax] < b.cen[ax]; });
        }
        uint32_t me = (uint32_t)nodes.size();
        nodes.push_back(BvhNode{});
        Box b0, b1;
        int32_t c0 = build(first, mid - first, depth + 1, b0);
        int32_t c1 = build(mid, first + count - mid, depth + 1, b1);
        BvhNode &n = nodes[me];
        for (int k = 0; k < 3; k++) { n.lo0[k] = b0.lo[k]; n.hi0[k] = b0.hi[k]; n.lo1[k] = b1.lo[k]; n.hi1[k] = b1.hi[k]; }
        n.c0 = c0; n.c1 = c1; n.pad[0] = n.pad[1] = 0;
        return (int32_t)me;
    }
};

} // namespace

namespace {
// f32 -> f16 bits with directed rounding (value must be finite and within the f16 range)
uint16_t half_bits(float f, bool up) {
    _Float16 h = (_Float16)f;                 // round to nearest even
    uint16_t b; std::memcpy(&b, &h, 2);
    const float back = (float)h;
    if (up ? back < f : back > f) {           // step one ulp toward the requested direction
        if (b == 0x8000u) b = 0;              // -0 -> +0 before stepping
        const bool neg = (b & 0x8000u) != 0;
        if (up) b = neg ? (uint16_t)(b - 1) : (uint16_t)(b + 1);
        else    b = neg ? (uint16_t)(b + 1) : (b == 0 ? (uint16_t)0x8001u : (uint16_t)(b - 1));
    }
    return b;
}
} // namespace

void build_bvh(const RtwSphere *spheres, uint32_t n, float t_begin, float t_end, BvhBuild &out) {
    out.nodes.clear(); out.nodes16.clear(); out.big.clear();
    out.root = std::numeric_limits<int32_t>::min();
    out.centre[0] = out.centre[1] = out.centre[2] = 0.0f;
    out.centre_radius = 0.0f; out.r_min = 0.0f; out.r_max = 0.0f; out.abs_max = 0.0f; out.depth = 0;
    if (n == 0) return;

    // "big" spheres: radius >= 16 x the median radius, at most RTW_MAX_BIG of them (largest first)
    std::vector<float> radii(n);
    for (uint32_t i = 0; i < n; i++) radii[i] = std::fabs(spheres[i].radius);
    std::vector<float> sorted = radii;
    std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
    const float r_med = sorted[n / 2];
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return radii[a] > radii[b]; });
    std::vector<char> is_big(n, 0);
    if (n > 2) {
        for (uint32_t k = 0; k < n && out.big.size() < RTW_MAX_BIG; k++) {
            uint32_t i = order[k];
            if (!(radii[i] >= 16.0f * r_med) && std::isfinite(radii[i])) break;
            is_big[i] = 1; out.big.push_back(i);
        }
    }
    std::sort(out.big.begin(), out.big.end());

    Builder b;
    Box centres; box_empty(centres);
    float rmin = FLT_MAX, rmax = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        if (is_big[i]) continue;
        const RtwSphere &s = spheres[i];
        Prim p; p.sphere = i;
        for (int k = 0; k < 3; k++) {
            // centre(t) = fl(origin + fl(velocity * t)) is monotone in t: the two ends bound it
            float c0 = s.center[k] + s.velocity[k] * t_begin;
            float c1 = s.center[k] + s.velocity[k] * t_end;
            float lo = std::min(c0, c1), hi = std::max(c0, c1);
            if (!(lo == lo) || !(hi == hi)) { lo = -FLT_MAX; hi = FLT_MAX; }
            centres.lo[k] = std::min(centres.lo[k], lo); centres.hi[k] = std::max(centres.hi[k], hi);
            p.cen[k] = 0.5f * lo + 0.5f * hi;
            p.box.lo[k] = std::nextafter(lo - radii[i], -FLT_MAX);
            p.box.hi[k] = std::nextafter(hi + radii[i], FLT_MAX);
        }
        rmin = std::min(rmin, radii[i]); rmax = std::max(rmax, radii[i]);
        b.prims.push_back(p);
    }
    if (b.prims.empty()) return;
    Box root_box;
    out.root = b.build(0, (uint32_t)b.prims.size(), 0, root_box);
    out.nodes.swap(b.nodes);
    out.depth = b.max_depth;
    double r2 = 0.0;
    for (int k = 0; k < 3; k++) {
        out.centre[k] = 0.5f * centres.lo[k] + 0.5f * centres.hi[k];
        double h = std::max(std::fabs((double)centres.hi[k] - out.centre[k]), std::fabs((double)centres.lo[k] - out.centre[k]));
        r2 += h * h;
        out.abs_max = std::max(out.abs_max, std::max(std::fabs(root_box.lo[k]), std::fabs(root_box.hi[k])));
    }
    out.centre_radius = (float)(std::sqrt(r2) * 1.000001);
    out.r_min = rmin; out.r_max = rmax;

    // f16 copy for the LDS-resident traversal: only when it fits and every coordinate is far inside the
    // f16 range (outward rounding then costs <= 2^-11 relative per plane)
    if (!out.nodes.empty() && out.nodes.size() <= RTW_LDS_NODES_MAX && n <= RTW_LDS_GEOM_MAX && out.abs_max < 30000.0f) {
        out.nodes16.resize(out.nodes.size());
        for (size_t i = 0; i < out.nodes.size(); i++) {
            const BvhNode &a = out.nodes[i]; BvhNode16 &b = out.nodes16[i];
            for (int k = 0; k < 3; k++) {
                b.plane[0][k][0] = half_bits(a.lo0[k], false); b.plane[0][k][1] = half_bits(a.hi0[k], true);
                b.plane[1][k][0] = half_bits(a.lo1[k], false); b.plane[1][k][1] = half_bits(a.hi1[k], true);
            }
            // child references as the traversal consumes them: a leaf as ~sphere (negative), an inner node as its BYTE OFFSET in the
            // f16 array (id * 32 <= 16352 < the END / DEAD codes 0x7FFE / 0x7FFF): with the array at LDS offset 0 that is the node's address
            b.c0 = (int16_t)(a.c0 >= 0 ? a.c0 * 32 : a.c0); b.c1 = (int16_t)(a.c1 >= 0 ? a.c1 * 32 : a.c1); b.pad = 0;
        }
    }
}

} // namespace rtw

// ---- queue order of the tiles ------------------------------------------------------------------------------------------------
namespace rtw {
namespace {
uint32_t gcd_u32(uint32_t x, uint32_t y) { while (y) { const uint32_t t = x % y; x = y; y = t; } return x; }
// q -> (q * m) % n with m near n / golden ratio and coprime to n: a bijection that sends neighbours far apart
std::vector<uint32_t> scatter_perm(uint32_t n) {
    std::vector<uint32_t> v(n);
    if (n < 3u) { for (uint32_t q = 0; q < n; q++) v[q] = q; return v; }
    uint32_t m = (uint32_t)((double)n * 0.6180339887) | 1u;
    while (gcd_u32(m, n) != 1u) m += 2u;
    for (uint32_t q = 0; q < n; q++) v[q] = (uint32_t)(((uint64_t)q * m) % n);
    return v;
}
bool ray_hits_box(const double o[3], const double d[3], const float lo[3], const float hi[3], double &t_entry) {
    double t0 = 0.0, t1 = 1e300;
    for (int k = 0; k < 3; k++) {
        if (d[k] == 0.0) { if (o[k] < lo[k] || o[k] > hi[k]) return false; continue; }
        double a = (lo[k] - o[k]) / d[k], b = (hi[k] - o[k]) / d[k];
        if (a > b) std::swap(a, b);
        t0 = std::max(t0, a); t1 = std::min(t1, b);
    }
    t_entry = t0;
    return t0 <= t1;
}
bool ray_hits_sphere(const double o[3], const double d[3], const float c[3], double r, double &t_entry) {
    const double oc[3] = { o[0] - c[0], o[1] - c[1], o[2] - c[2] };
    const double a = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], b = oc[0] * d[0] + oc[1] * d[1] + oc[2] * d[2];
    const double cc = oc[0] * oc[0] + oc[1] * oc[1] + oc[2] * oc[2] - r * r, disc = b * b - a * cc;
    if (!(disc >= 0.0) || !(a > 0.0)) return false;
    const double sq = std::sqrt(disc);
    t_entry = std::max(0.0, (-b - sq) / a);
    return (-b + sq) / a > 0.0;                        // some part of the sphere lies ahead of the origin
}
} // namespace

void scene_cull_from_bvh(const BvhBuild &bb, const RtwSphere *spheres, SceneCull &out) {
    out = SceneCull{};
    for (uint32_t i : bb.big) if (out.n_big < 16u) {
        float *d = out.big[out.n_big++];
        d[0] = spheres[i].center[0]; d[1] = spheres[i].center[1]; d[2] = spheres[i].center[2]; d[3] = std::fabs(spheres[i].radius);
    }
    if (bb.root == std::numeric_limits<int32_t>::min()) return;
    out.has_tree = 1u;
    if (bb.nodes.empty()) {                                   // a single tree sphere: its own box (static part; good enough for a heuristic)
        const RtwSphere &s = spheres[(uint32_t)~bb.root];
        for (int k = 0; k < 3; k++) { out.lo[k] = s.center[k] - std::fabs(s.radius) - bb.abs_max * 0.0f; out.hi[k] = s.center[k] + std::fabs(s.radius); }
        return;
    }
    const BvhNode &r = bb.nodes[0];
    for (int k = 0; k < 3; k++) { out.lo[k] = std::min(r.lo0[k], r.lo1[k]); out.hi[k] = std::max(r.hi0[k], r.hi1[k]); }
}

void build_tile_order(uint32_t mode, uint32_t tiles_x, uint32_t tiles_y, uint32_t k_base, uint32_t row_block, uint32_t part_index,
                      uint32_t part_count, const RtwCamera &cam, const SceneCull &cull, uint32_t tail_tiles, std::vector<uint32_t> &order) {
    const uint32_t n = tiles_x * tiles_y;
    order.resize(n);
    if (mode == 3u) { for (uint32_t q = 0; q < n; q++) order[q] = n - 1u - q; return; }
    if (mode != 1u && mode != 2u && mode != 4u && mode != 5u) { for (uint32_t q = 0; q < n; q++) order[q] = q; return; }
    const uint32_t g = 8u, ng = (n + g - 1u) / g;      // (group sizes 2..64 measure the same within noise: profiles/r02_tile_order.log)
    const std::vector<uint32_t> grp = scatter_perm(ng);
    uint32_t k = 0;
    for (uint32_t q = 0; q < ng; q++)
        for (uint32_t i = 0; i < g; i++) { const uint32_t t = grp[q] * g + i; if (t < n) order[k++] = t; }
    if ((mode != 2u && mode != 4u && mode != 5u) || cull.n_other || (!cull.has_tree && !cull.n_big)) {
        if (mode == 4u || mode == 5u) for (uint32_t q = 0; q < n; q++) order[q] = q;          // (no cost guess: raster)
        return;
    }
    // Mode 2: longest-processing-time-first by an ESTIMATE of a tile's cost, from its centre ray (no lens offset, no jitter):
    //   class 2  the ray enters the root box of the tree's spheres: the sphere field -- nearer means larger spheres on screen, more
    //            pixels that hit one, more inter-reflection: cost falls with the distance at which the ray reaches the ground (a big
    //            sphere) under the field; rays through the box that never reach it come after those
    //   class 1  it hits only a sphere kept outside the tree (the ground): one bounce, then mostly sky -- nearer first again
    //   class 0  it hits nothing: sky
    // Expensive tiles first, cheap tiles last: the launch ends on short paths (for a camera above the ground looking at the scene this
    // is close to "bottom rows first"; it follows the camera when that looks elsewhere).  `tail_tiles` is not needed by a full sort.
    (void)tail_tiles;
    struct Key { uint8_t cls; float dist; uint32_t tile; };
    std::vector<Key> keys(n);
    const double o[3] = { cam.origin[0], cam.origin[1], cam.origin[2] };
    const uint32_t rb = row_block ? row_block : 1u;
    for (uint32_t t = 0; t < n; t++) {
        const uint32_t tcol = t % tiles_x, krow = k_base + (t / tiles_x) * 8u + 4u;      // compact row of the partition -> image row
        const double j = part_count > 1u ? (double)(((krow / rb) * part_count + part_index) * rb + krow % rb) : (double)krow, i = tcol * 8.0 + 4.0;
        double d[3], len2 = 0.0;
        for (int a = 0; a < 3; a++) { d[a] = cam.pixel00[a] + cam.delta_u[a] * i + cam.delta_v[a] * j; len2 += d[a] * d[a]; }
        const double len = std::sqrt(len2);
        Key kx; kx.cls = 0; kx.dist = 0.0f; kx.tile = t;
        // distance along the centre ray to the nearest big sphere (the ground under the sphere field), else to the root box
        double te = 0.0, big = 1e300, box = 0.0;
        for (uint32_t b = 0; b < cull.n_big; b++) if (ray_hits_sphere(o, d, cull.big[b], cull.big[b][3], te)) big = std::min(big, te * len);
        if (cull.has_tree && ray_hits_box(o, d, cull.lo, cull.hi, box)) { kx.cls = 2; kx.dist = (float)(big < 1e300 ? big : box * len + 1e6); }
        else if (big < 1e300) { kx.cls = 1; kx.dist = (float)big; }
        if (!(kx.dist == kx.dist)) kx.dist = 0.0f;           // (degenerate cameras)
        keys[t] = kx;
    }
    if (mode == 5u) {
        // Mode 5: raster order inside each class, the classes in the order sphere field, bare ground, sky.  What ends a launch is the longest PATH
        // that starts late (profiles/r03_endtimes.log: with units of one sample at the end of the queue a wave still runs on for 0.3 ms on average and
        // 1.1 ms at most after the queue is empty -- a 50-bounce glass path at seven busy waves per SIMD); a tile of sky holds no such path.
        std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.cls > b.cls; });
        for (uint32_t q = 0; q < n; q++) order[q] = keys[q].tile;
        return;
    }
    std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.cls != b.cls ? a.cls > b.cls : a.dist < b.dist; });
    if (mode == 4u) {
        // Mode 4: the same cost order, but only as coarse as 32 buckets per class; inside a bucket the tiles keep RASTER order, so that
        // tiles handed out one after the other are mostly neighbours in the image (their rays meet the same nodes and spheres, their
        // bank rows are adjacent) -- the exact sort of mode 2 scatters the tiles of one distance over the whole width of the frame.
        uint32_t b0 = 0;
        while (b0 < n) {
            uint32_t b1 = b0; while (b1 < n && keys[b1].cls == keys[b0].cls) b1++;
            const uint32_t cnt = b1 - b0, per = (cnt + 31u) / 32u;
            for (uint32_t s0 = b0; s0 < b1; s0 += per) {
                const uint32_t s1 = std::min(b1, s0 + per);
                std::sort(keys.begin() + s0, keys.begin() + s1, [](const Key &a, const Key &b) { return a.tile < b.tile; });
            }
            b0 = b1;
        }
    }
    for (uint32_t q = 0; q < n; q++) order[q] = keys[q].tile;
}
} // namespace rtw

// Exposed for the CPU tests: the permutation RTW_OPT_TILE_ORDER = mode would use for a whole frame of this camera.
extern "C" int rtw_tile_order(uint32_t mode, uint32_t width, uint32_t height, const RtwCamera *cam, const RtwScene *scene, uint32_t *order, uint32_t cap) {
    using namespace rtw;
    if (!cam || !order || width == 0 || height == 0) return RTW_E_INVALID;
    const uint32_t tx = (width + 7) / 8, ty = (height + 7) / 8;
    if ((uint64_t)tx * ty > cap) return RTW_E_INVALID;
    SceneCull cull;
    if (scene && scene->n_spheres && scene->spheres) {
        BvhBuild bb;
        build_bvh(scene->spheres, scene->n_spheres, cam->time0, cam->time0 + cam->shutter, bb);
        scene_cull_from_bvh(bb, scene->spheres, cull);
        cull.n_other = scene->n_quads + scene->n_instances;
    }
    std::vector<uint32_t> v;
    build_tile_order(mode, tx, ty, 0, 8, 0, 1, *cam, cull, 0, v);
    std::copy(v.begin(), v.end(), order);
    return RTW_OK;
}

// Host-side self-check of the acceleration structure (no GPU): every tree sphere is reachable exactly once,
// its time-expanded bounds lie inside the box its parent stores for it and inside every ancestor's, the f16
// copy (when present) contains the f32 boxes, depth <= RTW_BVH_STACK, big + tree spheres == all spheres.
extern "C" int rtw_bvh_validate(const RtwScene *sc, float t_begin, float t_end, uint32_t *n_nodes, uint32_t *depth, uint32_t *n_big, uint32_t *has_f16) {
    using namespace rtw;
    if (!sc || (sc->n_spheres && !sc->spheres)) return RTW_E_INVALID;
    BvhBuild b;
    build_bvh(sc->spheres, sc->n_spheres, t_begin, t_end, b);
    if (n_nodes) *n_nodes = (uint32_t)b.nodes.size();
    if (depth) *depth = b.depth;
    if (n_big) *n_big = (uint32_t)b.big.size();
    if (has_f16) *has_f16 = b.nodes16.empty() ? 0u : 1u;
    std::vector<int> seen(sc->n_spheres, 0);
    for (uint32_t i : b.big) { if (i >= sc->n_spheres || seen[i]++) return RTW_E_INVALID; }
    if (b.depth > RTW_BVH_STACK) return RTW_E_INVALID;
    auto half = [](uint16_t x) { _Float16 v; std::memcpy(&v, &x, 2); return (float)v; };
    struct Item { int32_t ref; float lo[3], hi[3]; uint32_t d; };
    std::vector<Item> todo;
    if (b.root != std::numeric_limits<int32_t>::min()) {
        Item r; r.ref = b.root; r.d = 0;
        for (int k = 0; k < 3; k++) { r.lo[k] = -FLT_MAX; r.hi[k] = FLT_MAX; }
        todo.push_back(r);
    }
    while (!todo.empty()) {
        Item it = todo.back(); todo.pop_back();
        if (it.d > RTW_BVH_STACK) return RTW_E_INVALID;
        if (it.ref < 0) {
            const uint32_t s = (uint32_t)~it.ref;
            if (s >= sc->n_spheres || seen[s]++) return RTW_E_INVALID;
            const RtwSphere &sp = sc->spheres[s];
            for (int k = 0; k < 3; k++) for (float t : { t_begin, t_end }) {
                const float c = sp.center[k] + sp.velocity[k] * t, r = std::fabs(sp.radius);
                if (!(c - r >= it.lo[k] && c + r <= it.hi[k])) return RTW_E_INVALID;
            }
            continue;
        }
        if ((size_t)it.ref >= b.nodes.size()) return RTW_E_INVALID;
        const BvhNode &n = b.nodes[it.ref];
        for (int c = 0; c < 2; c++) {
            Item ch; ch.ref = c ? n.c1 : n.c0; ch.d = it.d + 1;
            for (int k = 0; k < 3; k++) {
                ch.lo[k] = c ? n.lo1[k] : n.lo0[k]; ch.hi[k] = c ? n.hi1[k] : n.hi0[k];
                if (!(ch.lo[k] >= it.lo[k] && ch.hi[k] <= it.hi[k] && ch.lo[k] <= ch.hi[k])) return RTW_E_INVALID;
                if (!b.nodes16.empty()) {
                    const BvhNode16 &h = b.nodes16[it.ref];
                    const float l16 = half(h.plane[c][k][0]), h16 = half(h.plane[c][k][1]);
                    if (!(l16 <= ch.lo[k] && h16 >= ch.hi[k])) return RTW_E_INVALID;
                    if ((int32_t)(c ? h.c1 : h.c0) != (ch.ref >= 0 ? ch.ref * 32 : ch.ref)) return RTW_E_INVALID;
                }
            }
            todo.push_back(ch);
        }
    }
    for (uint32_t i = 0; i < sc->n_spheres; i++) if (seen[i] != 1) return RTW_E_INVALID;
    return RTW_OK;
}
