// rtw_shim.hip -- extern "C" entry points of librtw_hip.so (include/rtw.h): context, scene upload,
// blocking render.  Host code only; kernels live in rtw_kernels.hip.  No CPU fallback of any kind:
// without a HIP device every render entry point fails with RTW_E_NO_DEVICE / RTW_E_HIP.
#include "rtw_kernels.h"
#include "rtw_host.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

using namespace rtw;

static thread_local int g_last_hip = 0;

#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) { g_last_hip = (int)e_; return RTW_E_HIP; } \
    } while (0)

struct rtw_ctx {
    int device = 0;
    int n_cu = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // scene
    bool has_scene = false;
    bool has_textures = false;
    DevScene sc{};
    float bg[3] = { 0, 0, 0 };
    void *d_geom = nullptr, *d_vel = nullptr, *d_mat = nullptr, *d_tex = nullptr, *d_texels = nullptr;
    DevGeom geom{};
    void *d_quads = nullptr, *d_inst = nullptr, *d_igeom = nullptr, *d_ivel = nullptr, *d_imat = nullptr, *d_iquads = nullptr;
    DevBvh bvh{};
    void *d_nodes = nullptr, *d_nodes16 = nullptr, *d_big_geom = nullptr, *d_big_vel = nullptr, *d_big_index = nullptr;
    // scratch
    uint32_t *d_queue = nullptr;
    unsigned long long *d_stats = nullptr;
    float *d_out = nullptr;
    size_t d_out_cap = 0;
    float *d_samples = nullptr;          // per-sample radiance bank (see rtw_ctx_render)
    size_t d_samples_cap = 0;
};

// The device records of a sphere list (sphere.rs:13-20 + materials.rs:15-20)
static void prepare_spheres(const RtwSphere *sp, uint32_t n, std::vector<f4> &geom, std::vector<f4> &vel, std::vector<DevMat> &mat, bool &moving) {
    geom.resize(n); vel.resize(n); mat.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const RtwSphere &s = sp[i];
        geom[i] = f4{ s.center[0], s.center[1], s.center[2], host_mul(s.radius, s.radius) };   // sphere.rs:105 radius*radius
        vel[i] = f4{ s.velocity[0], s.velocity[1], s.velocity[2], 0.0f };
        if (s.velocity[0] != 0.0f || s.velocity[1] != 0.0f || s.velocity[2] != 0.0f) moving = true;
        DevMat &m = mat[i];
        std::memset(&m, 0, sizeof m);
        for (int k = 0; k < 3; k++) {
            // tex < 0: (texel * 1.0) * col_mod hoisted (texture.rs:265, sphere.rs:145)
            m.cm[k] = s.tex < 0 ? host_mul(host_mul(s.tex_color[k], 1.0f), s.col_mod[k]) : s.col_mod[k];
            m.emitted[k] = s.emitted[k];
        }
        m.metallicness = s.metallicness; m.opacity = s.opacity; m.ir = s.ir; m.tex = s.tex;
        m.inv_ir = host_div(1.0f, s.ir);                         // materials.rs:113  1.0 / self.ir
        m.r0_front = schlick_r0(m.inv_ir); m.r0_back = schlick_r0(s.ir);   // materials.rs:99-100 for either ratio
    }
}

// Quad::new (quad.rs:84-110): n = u x v, normal = unit(n), d = normal . origin, w = n / (n . n) -- one rounding per
// written operation (this TU is built with -ffp-contract=off), the oracle computes the same independently.
static void prepare_quads(const RtwQuad *q, uint32_t n, std::vector<DevQuad> &out) {
    out.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        const RtwQuad &s = q[i];
        DevQuad &d = out[i];
        std::memset(&d, 0, sizeof d);
        const float nx = s.u[1] * s.v[2] - s.u[2] * s.v[1], ny = s.u[2] * s.v[0] - s.u[0] * s.v[2], nz = s.u[0] * s.v[1] - s.u[1] * s.v[0];
        const float len = std::sqrt(nx * nx + ny * ny + nz * nz);
        d.normal[0] = nx / len; d.normal[1] = ny / len; d.normal[2] = nz / len;
        d.d = d.normal[0] * s.origin[0] + d.normal[1] * s.origin[1] + d.normal[2] * s.origin[2];
        const float nn = nx * nx + ny * ny + nz * nz;
        d.w[0] = nx / nn; d.w[1] = ny / nn; d.w[2] = nz / nn;
        for (int k = 0; k < 3; k++) {
            d.origin[k] = s.origin[k]; d.u[k] = s.u[k]; d.v[k] = s.v[k];
            d.albedo[k] = host_mul(s.tex_color[k], 1.0f);            // texture.rs:265
            d.emitted[k] = s.emitted[k];
        }
        d.metallicness = s.metallicness; d.opacity = s.opacity; d.ir = s.ir; d.tex = s.tex;
    }
}

template <class T>
static int upload(void **dst, const std::vector<T> &src) {
    size_t bytes = src.size() * sizeof(T);
    if (bytes == 0) bytes = sizeof(T);
    HIP_TRY(hipMalloc(dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RTW_OK;
}

extern "C" {

int rtw_abi_version(void) { return RTW_ABI_VERSION; }

int rtw_last_hip_error(void) { return g_last_hip; }

int rtw_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *rtw_strerror(int status) {
    switch (status) {
    case RTW_OK: return "ok";
    case RTW_E_INVALID: return "invalid argument";
    case RTW_E_NO_DEVICE: return "no HIP device";
    case RTW_E_HIP: return "HIP runtime error";
    case RTW_E_NOMEM: return "out of memory";
    case RTW_E_UNSUPPORTED: return "not implemented on the device";
    case RTW_E_NO_SCENE: return "no scene set";
    default: return "unknown status";
    }
}

int rtw_ctx_create(int device, rtw_ctx **out) {
    if (!out) return RTW_E_INVALID;
    *out = nullptr;
    int n = rtw_device_count();
    if (n <= 0 || device < 0 || device >= n) return RTW_E_NO_DEVICE;
    rtw_ctx *c = new (std::nothrow) rtw_ctx();
    if (!c) return RTW_E_NOMEM;
    c->device = device;
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) { c->n_cu = prop.multiProcessorCount; e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking); }
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_queue, 64);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_stats, 16 * sizeof(unsigned long long));
    if (e != hipSuccess) { g_last_hip = (int)e; rtw_ctx_destroy(c); return RTW_E_HIP; }
    c->stream = c->own_stream;
    *out = c;
    return RTW_OK;
}

static void free_scene(rtw_ctx *c) {
    void **bufs[] = { &c->d_quads, &c->d_inst, &c->d_igeom, &c->d_ivel, &c->d_imat, &c->d_iquads,
                      &c->d_geom, &c->d_vel, &c->d_mat, &c->d_tex, &c->d_texels, &c->d_nodes, &c->d_nodes16, &c->d_big_geom, &c->d_big_vel, &c->d_big_index };
    for (void **b : bufs) { if (*b) (void)hipFree(*b); *b = nullptr; }
    c->has_scene = false;
}

void rtw_ctx_destroy(rtw_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    free_scene(c);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_samples) (void)hipFree(c->d_samples);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int rtw_ctx_set_stream(rtw_ctx *c, void *hip_stream) {
    if (!c) return RTW_E_INVALID;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RTW_OK;
}

int rtw_ctx_set_scene(rtw_ctx *c, const RtwScene *s, float t_begin, float t_end) {
    if (!c || !s || (s->n_spheres && !s->spheres)) return RTW_E_INVALID;
    if ((s->n_textures && !s->textures) || (s->n_texels && !s->texels)) return RTW_E_INVALID;
    if ((s->n_quads && !s->quads) || (s->n_instances && !s->instances)) return RTW_E_INVALID;
    if ((s->n_inst_spheres && !s->inst_spheres) || (s->n_inst_quads && !s->inst_quads)) return RTW_E_INVALID;
    for (uint32_t i = 0; i < s->n_spheres; i++)
        if (s->spheres[i].tex >= (int32_t)s->n_textures) return RTW_E_INVALID;
    for (uint32_t i = 0; i < s->n_inst_spheres; i++)
        if (s->inst_spheres[i].tex >= (int32_t)s->n_textures) return RTW_E_INVALID;
    for (uint32_t i = 0; i < s->n_quads; i++)
        if (s->quads[i].tex >= (int32_t)s->n_textures) return RTW_E_INVALID;
    for (uint32_t i = 0; i < s->n_inst_quads; i++)
        if (s->inst_quads[i].tex >= (int32_t)s->n_textures) return RTW_E_INVALID;
    for (uint32_t i = 0; i < s->n_instances; i++) {
        const RtwInstance &in = s->instances[i];
        if ((uint64_t)in.first_sphere + in.n_spheres > s->n_inst_spheres) return RTW_E_INVALID;
        if ((uint64_t)in.first_quad + in.n_quads > s->n_inst_quads) return RTW_E_INVALID;
        if (in.medium > RTW_MEDIUM_CONST_DENSITY) return RTW_E_INVALID;
    }
    for (uint32_t i = 0; i < s->n_textures; i++) {
        const RtwTexture &t = s->textures[i];
        if (t.row == 0 || t.col == 0 || (uint64_t)t.texel_offset + (uint64_t)t.row * t.col > s->n_texels) return RTW_E_INVALID;
    }
    HIP_TRY(hipSetDevice(c->device));
    free_scene(c);

    std::vector<f4> geom, vel;
    std::vector<DevMat> mat;
    bool moving = false;
    prepare_spheres(s->spheres, s->n_spheres, geom, vel, mat, moving);
    std::vector<RtwTexture> tex(s->textures, s->textures + s->n_textures);
    std::vector<float> texels(s->texels, s->texels + 3 * (size_t)s->n_texels);

    int rc;
    if ((rc = upload(&c->d_geom, geom)) || (rc = upload(&c->d_vel, vel)) || (rc = upload(&c->d_mat, mat)) ||
        (rc = upload(&c->d_tex, tex)) || (rc = upload(&c->d_texels, texels))) { free_scene(c); return rc; }

    // quads and instances (Scene::new, viewport.rs:122-135).  Their AABB trees (qaabb.rs, iaabb.rs) only prune list
    // walks over a handful of objects: the device walks the lists.
    {
        std::vector<DevQuad> quads, iquads;
        std::vector<f4> igeom, ivel; std::vector<DevMat> imat;
        std::vector<DevInstance> inst(s->n_instances);
        bool imoving = false;
        prepare_quads(s->quads, s->n_quads, quads);
        prepare_quads(s->inst_quads, s->n_inst_quads, iquads);
        prepare_spheres(s->inst_spheres, s->n_inst_spheres, igeom, ivel, imat, imoving);
        for (uint32_t i = 0; i < s->n_instances; i++) {
            const RtwInstance &in = s->instances[i];
            DevInstance &d = inst[i];
            std::memset(&d, 0, sizeof d);
            d.first_sphere = in.first_sphere; d.n_spheres = in.n_spheres; d.first_quad = in.first_quad; d.n_quads = in.n_quads;
            d.density = in.density; d.medium = in.medium;
            for (int k = 0; k < 3; k++) {
                d.tr[k] = in.translation[k];
                d.back[2 * k] = std::sin(-in.rotation[k]); d.back[2 * k + 1] = std::cos(-in.rotation[k]);   // rotated(-self.rotation) (instance.rs:258)
                d.fwd[2 * k] = std::sin(in.rotation[k]);   d.fwd[2 * k + 1] = std::cos(in.rotation[k]);     // vec3.rs:163-170
            }
        }
        if ((rc = upload(&c->d_quads, quads)) || (rc = upload(&c->d_iquads, iquads)) || (rc = upload(&c->d_inst, inst)) ||
            (rc = upload(&c->d_igeom, igeom)) || (rc = upload(&c->d_ivel, ivel)) || (rc = upload(&c->d_imat, imat))) { free_scene(c); return rc; }
        c->geom.quads = (const DevQuad *)c->d_quads; c->geom.iquads = (const DevQuad *)c->d_iquads;
        c->geom.inst = (const DevInstance *)c->d_inst;
        c->geom.igeom = (const f4 *)c->d_igeom; c->geom.ivel = (const f4 *)c->d_ivel; c->geom.imat = (const DevMat *)c->d_imat;
        c->geom.n_quads = s->n_quads; c->geom.n_inst = s->n_instances;
    }

    // acceleration structure (Scene::new_sphere builds the AABB tree, viewport.rs:90-105)
    BvhBuild bb;
    build_bvh(s->spheres, s->n_spheres, std::fmin(t_begin, t_end), std::fmax(t_begin, t_end), bb);
    std::vector<f4> big_geom, big_vel;
    for (uint32_t i : bb.big) { big_geom.push_back(geom[i]); big_vel.push_back(vel[i]); }
    if ((rc = upload(&c->d_nodes16, bb.nodes16))) { free_scene(c); return rc; }
    c->bvh.nodes16 = bb.nodes16.empty() ? nullptr : (const BvhNode16 *)c->d_nodes16;
    c->bvh.n_nodes = (uint32_t)bb.nodes.size();
    if ((rc = upload(&c->d_nodes, bb.nodes)) || (rc = upload(&c->d_big_geom, big_geom)) ||
        (rc = upload(&c->d_big_vel, big_vel)) || (rc = upload(&c->d_big_index, bb.big))) { free_scene(c); return rc; }
    c->bvh.nodes = (const BvhNode *)c->d_nodes;
    c->bvh.big_geom = (const f4 *)c->d_big_geom; c->bvh.big_vel = (const f4 *)c->d_big_vel;
    c->bvh.big_index = (const uint32_t *)c->d_big_index; c->bvh.n_big = (uint32_t)bb.big.size();
    c->bvh.root = bb.root; c->bvh.depth = bb.depth;
    c->bvh.cx = bb.centre[0]; c->bvh.cy = bb.centre[1]; c->bvh.cz = bb.centre[2];
    c->bvh.centre_radius = bb.centre_radius;
    c->bvh.r_max2 = bb.r_max * bb.r_max * 1.000001f;
    c->bvh.inv_2rmin = bb.r_min > 0.0f ? 1.0f / (2.0f * bb.r_min) : INFINITY;
    c->bvh.abs_max = bb.abs_max;

    c->sc.geom = (const f4 *)c->d_geom; c->sc.vel = (const f4 *)c->d_vel; c->sc.mat = (const DevMat *)c->d_mat;
    c->sc.tex = (const RtwTexture *)c->d_tex; c->sc.texels = (const float *)c->d_texels;
    c->sc.n = s->n_spheres; c->sc.moving = moving ? 1u : 0u;
    std::memcpy(c->bg, s->background, sizeof c->bg);
    c->has_textures = false;
    for (uint32_t i = 0; i < s->n_spheres; i++) if (s->spheres[i].tex >= 0) c->has_textures = true;
    c->has_scene = true;
    return RTW_OK;
}

int rtw_ctx_render(rtw_ctx *c, const RtwCamera *cam, const RtwParams *p, float *out_rgb, RtwStats *stats) {
    if (!c || !cam || !p || !out_rgb) return RTW_E_INVALID;
    if (!c->has_scene) return RTW_E_NO_SCENE;
    if (p->width == 0 || p->height == 0 || p->samples == 0) return RTW_E_INVALID;
    if (p->integrator > RTW_INTEGRATOR_RUST2 || p->sampler > RTW_SAMPLER_NO_RAND || p->accel > RTW_ACCEL_BVH) return RTW_E_INVALID;
    if (p->part_count > 1 && (p->row_block == 0 || p->part_index >= p->part_count)) return RTW_E_INVALID;
    if ((uint64_t)p->width * p->height >= (1ull << 32)) return RTW_E_INVALID;
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipSetDevice(c->device));

    // The BVH's pruning is proven against the reference's ROUNDED quadratic (DESIGN.md "Conservative traversal"), which presumes
    // that d.d is an ordinary f32.  Scattered directions are unit-scale by construction; camera directions are whatever the caller's
    // camera says (the reference never normalises them), and when |d|^2 can overflow or underflow f32 the reference's quadratic
    // degenerates -- NaN roots pass both range tests and are reported as hits (sphere.rs:118-121) -- which only the list walk
    // reproduces.  Bound |d| over the image from the camera (d = pixel00 + x delta_u + y delta_v, x in [0,W], y in [0,H]):
    // above by the triangle inequality, below by the distance of that plane from the origin; outside [1e-15, 1e15] walk the list.
    uint32_t accel = p->accel;
    if (accel == RTW_ACCEL_BVH) {
        const double P[3] = { cam->pixel00[0], cam->pixel00[1], cam->pixel00[2] };
        const double U[3] = { cam->delta_u[0], cam->delta_u[1], cam->delta_u[2] }, V[3] = { cam->delta_v[0], cam->delta_v[1], cam->delta_v[2] };
        auto norm = [](const double *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
        const double W = (double)p->width + 1.0, H = (double)p->height + 1.0;
        const double hi = norm(P) + W * norm(U) + H * norm(V);
        const double N[3] = { U[1] * V[2] - U[2] * V[1], U[2] * V[0] - U[0] * V[2], U[0] * V[1] - U[1] * V[0] };
        const double nn = norm(N);
        const double lo = nn > 0.0 ? std::fabs(P[0] * N[0] + P[1] * N[1] + P[2] * N[2]) / nn : 0.0;
        if (!(hi <= 1e15) || !(lo >= 1e-15)) accel = RTW_ACCEL_BRUTE;
    }

    KArgs a;
    std::memset(&a, 0, sizeof a);
    a.cam = *cam; a.sc = c->sc; a.bvh = c->bvh; a.geom = c->geom;
    if (p->flags & RTW_FLAG_GLOBAL_NODES) a.bvh.nodes16 = nullptr;
    a.width = p->width; a.height = p->height;
    const uint32_t n_rows = rtw_part_rows(p->height, p->row_block, p->part_index, p->part_count);
    a.row_block = p->row_block ? p->row_block : 1; a.part_index = p->part_index; a.part_count = p->part_count;
    a.tiles_x = (p->width + 7) / 8;
    a.n_samples = sampler_count(p->sampler, p->samples, &a.s_root);
    if (a.n_samples == 0) return RTW_E_INVALID;
    a.sampler = p->sampler; a.integrator = p->integrator; a.depth = p->depth;
    a.has_textures = c->has_textures ? 1u : 0u;
    a.seed_lo = (uint32_t)p->seed; a.seed_hi = (uint32_t)(p->seed >> 32);
    a.inv_gamma = host_div(1.0f, p->gamma);                       // viewport.rs:232
    a.mint = p->mint; a.maxt = p->maxt;
    std::memcpy(a.bg, c->bg, sizeof a.bg);
    a.queue = c->d_queue; a.stats = c->d_stats;

    // Work units are (tile, chunk of samples, pixel): a pixel's samples are spread over ceil(n / chunk_len) units
    // so that no lane owns more than chunk_len sequential paths (the slowest PIXEL used to set a ~50 ms tail).
    // Every sample's radiance is banked in HBM and added in order by resolve_kernel: 12 B per camera ray
    // (12.4 GB for 1920x1080x500) -- the image is rendered in bands of tile rows when that exceeds the budget.
    uint32_t chunk_len = 4;          // tuned on the bench frame: 4-6 is the flat optimum (1: 13.5, 2: 17.0, 4: 17.9, 8: 17.3, 16: 16.8 Gsegments/s)
    if (const char *e = getenv("RTW_CHUNK")) { int v = atoi(e); if (v >= 1 && v <= 4096) chunk_len = (uint32_t)v; }
    if (chunk_len > a.n_samples) chunk_len = a.n_samples;
    a.chunk_len = chunk_len;
    a.n_chunks = (a.n_samples + chunk_len - 1) / chunk_len;
    const uint64_t slots_per_tile_row = (uint64_t)a.tiles_x * a.n_chunks * 64ull * chunk_len;
    uint64_t budget = 48ull << 30;
    if (const char *e = getenv("RTW_SAMPLE_BUF_GB")) { double v = atof(e); if (v > 0.0) budget = (uint64_t)(v * (double)(1ull << 30)); }
    uint64_t max_slots = budget / 12; if (max_slots > 0xFFFFFFF0ull) max_slots = 0xFFFFFFF0ull;
    const uint32_t tile_rows = (n_rows + 7) / 8;
    uint64_t band_tile_rows = max_slots / (slots_per_tile_row ? slots_per_tile_row : 1);
    if (band_tile_rows == 0) return RTW_E_NOMEM;                  // one row of tiles does not fit the budget
    if (band_tile_rows > tile_rows) band_tile_rows = tile_rows;
    const size_t sample_bytes = (size_t)(band_tile_rows * slots_per_tile_row) * 12;
    if (c->d_samples_cap < sample_bytes) {
        if (c->d_samples) (void)hipFree(c->d_samples);
        c->d_samples = nullptr; c->d_samples_cap = 0;
        if (hipMalloc((void **)&c->d_samples, sample_bytes ? sample_bytes : 12) != hipSuccess) { (void)hipGetLastError(); return RTW_E_NOMEM; }
        c->d_samples_cap = sample_bytes;
    }
    a.samples = c->d_samples;

    const size_t out_bytes = (size_t)n_rows * p->width * 3 * sizeof(float);
    hipPointerAttribute_t attr;
    bool out_on_device = false;
    if (hipPointerGetAttributes(&attr, out_rgb) == hipSuccess) {
        out_on_device = attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
    } else (void)hipGetLastError();
    if (out_on_device) a.out = out_rgb;
    else {
        if (c->d_out_cap < out_bytes) {
            if (c->d_out) (void)hipFree(c->d_out);
            c->d_out = nullptr; c->d_out_cap = 0;
            HIP_TRY(hipMalloc((void **)&c->d_out, out_bytes ? out_bytes : 4));
            c->d_out_cap = out_bytes;
        }
        a.out = c->d_out;
    }

    // dynamic LDS layout of the BVH kernel for this tree
    if (accel == RTW_ACCEL_BVH) {
        const bool ldsn = a.bvh.nodes16 != nullptr;
        // sentinel + one entry per tree level + the slot above the top the descend step always writes
        uint32_t levels = c->bvh.depth + 3; if (levels < 4) levels = 4; if (levels > RTW_BVH_STACK + 3) levels = RTW_BVH_STACK + 3;
        uint32_t off = levels * RTW_BLOCK * (ldsn ? 2u : 4u);
        off = (off + 15u) & ~15u;
        if (ldsn) {
            a.lds_nodes_off = off; off += c->bvh.n_nodes * 32u; off = (off + 15u) & ~15u;
            // sphere geometry rides along only while the workgroup stays under 1/6 of the CU's 160 KiB, i.e. while
            // it does not cost a resident workgroup at the kernel's register budget (6 waves/SIMD)
            bool geom = off + c->sc.n * 16u <= 160u * 1024u / 6u;
            if (const char *e = getenv("RTW_LDS_GEOM")) geom = atoi(e) != 0 && c->sc.n <= RTW_LDS_GEOM_MAX;
            if (geom) { a.lds_geom_off = off; off += c->sc.n * 16u; }
        }
        a.lds_bytes = off;
    }

    // persistent grid: as many workgroups as the kernel's registers let be resident, capped by the work
    uint32_t per_cu = kernel_blocks_per_cu(a, c->sc.moving != 0, accel, c->bvh.nodes16 != nullptr && !(p->flags & RTW_FLAG_GLOBAL_NODES));
    if (const char *e = getenv("RTW_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v <= 8) per_cu = (uint32_t)v; }   // occupancy experiments

    HIP_TRY(hipMemsetAsync(c->d_stats, 0, 16 * sizeof(unsigned long long), c->stream));
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (uint32_t tr0 = 0; tr0 < tile_rows || tr0 == 0; tr0 += (uint32_t)band_tile_rows) {
        const uint32_t tr1 = tr0 + (uint32_t)band_tile_rows < tile_rows ? tr0 + (uint32_t)band_tile_rows : tile_rows;
        a.k_base = tr0 * 8; a.k_end = tr1 * 8 < n_rows ? tr1 * 8 : n_rows;
        a.n_tiles = a.tiles_x * (tr1 - tr0);
        a.total_work = a.n_tiles * a.n_chunks * 64u;              // < 2^32 by the slot bound above
        uint32_t grid = (uint32_t)c->n_cu * per_cu;
        const uint32_t need = (a.total_work + RTW_BLOCK - 1) / RTW_BLOCK;
        if (grid > need) grid = need ? need : 1;
        HIP_TRY(hipMemsetAsync(c->d_queue, 0, 64, c->stream));
        if (a.n_tiles) launch_render(a, c->sc.moving != 0, accel, grid, c->stream);
        HIP_TRY(hipGetLastError());
        if (tile_rows == 0) break;
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    unsigned long long h_stats[16];
    HIP_TRY(hipMemcpyAsync(h_stats, c->d_stats, sizeof h_stats, hipMemcpyDeviceToHost, c->stream));
    if (!out_on_device) HIP_TRY(hipMemcpyAsync(out_rgb, c->d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->camera_rays = h_stats[0]; stats->segments = h_stats[1];
        stats->sphere_tests = h_stats[2]; stats->node_tests = h_stats[3];
        stats->nan_pixels = (uint32_t)h_stats[4]; stats->rows = n_rows;
        stats->quad_tests = h_stats[14];
        stats->kernel_ms = ms;
        for (int k = 0; k < 3; k++) { stats->phase_steps[k] = h_stats[5 + k]; stats->phase_lanes[k] = h_stats[8 + k]; }
        if (getenv("RTW_STAMP_DUMP")) std::fprintf(stderr, "rtw stamp: wave-ticks traverse %llu leaf %llu shade %llu\n", h_stats[11], h_stats[12], h_stats[13]);
        if (getenv("RTW_ENDTIMES_DUMP") && h_stats[15]) {   // RTW_ENDTIMES build: [12] longest wave lifetime [13] sum of wave lifetimes [15] waves
            std::fprintf(stderr, "rtw endtimes: %llu waves, longest lifetime %llu ticks, mean lifetime %.1f %% of it\n", h_stats[15], h_stats[12],
                         100.0 * (double)h_stats[13] / (double)h_stats[15] / (double)h_stats[12]);
        }
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    return RTW_OK;
}

int rtw_ctx_render_multi(rtw_ctx *c, const RtwCamera *cam, const RtwParams *p, float fps, uint32_t start_frame, uint32_t n_frames,
                         float *out_rgb, RtwStats *stats) {
    if (!c || !cam || !p || !out_rgb || !(fps > 0.0f)) return RTW_E_INVALID;
    const size_t frame_floats = (size_t)rtw_part_rows(p->height, p->row_block, p->part_index, p->part_count) * p->width * 3;
    for (uint32_t i = 0; i < n_frames; i++) {                     // viewport.rs:256-266, one async_render per frame
        RtwCamera cm = *cam;
        cm.time0 = host_div((float)(start_frame + i), fps);       // viewport.rs:279  frame as f32 / fps
        int rc = rtw_ctx_render(c, &cm, p, out_rgb + (size_t)i * frame_floats, stats ? stats + i : nullptr);
        if (rc != RTW_OK) return rc;
    }
    return RTW_OK;
}

int rtw_render(const RtwCamera *cam, const RtwScene *scene, const RtwParams *params, float *out_rgb, RtwStats *stats) {
    if (!cam || !scene || !params) return RTW_E_INVALID;
    int dev = 0;
    if (rtw_device_count() <= 0) return RTW_E_NO_DEVICE;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    rtw_ctx *c = nullptr;
    int rc = rtw_ctx_create(dev, &c);
    if (rc != RTW_OK) return rc;
    rc = rtw_ctx_set_scene(c, scene, cam->time0, cam->time0 + cam->shutter);
    if (rc == RTW_OK) rc = rtw_ctx_render(c, cam, params, out_rgb, stats);
    rtw_ctx_destroy(c);
    return rc;
}

} // extern "C"
