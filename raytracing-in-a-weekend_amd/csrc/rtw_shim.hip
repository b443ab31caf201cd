// rtw_shim.hip -- extern "C" entry points of librtw_hip.so (include/rtw.h): context, scene upload,
// blocking render.  Host code only; kernels live in rtw_kernels.hip.  No CPU fallback of any kind:
// without a HIP device every render entry point fails with RTW_E_NO_DEVICE / RTW_E_HIP.
#include "rtw_kernels.h"
#include "rtw_host.h"

#include <link.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <new>
#include <utility>
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
    hipEvent_t ev_mark = nullptr;        // "the call began", recorded on this context's stream before anything of the call is launched anywhere (RtwStats.start_ms)
    // scene
    bool has_scene = false;
    bool has_textures = false;
    bool bvh_ok = true;                  // false: the tree is deeper than the device stack (never with build_bvh's invariant; checked anyway)
    double scene_span = 0.0;             // largest |coordinate| a sphere's centre reaches over [t_begin, t_end]; +inf when a centre, velocity or radius is not a number
                                         // or a centre / velocity is infinite (rtw_ctx_render: only ordinary scenes go through the tree)
    float t_begin = 0.0f, t_end = 0.0f;  // ray.time range the BVH bounds were expanded for (rtw_ctx_set_scene)
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
    unsigned long long *h_stats = nullptr;   // pinned: the counter read-back is a true async copy
    float *d_out = nullptr;
    size_t d_out_cap = 0;
    float *h_out = nullptr;              // pinned staging for a multi-GPU frame in PAGEABLE host memory: a device-to-host copy into pageable memory returns only
    size_t h_out_cap = 0;                // when it is done, which would hold the fork of rtw_mgpu_render at this device until it has rendered its share
    uint32_t *d_order = nullptr;         // RTW_OPT_TILE_ORDER: queue position -> tile, valid for order_key
    size_t d_order_cap = 0;
    TileOrderKey order_key{};
    SceneCull cull{};                    // what the tile-order heuristic knows about the scene (host copy, set by rtw_ctx_set_scene)
    uint32_t scene_serial = 0;           // bumped by every rtw_ctx_set_scene
    float *d_samples = nullptr;          // per-sample radiance bank (see render_enqueue)
    size_t d_samples_cap = 0;
    // options (rtw_ctx_set_option)
    uint32_t opt_chunk_len = 0;          // 0 = auto (render_enqueue); round 1 tuned a fixed 4 on the bench frame (1: 13.5, 2: 17.0, 4: 17.9, 8: 17.3, 16: 16.8 Gsegments/s)
    uint64_t opt_bank_bytes = 48ull << 30;
    int opt_lds_geom = -1;
    uint32_t opt_blocks_per_cu = 0;
    uint32_t opt_list_walk_max = RTW_LIST_WALK_MAX_DEFAULT;
    uint32_t opt_tile_order = 0;         // raster.  (The cost order, 2, was the default while units were 4 samples and grabs single blocks -- profiles/
                                         // r02_order_chunk_grid.log --; with today's units and grabs raster is 1.2 % ahead of it on the bench frame and ahead
                                         // on every other config too, profiles/r02_order_ab.log)
    uint32_t opt_sub_queues = 0;         // RTW_OPT_SUB_QUEUES: 0 = eight sub-queues (one per XCD) for all but tiny launches, 1 = a single queue
    double opt_tail_units = 0.0;         // RTW_OPT_TAIL_UNITS: blocks of short units per resident wave at the end of the queue (render_enqueue); 0 = off.
                                         // OFF by default: measured, it does not shorten a launch (profiles/r03_tail_units.log, r03_endtimes.log)
    uint32_t opt_grab_blocks = 2;        // RTW_OPT_GRAB_BLOCKS (profiles/r02_grab_sweep.log: 1 / 2 / 4 / 8 / a tile's 42 blocks = 83.9 / 82.9 / 83.7 / 86.8 / 107.9 ms on the bench frame)

    // cache of a per-call driver query (tens of microseconds: visible on small frames)
    std::map<std::pair<const void *, uint32_t>, uint32_t> occupancy;    // (kernel, dynamic LDS bytes) -> resident workgroups per CU
    bool attr_on_device = false; int attr_device = -1;   // memory kind of this call's out_rgb
    // the render in flight between render_enqueue and render_wait
    struct Pending {
        bool active = false;
        bool marked = false;             // ev_mark was recorded for this call (by rtw_mgpu_render, before any device was given work)
        uint32_t n_rows = 0;
        std::chrono::steady_clock::time_point t0;      // the call's begin on the host (rtw_mgpu_render: the same instant for every device)
        float enqueue_ms = 0.0f;         // host time from t0 until this context's launches had been issued
        // where the rendered rows go (render_copy / render_wait)
        float *dst = nullptr; bool scatter = false, direct = false, staged = false;
        uint32_t width = 0, height = 0, row_block = 1, part_index = 0, part_count = 1;
        size_t out_bytes = 0;
    } pend;
};

// Where a render's rows go.
struct OutSpec {
    float *base;        // compact [rows][width][3] buffer, or -- scatter -- the full [height][width][3] frame
    bool scatter;       // copy every row block to its image rows of the full frame (multi-GPU: no gather buffer, no de-interleave pass)
};

// The device records of a sphere list (sphere.rs:13-20 + materials.rs:15-20)
static void prepare_spheres(const RtwSphere *sp, uint32_t n, const RtwTexture *tex, std::vector<f4> &geom, std::vector<f4> &vel, std::vector<DevMat> &mat, bool &moving) {
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
        // the image's shape and place ride along in the material record: the texel fetch then depends on ONE load (this record), not on a second one
        // of the RtwTexture behind it (a SHADE step of a textured scene was three dependent global loads deep: material -> texture -> texel)
        if (s.tex >= 0 && tex) { m.tex_row = tex[s.tex].row; m.tex_col = tex[s.tex].col; m.tex_offset = tex[s.tex].texel_offset; }
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

// How many copies of the HIP runtime are mapped into this process.  librtw_hip.so links /opt/rocm's libamdhip64; PyTorch ships its own.  When torch
// is imported FIRST the dynamic linker gives this library the copy that is already loaded (same soname) and all is well; when this library is
// loaded first, a later `import torch` brings up a SECOND runtime whose device enumeration fails ("No HIP GPUs are available",
// gpurun_out/r02_debug_torch.log) -- torch is then blind for the rest of the process.  A context created in such a process fails loudly instead.
static int count_hip_runtimes_cb(struct dl_phdr_info *info, size_t, void *data) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "libamdhip64")) ++*(int *)data;
    return 0;
}

extern "C" {

int rtw_abi_version(void) { return RTW_ABI_VERSION; }

int rtw_hip_runtime_count(void) {
    int n = 0;
    dl_iterate_phdr(count_hip_runtimes_cb, &n);
    return n;
}

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
    case RTW_E_INTERNAL: return "a render kernel gave up (internal error): the image is incomplete";
    case RTW_E_RUNTIME_CONFLICT: return "two copies of the HIP runtime (libamdhip64) are loaded in this process -- typically PyTorch imported AFTER the first rtw_* call: "
                                        "import torch before using this library (the runtimes then share one copy), or link both against the same runtime";
    default: return "unknown status";
    }
}

int rtw_ctx_create(int device, rtw_ctx **out) {
    if (!out) return RTW_E_INVALID;
    *out = nullptr;
    if (rtw_hip_runtime_count() > 1) return RTW_E_RUNTIME_CONFLICT;
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
    if (e == hipSuccess) e = hipEventCreate(&c->ev_mark);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_queue, RTW_QUEUE_BYTES);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_stats, RTW_N_STATS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->h_stats, RTW_N_STATS * sizeof(unsigned long long), hipHostMallocDefault);
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
    if (c->h_stats) (void)hipHostFree(c->h_stats);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->ev_mark) (void)hipEventDestroy(c->ev_mark);
    if (c->d_samples) (void)hipFree(c->d_samples);
    if (c->d_order) (void)hipFree(c->d_order);
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
        if (t.emit_tex > s->n_textures) return RTW_E_INVALID;        // 1 + index of Rust2's emission image, 0 = none
    }
    HIP_TRY(hipSetDevice(c->device));
    free_scene(c);

    std::vector<f4> geom, vel;
    std::vector<DevMat> mat;
    bool moving = false;
    prepare_spheres(s->spheres, s->n_spheres, s->textures, geom, vel, mat, moving);
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
        prepare_spheres(s->inst_spheres, s->n_inst_spheres, s->textures, igeom, ivel, imat, imoving);
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
            rotation_coefficients(d.back, d.back_k); rotation_coefficients(d.fwd, d.fwd_k);
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
    c->bvh_ok = bb.depth <= RTW_BVH_STACK;        // build_bvh guarantees it; a deeper tree would overflow the per-lane LDS stack
    {   // The tree prunes on the premise that the reference's quadratic is computed in ORDINARY f32 (DESIGN.md "Conservative traversal").  A centre at 1e19 or
        // beyond (|oc|^2 overflows), or a NaN anywhere, makes its discriminant NaN, and a NaN root passes both range tests of sphere.rs:118-121: such a sphere is
        // "hit" by every ray that reaches it in list order with nothing accepted before -- an order-dependent answer only the list walk gives.  Record how far the
        // centres reach; rtw_ctx_render decides with the camera in hand.  (Radii: +-inf and huge ones are fine -- those spheres are tested exactly for every
        // query, and r^2 = inf gives -inf / +inf roots, no NaN; a NaN radius is not.)
        double span = 0.0;
        const double tmax = std::fmax(std::fabs((double)t_begin), std::fabs((double)t_end));
        for (uint32_t i = 0; i < s->n_spheres; i++) {
            const RtwSphere &q = s->spheres[i];
            double reach = 0.0;
            bool odd = q.radius != q.radius;
            // ... and a material that can turn a ray into one: the scattered direction is reflect * m + scatter * (1 - m), or a refraction with ratio ir or 1 / ir
            // (materials.rs:105-154) -- finite and of ordinary size exactly when m and ir are (a NaN or 1e30 there makes the NEXT ray NaN or |d|^2 overflow)
            {
                const double m = std::fabs((double)q.metallicness), ir = std::fabs((double)q.ir);
                if (!(m <= 1e10)) odd = true;
                if (q.opacity > 0.0f && !(ir >= 1e-10 && ir <= 1e10)) odd = true;         // (ir is read by the dielectric branch only)
            }
            for (int k = 0; k < 3; k++) {
                const double a = std::fabs((double)q.center[k]), v = std::fabs((double)q.velocity[k]);
                if (!(a <= 1.7e308) || !(v <= 1.7e308)) odd = true;                   // NaN or inf (NaN compares false: the negated tests catch it; fmax would drop it)
                else reach = std::fmax(reach, a + v * tmax);
            }
            span = std::fmax(span, odd ? HUGE_VAL : reach);
        }
        c->scene_span = span;
    }
    c->scene_serial++;
    scene_cull_from_bvh(bb, s->spheres, c->cull);
    c->cull.n_other = s->n_quads + s->n_instances;
    c->t_begin = std::fmin(t_begin, t_end); c->t_end = std::fmax(t_begin, t_end);
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

} // extern "C"

// Copy the compact rows of partition (row_block, part_index, part_count) from `src` (device) to their image rows of the
// full frame `dst` (host, or device memory of any GPU): the blocks a partition owns are equally spaced in the frame, so ONE
// strided 2-D copy moves all the full blocks, and a second, 1-D, the ragged last block when this partition owns it.
static int scatter_rows(rtw_ctx *c, const float *src, float *dst, uint32_t width, uint32_t height, uint32_t row_block,
                        uint32_t part_index, uint32_t part_count) {
    const size_t row_bytes = (size_t)width * 3 * sizeof(float);
    if (part_count <= 1) { HIP_TRY(hipMemcpyAsync(dst, src, row_bytes * height, hipMemcpyDefault, c->stream)); return RTW_OK; }
    const uint32_t n_blocks = (height + row_block - 1) / row_block;
    uint32_t mine = 0, full = 0;                                    // blocks b = part_index + k * part_count < n_blocks
    if (part_index < n_blocks) mine = (n_blocks - 1 - part_index) / part_count + 1;
    const bool ragged = height % row_block != 0;
    const bool own_last = mine > 0 && (part_index + (mine - 1) * part_count) == n_blocks - 1;
    full = mine - ((ragged && own_last) ? 1u : 0u);
    const size_t block_bytes = row_bytes * row_block;
    if (full) HIP_TRY(hipMemcpy2DAsync((char *)dst + (size_t)part_index * block_bytes, (size_t)part_count * block_bytes, src, block_bytes,
                                       block_bytes, full, hipMemcpyDefault, c->stream));
    if (full < mine) {
        const uint32_t b = part_index + full * part_count;
        HIP_TRY(hipMemcpyAsync((char *)dst + (size_t)b * block_bytes, (const char *)src + (size_t)full * block_bytes,
                               row_bytes * (height - b * row_block), hipMemcpyDefault, c->stream));
    }
    return RTW_OK;
}

// ... the same on the host, from the pinned staging buffer into a pageable frame (render_wait)
static void scatter_rows_host(const float *src, float *dst, uint32_t width, uint32_t height, uint32_t n_rows, uint32_t row_block, uint32_t part_index, uint32_t part_count) {
    const size_t row_floats = (size_t)width * 3;
    if (part_count <= 1) { std::memcpy(dst, src, row_floats * height * sizeof(float)); return; }
    for (uint32_t k = 0; k < n_rows; k += row_block) {               // compact rows [k, k + rows) are image rows [j, j + rows)
        const uint32_t j = ((k / row_block) * part_count + part_index) * row_block;
        const uint32_t rows = std::min(row_block, std::min(n_rows - k, height - j));
        std::memcpy(dst + (size_t)j * row_floats, src + (size_t)k * row_floats, rows * row_floats * sizeof(float));
    }
}

// A render is three phases: PREPARE (argument checks, first-use allocations, driver queries: may wait for the device), LAUNCH (memsets, kernels,
// events, counter read-back: never waits), COPY (the rows toward the caller's memory: never waits for the device either -- a pageable multi-GPU
// host frame is staged through pinned memory and finished on the host in render_wait).  rtw_ctx_render runs them back to back; rtw_mgpu_render
// runs each phase for EVERY device before the next phase for any, so that no device's work is held up by another's.
enum : unsigned { PH_PREPARE = 1u, PH_LAUNCH = 2u, PH_COPY = 4u, PH_ALL = 7u };
static int render_enqueue_impl(rtw_ctx *c, const RtwCamera *cam, const RtwParams *p, OutSpec out, unsigned phases);
static int render_copy(rtw_ctx *c);

// Launch one render on the context's stream; nothing here waits for the GPU (apart from a first-use hipMalloc).  On failure nothing of
// this call is left running: whatever was already launched is waited for, so the caller may free or reuse its buffers at once.
static int render_enqueue(rtw_ctx *c, const RtwCamera *cam, const RtwParams *p, OutSpec out, unsigned phases = PH_ALL) {
    const int rc = render_enqueue_impl(c, cam, p, out, phases);
    if (rc != RTW_OK && c && c->stream && !c->pend.active) {
        const int keep = g_last_hip;
        if (hipStreamSynchronize(c->stream) != hipSuccess) (void)hipGetLastError();
        g_last_hip = keep;
    }
    return rc;
}

static int render_enqueue_impl(rtw_ctx *c, const RtwCamera *cam, const RtwParams *p, OutSpec out, unsigned phases) {
    if (!c || !cam || !p || !out.base) return RTW_E_INVALID;
    if (c->pend.active) return RTW_E_INVALID;
    if (!c->has_scene) return RTW_E_NO_SCENE;
    if (p->width == 0 || p->height == 0 || p->samples == 0) return RTW_E_INVALID;
    if (p->integrator > RTW_INTEGRATOR_RUST2 || p->sampler > RTW_SAMPLER_NO_RAND || p->accel > RTW_ACCEL_BVH) return RTW_E_INVALID;
    if (p->part_count > 1 && (p->row_block == 0 || p->part_index >= p->part_count)) return RTW_E_INVALID;
    if (p->width > 65535u || p->height > 65535u) return RTW_E_INVALID;      // a lane keeps (column, row) in one register (rtw_kernels.hip Pixel)
    if (!c->pend.marked) c->pend.t0 = std::chrono::steady_clock::now();
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
        // ... and that the origin is one: past 2^60 (or NaN) the squares of the quadratic overflow and the same NaN roots appear
        const double O[3] = { cam->origin[0], cam->origin[1], cam->origin[2] }, CU[3] = { cam->u[0], cam->u[1], cam->u[2] }, CV[3] = { cam->v[0], cam->v[1], cam->v[2] };
        if (!(norm(O) + std::fabs((double)cam->lens_radius) * (norm(CU) + norm(CV)) <= 0x1p60)) accel = RTW_ACCEL_BRUTE;
        // ... and that the scene is one (rtw_ctx_set_scene: scene_span): b = oc . d and a c = (d . d)(oc . oc - r^2) must stay below f32's 3.4e38, so
        // |d| x (the farthest centre + the camera's distance) is kept below 1e18 (camera rays: `hi` above; scattered rays are unit-scale)
        if (!(std::fmax(hi, 2.0) * (c->scene_span + norm(O) + 1.0) <= 1e18)) accel = RTW_ACCEL_BRUTE;
        // ... and the accepted range: with a NaN bound the reference's `x < mint || x > maxt` rejects nothing, where the tree's interval test rejects everything
        if (p->mint != p->mint || p->maxt != p->maxt) accel = RTW_ACCEL_BRUTE;
        // The tree's bounds cover ray.time in [t_begin, t_end] only (rtw_ctx_set_scene): outside it a moving sphere can leave its box
        if (c->sc.moving) {
            const float ta = cam->time0, tb = cam->time0 + cam->shutter;
            if (!(std::fmin(ta, tb) >= c->t_begin && std::fmax(ta, tb) <= c->t_end)) accel = RTW_ACCEL_BRUTE;
            if (ta != ta || tb != tb) accel = RTW_ACCEL_BRUTE;         // (fmin / fmax drop a NaN: a NaN shutter makes every moving centre NaN, which only the list walk answers like the reference)
        }
        if (!c->bvh_ok) accel = RTW_ACCEL_BRUTE;
        // a handful of spheres: the list walk IS the fastest closest-hit (the traversal scheduler only costs)
        if (c->sc.n <= c->opt_list_walk_max) accel = RTW_ACCEL_BRUTE;
    }

    KArgs a;
    std::memset(&a, 0, sizeof a);
    a.cam = *cam; a.sc = c->sc; a.bvh = c->bvh; a.geom = c->geom;
    if (p->flags & RTW_FLAG_GLOBAL_NODES) a.bvh.nodes16 = nullptr;
    a.flags = p->flags;
    a.width = p->width; a.height = p->height;
    const uint32_t n_rows = rtw_part_rows(p->height, p->row_block, p->part_index, p->part_count);
    a.row_block = p->row_block ? p->row_block : 1; a.part_index = p->part_index; a.part_count = p->part_count;
    a.tiles_x = (p->width + 7) / 8;
    a.n_samples = sampler_count(p->sampler, p->samples, &a.s_root);
    if (a.n_samples == 0 || a.n_samples >= (1u << 24)) return RTW_E_INVALID;   // (sample index and samples left in the unit share a register)
    a.sampler = p->sampler; a.integrator = p->integrator; a.depth = p->depth;
    a.has_textures = c->has_textures ? 1u : 0u;
    a.seed_lo = (uint32_t)p->seed; a.seed_hi = (uint32_t)(p->seed >> 32);
    a.inv_gamma = host_div(1.0f, p->gamma);                       // viewport.rs:232
    a.mint = p->mint; a.maxt = p->maxt;
    std::memcpy(a.bg, c->bg, sizeof a.bg);
    a.queue = c->d_queue; a.stats = c->d_stats;
#ifdef RTW_ENDTIMES
    if (const char *e = getenv("RTW_ENDTIMES_REF")) a.endtimes_ref = std::strtoull(e, nullptr, 10);       // diagnostic build only
#endif

    // Work units are (tile, chunk of samples, pixel): a pixel's samples are spread over ceil(n / chunk_len) units
    // so that no lane owns more than chunk_len sequential paths (the slowest PIXEL used to set a ~50 ms tail).
    // Every sample's radiance is banked in HBM and added in order by resolve_kernel: 12 B per camera ray
    // (12.4 GB for 1920x1080x500) -- the image is rendered in bands of tile rows when that exceeds the budget.
    // RTW_FLAG_CHUNK_SUMS banks one partial sum per unit instead (bank_len = 1: one slot per unit and pixel).
    const bool sums = (p->flags & RTW_FLAG_CHUNK_SUMS) != 0;
    uint32_t chunk_len = (p->flags & RTW_FLAG_CHUNK_SUMS) ? RTW_SUM_CHUNK : c->opt_chunk_len;   // (the summation chunk is part of the image's definition)
    if (chunk_len == 0) {
        // auto: the unit length follows the size of the launch (profiles/r02_order_chunk_grid.log).  Long units amortise the per-unit work
        // (queue, pixel hash, bank addressing) but lengthen the tail of the launch, which is one unit deep: the whole bench frame
        // (660 units of 4 per lane) runs in 101.0 / 98.5 / 97.4 ms with units of 4 / 6 / 8, an eighth of it (82 per lane) in 13.46 /
        // 13.34 / 13.55, C2 (51 per lane) in 8.97 / 9.17 / 9.37.  A frame so small that one workgroup per CU is enough (below) gets
        // units of 2 (C1: 0.276 -> 0.261 ms).
        const uint64_t units4 = (uint64_t)a.tiles_x * ((n_rows + 7) / 8) * ((a.n_samples + 3) / 4) * 64ull;
        const uint64_t lanes = (uint64_t)c->n_cu * 1536ull;       // 24 resident waves per CU (6 per SIMD: the register budget of the specialised builds)
        // (units of 12 from 500 per lane: bench frame 97.8 -> 97.0 ms, C4 386.1 -> 383.7; 16..32 gain C4 another 0.6 % and lose the bench frame 0.3..0.9 %)
        chunk_len = units4 >= 500ull * lanes ? 12u : units4 >= 200ull * lanes ? 8u : units4 >= 64ull * lanes ? 6u : (units4 * 6ull < 16ull * lanes ? 2u : 4u);
    }
    if (chunk_len > a.n_samples) chunk_len = a.n_samples;
    a.chunk_len = chunk_len;
    a.n_chunks = (a.n_samples + chunk_len - 1) / chunk_len;
    a.bank_len = sums ? 1u : 0u;
    // the bank: one slot per (tile, sample, pixel) -- whatever the unit lengths --, or per (tile, unit, pixel) for the partial sums
    const uint64_t slots_per_tile_row = (uint64_t)a.tiles_x * (sums ? a.n_chunks : a.n_samples) * 64ull;
    uint64_t max_slots = c->opt_bank_bytes / 12; if (max_slots > 0xFFFFFFF0ull) max_slots = 0xFFFFFFF0ull;
    // (work items are counted in 32 bits as well: 64 per tile and unit, at most one unit per sample)
    const uint64_t items_per_tile_row = slots_per_tile_row;
    const uint32_t tile_rows = (n_rows + 7) / 8;
    uint64_t band_tile_rows = max_slots / (slots_per_tile_row ? slots_per_tile_row : 1);
    if (band_tile_rows > 0xFFFFFFF0ull / items_per_tile_row) band_tile_rows = 0xFFFFFFF0ull / items_per_tile_row;
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

    // destination: the kernels write compact rows either straight into the caller's device buffer or into the context's
    const size_t out_bytes = (size_t)n_rows * p->width * 3 * sizeof(float);
    // (asked on every call, a few microseconds: the same address can be host memory in one call and device memory in the next)
    bool pinned_host = false;
    {
        hipPointerAttribute_t attr;
        c->attr_on_device = false; c->attr_device = -1;
        if (hipPointerGetAttributes(&attr, out.base) == hipSuccess) {
            c->attr_on_device = attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
            pinned_host = attr.type == hipMemoryTypeHost;          // hipHostMalloc / hipHostRegister: a copy into it is a true asynchronous DMA
            c->attr_device = attr.device;
        } else (void)hipGetLastError();
    }
    const bool direct = c->attr_on_device && c->attr_device == c->device && !(out.scatter && p->part_count > 1);
    // A multi-GPU frame in pageable host memory (the reference's Img, a Vec, a numpy array): staged through this context's pinned buffer
    const bool staged = out.scatter && !c->attr_on_device && !pinned_host;
    if (staged && c->h_out_cap < out_bytes) {
        if (c->h_out) (void)hipHostFree(c->h_out);
        c->h_out = nullptr; c->h_out_cap = 0;
        if (hipHostMalloc((void **)&c->h_out, out_bytes ? out_bytes : 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return RTW_E_NOMEM; }
        c->h_out_cap = out_bytes;
    }
    if (direct) a.out = out.base;
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
        uint32_t levels = c->bvh.depth + 3; if (levels < 4) levels = 4;
        // Layout: f16 nodes FIRST (LDS offset 0: the hottest address of the kernel, the node fetch of every visit, then needs no base
        // register -- as the second block its offset was an SGPR the allocator spilled, one v_readlane per visit), then the per-lane
        // stack, then the optional sphere geometry.
        uint32_t off = 0;
        if (ldsn) { off = c->bvh.n_nodes * 32u; off = (off + 15u) & ~15u; }
        a.lds_stack_off = off;
        off += levels * RTW_BLOCK * (ldsn ? 2u : 4u);
        off = (off + 15u) & ~15u;
        if (ldsn) {
            // sphere geometry rides along only while the workgroup stays under its share (1/6 with 256 threads) of the CU's 160 KiB, i.e. while
            // it does not cost a resident workgroup at the kernel's register budget (6 waves/SIMD)
            bool geom = off + c->sc.n * 16u <= 160u * 1024u / (1536u / RTW_BLOCK);
            if (c->opt_lds_geom >= 0) geom = c->opt_lds_geom != 0 && c->sc.n <= RTW_LDS_GEOM_MAX;
            if (geom && kernel_has_lds_geom(a)) { a.lds_geom_off = off; off += c->sc.n * 16u; }
        }
        a.lds_bytes = off;
    }

    // persistent grid: as many workgroups as the kernel's registers let be resident, capped by the work
    uint32_t per_cu = c->opt_blocks_per_cu;
    if (per_cu == 0) {
        const bool ldsn = c->bvh.nodes16 != nullptr && !(p->flags & RTW_FLAG_GLOBAL_NODES);
        const auto key = std::make_pair(kernel_id(a, c->sc.moving != 0, accel, ldsn), a.lds_bytes);
        auto it = c->occupancy.find(key);
        if (it == c->occupancy.end()) it = c->occupancy.emplace(key, kernel_blocks_per_cu(a, c->sc.moving != 0, accel, ldsn)).first;
        per_cu = it->second;
    }

    if (!(phases & PH_LAUNCH)) return RTW_OK;          // PREPARE only: everything that can wait for the device is behind us

    if (!c->pend.marked) HIP_TRY(hipEventRecord(c->ev_mark, c->stream));
    HIP_TRY(hipMemsetAsync(c->d_stats, 0, RTW_N_STATS * sizeof(unsigned long long), c->stream));
#ifdef RTW_ENDTIMES
    HIP_TRY(hipMemsetAsync(c->d_stats + 24, 0xFF, sizeof(unsigned long long), c->stream));     // atomicMin targets (diagnostic build only)
    HIP_TRY(hipMemsetAsync(c->d_stats + 26, 0xFF, sizeof(unsigned long long), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_stats + 28, 0xFF, sizeof(unsigned long long), c->stream));
#endif
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    for (uint32_t tr0 = 0; tr0 < tile_rows || tr0 == 0; tr0 += (uint32_t)band_tile_rows) {
        const uint32_t tr1 = tr0 + (uint32_t)band_tile_rows < tile_rows ? tr0 + (uint32_t)band_tile_rows : tile_rows;
        a.k_base = tr0 * 8; a.k_end = tr1 * 8 < n_rows ? tr1 * 8 : n_rows;
        a.n_tiles = a.tiles_x * (tr1 - tr0);
        a.total_work = a.n_tiles * a.n_chunks * 64u;              // < 2^32 by the item bound above  (one region; refined below)
        a.reg_q1 = a.reg_q2 = a.n_tiles;
        for (int r = 0; r < 3; r++) { a.reg_len[r] = a.chunk_len; a.reg_nc[r] = a.n_chunks; }
        a.tile_order = nullptr;
        if (c->opt_tile_order && a.n_tiles > 2u && !((c->opt_tile_order == 2u || c->opt_tile_order == 5u) && c->cull.n_other)) {    // (quads / instances: no cost guess, raster order)
            // queue order of the tiles (rtw_host.cpp build_tile_order): built once per (mode, frame shape, camera, partition, scene), kept on the device
            const uint32_t tiles_y = a.n_tiles / a.tiles_x;
            TileOrderKey key;
            std::memset(&key, 0, sizeof key);
            key.mode = c->opt_tile_order; key.tiles_x = a.tiles_x; key.tiles_y = tiles_y; key.k_base = a.k_base;
            key.row_block = a.row_block; key.part_index = a.part_index; key.part_count = a.part_count; key.cam = *cam; key.scene_serial = c->scene_serial;
            key.cam.time0 = key.cam.shutter = key.cam.lens_radius = 0.0f;     // (the estimate uses the centre rays only: the frames of a clip share one order)
            // tiles worth of work in flight when the queue runs dry: four units for every lane of the full grid
            const uint64_t per_tile = 64ull * a.n_chunks;
            key.tail_tiles = (uint32_t)std::min<uint64_t>(a.n_tiles, ((uint64_t)c->n_cu * per_cu * RTW_BLOCK * 4ull + per_tile - 1) / per_tile);
            if (!c->d_order || std::memcmp(&key, &c->order_key, sizeof key) != 0) {
                std::vector<uint32_t> order;
                build_tile_order(c->opt_tile_order, a.tiles_x, tiles_y, a.k_base, a.row_block, a.part_index, a.part_count, *cam, c->cull, key.tail_tiles, order);
                if (c->d_order_cap < order.size()) {
                    if (c->d_order) (void)hipFree(c->d_order);
                    c->d_order = nullptr; c->d_order_cap = 0;
                    HIP_TRY(hipMalloc((void **)&c->d_order, order.size() * sizeof(uint32_t)));
                    c->d_order_cap = order.size();
                }
                HIP_TRY(hipMemcpyAsync(c->d_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
                HIP_TRY(hipStreamSynchronize(c->stream));          // (`order` is a local: the copy must be done before it goes away; once per frame shape)
                c->order_key = key;
            }
            a.tile_order = c->d_order;
        }
        // Small launches are latency-bound, not throughput-bound: a lane should own >= ~16 work units before another resident
        // workgroup per CU pays (profiles/r02_small_frame.log: C1 0.494 -> 0.276 ms with 1 workgroup per CU instead of 6, `First
        // frame` 3.15 -> 2.68 ms with 4; the 1080p frames are indifferent).  An explicit RTW_OPT_BLOCKS_PER_CU wins.
        uint32_t wg_per_cu = per_cu;
        if (c->opt_blocks_per_cu == 0) {
            const uint64_t lanes_per_wg_row = (uint64_t)c->n_cu * RTW_BLOCK * 16ull;
            const uint64_t want = ((uint64_t)a.total_work + lanes_per_wg_row - 1) / lanes_per_wg_row;
            if (want < wg_per_cu) wg_per_cu = (uint32_t)(want ? want : 1);
            // The specialised BVH builds are compiled for 7 waves/SIMD (72 VGPRs); the seventh workgroup per CU pays on the full 1080p frames
            // (bench frame 77.9 -> 76.4 ms, C4 319 -> 311, C5 96.8 -> 95.0) and costs on shorter launches (an eighth of the bench frame 10.9 ->
            // 11.1 ms, C2 7.18 -> 7.35: one more wave per SIMD to drain at the end): it is used from ~40 work units per lane
            // (profiles/r02_ab_waves7.log; profiles/r02_wg_per_cu.log: 6 / 7 workgroups per CU = 78.2 / 76.1 ms for the bench frame, 39.8 / 38.8 for
            // half of it, 20.75 / 20.40 for a quarter, 10.95 / 10.95 for an eighth, 7.07 / 7.11 for C2).
            if (wg_per_cu > 6u && (uint64_t)a.total_work < 40ull * c->n_cu * RTW_BLOCK * wg_per_cu) wg_per_cu = 6u;
        }
        uint32_t grid = (uint32_t)c->n_cu * wg_per_cu;
        {
            const uint32_t need = (a.total_work + RTW_BLOCK - 1) / RTW_BLOCK;
            if (grid > need) grid = need ? need : 1;
        }
        // Guided unit length (RTW_OPT_TAIL_UNITS = k; 0, the default, = one length for the whole launch).  The idea: the end of a
        // launch is one work unit deep per lane, so cut the LAST tiles of the queue into units of one sample and the tiles before them into units of
        // a third of the launch's length -- region 3 holds k blocks of (tile, 1 sample) for every resident wave, region 2 k blocks of (tile, L2
        // samples).  The image cannot change: the bank is indexed by (tile, sample, pixel) and the resolve adds in sample order.
        // MEASURED (profiles/r03_tail_units.log, r03_endtimes.log, r03_tail_order.log): it does what it says -- a wave runs on for 0.31 instead of
        // 0.40 ms on average after it finds the queue empty -- and gains nothing: an eighth of the bench frame 10.21 / 10.24 / 10.34 / 10.36 ms for
        // k = 0 / 1 / 2 / 4, C2 7.07 / 7.09 / 7.00 / 7.13, the whole frame 74.48 / 74.64 / 74.71 / 74.66.  What a launch waits for at its end is not
        // a unit but a PATH: with one-sample units the longest wait is still ~1.1 ms -- a 50-bounce glass path started just before the queue ran
        // dry, at ~20 us per segment with seven busy waves per SIMD -- and a path cannot be split.  Handing the sky tiles out last so that no such
        // path starts late (RTW_OPT_TILE_ORDER 5) does not help either.  Kept as an option; off.
        if (!sums && c->opt_tail_units > 0.0 && a.chunk_len > 1u && a.n_tiles > 1u) {
            const double waves = (double)grid * (RTW_BLOCK / 64u);
            const uint32_t L1 = a.chunk_len, L2 = L1 >= 12u ? 4u : (L1 >= 6u ? 2u : 1u), L3 = 1u;
            uint64_t t3 = (uint64_t)std::ceil(c->opt_tail_units * waves * L3 / (double)a.n_samples);
            uint64_t t2 = L2 > L3 ? (uint64_t)std::ceil(c->opt_tail_units * waves * L2 / (double)a.n_samples) : 0;
            if (t3 > a.n_tiles / 4u) t3 = a.n_tiles / 4u;                     // (small launches: at most a quarter of the tiles in either region)
            if (t2 > a.n_tiles / 4u) t2 = a.n_tiles / 4u;
            a.reg_q2 = a.n_tiles - (uint32_t)t3; a.reg_q1 = a.reg_q2 - (uint32_t)t2;
            a.reg_len[1] = L2; a.reg_len[2] = L3;
            for (int r = 1; r < 3; r++) a.reg_nc[r] = (a.n_samples + a.reg_len[r] - 1u) / a.reg_len[r];
            a.total_work = (a.reg_q1 * a.reg_nc[0] + (a.reg_q2 - a.reg_q1) * a.reg_nc[1] + (a.n_tiles - a.reg_q2) * a.reg_nc[2]) * 64u;   // <= tiles x samples x 64 < 2^32
        }
        {   // guided grabs (fetch_pixel): work left / (4 x resident waves), as a shift; at most RTW_OPT_GRAB_BLOCKS blocks (0: the blocks of one tile)
            a.sub_shift = (c->opt_sub_queues != 1u && a.n_tiles >= 64u && grid >= 64u) ? RTW_SUB_SHIFT : 0u;       // one sub-queue per XCD, unless the launch is tiny
            const uint32_t waves4 = (grid * (RTW_BLOCK / 64u) * 4u) >> a.sub_shift;
            a.grab_shift = 0; while (a.grab_shift < 31u && (1u << a.grab_shift) < waves4) a.grab_shift++;
            uint32_t blocks = c->opt_grab_blocks ? c->opt_grab_blocks : a.n_chunks;
            if (blocks > a.n_chunks) blocks = a.n_chunks;
            a.grab_max = (blocks ? blocks : 1u) * 64u;
        }
        HIP_TRY(hipMemsetAsync(c->d_queue, 0, RTW_QUEUE_BYTES, c->stream));
        if (a.n_tiles) launch_render(a, c->sc.moving != 0, accel, grid, c->stream);
        HIP_TRY(hipGetLastError());
        if (tile_rows == 0) break;
    }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, RTW_N_STATS * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    c->pend.active = true; c->pend.n_rows = n_rows;
    c->pend.dst = out.base; c->pend.scatter = out.scatter; c->pend.direct = direct; c->pend.staged = staged;
    c->pend.width = p->width; c->pend.height = p->height; c->pend.row_block = a.row_block; c->pend.part_index = p->part_index; c->pend.part_count = p->part_count;
    c->pend.out_bytes = out_bytes;
    c->pend.enqueue_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c->pend.t0).count();
    if (phases & PH_COPY) return render_copy(c);
    return RTW_OK;
}

// COPY phase: the rendered rows toward the caller's memory, queued behind the kernels on the context's stream.
static int render_copy(rtw_ctx *c) {
    if (!c || !c->pend.active) return RTW_E_INVALID;
    const rtw_ctx::Pending &q = c->pend;
    if (q.direct) return RTW_OK;                        // the kernels wrote into the caller's device buffer
    HIP_TRY(hipSetDevice(c->device));
    if (q.staged) {                                     // pinned staging now (a true asynchronous copy), the caller's pageable frame in render_wait
        HIP_TRY(hipMemcpyAsync(c->h_out, c->d_out, q.out_bytes, hipMemcpyDeviceToHost, c->stream));
    } else if (q.scatter) {
        const int rc = scatter_rows(c, c->d_out, q.dst, q.width, q.height, q.row_block, q.part_index, q.part_count);
        if (rc != RTW_OK) return rc;
    } else HIP_TRY(hipMemcpyAsync(q.dst, c->d_out, q.out_bytes, hipMemcpyDefault, c->stream));
    return RTW_OK;
}

// Wait for the render launched by render_enqueue and report its counters.
static int render_wait(rtw_ctx *c, RtwStats *stats) {
    if (!c || !c->pend.active) return RTW_E_INVALID;
    c->pend.active = false; c->pend.marked = false;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->pend.staged)
        scatter_rows_host(c->h_out, c->pend.dst, c->pend.width, c->pend.height, c->pend.n_rows, c->pend.row_block, c->pend.part_index, c->pend.part_count);
    float ms = 0.0f, start_ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    HIP_TRY(hipEventElapsedTime(&start_ms, c->ev_mark, c->ev0));
    const unsigned long long *h_stats = c->h_stats;
    const bool kernel_gave_up = h_stats[23] != 0ull;           // safety valve of the persistent loop (rtw_kernels.hip RTW_MAX_TRIPS)
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->camera_rays = h_stats[0]; stats->segments = h_stats[1];
        stats->sphere_tests = h_stats[2]; stats->node_tests = h_stats[3];
        stats->nan_pixels = (uint32_t)h_stats[4]; stats->rows = c->pend.n_rows;
        stats->quad_tests = h_stats[14];
        stats->kernel_ms = ms;
        stats->enqueue_ms = c->pend.enqueue_ms; stats->start_ms = start_ms;
        for (int k = 0; k < 3; k++) { stats->phase_steps[k] = h_stats[5 + k]; stats->phase_lanes[k] = h_stats[8 + k]; }
        for (int k = 3; k < 5; k++) { stats->phase_steps[k] = h_stats[16 + 2 * (k - 3)]; stats->phase_lanes[k] = h_stats[17 + 2 * (k - 3)]; }
#ifdef RTW_CENSUS                                   // diagnostic build only (scripts/gpu_census.py): lanes live in the sub-blocks of a SHADE step
        if (getenv("RTW_CENSUS_DUMP")) {
            static const char *names[CEN_N] = { "shading", "inflight (unit(d))", "miss (sky)", "hit (point, normal, material)", "dielectric branch", "schlick draw",
                                                "diffuse branch", "unit-vector loop trip", "bank store", "needs a work unit", "start_path", "lens-disk loop trip",
                                                "trav_begin", "depth exhausted", "trips through the scheduler (all phases)", "cooperative round" };
            std::fprintf(stderr, "rtw census: %llu SHADE steps\n", h_stats[7]);
            for (int k = 0; k < CEN_N; k++)
                if (h_stats[32 + 2 * k])
                    std::fprintf(stderr, "rtw census: %-32s executions %12llu (%.3f per SHADE step)  lanes %14llu  live fraction %.4f  lanes per SHADE step %.2f\n", names[k],
                                 h_stats[32 + 2 * k], (double)h_stats[32 + 2 * k] / (double)h_stats[7], h_stats[33 + 2 * k],
                                 (double)h_stats[33 + 2 * k] / (64.0 * (double)h_stats[32 + 2 * k]), (double)h_stats[33 + 2 * k] / (double)h_stats[7]);
        }
#endif
#if defined(RTW_STAMP) || defined(RTW_ENDTIMES)     // diagnostic builds only (scripts/gpu_endtimes.py)
        if (getenv("RTW_STAMP_DUMP")) std::fprintf(stderr, "rtw stamp: wave-ticks traverse %llu leaf %llu shade %llu   of shade (specialised builds): a. unit(d) + paths that end %llu, b. bank + next unit %llu, c-d. hit / path start + the rejection loop + second halves %llu, e. query begin %llu   (generic build: hit, bank + next unit, camera ray, query begin)\n",
                                                   h_stats[11], h_stats[12], h_stats[13], h_stats[16], h_stats[17], h_stats[18], h_stats[19]);
        if (getenv("RTW_ENDTIMES_DUMP") && h_stats[15] && getenv("RTW_ENDTIMES_REF")) {
            std::fprintf(stderr, "rtw endtimes histogram (waves ending in each 1/32 of the reference lifetime, bins 21/32 .. 32/32+; bin 0 also holds everything earlier):");
            for (int k = 0; k < 12; k++) std::fprintf(stderr, " %llu", h_stats[20 + k]);
            std::fprintf(stderr, "\n");
        }
        if (getenv("RTW_ENDTIMES_DUMP") && h_stats[15])
            std::fprintf(stderr, "rtw endtimes (100 MHz device clock): wave starts spread over %.1f us, first wave end %.1f us and last wave end %.1f us after the first start\n",
                         (double)(h_stats[25] - h_stats[24]) / 100.0, (double)(h_stats[26] - h_stats[24]) / 100.0, (double)(h_stats[27] - h_stats[24]) / 100.0);
        if (getenv("RTW_ENDTIMES_DUMP") && h_stats[15])
            std::fprintf(stderr, "rtw endtimes: the queue ran dry for the first wave %.1f us and for the last wave %.1f us after the first start; a wave then ran on for %.1f us on average, %.1f us at most\n",
                         (double)(h_stats[28] - h_stats[24]) / 100.0, (double)(h_stats[29] - h_stats[24]) / 100.0, (double)h_stats[31] / (double)h_stats[15] / 100.0, (double)h_stats[30] / 100.0);
        if (getenv("RTW_ENDTIMES_DUMP") && h_stats[15]) {   // RTW_ENDTIMES build: [12] longest wave lifetime [13] sum of wave lifetimes [15] waves
            if (h_stats[40]) {
                const double n = (double)h_stats[40];
                std::fprintf(stderr, "rtw endtimes: %llu waves ran on for 0.6 ms or more after finding the queue empty: on average %.0f us, %.0f scheduler trips (%.0f TRAVERSE steps with %.1f lanes, %.0f LEAF with %.1f, %.0f SHADE with %.1f), the busiest lane shaded %.1f queries, all lanes %.1f -> %.1f us per trip, %.1f us per query of the busiest lane\n",
                             h_stats[40], h_stats[41] / n / 100.0, h_stats[42] / n, h_stats[43] / n, h_stats[43] ? (double)h_stats[46] / h_stats[43] : 0.0, h_stats[44] / n, h_stats[44] ? (double)h_stats[47] / h_stats[44] : 0.0,
                             h_stats[45] / n, h_stats[45] ? (double)h_stats[48] / h_stats[45] : 0.0, h_stats[49] / n, h_stats[50] / n,
                             h_stats[42] ? h_stats[41] / 100.0 / h_stats[42] : 0.0, h_stats[49] ? h_stats[41] / 100.0 / h_stats[49] : 0.0);
            }
            std::fprintf(stderr, "rtw endtimes: %llu waves, longest lifetime %llu ticks, mean lifetime %.1f %% of it\n", h_stats[15], h_stats[12],
                         100.0 * (double)h_stats[13] / (double)h_stats[15] / (double)h_stats[12]);
        }
#endif
        stats->total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - c->pend.t0).count();
    }
    return kernel_gave_up ? RTW_E_INTERNAL : RTW_OK;
}

struct rtw_mgpu {
    std::vector<rtw_ctx *> ctx;
};

extern "C" {

int rtw_ctx_set_option(rtw_ctx *c, uint32_t key, double v) {
    if (!c || !(v == v)) return RTW_E_INVALID;
    switch (key) {
    case RTW_OPT_CHUNK_LEN:      if (!(v >= 0.0 && v <= 255.0)) return RTW_E_INVALID; c->opt_chunk_len = (uint32_t)v; return RTW_OK;
    case RTW_OPT_SAMPLE_BANK_GB: if (!(v > 0.0 && v <= 1048576.0)) return RTW_E_INVALID; c->opt_bank_bytes = (uint64_t)(v * (double)(1ull << 30)); return RTW_OK;
    case RTW_OPT_LDS_GEOM:       if (!(v >= -1.0 && v <= 1.0)) return RTW_E_INVALID; c->opt_lds_geom = (int)v; return RTW_OK;
    case RTW_OPT_BLOCKS_PER_CU:  if (!(v >= 0.0 && v <= 8.0)) return RTW_E_INVALID; c->opt_blocks_per_cu = (uint32_t)v; return RTW_OK;
    case RTW_OPT_LIST_WALK_MAX:  if (!(v >= 0.0 && v <= 4294967295.0)) return RTW_E_INVALID; c->opt_list_walk_max = (uint32_t)v; return RTW_OK;
    case RTW_OPT_TILE_ORDER:     if (!(v >= 0.0 && v <= 5.0 && v == (double)(uint32_t)v)) return RTW_E_INVALID; c->opt_tile_order = (uint32_t)v; return RTW_OK;
    case RTW_OPT_SUB_QUEUES:     if (!(v == 0.0 || v == 1.0)) return RTW_E_INVALID; c->opt_sub_queues = (uint32_t)v; return RTW_OK;
    case RTW_OPT_TAIL_UNITS:     if (!(v >= 0.0 && v <= 1024.0)) return RTW_E_INVALID; c->opt_tail_units = v; return RTW_OK;
    case RTW_OPT_GRAB_BLOCKS:    if (!(v >= 0.0 && v <= 65536.0 && v == (double)(uint32_t)v)) return RTW_E_INVALID; c->opt_grab_blocks = (uint32_t)v; return RTW_OK;
    default: return RTW_E_INVALID;
    }
}

int rtw_ctx_render(rtw_ctx *c, const RtwCamera *cam, const RtwParams *p, float *out_rgb, RtwStats *stats) {
    int rc = render_enqueue(c, cam, p, OutSpec{ out_rgb, false });
    if (rc != RTW_OK) return rc;
    return render_wait(c, stats);
}

// ---- one frame over several GPUs (rtw.h) -------------------------------------------------------------
int rtw_mgpu_create(const int *devices, uint32_t n, rtw_mgpu **out) {
    if (!out) return RTW_E_INVALID;
    *out = nullptr;
    if (!devices || n == 0 || n > 64) return RTW_E_INVALID;
    if (rtw_device_count() <= 0) return RTW_E_NO_DEVICE;
    rtw_mgpu *m = new (std::nothrow) rtw_mgpu();
    if (!m) return RTW_E_NOMEM;
    for (uint32_t k = 0; k < n; k++) {
        rtw_ctx *c = nullptr;
        int rc = rtw_ctx_create(devices[k], &c);
        if (rc != RTW_OK) { rtw_mgpu_destroy(m); return rc; }
        m->ctx.push_back(c);
    }
    // let every device write into the others' memory (frames that live in one GPU's HBM); failure is fine: the copies then go
    // through the host, which hipMemcpyDefault does by itself
    for (uint32_t k = 0; k < n; k++) for (uint32_t j = 0; j < n; j++) {
        if (devices[k] == devices[j]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[k], devices[j]) == hipSuccess && can) {
            (void)hipSetDevice(devices[k]);
            if (hipDeviceEnablePeerAccess(devices[j], 0) != hipSuccess) (void)hipGetLastError();   // (already enabled is an error code too)
        } else (void)hipGetLastError();
    }
    *out = m;
    return RTW_OK;
}

void rtw_mgpu_destroy(rtw_mgpu *m) {
    if (!m) return;
    for (rtw_ctx *c : m->ctx) rtw_ctx_destroy(c);
    delete m;
}

int rtw_mgpu_set_scene(rtw_mgpu *m, const RtwScene *scene, float t_begin, float t_end) {
    if (!m) return RTW_E_INVALID;
    for (rtw_ctx *c : m->ctx) { int rc = rtw_ctx_set_scene(c, scene, t_begin, t_end); if (rc != RTW_OK) return rc; }
    return RTW_OK;
}

int rtw_mgpu_set_option(rtw_mgpu *m, uint32_t key, double value) {
    if (!m) return RTW_E_INVALID;
    for (rtw_ctx *c : m->ctx) { int rc = rtw_ctx_set_option(c, key, value); if (rc != RTW_OK) return rc; }
    return RTW_OK;
}

int rtw_mgpu_render(rtw_mgpu *m, const RtwCamera *cam, const RtwParams *params, float *out_rgb, RtwStats *per_device, RtwStats *total) {
    if (!m || !cam || !params || !out_rgb || params->part_count > 1) return RTW_E_INVALID;
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n = (uint32_t)m->ctx.size();
    // fork: viewport.rs:236-240 spawns one task per row; here one asynchronous launch sequence per device, in three passes so that nothing
    // one device needs -- a first-use allocation, a driver query, a copy toward the caller's frame -- can delay the start of another:
    //   0. every device: mark the begin of the call on its stream, check the arguments, allocate what this frame shape needs (may wait)
    //   1. every device: memsets, kernels, events, counter read-back (never waits)
    //   2. every device: its rows toward the frame -- strided copies into device or pinned memory, pinned staging for a pageable frame
    int rc = RTW_OK;
    auto part = [&](uint32_t k) { RtwParams q = *params; q.row_block = params->row_block ? params->row_block : 8u; q.part_index = k; q.part_count = n; return q; };
    for (uint32_t k = 0; k < n && rc == RTW_OK; k++) {
        rtw_ctx *c = m->ctx[k];
        if (c->pend.active) { rc = RTW_E_INVALID; break; }
        c->pend.t0 = t0;
        if (hipSetDevice(c->device) != hipSuccess || hipEventRecord(c->ev_mark, c->stream) != hipSuccess) { (void)hipGetLastError(); rc = RTW_E_HIP; break; }
        c->pend.marked = true;
        const RtwParams q = part(k);
        rc = render_enqueue(c, cam, &q, OutSpec{ out_rgb, true }, PH_PREPARE);
    }
    uint32_t launched = 0;
    for (; launched < n && rc == RTW_OK; launched++) {
        const RtwParams q = part(launched);
        rc = render_enqueue(m->ctx[launched], cam, &q, OutSpec{ out_rgb, true }, PH_PREPARE | PH_LAUNCH);
        if (rc != RTW_OK) break;
    }
    for (uint32_t k = 0; k < launched && rc == RTW_OK; k++) rc = render_copy(m->ctx[k]);
    for (uint32_t k = 0; k < n; k++) m->ctx[k]->pend.marked = m->ctx[k]->pend.active && m->ctx[k]->pend.marked;
    // join: viewport.rs:241-244 awaits the tasks in order
    RtwStats sum; std::memset(&sum, 0, sizeof sum);
    for (uint32_t k = 0; k < launched; k++) {
        RtwStats st;
        const int rw = render_wait(m->ctx[k], &st);
        if (rw != RTW_OK) { if (rc == RTW_OK) rc = rw; continue; }
        if (per_device) per_device[k] = st;
        sum.camera_rays += st.camera_rays; sum.segments += st.segments; sum.sphere_tests += st.sphere_tests;
        sum.node_tests += st.node_tests; sum.quad_tests += st.quad_tests; sum.nan_pixels += st.nan_pixels; sum.rows += st.rows;
        for (int i = 0; i < 6; i++) { sum.phase_steps[i] += st.phase_steps[i]; sum.phase_lanes[i] += st.phase_lanes[i]; }
        if (st.kernel_ms > sum.kernel_ms) sum.kernel_ms = st.kernel_ms;
        if (st.enqueue_ms > sum.enqueue_ms) sum.enqueue_ms = st.enqueue_ms;          // (the last device's: the fork's length on the host)
        if (st.start_ms > sum.start_ms) sum.start_ms = st.start_ms;                // (the latest kernel start)
    }
    sum.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (total) *total = sum;
    return rc;
}

int rtw_render_multi_gpu(const int *devices, uint32_t n_devices, const RtwCamera *cam, const RtwScene *scene,
                         const RtwParams *params, float *out_rgb, RtwStats *per_device) {
    if (!cam || !scene || !params) return RTW_E_INVALID;
    rtw_mgpu *m = nullptr;
    int rc = rtw_mgpu_create(devices, n_devices, &m);
    if (rc != RTW_OK) return rc;
    rc = rtw_mgpu_set_scene(m, scene, cam->time0, cam->time0 + cam->shutter);
    if (rc == RTW_OK) rc = rtw_mgpu_render(m, cam, params, out_rgb, per_device, nullptr);
    rtw_mgpu_destroy(m);
    return rc;
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
