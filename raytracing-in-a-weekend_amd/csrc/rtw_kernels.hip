// rtw_kernels.hip -- the gfx950 render kernels.
//
// One kernel does the reference's L3+L4+L5: the pixel/sample driver
// (Rust/src/viewport.rs:270-305 render_row, :430-478 render, :479-516 render_no_rand), the
// integrator (Rust/src/viewport/ray_color.rs:12-92, written front-to-back instead of recursively)
// and the closest-hit + scatter (objects/sphere.rs:99-147, objects/materials.rs:105-154).
//
// Execution shape (wave64, CDNA4):
//   * persistent workgroups (4 waves); a lane owns ONE pixel at a time and walks its samples in
//     order, so the per-pixel f32 sum has the reference's order (viewport.rs:299 `color +=`);
//   * a lane whose path ends starts its pixel's next sample in the same loop trip ("regeneration"),
//     and a lane whose pixel is finished pulls the next pixel index from a global queue with ONE
//     wave-aggregated atomic (ballot + mbcnt), so every loop trip runs a closest-hit query with
//     (almost) all 64 lanes live irrespective of how path lengths differ across the tile;
//   * work items are ordered 8x8-tile-major so the 64 pixels a wave pulls together are one tile;
//   * the brute-force sphere loop reads {centre, r^2} with wave-uniform SCALAR loads: the VALU ops
//     per sphere take their sphere operands straight from SGPRs, no LDS or vector-memory traffic.
#include "rtw_kernels.h"

namespace rtw {

// ---- closest hit, brute force in list order (camera_tests.rs:19-33; `min_hit == None || min_hit > i`)
template <bool MOVING>
__device__ __forceinline__ void closest_brute(const DevScene &sc, v3 o, v3 d, float tm, float mint, float maxt,
                                              int &best, float &best_t) {
    cf4_ptr geom = (cf4_ptr)(uintptr_t)sc.geom;
    cf4_ptr vel = (cf4_ptr)(uintptr_t)sc.vel;
    const float a = dot(d, d);
    best = -1; best_t = 0.0f;
    const uint32_t n = sc.n;
    for (uint32_t s = 0; s < n; ++s) {
        f4 g = geom[s];
        float cx = g.x, cy = g.y, cz = g.z;
        if (MOVING) {                      // sphere.rs:100  origin + velocity * r.time
            f4 vv = vel[s];
            cx = cx + vv.x * tm; cy = cy + vv.y * tm; cz = cz + vv.z * tm;
        }
        float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
        float b = ocx * d.x + ocy * d.y + ocz * d.z;
        float c = (ocx * ocx + ocy * ocy + ocz * ocz) - g.w;
        float disc = b * b - a * c;
        if (!(disc < 0.0f)) {
            float sq = __builtin_sqrtf(disc);
            float x = (-b - sq) / a;
            if (x < mint) x = (-b + sq) / a;
            if (!(x < mint || x > maxt)) {
                if (best < 0 || best_t > x) { best = (int)s; best_t = x; }
            }
        }
    }
}

// ---- exact sphere test shared by the BVH paths (same arithmetic as closest_brute) ----------------
// Candidate order is not list order here, so ties are resolved explicitly toward the lower index --
// the sphere `min_hit > i` keeps in list order.
template <bool MOVING>
__device__ __forceinline__ void exact_sphere(f4 g, f4 vv, uint32_t s, v3 o, v3 d, float tm, float a, float mint, float maxt,
                                             int &best, float &best_t) {
    float cx = g.x, cy = g.y, cz = g.z;
    if (MOVING) { cx = cx + vv.x * tm; cy = cy + vv.y * tm; cz = cz + vv.z * tm; }
    float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
    float b = ocx * d.x + ocy * d.y + ocz * d.z;
    float c = (ocx * ocx + ocy * ocy + ocz * ocz) - g.w;
    float disc = b * b - a * c;
    if (!(disc < 0.0f)) {
        float sq = __builtin_sqrtf(disc);
        float x = (-b - sq) / a;
        if (x < mint) x = (-b + sq) / a;
        if (!(x < mint || x > maxt)) {
            if (x < best_t || (x == best_t && s < (uint32_t)best)) { best = (int)s; best_t = x; }
        }
    }
}

// ---- closest hit through the BVH ------------------------------------------------------------------
// Result-identical to closest_brute (tests/test_gpu_parity.py checks it bit for bit).  The tree only
// PRUNES; every surviving candidate runs the exact reference arithmetic above.  Pruning is made safe
// against the reference's own f32 rounding (DESIGN.md "Conservative traversal"):
//   the reference reports a hit when fl(b*b - a*c) >= 0, which implies the ray passes within
//   sqrt(r^2 + K u (|oc|^2 + r^2)) of the centre (K = 24 >= the 15 the error analysis needs, u = 2^-24),
//   and its t can be earlier than the geometric entry by at most sqrt(K u (|oc|^2 + r^2)) / |d|.
// So each ray is thickened by rho (slab tests against boxes inflated by rho: folded into the per-ray
// constants, zero cost per box) and the t-interval is widened by tau.  Spheres much larger than the
// rest ("big", e.g. the ground) stay outside the tree and are tested exactly first, which both keeps
// rho/tau small and gives an early best_t.
#define RTW_KU 1.4305115e-6f    /* 24 * 2^-24 */

template <bool MOVING>
__device__ __forceinline__ void closest_bvh(const DevScene &sc, const DevBvh &bv, int *stack, v3 o, v3 d, float tm,
                                            float mint, float maxt, int &best, float &best_t, uint32_t &n_nodes, uint32_t &n_tests) {
    const float a = dot(d, d);
    best = -1; best_t = maxt;
    // 1. big spheres: uniform loop, scalar loads
    {
        cf4_ptr bg = (cf4_ptr)(uintptr_t)bv.big_geom;
        cf4_ptr bvel = (cf4_ptr)(uintptr_t)bv.big_vel;
        for (uint32_t k = 0; k < bv.n_big; ++k) {
            f4 vv = MOVING ? bvel[k] : f4{ 0, 0, 0, 0 };
            exact_sphere<MOVING>(bg[k], vv, bv.big_index[k], o, d, tm, a, mint, maxt, best, best_t);
        }
        n_tests += bv.n_big;
    }
    int node = bv.root;
    if (node == (int)0x80000000) return;
    if (node < 0) {     // single-sphere tree
        const uint32_t s = (uint32_t)~node;
        exact_sphere<MOVING>(sc.geom[s], MOVING ? sc.vel[s] : f4{ 0, 0, 0, 0 }, s, o, d, tm, a, mint, maxt, best, best_t);
        n_tests++;
        return;
    }
    // 2. per-ray constants of the thick-ray slab test
    const float ex = o.x - bv.cx, ey = o.y - bv.cy, ez = o.z - bv.cz;
    const float M = __builtin_sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + bv.centre_radius;
    const float q = (M * M + bv.r_max2) * RTW_KU;
    const float sq_q = __builtin_sqrtf(q) * 1.0001f;
    float rho = fminf(q * bv.inv_2rmin, sq_q);
    rho = rho * 1.0001f + 4.8e-7f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + bv.abs_max);   // + slab-arithmetic slop (8u * magnitudes)
    const float tau_t = sq_q * 1.0001f / __builtin_sqrtf(a) + 1e-30f;
    float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    if (!(fabsf(d.x) >= 1e-20f)) ix = copysignf(1e20f, d.x);
    if (!(fabsf(d.y) >= 1e-20f)) iy = copysignf(1e20f, d.y);
    if (!(fabsf(d.z) >= 1e-20f)) iz = copysignf(1e20f, d.z);
    // t(lo) = (lo - rho - o) * inv = fma(lo, inv, -(o + rho) * inv);  t(hi) = fma(hi, inv, -(o - rho) * inv)
    const float kpx = -(o.x + rho) * ix, kpy = -(o.y + rho) * iy, kpz = -(o.z + rho) * iz;
    const float kmx = -(o.x - rho) * ix, kmy = -(o.y - rho) * iy, kmz = -(o.z - rho) * iz;
    const float lo_lim = mint - tau_t;
    float hi_lim = best_t + tau_t;

    const uint32_t tid = threadIdx.x;
    uint32_t sp = 0;
    for (;;) {
        const f4 *np = (const f4 *)(bv.nodes + node);
        const f4 n0 = np[0], n1 = np[1], n2 = np[2];
        const int c0 = bv.nodes[node].c0, c1 = bv.nodes[node].c1;
        n_nodes++;
        // child 0 box: lo0 = n0.xyz, hi0 = (n0.w, n1.x, n1.y); child 1: lo1 = (n1.z, n1.w, n2.x), hi1 = n2.yzw
        float t1, t2;
        t1 = __builtin_fmaf(n0.x, ix, kpx); t2 = __builtin_fmaf(n0.w, ix, kmx);
        float e0 = fminf(t1, t2), x0 = fmaxf(t1, t2);
        t1 = __builtin_fmaf(n0.y, iy, kpy); t2 = __builtin_fmaf(n1.x, iy, kmy);
        e0 = fmaxf(e0, fminf(t1, t2)); x0 = fminf(x0, fmaxf(t1, t2));
        t1 = __builtin_fmaf(n0.z, iz, kpz); t2 = __builtin_fmaf(n1.y, iz, kmz);
        e0 = fmaxf(e0, fminf(t1, t2)); x0 = fminf(x0, fmaxf(t1, t2));
        t1 = __builtin_fmaf(n1.z, ix, kpx); t2 = __builtin_fmaf(n2.y, ix, kmx);
        float e1 = fminf(t1, t2), x1 = fmaxf(t1, t2);
        t1 = __builtin_fmaf(n1.w, iy, kpy); t2 = __builtin_fmaf(n2.z, iy, kmy);
        e1 = fmaxf(e1, fminf(t1, t2)); x1 = fminf(x1, fmaxf(t1, t2));
        t1 = __builtin_fmaf(n2.x, iz, kpz); t2 = __builtin_fmaf(n2.w, iz, kmz);
        e1 = fmaxf(e1, fminf(t1, t2)); x1 = fminf(x1, fmaxf(t1, t2));
        bool h0 = e0 <= x0 && x0 >= lo_lim && e0 <= hi_lim;
        bool h1 = e1 <= x1 && x1 >= lo_lim && e1 <= hi_lim;
        if (h0 && c0 < 0) {
            const uint32_t s = (uint32_t)~c0;
            exact_sphere<MOVING>(sc.geom[s], MOVING ? sc.vel[s] : f4{ 0, 0, 0, 0 }, s, o, d, tm, a, mint, maxt, best, best_t);
            n_tests++; h0 = false;
        }
        if (h1 && c1 < 0) {
            const uint32_t s = (uint32_t)~c1;
            exact_sphere<MOVING>(sc.geom[s], MOVING ? sc.vel[s] : f4{ 0, 0, 0, 0 }, s, o, d, tm, a, mint, maxt, best, best_t);
            n_tests++; h1 = false;
        }
        hi_lim = best_t + tau_t;
        h0 = h0 && e0 <= hi_lim;
        h1 = h1 && e1 <= hi_lim;
        if (h0 && h1) {
            const bool near0 = e0 <= e1;
            stack[sp * RTW_BLOCK + tid] = near0 ? c1 : c0;
            sp++;
            node = near0 ? c0 : c1;
        } else if (h0) node = c0;
        else if (h1) node = c1;
        else {
            if (sp == 0) break;
            sp--;
            node = stack[sp * RTW_BLOCK + tid];
        }
    }
}

template <bool MOVING, int ACCEL>
__global__ __launch_bounds__(RTW_BLOCK) void render_kernel(const KArgs A) {
    const uint32_t lane = threadIdx.x & 63u;
    const DevScene &sc = A.sc;
    // per-lane traversal stack, [level][thread] so that a level is one conflict-free LDS row
    __shared__ int bvh_stack[ACCEL == RTW_ACCEL_BVH ? RTW_BVH_STACK * RTW_BLOCK : 1];

    // pixel state
    bool dead = false, have = false, newpath = false;
    uint32_t pi = 0, pj = 0, pk = 0, rng_base = 0, s = 0;
    v3 acc = mk(0, 0, 0);
    // path state
    uint32_t k = 0;
    v3 o = mk(0, 0, 0), d = mk(0, 0, 0), thr = mk(1, 1, 1), L = mk(0, 0, 0);
    float tm = 0.0f;
    bool poison = false;
    Rng rng; rng.state = 0; rng.inc = 1;
    // counters
    uint32_t n_seg = 0, n_rays = 0, n_nan = 0, n_nodes = 0, n_tests = 0;

    const v3 cam_o = ld3(A.cam.origin), cam_u = ld3(A.cam.u), cam_v = ld3(A.cam.v);
    const v3 p00 = ld3(A.cam.pixel00), du = ld3(A.cam.delta_u), dv = ld3(A.cam.delta_v);

    for (;;) {
        // ---- 1. lanes without a pixel pull the next work item (one atomic per wave) ------------
        const bool need = !have && !dead;
        const unsigned long long m = __ballot(need);
        if (m) {
            const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(A.queue, (uint32_t)__popcll(m));
            base = (uint32_t)__shfl((int)base, (int)leader);
            if (need) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t w = base + rank;
                if (w >= A.total_work) dead = true;
                else {
                    // 8x8-tile-major order over the compact rows of this partition
                    const uint32_t tile = w >> 6, p = w & 63u;
                    const uint32_t tcol = tile % A.tiles_x, trow = tile / A.tiles_x;
                    pi = tcol * 8u + (p & 7u);
                    pk = trow * 8u + (p >> 3);
                    if (pi < A.width && pk < A.n_rows) {
                        have = true; newpath = true; s = 0; acc = mk(0, 0, 0);
                        pj = pk;   // compact row -> image row (RtwParams row partition)
                        if (A.part_count > 1) pj = ((pk / A.row_block) * A.part_count + A.part_index) * A.row_block + (pk % A.row_block);
                        rng_base = rng_pixel_base(A.seed_lo, A.seed_hi, pj * A.width + pi);
                    }
                }
            }
        }
        if (__ballot(!dead) == 0ull) break;

        // ---- 2. start the next camera ray of this lane's pixel ---------------------------------
        if (have && newpath) {
            newpath = false;
            rng = rng_start(rng_base, s);
            thr = mk(1.0f, 1.0f, 1.0f); L = mk(0, 0, 0); poison = false; k = 0;
            if (A.sampler == RTW_SAMPLER_NO_RAND) {              // viewport.rs:498-503
                o = cam_o;
                d = (p00 + du * (float)pi) + dv * (float)pj;
                tm = 0.0f;
            } else {
                float jx, jy;
                if (A.sampler == RTW_SAMPLER_CENTRES) {          // Rust2/src/viewport.rs:92-104: direction first, then the disk draw
                    const uint32_t kx = s / A.s_root, ly = s % A.s_root;
                    jx = ((float)pi + ((float)kx + 0.5f) / (float)A.s_root) / (float)A.width;
                    jy = ((float)pj + ((float)ly + 0.5f) / (float)A.s_root) / (float)A.height;
                    float rx, ry; random_in_unit_disk(rng, rx, ry);
                    o = cam_o + mk(rx, ry, 0.0f) * A.cam.lens_radius;
                    tm = 0.0f;
                } else {
                    float rx, ry; random_in_unit_disk(rng, rx, ry);   // always drawn (viewport.rs:288)
                    o = cam_o + (cam_u * rx + cam_v * ry) * A.cam.lens_radius;
                    if (A.sampler == RTW_SAMPLER_ROW) {          // viewport.rs:290-297
                        jx = (float)pi + rng_f32(rng);
                        jy = (float)pj + rng_f32(rng);
                        tm = A.cam.time0 + A.cam.shutter * rng_f32(rng);
                    } else {                                     // viewport.rs:452-470 (x outer, y inner)
                        const uint32_t sx = s / A.s_root, sy = s % A.s_root;
                        jx = (float)pi + (((float)sx + rng_f32(rng)) / (float)A.s_root);
                        jy = (float)pj + (((float)sy + rng_f32(rng)) / (float)A.s_root);
                        tm = 0.0f;
                    }
                }
                d = (p00 + du * jx) + dv * jy;
            }
            n_rays++;
        }

        // ---- 3. one closest-hit query + scatter for every lane that owns a pixel ----------------
        if (have) {
            int best = -1; float best_t = 0.0f;
            const bool no_query = A.depth == 0 && A.integrator != RTW_INTEGRATOR_NORMAL;   // `if depth < 1 { return black }` (ray_color.rs:14-16)
            if (no_query) {}
            else if (ACCEL == RTW_ACCEL_BVH) closest_bvh<MOVING>(sc, A.bvh, bvh_stack, o, d, tm, A.mint, A.maxt, best, best_t, n_nodes, n_tests);
            else closest_brute<MOVING>(sc, o, d, tm, A.mint, A.maxt, best, best_t);
            bool finished = false;
            if (no_query) { L = mk(0, 0, 0); finished = true; }
            else if (n_seg++, best < 0) {
                v3 miss;
                if (A.integrator == RTW_INTEGRATOR_BG_COLOR) miss = ld3(A.bg);
                else if (A.integrator == RTW_INTEGRATOR_FLAG) miss = mk(0.0f, 0.0f, 1.0f);
                else miss = sky_gradient(d);
                L = L + miss * thr;
                finished = true;
            } else {
                f4 g = sc.geom[best];
                v3 c = mk(g.x, g.y, g.z);
                if (MOVING) { f4 vv = sc.vel[best]; c = c + mk(vv.x, vv.y, vv.z) * tm; }
                const v3 point = o + d * best_t;                 // r.at(x)
                const v3 normal = unit(point - c);               // sphere.rs:127
                const DevMat mat = sc.mat[best];
                if (A.integrator == RTW_INTEGRATOR_NORMAL) {     // C++/src/tests.cpp:91
                    L = mk(normal.x + 1.0f, normal.y + 1.0f, normal.z + 1.0f) * 0.5f;
                    finished = true;
                } else if (A.integrator == RTW_INTEGRATOR_FLAG && mat.metallicness != 1.0f) {
                    L = mk(1.0f, 1.0f, 0.0f) * thr;              // glass_tests.rs:35-37
                    finished = true;
                } else {
                    const v3 cm = sphere_albedo(sc, mat, normal);
                    float cos_theta;
                    const v3 nd = on_hit(mat, normal, d, rng, cos_theta);
                    if (A.integrator == RTW_INTEGRATOR_BG_COLOR) {   // ray_color.rs:64-88, front-to-back
                        // lambertian_scatter_pdf (materials.rs:5-13); pdf == 0 makes the reference's
                        // `color * pdf / pdf` a 0/0
                        const float pdf = cos_theta > 0.0f ? cos_theta * 0.318309886183790671538f : 0.0f;
                        if (mat.metallicness != 1.0f && !(pdf > 0.0f)) poison = true;
                        L = L + ld3(mat.emitted) * thr;
                    }
                    thr = thr * cm;
                    o = point; d = nd;
                    k++;
                    if (k >= A.depth) {                          // depth exhausted: the innermost call returns black
                        if (A.integrator != RTW_INTEGRATOR_BG_COLOR) L = mk(0, 0, 0);
                        finished = true;
                    }
                }
            }
            if (finished) {
                if (poison) { const float qn = __builtin_nanf(""); L = mk(qn, qn, qn); }
                acc = acc + L;                                   // viewport.rs:299
                s++;
                if (s >= A.n_samples) {
                    v3 col = acc / (float)A.n_samples;           // viewport.rs:301
                    // gamma_correct (viewport.rs:207-213).  x^1 is x: skipping the libm call keeps the
                    // gamma == 1 output bit-identical to the CPU (ocml powf is not exact there).
                    if (A.inv_gamma != 1.0f) col = mk(powf(col.x, A.inv_gamma), powf(col.y, A.inv_gamma), powf(col.z, A.inv_gamma));
                    float *px = A.out + 3 * ((size_t)pk * A.width + pi);
                    px[0] = col.x; px[1] = col.y; px[2] = col.z;
                    if (col.x != col.x || col.y != col.y || col.z != col.z) n_nan++;
                    have = false;
                } else newpath = true;
            }
        }
    }

    // ---- counters: wave reduce, one atomic per wave ---------------------------------------------
    unsigned long long seg = n_seg, rays = n_rays, nans = n_nan, nodes = n_nodes;
    unsigned long long tests = ACCEL == RTW_ACCEL_BVH ? (unsigned long long)n_tests : (unsigned long long)n_seg * sc.n;
    for (int off = 32; off > 0; off >>= 1) {
        seg += __shfl_down(seg, off);
        rays += __shfl_down(rays, off);
        nans += __shfl_down(nans, off);
        nodes += __shfl_down(nodes, off);
        tests += __shfl_down(tests, off);
    }
    if (lane == 0) {
        atomicAdd(&A.stats[0], rays);
        atomicAdd(&A.stats[1], seg);
        atomicAdd(&A.stats[2], tests);
        atomicAdd(&A.stats[3], nodes);
        atomicAdd(&A.stats[4], nans);
    }
}

typedef void (*kernel_fn)(const KArgs);
static kernel_fn pick_kernel(bool moving, uint32_t accel) {
    if (accel == RTW_ACCEL_BVH) return moving ? render_kernel<true, RTW_ACCEL_BVH> : render_kernel<false, RTW_ACCEL_BVH>;
    return moving ? render_kernel<true, RTW_ACCEL_BRUTE> : render_kernel<false, RTW_ACCEL_BRUTE>;
}

void launch_render(const KArgs &a, bool moving, uint32_t accel, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(pick_kernel(moving, accel), dim3(grid), dim3(RTW_BLOCK), 0, stream, a);
}

uint32_t kernel_blocks_per_cu(bool moving, uint32_t accel) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pick_kernel(moving, accel), RTW_BLOCK, 0) != hipSuccess || n < 1) n = 1;
    return (uint32_t)(n > 8 ? 8 : n);
}

} // namespace rtw
