// rtw_kernels.hip -- the gfx950 render kernels.
//
// One launch does the reference's L3+L4+L5: the pixel/sample driver
// (Rust/src/viewport.rs:270-305 render_row, :430-478 render, :479-516 render_no_rand), the
// integrator (Rust/src/viewport/ray_color.rs:12-92, written front-to-back instead of recursively)
// and the closest-hit + scatter (objects/sphere.rs:99-147, objects/materials.rs:105-154).
//
// Execution shape common to both render kernels (wave64, CDNA4):
//   * persistent workgroups (4 waves); the work unit is (8x8 tile, chunk of 4 consecutive samples, pixel of
//     the tile); a wave takes 64 units -- one tile x chunk -- from the global queue with ONE atomic into a
//     wave-local reserve and hands them to its lanes as they run dry (ballot + mbcnt);
//   * a lane whose path ends starts the next sample of its unit at once ("regeneration");
//   * every finished path banks its radiance (12 B) in HBM; resolve_kernel then adds a pixel's samples IN
//     SAMPLE ORDER, so the per-pixel f32 sum has the reference's order (viewport.rs:299 `color +=`) although
//     a pixel's samples were traced by many lanes.
//
// render_brute: every lane tests every sphere in list order; {centre, r^2} arrive by wave-uniform
//   SCALAR loads, so the VALU ops take their sphere operands straight from SGPRs.
// render_bvh:   per-lane BVH traversal.  Lanes of a wave diverge (different node counts, leaves at
//   different moments, paths ending at different moments), so the wave runs a tiny scheduler: every
//   lane is in one of three phases -- TRAVERSE (inner-node visits = two slab tests each), LEAF (one exact
//   sphere test), SHADE (finish a segment: scatter / sky, next sample, next unit, set up the next
//   traversal) -- and each loop trip executes ONE phase (ballot + popcount), the other lanes keep their
//   state (traversal stack in LDS) and wait.  That turns three nested divergent loops into one loop whose
//   body runs with most lanes live.  The phase is encoded in the lane's node register (Trav::node).
// GEOM builds of both add the rest of Scene::collision_normal -- quads, instances, media -- to the SHADE step.
//
// What this code is tuned for: on MI355X the render time follows the number of VALU instructions issued
//  -- so uniform work is kept on the scalar unit (lane-mask logic, counters as ballot popcounts,
// the work-unit decode once per 64 items) and the divergent branches share what they can (one unit(direction)).
#include "rtw_kernels.h"
#include <type_traits>

#ifndef RTW_MAX_TRIPS
#define RTW_MAX_TRIPS 0x30000000u   /* loop trips after which a wave of a persistent kernel gives up (safety valve; ~6000x what the bench frame needs per wave) */
#endif

namespace rtw {

// ================================================================================================
// shared pieces
// ================================================================================================

// SPEC != 0 is the specialisation for the common configuration -- ray_color_gradient, render_row sampler, depth >= 1
// (every BASELINE config): the integrator/sampler switches fold away at compile time.  SPEC == 1 additionally knows
// that no sphere carries an image texture (no atan2f/acosf code at all); SPEC == 2 keeps the texture lookup (C5).
// SPEC == 3 is SPEC == 1 with RTW_FLAG_CHUNK_SUMS (one partial sum per work unit in the bank); the generic build honours the flag at run
// time; a runtime test of it in the SPEC == 1 / 2 builds cost the bench frame 0.8 % (gpurun_out/r02_ab_chunk.log), hence a build of its own.
// SPEC == 4 is the reference's own demo configuration, presentation_image (main.rs:89-419): ray_color_bg_color with the render_row sampler; SPEC == 5 is
// Rust2's: its `ray_color` through its fixed-centre `render_row` (Rust2/src/viewport.rs:87-114); SPEC == 6 is ray_color_gradient through the stratified
// `Viewport::render` (viewport.rs:430-478: the reference's serial driver, quad_test's).  All three keep the generic build's step and fold the switches.
// SPEC == 0 reads everything from the (wave-uniform) kernel arguments.
constexpr bool gradient_spec(int spec) { return spec >= 1 && spec <= 3; }
template <int SPEC> __device__ __forceinline__ uint32_t integ(const KArgs &A) {
    return SPEC == 5 ? (uint32_t)RTW_INTEGRATOR_RUST2 : SPEC == 4 ? (uint32_t)RTW_INTEGRATOR_BG_COLOR : SPEC ? (uint32_t)RTW_INTEGRATOR_GRADIENT : A.integrator;
}
template <int SPEC> __device__ __forceinline__ uint32_t samp(const KArgs &A) {
    return SPEC == 6 ? (uint32_t)RTW_SAMPLER_STRATIFIED : SPEC == 5 ? (uint32_t)RTW_SAMPLER_CENTRES : SPEC ? (uint32_t)RTW_SAMPLER_ROW : A.sampler;
}

struct Pixel {            // the work unit a lane owns: a run of consecutive samples of one pixel.  Four registers (they live through every step of
                          // the persistent loop, and the specialised builds have 72): the shim keeps width, height < 2^16, samples < 2^24, units <= 255 samples
    uint32_t ij;          // column | image row << 16
    uint32_t rng_base;    // hash of (seed, pixel)
    uint32_t s;           // next sample index | samples of this unit still to trace (this one included) << 24
    uint32_t slot;        // where sample s goes in KArgs.samples (advances with s)
};
__device__ __forceinline__ uint32_t px_i(const Pixel &px) { return px.ij & 0xFFFFu; }
__device__ __forceinline__ uint32_t px_j(const Pixel &px) { return px.ij >> 16; }
__device__ __forceinline__ uint32_t px_s(const Pixel &px) { return px.s & 0xFFFFFFu; }
// the unit's next sample; true when that was its last
__device__ __forceinline__ bool px_advance(Pixel &px) { px.s += 1u - (1u << 24); return px.s < (1u << 24); }

struct Path {             // the path a lane is tracing
    v3 o, d;              // current ray
    float tm;             // ray.time
    v3 thr;               // product of col_mod so far (front-to-back)
    v3 L;                 // radiance gathered so far (emission; sky/background at the end)
    uint32_t k;           // closest-hit queries done
    Rng rng;
    bool poison;          // bg_color's 0/0 (ray_color.rs:72-75)
};

// Wave-local reserve of work items: the wave takes 64 consecutive items (one tile x chunk) from the global
// queue with ONE atomic and hands them to its lanes as they run dry, so the queue word sees one atomic per
// 64 units instead of one per refill (a single word saturates near 88 dequeues/us, MI355X_MICROARCH.md).
struct Reserve {                             // wave-uniform
    uint32_t next, end;                      // items [next, end) of ONE 64-item block: the 64 pixels of one (tile, chunk) unit
    uint32_t limit;                          // the wave holds items [next, limit) of sub-queue `sub`: whole 64-item blocks (fetch_pixel: guided grabs)
    uint32_t sub, tries;                     // the sub-queue this wave takes from, and how many sub-queues it has found empty
    uint32_t i0, k0;                         // the block's tile: first column, first compact row
    uint32_t s0, s1;                         // the block's chunk: samples [s0, s1)
    uint32_t slot0;                          // the block's place in the sample bank: slot of (its tile, its first sample, pixel 0)
#ifdef RTW_ENDTIMES
    unsigned long long t_dry;                // diagnostic build: device-wide time at which this wave first found the queue empty (0: not yet)
#endif
};

// Pull the next work item for the lanes with `need` set.  Must be called by all lanes of the wave
// that are currently active; returns true for lanes that got a valid unit.  `exhausted` is set for
// lanes that found the queue empty.  Lanes that get nothing (reserve ran out mid-way, padding pixel)
// simply ask again on the next trip.
// Sub-queue `sub` holds the tiles at queue positions sub, sub + S, .. (S = 2^sub_shift); its blocks are numbered tile by tile, a tile's units in
// sample order.  With the three regions of unit length: t1 / t2 = its tiles in region 1 / regions 1 + 2, b1 / b2 = the blocks before region 2 / 3.
struct SubQ { uint32_t t1, t2, b1, b2, total; };
__device__ __forceinline__ SubQ subq_layout(const KArgs &A, uint32_t sub) {
    const uint32_t up = (1u << A.sub_shift) - 1u - sub;            // (x + up) >> sub_shift = the queue positions below x that belong to this sub-queue
    SubQ q;
    q.t1 = (A.reg_q1 + up) >> A.sub_shift; q.t2 = (A.reg_q2 + up) >> A.sub_shift;
    const uint32_t t3 = (A.n_tiles + up) >> A.sub_shift;
    q.b1 = q.t1 * A.reg_nc[0]; q.b2 = q.b1 + (q.t2 - q.t1) * A.reg_nc[1];
    q.total = (q.b2 + (t3 - q.t2) * A.reg_nc[2]) * 64u;
    return q;
}

__device__ __forceinline__ bool fetch_pixel(const KArgs &A, bool need, Pixel &px, bool &exhausted, Reserve &rs) {
    const unsigned long long m = ballot64(need);
    if (m == 0ull) return false;
    if (rs.next == rs.end) {                                      // wave-uniform: refill the reserve
        if (rs.end == rs.limit) {
            // ... from the work queue.  The queue is 2^sub_shift (8, or 1 for small launches) sub-queues, each with a counter of its own 256 B
            // from the next: sub-queue j holds the tiles at queue positions j, j + 8, .. (so each keeps the host's order), a wave starts on
            // sub-queue blockIdx % 8 -- workgroups go round-robin to the 8 XCDs, so that is "the XCD's own" -- and moves on to the next one
            // when it finds one empty, until it has found all of them empty.  Why: a grab is an atomic with return value on which the
            // whole wave waits, and 6144 waves on ONE word made that wait ~2 ms of the 83 ms frame: one more such atomic per grab on the
            // same word costs +2.2 ms, on a word per XCD +0.4 ms (profiles/r02_ab_refill_probe.log).
            // A grab takes up to grab_max / 64 consecutive blocks (default 2) while plenty of work is left in the sub-queue -- what is
            // left, as this wave last saw it, divided by 4 x the waves that share it -- and single blocks towards the end and when
            // helping out on another sub-queue.  Consecutive blocks are consecutive sample chunks of the same 8x8 tile, so the wave's
            // lanes stay on the same 64 pixels longer: their rays are alike and its TRAVERSE / LEAF steps run fuller (0.623 -> 0.633 /
            // 0.505 -> 0.519 on the bench frame).  More blocks per grab fill the steps a little more (0.643 with a whole tile's 42) but
            // cost far more than that gains -- 4: +0 %, 8: +3 %, a tile: +29 % -- because a wave then sits on up to 17 ms of work the
            // others cannot take, and sizes its next grab by a queue position that old (profiles/r02_grab_sweep.log).
            const uint32_t n_sub = 1u << A.sub_shift;
            const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
            bool got = false;
            while (rs.tries < n_sub) {                            // (wave-uniform; at most n_sub failed grabs in a wave's life)
                const uint32_t total = subq_layout(A, rs.sub).total;     // items of sub-queue rs.sub
                uint32_t grab = 64u;
                if (rs.tries == 0u && rs.limit < total) {
                    grab = ((total - rs.limit) >> A.grab_shift) & ~63u;
                    grab = grab > A.grab_max ? A.grab_max : (grab < 64u ? 64u : grab);
                }
                uint32_t base = 0;
                if ((threadIdx.x & 63u) == leader) base = atomicAdd(A.queue + rs.sub * (RTW_QUEUE_STRIDE / 4u), grab);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
                if (base < total) {
                    rs.next = base;
                    rs.limit = total - base > grab ? base + grab : total;      // (no overflow: base < total)
                    got = true;
                    break;
                }
                rs.sub = (rs.sub + 1u) & (n_sub - 1u); rs.tries++; rs.limit = 0u;
            }
            if (!got) {
                rs.next = rs.limit = 0u;                          // nothing left anywhere: avail == 0 below tells the lanes
#ifdef RTW_ENDTIMES
                if (rs.t_dry == 0ull) rs.t_dry = wall_clock64();
#endif
            }
        }
        rs.end = rs.next + (rs.next < rs.limit ? 64u : 0u);       // the next 64-item block of what this wave holds
        if (rs.next < rs.limit) {
            // Work unit = (8x8 tile, chunk of chunk_len samples, pixel of the tile); the units of one tile are consecutive.  The
            // block's tile and chunk are the same for its 64 items: the three integer divisions run once per block, here.
            // The unit length depends on where in the queue the tile sits (KArgs.reg_*): three regions, so three cases of the same division.
            uint32_t u = rs.next >> 6;                             // block of the sub-queue
            const SubQ sq = subq_layout(A, rs.sub);
            uint32_t lt, chunk, len;
            if (u < sq.b1) { lt = u / A.reg_nc[0]; chunk = u - lt * A.reg_nc[0]; len = A.reg_len[0]; }
            else if (u < sq.b2) { u -= sq.b1; lt = u / A.reg_nc[1]; chunk = u - lt * A.reg_nc[1]; lt += sq.t1; len = A.reg_len[1]; }
            else { u -= sq.b2; lt = u / A.reg_nc[2]; chunk = u - lt * A.reg_nc[2]; lt += sq.t2; len = A.reg_len[2]; }
            const uint32_t qt = (lt << A.sub_shift) + rs.sub;      // ... -> queue position of its tile
            // queue position -> tile: raster order, or any permutation the host supplies (RTW_OPT_TILE_ORDER)
            const uint32_t tile = A.tile_order ? A.tile_order[qt] : qt;
            const uint32_t trow = tile / A.tiles_x, tcol = tile - trow * A.tiles_x;
            rs.i0 = tcol * 8u;
            rs.k0 = A.k_base + trow * 8u;
            rs.s0 = chunk * len;
            rs.s1 = rs.s0 + len < A.n_samples ? rs.s0 + len : A.n_samples;
            // [tile][sample][pixel] -- or, one partial sum per unit (RTW_FLAG_CHUNK_SUMS: a single region), [tile][chunk][pixel]
            rs.slot0 = (A.bank_len ? tile * A.n_chunks + chunk : tile * A.n_samples + rs.s0) * 64u;
        }
    }
    const uint32_t avail = rs.end - rs.next;
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    const uint32_t w = rs.next + rank;
    const uint32_t cnt = (uint32_t)__popcll(m);
    rs.next += cnt < avail ? cnt : avail;
    if (!need) return false;
    if (avail == 0u) { exhausted = true; return false; }          // the queue is empty
    if (rank >= avail) return false;                              // reserve ran out: next trip refills
    const uint32_t p = w & 63u;                                    // pixel of the tile
    const uint32_t i = rs.i0 + (p & 7u);
    const uint32_t k = rs.k0 + (p >> 3);                           // compact row of this partition
    if (!(i < A.width && k < A.k_end)) return false;               // padding item: ask again next trip
    uint32_t j = k;   // compact row -> image row (RtwParams row partition)
    if (A.part_count > 1) {
        // tiles are 8 rows tall and start on a multiple of 8, so with the usual 8-row blocks the block index is uniform too
        if (A.row_block == 8u) j = ((rs.k0 >> 3) * A.part_count + A.part_index) * 8u + (p >> 3);
        else j = ((k / A.row_block) * A.part_count + A.part_index) * A.row_block + (k % A.row_block);
    }
    px.ij = i | (j << 16);
    px.rng_base = rng_pixel_base(A.seed_lo, A.seed_hi, j * A.width + i);
    px.s = rs.s0 | ((rs.s1 - rs.s0) << 24);
    px.slot = rs.slot0 + (w & 63u);                              // [tile][sample][pixel of tile]: lanes that finish the same sample of neighbouring
                                                                   // pixels together fill whole sectors
    return rs.s0 < rs.s1;
}

// Camera ray of sample px.s (the sampler loops of viewport.rs / Rust2 viewport.rs).
// The camera (21 floats, the first 84 bytes of the kernel's argument block) is read HERE, with two scalar loads, each time a path starts --
// not through `A.cam`: values the optimiser sees as loop invariants are loaded once and kept for the whole persistent loop, which in this
// kernel (106 SGPRs, all in use) means spilled to lanes of a VGPR and brought back with a v_readlane -- a VALU instruction in a VALU-bound
// kernel -- at every use.  A scalar load from the (cached) argument block costs no VALU slot.
struct CamRegs { RtwCamera c; };
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f8v __attribute__((ext_vector_type(8)));
static_assert(sizeof(RtwCamera) == 84 && offsetof(KArgs, cam) == 0, "start_path reads KArgs.cam as 16 + 8 dwords at offset 0");
__device__ __forceinline__ RtwCamera load_camera() {
    f16v lo; f8v hi;
    asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx8 %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(lo), "=&s"(hi) : "s"(__builtin_amdgcn_kernarg_segment_ptr()));      // (early-clobber: the first load's result must not land on the base pair the second still reads)
    RtwCamera c;
    c.origin[0] = lo[0]; c.origin[1] = lo[1]; c.origin[2] = lo[2]; c.u[0] = lo[3]; c.u[1] = lo[4]; c.u[2] = lo[5];
    c.v[0] = lo[6]; c.v[1] = lo[7]; c.v[2] = lo[8]; c.pixel00[0] = lo[9]; c.pixel00[1] = lo[10]; c.pixel00[2] = lo[11];
    c.delta_u[0] = lo[12]; c.delta_u[1] = lo[13]; c.delta_u[2] = lo[14]; c.delta_v[0] = lo[15]; c.delta_v[1] = hi[0]; c.delta_v[2] = hi[1];
    c.lens_radius = hi[2]; c.time0 = hi[3]; c.shutter = hi[4];
    return c;
}

template <int SPEC>
__device__ __forceinline__ void start_path(const KArgs &A, const Pixel &px, Path &pt, Cen *cn = nullptr) {
    RTW_CEN(cn, CEN_START_PATH);
    const RtwCamera cam = load_camera();
    const v3 cam_o = ld3(cam.origin), p00 = ld3(cam.pixel00), du = ld3(cam.delta_u), dv = ld3(cam.delta_v);
    pt.rng = rng_start(px.rng_base, px_s(px));
    pt.thr = mk(1.0f, 1.0f, 1.0f); pt.L = mk(0, 0, 0); pt.poison = false; pt.k = 0;
    // (single assignment of pt.o / pt.d / pt.tm at the end: stores to different members on different
    //  branches get "sunk" into a phi of pointers by the optimiser, which forces the path state into scratch)
    v3 o, d; float tm = 0.0f;
    if (samp<SPEC>(A) == RTW_SAMPLER_NO_RAND) {                  // viewport.rs:498-503
        o = cam_o;
        d = (p00 + du * (float)px_i(px)) + dv * (float)px_j(px);
    } else {
        float jx, jy, rx, ry;
        if (samp<SPEC>(A) == RTW_SAMPLER_CENTRES) {              // Rust2/src/viewport.rs:92-104
            const uint32_t kx = px_s(px) / A.s_root, ly = px_s(px) % A.s_root;
            jx = ((float)px_i(px) + ((float)kx + 0.5f) / (float)A.s_root) / (float)A.width;
            jy = ((float)px_j(px) + ((float)ly + 0.5f) / (float)A.s_root) / (float)A.height;
            random_in_unit_disk(pt.rng, rx, ry, false, cn);
            o = cam_o + mk(rx, ry, 0.0f) * cam.lens_radius;
        } else {
            random_in_unit_disk(pt.rng, rx, ry, !SPEC && (A.flags & RTW_FLAG_CPP_DIFFUSE), cn);   // always drawn (viewport.rs:288)
            o = cam_o + (ld3(cam.u) * rx + ld3(cam.v) * ry) * cam.lens_radius;
            if (samp<SPEC>(A) == RTW_SAMPLER_ROW) {              // viewport.rs:290-297
                jx = rng_offset(pt.rng, (float)px_i(px));
                jy = rng_offset(pt.rng, (float)px_j(px));
                tm = cam.time0 + cam.shutter * rng_f32(pt.rng);
            } else {                                         // viewport.rs:452-470 (x outer, y inner)
                const uint32_t sx = px_s(px) / A.s_root, sy = px_s(px) % A.s_root;
                jx = (float)px_i(px) + (((float)sx + rng_f32(pt.rng)) / (float)A.s_root);
                jy = (float)px_j(px) + (((float)sy + rng_f32(pt.rng)) / (float)A.s_root);
            }
        }
        d = (p00 + du * jx) + dv * jy;
    }
    pt.o = o; pt.d = d; pt.tm = tm;
}

// ray_color_* when the closest-hit query found nothing: sky / background ends the path.
template <int SPEC>
__device__ __forceinline__ void shade_miss(const KArgs &A, Path &pt, v3 ud) {
    v3 miss;
    if (integ<SPEC>(A) == RTW_INTEGRATOR_BG_COLOR || integ<SPEC>(A) == RTW_INTEGRATOR_RUST2) miss = ld3(A.bg);
    else if (integ<SPEC>(A) == RTW_INTEGRATOR_FLAG) miss = mk(0.0f, 0.0f, 1.0f);
    else miss = sky_gradient(ud);
    // SPEC builds (ray_color_gradient): nothing has been gathered before the path ends, pt.L is 0 -- so L is not carried from step to step -- and
    // the oracle's `0 + x` is x itself except for x = -0, which the resolve pass's own `0 + ..` (the pixel's sum starts at +0) turns into the same +0
    if (gradient_spec(SPEC)) pt.L = miss * pt.thr;
    else pt.L = pt.L + miss * pt.thr;
}

// One step of ray_color_* at the closest hit, given as the reference's `Hit` (objects.rs:16-23): point, normal,
// col_mod, material.  Returns true when the path is finished (pt.L is then its radiance).
template <int SPEC>
__device__ __forceinline__ bool shade_surface(const KArgs &A, Path &pt, v3 ud, v3 point, v3 normal, v3 cm, MatP mat, v3 emitted, Cen *cn = nullptr) {
    if (integ<SPEC>(A) == RTW_INTEGRATOR_NORMAL) {             // C++/src/tests.cpp:91
        pt.L = mk(normal.x + 1.0f, normal.y + 1.0f, normal.z + 1.0f) * 0.5f;
        return true;
    }
    if (integ<SPEC>(A) == RTW_INTEGRATOR_FLAG && mat.metallicness != 1.0f) {
        pt.L = mk(1.0f, 1.0f, 0.0f) * pt.thr;                // glass_tests.rs:35-37
        return true;
    }
    if (integ<SPEC>(A) == RTW_INTEGRATOR_RUST2) {              // Rust2/src/viewport/ray_color.rs:17-31, front-to-back
        const v3 nd2 = on_hit_rust2(mat, normal, pt.d, pt.rng, A.flags);
        pt.L = pt.L + emitted * pt.thr;
        pt.thr = pt.thr * cm;
        pt.o = point; pt.d = nd2;
        pt.k++;
        if (pt.k >= A.depth) { pt.L = pt.L + ld3(A.bg) * pt.thr; return true; }   // depth 0 returns bg_color
        return false;
    }
    float cos_theta;
    const v3 nd = on_hit(mat, point, normal, pt.d, ud, pt.rng, cos_theta, SPEC ? 0u : A.flags, cn);
    if (integ<SPEC>(A) == RTW_INTEGRATOR_BG_COLOR) {           // ray_color.rs:64-88, front-to-back
        // lambertian_scatter_pdf (materials.rs:5-13); pdf == 0 makes the reference's `color * pdf / pdf` a 0/0
        const float pdf = cos_theta > 0.0f ? cos_theta * 0.318309886183790671538f : 0.0f;
        if (mat.metallicness != 1.0f && !(pdf > 0.0f)) pt.poison = true;
        pt.L = pt.L + emitted * pt.thr;
    }
    pt.thr = pt.thr * cm;
    pt.o = point; pt.d = nd;
    pt.k++;
    if (pt.k >= A.depth) {                                   // depth exhausted: the innermost call returns black
        RTW_CEN(cn, CEN_DEPTH_END);
        if (integ<SPEC>(A) != RTW_INTEGRATOR_BG_COLOR) pt.L = mk(0, 0, 0);
        return true;
    }
    return false;
}

// ... at top-level sphere `best` (Sphere::collision_normal's Hit, sphere.rs:124-146).
template <bool MOVING, int SPEC>
__device__ __forceinline__ bool shade_hit(const KArgs &A, Path &pt, v3 ud, int best, float best_t, Cen *cn = nullptr) {
    RTW_CEN(cn, CEN_HIT);
    const DevScene &sc = A.sc;
    f4 g = sc.geom[best];
    v3 c = mk(g.x, g.y, g.z);
    if (MOVING) { f4 vv = sc.vel[best]; c = c + mk(vv.x, vv.y, vv.z) * pt.tm; }
    const v3 point = pt.o + pt.d * best_t;                   // r.at(x)
    const v3 normal = unit(point - c);                       // sphere.rs:127
    const DevMat mat = sc.mat[best];
    v3 cm, emitted = ld3(mat.emitted);
    if ((SPEC == 0 || SPEC == 5) && integ<SPEC>(A) == RTW_INTEGRATOR_RUST2 && mat.tex >= 0) rust2_sphere_color(sc, mat, normal, cm, emitted);     // Rust2's own lookup rule
    else cm = (SPEC == 1 || SPEC == 3) ? ld3(mat.cm) : sphere_albedo(sc, mat, normal);
    return shade_surface<SPEC>(A, pt, ud, point, normal, cm, mat_params(mat), emitted, cn);
}

template <bool MOVING, int SPEC>
__device__ __forceinline__ bool shade(const KArgs &A, Path &pt, int best, float best_t, Cen *cn = nullptr) {
    RTW_CEN(cn, CEN_INFLIGHT);
    // unit(direction) feeds the scatter of the lanes that hit (materials.rs:111,143) and the sky of the lanes that missed
    // (ray_color.rs:38): computed once, ahead of the divergent branches, instead of once in each
    const v3 ud = unit(pt.d);
    if (best < 0) { RTW_CEN(cn, CEN_MISS); shade_miss<SPEC>(A, pt, ud); return true; }
    return shade_hit<MOVING, SPEC>(A, pt, ud, best, best_t, cn);
}

// ---- specialised builds: a SHADE step in two halves around ONE rejection loop (sample_ball_or_disk, rtw_device.h) -------------------------
// The hit half of shade<>() for the gradient integrator, for a lane whose path goes on (the kernel has dealt with misses and exhausted depths):
// point, normal, material and Material::on_hit up to its random unit vector.  pt.o / pt.thr are final afterwards; pt.d is final for a
// dielectric hit, and for a diffuse / metallic one (need_ball) it holds the mirror direction until on_hit_second() has the unit vector
// (nrm, metal, front: what that half needs of the hit).  `ud` = unit(pt.d), from the caller.  The bounce count lives in the kernel's flag word.
template <bool MOVING, int SPEC>
__device__ __forceinline__ void shade_hit_first(const KArgs &A, Path &pt, v3 ud, int best, float best_t, bool &need_ball, v3 &nrm, float &metal, bool &front, Cen *cn = nullptr) {
    RTW_CEN(cn, CEN_HIT);
    const DevScene &sc = A.sc;
    f4 g = sc.geom[best];
    v3 c = mk(g.x, g.y, g.z);
    if (MOVING) { f4 vv = sc.vel[best]; c = c + mk(vv.x, vv.y, vv.z) * pt.tm; }
    const v3 point = pt.o + pt.d * best_t;                   // r.at(x)
    const v3 normal = unit(point - c);                       // sphere.rs:127
    const DevMat mat = sc.mat[best];
    const v3 cm = (SPEC == 1 || SPEC == 3) ? ld3(mat.cm) : sphere_albedo(sc, mat, normal);
    v3 next; bool fr;
    const bool complete = on_hit_first(mat_params(mat), normal, pt.d, ud, pt.rng, next, fr, cn);
    pt.thr = pt.thr * cm;
    pt.o = point; pt.d = next;
    need_ball = !complete; nrm = normal; metal = mat.metallicness; front = fr;
}

// start_path<>() for the render_row sampler (viewport.rs:286-297), in two halves around the lens-disk draw.
__device__ __forceinline__ void start_path_first(const Pixel &px, Path &pt, Cen *cn = nullptr) {
    RTW_CEN(cn, CEN_START_PATH);
    pt.rng = rng_start(px.rng_base, px_s(px));
    pt.thr = mk(1.0f, 1.0f, 1.0f); pt.L = mk(0, 0, 0); pt.poison = false;
}
__device__ __forceinline__ void start_path_second(const Pixel &px, Path &pt, float rx, float ry) {
    const RtwCamera cam = load_camera();
    const v3 cam_o = ld3(cam.origin), p00 = ld3(cam.pixel00), du = ld3(cam.delta_u), dv = ld3(cam.delta_v);
    const v3 o = cam_o + (ld3(cam.u) * rx + ld3(cam.v) * ry) * cam.lens_radius;       // viewport.rs:288-289
    const float jx = rng_offset(pt.rng, (float)px_i(px));                           // viewport.rs:290-297
    const float jy = rng_offset(pt.rng, (float)px_j(px));
    const float tm = cam.time0 + cam.shutter * rng_f32(pt.rng);
    const v3 d = (p00 + du * jx) + dv * jy;
    pt.o = o; pt.d = d; pt.tm = tm;
}

// Scenes with quads / instances (generic build only): finish Scene::collision_normal (viewport.rs:136-150) for
// the query whose sphere part returned (best, best_t), then shade whichever object won.
template <bool MOVING, int SPEC>
__device__ __forceinline__ bool shade_geom(const KArgs &A, Path &pt, int best, float best_t, uint32_t &n_sph, uint32_t &n_quad) {
    GeomHit h;
    if (geom_closest(A.sc, A.geom, pt.o, pt.d, pt.tm, A.mint, A.maxt, best >= 0, best_t, pt.rng, h, n_sph, n_quad)) {
        mat_derive(h.m);
        return shade_surface<SPEC>(A, pt, unit(pt.d), h.point, h.normal, h.cm, h.m, h.emitted);
    }
    return shade<MOVING, SPEC>(A, pt, best, best_t);
}

// A path ended: bank its radiance in the sample buffer (the resolve kernel adds the samples of a pixel
// in sample order, viewport.rs:299).  Returns true when the unit is done.
template <int SPEC>
__device__ __forceinline__ bool finish_path(const KArgs &A, Pixel &px, Path &pt, Cen *cn = nullptr) {
    RTW_CEN(cn, CEN_BANK);
    if (pt.poison) { const float qn = __builtin_nanf(""); pt.L = mk(qn, qn, qn); }
    float3 *dst = reinterpret_cast<float3 *>(A.samples + 3 * (size_t)px.slot);
    if (SPEC == 3 || (SPEC == 0 && (A.flags & RTW_FLAG_CHUNK_SUMS))) {        // one slot per unit, the unit's samples added into it in sample order
        if (px_s(px) & (RTW_SUM_CHUNK - 1u)) {      // (chunk_len == RTW_SUM_CHUNK in this mode, units start on multiples of it)
            const float3 acc = *dst;
            pt.L = mk(acc.x, acc.y, acc.z) + pt.L;
        }
        *dst = make_float3(pt.L.x, pt.L.y, pt.L.z);
        return px_advance(px);
    }
    *dst = make_float3(pt.L.x, pt.L.y, pt.L.z);   // one 12-byte store
    px.slot += 64u;
    return px_advance(px);
}

// The pixel stage of the driver (viewport.rs:299-301): color = sum of the samples IN ORDER, / samples,
// powf(1/gamma), one coalesced-by-tile 12-byte store per pixel.  One lane per pixel; the per-sample radiances
// were banked by the render kernel as [tile][sample][pixel of tile].
__global__ __launch_bounds__(RTW_BLOCK) void resolve_kernel(const KArgs A) {
    const uint32_t w = blockIdx.x * RTW_BLOCK + threadIdx.x;
    const uint32_t tile = w >> 6, p = w & 63u;
    uint32_t nan = 0;
    if (tile < A.n_tiles) {
        const uint32_t tcol = tile % A.tiles_x, trow = tile / A.tiles_x;
        const uint32_t i = tcol * 8u + (p & 7u), k = A.k_base + trow * 8u + (p >> 3);
        if (i < A.width && k < A.k_end) {
            v3 acc = mk(0, 0, 0);
            // one slot per sample, in sample order -- however the render kernel cut them into units -- or (RTW_FLAG_CHUNK_SUMS) one per unit
            const uint32_t n_slots = A.bank_len ? A.n_chunks : A.n_samples;
            const float *src = A.samples + 3 * ((size_t)tile * n_slots * 64u + p);
            for (uint32_t q = 0; q < n_slots; q++) acc = acc + ld3(src + 3 * 64u * (size_t)q);     // viewport.rs:299
            v3 col = acc / (float)A.n_samples;                   // viewport.rs:301
            // gamma_correct (viewport.rs:207-213).  x^1 is x: skipping the libm call keeps the gamma == 1
            // output bit-identical to the CPU (ocml powf is not exact there).
            if (A.inv_gamma != 1.0f) col = mk(powf(col.x, A.inv_gamma), powf(col.y, A.inv_gamma), powf(col.z, A.inv_gamma));
            *reinterpret_cast<float3 *>(A.out + 3 * ((size_t)k * A.width + i)) = make_float3(col.x, col.y, col.z);
            if (col.x != col.x || col.y != col.y || col.z != col.z) nan = 1;
        }
    }
    const unsigned long long m = ballot64(nan != 0);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(&A.stats[4], (unsigned long long)__popcll(m));
}

__device__ __forceinline__ void flush_counters(const KArgs &A, uint32_t n_seg, uint32_t n_rays,
                                               unsigned long long tests, uint32_t n_nodes) {
    unsigned long long seg = n_seg, rays = n_rays, nodes = n_nodes;
    for (int off = 32; off > 0; off >>= 1) {
        seg += __shfl_down(seg, off);
        rays += __shfl_down(rays, off);
        nodes += __shfl_down(nodes, off);
        tests += __shfl_down(tests, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&A.stats[0], rays);
        atomicAdd(&A.stats[1], seg);
        atomicAdd(&A.stats[2], tests);
        atomicAdd(&A.stats[3], nodes);
    }
}

__device__ __forceinline__ void flush_tests(const KArgs &A, uint32_t n) {
    unsigned long long q = n;
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&A.stats[2], q);
}

__device__ __forceinline__ void flush_quads(const KArgs &A, uint32_t n_quad) {
    unsigned long long q = n_quad;
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off);
    if ((threadIdx.x & 63u) == 0) atomicAdd(&A.stats[14], q);
}

// ================================================================================================
// brute force: closest hit in list order (camera_tests.rs:19-33; `min_hit == None || min_hit > i`)
// ================================================================================================
template <bool MOVING>
__device__ __forceinline__ void closest_brute(const DevScene &sc, v3 o, v3 d, float tm, float mint, float maxt,
                                              int &best, float &best_t) {
    cf4_ptr geom = (cf4_ptr)(uintptr_t)sc.geom;
    cf4_ptr vel = (cf4_ptr)(uintptr_t)sc.vel;
    const float a = dot(d, d);
    const float ra = rcp_refined(a);
    const bool a_plain = ballot64(!in_range(a, 0x1p-20f, 0x1p20f)) == 0ull;      // see sphere_root()
    best = -1; best_t = 0.0f;
    const uint32_t n = sc.n;
    auto test = [&](f4 g, f4 vv, uint32_t s) {
        float cx = g.x, cy = g.y, cz = g.z;
        if (MOVING) { cx = cx + vv.x * tm; cy = cy + vv.y * tm; cz = cz + vv.z * tm; }   // sphere.rs:100  origin + velocity * r.time
        float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
        float b = ocx * d.x + ocy * d.y + ocz * d.z;
        float c = (ocx * ocx + ocy * ocy + ocz * ocz) - g.w;
        float disc = b * b - a * c;
        if (!(disc < 0.0f)) {
            const float x = sphere_root(b, disc, a, ra, a_plain, mint);
            const bool take = !(x < mint || x > maxt) && (best < 0 || best_t > x);      // (select form: no exec-mask regions)
            best = take ? (int)s : best; best_t = take ? x : best_t;
        }
    };
    const f4 zero = { 0, 0, 0, 0 };
    uint32_t s = 0;
    // four spheres per trip, software-pipelined: the scalar loads (one s_load_dwordx16) of the NEXT four are
    // issued before the current four are tested, so their latency hides behind ~70 VALU ops; list order is kept
    f4 n0 = zero, n1 = zero, n2 = zero, n3 = zero, w0 = zero, w1 = zero, w2 = zero, w3 = zero;
    if (n >= 4) {
        n0 = geom[0]; n1 = geom[1]; n2 = geom[2]; n3 = geom[3];
        if (MOVING) { w0 = vel[0]; w1 = vel[1]; w2 = vel[2]; w3 = vel[3]; }
    }
    for (; s + 4 <= n; s += 4) {
        const f4 g0 = n0, g1 = n1, g2 = n2, g3 = n3, v0 = w0, v1 = w1, v2 = w2, v3 = w3;
        if (s + 8 <= n) {
            n0 = geom[s + 4]; n1 = geom[s + 5]; n2 = geom[s + 6]; n3 = geom[s + 7];
            if (MOVING) { w0 = vel[s + 4]; w1 = vel[s + 5]; w2 = vel[s + 6]; w3 = vel[s + 7]; }
        }
        test(g0, v0, s); test(g1, v1, s + 1); test(g2, v2, s + 2); test(g3, v3, s + 3);
    }
    for (; s < n; ++s) test(geom[s], MOVING ? vel[s] : zero, s);
}

#ifndef RTW_GEOM_BRUTE_WAVES
#define RTW_GEOM_BRUTE_WAVES 4        /* waves per SIMD the GEOM list-walk build is compiled for: presentation_image 3 / 4 / 5 / 6 = 213 / 188 / 186 / 228 ms (profiles/r02_geom_waves.log) */
#endif
template <bool MOVING, int SPEC, bool GEOM>
__global__ __launch_bounds__(RTW_BLOCK, GEOM ? RTW_GEOM_BRUTE_WAVES : 1) void render_brute(const KArgs A) {
    bool dead = false, have = false, newpath = false;
    Pixel px; px.ij = px.rng_base = px.s = px.slot = 0;
    Reserve rs; rs.next = rs.end = rs.limit = rs.i0 = rs.k0 = rs.s0 = rs.s1 = rs.slot0 = rs.tries = 0; rs.sub = blockIdx.x & ((1u << A.sub_shift) - 1u);
#ifdef RTW_ENDTIMES
    rs.t_dry = 0ull;
#endif
    Path pt; pt.o = pt.d = pt.L = mk(0, 0, 0); pt.thr = mk(1, 1, 1); pt.tm = 0; pt.k = 0; pt.poison = false; pt.rng.state = 0; pt.rng.inc = 1;
    uint32_t n_seg = 0, n_rays = 0, n_isph = 0, n_quad = 0;
    uint32_t trips = 0; bool aborted = false;

    for (;;) {
        if (fetch_pixel(A, !have && !dead, px, dead, rs)) { have = true; newpath = true; }
        if (ballot64(!dead) == 0ull) break;
        if (++trips > RTW_MAX_TRIPS) { aborted = true; break; }       // safety valve, as in render_bvh
        if (have && newpath) { newpath = false; start_path<SPEC>(A, px, pt); n_rays++; }
        if (have) {
            bool finished;
            if (!SPEC && A.depth == 0 && A.integrator != RTW_INTEGRATOR_NORMAL) {   // `if depth < 1 { return black }` (ray_color.rs:14-16)
                pt.L = A.integrator == RTW_INTEGRATOR_RUST2 ? ld3(A.bg) : mk(0, 0, 0); finished = true;
            } else {
                int best; float best_t;
                closest_brute<MOVING>(A.sc, pt.o, pt.d, pt.tm, A.mint, A.maxt, best, best_t);
                n_seg++;
                finished = GEOM ? shade_geom<MOVING, SPEC>(A, pt, best, best_t, n_isph, n_quad) : shade<MOVING, SPEC>(A, pt, best, best_t);
            }
            if (finished) {
                if (finish_path<SPEC>(A, px, pt)) have = false;
                else newpath = true;
            }
        }
    }
    flush_counters(A, n_seg, n_rays, (unsigned long long)n_seg * A.sc.n + n_isph, 0);
    if (GEOM) flush_quads(A, n_quad);
    if (aborted && (threadIdx.x & 63u) == 0) atomicAdd(&A.stats[23], 1ull);
}

// ================================================================================================
// BVH
// ================================================================================================

// Exact sphere test for the BVH paths (same arithmetic as closest_brute).  Candidate order is not list
// order here, so ties are resolved explicitly toward the lower index -- the sphere `min_hit > i`
// keeps in list order.
template <bool MOVING>
__device__ __forceinline__ void exact_sphere(f4 g, f4 vv, uint32_t s, v3 o, v3 d, float tm, float a, float ra, bool a_plain, float mint, float maxt,
                                             int &best, float &best_t) {
    float cx = g.x, cy = g.y, cz = g.z;
    if (MOVING) { cx = cx + vv.x * tm; cy = cy + vv.y * tm; cz = cz + vv.z * tm; }
    float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
    float b = ocx * d.x + ocy * d.y + ocz * d.z;
    float c = (ocx * ocx + ocy * ocy + ocz * ocz) - g.w;
    float disc = b * b - a * c;
    if (!(disc < 0.0f)) {
        const float x = sphere_root(b, disc, a, ra, a_plain, mint);
        // (select form, no exec-mask regions: a NaN root fails `x < best_t` and `x == best_t` alike, as it does in the nested form)
        const bool take = !(x < mint || x > maxt) && (x < best_t || (x == best_t && s < (uint32_t)best));
        best = take ? (int)s : best; best_t = take ? x : best_t;
    }
}

// The tree only PRUNES; every surviving candidate runs the exact reference arithmetic above, so the
// result is identical to closest_brute (tests/test_gpu_parity.py checks it bit for bit).  Pruning is
// made safe against the reference's own f32 rounding (DESIGN.md "Conservative traversal"):
//   the reference reports a hit when fl(b*b - a*c) >= 0, which implies the ray passes within
//   sqrt(r^2 + K u (|oc|^2 + r^2)) of the centre (K = 24 >= the 15 the error analysis needs, u = 2^-24),
//   and its t can be earlier than the geometric entry by at most sqrt(K u (|oc|^2 + r^2)) / |d|.
// So each ray is thickened by rho (slab tests against boxes inflated by rho: folded into the per-ray
// constants, zero cost per box) and the t-interval is widened by tau.  Spheres much larger than the
// rest ("big", e.g. the ground) stay outside the tree and are tested exactly first, which both keeps
// rho/tau small and gives an early best_t.
#define RTW_KU 1.4305115e-6f    /* 24 * 2^-24 */
#ifndef RTW_S_HI
#define RTW_S_HI 52u            /* lanes waiting in SHADE that trigger a SHADE step.  Round 1 (raster tile order): 48: 26.93, 56: 27.08, 60/64: 26.94 G segments/s,
                                   C4 loses 3 % at 56.  Round 2 (scattered tile order, profiles/r02_ab_s_hi.log): 44 / 48 / 52 / 56 = 26.92 / 27.37 / 27.61 / 27.52
                                   on the bench frame; 52 against 48 on the other configs: C2 +3.6 %, C4 -1.3 %, C5 +0.0 % */
#endif
#ifndef RTW_TRAV_UNROLL
#define RTW_TRAV_UNROLL 3       /* node visits per scheduling decision (1: 13.2, 2: 14.4, 3: 14.9, 4: 14.2 Gsegments/s) */
#endif
#ifndef RTW_T_LO
#define RTW_T_LO 6u             /* below this many lanes in TRAVERSE and in LEAF, SHADE runs anyway */
#endif
#ifndef RTW_T_LO_DRAIN
#define RTW_T_LO_DRAIN RTW_T_LO /* the same once the work queue is empty.  1 ("SHADE only when no lane traverses any more") was measured: the bench frame and C4
                                   lose 0.6 %, C2 gains 1.6 %, `First frame` 5 %, an eighth of the bench frame nothing (profiles/r02_drain_ab.log) */
#endif

struct Trav {                // traversal state of one lane
    int node;                // the lane's PHASE is encoded here: 0 <= node < DEAD an inner node to visit (TRAVERSE; LDS variant: its byte offset);
                             // above END (as unsigned) the leaf ~node to test (LEAF; LDS variant: 16-bit codes, zero-extended);
                             // END: no query pending, the lane waits for a SHADE step; DEAD: finished
    uint32_t sp;             // LDS byte ADDRESS of the TOP entry of this lane's stack, base + (level * RTW_BLOCK + threadIdx.x) * sizeof(entry);
                             // level 0 holds the END sentinel, so a pop never has to ask whether the stack is empty
    int best; float best_t;  // closest accepted hit so far (best_t starts at maxt)
    float a, ra;             // d.d and rcp_refined(d.d) (sphere_root)
    float ix, iy, iz;        // 1/d
    float kpx, kpy, kpz;     // global-node variant: -(o + rho) / d (goes with a box's lo planes); LDS variant: the NEAR planes' constant
    float kmx, kmy, kmz;     // global-node variant: -(o - rho) / d (hi planes);                    LDS variant: the FAR planes' constant
    uint32_t selx, sely, selz;   // LDS variant: v_perm_b32 selector per axis, identity when the ray runs along +axis, half-swap otherwise
    float tau_t, lo_lim, hi_lim;
};

// Phase codes (stack entries are 16-bit when the nodes live in LDS, 32-bit otherwise).  Each phase test is ONE compare.
template <class S> struct Code;
template <> struct Code<short> { enum : int { END = 0x7FFF, DEAD = 0x7FFE }; };
template <> struct Code<int> { enum : int { END = 0x7FFFFFFF, DEAD = 0x7FFFFFFE }; };
template <class S> __device__ __forceinline__ bool in_trav(int node) { return (uint32_t)node < (uint32_t)Code<S>::DEAD; }
template <class S> __device__ __forceinline__ bool in_leaf(int node) { return (uint32_t)node > (uint32_t)Code<S>::END; }
// sphere index of a LEAF code: ~code, within the width of a stack entry
template <class S> __device__ __forceinline__ uint32_t leaf_sphere(int node) { return (uint32_t)node ^ (sizeof(S) == 2 ? 0xFFFFu : 0xFFFFFFFFu); }
template <class S> __device__ __forceinline__ bool in_shade(int node) { return node == (int)Code<S>::END; }

// Lanes of the wave for which `c` holds, as a 32-bit SGPR value.  The empty asm hides the popcount's origin from the
// optimiser, which otherwise carries it as 64 bits and compares it with VALU v_cmp_*_u64 on scalar operands.
__device__ __forceinline__ uint32_t lanes_in(bool c) {
    uint32_t n = (uint32_t)__popcll(ballot64(c));
    asm volatile("" : "+s"(n));
    return n;
}

// Begin a closest-hit query: big spheres, per-ray constants, root (which sets the phase).
// What a query's begin needs of DevBvh -- big_geom .. abs_max, 16 dwords -- read from the argument block with one scalar load, for the
// reason given at load_camera().
struct BvhBegin {
    const f4 *big_geom; const f4 *big_vel; const uint32_t *big_index;
    uint32_t n_big, depth; int32_t root;
    float cx, cy, cz, centre_radius, r_max2, inv_2rmin, abs_max;
};
static_assert(sizeof(BvhBegin) == 64 && offsetof(DevBvh, abs_max) - offsetof(DevBvh, big_geom) == 60 && offsetof(DevBvh, big_vel) == offsetof(DevBvh, big_geom) + 8 &&
              offsetof(DevBvh, n_big) == offsetof(DevBvh, big_geom) + 24 && offsetof(DevBvh, root) == offsetof(DevBvh, big_geom) + 32 &&
              offsetof(DevBvh, cx) == offsetof(DevBvh, big_geom) + 36, "BvhBegin mirrors DevBvh from big_geom on");
__device__ __forceinline__ BvhBegin load_bvh_begin() {
    typedef uint32_t u16v __attribute__((ext_vector_type(16)));
    u16v r;
    asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(__builtin_amdgcn_kernarg_segment_ptr()),
                 "n"(offsetof(KArgs, bvh) + offsetof(DevBvh, big_geom)));
    BvhBegin b;
    b.big_geom = (const f4 *)(uintptr_t)((uint64_t)r[0] | ((uint64_t)r[1] << 32));
    b.big_vel = (const f4 *)(uintptr_t)((uint64_t)r[2] | ((uint64_t)r[3] << 32));
    b.big_index = (const uint32_t *)(uintptr_t)((uint64_t)r[4] | ((uint64_t)r[5] << 32));
    b.n_big = r[6]; b.depth = r[7]; b.root = (int32_t)r[8];
    b.cx = __uint_as_float(r[9]); b.cy = __uint_as_float(r[10]); b.cz = __uint_as_float(r[11]); b.centre_radius = __uint_as_float(r[12]);
    b.r_max2 = __uint_as_float(r[13]); b.inv_2rmin = __uint_as_float(r[14]); b.abs_max = __uint_as_float(r[15]);
    return b;
}

// The stack pointer is an ABSOLUTE LDS byte address (the dynamic-LDS base is already folded in when a query begins), so an
// access is the ds instruction alone, no per-access address add.
template <class T> __device__ __forceinline__ T lds_get(uint32_t addr) { return *(const __attribute__((address_space(3))) T *)(uintptr_t)addr; }
template <class T> __device__ __forceinline__ void lds_put(uint32_t addr, T v) { *(__attribute__((address_space(3))) T *)(uintptr_t)addr = v; }
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }

template <bool MOVING, class S>
__device__ __forceinline__ void trav_begin(const KArgs &A, const Path &pt, Trav &tr, uint32_t sp0, bool a_plain_wave, bool &a_odd, Cen *cn = nullptr) {
    RTW_CEN(cn, CEN_TRAV_BEGIN);
    const BvhBegin bv = load_bvh_begin();
    const v3 o = pt.o, d = pt.d;
    tr.a = dot(d, d);
    tr.ra = rcp_refined(tr.a);
    a_odd = !in_range(tr.a, 0x1p-20f, 0x1p20f);           // (per lane; the caller folds it into the wave's sticky flag where the wave is whole)
    const bool a_plain = a_plain_wave && ballot64(a_odd) == 0ull;
    tr.best = -1; tr.best_t = A.maxt; tr.sp = sp0;           // the lane's level-0 slot (the sentinel)
    {   // big spheres: uniform loop, scalar loads
        cf4_ptr bg = (cf4_ptr)(uintptr_t)bv.big_geom;
        cf4_ptr bvel = (cf4_ptr)(uintptr_t)bv.big_vel;
        // (the sphere's index in the scene list through the constant address space too: as a generic pointer the compiler fetched it with a
        //  per-lane flat load of a wave-uniform address and waited for it -- a global-memory round trip in every SHADE step)
        typedef const uint32_t __attribute__((address_space(4))) *cu32_ptr;
        cu32_ptr bidx = (cu32_ptr)(uintptr_t)bv.big_index;
        for (uint32_t k = 0; k < bv.n_big; ++k) {
            f4 vv = MOVING ? bvel[k] : f4{ 0, 0, 0, 0 };
            exact_sphere<MOVING>(bg[k], vv, bidx[k], o, d, pt.tm, tr.a, tr.ra, a_plain, A.mint, A.maxt, tr.best, tr.best_t);
        }
    }
    if (bv.root == (int)0x80000000) { tr.node = (int)Code<S>::END; return; }   // no tree: the query is complete
    // (LDS variant: inner nodes by byte offset, trav_node_lds; every code zero-extended from 16 bits)
    tr.node = sizeof(S) == 2 ? (bv.root >= 0 ? bv.root * 32 : (int)((uint32_t)bv.root & 0xFFFFu)) : bv.root;
    // per-ray constants of the thick-ray slab test.  Everything here only feeds CONSERVATIVE bounds, so
    // the hardware approximations (v_sqrt_f32 / v_rcp_f32 / v_rsq_f32, <= 1 ulp) are used with the
    // 1e-4 relative safety factors below instead of the correctly-rounded sequences.
    const float ex = o.x - bv.cx, ey = o.y - bv.cy, ez = o.z - bv.cz;
#ifdef RTW_BEGIN_L2_NORM
    const float M = __builtin_amdgcn_sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + bv.centre_radius;
#else
    // an UPPER bound of |o - C| is all that is needed: the 1-norm (two additions with |.| modifiers instead of three products, two sums and a root).
    // It is at most sqrt(3) times the distance, which makes rho / tau -- 1e-6-sized paddings against spheres of radius 0.2 -- at most 3x / 1.7x as large
    const float M = (__builtin_fabsf(ex) + __builtin_fabsf(ey) + __builtin_fabsf(ez)) * 1.0001f + bv.centre_radius;
#endif
    const float q = (M * M + bv.r_max2) * RTW_KU;
    const float sq_q = __builtin_amdgcn_sqrtf(q) * 1.0001f;
    float rho = fminf(q * bv.inv_2rmin, sq_q);
    rho = rho * 1.0001f + 4.8e-7f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + bv.abs_max);   // + slab-arithmetic slop (8u * magnitudes)
    tr.tau_t = sq_q * 1.0002f * __builtin_amdgcn_rsqf(tr.a) + 1e-30f;
    float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
#ifdef RTW_BEGIN_GUARD_PER_LANE
    if (!(fabsf(d.x) >= 1e-20f)) ix = copysignf(1e20f, d.x);
    if (!(fabsf(d.y) >= 1e-20f)) iy = copysignf(1e20f, d.y);
    if (!(fabsf(d.z) >= 1e-20f)) iz = copysignf(1e20f, d.z);
#else
    // a component of the direction that is (nearly) zero or NaN gets a huge finite reciprocal of its sign: checked for the whole wave with one
    // three-way minimum, so that the three compare / select pairs are only executed by waves that hold such a ray
    if (ballot64(!(fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)) >= 1e-20f) || d.x != d.x || d.y != d.y || d.z != d.z) != 0ull) {
        if (!(fabsf(d.x) >= 1e-20f)) ix = copysignf(1e20f, d.x);
        if (!(fabsf(d.y) >= 1e-20f)) iy = copysignf(1e20f, d.y);
        if (!(fabsf(d.z) >= 1e-20f)) iz = copysignf(1e20f, d.z);
    }
#endif
    tr.ix = ix; tr.iy = iy; tr.iz = iz;
    if (sizeof(S) == 2) {
        // LDS nodes: a ray along +axis enters a box through lo and leaves through hi, one along -axis the other way round, so
        // {near, far} = {lo, hi} or {hi, lo} by the sign of 1/d alone -- chosen by one byte permute of the {lo, hi} dword instead of
        // a v_min / v_max pair per plane pair -- and with the box inflated by rho:  t_near = (near - o) / d - rho / |d|,
        // t_far = (far - o) / d + rho / |d|,  i.e. fma(plane, 1/d, -o/d -+ rho/|d|): no sign-dependent select in the constants either.
        const float ox = -o.x * ix, oy = -o.y * iy, oz = -o.z * iz;
        const float rx = rho * fabsf(ix), ry = rho * fabsf(iy), rz = rho * fabsf(iz);
        tr.kpx = ox - rx; tr.kpy = oy - ry; tr.kpz = oz - rz;
        tr.kmx = ox + rx; tr.kmy = oy + ry; tr.kmz = oz + rz;
        const uint32_t ID = 0x03020100u, FLIP = 0x03020100u ^ 0x01000302u;      // selector XOR mask: all ones where 1/d < 0
        tr.selx = ID ^ ((uint32_t)((int32_t)__float_as_uint(ix) >> 31) & FLIP);
        tr.sely = ID ^ ((uint32_t)((int32_t)__float_as_uint(iy) >> 31) & FLIP);
        tr.selz = ID ^ ((uint32_t)((int32_t)__float_as_uint(iz) >> 31) & FLIP);
    } else {
        // t(lo) = (lo - rho - o) * inv = fma(lo, inv, -(o + rho) * inv);  t(hi) = fma(hi, inv, -(o - rho) * inv)
        tr.kpx = -(o.x + rho) * ix; tr.kpy = -(o.y + rho) * iy; tr.kpz = -(o.z + rho) * iz;
        tr.kmx = -(o.x - rho) * ix; tr.kmy = -(o.y - rho) * iy; tr.kmz = -(o.z - rho) * iz;
    }
    tr.lo_lim = A.mint - tr.tau_t;
    tr.hi_lim = tr.best_t + tr.tau_t;
}


// After a leaf test: take the next entry off the stack (the sentinel of level 0 ends the query).
template <class S>
__device__ __forceinline__ void trav_pop(Trav &tr) {
    tr.node = (int)(uint32_t)lds_get<typename std::make_unsigned<S>::type>(tr.sp);
    tr.sp -= RTW_BLOCK * (uint32_t)sizeof(S);      // (may step below level 0 when the sentinel came off: sp is not used again before trav_begin)
}

template <class S> __device__ __forceinline__ int entry_to_node(uint32_t raw);
template <> __device__ __forceinline__ int entry_to_node<short>(uint32_t raw) { return (int)(raw & 0xFFFFu); }   // (v_and: fast class; a sign extension is a v_bfe)
template <> __device__ __forceinline__ int entry_to_node<int>(uint32_t raw) { return (int)raw; }

// Both slab results are in: descend into the nearer child and push the farther, or pop.  Select form, no exec-mask regions:
// the farther child is stored ABOVE the top unconditionally (it only counts when sp moves up), and the entry a pop would
// return was read by the caller before the box tests (`popped`), so its LDS latency hides behind them.
// c0 / c1 / popped are raw stack entries (for 16-bit entries: the id in the low half, upper bits ignored).
template <class S>
__device__ __forceinline__ void trav_descend(Trav &tr, float e0, float x0, float e1, float x1,
                                             uint32_t c0, uint32_t c1, uint32_t popped) {
    // hit <=> [max(entry, lo_lim), min(exit, hi_lim)] is non-empty (lo_lim <= hi_lim always: best_t >= mint): two min/max and
    // one compare per box instead of three compares and two mask ANDs
    const float n0 = fmaxf(e0, tr.lo_lim), n1 = fmaxf(e1, tr.lo_lim);
    const bool h0 = n0 <= fminf(x0, tr.hi_lim);
    const bool h1 = n1 <= fminf(x1, tr.hi_lim);
    // (plain i1 logic on already-evaluated compares: lane-mask arithmetic on the scalar unit, no VALU and no branches)
    const bool le = n0 <= n1;
    const bool both = h0 && h1, none = !(h0 || h1);
    const bool near0 = h0 && (!h1 || le);
    const uint32_t level = RTW_BLOCK * (uint32_t)sizeof(S);
    if (sizeof(S) == 2) {
        // c0 holds {c0, c1} as halves: rotate the nearer child into the low half (one select of the rotate amount + v_alignbit), the
        // farther one is then stored straight from the high half (ds_write_b16_d16_hi) -- 2 VALU for near AND far instead of 3.
        // The condition is built as ONE lane mask from the three compares' ballots (scalar unit): left to the selector,
        // select(a && b, ..) becomes two nested v_cndmask.
        const uint64_t H0 = __builtin_amdgcn_ballot_w64(h0), H1 = __builtin_amdgcn_ballot_w64(h1), LE = __builtin_amdgcn_ballot_w64(le);
        const bool far0 = __builtin_amdgcn_inverse_ballot_w64(~H0 | (H1 & ~LE));
        const uint32_t nf = __builtin_amdgcn_alignbit(c0, c0, far0 ? 16u : 0u);
        lds_put<unsigned short>(tr.sp + level, (unsigned short)(nf >> 16));
        tr.node = entry_to_node<S>(none ? popped : nf);
    } else {
        lds_put<S>(tr.sp + level, (S)(near0 ? c1 : c0));
        tr.node = entry_to_node<S>(none ? popped : (near0 ? c0 : c1));
    }
    tr.sp = (tr.sp + (both ? level : 0u)) - (none ? level : 0u);
}

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32a2 __attribute__((aligned(2)));
// f16 halves of a dword as f32 (scalar casts only: element access through f16 ext-vectors is miscompiled
// by this toolchain -- lanes came back undefined)
__device__ __forceinline__ float h_lo(unsigned int w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xFFFFu)); }
__device__ __forceinline__ float h_hi(unsigned int w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }

// One inner-node visit, nodes resident in LDS as f16 (BvhNode16): two ds_read_b128 instead of four
// global loads; the f16 planes feed v_fma_mix_f32 directly.
__device__ __forceinline__ void trav_node_lds(const u4 *lnodes, Trav &tr) {
    const uint32_t popped = lds_get<unsigned short>(tr.sp);
    // tr.node is the node's byte offset in the f16 array (BvhNode16.c0 / c1 hold inner children that way), which sits at LDS offset 0
    const uint32_t at = (uint32_t)tr.node;                            // (+ the array's LDS address, which is 0: checked once by the kernel)
    const u4 r0 = lds_get<u4>(at);
    const u3 r1 = lds_get<u3>(at + 16u);       // 12 of the 16 bytes: no dead destination register for the allocator to recycle early
#ifdef RTW_LDS_PROBE
    {   // experiment: how much LDS headroom is there?  RTW_LDS_PROBE extra (2-byte-aligned) dword reads per visit, results only consumed
        const uint32_t pa = (uint32_t)tr.node * 32u + ((tr.selx & 1u) << RTW_LDS_PROBE_SH);
        #pragma unroll
        for (int k = 0; k < RTW_LDS_PROBE; k++) { const uint32_t x = lds_get<u32a2>(pa + 4u * k); asm volatile("" :: "v"(x)); }
    }
#endif
    // r0 = box0 {lo hi}.x  box0 {lo hi}.y  box0 {lo hi}.z  box1 {lo hi}.x      r1 = box1 {lo hi}.y  box1 {lo hi}.z  {c0 c1}
    const uint32_t c0 = r1.z, c1 = 0;             // both ids in one dword (trav_descend<short> rotates it)
    // {near plane, far plane} of each (box, axis): the halves as stored or swapped, by the ray's direction along the axis
    const uint32_t ax = __builtin_amdgcn_perm(r0.x, r0.x, tr.selx), ay = __builtin_amdgcn_perm(r0.y, r0.y, tr.sely), az = __builtin_amdgcn_perm(r0.z, r0.z, tr.selz);
    const uint32_t bx = __builtin_amdgcn_perm(r0.w, r0.w, tr.selx), by = __builtin_amdgcn_perm(r1.x, r1.x, tr.sely), bz = __builtin_amdgcn_perm(r1.y, r1.y, tr.selz);
    const float e0 = fmaxf(fmaxf(__builtin_fmaf(h_lo(ax), tr.ix, tr.kpx), __builtin_fmaf(h_lo(ay), tr.iy, tr.kpy)), __builtin_fmaf(h_lo(az), tr.iz, tr.kpz));
    const float x0 = fminf(fminf(__builtin_fmaf(h_hi(ax), tr.ix, tr.kmx), __builtin_fmaf(h_hi(ay), tr.iy, tr.kmy)), __builtin_fmaf(h_hi(az), tr.iz, tr.kmz));
    const float e1 = fmaxf(fmaxf(__builtin_fmaf(h_lo(bx), tr.ix, tr.kpx), __builtin_fmaf(h_lo(by), tr.iy, tr.kpy)), __builtin_fmaf(h_lo(bz), tr.iz, tr.kpz));
    const float x1 = fminf(fminf(__builtin_fmaf(h_hi(bx), tr.ix, tr.kmx), __builtin_fmaf(h_hi(by), tr.iy, tr.kmy)), __builtin_fmaf(h_hi(bz), tr.iz, tr.kmz));
    trav_descend<short>(tr, e0, x0, e1, x1, c0, c1, popped);
}

// One inner-node visit: two slab tests, descend into the nearer child, push the farther.
__device__ __forceinline__ void trav_node(const DevBvh &bv, Trav &tr) {
    const uint32_t popped = lds_get<uint32_t>(tr.sp);
    const f4 *np = (const f4 *)(bv.nodes + tr.node);
    const f4 n0 = np[0], n1 = np[1], n2 = np[2];
    const uint32_t c0 = (uint32_t)bv.nodes[tr.node].c0, c1 = (uint32_t)bv.nodes[tr.node].c1;
    // child 0 box: lo0 = n0.xyz, hi0 = (n0.w, n1.x, n1.y); child 1: lo1 = (n1.z, n1.w, n2.x), hi1 = n2.yzw
    float t1, t2;
    t1 = __builtin_fmaf(n0.x, tr.ix, tr.kpx); t2 = __builtin_fmaf(n0.w, tr.ix, tr.kmx);
    float e0 = fminf(t1, t2), x0 = fmaxf(t1, t2);
    t1 = __builtin_fmaf(n0.y, tr.iy, tr.kpy); t2 = __builtin_fmaf(n1.x, tr.iy, tr.kmy);
    e0 = fmaxf(e0, fminf(t1, t2)); x0 = fminf(x0, fmaxf(t1, t2));
    t1 = __builtin_fmaf(n0.z, tr.iz, tr.kpz); t2 = __builtin_fmaf(n1.y, tr.iz, tr.kmz);
    e0 = fmaxf(e0, fminf(t1, t2)); x0 = fminf(x0, fmaxf(t1, t2));
    t1 = __builtin_fmaf(n1.z, tr.ix, tr.kpx); t2 = __builtin_fmaf(n2.y, tr.ix, tr.kmx);
    float e1 = fminf(t1, t2), x1 = fmaxf(t1, t2);
    t1 = __builtin_fmaf(n1.w, tr.iy, tr.kpy); t2 = __builtin_fmaf(n2.z, tr.iy, tr.kmy);
    e1 = fmaxf(e1, fminf(t1, t2)); x1 = fminf(x1, fmaxf(t1, t2));
    t1 = __builtin_fmaf(n2.x, tr.iz, tr.kpz); t2 = __builtin_fmaf(n2.w, tr.iz, tr.kmz);
    e1 = fmaxf(e1, fminf(t1, t2)); x1 = fminf(x1, fmaxf(t1, t2));
    trav_descend<int>(tr, e0, x0, e1, x1, c0, c1, popped);
}

#ifdef RTW_STAMP
// diagnostic build: wave-ticks (s_memtime) of the sub-steps of SHADE, summed into sub[k]
#define RTW_SUB_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t_now = __builtin_amdgcn_s_memtime(); sub[k] += t_now - t_sub; t_sub = t_now; } while (0)
#else
#define RTW_SUB_STAMP(k) do { } while (0)
#endif
#ifndef RTW_BVH_WAVES
#define RTW_BVH_WAVES 4        /* min waves per SIMD the register allocator must leave room for */
#endif
// GEOM builds: a ray that is not an ordinary one.  The constant-density medium draws its free path as ln(xi) / -density (instance.rs:24-26),
// and xi == 0 -- one draw in 2^24, dozens per frame of presentation_image -- makes that +inf: the scatter point, and with it the origin of
// the next ray, is inf or NaN.  The reference's sphere test ACCEPTS such a ray (every comparison with its NaN root is false,
// sphere.rs:118-121) and reports the first sphere of the list, where a tree prunes the ray at its root (the tree's bounds presume ordinary
// operands, DESIGN.md "Conservative traversal"; rtw_shim.hip keeps the camera inside them).  Such lanes walk the list as the reference does;
// the query is then complete.  (Found as ONE pixel of presentation_image at 64 spp that the tree rendered finite and the list walk NaN:
// tests/test_gpu_round3.py::test_a_ray_from_infinity_hits_what_the_reference_says.)
template <bool MOVING, class S>
__device__ __forceinline__ void wild_ray_query(const KArgs &A, const Path &pt, Trav &tr, uint32_t &n_tests) {
    const float m = (__builtin_fabsf(pt.o.x) + __builtin_fabsf(pt.o.y) + __builtin_fabsf(pt.o.z)) +
                    (__builtin_fabsf(pt.d.x) + __builtin_fabsf(pt.d.y) + __builtin_fabsf(pt.d.z));
    if (m < 0x1p60f) return;
    closest_brute<MOVING>(A.sc, pt.o, pt.d, pt.tm, A.mint, A.maxt, tr.best, tr.best_t);
    tr.node = (int)Code<S>::END;
    n_tests += A.sc.n - A.bvh.n_big;                           // (the big spheres are counted with every query, flush at the kernel's end)
}

#ifndef RTW_BVH_WAVES_GEOM
#define RTW_BVH_WAVES_GEOM 2   /* the generic GEOM builds (114 VGPRs, 4 waves as compiled); 3 / 4 / 5 / 6 measured, profiles/r03_ab_geom_waves.log */
#endif
#ifndef RTW_BVH_WAVES_GEOM_SPEC
#define RTW_BVH_WAVES_GEOM_SPEC 5   /* the GEOM builds of the common configuration (SPEC == 2; 97 - 104 VGPRs uncapped): profiles/r03_ab_geom_spec.log */
#endif
#ifndef RTW_BVH_WAVES_SPEC
#define RTW_BVH_WAVES_SPEC 7   /* the specialised builds (SPEC != 0) are compiled for 7 waves/SIMD = 72 VGPRs (two dwords of scratch in the static builds, ten in the MOVING
                                  ones); whether the seventh workgroup per CU is used depends on the LDS the tree needs and on the size of the launch (rtw_shim.hip) */
#endif
#ifndef RTW_BVH_WAVES_SPEC2
#define RTW_BVH_WAVES_SPEC2 RTW_BVH_WAVES_SPEC   /* the build that keeps the image-texture lookup (C5) */
#endif
// NODES: 0 = f32 nodes in global memory, 32-bit stack; 1 = f16 nodes in LDS, 16-bit stack; 2 = as 1, and the spheres' {centre, r^2} in LDS
// too (a build of its own: as a run-time choice the leaf test went through a flat load and a select of two addresses, 7 VALU).
// Which builds run the SHADE step in two halves around one rejection loop (4.2): the common configuration without quads / instances.
#ifndef RTW_BVH_WAVES_FOLDED
#define RTW_BVH_WAVES_FOLDED RTW_BVH_WAVES   /* builds with the generic step and folded switches (SPEC == 4 without GEOM) */
#endif
// (Not the MOVING build that keeps the texture lookup -- C5's: there the generic-shaped step with the switches folded in is level or ahead, 92.9 / 93.1 against
//  93.6 / 93.9 ms at 7 waves, profiles/r03_ab_two_halves_c5*.log -- the two-halves step's spills at 72 VGPRs eat what it saves.)
constexpr bool two_halves_step(int spec, bool geom, bool moving) {
#ifdef RTW_NO_TWO_HALVES
    return false;
#else
    return gradient_spec(spec) && !geom && !(moving && spec == 2);
#endif
}
template <bool MOVING, int NODES, int SPEC, bool GEOM>
__global__ __launch_bounds__(RTW_BLOCK, GEOM ? (SPEC != 0 ? RTW_BVH_WAVES_GEOM_SPEC : RTW_BVH_WAVES_GEOM) : (two_halves_step(SPEC, GEOM, MOVING) ? (SPEC == 2 ? RTW_BVH_WAVES_SPEC2 : RTW_BVH_WAVES_SPEC) : (gradient_spec(SPEC) ? RTW_BVH_WAVES_SPEC2 : (SPEC != 0 ? RTW_BVH_WAVES_FOLDED : RTW_BVH_WAVES)))) void render_bvh(const KArgs A) {
    constexpr bool LDSN = NODES != 0, geom_in_lds = NODES == 2;
    // LDS is all dynamic, sized by the host for THIS tree (rtw_shim.hip, render_enqueue_impl): -- LDS-node variants -- the f16 nodes at
    // offset 0, then the per-lane traversal stack [level][thread] (a level is one conflict-free row; depth + 3 levels: the sentinel,
    // one per tree level, and the slot above the top that the select-form descend always writes; 16-bit entries in the LDS-node
    // variants), then -- NODES == 2 -- {centre, r^2} of every sphere for the leaf tests.  Book-1: 15.5 KB + 7 KB (+ 7.8 KB).
    typedef typename std::conditional<LDSN, short, int>::type stack_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    *(stack_t *)(lds_raw + A.lds_stack_off + threadIdx.x * (uint32_t)sizeof(stack_t)) = (stack_t)Code<stack_t>::END;   // level 0: the sentinel (own slot, no sync needed)
    u4 *lnodes = (u4 *)lds_raw;                     // offset 0 (rtw_shim.hip: the node fetch needs no base register)
    // ... and the dynamic LDS itself starts at LDS address 0 (the kernel has no static __shared__), so a node's byte offset IS its address
    // (trav_node_lds).  The address is a link-time constant the compiler cannot see; should it ever not be 0 the launch fails loudly.
    if (LDSN && lds_addr(lds_raw) != 0u) { if (threadIdx.x == 0) atomicAdd(&A.stats[23], 1ull); return; }
    f4 *lgeom = (f4 *)(lds_raw + A.lds_geom_off);
    const DevScene &sc = A.sc;
    if (LDSN) {
        const u4 *src = (const u4 *)A.bvh.nodes16;
        for (uint32_t i = threadIdx.x; i < A.bvh.n_nodes * 2u; i += RTW_BLOCK) lnodes[i] = src[i];
        if (geom_in_lds) for (uint32_t i = threadIdx.x; i < sc.n; i += RTW_BLOCK) lgeom[i] = sc.geom[i];
        __syncthreads();
    }

    // Lane flags live in ONE VGPR: as separate bools the compiler keeps them as lane masks in SGPR pairs and re-merges
    // every one of them under exec on every trip of the loop (3 SALU each), although only SHADE steps change them.
    enum : uint32_t { F_HAVE = 1u,       // owns a work unit
                      F_INFLIGHT = 2u,   // a closest-hit query is in flight / complete and waiting to be shaded
                      F_NEWPATH = 4u,    // the next SHADE step starts the next sample of the unit
                      F_DONE = 8u,       // a finished path waits for the next SHADE step to bank it
                      F_K_SHIFT = 8u };  // specialised builds: the path's bounce count (Path.k) in bits 8..31
    uint32_t fl = 0u;
    Pixel px; px.ij = px.rng_base = px.s = px.slot = 0;
    Reserve rs; rs.next = rs.end = rs.limit = rs.i0 = rs.k0 = rs.s0 = rs.s1 = rs.slot0 = rs.tries = 0; rs.sub = blockIdx.x & ((1u << A.sub_shift) - 1u);
#ifdef RTW_ENDTIMES
    rs.t_dry = 0ull;
#endif
    Path pt; pt.o = pt.d = pt.L = mk(0, 0, 0); pt.thr = mk(1, 1, 1); pt.tm = 0; pt.k = 0; pt.poison = false; pt.rng.state = 0; pt.rng.inc = 1;
    Trav tr; tr.node = (int)Code<stack_t>::END; tr.sp = 0; tr.best = -1; tr.best_t = 0; tr.a = tr.ra = 1; tr.ix = tr.iy = tr.iz = 0;
    tr.kpx = tr.kpy = tr.kpz = tr.kmx = tr.kmy = tr.kmz = 0; tr.selx = tr.sely = tr.selz = 0; tr.tau_t = tr.lo_lim = tr.hi_lim = 0;
    // Work counters live in SGPRs: they are sums of ballot popcounts the scheduler computes anyway (node visits ==
    // lanes live in TRAVERSE steps, leaf tests == lanes live in LEAF steps), which keeps four VGPRs out of the loop.
    // (32-bit: one wave's share of a launch -- at most 2^32 sample slots per launch, rtw_ctx_render -- stays far below 2^32)
    uint32_t w_seg = 0, w_rays = 0;
    uint32_t t_lo = RTW_T_LO;                       // wave-uniform: drops to RTW_T_LO_DRAIN once the queue is empty
    uint32_t trips = 0; bool aborted = false;       // wave-uniform
    bool a_plain = true;                            // wave-uniform, sticky (trav_begin)
    uint32_t n_isph = 0, n_quad = 0;                // GEOM builds only: member-sphere and quad tests of the extra stage
    uint32_t c_steps[3] = { 0, 0, 0 };              // wave-uniform (SGPR) census of the scheduler
    uint32_t c_lanes[3] = { 0, 0, 0 };
#ifdef RTW_STAMP
    unsigned long long c_time[3] = { 0, 0, 0 };
    unsigned long long sub[5] = { 0, 0, 0, 0, 0 }, t_sub = 0;     // SHADE: hit / bank + next unit / camera ray / query begin / rest
#endif
#ifdef RTW_CENSUS
    Cen cen_store; for (int k = 0; k < CEN_N; k++) cen_store.n[k] = cen_store.w[k] = 0u;
    Cen *const cn = &cen_store;
#else
    Cen *const cn = nullptr;
#endif

#ifdef RTW_ENDTIMES
    const unsigned long long t_wave_start = __builtin_amdgcn_s_memtime();
    const unsigned long long rt_wave_start = wall_clock64();     // s_memrealtime: one 100 MHz clock for the whole device (s_memtime is per XCD)
    uint32_t dry_trips = 0, dry_steps[3] = { 0, 0, 0 }, dry_lanes[3] = { 0, 0, 0 };   // what the wave does after it has found the queue empty
    uint32_t dry_queries = 0;                                                          // per lane: queries shaded after that
#endif
    for (;;) {
        // ---- scheduler ---------------------------------------------------------------------------
        // SHADE is the expensive step (several hundred instructions): it runs when enough lanes have piled up
        // in it (RTW_S_HI) or when little traversal work is left to hide behind (RTW_T_LO); otherwise the
        // larger of the two traversal queues runs.
        uint32_t nT = lanes_in(in_trav<stack_t>(tr.node));
        const uint32_t nL = lanes_in(in_leaf<stack_t>(tr.node));
        const uint32_t nS = lanes_in(in_shade<stack_t>(tr.node));
        if ((nT | nL | nS) == 0u) break;                     // every lane is DEAD
        // Safety valve of the persistent loop: a wave that has taken an absurd number of scheduler trips (the bench frame needs ~130 k per
        // wave) gives up instead of hanging the GPU; the launch then reports RTW_E_INTERNAL (stats[23] counts such waves).
        if (++trips > RTW_MAX_TRIPS) { aborted = true; break; }
        RTW_CEN(cn, CEN_BIG_ROOT);                 // (census build: trips through the scheduler)
        const bool run_shade = nS >= RTW_S_HI || (nT < t_lo && nL < t_lo && nS > 0u);
        const bool run_leaf = nL > nT;
#ifdef RTW_ENDTIMES
        if (rs.t_dry) {
            dry_trips++;
            const int ph = run_shade ? 2 : (run_leaf ? 1 : 0);
            dry_steps[ph]++; dry_lanes[ph] += ph == 2 ? nS : (ph == 1 ? nL : nT);
            if (run_shade && in_shade<stack_t>(tr.node) && (fl & F_INFLIGHT)) dry_queries++;
        }
#endif
#ifdef RTW_STAMP
        const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
        const int which = run_shade ? 2 : (run_leaf ? 1 : 0);
#endif
        bool burst = false;
        if (run_shade) {
            c_steps[2]++; c_lanes[2] += nS;
            // a. the closest-hit query this lane was waiting on is complete: scatter, or end the path
            bool need_unit = false, started = false, a_odd = false;
            const bool shading = in_shade<stack_t>(tr.node);
            w_seg += (uint32_t)__popcll(ballot64(shading && (fl & F_INFLIGHT) != 0u));
#ifdef RTW_STAMP
            t_sub = t_begin;
#endif
            if constexpr (two_halves_step(SPEC, GEOM, MOVING)) {
            // ---- specialised builds: the step in two halves around one rejection loop for the scatter directions AND the lens samples ----
            bool need_ball = false, need_disk = false, front = false;
            // a. the closest-hit query this lane was waiting on is complete.  Paths that END here -- the ray missed (sky), or its depth is
            //    exhausted (black) -- are finished first, so that their lanes' next camera rays and the hits' scatter directions meet in ONE
            //    rejection loop further down; the lanes that hit keep unit(direction) until then.
            v3 nrm, ud; float metal;
            bool hit = false;
            if (shading) {
                RTW_CEN(cn, CEN_SHADING);
                if (fl & F_INFLIGHT) {
                    fl &= ~F_INFLIGHT;
                    RTW_CEN(cn, CEN_INFLIGHT);
                    ud = unit(pt.d);
                    // (the radiance of a path that ends -- sky, or black at exhausted depth -- is formed in the bank block below, where it is
                    //  stored: formed here it lived across the join, three registers that the MOVING builds spilled and reloaded in every step)
                    if (tr.best < 0) { RTW_CEN(cn, CEN_MISS); fl |= F_DONE; }
                    else if ((fl >> F_K_SHIFT) + 1u >= A.depth) { RTW_CEN(cn, CEN_DEPTH_END); fl |= F_DONE; }   // depth exhausted: the innermost call returns black
                    else { hit = true; fl += 1u << F_K_SHIFT; }         // (the bounce count of these builds: Path.k is not carried)
                }
            }
            RTW_SUB_STAMP(0);
            if (shading) {
                if (fl & F_DONE) {
                    fl &= ~F_DONE;
                    // in these builds F_DONE is only ever set a few lines up, in this same step: tr.best still tells a miss from an exhausted depth
                    if (tr.best < 0) shade_miss<SPEC>(A, pt, ud); else pt.L = mk(0, 0, 0);
                    if (finish_path<SPEC>(A, px, pt, cn)) fl &= ~F_HAVE; else fl |= F_NEWPATH;
                }
                need_unit = (fl & F_HAVE) == 0u;
                if (need_unit) RTW_CEN(cn, CEN_NEED_UNIT);
            }
            // b. next work unit (every lane of the wave: the reserve stays wave-uniform; see the generic path below)
            bool exhausted = false;
            const bool got = fetch_pixel(A, need_unit, px, exhausted, rs);
            if (ballot64(exhausted) != 0ull) t_lo = RTW_T_LO_DRAIN;
            RTW_SUB_STAMP(1);
            if (shading) {
                if (got) fl |= F_HAVE | F_NEWPATH;
                if (exhausted) tr.node = (int)Code<stack_t>::DEAD;
                // c. next camera ray, up to its lens sample
                if ((fl & F_HAVE) && (fl & F_NEWPATH)) { fl &= F_HAVE | F_INFLIGHT | F_DONE; start_path_first(px, pt, cn); need_disk = true; started = true; }   // (clears F_NEWPATH and the bounce count)
            }
            //    ... the hit, up to its random unit vector
            if (hit) shade_hit_first<MOVING, SPEC>(A, pt, ud, tr.best, tr.best_t, need_ball, nrm, metal, front, cn);
            // d. ONE rejection loop: points of the unit ball for the lanes that scatter, of the unit disk for the lanes that start a path
            float sx, sy, sz, sl2;
            sample_ball_or_disk(pt.rng, need_ball || need_disk, need_ball, sx, sy, sz, sl2, cn);
            if (need_ball) pt.d = on_hit_second(metal, nrm, pt.d, front, mk(sx, sy, sz), sl2);     // (sl2: see RTW_UNIT_FROM_L2, rtw_device.h)
            if (need_disk) start_path_second(px, pt, sx, sy);
            RTW_SUB_STAMP(2);
            // e. start the next closest-hit query
            if (shading && (fl & F_HAVE)) {
                trav_begin<MOVING, stack_t>(A, pt, tr, lds_addr(lds_raw) + A.lds_stack_off + threadIdx.x * (uint32_t)sizeof(stack_t), a_plain, a_odd, cn);
                fl |= F_INFLIGHT;
            }
            RTW_SUB_STAMP(3);
            } else {
            // ---- generic build (every integrator / sampler / dialect, quads and instances) ----
            if (shading) {
                RTW_CEN(cn, CEN_SHADING);
                if (fl & F_INFLIGHT) {
                    fl &= ~F_INFLIGHT;
                    const bool done = GEOM ? shade_geom<MOVING, SPEC>(A, pt, tr.best, tr.best_t, n_isph, n_quad) : shade<MOVING, SPEC>(A, pt, tr.best, tr.best_t, cn);
                    if (done) fl |= F_DONE;
                }
            }
            RTW_SUB_STAMP(0);
            if (shading) {
                if (fl & F_DONE) { fl &= ~F_DONE; if (finish_path<SPEC>(A, px, pt, cn)) fl &= ~F_HAVE; else fl |= F_NEWPATH; }
                need_unit = (fl & F_HAVE) == 0u;
                if (need_unit) RTW_CEN(cn, CEN_NEED_UNIT);
            }
            // b. next work unit.  Executed by EVERY lane of the wave (not only the ones in SHADE): the wave's
            //    reserve must stay wave-uniform, which it only does if all lanes run its bookkeeping.
            bool exhausted = false;
            const bool got = fetch_pixel(A, need_unit, px, exhausted, rs);
            // Once the queue is empty the wave only drains: its lanes die one by one, S_HI can no longer be reached and SHADE runs on the
            // "few lanes traverse" rule alone -- after finding the queue empty a wave runs on for 0.38 ms on average and 1.2 ms at most
            // (profiles/r02_endtimes.log).  The threshold of that rule can differ while draining (RTW_T_LO_DRAIN; measured, not better).
            // (HERE, where every lane of the wave is active: t_lo steers the scheduler and must stay wave-uniform.)
            if (ballot64(exhausted) != 0ull) t_lo = RTW_T_LO_DRAIN;
            RTW_SUB_STAMP(1);
#ifdef RTW_STAMP
            if (shading) {
                if (got) fl |= F_HAVE | F_NEWPATH;
                if (exhausted) tr.node = (int)Code<stack_t>::DEAD;
                if ((fl & F_HAVE) && (fl & F_NEWPATH)) { fl &= ~F_NEWPATH; start_path<SPEC>(A, px, pt); started = true; }
            }
            RTW_SUB_STAMP(2);
#endif
            if (shading) {
                if (got) fl |= F_HAVE | F_NEWPATH;
                if (exhausted) tr.node = (int)Code<stack_t>::DEAD;
                if (fl & F_HAVE) {
                    // c. next camera ray (a lane whose path continues keeps its scattered ray)
                    if (fl & F_NEWPATH) { fl &= ~F_NEWPATH; start_path<SPEC>(A, px, pt, cn); started = true; }
                    // d. start the next closest-hit query
                    if (!SPEC && A.depth == 0 && A.integrator != RTW_INTEGRATOR_NORMAL) {   // `if depth < 1 { return black }` (ray_color.rs:14-16)
                        pt.L = A.integrator == RTW_INTEGRATOR_RUST2 ? ld3(A.bg) : mk(0, 0, 0);
                        fl |= F_DONE;                                      // banked on the next SHADE trip
                    } else {
                        trav_begin<MOVING, stack_t>(A, pt, tr, lds_addr(lds_raw) + A.lds_stack_off + threadIdx.x * (uint32_t)sizeof(stack_t), a_plain, a_odd, cn);
                        if constexpr (GEOM) wild_ray_query<MOVING, stack_t>(A, pt, tr, n_isph);
                        fl |= F_INFLIGHT;
                    }
                }
            }
            RTW_SUB_STAMP(3);
            }
            w_rays += (uint32_t)__popcll(ballot64(started));
            // Wave-uniform and STICKY: lanes of this wave still test leaves of queries begun in earlier SHADE steps, so once any lane's d.d
            // has left [2^-20, 2^20] the wave stays on the generic sqrt / division (same bits, a few more instructions) for good.
            if (ballot64(a_odd) != 0ull) a_plain = false;
#ifndef RTW_STAMP
            nT = lanes_in(in_trav<stack_t>(tr.node)); burst = nT >= 33u;      // (as after a LEAF step)
#endif
        } else {
            burst = !run_leaf;
            if (run_leaf) {
                c_steps[1]++; c_lanes[1] += nL;
                if (in_leaf<stack_t>(tr.node)) {
                    const uint32_t s = leaf_sphere<stack_t>(tr.node);
                    const f4 gs = geom_in_lds ? lgeom[s] : sc.geom[s];
                    exact_sphere<MOVING>(gs, MOVING ? sc.vel[s] : f4{ 0, 0, 0, 0 }, s, pt.o, pt.d, pt.tm, tr.a, tr.ra, a_plain, A.mint, A.maxt, tr.best, tr.best_t);
                    tr.hi_lim = tr.best_t + tr.tau_t;
                    trav_pop<stack_t>(tr);
                }
#ifndef RTW_STAMP
                // straight on to the bursts, without asking the scheduler, when more than half of the lanes are in TRAVERSE after the pops
                // (the same shortcut, and the same argument, as between bursts)
                nT = lanes_in(in_trav<stack_t>(tr.node)); burst = nT >= 33u;
#endif
            }
        }
        if (burst) {
            // RTW_TRAV_UNROLL node visits per scheduling decision: the scheduler's ballots and branches are
            // paid once per burst; lanes that leave TRAVERSE (leaf reached / query done) sit out the rest of it.
            // Bursts follow one another without a trip through the scheduler while MORE THAN HALF of the lanes are still in TRAVERSE: with
            // 33 or more lanes there, LEAF and SHADE hold at most 31 between them, so neither "nL > nT" nor "nS >= RTW_S_HI" (52) can be true
            // and the scheduler would say TRAVERSE again -- the same decisions from one ballot instead of three, and without the register
            // copies the compiler puts at the joins of the three-way branch for state that only LEAF and SHADE change (~16 v_mov per burst).
            static_assert(RTW_S_HI > 31u && RTW_T_LO <= 33u, "the shortcut below assumes the thresholds of the scheduler");
            uint32_t live = nT;
            for (;;) {
                // (the burst's census is summed in two scalars of its own and added once: kept in c_steps / c_lanes directly, the register
                // allocator -- out of SGPRs in this kernel -- holds the totals in VGPRs and every visit paid a v_add for each)
                uint32_t b_steps = 0, b_lanes = 0;
                for (int u = 0; u < RTW_TRAV_UNROLL; u++) {
                    b_steps++; b_lanes += live;
                    if (in_trav<stack_t>(tr.node)) { if (LDSN) trav_node_lds((const u4 *)lnodes, tr); else trav_node(A.bvh, tr); }
                    if (u + 1 >= RTW_TRAV_UNROLL) break;
                    live = lanes_in(in_trav<stack_t>(tr.node));
                    if (live == 0u) break;
                }
                asm volatile("" : "+s"(b_steps), "+s"(b_lanes));
                c_steps[0] += b_steps; c_lanes[0] += b_lanes;
#ifdef RTW_STAMP
                break;                                       // (diagnostic build: every burst is timed as a trip of the outer loop)
#else
                if (live == 0u) break;
                live = lanes_in(in_trav<stack_t>(tr.node));
                if (live < 33u) break;
                if (++trips > RTW_MAX_TRIPS) break;          // (the outer loop's valve fires on its next trip)
#endif
            }
        }
#ifdef RTW_STAMP
        // diagnostic build only: wave-ticks per phase (s_memtime), written to stats[11..13]
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        c_time[which] += __builtin_amdgcn_s_memtime() - t_begin;
#endif
    }
    if (GEOM) { flush_quads(A, n_quad); flush_tests(A, n_isph); }
#ifdef RTW_CENSUS
    for (int k = 0; k < CEN_N; k++) {          // diagnostic build: stats[32 + 2k] = wave-level executions of sub-block k, [33 + 2k] = lanes live in them
        unsigned long long w = cn->w[k], n = cn->n[k];
        for (int off = 32; off > 0; off >>= 1) { w += __shfl_down(w, off); n += __shfl_down(n, off); }
        if ((threadIdx.x & 63u) == 0) { atomicAdd(&A.stats[32 + 2 * k], w); atomicAdd(&A.stats[33 + 2 * k], n); }
    }
#endif
#ifdef RTW_ENDTIMES
    uint32_t dry_q_max = dry_queries, dry_q_sum = dry_queries;
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = __shfl_down(dry_q_max, off); dry_q_max = o > dry_q_max ? o : dry_q_max; dry_q_sum += __shfl_down(dry_q_sum, off); }
#endif
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&A.stats[0], (unsigned long long)w_rays);
        atomicAdd(&A.stats[1], (unsigned long long)w_seg);
        atomicAdd(&A.stats[2], (unsigned long long)c_lanes[1] + (unsigned long long)w_seg * A.bvh.n_big);   // leaf tests + the big-sphere pre-pass of every query
        atomicAdd(&A.stats[3], (unsigned long long)c_lanes[0]);
        for (int k = 0; k < 3; k++) { atomicAdd(&A.stats[5 + k], (unsigned long long)c_steps[k]); atomicAdd(&A.stats[8 + k], (unsigned long long)c_lanes[k]); }
        if (aborted) atomicAdd(&A.stats[23], 1ull);
#ifdef RTW_STAMP
        for (int k = 0; k < 3; k++) atomicAdd(&A.stats[11 + k], c_time[k]);
        for (int k = 0; k < 4; k++) atomicAdd(&A.stats[16 + k], sub[k]);
#endif
#ifdef RTW_ENDTIMES
        // diagnostic build only: when do the waves of a launch start and finish?  (overwrites the census slots)
        const unsigned long long t_end = __builtin_amdgcn_s_memtime();
        atomicMax(&A.stats[12], t_end - t_wave_start);     // longest wave lifetime (s_memtime is per XCD: only differences within a wave mean anything)
        if (A.endtimes_ref) {                              // histogram of the lifetimes in 1/32 of a reference lifetime (a previous run's longest): stats[20 + 0..11] = the last 12 bins below / at it
            const unsigned long long bin = (t_end - t_wave_start) * 32ull / A.endtimes_ref;
            const unsigned long long b = bin >= 32ull ? 11ull : (bin >= 21ull ? bin - 21ull : 0ull);
            atomicAdd(&A.stats[20 + b], 1ull);
        }
        atomicAdd(&A.stats[13], t_end - t_wave_start);     // sum of the waves' lifetimes
        atomicAdd(&A.stats[15], 1ull);                     // waves
        const unsigned long long rt_end = wall_clock64();
        atomicMin(&A.stats[24 + 8 - 8], rt_wave_start + 0ull);      // [24] earliest wave start   (device-wide 100 MHz ticks; [24], [26] are pre-set to ~0ull by the shim)
        atomicMax(&A.stats[25], rt_wave_start);            // [25] latest wave start
        atomicMin(&A.stats[26], rt_end);                   // [26] earliest wave end
        atomicMax(&A.stats[27], rt_end);                   // [27] latest wave end
        if (rs.t_dry) { atomicMin(&A.stats[28], rs.t_dry); atomicMax(&A.stats[29], rs.t_dry); atomicMax(&A.stats[30], rt_end - rs.t_dry); atomicAdd(&A.stats[31], rt_end - rs.t_dry); }
        if (rs.t_dry && rt_end - rs.t_dry >= 60000ull) {   // the waves that ran on for 0.6 ms or more: [40] how many, [41] ticks, [42] trips, [43..45] steps T / L / S, [46..48] their lanes, [49] max queries of a lane, [50] sum
            atomicAdd(&A.stats[40], 1ull); atomicAdd(&A.stats[41], rt_end - rs.t_dry); atomicAdd(&A.stats[42], (unsigned long long)dry_trips);
            for (int k = 0; k < 3; k++) { atomicAdd(&A.stats[43 + k], (unsigned long long)dry_steps[k]); atomicAdd(&A.stats[46 + k], (unsigned long long)dry_lanes[k]); }
            atomicAdd(&A.stats[49], (unsigned long long)dry_q_max); atomicAdd(&A.stats[50], (unsigned long long)dry_q_sum);
        }
#endif
    }
}

// ================================================================================================
// host side
// ================================================================================================
typedef void (*kernel_fn)(const KArgs);
static bool is_common_config(const KArgs &a) {
    return a.integrator == RTW_INTEGRATOR_GRADIENT && a.sampler == RTW_SAMPLER_ROW && a.depth >= 1 &&
           (a.flags & (RTW_FLAG_CPP_DIELECTRIC | RTW_FLAG_CPP_DIFFUSE)) == 0u;
}
static bool is_demo_config(const KArgs &a) {                   // presentation_image's: ray_color_bg_color through render_row
    return a.integrator == RTW_INTEGRATOR_BG_COLOR && a.sampler == RTW_SAMPLER_ROW && a.depth >= 1 &&
           (a.flags & (RTW_FLAG_CPP_DIELECTRIC | RTW_FLAG_CPP_DIFFUSE)) == 0u;
}
static bool is_serial_config(const KArgs &a) {                 // Viewport::render's: ray_color_gradient, stratified
    return a.integrator == RTW_INTEGRATOR_GRADIENT && a.sampler == RTW_SAMPLER_STRATIFIED && a.depth >= 1 &&
           (a.flags & (RTW_FLAG_CPP_DIELECTRIC | RTW_FLAG_CPP_DIFFUSE)) == 0u;
}
static bool is_rust2_config(const KArgs &a) {                  // Rust2's: ray_color through its fixed-centre render_row
    return a.integrator == RTW_INTEGRATOR_RUST2 && a.sampler == RTW_SAMPLER_CENTRES && a.depth >= 1 &&
           (a.flags & (RTW_FLAG_CPP_DIELECTRIC | RTW_FLAG_CPP_DIFFUSE)) == 0u;
}
template <int SPEC>
static kernel_fn pick_kernel_spec(bool moving, uint32_t accel, int nodes) {
    if (accel == RTW_ACCEL_BVH) {
        if (nodes == 2) return moving ? render_bvh<true, 2, SPEC, false> : render_bvh<false, 2, SPEC, false>;
        if (nodes == 1) return moving ? render_bvh<true, 1, SPEC, false> : render_bvh<false, 1, SPEC, false>;
        return moving ? render_bvh<true, 0, SPEC, false> : render_bvh<false, 0, SPEC, false>;
    }
    return moving ? render_brute<true, SPEC, false> : render_brute<false, SPEC, false>;
}
// quads / instances in the scene: the step of the generic build (SPEC == 0: everything from the kernel arguments; SPEC == 2: the common configuration folded
// in at compile time, sphere textures kept) with the extra closest-hit stage (sphere geometry always global: kernel_has_lds_geom)
template <int SPEC>
static kernel_fn pick_kernel_geom(bool moving, uint32_t accel, int nodes) {
    if (accel == RTW_ACCEL_BVH) {
        if (nodes) return moving ? render_bvh<true, 1, SPEC, true> : render_bvh<false, 1, SPEC, true>;
        return moving ? render_bvh<true, 0, SPEC, true> : render_bvh<false, 0, SPEC, true>;
    }
    return moving ? render_brute<true, SPEC, true> : render_brute<false, SPEC, true>;
}
static kernel_fn pick_kernel(const KArgs &a, bool moving, uint32_t accel, bool lds_nodes) {
    const int nodes = lds_nodes ? (a.lds_geom_off ? 2 : 1) : 0;
    if (a.geom.n_quads || a.geom.n_inst) {
#ifndef RTW_GEOM_GENERIC_ONLY
        if (is_common_config(a) && !(a.flags & RTW_FLAG_CHUNK_SUMS)) return pick_kernel_geom<2>(moving, accel, nodes);
        if (is_demo_config(a) && !(a.flags & RTW_FLAG_CHUNK_SUMS)) return pick_kernel_geom<4>(moving, accel, nodes);
        if (is_rust2_config(a) && !(a.flags & RTW_FLAG_CHUNK_SUMS)) return pick_kernel_geom<5>(moving, accel, nodes);
        if (is_serial_config(a) && !(a.flags & RTW_FLAG_CHUNK_SUMS)) return pick_kernel_geom<6>(moving, accel, nodes);
#endif
        return pick_kernel_geom<0>(moving, accel, nodes);
    }
#ifndef RTW_GEOM_GENERIC_ONLY
    if (is_demo_config(a) && !(a.flags & RTW_FLAG_CHUNK_SUMS)) return pick_kernel_spec<4>(moving, accel, nodes);    // (the generic build's step, switches folded in)
    if (is_rust2_config(a) && !(a.flags & RTW_FLAG_CHUNK_SUMS)) return pick_kernel_spec<5>(moving, accel, nodes);
    if (is_serial_config(a) && !(a.flags & RTW_FLAG_CHUNK_SUMS)) return pick_kernel_spec<6>(moving, accel, nodes);
#endif
    if (!is_common_config(a)) return pick_kernel_spec<0>(moving, accel, nodes);
    if (a.flags & RTW_FLAG_CHUNK_SUMS) return a.has_textures ? pick_kernel_spec<0>(moving, accel, nodes) : pick_kernel_spec<3>(moving, accel, nodes);
    return a.has_textures ? pick_kernel_spec<2>(moving, accel, nodes) : pick_kernel_spec<1>(moving, accel, nodes);
}

bool kernel_has_lds_geom(const KArgs &a) { return !(a.geom.n_quads || a.geom.n_inst); }

void launch_render(const KArgs &a, bool moving, uint32_t accel, uint32_t grid, hipStream_t stream) {
    hipLaunchKernelGGL(pick_kernel(a, moving, accel, a.bvh.nodes16 != nullptr), dim3(grid), dim3(RTW_BLOCK), a.lds_bytes, stream, a);
    hipLaunchKernelGGL(resolve_kernel, dim3((a.n_tiles * 64u + RTW_BLOCK - 1) / RTW_BLOCK), dim3(RTW_BLOCK), 0, stream, a);
}

uint32_t kernel_blocks_per_cu(const KArgs &a, bool moving, uint32_t accel, bool lds_nodes) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, pick_kernel(a, moving, accel, lds_nodes), RTW_BLOCK, a.lds_bytes) != hipSuccess || n < 1) n = 1;
    return (uint32_t)(n > 8 ? 8 : n);
}

const void *kernel_id(const KArgs &a, bool moving, uint32_t accel, bool lds_nodes) {
    return (const void *)pick_kernel(a, moving, accel, lds_nodes);
}

} // namespace rtw
