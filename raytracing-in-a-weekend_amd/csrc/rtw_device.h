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
//   Rust/src/objects/quad.rs:37-81         Quad::collision_normal
//   Rust/src/objects/instance.rs:250-310   Instance::collision_normal (+ const_density :24-26)
//   Rust/src/vec3.rs:161-181               Vec3::rotated
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rtw.h"

namespace rtw {

// Lane mask of a predicate.  HIP's __ballot(int) compares the predicate, widened to an int, with zero: for a bool that lives in an SGPR pair (a
// flag carried from a divergent branch) the compiler then materialises 0 / 1 in a VGPR and compares it again -- two VALU instructions of the slow class
// for what is an s_and with exec.  The builtin takes the i1 itself.
__device__ __forceinline__ unsigned long long ballot64(bool c) { return __builtin_amdgcn_ballot_w64(c); }

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
// ---- IEEE sqrt and division without the exponent-range machinery -----------------------------------------------------
// hipcc expands a correctly rounded f32 sqrt to 17 and a division to 12 instructions; most of them (pre-scaling, v_div_scale /
// v_div_fmas / v_div_fixup, the zero / inf / NaN classes) only serve operands near the ends of the exponent range.  The helpers
// below are the SAME arithmetic with those steps -- identities for "plain" operands -- left out; callers check the operand
// ranges for the whole wave (one ballot) and fall back to the generic expansion otherwise, so the bits never differ.
//   sqrt_plain(x):  x in [2^-96, 2^127)        v_sqrt_f32 (<= 1 ulp) + the two one-ulp residual corrections
//   rcp_refined(d), div_plain(n, d, r):  |d| in [2^-40, 2^40], |n| in [2^-60, 2^40]    v_rcp_f32 + one Newton step (shared by every
//                   quotient over d), then the three-fma refinement of n * r
__device__ __forceinline__ float sqrt_plain(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_dn = __builtin_fmaf(-s_dn, s, x), e_up = __builtin_fmaf(-s_up, s, x);
    const float t = (0.0f >= e_dn) ? s_dn : s;
    return (0.0f < e_up) ? s_up : t;
}
// sqrt(x), correctly rounded: the plain sequence when every active lane's x is in [2^-96, 2^127), hipcc's expansion otherwise (same bits)
__device__ __forceinline__ float sqrt_ieee(float x) {
    const bool plain = (__float_as_uint(x) - 0x0F800000u) < (0x7F000000u - 0x0F800000u);
    if (ballot64(!plain) == 0ull) return sqrt_plain(x);
    return __builtin_sqrtf(x);
}
__device__ __forceinline__ float rcp_refined(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float div_plain(float n, float d, float r) {
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, n), r, q);
}
// The accepted root of a sphere test, `(-b - sqrt(disc)) / a`, or `(-b + sqrt(disc)) / a` when that one lies before mint (sphere.rs:106-116),
// for lanes with disc >= 0 (the caller's exec mask).  `ra` = rcp_refined(a); `a_plain` (wave-uniform) says every lane's a is in
// [2^-20, 2^20].  With disc in [2^-60, 2^96] in every active lane as well, the plain sequences give the bits of the generic expansions:
// sqrt(disc) >= 2^-30 makes each numerator either 0 (quotient +-0 either way: below mint, never stored) or >= 2^-54 in magnitude (it is the
// rounded sum of two floats of which one is >= 2^-30), so no residual of div_plain can leave the normal range (they are multiples of
// ulp(a) * ulp(q) >= 2^-43 * 2^-98); disc <= 2^96 also bounds |b| < 2^64 (b * b would have overflowed), so no quotient overflows.
// One integer range compare and one ballot per test; any other lane (disc = -0, a denormal, inf, NaN) sends the wave down the generic code.
__device__ __forceinline__ float sphere_root(float b, float disc, float a, float ra, bool a_plain, float mint) {
    const bool plain = (__float_as_uint(disc) - 0x21800000u) <= (0x6F800000u - 0x21800000u);      // 2^-60 <= disc <= 2^96
    float x;
    if (a_plain && ballot64(!plain) == 0ull) {
        const float sq = sqrt_plain(disc);
        x = div_plain(-b - sq, a, ra);
        if (x < mint) x = div_plain(-b + sq, a, ra);
    } else {
        const float sq = __builtin_sqrtf(disc);
        x = (-b - sq) / a;
        if (x < mint) x = (-b + sq) / a;
    }
    return x;
}
__device__ __forceinline__ bool in_range(float v, float lo, float hi) { const float a = __builtin_fabsf(v); return a >= lo && a <= hi; }

// Diagnostic build (-DRTW_CENSUS): how many lanes are live in each sub-block of a SHADE step.  Per-lane counters (summed over the wave at
// the end of the kernel): n[k] counts the lanes that executed block k, w[k] its wave-level executions (the first active lane counts).
// The product build compiles none of it (RTW_CEN expands to nothing, the pointer parameters are null and fold away).
enum { CEN_SHADING = 0, CEN_INFLIGHT, CEN_MISS, CEN_HIT, CEN_DIELECTRIC, CEN_SCHLICK, CEN_DIFFUSE, CEN_UV_TRIP, CEN_BANK, CEN_NEED_UNIT,
       CEN_START_PATH, CEN_DISK_TRIP, CEN_TRAV_BEGIN, CEN_DEPTH_END, CEN_BIG_ROOT, CEN_COOP, CEN_N };
struct Cen { uint32_t n[CEN_N], w[CEN_N]; };
#ifdef RTW_CENSUS
__device__ __forceinline__ void cen_count(Cen *cn, int k) {
    if (!cn) return;
    const unsigned long long m = ballot64(true);
    cn->n[k]++;
    cn->w[k] += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u ? 1u : 0u;
}
#define RTW_CEN(cn, k) cen_count(cn, k)
#else
#define RTW_CEN(cn, k) do { } while (0)
#endif

// unit(a) = a / sqrt(a.a) (vec3.rs:213), correctly rounded sqrt and divisions.  hipcc expands every IEEE f32 sqrt to 17 and every
// IEEE division to 12 instructions; most of those only serve operands near the ends of the exponent range (pre-scaling,
// v_div_scale / v_div_fmas / v_div_fixup, the zero / inf / NaN classes).  When every lane of the wave has all three components
// in [2^-40, 2^40] none of that can trigger: the sequence below is then the SAME arithmetic hipcc emits -- v_sqrt_f32 plus the
// two one-ulp residual corrections; v_rcp_f32, its Newton step, and the three-fma quotient refinement -- with the scaling
// steps (identities there) removed and the reciprocal shared by the three quotients: 36 instead of 58 VALU, same bits.
// Any lane outside the range (a zero component, a denormal, an overflow) sends the whole wave down the generic expansion.
__device__ __forceinline__ v3 unit(v3 a) {
    const float lo = fminf(fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    const float hi = fmaxf(fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z));
    const bool plain = lo >= 0x1p-40f && hi <= 0x1p40f;
    if (ballot64(!plain) == 0ull) {
        const float s = sqrt_plain(a.x * a.x + a.y * a.y + a.z * a.z);          // argument in [2^-80, 2^82)
        const float r = rcp_refined(s);
        return mk(div_plain(a.x, s, r), div_plain(a.y, s, r), div_plain(a.z, s, r));
    }
    return a / len(a);
}
// vec3.rs:256  self - (n * 2.0) * self.dot(n)
__device__ __forceinline__ v3 reflect(v3 a, v3 n) { return a - (n * 2.0f) * dot(a, n); }
__device__ __forceinline__ bool close_to_zero(v3 a) {
    return __builtin_fabsf(a.x) < 1e-7f && __builtin_fabsf(a.y) < 1e-7f && __builtin_fabsf(a.z) < 1e-7f;
}

// ---- RNG: one short stream per (seed, pixel, sample): a 32-bit LCG (the PCG family's multiplier) with a per-stream odd increment, start
// state and increment from a lowbias32 hash chain; a draw is the top 24 bits of the state.  Same definition as oracle/rtw_oracle.c
// (independently written there).  Until the middle of round 2 PCG's RXS-M-XS output permutation ran on top: 7 more integer instructions
// per draw, ~27 draws per SHADE step of a wave = 9 % of the frame (profiles/r02_ab_rng_cost.log), and nothing measurable in return for
// streams of a few dozen draws from hashed starts (tests/test_rng_quality.py; DESIGN.md "RNG").
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
// the top 24 bits of the next output, as a float: n in [0, 2^24), exact
__device__ __forceinline__ float rng_n24(Rng &r) {
    const uint32_t old = r.state;
    r.state = old * 747796405U + r.inc;
    return (float)(old >> 8);
}
__device__ __forceinline__ float rng_f32(Rng &r) { return rng_n24(r) * (1.0f / 16777216.0f); }
// xi * 2 + (-1) and c + xi in ONE instruction each: n * 2^-k is exact (n < 2^24, power-of-two scale), so the fused
// multiply-add rounds exactly once, at the same place as the reference's final addition -- bit-identical to
// `random::<f32>() * (max - min) + min` (vec3.rs:221-227) and to `i as f32 + random::<f32>()` (viewport.rs:290).
__device__ __forceinline__ float rng_sym(Rng &r) { return __builtin_fmaf(rng_n24(r), 1.0f / 8388608.0f, -1.0f); }
__device__ __forceinline__ float rng_offset(Rng &r, float c) { return __builtin_fmaf(rng_n24(r), 1.0f / 16777216.0f, c); }
// vec3.rs:228-239.  `strict` (RTW_FLAG_CPP_DIFFUSE): the C++ twin accepts |p|^2 < 1 (C++/src/vec3.cpp:28-34), Rust <= 1.
__device__ __forceinline__ v3 random_unit_vec(Rng &r, bool strict = false, Cen *cn = nullptr) {
    v3 p;
    for (;;) {
        RTW_CEN(cn, CEN_UV_TRIP);
        p.x = rng_sym(r);                  // xi * (max - min) + min
        p.y = rng_sym(r);
        p.z = rng_sym(r);
        const float l2 = p.x * p.x + p.y * p.y + p.z * p.z;
        if (strict ? l2 < 1.0f : l2 <= 1.0f) break;
    }
    return unit(p);
}
// vec3.rs:240-254 (C++/headers/vec3.h:35-41 with `strict`)
__device__ __forceinline__ void random_in_unit_disk(Rng &r, float &px, float &py, bool strict = false, Cen *cn = nullptr) {
    for (;;) {
        RTW_CEN(cn, CEN_DISK_TRIP);
        px = rng_sym(r);                   // xi * 2.0 - 1.0
        py = rng_sym(r);
        const float l2 = px * px + py * py;
        if (strict ? l2 < 1.0f : l2 <= 1.0f) break;
    }
}

// Both rejection samplers of a SHADE step in ONE loop (specialised builds).  In a SHADE step the lanes whose ray hit a non-dielectric
// sphere need random_unit_vec (vec3.rs:228-239: three draws per candidate) and the lanes that start a new path need random_in_unit_disk
// (vec3.rs:240-254: two draws per candidate) -- different lanes, the same instructions.  Run one after the other the two loops cost
// 5.55 + 2.76 trips per step at 0.16 / 0.15 live lanes (profiles/r03_census_baseline.txt); as one loop the disk lanes ride along in the
// trips the ball lanes need anyway.  A disk lane's third draw is switched off by per-lane constants rather than by an exec-mask region:
// its LCG step multiplies by 1 and adds 0 (the state stays), its z is fma(n, 0, 0) = +0, and (x*x + y*y) + 0 has the bits of x*x + y*y.
// Every lane consumes exactly the draws the reference's loops consume, in their order.  `ball`: this lane wants a point of the unit
// ball (else of the unit disk); only lanes with `need` enter.
#ifdef RTW_COOP_SAMPLER
// EXPERIMENT (measured and rejected, profiles/r03_ab_coop.log): the wave-cooperative form of the loop.  Once at most eight
// lanes are still without an accepted candidate, ONE wave-uniform round tests the next eight candidates of each of them in parallel: the wave is
// eight blocks of eight lanes, the k-th pending lane ("owner") hands its stream to block k (ds_permute to the block's first lane, ds_swizzle
// over the block), lane c of the block jumps the LCG ahead by c candidates (s' = A^j s + (1 + A + .. + A^(j-1)) inc, constants from a table),
// draws and tests candidate c exactly as the plain loop would have, and the owner takes the FIRST accepted one of its block (a byte of the accept
// ballot) with the state after it (ds_bpermute) -- the same draws, the same decisions, the same bits as the sequential loop.
struct JumpTable { uint32_t a[16], c[16]; };
__host__ __device__ constexpr JumpTable make_jump_table() {
    JumpTable t{};
    for (int e = 0; e < 16; e++) {                       // entries 0..7: candidate c of a DISK stream (2 draws each); 8..15: of a BALL stream (3 draws each)
        const int j = (e < 8 ? 2 : 3) * (e & 7);
        uint32_t a = 1u, c = 0u;
        for (int k = 0; k < j; k++) { c = c * 747796405U + 1u; a = a * 747796405U; }     // s -> A s + inc, j times: (a, c) o (A, 1)
        t.a[e] = a; t.c[e] = c;
    }
    return t;
}
__device__ __constant__ JumpTable RTW_JUMP = make_jump_table();

__device__ __forceinline__ void coop_round(Rng &r, bool &pend, bool ball, float &x, float &y, float &z, float &len2, unsigned long long P, uint32_t lane, Cen *cn) {
    RTW_CEN(cn, CEN_COOP);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(P >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)P, 0u));
    // owners push (state, inc with the ball flag in its always-set low bit) to the first lane of their block; every other lane pushes to a lane
    // that is nobody's first lane (its own block's last), so that no push can collide with an owner's
    const int dst = (int)(pend ? rank << 5 : (lane << 2) | 28u);
    const int t1 = __builtin_amdgcn_ds_permute(dst, (int)r.state);
    const int t2 = __builtin_amdgcn_ds_permute(dst, (int)((r.inc & ~1u) | (ball ? 1u : 0u)));
    const uint32_t b1 = (uint32_t)__builtin_amdgcn_ds_swizzle(t1, 0x18), b2 = (uint32_t)__builtin_amdgcn_ds_swizzle(t2, 0x18);   // lane & 0x18 of each half: the block's first lane
    const bool bball = (b2 & 1u) != 0u;
    const uint32_t binc = b2 | 1u, c = lane & 7u, e = c + (bball ? 8u : 0u);
    uint32_t st = RTW_JUMP.a[e] * b1 + RTW_JUMP.c[e] * binc;                     // the owner's stream, c candidates ahead
    float cx, cy, cz;
    { uint32_t old = st; st = old * 747796405U + binc; cx = __builtin_fmaf((float)(old >> 8), 1.0f / 8388608.0f, -1.0f); }
    { uint32_t old = st; st = old * 747796405U + binc; cy = __builtin_fmaf((float)(old >> 8), 1.0f / 8388608.0f, -1.0f); }
    { const uint32_t old = st; st = bball ? old * 747796405U + binc : old; cz = bball ? __builtin_fmaf((float)(old >> 8), 1.0f / 8388608.0f, -1.0f) : 0.0f; }
    const float l2 = cx * cx + cy * cy + cz * cz;
    const unsigned long long M = ballot64(l2 <= 1.0f);
    // the owner's block is byte `rank` of the accept mask; its lowest set bit is the first accepted candidate
    const uint32_t mine = (uint32_t)(M >> (rank * 8u)) & 0xFFu;
    const uint32_t cstar = mine ? (uint32_t)__builtin_ctz(mine) : 8u;
    const int src = (int)((rank * 8u + (cstar < 8u ? cstar : 7u)) << 2);
    const float nx = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(cx)));
    const float ny = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(cy)));
    const float nl = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(l2)));
    const float nz = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(cz)));
    const uint32_t ns = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)st);
    if (pend) {
        r.state = ns;                                   // after the accepted candidate, or after all eight
        if (cstar < 8u) { x = nx; y = ny; z = nz; len2 = nl; pend = false; }
    }
}
#ifndef RTW_COOP_MAX
#define RTW_COOP_MAX 8
#endif
__device__ __forceinline__ void sample_ball_or_disk(Rng &r, bool need, bool ball, float &x, float &y, float &z, float &len2, Cen *cn = nullptr) {
    bool pend = need;
    const uint32_t am = ball ? 747796405U : 1U, ai = ball ? r.inc : 0U;
    const float kz = ball ? (1.0f / 8388608.0f) : 0.0f, oz = ball ? -1.0f : 0.0f;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    for (;;) {
        if (pend) {
            RTW_CEN(cn, CEN_UV_TRIP);
            x = rng_sym(r);
            y = rng_sym(r);
            const uint32_t old = r.state;
            r.state = old * am + ai;
            z = __builtin_fmaf((float)(old >> 8), kz, oz);
            const float l2 = x * x + y * y + z * z;
            len2 = l2;
            if (l2 <= 1.0f) pend = false;
        }
        const unsigned long long P = ballot64(pend);
        if (P == 0ull) break;
        if ((uint32_t)__popcll(P) <= (uint32_t)RTW_COOP_MAX) coop_round(r, pend, ball, x, y, z, len2, P, lane, cn);
    }
}
#else
__device__ __forceinline__ void sample_ball_or_disk(Rng &r, bool need, bool ball, float &x, float &y, float &z, float &len2, Cen *cn = nullptr) {
    if (need) {
        const uint32_t am = ball ? 747796405U : 1U, ai = ball ? r.inc : 0U;
        const float kz = ball ? (1.0f / 8388608.0f) : 0.0f, oz = ball ? -1.0f : 0.0f;
#ifdef RTW_EXPERIMENT_CAP_TRIPS
        int trip = 0;
#endif
        for (;;) {
            RTW_CEN(cn, CEN_UV_TRIP);
            x = rng_sym(r);
            y = rng_sym(r);
            const uint32_t old = r.state;
            r.state = old * am + ai;
            z = __builtin_fmaf((float)(old >> 8), kz, oz);
            const float l2 = x * x + y * y + z * z;
            len2 = l2;                               // |p|^2 of the accepted point: unit_of_ball_point() divides by its root
            if (l2 <= 1.0f) break;
#ifdef RTW_EXPERIMENT_CAP_TRIPS      /* WRONG IMAGES: an upper bound on what any scheme that shortens the loop's tail can gain (profiles/r03_ab_coop_bound.log) */
            if (++trip >= RTW_EXPERIMENT_CAP_TRIPS) { const float s = 0.5f * __builtin_amdgcn_rsqf(l2); x *= s; y *= s; z *= s; break; }
#endif
        }
    }
}
#endif

// ---- device scene ------------------------------------------------------------------------------
// Hot, wave-uniform stream (scalar loads):  geom[i] = {cx, cy, cz, r*r},  vel[i] = {vx, vy, vz, 0}.
// Cold, per-lane record fetched only for the sphere a lane hit:
struct DevMat {
    float cm[3];          // tex < 0: tex_color * col_mod (sphere.rs:145, hoisted); else col_mod
    float metallicness;
    float opacity;
    float ir;
    int32_t tex;
    float inv_ir;         // 1.0 / ir                        } the material-only divisions of the dielectric branch
    float emitted[3];
    float r0_front;       // ((1 - 1/ir) / (1 + 1/ir))^2     } (materials.rs:99-100,113), done once on the host: the same
    float r0_back;        // ((1 - ir) / (1 + ir))^2         } single IEEE f32 operations, hoisted like radius * radius
    uint32_t tex_row, tex_col, tex_offset;   // tex >= 0: RtwTexture.row / col / texel_offset of that image, copied by the host (one dependent load less per textured hit)
};
static_assert(sizeof(DevMat) == 64, "DevMat is four f4 rows");

// The f32 operations of `reflectance` that depend on the material alone (materials.rs:99-100): r0 = ((1-x)/(1+x))^2.
// __host__ __device__: the shim evaluates it once per sphere, the quad path per hit -- IEEE on both, same bits.
__host__ __device__ __forceinline__ float schlick_r0(float ref_idx) {
    float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
    return r0 * r0;
}

struct DevScene {
    const f4 *geom;
    const f4 *vel;
    const DevMat *mat;
    const RtwTexture *tex;
    const float *texels;
    uint32_t n;
    uint32_t moving;
};

// `Quad` with the derived fields of Quad::new (quad.rs:96-108), seven 16-byte rows read by scalar loads.
struct DevQuad {
    float origin[3]; float d;             // d = normal . origin
    float u[3];      float metallicness;
    float v[3];      float opacity;
    float normal[3]; float ir;            // unit(u x v)
    float w[3];      int32_t tex;         // n / (n . n)
    float albedo[3]; uint32_t pad0;       // tex < 0: the 1x1 texel * 1.0 (texture.rs:265)
    float emitted[3]; uint32_t pad1;
};
static_assert(sizeof(DevQuad) == 112, "DevQuad is seven f4 rows");

// `Instance` (instance.rs:27-38): member ranges, translation, and sin/cos of -rotation / +rotation
// (Vec3::rotated recomputes them per call, vec3.rs:163-170; hoisted to the host, same libm).
struct DevInstance {
    uint32_t first_sphere, n_spheres, first_quad, n_quads;
    float tr[3]; float density;
    float back[6]; float fwd[6];          // asin, acos, bsin, bcos, csin, ccos
    // the four coefficient expressions of Vec3::rotated that involve the angles alone (vec3.rs:173-178), evaluated once on the host
    // with the same f32 operations in the same order:  asin*bsin*ccos - asin*ccos,  acos*bsin*ccos + asin*csin,
    // asin*bsin*csin + acos*ccos,  acos*bsin*csin - asin*ccos
    float back_k[4]; float fwd_k[4];
    uint32_t medium; uint32_t pad[3];
};
static_assert(sizeof(DevInstance) == 128, "DevInstance is eight f4 rows");

// (host side of the above; __host__ __device__ so that both compilers see one definition)
__host__ __device__ __forceinline__ void rotation_coefficients(const float *q, float *k) {
    const float as = q[0], ac = q[1], bs = q[2], cs = q[4], cc = q[5];
    k[0] = as * bs * cc - as * cc;
    k[1] = ac * bs * cc + as * cs;
    k[2] = as * bs * cs + ac * cc;
    k[3] = ac * bs * cs - as * cc;
}

struct DevGeom {                          // everything of `Scene` that is not a top-level sphere
    const DevQuad *quads;
    const DevInstance *inst;
    const f4 *igeom;                      // instance-member spheres: same three streams as DevScene
    const f4 *ivel;
    const DevMat *imat;
    const DevQuad *iquads;
    uint32_t n_quads, n_inst;
};

// Rust `f as usize` (saturating, NaN -> 0), then clamped to the image like the oracle
__device__ __forceinline__ uint32_t tex_index(float f, uint32_t last) {
    if (!(f > 0.0f)) return 0;
    if (f >= (float)last) return last;
    return (uint32_t)f;
}

// ---- the two libm calls of the spherical UV (sphere.rs:132-133: f32::atan2, f32::acos) ---------------------------------------------------
// Rust's f32::atan2 / acos are the platform libm's, accurate to about an ulp; which ulp decides nothing but on which side of a texel edge a
// hit within ~1e-7 of it falls ("a texel edge can move by an ulp").  ocml's atan2f + acosf cost ~130 VALU instructions of every
// SHADE step of a textured scene (C5: +9 % of the frame, profiles/r02_c5_parts.log), most of it for arguments a unit normal cannot have.
// The two below are the same classic reductions with minimax polynomials (fitted for this file; max error against f64 over 4 M random unit
// normals, evaluated in f32 with one rounding per operation: atan2 2.03 ulp, acos 1.25 ulp; texel choice against a correctly rounded libm:
// identical for textures up to 4 x 4, 2 texels per million hits for a 128-texel-wide image -- tests/test_round3_cpu.py re-measures both from
// the oracle's copy of the same code).  There is NO second path through ocml behind a range check (tried: with both inlined the C5 kernel went
// from 52 to 80 bytes of scratch and from 93.3 to 101.6 ms, profiles/r03_ab_c5_uv.log): the two functions are total -- atan2(+-0, +-0) and NaN
// arguments give libm's answers, tiny arguments are rescaled, acos of |x| > 1 is NaN.
//   atan2(y, x):  t = min / max (correctly rounded), atan t = t + t s P(s), s = t^2, then the octant fix-ups
__device__ __forceinline__ float atan2_plain(float y, float x) {
    const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
    const float k = fmaxf(ax, ay) < 0x1p-60f ? 0x1p80f : 1.0f;                      // (a direction's arctangent does not depend on its length)
    const float mx = fmaxf(ax, ay) * k, mn = fminf(ax, ay) * k;
    const float t = mx == 0.0f ? 0.0f : div_plain(mn, mx, rcp_refined(mx));           // atan2(+-0, +-0) = +-0 or +-pi: t = 0
    const float s = t * t;
    float p = 0.002974563976749778f;
    p = __builtin_fmaf(p, s, -0.016581078991293907f); p = __builtin_fmaf(p, s, 0.043553370982408524f); p = __builtin_fmaf(p, s, -0.07580564171075821f);
    p = __builtin_fmaf(p, s, 0.10678933560848236f);   p = __builtin_fmaf(p, s, -0.14214207231998444f); p = __builtin_fmaf(p, s, 0.19994136691093445f);
    p = __builtin_fmaf(p, s, -0.3333316743373871f);
    float a = __builtin_fmaf(t * s, p, t);
    a = ay > ax ? 1.57079632679489661923f - a : a;
    a = __float_as_int(x) < 0 ? 3.14159265358979323846f - a : a;                     // the SIGN of x, so that atan2(y, -0) is libm's
    a = (x != x || y != y) ? __builtin_nanf("") : a;
    return __builtin_copysignf(a, y);
}
//   acos(x), |x| <= 1:  |x| <= 1/2: pi/2 - asin x;  else 2 asin sqrt((1 - |x|) / 2), reflected for x < 0;  asin q = q + q s R(s)
__device__ __forceinline__ float acos_plain(float x) {
    const float ax = __builtin_fabsf(x);
    const bool small = ax <= 0.5f;
    const float s = small ? x * x : (1.0f - ax) * 0.5f;
    const float q = small ? x : sqrt_plain(s);
    float p = 0.04221854731440544f;
    p = __builtin_fmaf(p, s, 0.02414761111140251f); p = __builtin_fmaf(p, s, 0.04547709599137306f); p = __builtin_fmaf(p, s, 0.07495241612195969f);
    p = __builtin_fmaf(p, s, 0.16666753590106964f);
    const float r = __builtin_fmaf(q * s, p, q);
    const float two = r + r;
    return small ? 1.57079632679489661923f - r : (x < 0.0f ? 3.14159265358979323846f - two : two);
}
// (u, v) of a sphere's outward normal (sphere.rs:132-133)
__device__ __forceinline__ void sphere_uv(v3 normal, float &u, float &v) {
    const float PI = 3.14159265358979323846f, FRAC_1_PI = 0.318309886183790671538f;
    u = (atan2_plain(-normal.z, normal.x) + PI) * FRAC_1_PI * 0.5f;
    v = 1.0f - (FRAC_1_PI * acos_plain(-normal.y));
}

// sphere.rs:129-146 + texture.rs:259-267
__device__ __forceinline__ v3 sphere_albedo(const DevScene &sc, const DevMat &m, v3 normal) {
    if (m.tex < 0) return ld3(m.cm);
    float u, v;
    sphere_uv(normal, u, v);
    uint32_t tx = tex_index(floorf(u * (float)(m.tex_row - 1)), m.tex_row - 1);
    uint32_t ty = tex_index(floorf(v * (float)(m.tex_col - 1)), m.tex_col - 1);
    const float *px = sc.texels + 3 * (size_t)(m.tex_offset + ty * m.tex_row + tx);
    return (ld3(px) * 1.0f) * ld3(m.cm);
}

// Rust2's ImageTexture::color_at (Rust2/src/objects/texture.rs:94-105), reached from Rust2's Sphere::color (Rust2/src/objects/sphere.rs:92-107):
// multiplied = img[x * width + y] with x = (u * width) as usize, y = (v * height) as usize -- scaled by the size, not size - 1, and indexed
// TRANSPOSED (part of the contract: SURVEY.md 8 a10) --; emmited = emmit_img[emmit_x * emmit_width + emmit_y] with floor() before the casts.
// Where the reference would panic (an index past the Vec: u == 1, tall images) the index is clamped to the last texel, as in the oracle.
__device__ __forceinline__ uint32_t rust2_texel_index(float u, float v, uint32_t width, uint32_t height, bool floor_first) {
    const float fx = u * (float)width, fy = v * (float)height;
    const uint64_t x = tex_index(floor_first ? floorf(fx) : fx, 0xFFFFFFFFu), y = tex_index(floor_first ? floorf(fy) : fy, 0xFFFFFFFFu);
    const uint64_t idx = x * (uint64_t)width + y, last = (uint64_t)width * height - 1u;
    return (uint32_t)(idx > last ? last : idx);
}
__device__ __forceinline__ void rust2_sphere_color(const DevScene &sc, const DevMat &m, v3 normal, v3 &multiplied, v3 &emmited) {
    const RtwTexture t = sc.tex[m.tex];
    float u, v;
    sphere_uv(normal, u, v);
    multiplied = ld3(sc.texels + 3 * (size_t)(t.texel_offset + rust2_texel_index(u, v, t.row, t.col, false))) * ld3(m.cm);   // (cm = col_mod: 1 in a Rust2 scene)
    if (t.emit_tex != 0u) {
        const RtwTexture e = sc.tex[t.emit_tex - 1u];
        emmited = ld3(sc.texels + 3 * (size_t)(e.texel_offset + rust2_texel_index(u, v, e.row, e.col, true)));
    }
}

// materials.rs:89-97
__device__ __forceinline__ v3 refract(v3 uv, v3 n, float etai_over_etat) {
    float cos_theta = dot(-uv, n);
    if (cos_theta > 1.0f) cos_theta = 1.0f;
    v3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    v3 r_out_parallel = n * -sqrt_ieee(__builtin_fabsf(1.0f - len2(r_out_perp)));
    return r_out_perp + r_out_parallel;
}
// materials.rs:98-103, powi(5) = x * ((x*x)*(x*x))
__device__ __forceinline__ float reflectance(float cosine, float r0) {   // r0 = schlick_r0(ref_idx)
    float x = 1.0f - cosine;
    float x2 = x * x;
    float x5 = x * (x2 * x2);
    return r0 + (1.0f - r0) * x5;
}

// Material::on_hit (materials.rs:105-154) followed by the degenerate-direction fix-up of
// ray_color_* (ray_color.rs:31-33).  Returns the next direction; cos_theta for bg_color.
// The scalars of `Material` on_hit reads (materials.rs:15-20) plus the three values derived from `ir` alone.
struct MatP { float metallicness, opacity, ir, inv_ir, r0_front, r0_back; };
__device__ __forceinline__ MatP mat_params(float metallicness, float opacity, float ir) {   // quads: derived values follow in mat_derive()
    MatP m; m.metallicness = metallicness; m.opacity = opacity; m.ir = ir;
    m.inv_ir = m.r0_front = m.r0_back = 0.0f;
    return m;
}
// ... once per query, for the quad that won (the values are read by the dielectric branch only)
__device__ __forceinline__ void mat_derive(MatP &m) {
    if (m.opacity > 0.0f) { m.inv_ir = 1.0f / m.ir; m.r0_front = schlick_r0(m.inv_ir); m.r0_back = schlick_r0(m.ir); }
}
__device__ __forceinline__ MatP mat_params(const DevMat &d) {                                // ... or by the host (spheres)
    MatP m; m.metallicness = d.metallicness; m.opacity = d.opacity; m.ir = d.ir;
    m.inv_ir = d.inv_ir; m.r0_front = d.r0_front; m.r0_back = d.r0_back;
    return m;
}
// `ud` is unit(dir), computed by the caller (the sky of a missing lane needs the same expression: one copy for the wave).
// `flags`: RTW_FLAG_CPP_* select the C++ twin's dialect (generic build only; a compile-time 0 elsewhere).
__device__ __forceinline__ v3 on_hit(const MatP m, v3 point, v3 normal, v3 dir, v3 ud, Rng &rng, float &cos_theta, uint32_t flags, Cen *cn = nullptr) {
    const bool front = !(dot(dir, normal) > 0.0f);
    // Shared by both branches: unit(dir), and its mirror direction.  reflect(ud, -n) == reflect(ud, n)
    // bit for bit ((-n*2) * dot(ud,-n) == (n*2) * dot(ud,n): negation is exact and commutes with the
    // rounded products and sums), so the dielectric's reflect about the ray-facing normal is this too.
    const v3 refl = reflect(ud, normal);
    v3 next;
    if (m.opacity > 0.0f) {
        RTW_CEN(cn, CEN_DIELECTRIC);
        const v3 n = front ? normal : -normal;
        const float ratio = front ? m.inv_ir : m.ir;
        float ct = dot(-ud, n);
        if (ct > 1.0f) ct = 1.0f;
        const float st = sqrt_ieee(1.0f - ct * ct);
        const bool cannot_refract = ratio * st > 1.0f;
        const float rfl = reflectance(ct, front ? m.r0_front : m.r0_back);
        bool do_reflect = cannot_refract;
        // xi drawn only when refraction is possible; never by the C++ twin (Schlick term commented out, C++/headers/materials.h:106)
        if (!do_reflect && !(flags & RTW_FLAG_CPP_DIELECTRIC)) { RTW_CEN(cn, CEN_SCHLICK); do_reflect = rfl > rng_f32(rng); }
        next = do_reflect ? refl : refract(ud, n, ratio);
        cos_theta = 0.0f;
    } else if (flags & RTW_FLAG_CPP_DIFFUSE) {
        // C++/headers/materials.h:113-117: sc = uniform_scatter(h, r).direction * (1 - m); reflect = metallic(h, r);
        // direction = reflect.direction * m + sc -- with uniform_scatter = unit((point + normal + rand_unit) - point) and
        // metallic = unit(reflect(unit(d), n)) (C++/src/materials.cpp:4-13), then near_zero -> normal (C++/src/sphere.cpp:29-31)
        const v3 target = (point + normal) + random_unit_vec(rng, true);
        const v3 sdir = unit(target - point);
        const v3 mdir = unit(refl);
        next = mdir * m.metallicness + sdir * (1.0f - m.metallicness);
        cos_theta = (m.metallicness != 1.0f) ? dot(sdir, normal) : 0.0f;
        if (__builtin_fabsf(next.x) < 1e-8f && __builtin_fabsf(next.y) < 1e-8f && __builtin_fabsf(next.z) < 1e-8f) next = normal;
        return next;
    } else {
        RTW_CEN(cn, CEN_DIFFUSE);
        const v3 target = normal + random_unit_vec(rng, false, cn);      // drawn even for mirrors (materials.rs:142)
        const v3 sc = close_to_zero(target) ? normal : target;
        next = refl * m.metallicness + sc * (1.0f - m.metallicness);
        cos_theta = (m.metallicness != 1.0f) ? dot(sc, normal) : 0.0f;
    }
    if (close_to_zero(next)) next = front ? normal : normal * -1.0f;
    return next;
}

// Material::on_hit in two halves around the merged rejection loop (specialised builds: Rust dialect, gradient integrator).
// First half: everything up to the random unit vector.  A dielectric lane is complete after it (returns true, `next` is its direction); a
// diffuse / metallic lane gets its mirror direction in `next` and waits for its unit vector.
__device__ __forceinline__ bool on_hit_first(const MatP m, v3 normal, v3 dir, v3 ud, Rng &rng, v3 &next, bool &front, Cen *cn = nullptr) {
    front = !(dot(dir, normal) > 0.0f);
    const v3 refl = reflect(ud, normal);          // (shared by both branches: see on_hit)
    if (m.opacity > 0.0f) {
        RTW_CEN(cn, CEN_DIELECTRIC);
        const v3 n = front ? normal : -normal;
        const float ratio = front ? m.inv_ir : m.ir;
        float ct = dot(-ud, n);
        if (ct > 1.0f) ct = 1.0f;
        const float st = sqrt_ieee(1.0f - ct * ct);
        const bool cannot_refract = ratio * st > 1.0f;
        const float rfl = reflectance(ct, front ? m.r0_front : m.r0_back);
        bool do_reflect = cannot_refract;
        if (!do_reflect) { RTW_CEN(cn, CEN_SCHLICK); do_reflect = rfl > rng_f32(rng); }
        next = do_reflect ? refl : refract(ud, n, ratio);
        if (close_to_zero(next)) next = front ? normal : normal * -1.0f;
        return true;
    }
    RTW_CEN(cn, CEN_DIFFUSE);
    next = refl;
    return false;
}
// unit(p) for an accepted point of the rejection sampler, whose |p|^2 = `l2` the sampler has just computed (the same expression, x*x + y*y + z*z).
// The components are multiples of 2^-23 in [-1, 1): "every component in [2^-40, 2^40]" (unit()'s condition for the plain sequences) is "no component
// is zero" here, and l2 is in [2^-46, 1].
__device__ __forceinline__ v3 unit_of_ball_point(v3 p, float l2) {
    const float lo = fminf(fminf(__builtin_fabsf(p.x), __builtin_fabsf(p.y)), __builtin_fabsf(p.z));
    if (ballot64(!(lo > 0.0f)) == 0ull) {
        const float s = sqrt_plain(l2);
        const float r = rcp_refined(s);
        return mk(div_plain(p.x, s, r), div_plain(p.y, s, r), div_plain(p.z, s, r));
    }
    return p / __builtin_sqrtf(l2);
}
// Second half: `p` is the accepted point of the unit ball (sample_ball_or_disk) and `l2` its squared length, `refl` the mirror direction of the first half.
__device__ __forceinline__ v3 on_hit_second(float metallicness, v3 normal, v3 refl, bool front, v3 p, float l2) {
    // random_unit_vec = unit(accepted point) (vec3.rs:238).  Handing the sampler's |p|^2 to the normalisation saves 7 VALU per SHADE step and costs more
    // than that: the value is one more live register across the loop, and at 72 VGPRs the allocator then spills the traversal's tau -- a scratch
    // load in every LEAF step (C4 -0.9 %, bench frame +-0; profiles/r03_ab_l2pass.log).  Off.
#ifdef RTW_UNIT_FROM_L2
    const v3 target = normal + unit_of_ball_point(p, l2);
#else
    (void)l2;
    const v3 target = normal + unit(p);
#endif
    const v3 sc = close_to_zero(target) ? normal : target;
    v3 next = refl * metallicness + sc * (1.0f - metallicness);
    if (close_to_zero(next)) next = front ? normal : normal * -1.0f;
    return next;
}

// Rust2's Material trait objects (Rust2/src/objects/material.rs): MirrorGlass :130-162 (the Rust
// dielectric's arithmetic), Mirror :75-83 (reflects the un-normalised direction), Lambertian :25-36
// (unit(n + random_unit_vec())).  No degenerate-direction fix-up in Rust2's ray_color.
__device__ __forceinline__ v3 on_hit_rust2(const MatP m, v3 normal, v3 dir, Rng &rng, uint32_t flags) {
    if (m.opacity > 0.0f) {
        const bool front = !(dot(dir, normal) > 0.0f);
        const v3 n = front ? normal : -normal;
        const float ratio = front ? m.inv_ir : m.ir;
        const v3 ud = unit(dir);
        float ct = dot(-ud, n);
        if (ct > 1.0f) ct = 1.0f;
        const float st = sqrt_ieee(1.0f - ct * ct);
        const bool cannot_refract = ratio * st > 1.0f;
        const float rfl = reflectance(ct, front ? m.r0_front : m.r0_back);
        bool do_reflect = cannot_refract;
        if (!do_reflect && !(flags & RTW_FLAG_CPP_DIELECTRIC)) do_reflect = rfl > rng_f32(rng);
        return do_reflect ? reflect(ud, n) : refract(ud, n, ratio);
    }
    if (m.metallicness == 1.0f) return reflect(dir, normal);
    return unit(normal + random_unit_vec(rng));
}

// ---- quads and instances ----------------------------------------------------------
// What a closest-hit query reports when the winner is not a top-level sphere: `Hit` (objects.rs:16-23).
struct GeomHit {
    float t;
    v3 point, normal, cm;
    MatP m;
    v3 emitted;
};

// Vec3::rotated (vec3.rs:161-181) as written; q = {asin, acos, bsin, bcos, csin, ccos}, k = rotation_coefficients(q)
__device__ __forceinline__ v3 rotated(v3 a, const float *q, const float *k) {
    const float as = q[0], ac = q[1], bs = q[2], bc = q[3], cs = q[4], cc = q[5];
    v3 o;
    o.x = a.x * bc * cc + a.y * k[0] + a.z * k[1];
    o.y = a.x * bc * cs + a.y * k[2] + a.z * k[3];
    o.z = a.x * -bs + a.y * as * bc + a.z * ac * bc;
    return o;
}

// ln(x), x a positive normal f32: f64 atanh series, rounded once.  The same operations as oracle/rtw_oracle.c
// ln_f32 (f64 add/mul/div are IEEE on both sides, no contraction), correctly rounded for every xi = k 2^-24.
__device__ __forceinline__ float ln_f32(float xf) {
    if (xf == 0.0f) return -__builtin_inff();
    const uint32_t bits = __builtin_bit_cast(uint32_t, xf);
    int e = (int)(bits >> 23) - 127;
    const float mf = __builtin_bit_cast(float, (bits & 0x007FFFFFu) | 0x3F800000u);
    double m = (double)mf;
    if (mf > 1.41421354f) { m = m * 0.5; e += 1; }
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s;
    double p = 1.0 / 25.0;
    p = 1.0 / 23.0 + z * p; p = 1.0 / 21.0 + z * p; p = 1.0 / 19.0 + z * p; p = 1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p; p = 1.0 / 13.0 + z * p; p = 1.0 / 11.0 + z * p; p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;  p = 1.0 / 5.0 + z * p;  p = 1.0 / 3.0 + z * p;
    const double lm = 2.0 * s + (2.0 * s) * (z * p);
    return (float)((double)e * 0.6931471805599453094 + lm);
}

#ifdef RTW_GEOM_EAGER_RECORDS
// Quad::collision_normal (quad.rs:37-81) against quad `qi` of `quads`; on Some(hit) that is strictly closer
// than the current one (`min_hit == None || min_hit > i`) it replaces h.
__device__ __forceinline__ void quad_test(const DevScene &sc, const DevQuad *quads, uint32_t qi, v3 o, v3 d,
                                          float mint, float maxt, bool &found, GeomHit &h) {
    cf4_ptr q = (cf4_ptr)(uintptr_t)(quads + qi);
    const f4 r0 = q[0], r3 = q[3];
    const v3 normal = mk(r3.x, r3.y, r3.z);
    const float denominator = dot(normal, d);
    if (__builtin_fabsf(denominator) <= 1e-8f) return;
    // (Measured and rejected, round 3: the same correctly rounded quotient through div_plain when every lane's operands are ordinary saves 6 of the
    //  division's 12 instructions and costs a range check, a ballot and a second code path per quad test: presentation_image 183.5 instead of 170.3 ms,
    //  images identical -- profiles/r03_ab_quad_div.log.  -DRTW_QUAD_PLAIN_DIV builds it.)
    const float numerator = r0.w - dot(normal, o);
    float t;
#ifdef RTW_QUAD_PLAIN_DIV
    const bool plain = in_range(denominator, 0x1p-40f, 0x1p40f) && (numerator == 0.0f || in_range(numerator, 0x1p-60f, 0x1p40f));
    if (ballot64(!plain) == 0ull) t = div_plain(numerator, denominator, rcp_refined(denominator));
    else
#endif
        t = numerator / denominator;
    if (t < mint || t > maxt) return;
    if (found && !(h.t > t)) return;                           // cannot replace the current hit (`min_hit > i` is strict): skip the interior test
    const f4 r1 = q[1], r2 = q[2], r4 = q[4];
    const v3 point = o + d * t;
    const v3 planar = point - mk(r0.x, r0.y, r0.z);
    const v3 qu = mk(r1.x, r1.y, r1.z), qv = mk(r2.x, r2.y, r2.z), w = mk(r4.x, r4.y, r4.z);
    const v3 pxv = mk(planar.y * qv.z - planar.z * qv.y, planar.z * qv.x - planar.x * qv.z, planar.x * qv.y - planar.y * qv.x);
    const v3 uxp = mk(qu.y * planar.z - qu.z * planar.y, qu.z * planar.x - qu.x * planar.z, qu.x * planar.y - qu.y * planar.x);
    const float alfa = dot(w, pxv), beta = dot(w, uxp);
    if (alfa < 0.0f || alfa > 1.0f || beta < 0.0f || beta > 1.0f) return;
    const f4 r5 = q[5], r6 = q[6];
    // (a scalar copy first: __builtin_bit_cast applied directly to an ext-vector ELEMENT reads element 0 with this
    //  toolchain -- seen in the ISA as the texture id compared against w.x)
    const float tex_bits = r4.w;
    const int32_t tex = __float_as_int(tex_bits);
    v3 cm = mk(r5.x, r5.y, r5.z);
    if (tex >= 0) {                                            // quad.rs:64-79
        const RtwTexture tx = sc.tex[tex];
        const uint32_t ix = alfa != 1.0f ? tex_index(floorf(alfa * (float)tx.row), tx.row - 1) : tx.row - 1;
        const uint32_t iy = beta != 1.0f ? tex_index(floorf(beta * (float)tx.col), tx.col - 1) : tx.col - 1;
        cm = ld3(sc.texels + 3 * (size_t)(tx.texel_offset + iy * tx.row + ix)) * 1.0f;
    }
    found = true;
    h.t = t; h.point = point; h.normal = normal; h.cm = cm;
    h.m = mat_params(r1.w, r2.w, r3.w);
    h.emitted = mk(r6.x, r6.y, r6.z);
}

// Closest member of an instance: its spheres in list order, then its quads (instance.rs:263-273).
__device__ __forceinline__ bool instance_members(const DevScene &sc, const DevGeom &g, const DevInstance &in, v3 o, v3 d, float tm,
                                                 float mint, float maxt, GeomHit &h, uint32_t &n_sph, uint32_t &n_quad) {
    bool found = false;
    int best = -1; float best_t = 0.0f;
    const float a = dot(d, d);
    cf4_ptr geom = (cf4_ptr)(uintptr_t)g.igeom, vel = (cf4_ptr)(uintptr_t)g.ivel;
    for (uint32_t k = 0; k < in.n_spheres; ++k) {              // sphere.rs:99-147, as closest_brute
        const uint32_t s = in.first_sphere + k;
        const f4 gg = geom[s], vv = vel[s];
        const float cx = gg.x + vv.x * tm, cy = gg.y + vv.y * tm, cz = gg.z + vv.z * tm;
        const float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
        const float b = ocx * d.x + ocy * d.y + ocz * d.z;
        const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - gg.w;
        const float disc = b * b - a * c;
        if (!(disc < 0.0f)) {
            const float sq = __builtin_sqrtf(disc);
            float x = (-b - sq) / a;
            if (x < mint) x = (-b + sq) / a;
            if (!(x < mint || x > maxt)) {
                if (best < 0 || best_t > x) { best = (int)s; best_t = x; }
            }
        }
    }
    n_sph += in.n_spheres;
    if (best >= 0) {
        const f4 gg = g.igeom[best], vv = g.ivel[best];
        const v3 c = mk(gg.x, gg.y, gg.z) + mk(vv.x, vv.y, vv.z) * tm;
        const DevMat mat = g.imat[best];
        found = true;
        h.t = best_t; h.point = o + d * best_t; h.normal = unit(h.point - c);
        h.cm = sphere_albedo(sc, mat, h.normal);
        h.m = mat_params(mat);
        h.emitted = ld3(mat.emitted);
    }
    for (uint32_t k = 0; k < in.n_quads; ++k) quad_test(sc, g.iquads, in.first_quad + k, o, d, mint, maxt, found, h);
    n_quad += in.n_quads;
    return found;
}

// The part of Scene::collision_normal (viewport.rs:136-150) after the top-level spheres: quads, then instances.
// `found`/`h.t` enter as the sphere result (h.t = its t); returns true when a quad or an instance wins.
__device__ __forceinline__ bool geom_closest(const DevScene &sc, const DevGeom &g, v3 o, v3 d, float tm, float mint, float maxt,
                                             bool sphere_found, float sphere_t, Rng &rng, GeomHit &h, uint32_t &n_sph, uint32_t &n_quad) {
    bool found = sphere_found, won = false;
    h.t = sphere_t;
    {   // q_hit: closest quad first, then compared with s_hit
        GeomHit qh; bool qfound = false;
        for (uint32_t k = 0; k < g.n_quads; ++k) quad_test(sc, g.quads, k, o, d, mint, maxt, qfound, qh);
        n_quad += g.n_quads;
        if (qfound && (!found || h.t > qh.t)) { h = qh; found = true; won = true; }
    }
    GeomHit ih; bool ifound = false;
    for (uint32_t i = 0; i < g.n_inst; ++i) {                 // Instance::collision_normal (instance.rs:250-310)
        const DevInstance in = g.inst[i];
        const v3 tr = ld3(in.tr);
        const v3 lo = rotated(o - tr, in.back, in.back_k), ld = rotated(d, in.back, in.back_k);
        GeomHit c;
        if (!instance_members(sc, g, in, lo, ld, tm, mint, maxt, c, n_sph, n_quad)) continue;
        if (in.medium == RTW_MEDIUM_CONST_DENSITY) {           // const_density (:24-26)
            const float distance = ln_f32(rng_f32(rng)) / -in.density;
            if (distance >= 0.0f) {
                const v3 o2 = c.point + ld * distance;
                GeomHit second;
                if (!instance_members(sc, g, in, o2, ld, tm, mint, maxt, second, n_sph, n_quad)) continue;   // left the volume first
                c.point = o2;
                c.normal = random_unit_vec(rng);
            }
        }
        c.point = rotated(c.point, in.fwd, in.fwd_k) + tr;
        c.normal = rotated(c.normal, in.fwd, in.fwd_k);
        if (!ifound || ih.t > c.t) { ih = c; ifound = true; }
    }
    if (ifound && (!found || h.t > ih.t)) { h = ih; won = true; }
    return won;
}

#else
// ---- the closest-hit walk over quads and instances with the `Hit` record formed ONCE, for the winner -------------------------------------
// The reference fills a whole `Hit` (point, normal, texel, material, emission: 17 words) for every candidate that replaces the current
// one; what the comparisons read is `t` alone, and `t` is the same number in an instance's local frame and in the world's.  So the walk
// keeps (t, which member) and the record is rebuilt afterwards with the same operations on the same operands -- bit for bit what the
// eager form produces (RTW_GEOM_EAGER_RECORDS builds that form; presentation_image A/B in profiles/r03_ab_geom_records.log).

// The tests of Quad::collision_normal (quad.rs:37-63) against quad `qi`: true when it is a Some(hit) that replaces the current one
// (`min_hit == None || min_hit > i`, strict); t_out = its t.
__device__ __forceinline__ bool quad_pick(const DevQuad *quads, uint32_t qi, v3 o, v3 d, float mint, float maxt, bool found, float cur_t, float &t_out) {
    cf4_ptr q = (cf4_ptr)(uintptr_t)(quads + qi);
    const f4 r0 = q[0], r3 = q[3];
    const v3 normal = mk(r3.x, r3.y, r3.z);
    const float denominator = dot(normal, d);
    if (__builtin_fabsf(denominator) <= 1e-8f) return false;
    const float numerator = r0.w - dot(normal, o);
    const float t = numerator / denominator;
    if (t < mint || t > maxt) return false;
    if (found && !(cur_t > t)) return false;                   // cannot replace the current hit: skip the interior test
    const f4 r1 = q[1], r2 = q[2], r4 = q[4];
    const v3 point = o + d * t;
    const v3 planar = point - mk(r0.x, r0.y, r0.z);
    const v3 qu = mk(r1.x, r1.y, r1.z), qv = mk(r2.x, r2.y, r2.z), w = mk(r4.x, r4.y, r4.z);
    const v3 pxv = mk(planar.y * qv.z - planar.z * qv.y, planar.z * qv.x - planar.x * qv.z, planar.x * qv.y - planar.y * qv.x);
    const v3 uxp = mk(qu.y * planar.z - qu.z * planar.y, qu.z * planar.x - qu.x * planar.z, qu.x * planar.y - qu.y * planar.x);
    const float alfa = dot(w, pxv), beta = dot(w, uxp);
    if (alfa < 0.0f || alfa > 1.0f || beta < 0.0f || beta > 1.0f) return false;
    t_out = t;
    return true;
}

// The `Hit` of quad `qi` at parameter t (quad.rs:64-81); qi is per lane here (vector loads).
__device__ __forceinline__ void quad_record(const DevScene &sc, const DevQuad *quads, uint32_t qi, v3 o, v3 d, float t, GeomHit &h) {
    const DevQuad &q = quads[qi];
    const v3 point = o + d * t;
    v3 cm = ld3(q.albedo);
    const int32_t tex = q.tex;
    if (tex >= 0) {                                            // quad.rs:64-79
        const v3 planar = point - ld3(q.origin);
        const v3 qu = ld3(q.u), qv = ld3(q.v), w = ld3(q.w);
        const v3 pxv = mk(planar.y * qv.z - planar.z * qv.y, planar.z * qv.x - planar.x * qv.z, planar.x * qv.y - planar.y * qv.x);
        const v3 uxp = mk(qu.y * planar.z - qu.z * planar.y, qu.z * planar.x - qu.x * planar.z, qu.x * planar.y - qu.y * planar.x);
        const float alfa = dot(w, pxv), beta = dot(w, uxp);
        const RtwTexture tx = sc.tex[tex];
        const uint32_t ix = alfa != 1.0f ? tex_index(floorf(alfa * (float)tx.row), tx.row - 1) : tx.row - 1;
        const uint32_t iy = beta != 1.0f ? tex_index(floorf(beta * (float)tx.col), tx.col - 1) : tx.col - 1;
        cm = ld3(sc.texels + 3 * (size_t)(tx.texel_offset + iy * tx.row + ix)) * 1.0f;
    }
    h.t = t; h.point = point; h.normal = ld3(q.normal); h.cm = cm;
    h.m = mat_params(q.metallicness, q.opacity, q.ir);
    h.emitted = ld3(q.emitted);
}

// Closest member of an instance: its spheres in list order, then its quads (instance.rs:263-273).  code >= 0: member sphere `code`;
// code < 0: member quad ~code.
__device__ __forceinline__ bool instance_pick(const DevGeom &g, const DevInstance &in, v3 o, v3 d, float tm, float mint, float maxt,
                                              float &t_out, int &code, uint32_t &n_sph, uint32_t &n_quad) {
    int best = -1; float best_t = 0.0f;
    const float a = dot(d, d);
    cf4_ptr geom = (cf4_ptr)(uintptr_t)g.igeom, vel = (cf4_ptr)(uintptr_t)g.ivel;
    for (uint32_t k = 0; k < in.n_spheres; ++k) {              // sphere.rs:99-147, as closest_brute
        const uint32_t s = in.first_sphere + k;
        const f4 gg = geom[s], vv = vel[s];
        const float cx = gg.x + vv.x * tm, cy = gg.y + vv.y * tm, cz = gg.z + vv.z * tm;
        const float ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
        const float b = ocx * d.x + ocy * d.y + ocz * d.z;
        const float c = (ocx * ocx + ocy * ocy + ocz * ocz) - gg.w;
        const float disc = b * b - a * c;
        if (!(disc < 0.0f)) {
            const float sq = __builtin_sqrtf(disc);
            float x = (-b - sq) / a;
            if (x < mint) x = (-b + sq) / a;
            if (!(x < mint || x > maxt)) {
                if (best < 0 || best_t > x) { best = (int)s; best_t = x; }
            }
        }
    }
    n_sph += in.n_spheres;
    bool found = best >= 0;
    for (uint32_t k = 0; k < in.n_quads; ++k) {
        float t;
        if (quad_pick(g.iquads, in.first_quad + k, o, d, mint, maxt, found, best_t, t)) { found = true; best_t = t; best = ~(int)(in.first_quad + k); }
    }
    n_quad += in.n_quads;
    t_out = best_t; code = best;
    return found;
}

// The `Hit` of an instance's member `code` at parameter t, in the instance's frame.
__device__ __forceinline__ void member_record(const DevScene &sc, const DevGeom &g, int code, v3 o, v3 d, float tm, float t, GeomHit &h) {
    if (code >= 0) {
        const f4 gg = g.igeom[code], vv = g.ivel[code];
        const v3 c = mk(gg.x, gg.y, gg.z) + mk(vv.x, vv.y, vv.z) * tm;
        const DevMat mat = g.imat[code];
        h.t = t; h.point = o + d * t; h.normal = unit(h.point - c);
        h.cm = sphere_albedo(sc, mat, h.normal);
        h.m = mat_params(mat);
        h.emitted = ld3(mat.emitted);
    } else {
        quad_record(sc, g.iquads, (uint32_t)~code, o, d, t, h);
    }
}

// The part of Scene::collision_normal (viewport.rs:136-150) after the top-level spheres: quads, then instances.
// `sphere_found` / `sphere_t` are the sphere result; returns true when a quad or an instance wins, h = its Hit.
__device__ __forceinline__ bool geom_closest(const DevScene &sc, const DevGeom &g, v3 o, v3 d, float tm, float mint, float maxt,
                                             bool sphere_found, float sphere_t, Rng &rng, GeomHit &h, uint32_t &n_sph, uint32_t &n_quad) {
    bool found = sphere_found;
    float ht = sphere_t;
    int win = 0;                                               // 1: top-level quad qk, 2: instance ii
    uint32_t qk = 0;
    {   // q_hit: closest quad first, then compared with s_hit
        bool qfound = false; float qt = 0.0f;
        for (uint32_t k = 0; k < g.n_quads; ++k) {
            float t;
            if (quad_pick(g.quads, k, o, d, mint, maxt, qfound, qt, t)) { qfound = true; qt = t; qk = k; }
        }
        n_quad += g.n_quads;
        if (qfound && (!found || ht > qt)) { ht = qt; found = true; win = 1; }
    }
    bool ifound = false, imed = false;
    float it = 0.0f; uint32_t ii = 0; int icode = 0;
    v3 imp = mk(0, 0, 0), imn = mk(0, 0, 0);                   // the medium's scatter point (instance frame) and direction
    for (uint32_t i = 0; i < g.n_inst; ++i) {                 // Instance::collision_normal (instance.rs:250-310)
        const DevInstance in = g.inst[i];
        const v3 tr = ld3(in.tr);
        const v3 lo = rotated(o - tr, in.back, in.back_k), ld = rotated(d, in.back, in.back_k);
        float ct; int code;
        if (!instance_pick(g, in, lo, ld, tm, mint, maxt, ct, code, n_sph, n_quad)) continue;
        bool med = false; v3 mp = mk(0, 0, 0), mn = mk(0, 0, 0);
        if (in.medium == RTW_MEDIUM_CONST_DENSITY) {           // const_density (:24-26)
            const float distance = ln_f32(rng_f32(rng)) / -in.density;
            if (distance >= 0.0f) {
                const v3 o2 = (lo + ld * ct) + ld * distance;
                float t2; int c2;
                if (!instance_pick(g, in, o2, ld, tm, mint, maxt, t2, c2, n_sph, n_quad)) continue;   // left the volume first
                med = true; mp = o2; mn = random_unit_vec(rng);
            }
        }
        if (!ifound || it > ct) { it = ct; ii = i; icode = code; imed = med; imp = mp; imn = mn; ifound = true; }
    }
    if (ifound && (!found || ht > it)) win = 2;
    if (win == 1) {
        quad_record(sc, g.quads, qk, o, d, ht, h);
    } else if (win == 2) {
        const DevInstance &in = g.inst[ii];
        const v3 tr = ld3(in.tr);
        const v3 lo = rotated(o - tr, in.back, in.back_k), ld = rotated(d, in.back, in.back_k);
        member_record(sc, g, icode, lo, ld, tm, it, h);
        if (imed) { h.point = imp; h.normal = imn; }
        h.point = rotated(h.point, in.fwd, in.fwd_k) + tr;
        h.normal = rotated(h.normal, in.fwd, in.fwd_k);
    }
    return win != 0;
}
#endif

// ray_color.rs:38-40
__device__ __forceinline__ v3 sky_gradient(v3 ud) {      // ud = unit(direction)
    float t = 0.5f * (ud.y + 1.0f);
    return mk((1.0f - t) + t * 0.5f, (1.0f - t) + t * 0.7f, 1.0f);
}

} // namespace rtw
