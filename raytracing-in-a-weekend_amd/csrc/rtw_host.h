// rtw_host.h -- host-side C++ mirror of the reference constructors (Viewport::new, Sphere::new,
// Scene::new_sphere's acceleration build) and the BASELINE scene generators.  Pure host code.
#pragma once
#include <cstdint>
#include <vector>
#include "rtw.h"

namespace rtw {

// f32 helpers: this library is built with -ffp-contract=off, so these are single IEEE operations.
static inline float host_mul(float a, float b) { return a * b; }
static inline float host_div(float a, float b) { return a / b; }

// Rays per pixel a sampler traces for Viewport.samples (viewport.rs:443; Rust2 viewport.rs:90).
uint32_t sampler_count(uint32_t sampler, uint32_t samples, uint32_t *s_root);

// ---- acceleration structure ---------------------------------------------------------------------
// Binary BVH, single-sphere leaves, both child boxes stored in the parent (one 64-byte fetch per
// visit = two slab tests).  child >= 0: inner node index; child < 0: leaf holding sphere ~child.
struct BvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t c0, c1;
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// Compact node for the LDS-resident variant: the same two child boxes as f16, rounded OUTWARD (lo down,
// hi up: the tree only prunes, so larger boxes stay conservative), 16-bit child ids.  32 bytes.
// One dword per (box, axis) holding {lo, hi}: the kernel swaps the halves with ONE v_perm_b32 when the ray
// runs against the axis and has its {near plane, far plane} without a min / max pair.
struct BvhNode16 {
    uint16_t plane[2][3][2];   // [child box][axis]{lo, hi}
    int16_t c0, c1;
    uint32_t pad;
};
static_assert(sizeof(BvhNode16) == 32, "BvhNode16 must be 32 bytes");
#define RTW_LDS_NODES_MAX 512 // inner nodes the LDS variant holds (16 KB)
#define RTW_LDS_GEOM_MAX 640  // spheres whose {centre, r^2} the LDS variant also keeps on chip for the leaf tests (10 KB)

#define RTW_MAX_BIG 16        // spheres far larger than the rest are tested exactly, outside the tree
#define RTW_BVH_STACK 24      // builder guarantees depth <= RTW_BVH_STACK (median-split fallback near the limit)

struct BvhBuild {
    std::vector<BvhNode> nodes;          // nodes[0] is the root (absent when < 2 tree spheres)
    std::vector<BvhNode16> nodes16;      // f16 copy, filled only when it is usable (see build_bvh)
    std::vector<uint32_t> big;           // sphere indices tested by the uniform pre-pass
    int32_t root;                        // node index, or ~sphere for a single tree sphere, or INT32_MIN if empty
    // per-ray padding constants (DESIGN.md "Conservative traversal"): over the TREE spheres only
    float centre[3];                     // C: centre of the box of sphere centres
    float centre_radius;                 // R_c: max |c_s - C| (time-expanded)
    float r_min, r_max;                  // radius range of the tree spheres
    float abs_max;                       // largest |coordinate| of any tree box
    uint32_t depth;
};

// Bounds cover centre(t) = origin + velocity * t for t in [t_begin, t_end] (sphere.rs:100); the
// reference's own AABB ignores velocity (aabb/aabb.rs:27-39) and so culls moving spheres wrongly.
void build_bvh(const RtwSphere *spheres, uint32_t n, float t_begin, float t_end, BvhBuild &out);

// ---- queue order of the 8x8 tiles (RTW_OPT_TILE_ORDER) -------------------------------------------------------------------
// What the ordering heuristic may know about the scene: the spheres kept outside the tree (the ground) and the root box of
// the tree's spheres.  n_other != 0 (quads / instances present) switches the cheap-tile guess off.
struct SceneCull {
    float big[16][4];            // centre, radius
    uint32_t n_big = 0;
    uint32_t has_tree = 0;
    float lo[3] = { 0, 0, 0 }, hi[3] = { 0, 0, 0 };   // root box of the tree (time-expanded)
    uint32_t n_other = 0;
};
// fills has_tree / lo / hi from a built tree
void scene_cull_from_bvh(const BvhBuild &bb, const RtwSphere *spheres, SceneCull &out);
struct TileOrderKey {            // everything the order depends on (compared bytewise)
    uint32_t mode, tiles_x, tiles_y, k_base, row_block, part_index, part_count, scene_serial, tail_tiles;
    RtwCamera cam;
};
// mode 1: groups of 8 consecutive tiles scattered over the frame by a multiplicative bijection (waves then work on a mix of cheap and
//         expensive image regions at any time instead of all on the same band);
// mode 2: longest-processing-time-first by an estimated cost per tile (class of the tile's centre ray -- sphere field, ground only,
//         sky -- then distance, nearer first): the launch ends on its cheapest tiles;
// mode 3: reverse raster (bottom rows first).
// `order` receives a permutation of [0, tiles_x * tiles_y).  The image never depends on it.
void build_tile_order(uint32_t mode, uint32_t tiles_x, uint32_t tiles_y, uint32_t k_base, uint32_t row_block, uint32_t part_index,
                      uint32_t part_count, const RtwCamera &cam, const SceneCull &cull, uint32_t tail_tiles, std::vector<uint32_t> &order);

} // namespace rtw
