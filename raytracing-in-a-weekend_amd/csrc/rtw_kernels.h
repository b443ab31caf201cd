// rtw_kernels.h -- kernel argument block shared by rtw_kernels.hip (device) and rtw_shim.hip (host).
#pragma once
#include "rtw_device.h"
#include "rtw_host.h"

#define RTW_QUEUE_BYTES 4096u   // the work queue's counters (KArgs.queue): up to 8 sub-queues ...
#define RTW_QUEUE_STRIDE 256u   // ... bytes apart
#ifndef RTW_SUB_SHIFT
#define RTW_SUB_SHIFT 3u        // log2 of the number of sub-queues (8: one per XCD; 16 and 32 measured no better, profiles/r02_subq_count.log)
#endif
#ifndef RTW_BLOCK
#define RTW_BLOCK 256   // 4 waves per workgroup
#endif
#define RTW_N_STATS 64  // 64-bit counters a render launch accumulates (KArgs.stats); [32..63] are used by the -DRTW_CENSUS diagnostic build only
#ifndef RTW_LIST_WALK_MAX_DEFAULT
#define RTW_LIST_WALK_MAX_DEFAULT 48u  // RTW_OPT_LIST_WALK_MAX: scenes this small walk the list even when the BVH is asked for (measured crossover ~56 spheres: profiles/r02_crossover.log)
#endif

namespace rtw {

// Device view of the acceleration structure built by build_bvh() (rtw_host.cpp).
struct DevBvh {
    const BvhNode *nodes;
    const BvhNode16 *nodes16;     // f16 copy for the LDS-resident variant, or null
    uint32_t n_nodes;
    const f4 *big_geom;           // {cx, cy, cz, r*r} of the spheres kept outside the tree
    const f4 *big_vel;
    const uint32_t *big_index;    // their indices in the scene list
    uint32_t n_big;
    uint32_t depth;               // tree depth (the stack needs depth + 3 levels: sentinel, one per level, the slot above the top)
    int32_t root;                 // node index; ~sphere when the tree is a single leaf; INT32_MIN when empty
    float cx, cy, cz;             // centre C of the tree spheres' centres
    float centre_radius;          // R_c
    float r_max2;                 // r_max^2
    float inv_2rmin;              // 1 / (2 r_min), +inf when r_min == 0
    float abs_max;                // largest |coordinate| of the root box
};

struct KArgs {
    RtwCamera cam;
    DevScene  sc;
    DevBvh    bvh;
    DevGeom   geom;               // quads and instances (n_quads == n_inst == 0 for sphere-only scenes)
    uint32_t width, height;       // full image
    uint32_t k_base, k_end;       // compact rows [k_base, k_end) of this partition rendered by this launch (a band)
    uint32_t n_tiles;             // tiles_x * ceil((k_end - k_base) / 8)
    uint32_t chunk_len, n_chunks; // samples per work unit and units per pixel of the queue's FIRST region (all of it with RTW_FLAG_CHUNK_SUMS)
    uint32_t bank_len;            // 0: the bank holds one slot per sample, [tile][sample][pixel] -- independent of how the samples are cut into units;
                                  // 1 (RTW_FLAG_CHUNK_SUMS): one slot per unit, [tile][chunk][pixel]
    // The unit length is GUIDED within a launch: the tiles at queue positions [0, reg_q1) are cut into units of reg_len[0] samples, [reg_q1, reg_q2) into
    // units of reg_len[1], the rest -- the end of the queue -- into units of reg_len[2] (reg_nc[r] = ceil(n_samples / reg_len[r]) units per pixel):
    // long units amortise the per-unit work while plenty is left, short ones keep the drain of the launch short (rtw_shim.hip)
    uint32_t reg_q1, reg_q2;
    uint32_t reg_len[3], reg_nc[3];
    uint32_t flags;               // RtwParams.flags (RTW_FLAG_CPP_*: generic build only)
    float *samples;               // per-sample radiance, [n_tiles][n_samples][64][3]  (RTW_FLAG_CHUNK_SUMS: [n_tiles][n_chunks][64][3])
    uint32_t row_block, part_index, part_count;
    uint32_t tiles_x;             // ceil(width / 8)
    uint32_t total_work;          // work items of the launch: 64 per (tile, unit)
    uint32_t sub_shift;           // the work queue is 2^sub_shift sub-queues (tiles dealt round-robin in queue order), counters RTW_QUEUE_STRIDE bytes apart
    uint32_t grab_shift, grab_max;   // a wave takes min(grab_max, (work left >> grab_shift) rounded down to whole blocks, at least one block) items per queue atomic
    const uint32_t *tile_order;   // queue position -> tile (a permutation of [0, n_tiles)), or null = raster order
    uint32_t n_samples;           // rays per pixel actually traced (sampler-dependent)
    uint32_t s_root;              // strata per axis (STRATIFIED / CENTRES)
    uint32_t sampler, integrator, depth;
    uint32_t has_textures;        // any sphere with an image texture (selects the generic kernel)
    uint32_t lds_bytes;           // dynamic LDS of the BVH kernel: f16 nodes | stack | sphere geometry
    uint32_t lds_stack_off, lds_geom_off;   // byte offsets (16-aligned) of the per-lane stack and of the sphere geometry (0: geometry stays in
                                            // global memory); the f16 nodes sit at offset 0
    uint32_t seed_lo, seed_hi;
    float inv_gamma, mint, maxt;
    float bg[3];
    float *out;                   // [rows of the partition][width][3]
    uint32_t *queue;              // work-item counter, zeroed before launch
    unsigned long long endtimes_ref;   // -DRTW_ENDTIMES builds: reference wave lifetime for the histogram (0 = none)
    unsigned long long *stats;    // [0] camera rays [1] segments [2] sphere tests [3] node tests [4] nan pixels [5..7] phase steps [8..10] phase lanes [14] quad tests [16..19] steps, lanes of phases 3 (switch), 4 (new path)
};

// accel: RTW_ACCEL_BRUTE, RTW_ACCEL_BVH; the BVH launch picks the LDS-resident variant when a.bvh.nodes16 != null
void launch_render(const KArgs &a, bool moving, uint32_t accel, uint32_t grid, hipStream_t stream);
// Resident workgroups per CU for the kernel variant (occupancy API), >= 1.
uint32_t kernel_blocks_per_cu(const KArgs &a, bool moving, uint32_t accel, bool lds_nodes);
// Is there a build of the BVH kernel for this configuration that reads the spheres' {centre, r^2} from LDS?  (KArgs.lds_geom_off may only be set then)
bool kernel_has_lds_geom(const KArgs &a);
// The kernel variant launch_render would pick, as an opaque id (key of the per-context occupancy cache).
const void *kernel_id(const KArgs &a, bool moving, uint32_t accel, bool lds_nodes);

} // namespace rtw
