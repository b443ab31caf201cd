// wave_sim.cpp -- a design tool, not product: replays the BVH kernels' in-wave phase scheduler on the CPU.
//
// It traces the bench frame's paths with a plain f32 path tracer over the product's own BVH (build_bvh, rtw_host.cpp),
// records for every closest-hit query the ORDER of its inner-node visits and leaf tests, and then simulates ONE wavefront
// pulling (tile, chunk) blocks exactly like fetch_pixel() does, under different scheduling policies.  Output: SIMD
// efficiency per phase and relative time per segment -- the quantities the kernel's census (RtwStats.phase_*) measures on the
// GPU -- so that policies can be compared before they are written in HIP.  Costs per wave-step are parameters (cycles).
//
//   build:  /opt/rocm/lib/llvm/bin/clang++ -O2 -std=c++17 -I../../include -I../../raytracing-in-a-weekend_amd/csrc
//           wave_sim.cpp ../../raytracing-in-a-weekend_amd/csrc/rtw_host.cpp -o wave_sim
#include "rtw_host.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
using namespace rtw;

struct V { float x, y, z; };
static V operator+(V a, V b) { return { a.x + b.x, a.y + b.y, a.z + b.z }; }
static V operator-(V a, V b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
static V operator*(V a, float s) { return { a.x * s, a.y * s, a.z * s }; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V unit(V a) { float l = std::sqrt(dot(a, a)); return a * (1.0f / l); }

struct Scene { std::vector<RtwSphere> sp; BvhBuild bvh; };
static std::mt19937 g_rng(12345);
static float rnd() { return (g_rng() >> 8) * (1.0f / 16777216.0f); }
static V rand_unit() { for (;;) { V p = { 2 * rnd() - 1, 2 * rnd() - 1, 2 * rnd() - 1 }; if (dot(p, p) <= 1.0f) return unit(p); } }

static bool hit_sphere(const RtwSphere &s, V o, V d, float mint, float maxt, float &t) {
    V oc = o - V{ s.center[0], s.center[1], s.center[2] };
    float a = dot(d, d), b = dot(oc, d), c = dot(oc, oc) - s.radius * s.radius, disc = b * b - a * c;
    if (disc < 0) return false;
    float sq = std::sqrt(disc), x = (-b - sq) / a;
    if (x < mint) x = (-b + sq) / a;
    if (x < mint || x > maxt) return false;
    t = x; return true;
}
static bool hit_box(const float *lo, const float *hi, V o, V inv, float tmin, float tmax, float &entry) {
    float t0 = tmin, t1 = tmax;
    const float oo[3] = { o.x, o.y, o.z }, ii[3] = { inv.x, inv.y, inv.z };
    for (int k = 0; k < 3; k++) {
        float a = (lo[k] - oo[k]) * ii[k], b = (hi[k] - oo[k]) * ii[k];
        if (a > b) std::swap(a, b);
        t0 = std::max(t0, a); t1 = std::min(t1, b);
    }
    entry = t0;
    return t0 <= t1;
}

// One closest-hit query; `ops` receives 'N' per inner-node visit and 'L' per leaf test, in execution order.
static int closest(const Scene &S, V o, V d, float &best_t, std::string &ops) {
    const float mint = 0.001f;
    int best = -1; best_t = 1e5f;
    for (uint32_t i : S.bvh.big) { float t; if (hit_sphere(S.sp[i], o, d, mint, best_t, t) && t < best_t) { best_t = t; best = (int)i; } }
    if (S.bvh.root == std::numeric_limits<int32_t>::min()) return best;
    V inv = { 1.0f / d.x, 1.0f / d.y, 1.0f / d.z };
    int32_t stack[64]; int sp = 0; int32_t node = S.bvh.root;
    for (;;) {
        if (node >= 0) {
            ops.push_back('N');
            const BvhNode &n = S.bvh.nodes[node];
            float e0, e1;
            bool h0 = hit_box(n.lo0, n.hi0, o, inv, mint, best_t, e0), h1 = hit_box(n.lo1, n.hi1, o, inv, mint, best_t, e1);
            if (h0 && h1) { if (e0 <= e1) { stack[sp++] = n.c1; node = n.c0; } else { stack[sp++] = n.c0; node = n.c1; } continue; }
            if (h0) { node = n.c0; continue; }
            if (h1) { node = n.c1; continue; }
        } else {
            ops.push_back('L');
            uint32_t s = (uint32_t)~node; float t;
            if (hit_sphere(S.sp[s], o, d, mint, best_t, t) && t < best_t) { best_t = t; best = (int)s; }
        }
        if (sp == 0) break;
        node = stack[--sp];
    }
    return best;
}


// ---- uniform grid over the tree spheres (design study): cells hold sphere lists; a query = DDA steps ('N') + exact tests ('L') ----
struct Grid {
    float lo[3], inv_cell[3], cell[3]; int n[3];
    std::vector<std::vector<uint32_t>> cells; std::vector<float> ymax;
};
static Grid g_grid; static bool g_use_grid = false;
static void build_grid(const Scene &S, float target_per_cell) {
    Grid &G = g_grid;
    float lo[3] = { 1e30f, 1e30f, 1e30f }, hi[3] = { -1e30f, -1e30f, -1e30f };
    std::vector<char> big(S.sp.size(), 0); for (uint32_t i : S.bvh.big) big[i] = 1;
    size_t cnt = 0;
    for (size_t i = 0; i < S.sp.size(); i++) if (!big[i]) { cnt++; for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], S.sp[i].center[k] - S.sp[i].radius); hi[k] = std::max(hi[k], S.sp[i].center[k] + S.sp[i].radius); } }
    // cells ~ cube root rule on the two large axes; y gets 1 cell when the scene is flat
    float ext[3] = { hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2] };
    float vol = ext[0] * ext[1] * ext[2];
    float cs = std::cbrt(vol * target_per_cell / cnt);
    if (getenv("SIM_CELL")) cs = (float)atof(getenv("SIM_CELL"));          // cell edge given directly (all axes)
    for (int k = 0; k < 3; k++) { G.n[k] = std::max(1, (int)std::floor(ext[k] / cs)); G.cell[k] = ext[k] / G.n[k]; G.inv_cell[k] = 1.0f / G.cell[k]; G.lo[k] = lo[k]; }
    G.cells.assign((size_t)G.n[0] * G.n[1] * G.n[2], {});
    for (size_t i = 0; i < S.sp.size(); i++) if (!big[i]) {
        int a[3], b[3];
        for (int k = 0; k < 3; k++) {
            a[k] = std::max(0, std::min(G.n[k] - 1, (int)std::floor((S.sp[i].center[k] - S.sp[i].radius - lo[k]) * G.inv_cell[k])));
            b[k] = std::max(0, std::min(G.n[k] - 1, (int)std::floor((S.sp[i].center[k] + S.sp[i].radius - lo[k]) * G.inv_cell[k])));
        }
        for (int x = a[0]; x <= b[0]; x++) for (int y = a[1]; y <= b[1]; y++) for (int z = a[2]; z <= b[2]; z++) G.cells[((size_t)x * G.n[1] + y) * G.n[2] + z].push_back((uint32_t)i);
    }
    size_t refs = 0, nonempty = 0; for (auto &c : G.cells) { refs += c.size(); nonempty += !c.empty(); }
    printf("grid %d x %d x %d, cell %.2f x %.2f x %.2f, %zu sphere refs (%.2f per sphere), %zu / %zu cells non-empty\n", G.n[0], G.n[1], G.n[2], G.cell[0], G.cell[1], G.cell[2],
           refs, (double)refs / cnt, nonempty, G.cells.size());
}
static int closest_grid(const Scene &S, V o, V d, float &best_t, std::string &ops) {
    const Grid &G = g_grid;
    const float mint = 0.001f;
    int best = -1; best_t = 1e5f;
    for (uint32_t i : S.bvh.big) { float t; if (hit_sphere(S.sp[i], o, d, mint, best_t, t) && t < best_t) { best_t = t; best = (int)i; } }
    // clip the ray to the grid box
    float oo[3] = { o.x, o.y, o.z }, dd[3] = { d.x, d.y, d.z };
    float t0 = mint, t1 = best_t;
    for (int k = 0; k < 3; k++) {
        float hi = G.lo[k] + G.cell[k] * G.n[k];
        if (dd[k] == 0) { if (oo[k] < G.lo[k] || oo[k] > hi) return best; continue; }
        float a = (G.lo[k] - oo[k]) / dd[k], b = (hi - oo[k]) / dd[k]; if (a > b) std::swap(a, b);
        t0 = std::max(t0, a); t1 = std::min(t1, b);
    }
    ops.push_back('N');                       // the set-up step (clip + first cell) costs about one step
    if (t0 > t1) return best;
    int c[3], step[3]; float tn[3], dt[3];
    for (int k = 0; k < 3; k++) {
        float p = oo[k] + dd[k] * t0;
        c[k] = std::max(0, std::min(G.n[k] - 1, (int)std::floor((p - G.lo[k]) * G.inv_cell[k])));
        step[k] = dd[k] > 0 ? 1 : -1;
        if (dd[k] == 0) { tn[k] = 1e30f; dt[k] = 1e30f; }
        else { float edge = G.lo[k] + (c[k] + (dd[k] > 0 ? 1 : 0)) * G.cell[k]; tn[k] = (edge - oo[k]) / dd[k]; dt[k] = G.cell[k] / std::fabs(dd[k]); }
    }
    uint32_t last = 0xFFFFFFFFu;
    for (;;) {
        const auto &cell = G.cells[((size_t)c[0] * G.n[1] + c[1]) * G.n[2] + c[2]];
        float t_exit = std::min(tn[0], std::min(tn[1], tn[2]));
        for (uint32_t s : cell) {
            if (s == last) continue; last = s;
            ops.push_back('L'); float t;
            if (hit_sphere(S.sp[s], o, d, mint, best_t, t) && t < best_t) { best_t = t; best = (int)s; }
        }
        if (best_t <= t_exit || t_exit >= t1) break;
        int k = tn[0] <= tn[1] ? (tn[0] <= tn[2] ? 0 : 2) : (tn[1] <= tn[2] ? 1 : 2);
        c[k] += step[k]; tn[k] += dt[k];
        if (c[k] < 0 || c[k] >= G.n[k]) break;
        ops.push_back('N');
    }
    return best;
}

struct PathTrace { std::vector<std::string> seg; };     // one string of ops per closest-hit query of the path

static PathTrace trace_path(const Scene &S, const RtwCamera &cam, uint32_t i, uint32_t j, uint32_t depth) {
    PathTrace pt;
    float rx, ry; do { rx = 2 * rnd() - 1; ry = 2 * rnd() - 1; } while (rx * rx + ry * ry > 1.0f);
    V u = { cam.u[0], cam.u[1], cam.u[2] }, v = { cam.v[0], cam.v[1], cam.v[2] };
    V o = V{ cam.origin[0], cam.origin[1], cam.origin[2] } + (u * rx + v * ry) * cam.lens_radius;
    float jx = i + rnd(), jy = j + rnd();
    V d = V{ cam.pixel00[0], cam.pixel00[1], cam.pixel00[2] } + V{ cam.delta_u[0], cam.delta_u[1], cam.delta_u[2] } * jx + V{ cam.delta_v[0], cam.delta_v[1], cam.delta_v[2] } * jy;
    for (uint32_t k = 0; k < depth; k++) {
        std::string ops; float t;
        int best = g_use_grid ? closest_grid(S, o, d, t, ops) : closest(S, o, d, t, ops);
        pt.seg.push_back(ops);
        if (best < 0) break;
        const RtwSphere &s = S.sp[best];
        V p = o + d * t, n = unit(p - V{ s.center[0], s.center[1], s.center[2] }), ud = unit(d);
        bool front = !(dot(d, n) > 0);
        V next;
        if (s.opacity > 0) {
            V nn = front ? n : n * -1.0f; float ratio = front ? 1.0f / s.ir : s.ir;
            float ct = std::min(dot(ud * -1.0f, nn), 1.0f), st = std::sqrt(1 - ct * ct);
            float r0 = (1 - ratio) / (1 + ratio); r0 *= r0;
            float refl = r0 + (1 - r0) * std::pow(1 - ct, 5.0f);
            if (ratio * st > 1.0f || refl > rnd()) next = ud - nn * (2 * dot(ud, nn));
            else { V perp = (ud + nn * ct) * ratio; next = perp + nn * -std::sqrt(std::fabs(1 - dot(perp, perp))); }
        } else {
            V sc = n + rand_unit(), refl = ud - n * (2 * dot(ud, n));
            next = refl * s.metallicness + sc * (1 - s.metallicness);
        }
        o = p; d = next;
    }
    return pt;
}

// ---- the wave ------------------------------------------------------------------------------------------------------
struct Costs { double T = 180, L = 230, S = 2800, X = 520, Sbase = 0; };   // cycles per wave-step
struct Policy {
    int paths = 1;          // paths per lane
    unsigned s_hi = 48, t_lo = 6, x_hi = 16, burst = 3, l_hi = 0;   // l_hi != 0: LEAF runs only with >= l_hi lanes (or when nothing traverses)
    bool x_when_any = false;
};
struct Lane {
    // traversal side
    const std::string *ops = nullptr; size_t pos = 0; int tpath = -1;    // tpath: index into `live`, -1 none
    bool t_done = false;                                                    // query complete, result waiting
    // waiting side (2-path policy)
    int wpath = -1; int wstate = 0;                                         // 0 EMPTY 1 READY 2 SHADE 3 DEAD
};
struct LivePath { const PathTrace *p; size_t seg; uint32_t unit_left; };

struct Result { double cyc = 0; unsigned long long steps[4] = { 0, 0, 0, 0 }, lanes[4] = { 0, 0, 0, 0 }, segments = 0;
    unsigned long long idleT[4] = { 0, 0, 0, 0 }; /* during T steps: lanes in leaf / waiting shade / dead / (2-path) waiting switch */ };

// blocks: each block = 64 lanes' units; a unit = chunk_len consecutive paths of one pixel
static Result simulate(const std::vector<std::vector<std::vector<PathTrace>>> &blocks, const Policy &P, const Costs &C) {
    Result R;
    size_t next_block = 0, next_in_block = 0;
    struct Unit { const std::vector<PathTrace> *paths; size_t next; };
    auto fetch = [&](Unit &u) -> bool {
        if (next_block >= blocks.size()) return false;
        u.paths = &blocks[next_block][next_in_block]; u.next = 0;
        if (++next_in_block == 64) { next_in_block = 0; next_block++; }
        return true;
    };
    std::vector<Lane> lane(64);
    // per lane per side: the unit it works through
    std::vector<Unit> unitT(64), unitW(64);
    std::vector<const PathTrace *> pathT(64, nullptr), pathW(64, nullptr);
    std::vector<size_t> segT(64, 0), segW(64, 0);
    std::vector<bool> haveT(64, false), haveW(64, false), exhausted(64, false);

    auto in_trav = [&](int l) { return lane[l].ops && lane[l].pos < lane[l].ops->size() && (*lane[l].ops)[lane[l].pos] == 'N'; };
    auto in_leaf = [&](int l) { return lane[l].ops && lane[l].pos < lane[l].ops->size() && (*lane[l].ops)[lane[l].pos] == 'L'; };
    auto begin_query = [&](int l) {           // start the traversal of pathT[l]'s current segment
        lane[l].ops = &pathT[l]->seg[segT[l]]; lane[l].pos = 0; lane[l].t_done = lane[l].ops->empty();
    };
    auto advance = [&](int l) { if (++lane[l].pos >= lane[l].ops->size()) lane[l].t_done = true; };

    if (P.paths == 1) {
        // ---- the round-1 scheduler: lane states TRAVERSE / LEAF / waits-for-SHADE / DEAD ----
        std::vector<int> state(64, 2);   // 2 = wants SHADE (initially: needs a unit)
        for (;;) {
            unsigned nT = 0, nL = 0, nS = 0;
            for (int l = 0; l < 64; l++) { if (state[l] == 3) continue; if (state[l] == 2) nS++; else if (in_trav(l)) nT++; else if (in_leaf(l)) nL++; }
            if (!(nT | nL | nS)) break;
            bool run_shade = nS >= P.s_hi || (nT < P.t_lo && nL < P.t_lo && nS > 0);
            if (run_shade) {
                R.steps[2]++; R.lanes[2] += nS; R.cyc += C.S;
                for (int l = 0; l < 64; l++) if (state[l] == 2) {
                    bool need_new_path = !haveT[l];
                    if (haveT[l]) {                                    // shade the finished query
                        R.segments++;
                        if (++segT[l] >= pathT[l]->seg.size()) need_new_path = true;
                    }
                    if (need_new_path) {
                        if (!haveT[l] || unitT[l].next >= unitT[l].paths->size()) { if (!fetch(unitT[l])) { state[l] = 3; haveT[l] = false; continue; } }
                        pathT[l] = &(*unitT[l].paths)[unitT[l].next++]; segT[l] = 0; haveT[l] = true;
                    }
                    begin_query(l);
                    state[l] = lane[l].t_done ? 2 : 0;
                }
            } else if (P.l_hi ? !(nL >= P.l_hi || nT == 0) : nT >= nL) {
                unsigned live = nT;
                for (unsigned u = 0; u < P.burst && live; u++) {
                    R.steps[0]++; R.lanes[0] += live; R.cyc += C.T;
                    for (int l = 0; l < 64; l++) { if (state[l] == 3) R.idleT[2]++; else if (state[l] == 2) R.idleT[1]++; else if (in_leaf(l)) R.idleT[0]++; }
                    for (int l = 0; l < 64; l++) if (state[l] == 0 && in_trav(l)) { advance(l); if (lane[l].t_done) state[l] = 2; }
                    live = 0; for (int l = 0; l < 64; l++) if (state[l] == 0 && in_trav(l)) live++;
                }
            } else {
                R.steps[1]++; R.lanes[1] += nL; R.cyc += C.L;
                for (int l = 0; l < 64; l++) if (state[l] == 0 && in_leaf(l)) { advance(l); if (lane[l].t_done) state[l] = 2; }
            }
        }
        return R;
    }
    // ---- two paths per lane: T side {0 TRAV, 1 DONE, 2 NONE}, W side {0 EMPTY, 1 READY, 2 SHADE, 3 DEAD} ----
    std::vector<int> ts(64, 2), ws(64, 0);
    for (;;) {
        unsigned nT = 0, nL = 0, nX = 0, nS = 0;
        std::vector<char> canx(64, 0);
        for (int l = 0; l < 64; l++) {
            if (ts[l] == 0) { if (in_trav(l)) nT++; else if (in_leaf(l)) nL++; }
            bool idle = ts[l] != 0;
            canx[l] = idle && ws[l] != 2 && (ts[l] == 1 || ws[l] == 1);
            if (canx[l]) nX++;
            if (ws[l] == 2 || ws[l] == 0) nS++;
        }
        if (!(nT | nL | nX | nS)) break;
        bool starved = nT < P.t_lo && nL < P.t_lo;
        bool run_shade = nS >= P.s_hi || (starved && nS > 0 && nS >= nX);
        bool run_switch = !run_shade && (nX >= P.x_hi || (starved && nX > 0));
        if (run_shade) {
            R.steps[2]++; R.lanes[2] += nS; R.cyc += C.S;
            for (int l = 0; l < 64; l++) if (ws[l] == 2 || ws[l] == 0) {
                bool need_new_path = ws[l] == 0;
                if (ws[l] == 2) { R.segments++; if (++segW[l] >= pathW[l]->seg.size()) need_new_path = true; }
                if (need_new_path) {
                    if (!haveW[l] || unitW[l].next >= unitW[l].paths->size()) { if (!fetch(unitW[l])) { ws[l] = 3; haveW[l] = false; continue; } haveW[l] = true; }
                    pathW[l] = &(*unitW[l].paths)[unitW[l].next++]; segW[l] = 0;
                }
                ws[l] = 1;
            }
        } else if (run_switch) {
            R.steps[3]++; R.lanes[3] += nX; R.cyc += C.X;
            for (int l = 0; l < 64; l++) if (canx[l]) {
                bool handover = ts[l] == 1, begin = ws[l] == 1;
                std::swap(pathT[l], pathW[l]); std::swap(segT[l], segW[l]); std::swap(unitT[l], unitW[l]);
                { bool t = haveT[l]; haveT[l] = haveW[l]; haveW[l] = t; }
                int nws = handover ? 2 : (ws[l] == 3 ? 3 : 0);
                ws[l] = nws;
                if (begin) { begin_query(l); ts[l] = lane[l].t_done ? 1 : 0; } else { ts[l] = 2; lane[l].ops = nullptr; }
            }
        } else if (nT >= nL) {
            unsigned live = nT;
            for (unsigned u = 0; u < P.burst && live; u++) {
                R.steps[0]++; R.lanes[0] += live; R.cyc += C.T;
                for (int l = 0; l < 64; l++) if (ts[l] == 0 && in_trav(l)) { advance(l); if (lane[l].t_done) ts[l] = 1; }
                live = 0; for (int l = 0; l < 64; l++) if (ts[l] == 0 && in_trav(l)) live++;
            }
        } else {
            R.steps[1]++; R.lanes[1] += nL; R.cyc += C.L;
            for (int l = 0; l < 64; l++) if (ts[l] == 0 && in_leaf(l)) { advance(l); if (lane[l].t_done) ts[l] = 1; }
        }
    }
    return R;
}


// ---- idealised intra-wave pool: K paths per wave live in a pool; a lane whose query is done drops the result into the pool and takes ANY
// ready ray (step X, cost C.X); SHADE takes up to 64 finished paths from the pool (and refills empty pool slots with new paths).
static Result simulate_pool(const std::vector<std::vector<std::vector<PathTrace>>> &blocks, unsigned K, unsigned s_hi, unsigned x_hi, unsigned t_lo, unsigned burst, const Costs &C) {
    Result R;
    size_t next_block = 0, next_in_block = 0;
    struct Unit { const std::vector<PathTrace> *paths = nullptr; size_t next = 0; };
    auto fetch = [&](Unit &u) -> bool {
        if (next_block >= blocks.size()) return false;
        u.paths = &blocks[next_block][next_in_block]; u.next = 0;
        if (++next_in_block == 64) { next_in_block = 0; next_block++; }
        return true;
    };
    struct Slot { Unit unit; const PathTrace *p = nullptr; size_t seg = 0; int st = 0; /* 0 EMPTY 1 READY 2 TRAV(on a lane) 3 DONE 4 DEAD */ };
    std::vector<Slot> pool(K);
    struct L { int slot = -1; size_t pos = 0; };
    std::vector<L> lane(64);
    auto ops = [&](int l) -> const std::string & { return pool[lane[l].slot].p->seg[pool[lane[l].slot].seg]; };
    for (;;) {
        unsigned nT = 0, nL = 0, nIdle = 0, nReady = 0, nDone = 0, nEmpty = 0;
        for (auto &s : pool) { if (s.st == 1) nReady++; else if (s.st == 3) nDone++; else if (s.st == 0) nEmpty++; }
        for (int l = 0; l < 64; l++) { if (lane[l].slot < 0) nIdle++; else if (ops(l)[lane[l].pos] == 'N') nT++; else nL++; }
        unsigned nX = std::min(nIdle, nReady);
        unsigned nS = nDone + nEmpty;
        if (!(nT | nL | nX | nS)) break;
        bool starved = nT < t_lo && nL < t_lo;
        bool run_shade = nS >= s_hi || (starved && nS > 0 && nX == 0);
        bool run_x = !run_shade && (nX >= x_hi || (starved && nX > 0));
        if (run_shade) {
            unsigned served = 0;
            for (auto &s : pool) {
                if (served == 64) break;
                if (s.st != 3 && s.st != 0) continue;
                served++;
                bool need_new = s.st == 0;
                if (s.st == 3) { R.segments++; if (++s.seg >= s.p->seg.size()) need_new = true; }
                if (need_new) {
                    if (!s.unit.paths || s.unit.next >= s.unit.paths->size()) { if (!fetch(s.unit)) { s.st = 4; continue; } }
                    s.p = &(*s.unit.paths)[s.unit.next++]; s.seg = 0;
                }
                s.st = s.p->seg[s.seg].empty() ? 3 : 1;
            }
            R.steps[2]++; R.lanes[2] += served; R.cyc += C.S;
        } else if (run_x) {
            unsigned done = 0;
            for (int l = 0; l < 64 && done < nX; l++) if (lane[l].slot < 0) {
                for (unsigned k = 0; k < K; k++) if (pool[k].st == 1) { pool[k].st = 2; lane[l].slot = (int)k; lane[l].pos = 0; done++; break; }
            }
            R.steps[3]++; R.lanes[3] += nX; R.cyc += C.X;
        } else if (nT >= nL) {
            unsigned live = nT;
            for (unsigned u = 0; u < burst && live; u++) {
                R.steps[0]++; R.lanes[0] += live; R.cyc += C.T;
                live = 0;
                for (int l = 0; l < 64; l++) if (lane[l].slot >= 0 && ops(l)[lane[l].pos] == 'N') {
                    if (++lane[l].pos >= ops(l).size()) { pool[lane[l].slot].st = 3; lane[l].slot = -1; }
                    else if (ops(l)[lane[l].pos] == 'N') live++;
                }
            }
        } else {
            R.steps[1]++; R.lanes[1] += nL; R.cyc += C.L;
            for (int l = 0; l < 64; l++) if (lane[l].slot >= 0 && ops(l)[lane[l].pos] == 'L') {
                if (++lane[l].pos >= ops(l).size()) { pool[lane[l].slot].st = 3; lane[l].slot = -1; }
            }
        }
    }
    return R;
}

// ---- one path per lane, SHADE split into H (scatter at a hit), N (path ended: bank, next sample / unit, camera ray) and B (trav_begin) ----
struct SplitCosts { double H = 1500, N = 700, B = 450, fusedS = 2800; };
static bool g_fuse_b = false;
static Result simulate_split(const std::vector<std::vector<std::vector<PathTrace>>> &blocks, unsigned h_hi, unsigned n_hi, unsigned b_hi, unsigned t_lo, unsigned burst,
                             const Costs &C, const SplitCosts &SC, double *cyc_parts) {
    Result R;
    size_t next_block = 0, next_in_block = 0;
    struct Unit { const std::vector<PathTrace> *paths = nullptr; size_t next = 0; };
    auto fetch = [&](Unit &u) -> bool {
        if (next_block >= blocks.size()) return false;
        u.paths = &blocks[next_block][next_in_block]; u.next = 0;
        if (++next_in_block == 64) { next_in_block = 0; next_block++; }
        return true;
    };
    // lane state: 0 TRAV/LEAF (by ops), 1 wants H, 2 wants N, 3 wants B, 4 dead
    struct L { int st = 2; Unit unit; const PathTrace *p = nullptr; size_t seg = 0, pos = 0; bool have = false; };
    std::vector<L> lane(64);
    auto opsof = [&](int l) -> const std::string & { return lane[l].p->seg[lane[l].seg]; };
    unsigned long long stepsH = 0, lanesH = 0, stepsN = 0, lanesN = 0, stepsB = 0, lanesB = 0;
    for (;;) {
        unsigned nT = 0, nL = 0, nH = 0, nN = 0, nB = 0;
        for (int l = 0; l < 64; l++) {
            if (lane[l].st == 0) { if (opsof(l)[lane[l].pos] == 'N') nT++; else nL++; }
            else if (lane[l].st == 1) nH++; else if (lane[l].st == 2) nN++; else if (lane[l].st == 3) nB++;
        }
        if (!(nT | nL | nH | nN | nB)) break;
        bool starved = nT < t_lo && nL < t_lo;
        int run = -1;                                     // 0 T 1 L 2 H 3 N 4 B
        if (nB >= b_hi) run = 4; else if (nH >= h_hi) run = 2; else if (nN >= n_hi) run = 3;
        else if (starved && (nB | nH | nN)) { run = nB >= nH && nB >= nN ? 4 : (nH >= nN ? 2 : 3); }
        else run = nT >= nL ? 0 : 1;
        auto done_query = [&](int l) {                    // traversal of the current segment finished: hit or miss?
            const bool last = lane[l].seg + 1 >= lane[l].p->seg.size();
            lane[l].st = last ? 2 : 1;                    // (the last query of a path is the one that ended it: miss / depth)
        };
        if (run == 2) {
            stepsH++; lanesH += nH; R.cyc += SC.H; cyc_parts[0] += SC.H;
            if (g_fuse_b) { R.cyc += SC.B; cyc_parts[0] += SC.B; }
            for (int l = 0; l < 64; l++) if (lane[l].st == 1) { R.segments++; lane[l].seg++; lane[l].st = 3; if (g_fuse_b) { lane[l].pos = 0; lane[l].st = 0; if (opsof(l).empty()) done_query(l); } }
        } else if (run == 3) {
            stepsN++; lanesN += nN; R.cyc += SC.N; cyc_parts[1] += SC.N;
            if (g_fuse_b) { R.cyc += SC.B; cyc_parts[1] += SC.B; }
            for (int l = 0; l < 64; l++) if (lane[l].st == 2) {
                if (lane[l].have) R.segments++;
                if (!lane[l].have || lane[l].unit.next >= lane[l].unit.paths->size()) { if (!fetch(lane[l].unit)) { lane[l].st = 4; lane[l].have = false; continue; } }
                lane[l].p = &(*lane[l].unit.paths)[lane[l].unit.next++]; lane[l].seg = 0; lane[l].have = true; lane[l].st = 3;
                if (g_fuse_b) { lane[l].pos = 0; lane[l].st = 0; if (opsof(l).empty()) done_query(l); }
            }
        } else if (run == 4) {
            stepsB++; lanesB += nB; R.cyc += SC.B; cyc_parts[2] += SC.B;
            for (int l = 0; l < 64; l++) if (lane[l].st == 3) { lane[l].pos = 0; lane[l].st = 0; if (opsof(l).empty()) done_query(l); }
        } else if (run == 0) {
            unsigned live = nT;
            for (unsigned u = 0; u < burst && live; u++) {
                R.steps[0]++; R.lanes[0] += live; R.cyc += C.T;
                live = 0;
                for (int l = 0; l < 64; l++) if (lane[l].st == 0 && opsof(l)[lane[l].pos] == 'N') {
                    if (++lane[l].pos >= opsof(l).size()) done_query(l); else if (opsof(l)[lane[l].pos] == 'N') live++;
                }
            }
        } else {
            R.steps[1]++; R.lanes[1] += nL; R.cyc += C.L;
            for (int l = 0; l < 64; l++) if (lane[l].st == 0 && opsof(l)[lane[l].pos] == 'L') { if (++lane[l].pos >= opsof(l).size()) done_query(l); }
        }
    }
    R.steps[2] = stepsH; R.lanes[2] = lanesH; R.steps[3] = stepsN; R.lanes[3] = lanesN;
    cyc_parts[3] = (double)stepsB; cyc_parts[4] = (double)lanesB;
    return R;
}

int main(int argc, char **argv) {
    uint32_t ns = 0, nt = 0, nx = 0;
    rtw_scene_generate(RTW_SCENE_C2_BOOK1_FINAL, 42, nullptr, 0, &ns, nullptr, 0, &nt, nullptr, 0, &nx);
    Scene S; S.sp.resize(ns);
    rtw_scene_generate(RTW_SCENE_C2_BOOK1_FINAL, 42, S.sp.data(), ns, &ns, nullptr, 0, &nt, nullptr, 0, &nx);
    build_bvh(S.sp.data(), ns, 0, 0, S.bvh);
    if (getenv("SIM_GRID")) { g_use_grid = true; build_grid(S, (float)atof(getenv("SIM_GRID"))); }
    if (getenv("SIM_SCENE")) {                 // other generator scenes: 4 = dielectric-heavy
        uint32_t which = (uint32_t)atoi(getenv("SIM_SCENE"));
        rtw_scene_generate(which, 42, nullptr, 0, &ns, nullptr, 0, &nt, nullptr, 0, &nx);
        S.sp.resize(ns); std::vector<RtwTexture> tx(nt ? nt : 1); std::vector<float> tl(3 * (nx ? nx : 1));
        rtw_scene_generate(which, 42, S.sp.data(), ns, &ns, tx.data(), nt, &nt, tl.data(), nx, &nx);
        build_bvh(S.sp.data(), ns, 0, 0, S.bvh);
        if (g_use_grid) build_grid(S, (float)atof(getenv("SIM_GRID")));
    }
    RtwCamera cam; RtwParams p;
    rtw_scene_default_view(RTW_SCENE_C5_MOTION_CHECKER, &cam, &p);
    const uint32_t chunk = 4, n_tiles = argc > 1 ? atoi(argv[1]) : 40, chunks_per_tile = argc > 2 ? atoi(argv[2]) : 6;
    // blocks = (tile, chunk): 64 pixels x `chunk` paths, consecutive chunks of a tile are consecutive in the queue
    std::vector<std::vector<std::vector<PathTrace>>> blocks;
    std::mt19937 pick(7);
    unsigned long long nseg = 0, nN = 0, nL = 0;
    for (uint32_t t = 0; t < n_tiles; t++) {
        uint32_t tx = pick() % (p.width / 8), ty = pick() % (p.height / 8);
        for (uint32_t c = 0; c < chunks_per_tile; c++) {
            std::vector<std::vector<PathTrace>> blk(64);
            for (uint32_t q = 0; q < 64; q++) for (uint32_t s = 0; s < chunk; s++) {
                blk[q].push_back(trace_path(S, cam, tx * 8 + (q & 7), ty * 8 + (q >> 3), p.depth));
                for (auto &sg : blk[q].back().seg) { nseg++; for (char ch : sg) (ch == 'N' ? nN : nL)++; }
            }
            blocks.push_back(std::move(blk));
        }
    }
    printf("traced %zu blocks: %llu segments, %.2f node visits + %.2f leaf tests per segment, %.2f segments per path\n", blocks.size(), nseg,
           (double)nN / nseg, (double)nL / nseg, (double)nseg / (blocks.size() * 64.0 * chunk));
    Costs C;
    if (getenv("SIM_X")) C.X = atof(getenv("SIM_X"));
    if (getenv("SIM_T")) C.T = atof(getenv("SIM_T"));
    if (getenv("SIM_L")) C.L = atof(getenv("SIM_L"));
    if (getenv("SIM_S")) C.S = atof(getenv("SIM_S"));
    auto report = [&](const char *name, const Policy &P) {
        Result R = simulate(blocks, P, C);
        auto eff = [&](int k) { return R.steps[k] ? R.lanes[k] / (64.0 * R.steps[k]) : 0.0; };
        double tt = R.steps[0] * C.T, tl = R.steps[1] * C.L, tsd = R.steps[2] * C.S, tx = R.steps[3] * C.X;
        printf("%-44s cyc/seg %7.1f | T %.3f (%4.1f%%) L %.3f (%4.1f%%) S %.3f (%4.1f%%) X %.3f (%4.1f%%)\n", name, R.cyc / R.segments,
               eff(0), 100 * tt / R.cyc, eff(1), 100 * tl / R.cyc, eff(2), 100 * tsd / R.cyc, eff(3), 100 * tx / R.cyc);
        if (R.idleT[0] + R.idleT[1] + R.idleT[2]) printf("    during T steps: %.3f of the lanes sit in LEAF, %.3f wait for SHADE, %.3f are dead\n",
            R.idleT[0] / (64.0 * R.steps[0]), R.idleT[1] / (64.0 * R.steps[0]), R.idleT[2] / (64.0 * R.steps[0]));
    };
    Policy base; report("1 path: s_hi 48 t_lo 6 (round 1)", base);
    if (argc > 3 && !strcmp(argv[3], "split")) {
        SplitCosts SC;
        if (getenv("SIM_H")) SC.H = atof(getenv("SIM_H"));
        if (getenv("SIM_N")) SC.N = atof(getenv("SIM_N"));
        if (getenv("SIM_B")) SC.B = atof(getenv("SIM_B"));
        g_fuse_b = getenv("SIM_FUSE_B") != nullptr;
        for (unsigned hh : { 32u, 40u, 48u, 56u }) for (unsigned nh : { 16u, 24u, 32u, 48u }) for (unsigned bh : { 16u, 32u, 48u }) {
            double parts[5] = { 0, 0, 0, 0, 0 };
            Result R = simulate_split(blocks, hh, nh, bh, 6, 3, C, SC, parts);
            auto eff = [&](int k) { return R.steps[k] ? R.lanes[k] / (64.0 * R.steps[k]) : 0.0; };
            printf("split h_hi %2u n_hi %2u b_hi %2u: cyc/seg %7.1f | T %.3f (%4.1f%%) L %.3f (%4.1f%%) H %.3f (%4.1f%%) N %.3f (%4.1f%%) B %.3f (%4.1f%%)\n", hh, nh, bh,
                   R.cyc / R.segments, eff(0), 100 * R.steps[0] * C.T / R.cyc, eff(1), 100 * R.steps[1] * C.L / R.cyc, eff(2), 100 * parts[0] / R.cyc,
                   eff(3), 100 * parts[1] / R.cyc, parts[3] ? parts[4] / (64.0 * parts[3]) : 0.0, 100 * parts[2] / R.cyc);
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "pool")) {
        for (unsigned K : { 96u, 128u, 192u, 256u }) for (unsigned sh : { 48u, 64u }) for (unsigned xh : { 8u, 16u, 32u }) {
            Result R = simulate_pool(blocks, K, sh, xh, 6, 3, C);
            auto eff = [&](int k) { return R.steps[k] ? R.lanes[k] / (64.0 * R.steps[k]) : 0.0; };
            printf("pool K %3u s_hi %2u x_hi %2u: cyc/seg %7.1f | T %.3f (%4.1f%%) L %.3f (%4.1f%%) S %.3f (%4.1f%%) X %.3f (%4.1f%%)\n", K, sh, xh, R.cyc / R.segments,
                   eff(0), 100 * R.steps[0] * C.T / R.cyc, eff(1), 100 * R.steps[1] * C.L / R.cyc, eff(2), 100 * R.steps[2] * C.S / R.cyc, eff(3), 100 * R.steps[3] * C.X / R.cyc);
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "leaf")) {
        for (unsigned sh : { 48u, 52u }) for (unsigned lh : { 0u, 8u, 16u, 24u, 32u, 40u, 48u }) {
            Policy q; q.s_hi = sh; q.l_hi = lh;
            char nm[96]; snprintf(nm, sizeof nm, "1 path: s_hi %u l_hi %u", sh, lh);
            report(nm, q);
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "grid1")) {
        for (unsigned sh : { 32u, 40u, 48u, 56u, 62u }) for (unsigned tl : { 2u, 6u, 12u, 20u }) for (unsigned b : { 2u, 3u, 5u }) {
            Policy q; q.s_hi = sh; q.t_lo = tl; q.burst = b;
            char nm[96]; snprintf(nm, sizeof nm, "1 path: s_hi %u t_lo %u burst %u", sh, tl, b);
            report(nm, q);
        }
        return 0;
    }
    for (unsigned sh : { 40u, 48u, 56u }) for (unsigned xh : { 1u, 4u, 8u, 16u, 32u }) for (unsigned tl : { 6u, 12u }) {
        Policy q; q.paths = 2; q.s_hi = sh; q.x_hi = xh; q.t_lo = tl;
        char nm[96]; snprintf(nm, sizeof nm, "2 paths: s_hi %u x_hi %u t_lo %u", sh, xh, tl);
        report(nm, q);
    }
    return 0;
}
