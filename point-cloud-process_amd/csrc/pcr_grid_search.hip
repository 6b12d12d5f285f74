// Exact 1-NN search over the multi-level voxel-hash grid (see pcr_grid.hip for the layout).
//
// Queries are processed in WAVE TILES of 32 consecutive records of the Morton-sorted query cloud (a rigid transform
// keeps a tile spatially compact, so the cloud is sorted once per pair).
//
//   wave tile   (wtile_search) one wave per tile, no block-level barrier: box of the cubes its queries claim (cells + 1
//               ring at first, bound balls later; inside the ICP loop the previous pass's neighbour bounds every ball at
//               once), directory through the 2x2x2-block table (one probe per lane), the box's points staged ONCE into the
//               wave's LDS slice as binary32 coordinates about the box centre, every query compared with every staged point
//               on the matrix cores (expanded form) with a rigorous bound on its rounding error, the winner by the direct form.
//   descent     (hard_search) what a tile cannot prove (ambiguous filter results, balls that need too many cells or points,
//               no neighbour inside the gate), one wave per query: pruned descent of the nested cell hierarchy (cells are
//               contiguous runs of the Morton-sorted target at every level), 64 children box-tested per step.
//   ICP pass    (grid_pass_kernel) ONE launch per iteration of the device-resident loop: tile, moments (fixed point), a
//               device-side work queue for the open queries served by the waves whose tile is done, and the last wave solves
//               the 3x3 step and tests convergence, so the host is out of the iteration.  grid_drain_kernel: second launch of
//               the throughput variant.
//   stand-alone grid_wtile_kernel -> grid_hard_kernel (-> grid_accumulate_kernel): the nn1 API, ungated runs, the host loop.
// Round 1's stage was a 64-query tile per 256-thread block (4 barriers per tile, ~980 staged candidates per query, all
// resident tiles in lockstep): 40 us for the same pass that takes the wave tiles ~30 us with a third of the hard-stage
// work; DESIGN.md section 7 keeps its measurements.  Measured alternatives before that, all exact, all slower on the
// 120k KITTI-shaped pair: 8 lanes/query over the 27 cells (117 us), flattened directory (174 us), prune-then-visit
// (88-180 us) -- dependent lookup->scan chains and uncoalesced 32-B reads dominate there.
// The unresolved queries of a tile are appended to one of 32 lists with a single atomic per wave (a per-query global
// append saturates at ~88 appends/us on this chip and cost more than the search).  Pruning only ever skips a cell whose
// box distance exceeds a bound that is itself >= the final answer, so every stage returns the exact nearest neighbour
// (lowest index on ties).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include "pcr_grid_dev.h"
#include "pcr_icp_step.h"

#ifndef PCR_HARD_SCAN_T
#define PCR_HARD_SCAN_T 192
#endif
constexpr unsigned int HARD_SCAN_T = PCR_HARD_SCAN_T;   // the hard stage scans cells up to this size, descends into bigger ones
constexpr int HARD_STACK = 160;
constexpr long long ID_NONE = 0x7fffffffffffffffll;
// The unresolved queries go to H_NLIST sub-lists (tile t appends to list t % H_NLIST): one returning atomic per tile on
// a single counter serialises at ~88/us on this chip -- 21 us for the 1875 tiles of a 120k scan, all finishing together.
// Counter j lives at hard_count[H_CSTRIDE * j] (separate cache lines); list j is hard_list[j * cap, (j + 1) * cap).
constexpr int H_NLIST = 32;
constexpr int H_CSTRIDE = 32;
__host__ __device__ static inline unsigned int hard_list_cap(long long nq) {
    // tiles (of 64 or 16 queries) append to list (tile % H_NLIST): room for every query of the lists' fair share of tiles
    const long long tiles = (nq + 15) / 16;
    return (unsigned int)(((tiles + H_NLIST - 1) / H_NLIST) * 16 + 64);
}

template <int G>
__device__ static inline void group_best(double& bd2, long long& bid, unsigned int& bpos) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const double od2 = __shfl_xor(bd2, off, 64);
        const long long oid = __shfl_xor(bid, off, 64);
        const unsigned int opos = __shfl_xor(bpos, off, 64);
        if (better(od2, oid, bd2, bid)) { bd2 = od2; bid = oid; bpos = opos; }
    }
}

struct wt_xyz;   // (the 24 coordinate bytes of a record, defined below)
__device__ static inline void scan_range(const pcr_pt* __restrict__ pts, unsigned int s, unsigned int e, unsigned int step, double ax,
                                         double ay, double az, double& bd2, long long& bid, unsigned int& bpos, double* bw) {
    for (unsigned int j = s; j < e; j += step) {
        const pcr_pt b = pts[j];
        const double d2 = dist2(ax, ay, az, b);
        if (better(d2, b.id, bd2, bid)) { bd2 = d2; bid = b.id; bpos = j; bw[0] = b.x; bw[1] = b.y; bw[2] = b.z; }
    }
}

__device__ static inline float fmin3(float a, float b, float c) {  // v_min3_f32 (inputs are never NaN here)
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ static inline double sq_pos(double v) {
    v = fmax(v, 0.0);
    return v * v;
}

// --------------------------------------------------------------- wave tile
// Second-generation tile stage: ONE WAVE = one tile of 16 consecutive queries, no block-level barrier anywhere.
// Why: with 64-query tiles every query was compared with the union of 64 neighbourhoods (~980 staged points per
// query on a KITTI scan: the kernel spent its time at the binary32 VALU peak on candidates no query needed), and the
// four waves of a block marched through load / directory / staging / filter in lockstep, so all resident tiles hit
// the memory system and then the VALU together.  A 16-query tile's box holds 3-4x fewer points at the same
// directory and staging cost per query, and independent waves de-synchronise by themselves.
//   lanes: query = lane & 15 (four copies), candidate slice = lane >> 4.
//   Up to WT_PASSES passes.  In every pass each still-open query claims a cube of half-width rho around itself:
//   rho = sqrt(bound) once it has a candidate (bound = the filter's rigorous upper bound U on the winner's squared
//   distance), otherwise cell0 * 4^pass (a guess, capped by the gate).  The pass box is the bounding box of the
//   claimed cubes in cells of the finest level that keeps it within WT_MAXC cells / 64 two-cell blocks; everything in
//   the box is staged and filtered.  A query is proven when the ball of radius sqrt(min(U, gate)) lies inside the box
//   that was actually staged and the filter's second-best exceeds U.  What is still open after the last pass
//   (ambiguous filter results, boxes with too many points, clamped coordinates) goes to the hard stage.
#ifndef PCR_WT_Q
#define PCR_WT_Q 32
#endif
constexpr int WT_Q = PCR_WT_Q;      // queries per wave tile (16 or 32)
#ifndef PCR_WT_MAXC
#define PCR_WT_MAXC 384
#endif
constexpr int WT_MAXC = PCR_WT_MAXC;        // cells in a wave-tile box (a 7 x 7 x 7 box = 64 two-cell blocks fits: the gate ball of a sparse-region query at 0.8 m cells)
#ifndef PCR_WT_PR
#define PCR_WT_PR 192
#endif
constexpr int WT_PR = PCR_WT_PR;    // points staged per round (chunks of 64, loaded back to back)
constexpr int WT_CH = WT_PR / 64;
static_assert(WT_PR % 64 == 0, "whole chunks");
#ifndef PCR_WT_PASSES
#define PCR_WT_PASSES 3
#endif
constexpr int WT_PASSES = PCR_WT_PASSES;
#ifndef PCR_WT_PASSES_SEEDED
#define PCR_WT_PASSES_SEEDED 1
#endif
constexpr int WT_PASSES_SEEDED = WT_PASSES < PCR_WT_PASSES_SEEDED ? WT_PASSES : PCR_WT_PASSES_SEEDED;
#ifndef PCR_WT_HEAVY_STOP
#define PCR_WT_HEAVY_STOP 384   // (192: 78.8-81.2, 384: 77.6-79.0, 576: 79.8-80.0 us for a one-iteration registration of the 120k pair; without: 81.4-82.0)
#endif
constexpr int WT_HEAVY_STOP = PCR_WT_HEAVY_STOP;
constexpr int WT_ROUNDS_SMALL = 768 / WT_PR, WT_ROUNDS_LARGE = 2304 / WT_PR;   // staged-point caps of 768 / 2304 per tile

#ifndef PCR_WT_MFMA
#define PCR_WT_MFMA 1   // filter on the matrix cores (v_mfma_f32_32x32x2_f32); 0 = packed binary32 VALU filter
#endif
struct wtile_lds {
    alignas(16) float px[WT_PR + 8], py[WT_PR + 8], pz[WT_PR + 8];
#if PCR_WT_MFMA
    alignas(16) float pn[WT_PR + 8];   // |p|^2 of the staged point (binary32, from the rounded local coordinates)
#endif
    unsigned int ppos[WT_PR];
    unsigned int c_start[WT_MAXC];
    unsigned int c_off[WT_MAXC + 1];
    unsigned short own[WT_PR];
};

// LDS hand-off between the lanes of ONE wave: LDS operations of a wave execute in order, so no hardware wait is
// needed, but the compiler must neither forward a lane's own earlier store to its load nor move accesses across
__device__ static inline void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ static inline unsigned int wave_incl_scan_max(unsigned int v) {
    v = max(v, dpp_u32<0x111, 0xf>(0u, v));
    v = max(v, dpp_u32<0x112, 0xf>(0u, v));
    v = max(v, dpp_u32<0x114, 0xf>(0u, v));
    v = max(v, dpp_u32<0x118, 0xf>(0u, v));
    v = max(v, dpp_u32<0x142, 0xa>(0u, v));
    v = max(v, dpp_u32<0x143, 0xc>(0u, v));
    return v;
}
// binary64 wave total in lane 63 (inclusive-scan pattern; lanes without a source add +0.0)
template <int CTRL, int ROW_MASK>
__device__ static inline double dpp_add_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);
}
__device__ static inline double wave_total_f64(double v) {
    v = dpp_add_f64<0x111, 0xf>(v);
    v = dpp_add_f64<0x112, 0xf>(v);
    v = dpp_add_f64<0x114, 0xf>(v);
    v = dpp_add_f64<0x118, 0xf>(v);
    v = dpp_add_f64<0x142, 0xa>(v);
    v = dpp_add_f64<0x143, 0xc>(v);
    return v;
}

// all-reduce inside every row of 16 lanes (= the 16 queries of a candidate slice): quad swaps, half mirror, mirror
__device__ static inline int row16_min(int v) {
    // (old = 0 with bound_ctrl: every lane has a source under these permutations, and the compiler can then fold the move into the
    // v_min / v_max -- with old = v it emitted v_mov + v_mov_dpp + v_min per step, 72 instructions for the six reductions of a tile)
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));  // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}
__device__ static inline int row16_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
    return v;
}

struct wt_xyz { double x, y, z; };  // the 24 coordinate bytes of a pcr_pt record

// the 24 coordinate bytes of a record by GLOBAL loads: the grid view's pointers are generic, and copying a struct through an
// address_space(1) pointer still comes out as flat loads (the aggregate copy drops the address space); three doubles do not
__device__ static inline wt_xyz ld_xyz_global(const void* p) {
    const __attribute__((address_space(1))) double* g = (const __attribute__((address_space(1))) double*)reinterpret_cast<const double*>(p);
    return wt_xyz{g[0], g[1], g[2]};
}

// pointers read from the device copy of the grid view are generic; tell the compiler they are global memory
template <typename T>
__device__ static inline const T* as_global(const T* p) {
    return (const T*)(const __attribute__((address_space(1))) T*)p;
}

// what a wave tile leaves behind in every lane (the 64 / WT_Q copies of a query agree)
struct wt_state {
    double ax, ay, az;        // the transformed query
    long long qi;
    float bound2;             // squared radius that provably holds the nearest neighbour (or the gate)
    unsigned int cand_pos;    // the candidate behind bound2 (POS_NONE: only the gate, or nothing, bounds the search)
    unsigned int won;         // proven nearest neighbour; POS_NONE = nothing within the gate, or still open
    bool open, clamped, qvalid;
    unsigned int staged;      // target points in the tile's boxes, summed over its passes (wave-uniform)
    unsigned long long dbg_pairs;
    unsigned int dbg_passes;
    unsigned long long probe;   // 1: the probe words (started counts of all groups) read in the middle of the tile's first pass were all complete
};

// what a tile needs from memory before anything else: requested in one go by the caller (with whatever else it needs then)
struct wt_pre {
    pcr_pt p;                 // the query record
    unsigned int seed_pos;    // ICP passes after the first: the previous pass's neighbour (res_pos) ...
    wt_xyz seed_b;            // ... and its coordinates (prev_xyz); stale where seed_pos is POS_NONE
};
__device__ __forceinline__ static void wtile_preload(const unsigned int tile, const int lane, const pcr_pt* __restrict__ q, const long long nq,
                                                     const unsigned int* __restrict__ res_pos, const wt_xyz* __restrict__ prev_xyz, wt_pre& P) {
    const long long qi = (long long)tile * WT_Q + (lane & (WT_Q - 1));
    P.seed_pos = POS_NONE;
    P.seed_b = wt_xyz{0.0, 0.0, 0.0};
    P.p.x = P.p.y = P.p.z = 0.0; P.p.id = 0;
    if (qi < nq) {
        // the seed travels with the query record: three dependent round trips (record -> res_pos -> prev_xyz) measured 2.7 us
        // at the head of every tile
        if (prev_xyz) {
            P.seed_pos = res_pos[qi];
            P.seed_b = prev_xyz[qi];
        }
        P.p = q[qi];
    }
}

#ifndef PCR_WT_WAVES
#define PCR_WT_WAVES 4   // keep the register allocator at <= 128 VGPRs: it drifts to 130-145 (3 waves per SIMD) on small edits
#endif
// The tile search proper, shared by the stand-alone tile kernel (nn1 API, host loop) and the one-kernel ICP pass.
// `tile` = index of the wave's run of WT_Q queries; proven queries have their res_pos (res_d2) written here.
__device__ __forceinline__ static void wtile_search(const pcr_grid_view& gv, wtile_lds* L, const unsigned int tile, const int lane, const wt_pre& P,
                                                    pcr_pt* __restrict__ q, const long long nq, const pcr_xform& x, const int has_x, const int write_back,
                                                    const double max_d2, const int gated, const unsigned int pcap, unsigned int* __restrict__ res_pos,
                                                    double* __restrict__ res_d2, unsigned long long* __restrict__ dbg,
                                                    const wt_xyz* __restrict__ prev_xyz, wt_state& S, const unsigned long long* probe_p = nullptr,
                                                    const unsigned int probe_stride = 0, const unsigned int probe_groups = 0, const unsigned int probe_total = 0) {
#if !PCR_WT_MFMA
    typedef float f2 __attribute__((ext_vector_type(2)));
#endif
    typedef float f4 __attribute__((ext_vector_type(4)));
    const pcr_pt* __restrict__ g_pts = as_global(gv.pts);
#ifdef PCR_WT_DIAG   // phase stamps + "why still open" counters for scripts/wt_stamps.py: a diagnostic build only (they cost registers)
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long t_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = t_start;
#define WT_STAMP(i) do { if (dbg) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_ph[i] += t_now - t_last; t_last = t_now; } } while (0)
#define WT_WHY(v) do { why = (v); } while (0)
    int why = 0;   // why the query was still open after its last pass (1 level, 2 too many points, 3 ambiguous, 4 ball out of the box)
#else
#define WT_STAMP(i) do {} while (0)
#define WT_WHY(v) do {} while (0)
#endif
    const long long qi = (long long)tile * WT_Q + (lane & (WT_Q - 1));
    const bool qvalid = qi < nq;
    double ax = 0, ay = 0, az = 0;
    bool clamped = false;
    const unsigned int seed_pos = P.seed_pos;
    const wt_xyz seed_b = P.seed_b;
    if (qvalid) {
        const pcr_pt p = P.p;
        ax = p.x; ay = p.y; az = p.z;
        if (has_x) {
            xform_apply(x, p, &ax, &ay, &az);
            // in-place transform of the source (main.py:110); every lane of this wave has loaded its record by now
            // and no other wave reads these 16 records
            if (write_back && lane < WT_Q) {
                pcr_pt o;
                o.x = ax; o.y = ay; o.z = az; o.id = p.id;
                q[qi] = o;
            }
        }
        cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
    }
    
    WT_STAMP(0);
    // per-query state across the passes (identical in the four lanes of a query after every merge)
    bool open = qvalid;                         // not yet proven
    const float gate2 = gated ? (float)max_d2 * (1.0f + 1e-6f) : INFINITY;   // rounded up: only ever used as an outer bound
    float bound2 = gate2;                       // squared radius that provably holds the nearest neighbour (or the gate)
    unsigned int cand_pos = POS_NONE, won = POS_NONE;
    if (prev_xyz && qvalid && !clamped) {
        // ICP iterations after the first: res_pos still holds every query's neighbour of the PREVIOUS pass (the target
        // never changes) and prev_xyz its coordinates (the accumulate kernel, which gathers the record anyway, leaves them
        // there: no dependent gather here).  The exact distance from the query's new position to that point bounds the
        // search ball at once, so the first box is the bounding box of tight balls instead of cells + 1 ring, and one
        // pass proves almost every query.
        const unsigned int pp = seed_pos;
        if (pp != POS_NONE) {
            const wt_xyz b = seed_b;
            const double dx = ax - b.x, dy = ay - b.y, dz = az - b.z;
            const float up = (float)((dx * dx + dy * dy) + dz * dz) * (1.0f + 2e-7f) + 1e-37f;   // rounded up
            if (up < bound2) { bound2 = up; cand_pos = pp; }
        }
    }
    unsigned long long dbg_pairs = 0, probe_v = 0;
    unsigned int dbg_passes = 0, staged = 0;
    // seeded tiles (ICP passes after the first) settle 97 % of their queries in pass 0; a third pass only feeds the kernel's tail
    const int max_passes = prev_xyz ? WT_PASSES_SEEDED : WT_PASSES;
    unsigned int staged_done = 0;   // points actually staged so far (wave-uniform)
    float ring = 1.0f;   // half-width, in level-0 cells, of the cube a query without a candidate claims (wave-uniform)
#pragma unroll 1
    for (int pass = 0; pass < max_passes; ++pass) {
        const bool part = open && !clamped;
        if (!__any(part)) break;
        // unseeded passes: a tile that has already staged a lot leaves what is still open to the queue instead of staging another
        // box -- the first pass of an ICP ends with its slowest (three-pass) tiles, and since dense tiles retry with their own cells
        // (below) the queue has the room: slowest tile 39.6 -> 31.0 us, 4 366 -> 4 998 items, one-iteration registration 82 -> 78.5 us
        if (pass > 0 && staged_done > (unsigned int)WT_HEAVY_STOP) break;
        // ---- the cube every open query claims, in level-0 cell coordinates
        int mn[3], mx[3];
        {
            const float rho = (cand_pos != POS_NONE) ? sqrtf(bound2) * (1.0f + 1e-6f) : fminf((float)gv.cell0 * ring, sqrtf(gate2));
            // half-width in cells; the claim only steers the box (what is proven is decided against the box actually staged)
            const double rc = fmin((double)rho * gv.inv_cell0, 262144.0);
            // the query in level-0 cell units (clamped queries never take part)
            const double tx = (ax - gv.lo[0]) * gv.inv_cell0, ty = (ay - gv.lo[1]) * gv.inv_cell0, tz = (az - gv.lo[2]) * gv.inv_cell0;
            const int cmax = (int)PCR_COORD_MAX;
            mn[0] = part ? max(0, min(cmax, (int)floor(tx - rc) + (int)PCR_COORD_BIAS)) : 0x7fffffff;
            mx[0] = part ? max(0, min(cmax, (int)floor(tx + rc) + (int)PCR_COORD_BIAS)) : -1;
            mn[1] = part ? max(0, min(cmax, (int)floor(ty - rc) + (int)PCR_COORD_BIAS)) : 0x7fffffff;
            mx[1] = part ? max(0, min(cmax, (int)floor(ty + rc) + (int)PCR_COORD_BIAS)) : -1;
            mn[2] = part ? max(0, min(cmax, (int)floor(tz - rc) + (int)PCR_COORD_BIAS)) : 0x7fffffff;
            mx[2] = part ? max(0, min(cmax, (int)floor(tz + rc) + (int)PCR_COORD_BIAS)) : -1;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            int lo_k = row16_min(mn[k]), hi_k = row16_max(mx[k]);   // every row of 16 lanes
            if (WT_Q >= 32) {   // queries 16..31 sit in row 1 (and 32..63 in rows 2, 3)
                lo_k = min(__builtin_amdgcn_readlane(lo_k, 0), __builtin_amdgcn_readlane(lo_k, 16));
                hi_k = max(__builtin_amdgcn_readlane(hi_k, 0), __builtin_amdgcn_readlane(hi_k, 16));
            }
            if (WT_Q == 64) {
                const int lo2 = min(__builtin_amdgcn_readlane(row16_min(mn[k]), 32), __builtin_amdgcn_readlane(row16_min(mn[k]), 48));
                const int hi2 = max(__builtin_amdgcn_readlane(row16_max(mx[k]), 32), __builtin_amdgcn_readlane(row16_max(mx[k]), 48));
                lo_k = min(lo_k, lo2);
                hi_k = max(hi_k, hi2);
            }
            mn[k] = __builtin_amdgcn_readfirstlane(lo_k);
            mx[k] = __builtin_amdgcn_readfirstlane(hi_k);
        }
        // finest level whose box has <= WT_MAXC cells and <= 64 blocks (wave-uniform: scalar registers)
        int level = -1, blo0 = 0, blo1 = 0, blo2 = 0, d0 = 0, d1 = 0, d2 = 0;
        const int n_levels = gv.levels;
        for (int l = 0; l < n_levels; ++l) {   // all operands are wave-uniform 32-bit integers: scalar ALU
            const int b0 = mn[0] >> (2 * l), b1 = mn[1] >> (2 * l), b2 = mn[2] >> (2 * l);
            const int e0 = (mx[0] >> (2 * l)) - b0 + 1, e1 = (mx[1] >> (2 * l)) - b1 + 1, e2 = (mx[2] >> (2 * l)) - b2 + 1;
            if (e0 > WT_MAXC || e1 > WT_MAXC || e2 > WT_MAXC || e0 * e1 > WT_MAXC || e0 * e1 * e2 > WT_MAXC) continue;
            const int n0 = ((b0 + e0 - 1) >> 1) - (b0 >> 1) + 1, n1 = ((b1 + e1 - 1) >> 1) - (b1 >> 1) + 1, n2 = ((b2 + e2 - 1) >> 1) - (b2 >> 1) + 1;
            if (n0 * n1 * n2 > 64) continue;
            level = l; blo0 = b0; blo1 = b1; blo2 = b2; d0 = e0; d1 = e1; d2 = e2;
            break;
        }
        level = __builtin_amdgcn_readfirstlane(level);
        if (level < 0) { if (part) WT_WHY(1); break; }
        blo0 = __builtin_amdgcn_readfirstlane(blo0); blo1 = __builtin_amdgcn_readfirstlane(blo1); blo2 = __builtin_amdgcn_readfirstlane(blo2);
        d0 = __builtin_amdgcn_readfirstlane(d0); d1 = __builtin_amdgcn_readfirstlane(d1); d2 = __builtin_amdgcn_readfirstlane(d2);
        ++dbg_passes;
        WT_STAMP(1);
        // ---- directory: one 2x2x2 block per lane (<= 64 blocks by the level choice)
        unsigned int ncell = 0;
        bool big_child = false;
        {
            const int lim = (int)(PCR_COORD_MAX >> (2 * level));
            const int bx0 = blo0 >> 1, by0 = blo1 >> 1, bz0 = blo2 >> 1;
            const int nb0 = ((blo0 + d0 - 1) >> 1) - bx0 + 1, nb1 = ((blo1 + d1 - 1) >> 1) - by0 + 1, nb2 = ((blo2 + d2 - 1) >> 1) - bz0 + 1;
            const int nb01 = nb0 * nb1, nblk = nb01 * nb2;
            pcr_block_slot e;
            bool have = false;
            int BX = 0, BY = 0, BZ = 0;
            if (lane < nblk) {
                // block index -> (ix, iy, iz) without an integer division (lane < 64: the float quotient is fixed up)
                // (v_rcp_f32: within 1 ulp, and the quotient is fixed up either way; a true division is a 12-instruction sequence)
                int iz = (int)((float)lane * __builtin_amdgcn_rcpf((float)nb01));
                iz -= (iz * nb01 > lane);
                iz += ((iz + 1) * nb01 <= lane);
                const int rem = lane - iz * nb01;
                int iy = (int)((float)rem * __builtin_amdgcn_rcpf((float)nb0));
                iy -= (iy * nb0 > rem);
                iy += ((iy + 1) * nb0 <= rem);
                const int ix = rem - iy * nb0;
                BX = bx0 + ix; BY = by0 + iy; BZ = bz0 + iz;
                const int blim = lim >> 1;
                if (BX >= 0 && BY >= 0 && BZ >= 0 && BX <= blim && BY <= blim && BZ <= blim)
                    have = lookup_block(as_global(gv.btable[level]), gv.bmask[level], (unsigned int)BX, (unsigned int)BY, (unsigned int)BZ, &e);
            }
            // occupied children inside the box -> compact cell list (start, count) in LDS, in slot order.  A block with a child too
            // large for its 16-bit count (> 65 535 points in one cell: duplicates) makes the tile give up below -- its queries go to
            // the exact descent, which splits such cells 64 ways; asking the cell table here instead cost eight unrolled four-load
            // probes of code and registers for a case no scan ever produces
            big_child = __any(have && e.flags != 0);
            unsigned int run = have ? e.start : 0u;
#pragma unroll
            for (int ch = 0; ch < 8; ++ch) {
                const unsigned int cs = run, cn = have ? (unsigned int)e.cnt[ch] : 0u;
                run += cn;
                const int X = 2 * BX + (ch & 1) - blo0, Y = 2 * BY + ((ch >> 1) & 1) - blo1, Z = 2 * BZ + (ch >> 2) - blo2;
                const bool in = have && cn > 0 && X >= 0 && Y >= 0 && Z >= 0 && X < d0 && Y < d1 && Z < d2;
                const unsigned long long m = __ballot(in);
                if (in) {
                    const unsigned int slot = ncell + __popcll(m & ((1ull << lane) - 1ull));
                    L->c_start[slot] = cs;
                    L->c_off[slot] = cn;
                }
                ncell += __popcll(m);
            }
        }
        // (the one-launch ICP pass wants device-scope words read here, a few microseconds into the launch -- "have all tiles of the
        // launch started?": the started count of every group -- without paying a round trip for them at the end of the tile)
        if (probe_p && pass == 0 && lane < (int)probe_groups)   // one word per group, compared after the tile: nobody waits for it here
            probe_v = __hip_atomic_load(probe_p + (size_t)probe_stride * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_sync();
        WT_STAMP(2);
        // ---- exclusive prefix of the cell counts in slot order (chunks of 64 cells, running total carried along)
        unsigned int total = 0;
        for (unsigned int c0 = 0; c0 < ncell; c0 += 64) {
            const unsigned int c = c0 + lane;
            const unsigned int cn = c < ncell ? L->c_off[c] : 0u;
            unsigned int t = 0;
            const unsigned int pre = wave_excl_scan_u32(cn, lane, &t);
            if (c < ncell) L->c_off[c] = total + pre;
            total += t;
        }
        if (big_child) total = pcap + 1u;   // (more than the cap: the tile gives up)
        wave_sync();
        WT_STAMP(3);
        staged += total;
        // (one-launch ICP pass: a tile that turns out heavy takes issue priority for the rest of its stage -- the caller already gave
        // it what last pass's load suggested; this also covers the unseeded first pass, which has no history: 92 -> 87 us)
        if (probe_groups) {
            if (staged > 3 * WT_PR) __builtin_amdgcn_s_setprio(3);
            else if (staged > 2 * WT_PR) __builtin_amdgcn_s_setprio(2);
            else if (staged > WT_PR) __builtin_amdgcn_s_setprio(1);
        }
        if (total > pcap) {
            // Too many points to stage around this tile.  Unseeded passes: "cells + one ring" around 32 queries in a dense part of the
            // scan (the first pass of an ICP left 329 of 3 752 tiles like that, 9 109 of its 9 116 queue items, and every further pass
            // of such a tile claimed MORE) -- the next pass claims the queries' own cells only: almost every query finds a candidate
            // centimetres away there, and the pass after that stages the tight balls.  If even that is too much, the tile stops
            // at once and the queue takes over (seeded passes: their claims are tight balls already).
            if (part) WT_WHY(2);
            if (prev_xyz || ring == 0.0f) break;
            ring = 0.0f;
            continue;
        }
        staged_done += total;
        if (dbg) dbg_pairs += (unsigned long long)total * (unsigned long long)__popcll(__ballot(part && lane < WT_Q));
        const float cellLf = (float)gv.cell0 * (float)(1 << (2 * level));
        const double cellL = gv.cell0 * (double)(1ll << (2 * level));
        const int blL = (int)(PCR_COORD_BIAS >> (2 * level));
        const double ox = gv.lo[0] + ((double)(blo0 - blL) + 0.5 * d0) * cellL;
        const double oy = gv.lo[1] + ((double)(blo1 - blL) + 0.5 * d1) * cellL;
        const double oz = gv.lo[2] + ((double)(blo2 - blL) + 0.5 * d2) * cellL;
        const float qxf = (float)(ax - ox), qyf = (float)(ay - oy), qzf = (float)(az - oz);
        float fm = INFINITY, fs = INFINITY, s_in = INFINITY;
        unsigned int bpos = POS_NONE;
#if PCR_WT_MFMA
        static_assert(WT_Q == 32, "the matrix-core filter maps the 32 queries of a tile to the 32 columns of a tile product");
        typedef float f16v __attribute__((ext_vector_type(16)));
        const float mf_b1 = lane < 32 ? -2.0f * qxf : -2.0f * qyf, mf_b2 = lane < 32 ? -2.0f * qzf : 1.0f;
        float bd = INFINITY;   // direct-form squared distance of this lane's best point so far
#endif
        if (total > 0) {
#if !PCR_WT_MFMA
            const f2 qx2 = {qxf, qxf}, qy2 = {qyf, qyf}, qz2 = {qzf, qzf};
            const int slice = lane / WT_Q;
#endif
            unsigned int carry = 0;  // owner cell of the last position of the previous round
            // Addresses and loads of one round (positions [b, b + n) of the box): owner cell of every staged position -- mark the
            // first position of each cell (slot ids increase with the position, cells are non-empty), then a running maximum over
            // the positions -- and the (up to) three target records of this lane, requested back to back: one memory round trip.
            // (Every element of R is assigned on every path: left conditionally unassigned, the nine doubles became loop-carried
            // values -- 18 registers alive through the filter and the directory: 102 -> 87 VGPRs for the stand-alone tile kernel.)
            auto request = [&](const unsigned int b, const unsigned int n, unsigned int (&J)[WT_CH], wt_xyz (&R)[WT_CH]) {
#pragma unroll
                for (int c3 = 0; c3 < WT_CH; ++c3) L->own[64 * c3 + lane] = 0;
                wave_sync();
                for (unsigned int c = lane; c < ncell; c += 64) {
                    const unsigned int f = L->c_off[c];
                    if (f >= b && f < b + n) L->own[f - b] = (unsigned short)c;
                }
                wave_sync();
                unsigned int ow[WT_CH];
#pragma unroll
                for (int c3 = 0; c3 < WT_CH; ++c3) ow[c3] = (unsigned int)L->own[64 * c3 + lane];
#pragma unroll
                for (int c3 = 0; c3 < WT_CH; ++c3) {
                    unsigned int v = max(wave_incl_scan_max(ow[c3]), carry);
                    carry = (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
                    const unsigned int k = 64 * c3 + lane;
                    const unsigned int vv = k < n ? v : 0u;
                    J[c3] = L->c_start[vv] + (b + k - L->c_off[vv]);
                }
#pragma unroll
                for (int c3 = 0; c3 < WT_CH; ++c3) {
                    R[c3] = wt_xyz{0.0, 0.0, 0.0};
                    // (a GLOBAL load, spelled out: a flat load also counts on the LDS counter -- every LDS wait of the filter would wait
                    // for the records in flight)
                    if (64 * c3 + lane < n) R[c3] = ld_xyz_global(&g_pts[J[c3]]);
                }
            };
            // The rounds are software-pipelined: the records of round r + 1 are requested BEFORE the filter of round r runs, so a tile
            // of several rounds -- the launch's last finishers -- pays one gather round trip, not one per round.
            unsigned int jj[WT_CH];
            wt_xyz rec[WT_CH];
            request(0u, min(total, (unsigned int)WT_PR), jj, rec);
#pragma unroll 1
            for (unsigned int base = 0; base < total; base += WT_PR) {
                const unsigned int cnt = min(total - base, (unsigned int)WT_PR);
#pragma unroll
                for (int c3 = 0; c3 < WT_CH; ++c3) {
                    const unsigned int k = 64 * c3 + lane;
                    if (k < cnt) {
                        const float fx = (float)(rec[c3].x - ox), fy = (float)(rec[c3].y - oy), fz = (float)(rec[c3].z - oz);
                        L->px[k] = fx; L->py[k] = fy; L->pz[k] = fz;
#if PCR_WT_MFMA
                        L->pn[k] = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
#endif
                        L->ppos[k] = jj[c3];
                    }
                }
#if PCR_WT_MFMA
                // pad the last block of 32 with points nobody can win (finite: no inf - inf in the expanded form)
                if (lane < 32 && cnt + lane < ((cnt + 31u) & ~31u)) { L->px[cnt + lane] = 1e15f; L->py[cnt + lane] = 0.0f; L->pz[cnt + lane] = 0.0f; L->pn[cnt + lane] = 1e30f; }
#else
                if (lane < 8) { L->px[cnt + lane] = 1e30f; L->py[cnt + lane] = 0.0f; L->pz[cnt + lane] = 0.0f; }  // pad the last group of 8
#endif
                wave_sync();
                unsigned int jn[WT_CH];
                wt_xyz rn[WT_CH];
#pragma unroll
                for (int c3 = 0; c3 < WT_CH; ++c3) { jn[c3] = 0u; rn[c3] = wt_xyz{0.0, 0.0, 0.0}; }
                if (base + WT_PR < total) request(base + WT_PR, min(total - base - WT_PR, (unsigned int)WT_PR), jn, rn);   // in flight during the filter
                WT_STAMP(4);
#if PCR_WT_MFMA
                // Filter on the matrix cores.  Expanded form |p|^2 - 2 q.p (the query's own |q|^2 is added at the end): one 32 x 32
                // tile = 32 staged points (rows) x the 32 queries (columns), K = 4 = two v_mfma_f32_32x32x2_f32:
                //   A rows (px, py | pz, |p|^2): lane l supplies point l % 32, k = l / 32 -- two 4-byte LDS reads per lane and block
                //   B cols (-2qx, -2qy | -2qz, 1): two registers per lane for the whole pass
                //   C: lane l holds query l % 32 against the 16 points 8 g + 4 (l / 32) + e of the block (g, e = 0..3)
                // so the two lanes of a query see half a block each -- the two candidate slices of the packed-VALU filter it replaces,
                // which read 96 bytes of LDS per lane for every 8 points and kept the VALU busy with 35 instructions for them (the
                // filter was half of a heavy tile's lifetime, bound by LDS return bandwidth and VALU issue of the whole CU).
                // The tile values only ORDER blocks: the winner inside the best block is found by the direct form below, and every
                // other block enters the proof through its minimum minus the bound d_arith on the expanded form's rounding error.
                {
                    int rk = -1;
                    const unsigned int nblk = (cnt + 31u) >> 5;
                    const float* const arow1 = lane < 32 ? L->px : L->py;
                    const float* const arow2 = lane < 32 ? L->pz : L->pn;
#pragma unroll 1
                    for (unsigned int blk = 0; blk < nblk; ++blk) {
                        const float a1 = arow1[32 * blk + (lane & 31)], a2 = arow2[32 * blk + (lane & 31)];
                        f16v acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, mf_b1, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, mf_b2, acc, 0, 0, 0);
                        float m = fmin3(acc[0], acc[1], acc[2]);
                        m = fmin3(m, acc[3], acc[4]);
                        m = fmin3(m, acc[5], acc[6]);
                        m = fmin3(m, acc[7], acc[8]);
                        m = fmin3(m, acc[9], acc[10]);
                        m = fmin3(m, acc[11], acc[12]);
                        m = fmin3(m, acc[13], acc[14]);
                        m = fminf(m, acc[15]);
                        const bool lt = part && m < fm;
                        if (part) {
                            fs = __builtin_amdgcn_fmed3f(m, fm, fs);
                            fm = __builtin_amdgcn_fmed3f(m, fm, -INFINITY);
                        }
                        rk = lt ? (int)blk : rk;
                    }
                    if (rk >= 0) {
                        // the 16 points this lane saw in its best block, by the direct form
                        float best = INFINITY, second = INFINITY;
                        int bi = 0;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int k = 32 * rk + 8 * g + 4 * (lane >> 5);
                            const f4 bx4 = *reinterpret_cast<const f4*>(&L->px[k]);
                            const f4 by4 = *reinterpret_cast<const f4*>(&L->py[k]);
                            const f4 bz4 = *reinterpret_cast<const f4*>(&L->pz[k]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float ddx = qxf - bx4[e], ddy = qyf - by4[e], ddz = qzf - bz4[e];
                                const float dj = __builtin_fmaf(ddz, ddz, __builtin_fmaf(ddy, ddy, ddx * ddx));
                                if (dj < best) { second = best; best = dj; bi = k + e; }
                                else if (dj < second) second = dj;
                            }
                        }
                        if (best < bd) {
                            s_in = fminf(s_in, fminf(bd, second));
                            bd = best;
                            bpos = L->ppos[bi];
                        } else s_in = fminf(s_in, best);
                    }
                }
#else
                // filter: slice s takes groups s, s+4, ... of 8 staged points (see grid_tile_kernel for the arithmetic)
                if (part) {
                    int rk = -1;
                    for (unsigned int k0 = slice * 8; k0 < cnt; k0 += 8 * (64 / WT_Q)) {
                        f2 d[4];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            const unsigned int k = k0 + 4 * u;
                            const f4 bx4 = *reinterpret_cast<const f4*>(&L->px[k]);
                            const f4 by4 = *reinterpret_cast<const f4*>(&L->py[k]);
                            const f4 bz4 = *reinterpret_cast<const f4*>(&L->pz[k]);
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const f2 bx = h ? f2{bx4.z, bx4.w} : f2{bx4.x, bx4.y};
                                const f2 by = h ? f2{by4.z, by4.w} : f2{by4.x, by4.y};
                                const f2 bz = h ? f2{bz4.z, bz4.w} : f2{bz4.x, bz4.y};
                                const f2 dx = qx2 - bx, dy = qy2 - by, dz = qz2 - bz;
                                f2 t = dx * dx;
                                t = __builtin_elementwise_fma(dy, dy, t);
                                d[2 * u + h] = __builtin_elementwise_fma(dz, dz, t);
                            }
                        }
                        float m8 = fmin3(d[0].x, d[0].y, d[1].x);
                        m8 = fmin3(m8, d[1].y, d[2].x);
                        m8 = fmin3(m8, d[2].y, d[3].x);
                        m8 = fmin3(m8, d[3].y, d[3].y);
                        const bool lt = m8 < fm;
                        fs = __builtin_amdgcn_fmed3f(m8, fm, fs);
                        fm = __builtin_amdgcn_fmed3f(m8, fm, -INFINITY);
                        rk = lt ? (int)k0 : rk;
                    }
                    if (rk >= 0) {
                        float best = INFINITY, second = INFINITY;
                        int bi = rk;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float ddx = qxf - L->px[rk + j], ddy = qyf - L->py[rk + j], ddz = qzf - L->pz[rk + j];
                            const float dj = __builtin_fmaf(ddz, ddz, __builtin_fmaf(ddy, ddy, ddx * ddx));
                            if (dj < best) { second = best; best = dj; bi = rk + j; }
                            else if (dj < second) second = dj;
                        }
                        bpos = L->ppos[bi];
                        s_in = second;
                    }
                }
#endif
                wave_sync();
#pragma unroll
                for (int c3 = 0; c3 < WT_CH; ++c3) { jj[c3] = jn[c3]; rec[c3] = rn[c3]; }
                WT_STAMP(5);
            }
#if PCR_WT_MFMA
            {
                // from here on fm = the winner's direct-form value, fs = a lower bound of every other staged point's value: the blocks'
                // minima come from the expanded form in binary32 -- |q|^2, |p|^2 (3 roundings each), 4 multiply-adds (counted as 8
                // roundings, in case they are not fused), 1 addition, all on magnitudes <= (|q| + |p|)^2 <= 12 R^2 -- within
                // 126 u R^2 of the exact value for the rounded coordinates (u = 2^-24); 160 u R^2 is subtracted
                const float Rm = 0.5f * cellLf * (float)max(d0, max(d1, d2));
                const float d_arith = 160.0f * 5.9604644775390625e-08f * Rm * Rm * (1.0f + 1e-6f);
                const float qn = __builtin_fmaf(qzf, qzf, __builtin_fmaf(qyf, qyf, qxf * qxf));
                fs = fminf((fs + qn) - d_arith, s_in);
                fm = bd;
            }
#else
            fs = fminf(fs, s_in);
#endif
        }
        // ---- merge the four candidate slices of every query
#pragma unroll
        for (int off = WT_Q; off < 64; off <<= 1) {
            const float om = __shfl_xor(fm, off, 64), os = __shfl_xor(fs, off, 64);
            const unsigned int op = __shfl_xor(bpos, off, 64);
            fs = fminf(fmaxf(fm, om), fminf(fs, os));
            if (om < fm) { fm = om; bpos = op; }
        }
        if (part) {
            // binary32 with outward rounding throughout: U bounds the winner's true squared distance from above
            float U = INFINITY;
            bool ambiguous = false;
            if (bpos != POS_NONE) {
                // coordinate error of the filter: |local coordinate| <= R, binary32 conversion + subtraction
                const float R = 0.5f * cellLf * (float)max(d0, max(d1, d2));
                const float e = 8.0f * 5.9604644775390625e-08f * R * (1.0f + 1e-6f);
                const float dhi = sqrtf(fm) * (1.0f + 2e-6f) + e;
                U = (dhi + e) * (dhi + e) * (1.0f + 2e-6f);
                ambiguous = fs <= U;
            }
            // distance from the query to the boundary of the staged box, in the filter's own local coordinates
            // (error of a local coordinate <= e/8 * 2, covered by the slack)
            float db = fminf(fminf(0.5f * d0 * cellLf - fabsf(qxf), 0.5f * d1 * cellLf - fabsf(qyf)), 0.5f * d2 * cellLf - fabsf(qzf));
            db = fmaxf(db - cellLf * (float)max(d0, max(d1, d2)) * 4e-6f, 0.0f);
            const float reach2 = fminf(U, bound2);   // the nearest neighbour (if within the gate) lies within this squared radius
            if (!ambiguous && reach2 <= db * db * (1.0f - 2e-6f)) {
                // proven: the filter's winner is the exact nearest neighbour, or nothing lies within the gate
                open = false;
                won = bpos;
                if (lane < WT_Q) {
                    if (res_d2) {
                        double dd = DBL_MAX;
                        if (bpos != POS_NONE) {
                            const pcr_pt bb = g_pts[bpos];
                            dd = dist2(ax, ay, az, bb);   // exact, direct form (the nn1 API reports it)
                        }
                        res_d2[qi] = dd;
                    }
                    if (!prev_xyz || bpos != seed_pos) res_pos[qi] = bpos;   // (seeded passes: most neighbours do not change)
                }
            } else {
                WT_WHY(ambiguous ? 3 : 4);
                if (bpos != POS_NONE && U < bound2) {
                    bound2 = U;       // a real candidate: the next pass (or the hard stage) searches inside its ball
                    cand_pos = bpos;
                }
            }
        }
        ring = ring == 0.0f ? 1.0f : ring * 4.0f;   // queries still without a candidate claim a wider cube next
    }
    WT_STAMP(6);
    S.ax = ax; S.ay = ay; S.az = az;
    S.qi = qi;
    S.bound2 = bound2;
    S.cand_pos = cand_pos;
    S.won = won;
    S.open = open; S.clamped = clamped; S.qvalid = qvalid;
    S.staged = staged;
    S.dbg_pairs = dbg_pairs; S.dbg_passes = dbg_passes;
    {   // probe: every group's count complete?  (group l has total / groups tiles, the first total % groups groups one more)
        const bool ok = lane < (int)probe_groups && (unsigned int)probe_v == probe_total / (probe_groups ? probe_groups : 1u) + ((unsigned int)lane < probe_total % (probe_groups ? probe_groups : 1u) ? 1u : 0u);
        S.probe = (probe_groups && __ballot(ok) == (probe_groups >= 64 ? ~0ull : (1ull << probe_groups) - 1ull)) ? 1ull : 0ull;
    }
#ifdef PCR_WT_DIAG
    if (dbg) {
        const bool unres = open && lane < WT_Q;
        for (int r = 0; r < 5; ++r) {
            const unsigned int n_r = (unsigned int)__popcll(__ballot(unres && (clamped ? 0 : why) == r));
            if (lane == 0 && n_r) atomicAdd(&dbg[(1 << 15) + r], (unsigned long long)n_r);
        }
        if (lane == 0) {   // every wave's phases; slot 7 = points staged by the tile
            t_ph[7] = staged;
            for (int i = 0; i < 8; ++i) dbg[(1 << 16) + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + i] = t_ph[i];
        }
    }
#endif
}

// tile index of a wave: every XCD (own L2) gets one contiguous run of the Morton-sorted queries
__device__ static inline unsigned int wtile_block(int xcd_remap) {
    unsigned int blk = blockIdx.x;
    if (xcd_remap) {
        const unsigned int per = gridDim.x >> 3, main = per << 3;
        if (blk < main) blk = (blk & 7u) * per + (blk >> 3);
    }
    return blk;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCR_WT_WAVES, 8)))
grid_wtile_kernel(const pcr_grid_view* __restrict__ gvp, pcr_pt* __restrict__ q, long long nq, pcr_xform x, int has_x, int write_back, double max_d2,
                  int gated, int xcd_remap, unsigned int pcap, unsigned int* __restrict__ res_pos, double* __restrict__ res_d2,
                  work_item* __restrict__ hard_list, unsigned int* __restrict__ hard_count, unsigned long long* __restrict__ dbg,
                  const pcr_icp_dev_state* __restrict__ st, const wt_xyz* __restrict__ prev_xyz) {
    __shared__ wtile_lds s_lds[4];
    if (st) {
        if (st->stop) return;
        x = st->x;
    }
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memtime() : 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int blk = wtile_block(xcd_remap);
    wt_state S;
    wt_pre P;
    wtile_preload(blk * 4 + wave, lane, q, nq, res_pos, prev_xyz, P);
    wtile_search(*gvp, &s_lds[wave], blk * 4 + wave, lane, P, q, nq, x, has_x, write_back, max_d2, gated, pcap, res_pos, res_d2, dbg, prev_xyz, S);
    // ---- what is still open goes to the hard stage: one append per wave
    const bool unres = S.open && lane < WT_Q;
    const unsigned long long m = __ballot(unres);
    if (m) {
        const unsigned int n_unres = __popcll(m);
        unsigned int base = 0;
        const unsigned int hl = (blk * 4 + wave) % H_NLIST;
        if (lane == 0) base = hl * hard_list_cap(nq) + atomicAdd(hard_count + H_CSTRIDE * hl, n_unres);
        base = __shfl(base, 0, 64);
        if (unres) {
            work_item it;
            it.ax = S.ax; it.ay = S.ay; it.az = S.az;
            it.best_d2 = (S.clamped || S.cand_pos == POS_NONE) ? DBL_MAX : (double)S.bound2;
            it.best_pos = S.clamped ? POS_NONE : S.cand_pos;
            it.qi = (unsigned int)S.qi;
            hard_list[base + __popcll(m & ((1ull << lane) - 1ull))] = it;
        }
    }
    if (dbg && lane == 0) {
        atomicAdd(&dbg[blockIdx.x * 4 + 1], S.dbg_pairs);
        atomicAdd(&dbg[blockIdx.x * 4 + 2], (unsigned long long)S.dbg_passes);
        atomicAdd(&dbg[blockIdx.x * 4 + 3], (unsigned long long)__popcll(m));
        atomicMax(&dbg[blockIdx.x * 4 + 0], __builtin_amdgcn_s_memtime() - t_start);
    }
}

// ------------------------------------------------------------- moments
// one gated correspondence into the 19 running Procrustes moments about `origin` (same arithmetic everywhere it is used)
__device__ static inline void moments_add(double* __restrict__ m, const double origin[3], double ax, double ay, double az, const pcr_pt& b,
                                          double max_d2, int gated) {
    const double d2 = dist2(ax, ay, az, b);
    if (gated && !(d2 < max_d2)) return;
    const double a0 = ax - origin[0], a1 = ay - origin[1], a2 = az - origin[2];
    const double b0 = b.x - origin[0], b1 = b.y - origin[1], b2 = b.z - origin[2];
    m[0] += 1.0;
    m[1] += a0; m[2] += a1; m[3] += a2;
    m[4] += b0; m[5] += b1; m[6] += b2;
    m[7] += b0 * a0; m[8] += b0 * a1; m[9] += b0 * a2;
    m[10] += b1 * a0; m[11] += b1 * a1; m[12] += b1 * a2;
    m[13] += b2 * a0; m[14] += b2 * a1; m[15] += b2 * a2;
    m[16] += (a0 * a0 + a1 * a1) + a2 * a2;
    m[17] += (b0 * b0 + b1 * b1) + b2 * b2;
    m[18] += d2;
}

// -------------------------------------------------------------- hard stage
struct hard_entry {
    unsigned int start, end;
    unsigned int x, y, z;
    int level;
};

__device__ static inline double box_dist2(const pcr_grid_view& gv, int level, double cell, unsigned int X, unsigned int Y, unsigned int Z,
                                          double ax, double ay, double az) {
    const int bl = (int)(PCR_COORD_BIAS >> (2 * level));
    const double slack = cell * 1e-9;
    const double x0 = gv.lo[0] + (double)((int)X - bl) * cell;
    const double y0 = gv.lo[1] + (double)((int)Y - bl) * cell;
    const double z0 = gv.lo[2] + (double)((int)Z - bl) * cell;
    const double dx = sq_pos(fmax(x0 - ax, ax - (x0 + cell)) - slack);
    const double dy = sq_pos(fmax(y0 - ay, ay - (y0 + cell)) - slack);
    const double dz = sq_pos(fmax(z0 - az, az - (z0 + cell)) - slack);
    return (dx + dy) + dz;
}

__device__ static inline double wave_min(double v) { return wave_min_f64(v); }   // (DPP network: pcr_grid_dev.h)

struct hard_lds {
    hard_entry stack[HARD_STACK];
    unsigned int fl_off[64], fl_start[64];  // flattened scan directory: exclusive point offset / start of every small cell
};

// One step of the descent.  Every lane holds (at most) one candidate cell.  Cells that are small (or
// cannot be split further) are read NOW, all together: their ranges are concatenated and the 64 lanes
// stride over the concatenation, so the reads of a step are independent of each other (one memory round
// trip instead of one per cell).  Cells that are still big are pushed for a later split, nearest last.
__device__ static inline void hard_disperse(const pcr_grid_view& gv, hard_lds* L, int& sp, int lane, bool valid, unsigned int s, unsigned int e,
                                            unsigned int X, unsigned int Y, unsigned int Z, int lvl, double bdist, double ax, double ay,
                                            double az, double& bd2, long long& bid, unsigned int& bpos, double& bound2, unsigned int& n_pts, double* bw) {
    const unsigned int cnt = e - s;
    const bool small = valid && (lvl == 0 || cnt <= HARD_SCAN_T);
    const unsigned long long m_small = __ballot(small);
    if (m_small) {
        // exclusive prefix of the small cells' sizes over the lanes
        const unsigned int inc = wave_incl_scan_add(small ? cnt : 0u);
        const unsigned int total = (unsigned int)__builtin_amdgcn_readlane((int)inc, 63);
        const int n_small = __popcll(m_small);
        if (small) {
            const int r = __popcll(m_small & ((1ull << lane) - 1ull));
            L->fl_off[r] = inc - cnt;
            L->fl_start[r] = s;
        }
        n_pts += total;
        for (unsigned int t0 = lane; t0 < total; t0 += 128) {
            // two independent reads per trip
            unsigned int jj[2];
            bool ok[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned int t = t0 + 64 * u;
                ok[u] = t < total;
                int lo = 0, hi = n_small - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (L->fl_off[mid] <= t) lo = mid;
                    else hi = mid - 1;
                }
                jj[u] = L->fl_start[lo] + (t - L->fl_off[lo]);
            }
            pcr_pt b0 = pcr_pt{0.0, 0.0, 0.0, 0}, b1 = pcr_pt{0.0, 0.0, 0.0, 0};   // (assigned on every path: see the staging rounds of the tile)
            if (ok[0]) b0 = gv.pts[jj[0]];
            if (ok[1]) b1 = gv.pts[jj[1]];
            if (ok[0]) {
                const double d2 = dist2(ax, ay, az, b0);
                if (better(d2, b0.id, bd2, bid)) { bd2 = d2; bid = b0.id; bpos = jj[0]; bw[0] = b0.x; bw[1] = b0.y; bw[2] = b0.z; }
            }
            if (ok[1]) {
                const double d2 = dist2(ax, ay, az, b1);
                if (better(d2, b1.id, bd2, bid)) { bd2 = d2; bid = b1.id; bpos = jj[1]; bw[0] = b1.x; bw[1] = b1.y; bw[2] = b1.z; }
            }
        }
        bound2 = fmin(bound2, wave_min(bd2));
    }
    // big cells: push (far ones first so that the nearest is split first); without room they are read whole
    const bool big = valid && !small && bdist <= bound2;
    const unsigned long long m_far = __ballot(big && bdist > 0.0), m_near = __ballot(big && !(bdist > 0.0));
    const int n_big = __popcll(m_far) + __popcll(m_near);
    if (n_big == 0) return;
    if (sp + n_big <= HARD_STACK) {
        const unsigned long long below = (1ull << lane) - 1ull;
        int slot = -1;
        if (big && bdist > 0.0) slot = __popcll(m_far & below);
        else if (big) slot = __popcll(m_far) + __popcll(m_near & below);
        if (slot >= 0) {
            hard_entry en;
            en.start = s; en.end = e; en.x = X; en.y = Y; en.z = Z; en.level = lvl;
            L->stack[sp + slot] = en;
        }
        sp += n_big;
    } else {
        unsigned long long mb = m_far | m_near;
        while (mb) {
            const int src = __ffsll((long long)mb) - 1;
            mb &= mb - 1;
            const unsigned int ss = __shfl(s, src, 64), ee = __shfl(e, src, 64);
            scan_range(gv.pts, ss + lane, ee, 64, ax, ay, az, bd2, bid, bpos, bw);
            n_pts += ee - ss;
        }
        bound2 = fmin(bound2, wave_min(bd2));
    }
}

// Searches one query exactly: pruned descent of the nested cell hierarchy by the 64 lanes of the calling wave.
// `bound2` bounds the search from above (gate and / or the tile stage's candidate, DBL_MAX = nothing); `have_cand` says a
// real point lies inside it.  On return every lane holds the same (bd2, bid, bpos); POS_NONE = nothing inside the bound.
__device__ __forceinline__ static void hard_search(const pcr_grid_view& gv, hard_lds* L, const int lane, const double ax, const double ay, const double az,
                                                   double bound2, bool have_cand, double& bd2, long long& bid, unsigned int& bpos,
                                                   unsigned int& h_steps, unsigned int& h_pts, int& s_level0, double* win = nullptr) {
    double bw[3] = {0.0, 0.0, 0.0};   // coordinates of this lane's best point (the scans have the record in hand: no gather for them later)
    const int top = gv.levels - 1;
    bd2 = DBL_MAX;
    bid = ID_NONE;
    bpos = POS_NONE;
    bool clamped = false;
    const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
    const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
    const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
    if (!clamped && !have_cand) {
        // nothing known yet (tile too large to stage): the query's own level-0 cell gives a first bound
        unsigned int s, e;
        if (lookup_cell(gv.table[0], gv.mask[0], (unsigned int)cx, (unsigned int)cy, (unsigned int)cz, &s, &e)) {
            scan_range(gv.pts, s + lane, e, 64, ax, ay, az, bd2, bid, bpos, bw);
            bound2 = fmin(bound2, wave_min(bd2));
        }
    }
    // Level schedule.  With a real candidate in hand the search starts at the smallest level whose
    // 3x3x3 block covers the bound ball.  Without one (only the gate, or nothing, bounds the search)
    // it starts at level 0 and grows: the first candidate found shrinks the ball, usually long before
    // the block would have to cover the whole gate radius in a dense part of the scan.
    auto level_for = [&](double b2, int from) {
        double c = gv.cell0 * (double)(1ll << (2 * from));
        for (int l = from; l <= top; ++l) {
            const double safe = c * (1.0 - 1e-9);
            if (safe * safe >= b2) return l;
            c *= 4.0;
        }
        return top + 1;  // not even the top level's block covers the ball
    };
    have_cand = have_cand || __any(bd2 < DBL_MAX);
#ifndef PCR_HARD_START_MAX
#define PCR_HARD_START_MAX 99
#endif
    // (PCR_HARD_START_MAX: the search never STARTS above this level even when the candidate's ball asks for it -- in a dense cloud the
    // true neighbour is centimetres away while the candidate, last pass's neighbour seen from the moved query, may be decimetres
    // away: the block at a fine level shrinks the ball before the coarse cells it would have touched are split one by one)
    int s_level = clamped ? top + 1 : (have_cand ? level_for(bound2, 0) : 0);
    if (s_level <= top && s_level > PCR_HARD_START_MAX) s_level = PCR_HARD_START_MAX;
    s_level0 = s_level;
    int sp = 0;  // wave-uniform stack pointer
    while (s_level <= top) {
        const int lvl = s_level;
        const double cell = gv.cell0 * (double)(1ll << (2 * lvl));
        bool valid = lane < 27;
        const int X = (cx >> (2 * lvl)) + (lane % 3 - 1), Y = (cy >> (2 * lvl)) + ((lane / 3) % 3 - 1), Z = (cz >> (2 * lvl)) + (lane / 9 - 1);
        const int lim = (int)(PCR_COORD_MAX >> (2 * lvl));
        valid = valid && X >= 0 && Y >= 0 && Z >= 0 && X <= lim && Y <= lim && Z <= lim;
        unsigned int s = 0, e = 0;
        double bdist = 0.0;
        if (valid) {
            bdist = box_dist2(gv, lvl, cell, (unsigned int)X, (unsigned int)Y, (unsigned int)Z, ax, ay, az);
            valid = bdist <= bound2 && lookup_cell(gv.table[lvl], gv.mask[lvl], (unsigned int)X, (unsigned int)Y, (unsigned int)Z, &s, &e);
        }
        hard_disperse(gv, L, sp, lane, valid, s, e, (unsigned int)X, (unsigned int)Y, (unsigned int)Z, lvl, bdist, ax, ay, az, bd2, bid, bpos,
                      bound2, h_pts, bw);
        ++h_steps;
        while (sp > 0) {
            --sp;
            const hard_entry en = L->stack[sp];  // same address in every lane: LDS broadcast
            const double ecell = gv.cell0 * (double)(1ll << (2 * en.level));
            if (box_dist2(gv, en.level, ecell, en.x, en.y, en.z, ax, ay, az) > bound2) continue;
            // split: one child per lane, box test against the bound, probe
            const int cl = en.level - 1;
            const unsigned int CX = en.x * 4u + (lane & 3), CY = en.y * 4u + ((lane >> 2) & 3), CZ = en.z * 4u + (lane >> 4);
            const double cdist = box_dist2(gv, cl, ecell * 0.25, CX, CY, CZ, ax, ay, az);
            unsigned int cs = 0, ce = 0;
            const bool cvalid = cdist <= bound2 && lookup_cell(gv.table[cl], gv.mask[cl], CX, CY, CZ, &cs, &ce);
            hard_disperse(gv, L, sp, lane, cvalid, cs, ce, CX, CY, CZ, cl, cdist, ax, ay, az, bd2, bid, bpos, bound2, h_pts, bw);
            ++h_steps;
        }
        const double safe = cell * (1.0 - 1e-9);
        if (safe * safe >= bound2) break;  // the block just searched covers the bound ball: exact
        s_level = (bound2 < DBL_MAX) ? level_for(bound2, lvl + 1) : lvl + 1;
    }
    if (s_level > top) {
        // the ball is not covered by any level's 3x3x3 block (query far outside the grid, or no bound at all):
        // descend from the <= 8 root cells that hold the whole target
        const double cell = gv.cell0 * (double)(1ll << (2 * top));
        const int b0 = (int)(PCR_COORD_BIAS >> (2 * top));
        bool valid = lane < 8;
        const unsigned int X = b0 + (lane & 1), Y = b0 + ((lane >> 1) & 1), Z = b0 + ((lane >> 2) & 1);
        unsigned int s = 0, e = 0;
        double bdist = 0.0;
        if (valid) {
            bdist = box_dist2(gv, top, cell, X, Y, Z, ax, ay, az);
            valid = bdist <= bound2 && lookup_cell(gv.table[top], gv.mask[top], X, Y, Z, &s, &e);
        }
        hard_disperse(gv, L, sp, lane, valid, s, e, X, Y, Z, top, bdist, ax, ay, az, bd2, bid, bpos, bound2, h_pts, bw);
        ++h_steps;
        while (sp > 0) {
            --sp;
            const hard_entry en = L->stack[sp];
            const double ecell = gv.cell0 * (double)(1ll << (2 * en.level));
            if (box_dist2(gv, en.level, ecell, en.x, en.y, en.z, ax, ay, az) > bound2) continue;
            const int cl = en.level - 1;
            const unsigned int CX = en.x * 4u + (lane & 3), CY = en.y * 4u + ((lane >> 2) & 3), CZ = en.z * 4u + (lane >> 4);
            const double cdist = box_dist2(gv, cl, ecell * 0.25, CX, CY, CZ, ax, ay, az);
            unsigned int cs = 0, ce = 0;
            const bool cvalid = cdist <= bound2 && lookup_cell(gv.table[cl], gv.mask[cl], CX, CY, CZ, &cs, &ce);
            hard_disperse(gv, L, sp, lane, cvalid, cs, ce, CX, CY, CZ, cl, cdist, ax, ay, az, bd2, bid, bpos, bound2, h_pts, bw);
            ++h_steps;
        }
    }
    // the lanes' bests meet: the smallest distance over the wave (DPP), then -- among the lanes that hold it: one, unless two points are
    // exactly equidistant -- the smallest id; the winner's lane hands everything out by v_readlane (no LDS, no butterfly)
    {
        const double m = wave_min(bd2);
        unsigned long long cm = __ballot(bpos != POS_NONE && bd2 == m);
        if (!cm) { bd2 = DBL_MAX; bid = ID_NONE; bpos = POS_NONE; if (win) { win[0] = win[1] = win[2] = 0.0; } return; }
        int wl = (int)__ffsll((long long)cm) - 1;
        cm &= cm - 1;
        if (cm) {   // (rare) ties in distance: lowest id
            long long best_id = ((long long)__builtin_amdgcn_readlane((int)(bid >> 32), wl) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)bid, wl);
            while (cm) {
                const int l = (int)__ffsll((long long)cm) - 1;
                cm &= cm - 1;
                const long long id_l = ((long long)__builtin_amdgcn_readlane((int)(bid >> 32), l) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)bid, l);
                if (id_l < best_id) { best_id = id_l; wl = l; }
            }
        }
        bd2 = m;
        bid = ((long long)__builtin_amdgcn_readlane((int)(bid >> 32), wl) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)bid, wl);
        bpos = (unsigned int)__builtin_amdgcn_readlane((int)bpos, wl);
        if (win) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                win[k] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bw[k]), wl), __builtin_amdgcn_readlane(__double2loint(bw[k]), wl));
        }
    }
}

// (115 VGPRs -> 4 waves per SIMD; forcing 5, 6 or 8 with amdgpu_waves_per_eu spills and measured 31, 39, 50 us against 29)
__global__ void __launch_bounds__(256)
grid_hard_kernel(pcr_grid_view gv, const work_item* __restrict__ list, const unsigned int* __restrict__ count_p, long long nq, double max_d2, int gated,
                 unsigned int* __restrict__ res_pos, double* __restrict__ res_d2, unsigned long long* __restrict__ dbg,
                 const pcr_icp_dev_state* __restrict__ st) {
    __shared__ hard_lds s_lds[4];
    if (st && st->stop) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    hard_lds* L = &s_lds[wave];
    // exclusive prefix of the sub-list lengths (lanes 0..H_NLIST-1 hold one list each)
    unsigned int l_cnt = lane < H_NLIST ? count_p[H_CSTRIDE * lane] : 0u;
    unsigned int l_inc = l_cnt;
#pragma unroll
    for (int off = 1; off < H_NLIST; off <<= 1) {
        const unsigned int o = __shfl_up(l_inc, off, 64);
        if (lane >= off) l_inc += o;
    }
    const unsigned int count = __shfl(l_inc, H_NLIST - 1, 64);
    const unsigned int l_exc = l_inc - l_cnt;
    const unsigned int l_cap = hard_list_cap(nq);
    for (unsigned int w = blockIdx.x * 4 + wave; w < count; w += gridDim.x * 4) {
        const int hl = (int)__ffsll((long long)__ballot(lane < H_NLIST && l_exc <= w && w < l_exc + l_cnt)) - 1;  // exactly one list holds item w
        const work_item it = list[(size_t)hl * l_cap + (w - __shfl(l_exc, hl, 64))];
        const unsigned long long h_t0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
        unsigned int h_steps = 0, h_pts = 0;
        int s_level0 = 0;
        double bd2;
        long long bid;
        unsigned int bpos;
        // it.best_d2 is an UPPER bound of the tile stage's candidate's squared distance: it bounds the search, and the candidate
        // itself is met again by the scan (it lies inside the bound) -- reading its record here would be one more dependent
        // memory round trip in a chain of four
        hard_search(gv, L, lane, it.ax, it.ay, it.az, gated ? fmin(it.best_d2, max_d2) : it.best_d2, it.best_pos != POS_NONE, bd2, bid, bpos, h_steps,
                    h_pts, s_level0);
        if (lane == 0) {
            res_pos[it.qi] = bpos;
            if (res_d2) res_d2[it.qi] = bd2;
            if (dbg && w < 60000) {
                dbg[(1 << 17) + w * 4 + 0] = __builtin_amdgcn_s_memtime() - h_t0;
                dbg[(1 << 17) + w * 4 + 1] = ((unsigned long long)h_steps << 32);
                dbg[(1 << 17) + w * 4 + 2] = h_pts;
                dbg[(1 << 17) + w * 4 + 3] = (unsigned long long)(it.best_pos != POS_NONE) | ((unsigned long long)(s_level0 + 1) << 8);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ one-kernel ICP pass
// One launch per ICP iteration of the device-resident loop: every wave runs its tile, adds the moments of the queries it
// proved, hands what it could not prove to a work queue, then serves that queue (one query per wave, hard_search) until
// its group of tiles has nothing left; the wave that leaves last solves the Procrustes step.  Against tile kernel ->
// hard kernel this removes a launch boundary and, more important, the hard items start while slow tiles are still
// running (the tile stage is one generation of waves: its tail left most of the chip idle).
//
// Moments meet in 64-bit FIXED-POINT accumulators (value * 2^F, F chosen on the host so that N * R^2 * 2^F < 2^61):
// integer additions commute, so waves may add in any order -- device-scope atomics, no slabs -- and the totals are still
// bitwise reproducible.  Every correspondence is rounded to 2^-F once, on its own (<= N * 2^-F / 2 on a sum bounded by N R^2:
// < 1e-12 relative at 120 000 points, three orders below the parity bar), so the totals are also independent of how the
// queries are grouped into tiles and queue items.  32 sets of 20 words
// (measured per pass: 8 sets 53.5 us -- the atomics queue at their lines --, 16: 41.9, 32: 41.0, 64: 42.1, 128: 42.5, 1024: +14 us
// for the finishing read).
//
// Everything waves tell each other INSIDE the launch goes through device-scope atomics (performed at the coherence
// point: the L2 of another XCD never holds a stale copy) and never needs a fence (an agent-scope release is an L2
// write-back on this chip: 2048 blocks doing one cost 40 us):
//   queue     32 groups (wave w belongs to group w % 32).  Per group ONE 64-bit word
//                 done << 44 | claimed << 22 | reserved
//             (tiles of the group through with their tile stage; slot indices handed out; slots reserved).  A tile leaves its
//             stage with ONE vector atomic: its k-th open query reserves a slot in group (g + k) % 32 (a tile in a sparse part of
//             the scan leaves up to 32 queries open: all in its own group measured 32..250 items per group), the lane of k = 0
//             also reports the tile done in its own group and claims the wave's first slot to serve.  Then it stores its items;
//             every 8-byte word of an item is self-validating (all-ones = not written yet; the consumer puts that back), so no
//             ordering between the words or against the counter is needed.
//   claims    a free wave takes the NEXT slot index of its group with one returning add -- whether or not an item is there yet --
//             and then polls ITS OWN slot: waiting waves sit on private addresses, an item lands in front of the wave that
//             serves it, and the group word only sees one add per reservation and per claim.  (Polling the group word and claiming
//             by compare-and-swap measured 430 us per pass with 117 waves per word, 88 us with 15: same-address accesses
//             serialise at the memory side at ~100 ns each, and few waves per group balance badly.)  When the last tile of the
//             last group has reported done, every reservation of the launch is in place: that wave reads the final slot counts R
//             and POISONS the slots [R, R + waves of the group) of every group; every wave still waiting (each holds exactly one
//             index >= R) reads that and leaves.
//   waiting   holding an index means waiting for it -- for an item from ANY tile, or for the poison that needs every tile -- so a
//             wave may only WAIT if every tile of the launch has started (they are resident and will finish): on a busy GPU
//             (several pairs in flight, clouds of several wave generations) waiting waves would keep the slots the missing tiles
//             need.  Until then a free wave only takes items that are already there (compare-and-swap while claimed < reserved)
//             and leaves otherwise.  Progress: the last generation of waves waits, and serves what the earlier ones left.
//   variants  inline_queue = 0: the launch only publishes and grid_drain_kernel (below) serves the queues -- the host's choice
//             while several ICP loops of the process are in flight; same results bit for bit.
//   finish    vmcnt(0) (own atomics acknowledged), then a two-level ticket (one word per group, then a root).
#ifndef PCR_ACC_SETS
#define PCR_ACC_SETS 32
#endif
constexpr int ACC_SETS = PCR_ACC_SETS;
#ifndef PCR_PASS_GROUPS
#define PCR_PASS_GROUPS 32
#endif
constexpr int PASS_GROUPS = PCR_PASS_GROUPS;
constexpr int PASS_SYNC_STRIDE = 32;                      // 64-bit words per group: queue word at 0, started at 16, ticket at 17 (other line)
constexpr int PASS_SYNC_WORDS = PASS_SYNC_STRIDE * (PASS_GROUPS + 1);   // + root ticket (0) / error word (16)
constexpr unsigned long long ITEM_NONE = ~0ull;           // "not written yet": never a valid word of an item
// Last word of an item = query index | binary32 bits of its candidate's bound << 32 (finite, or +inf).  A POISON word carries 0xfffffffe
// (no such bound) above the id of the pass that wrote it: poison of an earlier pass reads as "not written yet", so nobody ever has to
// clean it up (the whole queue is re-initialised per ICP call).
constexpr unsigned int POISON_HI = 0xfffffffeu;
__device__ static inline unsigned long long item_poison(unsigned int pass_id) { return ((unsigned long long)POISON_HI << 32) | pass_id; }
#ifndef PCR_PASS_SPIN_LIMIT
#define PCR_PASS_SPIN_LIMIT (1u << 21)
#endif
// polls (>= 1 us each) before a waiting wave gives up.  Giving up is SAFE: the wave says so with its ticket, and the launch's last
// wave then walks every reserved slot and serves what is still there (all tiles are done by then: every reserved item has been
// written) -- same integers into the accumulators, same result bit for bit.  Never seen with the default; a build with
// -DPCR_PASS_SPIN_LIMIT=2 makes almost every waiter give up (tests/test_gpu_queue_giveup.py).
constexpr unsigned int PASS_SPIN_LIMIT = PCR_PASS_SPIN_LIMIT;
constexpr int Q_BITS = 22;                                // reserved / claimed fields; done has the upper 20 bits
constexpr unsigned long long Q_MASK = (1ull << Q_BITS) - 1ull;
constexpr long long PASS_MAX_NQ = 1ll << 26;              // 32 groups x 2^22 slots, with room for the poison range
struct pass_args {
    unsigned long long* items;            // [PASS_GROUPS][cap][4], all-ones between launches
    unsigned long long* sync;             // [PASS_GROUPS + 1][PASS_SYNC_STRIDE], zero between launches
    unsigned int cap;
    unsigned long long* acc;              // [ACC_SETS][PCR_NMOM], zero between launches
    wt_xyz* prev_xyz;
    double scale, inv_scale;              // 2^F, 2^-F
    pcr_icp_dev_state* st;
    pcr_icp_loop_args la;
    unsigned int* tile_cost;              // [waves]: points every tile staged in the last pass (issue priority of this one)
    unsigned int pass_id;                 // passes enqueued so far in this ICP call
    unsigned long long* host_block;       // (or null) the context's pinned landing block as the device sees it: the wave that finishes the LAST pass
    unsigned int notify;                  // of a chunk (notify != 0), or the pass that stops the loop, writes state + log + flag there itself
    unsigned int* plog;                   // per-pass log of the call (or null): s_memrealtime stamps, 100 MHz, low 32 bits -- [pass] end of the pass,
                                          // [256 + pass] start of the drain launch (0: one-launch pass), [512 + pass] queue items, [768] start of the call
};
constexpr int PASS_LOG_WORDS = 3 * PCR_ICP_MAX_LOG + 8;   // 32-bit words
// the context's mapped landing block (8 KiB): loop state | pass log | ... | completion flag (last word)
constexpr size_t HOST_LOG_OFF = (sizeof(pcr_icp_dev_state) + 63) & ~(size_t)63;
constexpr size_t HOST_FLAG_OFF = 8192 - 8;
static_assert(HOST_LOG_OFF + 4 * PASS_LOG_WORDS <= HOST_FLAG_OFF, "state + pass log + flag fit the landing block");
__host__ __device__ static inline unsigned int pass_item_cap(long long nq) {   // room for every query of the group's tiles
    const long long tiles = (nq + WT_Q - 1) / WT_Q + 4;
    return (unsigned int)(((tiles + PASS_GROUPS - 1) / PASS_GROUPS) * (WT_Q + 1));   // + the poison range
}

// value * 2^F rounded to the nearest integer (ties to even) for |value * 2^F| <= 2^51 -- the host picks F so -- by the
// add-a-big-constant trick: one multiply-add and one 64-bit subtraction (__double2ll_rn is a ~12-instruction sequence, and
// every lane of a tile converts 19 moments)
__device__ static inline long long to_fixed(double v, double scale) {
    constexpr double BIG = 6755399441055744.0;   // 1.5 * 2^52
    return __double_as_longlong(__builtin_fma(v, scale, BIG)) - __double_as_longlong(BIG);
}

// lanes 0..18 of the calling wave add one moment each (one vector atomic instruction); mk = this lane's moment
__device__ static inline void acc_fixed_add(unsigned long long* __restrict__ acc, unsigned int set, int lane, double mk, double scale) {
    if (lane < PCR_NMOM - 1 && mk != 0.0)
        atomicAdd(acc + (size_t)(set % ACC_SETS) * PCR_NMOM + lane, (unsigned long long)to_fixed(mk, scale));   // two's complement
}

__device__ static inline void acc_fixed_add_ll(unsigned long long* __restrict__ acc, unsigned int set, int lane, long long v) {
    if (lane < PCR_NMOM - 1 && v != 0) atomicAdd(acc + (size_t)(set % ACC_SETS) * PCR_NMOM + lane, (unsigned long long)v);
}

__device__ static inline unsigned long long ld_dev(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ static inline void st_dev(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

union pass_lds {
    wtile_lds t;
    hard_lds h;
    double xch[WT_Q * (PCR_NMOM - 1)];
    long long xll[WT_Q * (PCR_NMOM - 1)];
};

// one queued query, served by the calling wave: exact search, then its moments (rounded to fixed point once) and its result
__device__ __forceinline__ static void pass_serve_item(const pcr_grid_view& gv, pass_lds* L, const int lane, const unsigned long long w, const double max_d2,
                                                       const pass_args& A, unsigned int* __restrict__ res_pos, unsigned long long* __restrict__ dbg,
                                                       const unsigned int dbg_slot, const unsigned long long t_start) {
    // lanes 0..3 hold the item's four words
    const unsigned int w_lo = (unsigned int)w, w_hi = (unsigned int)(w >> 32);
    const double ax = __hiloint2double(__builtin_amdgcn_readlane((int)w_hi, 0), __builtin_amdgcn_readlane((int)w_lo, 0));
    const double ay = __hiloint2double(__builtin_amdgcn_readlane((int)w_hi, 1), __builtin_amdgcn_readlane((int)w_lo, 1));
    const double az = __hiloint2double(__builtin_amdgcn_readlane((int)w_hi, 2), __builtin_amdgcn_readlane((int)w_lo, 2));
    const unsigned int qi = (unsigned int)__builtin_amdgcn_readlane((int)w_lo, 3);
    const float cand_b2 = __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)w_hi, 3));
    const unsigned long long h_t0 = dbg ? __builtin_amdgcn_s_memtime() : 0;
    unsigned int h_steps = 0, h_pts = 0;
    int s_level0 = 0;
    double bd2;
    long long bid;
    unsigned int bpos;
    const bool have_cand = cand_b2 < INFINITY;
    double win[3];
    hard_search(gv, &L->h, lane, ax, ay, az, have_cand ? fmin((double)cand_b2, max_d2) : max_d2, have_cand, bd2, bid, bpos, h_steps, h_pts, s_level0, win);
    wave_sync();
    if (bpos != POS_NONE) {   // wave-uniform: after the merge every lane holds the same result
        if (lane == 0) {
            pcr_pt b;   // (the descent had the winner's record in hand: no gather)
            b.x = win[0]; b.y = win[1]; b.z = win[2]; b.id = 0;
            A.prev_xyz[qi] = wt_xyz{b.x, b.y, b.z};
            double m[PCR_NMOM];
#pragma unroll
            for (int k = 0; k < PCR_NMOM; ++k) m[k] = 0.0;
            moments_add(m, gv.origin, ax, ay, az, b, max_d2, 1);
#pragma unroll
            for (int k = 0; k < PCR_NMOM - 1; ++k) L->xch[k] = m[k];
        }
        wave_sync();
        acc_fixed_add(A.acc, qi, lane, lane < PCR_NMOM - 1 ? L->xch[lane] : 0.0, A.scale);
        wave_sync();
    }
    if (lane == 0) {
        res_pos[qi] = bpos;
        if (dbg && dbg_slot < 60000) {
            dbg[(1 << 17) + dbg_slot * 4 + 0] = __builtin_amdgcn_s_memtime() - h_t0;
            dbg[(1 << 17) + dbg_slot * 4 + 1] = ((unsigned long long)h_steps << 32) | (unsigned int)((h_t0 - t_start) >> 4);
            dbg[(1 << 17) + dbg_slot * 4 + 2] = h_pts;
            dbg[(1 << 17) + dbg_slot * 4 + 3] = (unsigned long long)have_cand | ((unsigned long long)(s_level0 + 1) << 8);
        }
    }
}

// The wave that leaves a pass last: totals of the fixed-point accumulators -> Procrustes step -> convergence test -> loop state.
// lane l < 60 owns moment l % 20 of every third set: independent loads, integer sums (order irrelevant), words zeroed behind.
// The head of the loop state travels with them (one 8-byte word per lane) into LDS: the step reads and updates ~50 of its words.
__device__ __forceinline__ static void pass_finish(const pcr_grid_view& gv, pass_lds* L, const int lane, const pass_args& A, unsigned long long* root,
                                                   unsigned long long* __restrict__ dbg) {
    constexpr int HEAD_WORDS = (int)(pcr::ICP_STATE_HEAD_BYTES / 8);
    static_assert(pcr::ICP_STATE_HEAD_BYTES % 8 == 0 && HEAD_WORDS <= 64, "state head: one word per lane");
    double* const head = L->xch + 32;   // 16-byte aligned, behind the 20 moments
    unsigned long long hw = 0, err_w = 0;
    if (lane < HEAD_WORDS) hw = reinterpret_cast<const unsigned long long*>(A.st)[lane];
    if (lane == 63) err_w = ld_dev(root + 16);   // (with the other loads: one round trip)
    const int mom = lane % PCR_NMOM, part = lane / PCR_NMOM;
    long long sum = 0;
    if (part < 3) {
        constexpr int PER = (ACC_SETS + 2) / 3;
        unsigned long long v[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int set = part + 3 * j;
            v[j] = set < ACC_SETS ? ld_dev(A.acc + (size_t)set * PCR_NMOM + mom) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int set = part + 3 * j;
            if (v[j]) st_dev(A.acc + (size_t)set * PCR_NMOM + mom, 0ull);
            sum += (long long)v[j];
        }
    }
    if (dbg && lane == 0) dbg[(1 << 19) - 4] = __builtin_amdgcn_s_memrealtime();
    sum += __shfl(sum, lane + PCR_NMOM, 64) + __shfl(sum, lane + 2 * PCR_NMOM, 64);   // lanes 0..19: all three parts
    wave_sync();
    if (dbg && lane == 0) dbg[(1 << 19) - 3] = __builtin_amdgcn_s_memrealtime();
    if (lane < PCR_NMOM) L->xch[lane] = (double)sum * A.inv_scale;
    if (lane < HEAD_WORDS) reinterpret_cast<unsigned long long*>(head)[lane] = hw;
    const bool gave_up = __builtin_amdgcn_readlane((int)(unsigned int)err_w, 63) != 0;
    wave_sync();
    if (lane == 0) {
        pcr_icp_dev_state* hs = reinterpret_cast<pcr_icp_dev_state*>(head);
        if (gave_up) { hs->status = PCR_E_HIP; hs->stop = 1; st_dev(root + 16, 0ull); }
        else pcr::icp_step(hs, L->xch, gv.origin, A.la, A.st->r_diff, A.st->t_diff);
    }
    wave_sync();
    if (dbg && lane == 0) dbg[(1 << 19) - 2] = __builtin_amdgcn_s_memrealtime();
    if (A.plog && lane == 0 && A.pass_id < (unsigned int)PCR_ICP_MAX_LOG) A.plog[A.pass_id] = (unsigned int)__builtin_amdgcn_s_memrealtime();
    if (lane < HEAD_WORDS) reinterpret_cast<unsigned long long*>(A.st)[lane] = reinterpret_cast<const unsigned long long*>(head)[lane];
    if (dbg && lane == 0) dbg[(1 << 19) - 1] = __builtin_amdgcn_s_memrealtime();
    // The host's copy, written from HERE when the host waits for this pass (the last one of its chunk, or the one that stops the
    // loop): on this pool whatever small operation follows the last big kernel of a call -- a copy, a one-block kernel, an event --
    // starts 16-45 ms late every 10th-30th call, so the call's result must not depend on one.  State head
    // from LDS, the logs from memory (this pass's own entries were stored by lane 0 above: acknowledged first), the pass log, then
    // a system-scope release and the flag the host polls.
    if (A.host_block) {
        const int stop_now = reinterpret_cast<const pcr_icp_dev_state*>(head)->stop;   // (LDS: the same in every lane)
        if (A.notify || stop_now) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            wave_sync();
            constexpr int STATE_WORDS = (int)(sizeof(pcr_icp_dev_state) / 8);
            if (lane < HEAD_WORDS) A.host_block[lane] = reinterpret_cast<const unsigned long long*>(head)[lane];
            for (int w = HEAD_WORDS + lane; w < STATE_WORDS; w += 64) A.host_block[w] = ld_dev(reinterpret_cast<const unsigned long long*>(A.st) + w);
            if (A.plog)
                for (int w = lane; w < PASS_LOG_WORDS / 2; w += 64) A.host_block[HOST_LOG_OFF / 8 + w] = ld_dev(reinterpret_cast<const unsigned long long*>(A.plog) + w);
            __threadfence_system();
            wave_sync();
            if (lane == 0) __hip_atomic_store(A.host_block + HOST_FLAG_OFF / 8, (unsigned long long)A.pass_id + 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

#ifndef PCR_PASS_DIAG
#define PCR_PASS_DIAG 0
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCR_WT_WAVES, 8)))
grid_pass_kernel(const pcr_grid_view* __restrict__ gvp, pcr_grid_view gv, pcr_pt* __restrict__ q, long long nq, double max_d2, int xcd_remap,
                 unsigned int pcap_arg, unsigned int pcap_busy, unsigned int* __restrict__ res_pos, unsigned long long* __restrict__ dbg_arg, int use_prev, int inline_queue,
                 pass_args A) {
    // the kernel's own stamps and counters (scripts/pass_stamps.py, wt_stamps.py, hard_stamps.py) are compiled in by -DPCR_PASS_DIAG=1 only:
    // as a run-time switch they kept ~14 scalar registers alive through the whole kernel, which spills scalars into vector lanes as it is
    unsigned long long* const dbg = PCR_PASS_DIAG ? dbg_arg : nullptr;
    __shared__ pass_lds s_lds[4];
    // (the wave index is the same in every lane: through readfirstlane everything derived from it -- the tile, its queue group, half a
    // dozen pointers -- lives in scalar registers instead of occupying a vector register pair each)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned int tile = wtile_block(xcd_remap) * 4 + wave;
    // the tile's records and the loop state are requested together; a pass enqueued behind a stop leaves before any side effect
    wt_pre P;
    wtile_preload(tile, lane, q, nq, res_pos, use_prev ? A.prev_xyz : nullptr, P);
    const unsigned int last_cost = use_prev ? A.tile_cost[tile] : 0u;
    // Staging cap of this pass's tiles.  A launch of several wave generations does not end with its slowest tile but with the sum of
    // all of them, and a query a tile leaves open costs a whole wave of the drain launch 10-30 us: while many queries overflow their
    // tiles -- the unseeded first pass, and the passes of a source that still moves by decimetres per iteration in a cloud whose
    // neighbours are centimetres apart (1 M x 1 M, eight overlapping frames, 0.8 m off: 200 000 queue items per pass, every one "too
    // many points in the box") -- the tiles stage up to pcap_busy points (items 203 000 -> 76 000, drain 1.46 -> 0.73 ms for 0.3 ms more
    // in the tiles); once the last pass's queue was short (< 1/16 of the queries) the small cap is the faster one again.
    unsigned int pcap = pcap_arg;
    if (pcap_busy > pcap_arg && A.plog) {
        const unsigned int prev_items = A.pass_id > 0 && A.pass_id <= (unsigned int)PCR_ICP_MAX_LOG ? A.plog[2 * PCR_ICP_MAX_LOG + A.pass_id - 1] : 0xffffffffu;
        if ((unsigned long long)prev_items * 16ull > (unsigned long long)nq) pcap = pcap_busy;
        pcap = (unsigned int)__builtin_amdgcn_readfirstlane((int)pcap);
    }
    const pcr_xform x = A.st->x;
    if (A.st->stop) return;
    // The launch ends with its slowest tiles (several staging rounds in the densest part of the scan), and while all tiles run the
    // SIMDs are VALU-issue bound with four waves each: a tile that was heavy in the last pass gets issue priority over its
    // neighbours -- those have slack, they would only wait in the work queue.
    {
        const unsigned int lc = (unsigned int)__builtin_amdgcn_readfirstlane((int)last_cost);
        const unsigned int c = lc & 0xffffu, had_open = lc >> 16;   // points staged / queries left open in the last pass
        if (c > 3 * WT_PR) __builtin_amdgcn_s_setprio(3);
        else if (c > 2 * WT_PR) __builtin_amdgcn_s_setprio(2);
        else if (c > WT_PR || had_open) __builtin_amdgcn_s_setprio(1);   // (open queries: the sooner they are queued, the sooner they are served)
    }
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long rt_start = dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    unsigned long long rt_tile = 0, rt_acc = 0;
    unsigned int n_items = 0, n_polls = 0, n_casfail = 0;
    pass_lds* L = &s_lds[wave];
    const unsigned int n_waves = gridDim.x * 4;
    const unsigned int g = tile % PASS_GROUPS;
    const unsigned int g_tiles = n_waves / PASS_GROUPS + (g < n_waves % PASS_GROUPS ? 1u : 0u);
    unsigned long long* const g_q = A.sync + (size_t)PASS_SYNC_STRIDE * g;   // done << 44 | claimed << 22 | reserved
    unsigned long long* const g_started = g_q + 16;
    unsigned long long* const g_ticket = g_q + 17;
    unsigned long long* const items = A.items + (size_t)g * A.cap * 4;
    if (lane == 0) __hip_atomic_fetch_add(g_started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // nobody waits for the reply
    const unsigned int n_groups = n_waves < (unsigned int)PASS_GROUPS ? n_waves : (unsigned int)PASS_GROUPS;
    // ---- tile
    wt_state S;
    wtile_search(*gvp, &L->t, tile, lane, P, q, nq, x, 1, 1, max_d2, 1, pcap, res_pos, nullptr, dbg, use_prev ? A.prev_xyz : nullptr, S, inline_queue ? A.sync + 16 : nullptr,
                 (unsigned int)PASS_SYNC_STRIDE, n_groups, n_waves);
    wave_sync();
    __builtin_amdgcn_s_setprio(0);
    {
        const unsigned int n_open = (unsigned int)__popcll(__ballot(S.open && lane < WT_Q));   // (a ballot is taken by all lanes)
        if (lane == 0) A.tile_cost[tile] = (S.staged < 0xffffu ? S.staged : 0xffffu) | (n_open << 16);
    }
    if (dbg) rt_tile = __builtin_amdgcn_s_memrealtime();
    // ---- leave the tile stage.  First the target records of the proven queries are requested (for the moments; a neighbour that did
    // not change came with the seed), then the open queries reserve their queue slots, then the tile reports done -- and, when the
    // whole launch has started, so that waiting is allowed, claims this wave's first slot to serve with the same add.
    const bool proven = lane < WT_Q && S.won != POS_NONE;
    pcr_pt nb;
    nb.x = nb.y = nb.z = 0.0; nb.id = 0;
    const bool nb_known = proven && S.won == P.seed_pos;   // same neighbour as in the last pass: its coordinates came with the seed
    if (proven && !nb_known) nb = as_global(gv.pts)[S.won];
    // Every tile of the LAUNCH started?  (A wave that waits for work waits for items from any tile: all of them must be resident.)
    // Read in the middle of the tile -- a launch of one wave generation is complete within a microsecond, and the counts only
    // grow --; only if that read came too early is it repeated here.
    bool all_started = false;
    if (inline_queue) {
        all_started = S.probe != 0;
        if (!all_started) {
            unsigned long long sv = 0;
            if (lane < (int)n_groups) sv = ld_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane + 16);
            const bool ok = lane < (int)n_groups && (unsigned int)sv == n_waves / PASS_GROUPS + ((unsigned int)lane < n_waves % PASS_GROUPS ? 1u : 0u);
            all_started = __ballot(ok) == (n_groups >= 64 ? ~0ull : (1ull << n_groups) - 1ull);
        }
    }
    const bool unres = S.open && lane < WT_Q;
    const unsigned long long um = __ballot(unres);
    unsigned int mine = 0;     // slot index this wave owns (claimed below, or in the loop)
    bool have_claim = false;
    unsigned long long own = 0;   // reply of the add that reports the tile done (used after the moments: its round trip hides behind them)
    unsigned long long* const root = A.sync + (size_t)PASS_SYNC_STRIDE * PASS_GROUPS;
    {
        // The k-th open query of the tile reserves a slot in group (g + k) mod groups, all of them by ONE vector atomic -- a tile in a
        // sparse part of the scan leaves up to 32 queries open, and all of them in the tile's own group measured 32..250 items per
        // group (the fullest group finished 9 us after the emptiest).  Only when the replies are back -- the reservations are in
        // place -- does lane 0 report the tile done in its own group (and claim the wave's first slot to serve): "every tile done"
        // must imply "every reservation made", or the poison below could land in a slot that is reserved a moment later.
        const unsigned int rank = (unsigned int)__popcll(um & ((1ull << lane) - 1ull));
        const unsigned int tg = (g + rank) % n_groups;
        unsigned long long slot_w = 0;
        if (unres) slot_w = __hip_atomic_fetch_add(A.sync + (size_t)PASS_SYNC_STRIDE * tg, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" :: "v"(slot_w) : "memory");   // replies first
        if (lane == 0)
            own = __hip_atomic_fetch_add(g_q, (1ull << (2 * Q_BITS)) | (all_started ? 1ull << Q_BITS : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (unres) {
            unsigned long long* it = A.items + ((size_t)tg * A.cap + (size_t)(slot_w & Q_MASK)) * 4;
            const bool cand = !S.clamped && S.cand_pos != POS_NONE;
            // a NaN coordinate must not look like "not written yet"
            st_dev(it + 0, S.ax == S.ax ? (unsigned long long)__double_as_longlong(S.ax) : 0x7ff8000000000000ull);
            st_dev(it + 1, S.ay == S.ay ? (unsigned long long)__double_as_longlong(S.ay) : 0x7ff8000000000000ull);
            st_dev(it + 2, S.az == S.az ? (unsigned long long)__double_as_longlong(S.az) : 0x7ff8000000000000ull);
            st_dev(it + 3, (unsigned long long)(unsigned int)S.qi | ((unsigned long long)__float_as_uint(cand ? S.bound2 : INFINITY) << 32));
        }
    }
    // ---- moments of the proven queries.  Every correspondence is rounded to the fixed-point grid ONCE, by itself; from there on
    // only integers are added (here through the wave's LDS slice, then by the atomics): the totals do not depend on how queries
    // are grouped into tiles or queue items, or on any order.
    {
        double m[PCR_NMOM];
#pragma unroll
        for (int k = 0; k < PCR_NMOM; ++k) m[k] = 0.0;
        if (proven) {
            if (nb_known) { nb.x = P.seed_b.x; nb.y = P.seed_b.y; nb.z = P.seed_b.z; }
            else A.prev_xyz[S.qi] = wt_xyz{nb.x, nb.y, nb.z};   // seed of the next pass's tile
            moments_add(m, gv.origin, S.ax, S.ay, S.az, nb, max_d2, 1);
        }
        if (lane < WT_Q) {
#pragma unroll
            for (int k = 0; k < PCR_NMOM - 1; ++k) L->xll[lane * (PCR_NMOM - 1) + k] = to_fixed(m[k], A.scale);
        }
        wave_sync();
        long long tot = 0;
        if (lane < PCR_NMOM - 1) {
#pragma unroll 8
            for (int j = 0; j < WT_Q; ++j) tot += L->xll[j * (PCR_NMOM - 1) + lane];
        }
        acc_fixed_add_ll(A.acc, tile, lane, tot);
        wave_sync();
    }
    // ---- the reply of the done-add: this wave's first slot, and whether it was the last tile of its group
    {
        {
            const unsigned int o_lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)own), o_hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(own >> 32));
            own = ((unsigned long long)o_hi << 32) | o_lo;
        }
        if (all_started) {
            have_claim = true;
            mine = (unsigned int)((own >> Q_BITS) & Q_MASK);
        }
        if ((unsigned int)(own >> (2 * Q_BITS)) == g_tiles - 1u && inline_queue) {
            // last tile of its group.  The last GROUP to get there knows that every reservation of the launch is in place: it
            // poisons the slots [R, R + waves of the group) of every group -- whoever waits there holds an index that will never
            // be filled (each wave exactly one), reads that and leaves.
            int fin = 0;
            if (lane == 0) fin = __hip_atomic_fetch_add(root + 8, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(n_groups - 1u) ? 1 : 0;
            if (__builtin_amdgcn_readfirstlane(fin)) {
                if (lane == 0) st_dev(root + 8, 0ull);
                unsigned long long qw = 0;
                if (lane < (int)n_groups) qw = ld_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane);
                const unsigned int R_l = (unsigned int)(qw & Q_MASK);
                for (unsigned int gg = 0; gg < n_groups; ++gg) {
                    const unsigned int R = (unsigned int)__builtin_amdgcn_readlane((int)R_l, (int)gg);
                    const unsigned int gt = n_waves / PASS_GROUPS + (gg < n_waves % PASS_GROUPS ? 1u : 0u);
                    unsigned long long* const base_g = A.items + (size_t)gg * A.cap * 4;
                    for (unsigned int k = lane; k < gt; k += 64) st_dev(base_g + (size_t)(R + k) * 4 + 3, item_poison(A.pass_id));
                }
            }
        }
    }
    if (dbg) rt_acc = __builtin_amdgcn_s_memrealtime();
    if (dbg && lane == 0) {   // per-block counters of scripts/wt_stamps.py and bench.py (filter pairs, passes, open queries, cycles)
        atomicAdd(&dbg[blockIdx.x * 4 + 1], S.dbg_pairs);
        atomicAdd(&dbg[blockIdx.x * 4 + 2], (unsigned long long)S.dbg_passes);
        atomicAdd(&dbg[blockIdx.x * 4 + 3], (unsigned long long)__popcll(um));
        atomicMax(&dbg[blockIdx.x * 4 + 0], __builtin_amdgcn_s_memtime() - t_start);
    }
    if (!inline_queue) return;   // throughput variant: grid_drain_kernel serves the queues and finishes
    // ---- serve the group's queue
    bool failed = false, left_by_poison = false;
    for (;;) {
        if (!have_claim) {
            if (!all_started) {
                unsigned long long sv = 0;
                if (lane < (int)n_groups) sv = ld_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane + 16);
                const bool ok = lane < (int)n_groups && (unsigned int)sv == n_waves / PASS_GROUPS + ((unsigned int)lane < n_waves % PASS_GROUPS ? 1u : 0u);
                all_started = __ballot(ok) == (n_groups >= 64 ? ~0ull : (1ull << n_groups) - 1ull);
            }
            if (all_started) {
                unsigned long long old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(g_q, 1ull << Q_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned int o_lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)old), o_hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(old >> 32));
                old = ((unsigned long long)o_hi << 32) | o_lo;
                mine = (unsigned int)((old >> Q_BITS) & Q_MASK);   // an item, or the poison, will turn up there
            } else {
                // tiles of the group are not resident yet: take what is there, never hold a slot waiting for them
                unsigned long long qw = 0;
                if (lane == 0) qw = ld_dev(g_q);
                const unsigned int q_lo = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)qw), q_hi = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(qw >> 32));
                qw = ((unsigned long long)q_hi << 32) | q_lo;
                mine = (unsigned int)((qw >> Q_BITS) & Q_MASK);
                if (mine >= (unsigned int)(qw & Q_MASK)) break;
                int got = 0;
                if (lane == 0) {
                    unsigned long long expect = qw;
                    got = __hip_atomic_compare_exchange_strong(g_q, &expect, qw + (1ull << Q_BITS), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                }
                if (!__builtin_amdgcn_readfirstlane(got)) { ++n_casfail; continue; }   // somebody else moved the word: look again
            }
        }
        have_claim = false;
        // wait for the item (or the poison) in slot `mine`
        unsigned long long* it = items + (size_t)mine * 4;
        unsigned long long w = ITEM_NONE;
        bool poisoned = false;
        for (unsigned int spins = 0;; ++spins) {
            if (lane < 4) w = ld_dev(it + lane);
            // word 3: this pass's poison -> leave; an earlier pass's -> as good as not written
            const unsigned int w3_lo = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w, 3), w3_hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(w >> 32), 3);
            poisoned = w3_hi == POISON_HI && w3_lo == A.pass_id;
            const unsigned long long mk = __ballot(lane < 4 && w != ITEM_NONE && !(lane == 3 && w3_hi == POISON_HI));
            if (poisoned || mk == 0xfull) break;
            if (spins >= PASS_SPIN_LIMIT) { failed = true; break; }
            ++n_polls;
#ifndef PCR_SLOT_SLEEP
#define PCR_SLOT_SLEEP 16   // (1 / 4 / 16 / 32 / 64 measured within noise of each other; fewer polls = less traffic)
#endif
            __builtin_amdgcn_s_sleep(PCR_SLOT_SLEEP);
        }
        if (failed) break;
        if (poisoned) { left_by_poison = true; break; }
        if (lane < 4) st_dev(it + lane, ITEM_NONE);   // the slot is clean for the next launch
        ++n_items;
        pass_serve_item(gv, L, lane, w, max_d2, A, res_pos, dbg, g + PASS_GROUPS * mine, t_start);
    }
    const unsigned long long rt_loop = dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    // ---- the wave that leaves last converts the totals, solves the Procrustes step and tests convergence
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    {
        // Tickets carry a second count in their upper half: waves that left the queue by reading the poison.  Such a wave held an
        // index >= the final slot count of its group, so every slot below it had been claimed: a group with one of them has nothing
        // unclaimed, and when every group has one the last wave need not look at the queue words at all.
        // (third field, bits 48+: waves that gave up waiting -- each holds a slot index whose item came, or will have come, after it left)
        if (lane == 0) {
            const unsigned long long tg_old = __hip_atomic_fetch_add(g_ticket, 1ull | (left_by_poison ? 1ull << 32 : 0ull) | (failed ? 1ull << 48 : 0ull), __ATOMIC_RELAXED,
                                                                     __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned int)tg_old == g_tiles - 1u) {
                const bool g_clean = ((tg_old >> 32) & 0xffffull) != 0 || left_by_poison;
                const bool g_failed = (tg_old >> 48) != 0 || failed;
                const unsigned long long tr_old = __hip_atomic_fetch_add(root, 1ull | (g_clean ? 1ull << 32 : 0ull) | (g_failed ? 1ull << 48 : 0ull), __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned int)tr_old == n_groups - 1u) {
                    last = ((unsigned int)((tr_old >> 32) & 0xffffull) + (g_clean ? 1u : 0u) == n_groups) ? 1 : 2;   // 2: look for unclaimed items
                    if ((tr_old >> 48) != 0 || g_failed) last = 3;                                                    // 3: somebody gave up: look at every reserved slot
                }
            }
        }
    }
    if (dbg && lane == 0) {
        unsigned long long* d = dbg + (1 << 19) + (size_t)(blockIdx.x * 4 + wave) * 8;
        d[0] = rt_start; d[1] = rt_tile; d[2] = rt_acc; d[3] = rt_loop; d[4] = n_items; d[5] = n_polls | ((unsigned long long)n_casfail << 32); d[6] = S.dbg_pairs | ((unsigned long long)S.dbg_passes << 40) | ((unsigned long long)__popcll(um) << 48); d[7] = __builtin_amdgcn_s_memrealtime();
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    // ---- last wave of the launch.  Nobody is OBLIGED to wait for work (a wave that cannot see every tile of the launch started
    // leaves when it finds nothing), so items reserved after the last wave of their group left may still sit in the queue: unless
    // every group reported a wave that left by the poison, look for them and serve them here.  Then clean the queue words.
    if (last == 3 && dbg_arg && lane == 0) atomicAdd(&dbg_arg[(1 << 19) - 16], 1ull);   // (PCR_DEBUG_STAMPS: passes in which a waiter gave up)
    if (last >= 2) {
        unsigned long long qw = 0;
        if (lane < (int)n_groups) qw = ld_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane);
        const unsigned int r_l = (unsigned int)(qw & Q_MASK);
        // (a wave gave up: its slot lies BELOW the claimed count -- every reserved slot is looked at; served ones read all-ones)
        const unsigned int c_l = last == 3 ? 0u : (unsigned int)((qw >> Q_BITS) & Q_MASK);
        unsigned long long left = __ballot(lane < (int)n_groups && c_l < r_l);
        while (left) {
            const int gg = (int)__ffsll((long long)left) - 1;
            left &= left - 1;
            const unsigned int c0 = (unsigned int)__builtin_amdgcn_readlane((int)c_l, gg), r0 = (unsigned int)__builtin_amdgcn_readlane((int)r_l, gg);
            for (unsigned int i = c0; i < r0; ++i) {
                unsigned long long* it = A.items + ((size_t)gg * A.cap + i) * 4;
                unsigned long long w = ITEM_NONE;
                bool arrived = false;
                // (every tile of the launch is done -- its wave took its ticket behind its stores --, so an item that is not there by now
                // was served: one look suffices when sweeping after a give-up; the unclaimed range keeps its patience)
                const unsigned int patience = last == 3 ? 64u : (PASS_SPIN_LIMIT < (1u << 21) ? (1u << 21) : PASS_SPIN_LIMIT);
                bool served_before = false;
                for (unsigned int spins = 0; spins < patience; ++spins) {
                    if (lane < 4) w = ld_dev(it + lane);
                    const unsigned int w3_hi = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)(w >> 32), 3);
                    if (__ballot(lane < 4 && w != ITEM_NONE && !(lane == 3 && w3_hi == POISON_HI)) == 0xfull) { arrived = true; break; }
                    if (last == 3 && __ballot(lane < 4 && w == ITEM_NONE) == 0xfull) { served_before = true; break; }   // taken and cleaned by the wave that served it
                    __builtin_amdgcn_s_sleep(2);
                }
                if (served_before) continue;
                if (lane < 4) st_dev(it + lane, ITEM_NONE);
                // an item that never arrived (its words would decode to query 0xffffffff) is never served: the pass is flagged
                // failed instead, pass_finish reports PCR_E_HIP for the call
                if (!arrived) { if (lane == 0) st_dev(root + 16, 1ull); continue; }
                pass_serve_item(gv, L, lane, w, max_d2, A, res_pos, dbg, 59999u, t_start);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the leftovers' atomics
    }
    if (lane < (int)n_groups) {   // (stores: nobody waits for them)
        st_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane, 0ull);
        st_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane + 16, 0ull);
        st_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane + 17, 0ull);
    }
    if (lane == 0) st_dev(root, 0ull);
    pass_finish(gv, L, lane, A, root, dbg);
}

// Throughput variant of the pass, used while several ICP loops of the process are in flight: waves that wait for work buy
// latency for one pair with slots the other pairs' kernels could use (4 pairs in flight: 1.5e9 correspondences/s against
// 3.1e9), and without waiting the last tile of a group would serve what is left of its queue alone (430 us).  So the tile
// kernel only publishes (publish_only), and this second launch serves the queues with a static partition -- all counts are
// final at the kernel boundary -- and finishes.  What a wave adds to the accumulators (one rounding per tile, one per item)
// never depends on who serves an item or when: both variants give the same result bit for bit.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCR_WT_WAVES, 8)))
grid_drain_kernel(pcr_grid_view gv, double max_d2, unsigned int* __restrict__ res_pos, unsigned long long* __restrict__ dbg, pass_args A) {
    __shared__ pass_lds s_lds[4];
    if (A.st->stop) return;
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memtime() : 0;
    const unsigned long long t_rt_start = __builtin_amdgcn_s_memrealtime();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pass_lds* L = &s_lds[wave];
    // exclusive prefix of the groups' item counts (lanes 0..PASS_GROUPS-1 hold one group each; written by the previous launch)
    static_assert(PASS_GROUPS <= 64, "one group per lane");
    const unsigned int l_cnt = lane < PASS_GROUPS ? (unsigned int)(A.sync[(size_t)PASS_SYNC_STRIDE * lane] & Q_MASK) : 0u;
    unsigned int total = 0;
    const unsigned int l_exc = wave_excl_scan_u32(l_cnt, lane, &total);
    const unsigned int n_waves = gridDim.x * 4;
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0) dbg[(1 << 19) - 8] = total;   // diagnostics (PCR_DEBUG_STAMPS): items of this pass
    if (A.plog && blockIdx.x == 0 && threadIdx.x == 0 && A.pass_id < (unsigned int)PCR_ICP_MAX_LOG) {
        // (launched behind the tile kernel on the same stream: the first wave of this launch starts when that one has ended)
        A.plog[PCR_ICP_MAX_LOG + A.pass_id] = (unsigned int)t_rt_start | 1u;
        A.plog[2 * PCR_ICP_MAX_LOG + A.pass_id] = total;
    }
    for (unsigned int w_i = blockIdx.x * 4 + wave; w_i < total; w_i += n_waves) {
        const int grp = (int)__ffsll((long long)__ballot(lane < PASS_GROUPS && l_exc <= w_i && w_i < l_exc + l_cnt)) - 1;   // exactly one group holds item w_i
        const unsigned int idx = w_i - (unsigned int)__builtin_amdgcn_readlane((int)l_exc, grp);
        unsigned long long* it = A.items + ((size_t)grp * A.cap + idx) * 4;
        unsigned long long w = ITEM_NONE;
        if (lane < 4) {
            w = it[lane];
            it[lane] = ITEM_NONE;   // the slot is clean for the next pass
        }
        pass_serve_item(gv, L, lane, w, max_d2, A, res_pos, dbg, w_i, t_start);
    }
    // ---- two-level ticket over the blocks' waves (group = wave index % PASS_GROUPS), then the finish
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long* const root = A.sync + (size_t)PASS_SYNC_STRIDE * PASS_GROUPS;
    const unsigned int wid = blockIdx.x * 4 + wave;
    const unsigned int n_groups = n_waves < (unsigned int)PASS_GROUPS ? n_waves : (unsigned int)PASS_GROUPS;
    const unsigned int g = wid % n_groups;
    const unsigned int g_size = n_waves / n_groups + (g < n_waves % n_groups ? 1u : 0u);
    unsigned long long* const g_ticket = A.sync + (size_t)PASS_SYNC_STRIDE * g + 17;
    int last = 0;
    if (lane == 0) {
        if (__hip_atomic_fetch_add(g_ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(g_size - 1u)) {
            st_dev(g_ticket, 0ull);
            if (__hip_atomic_fetch_add(root, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)(n_groups - 1u)) {
                st_dev(root, 0ull);
                last = 1;
            }
        }
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    // queue words of the tile kernel: clean for the next pass (every wave of this launch has read them: it took its ticket after)
    if (lane < PASS_GROUPS) {
        st_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane, 0ull);
        st_dev(A.sync + (size_t)PASS_SYNC_STRIDE * lane + 16, 0ull);
    }
    pass_finish(gv, L, lane, A, root, dbg);
}

// --------------------------------------------------------------- epilogues
// nn1: sorted position -> original target index, gate, scatter to the query's original slot
__global__ void grid_finalize_nn1_kernel(pcr_grid_view gv, const pcr_pt* __restrict__ q, long long nq, const unsigned int* __restrict__ res_pos,
                                         const double* __restrict__ res_d2, double max_d2, int gated, int* __restrict__ idx_out,
                                         double* __restrict__ d2_out, unsigned int* __restrict__ hard_count) {
    const long long qi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi < H_NLIST) hard_count[H_CSTRIDE * qi] = 0;  // next search starts with empty lists (stream-ordered)
    if (qi >= nq) return;
    const unsigned int pos = res_pos[qi];
    const double d2 = res_d2[qi];
    const long long qid = q[qi].id;
    int id = -1;
    if (pos != POS_NONE && (!gated || d2 < max_d2)) id = (int)gv.pts[pos].id;
    idx_out[qid] = id;
    d2_out[qid] = (pos == POS_NONE) ? INFINITY : d2;
}

// ICP: gate + Procrustes moments about gv.origin.  One partial slab of PCR_NMOM doubles per
// block, summed in fixed order by reduce_partials_kernel (bitwise reproducible run to run).
__global__ void __launch_bounds__(256)
grid_accumulate_kernel(pcr_grid_view gv, const pcr_pt* __restrict__ q, long long nq, pcr_xform x, int apply_x,
                       const unsigned int* __restrict__ res_pos, double max_d2, int gated, double* __restrict__ partials,
                       unsigned int* __restrict__ ticket, double* __restrict__ out, unsigned int* __restrict__ hard_count,
                       pcr_icp_dev_state* __restrict__ st, pcr_icp_loop_args la, wt_xyz* __restrict__ prev_xyz) {
    __shared__ double s_part[4][PCR_NMOM];
    if (st && st->stop) return;
    double m[PCR_NMOM];
#pragma unroll
    for (int k = 0; k < PCR_NMOM; ++k) m[k] = 0.0;
    // batches of 4 queries per thread: the 4 result slots, the 4 query records and then the 4 matched
    // target records are each loaded together (three memory round trips per batch instead of twelve)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long q0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; q0 < nq; q0 += 4 * stride) {
        unsigned int pos[4];
        pcr_pt p[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long qi = q0 + u * stride;
            pos[u] = qi < nq ? res_pos[qi] : POS_NONE;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long qi = q0 + u * stride;
            if (qi < nq) p[u] = q[qi];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pos[u] != POS_NONE) b[u] = gv.pts[pos[u]];
        if (prev_xyz) {  // seed of the next pass's wave tiles (coalesced 24-byte stores, off every critical path)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (pos[u] != POS_NONE) prev_xyz[q0 + u * stride] = wt_xyz{b[u].x, b[u].y, b[u].z};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (pos[u] == POS_NONE) continue;
            double ax = p[u].x, ay = p[u].y, az = p[u].z;
            if (apply_x) xform_apply(x, p[u], &ax, &ay, &az);
            const double d2 = dist2(ax, ay, az, b[u]);
            if (gated && !(d2 < max_d2)) continue;
            const double a0 = ax - gv.origin[0], a1 = ay - gv.origin[1], a2 = az - gv.origin[2];
            const double b0 = b[u].x - gv.origin[0], b1 = b[u].y - gv.origin[1], b2 = b[u].z - gv.origin[2];
            m[0] += 1.0;
            m[1] += a0; m[2] += a1; m[3] += a2;
            m[4] += b0; m[5] += b1; m[6] += b2;
            m[7] += b0 * a0; m[8] += b0 * a1; m[9] += b0 * a2;
            m[10] += b1 * a0; m[11] += b1 * a1; m[12] += b1 * a2;
            m[13] += b2 * a0; m[14] += b2 * a1; m[15] += b2 * a2;
            m[16] += (a0 * a0 + a1 * a1) + a2 * a2;
            m[17] += (b0 * b0 + b1 * b1) + b2 * b2;
            m[18] += d2;
        }
    }
    // wave totals on the DPP network (fixed order, total in lane 63): a __shfl_xor butterfly is 6 dependent ds_bpermute round
    // trips per moment -- 114 of them were most of this kernel's wave lifetime
#pragma unroll
    for (int k = 0; k < PCR_NMOM - 1; ++k) m[k] = wave_total_f64(m[k]);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < PCR_NMOM; ++k) s_part[wave][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < PCR_NMOM) {
        const double v = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
        partials[(long long)blockIdx.x * PCR_NMOM + threadIdx.x] = v;
    }
    if (ticket == nullptr) {
        // host-sum mode: `partials` is pinned host memory; the host adds the slabs after the stream sync (no ticket,
        // no last-block pass: two dependent round trips fewer on the critical path of an ICP iteration)
        if (blockIdx.x == 0 && threadIdx.x < H_NLIST) hard_count[H_CSTRIDE * threadIdx.x] = 0;
        return;
    }
    // ---- the block that arrives last sums the slabs in fixed order (saves a launch boundary).
    // Hand-off per the CDNA4 recipe: drained stores -> barrier -> agent-scope release -> ticket;
    // last arriver: agent-scope acquire -> barrier -> plain loads.
    __shared__ unsigned int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == gridDim.x - 1) ? 1u : 0u;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *ticket = 0;  // ready for the next launch (stream-ordered)
        }
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < H_NLIST) hard_count[H_CSTRIDE * threadIdx.x] = 0;
    __shared__ double s_red[8][32];
    const int k = threadIdx.x & 31, slice = threadIdx.x >> 5;  // 8 strided slices, then a fixed tree
    double v = 0.0;
    if (k < PCR_NMOM) {
        // <= 256 slabs: up to 32 per slice, loaded 8 at a time (independent loads), added in slab order
        for (int b0 = slice; b0 < (int)gridDim.x; b0 += 64) {
            double t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 8 * u;
                t[u] = b < (int)gridDim.x ? partials[(long long)b * PCR_NMOM + k] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
    }
    s_red[slice][k] = v;
    __syncthreads();
    __shared__ double s_m[PCR_NMOM];
    if (slice == 0 && k < PCR_NMOM) {
        const double tot = ((s_red[0][k] + s_red[1][k]) + (s_red[2][k] + s_red[3][k])) + ((s_red[4][k] + s_red[5][k]) + (s_red[6][k] + s_red[7][k]));
        if (out) out[k] = tot;
        s_m[k] = tot;
    }
    if (st) {
        // device-resident loop: Procrustes + convergence test here, the next pass reads st->x (kernel boundary orders it)
        __syncthreads();
        if (threadIdx.x == 0) pcr::icp_step(st, s_m, gv.origin, la, st->r_diff, st->t_diff);
    }
}

// ------------------------------------------------------------------- host
struct grid_scratch {
    unsigned int* res_pos = nullptr;
    double* res_d2 = nullptr;
    unsigned long long* acc = nullptr;  // [ACC_SETS][PCR_NMOM] fixed-point moment accumulators + [PASS_SYNC_WORDS] queue words (one-kernel pass only)
    unsigned long long* items = nullptr;   // [PASS_GROUPS][pass_item_cap][4] work queue of the one-kernel pass
    unsigned int* tile_cost = nullptr;     // [waves of the pass]
    void* prev_xyz = nullptr;         // [nq] x 24 B: coordinates of every query's neighbour of the last ICP pass (device loop only)
    work_item* hard_list = nullptr;   // [nq] worst case
    unsigned int* hard_count = nullptr;
    int64_t nq = 0;
};

static int grid_scratch_alloc(pcr_ctx* ctx, int64_t nq, bool want_d2, unsigned int* hard_count, grid_scratch* sc) {
    int rc;
    sc->nq = nq;
    sc->hard_count = hard_count;  // zero at context creation, reset by the epilogue kernels
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * nq, (void**)&sc->res_pos))) return rc;
    if (want_d2 && (rc = pcr_dev_alloc(ctx, sizeof(double) * nq, (void**)&sc->res_d2))) {
        pcr_dev_free(ctx, sc->res_pos, sizeof(unsigned int) * nq);
        sc->res_pos = nullptr;
        return rc;
    }
    if ((rc = pcr_dev_alloc(ctx, sizeof(work_item) * (size_t)H_NLIST * hard_list_cap(nq), (void**)&sc->hard_list))) {
        pcr_dev_free(ctx, sc->res_pos, sizeof(unsigned int) * nq);
        if (sc->res_d2) pcr_dev_free(ctx, sc->res_d2, sizeof(double) * nq);
        sc->res_pos = nullptr; sc->res_d2 = nullptr;
        return rc;
    }
    return PCR_OK;
}

static void grid_scratch_free(pcr_ctx* ctx, grid_scratch* sc) {
    if (sc->res_pos) pcr_dev_free(ctx, sc->res_pos, sizeof(unsigned int) * sc->nq);
    if (sc->res_d2) pcr_dev_free(ctx, sc->res_d2, sizeof(double) * sc->nq);
    if (sc->hard_list) pcr_dev_free(ctx, sc->hard_list, sizeof(work_item) * (size_t)H_NLIST * hard_list_cap(sc->nq));
    if (sc->prev_xyz) pcr_dev_free(ctx, sc->prev_xyz, 24 * (size_t)sc->nq);
    if (sc->items) pcr_dev_free(ctx, sc->items, 32 * (size_t)PASS_GROUPS * pass_item_cap(sc->nq));
    if (sc->tile_cost) pcr_dev_free(ctx, sc->tile_cost, sizeof(unsigned int) * (size_t)((sc->nq + 4 * WT_Q - 1) / (4 * WT_Q)) * 4);
    sc->tile_cost = nullptr;
    if (sc->acc) pcr_dev_free(ctx, sc->acc, sizeof(unsigned long long) * (ACC_SETS * PCR_NMOM + PASS_SYNC_WORDS + PASS_LOG_WORDS / 2));
    sc->res_pos = nullptr; sc->res_d2 = nullptr; sc->hard_list = nullptr; sc->prev_xyz = nullptr; sc->items = nullptr; sc->acc = nullptr;
}

// Enqueues the search stages on `stream` over the `nq` records at `q` (a whole Morton-sorted cloud or a run of it);
// leaves res_pos (and res_d2 when the scratch has it) on the device.  `st` != null: device-resident ICP loop.
static unsigned int wtile_point_cap(const pcr_ctx* ctx, int64_t nq) {
    static const int wt_rounds_env = getenv("PCR_WT_ROUNDS") ? atoi(getenv("PCR_WT_ROUNDS")) : 0;
    const int nblocks = (int)((nq + 63) / 64);
    return (unsigned int)WT_PR * (wt_rounds_env > 0 ? wt_rounds_env : (nblocks > 8 * ctx->cu_count ? WT_ROUNDS_LARGE : WT_ROUNDS_SMALL));
}
static unsigned int wtile_busy_cap() {   // PCR_WT_ROUNDS_BUSY: staging rounds of a tile while the queue of the last pass was long (see grid_pass_kernel)
    static const int env = getenv("PCR_WT_ROUNDS_BUSY") ? atoi(getenv("PCR_WT_ROUNDS_BUSY")) : 0;
    return (unsigned int)WT_PR * (unsigned int)(env > 0 ? env : 48);
}
static int wtile_xcd_remap() {
    static const int xcd_remap = getenv("PCR_TILE_XCD") ? atoi(getenv("PCR_TILE_XCD")) : 1;
    return xcd_remap;
}

static int grid_search_enqueue(pcr_ctx* ctx, const pcr_index* idx, pcr_pt* q, int64_t nq, hipStream_t stream, const pcr_xform* x, int write_back,
                               double max_d2, bool gated, bool mark, grid_scratch* sc, const pcr_icp_dev_state* st, bool use_prev = false) {
    const int nblocks = (int)((nq + 63) / 64);
    pcr_xform xi;
    pcr_xform_from_T(nullptr, &xi);
    if (mark) pcr_prof_mark(ctx, 0);
    {
        if (ctx->d_debug) hipMemsetAsync(ctx->d_debug, 0, sizeof(unsigned long long) * ((1 << 16) + 8 * (size_t)nblocks), stream);
        const int wblocks = (int)((nq + 4 * WT_Q - 1) / (4 * WT_Q));
        hipLaunchKernelGGL(grid_wtile_kernel, dim3(wblocks), dim3(256), 0, stream, (const pcr_grid_view*)idx->d_view, q, (long long)nq, x ? *x : xi, (x || st) ? 1 : 0,
                           write_back, max_d2, gated ? 1 : 0, wtile_xcd_remap(), wtile_point_cap(ctx, nq), sc->res_pos, sc->res_d2, sc->hard_list, sc->hard_count,
                           ctx->d_debug, st, (const wt_xyz*)(use_prev ? sc->prev_xyz : nullptr));
    }
    if (mark) pcr_prof_mark(ctx, 1);
    // a fixed grid of waves walks the hard list (its length is only known on the device)
    const long long want = (nq + 3) / 4;
    // 8 blocks of 4 waves per CU: twice what is resident at 4 waves per SIMD (4 x CU measured the same, 2 x CU 30 % slower)
    const int g3 = (int)(want < 8ll * ctx->cu_count ? (want < 1 ? 1 : want) : 8ll * ctx->cu_count);
    hipLaunchKernelGGL(grid_hard_kernel, dim3(g3), dim3(256), 0, stream, idx->view, (const work_item*)sc->hard_list,
                       (const unsigned int*)sc->hard_count, (long long)nq, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2, ctx->d_debug, st);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

static int grid_search_launch(pcr_ctx* ctx, const pcr_index* idx, pcr_pt* q, int64_t nq, hipStream_t stream, unsigned int* hard_count,
                              const pcr_xform* x, int write_back, double max_d2, bool gated, bool want_d2, bool mark, grid_scratch* sc) {
    int rc = grid_scratch_alloc(ctx, nq, want_d2, hard_count, sc);
    if (rc) return rc;
    rc = grid_search_enqueue(ctx, idx, q, nq, stream, x, write_back, max_d2, gated, mark, sc, nullptr);
    if (rc) grid_scratch_free(ctx, sc);
    return rc;
}

int pcr_grid_nn1(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_xform* x, double max_d2, int32_t* d_idx, double* d_d2) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    grid_scratch sc;
    // tiles are runs of the Morton-sorted query cloud (sorted once; rigid motion keeps them compact)
    int rc = pcr_cloud_morton_sort(ctx, qc, idx->cell);
    if (rc) return rc;
    const int64_t nq = qc->n;
    rc = grid_search_launch(ctx, idx, qc->d, nq, ctx->stream, ctx->d_counters + PCR_HARD_COUNTERS, x, 0, max_d2, gated, true, true, &sc);
    if (rc) return rc;
    const int grid = (int)((nq + 255) / 256);
    hipLaunchKernelGGL(grid_finalize_nn1_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (const pcr_pt*)qc->d, (long long)nq,
                       (const unsigned int*)sc.res_pos, (const double*)sc.res_d2, max_d2, gated ? 1 : 0, d_idx, d_d2, sc.hard_count);
    PCR_HIP(ctx, hipGetLastError());
    grid_scratch_free(ctx, &sc);
    return PCR_OK;
}

int pcr_grid_icp_pass(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_xform* x, double max_d2, int write_back,
                      double* d_moments) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    const bool was_sorted = qc->morton_sorted;
    int rc = pcr_cloud_morton_sort(ctx, qc, idx->cell);
    if (rc) return rc;
    const int64_t nq = qc->n;
    // Lanes: the pass over one pair is a chain of three dependent launches and is latency-bound; runs of the sorted
    // source are independent, so they go down separate streams and overlap each other's stalls.  Every lane reduces its
    // own moments (fixed order); the host adds the lanes in lane order.  One lane while profiling (per-kernel events).
    // Host-sum mode (the ICP loop's zero-copy read-back, not while profiling): every block's partial slab goes straight
    // to pinned host memory and the host adds them after the sync.
    const bool host_sum = (d_moments == ctx->h_pinned) && !ctx->profile && ctx->h_slabs != nullptr;
    int lanes = host_sum ? ctx->icp_lanes : 1;
    if (lanes > PCR_MAX_LANES) lanes = PCR_MAX_LANES;
    while (lanes > 1 && nq < (int64_t)lanes * 8192) --lanes;
    if (!host_sum) {
        grid_scratch sc;
        rc = grid_search_launch(ctx, idx, qc->d, nq, ctx->stream, ctx->d_counters + PCR_HARD_COUNTERS, x, write_back, max_d2, gated, false, true, &sc);
        if (rc) return rc;
        int grid = (int)((nq + 1023) / 1024);  // four queries per thread
        if (grid > ctx->cu_count) grid = ctx->cu_count;
        if ((rc = pcr_ensure_scratch(ctx, sizeof(double) * PCR_NMOM * (size_t)grid))) {
            grid_scratch_free(ctx, &sc);
            return rc;
        }
        // after a write-back pass the cloud already holds the transformed points
        pcr_prof_mark(ctx, 2);
        hipLaunchKernelGGL(grid_accumulate_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (const pcr_pt*)qc->d, (long long)nq, *x,
                           write_back ? 0 : 1, (const unsigned int*)sc.res_pos, max_d2, gated ? 1 : 0, ctx->d_partials, ctx->d_counters + 64,
                           d_moments, sc.hard_count, (pcr_icp_dev_state*)nullptr, pcr_icp_loop_args{}, (wt_xyz*)nullptr);
        pcr_prof_mark(ctx, 3);
        pcr_prof_mark(ctx, 4);
        PCR_HIP(ctx, hipGetLastError());
        pcr_prof_finish(ctx);
        grid_scratch_free(ctx, &sc);
        return PCR_OK;
    }
    if ((rc = pcr_ctx_lanes(ctx, lanes))) return rc;
    if (!was_sorted) PCR_HIP(ctx, pcr_sync(ctx->stream));  // the sort ran on the main stream
    grid_scratch sc[PCR_MAX_LANES];
    int grids[PCR_MAX_LANES];
    const int64_t per = ((nq + lanes - 1) / lanes + 1023) / 1024 * 1024;
    int used = 0;
    for (int l = 0; l < lanes && rc == PCR_OK; ++l) {
        const int64_t q0 = per * l, q1 = q0 + per < nq ? q0 + per : nq;
        if (q0 >= q1) break;
        ++used;
        hipStream_t st = lanes == 1 ? ctx->stream : ctx->lane_stream[l];
        rc = grid_search_launch(ctx, idx, qc->d + q0, q1 - q0, st, ctx->d_counters + PCR_HARD_COUNTERS + 1024 * l, x, write_back, max_d2, gated, false, false, &sc[l]);
        if (rc) break;
        int grid = (int)((q1 - q0 + 1023) / 1024);
        if (grid > PCR_SLABS_PER_LANE) grid = PCR_SLABS_PER_LANE;
        grids[l] = grid;
        // slabs go straight to pinned host memory; no ticket
        hipLaunchKernelGGL(grid_accumulate_kernel, dim3(grid), dim3(256), 0, st, idx->view, (const pcr_pt*)(qc->d + q0), (long long)(q1 - q0), *x,
                           write_back ? 0 : 1, (const unsigned int*)sc[l].res_pos, max_d2, gated ? 1 : 0,
                           ctx->h_slabs + (size_t)PCR_NMOM * PCR_SLABS_PER_LANE * l, (unsigned int*)nullptr, (double*)nullptr, sc[l].hard_count,
                           (pcr_icp_dev_state*)nullptr, pcr_icp_loop_args{}, (wt_xyz*)nullptr);
    }
    hipError_t e = hipGetLastError();
    for (int l = 0; l < used; ++l) {
        const hipError_t es = pcr_sync(lanes == 1 ? ctx->stream : ctx->lane_stream[l]);
        if (e == hipSuccess) e = es;
        grid_scratch_free(ctx, &sc[l]);
    }
    if (rc) return rc;
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PCR_E_HIP; }
    // fixed-order sum on the host: lanes in order, slabs in order, 8 interleaved accumulators then a fixed tree
    double m[PCR_NMOM];
    for (int k = 0; k < PCR_NMOM; ++k) m[k] = 0.0;
    for (int l = 0; l < used; ++l) {
        const double* slab = ctx->h_slabs + (size_t)PCR_NMOM * PCR_SLABS_PER_LANE * l;
        double acc[8][PCR_NMOM];
        for (int j = 0; j < 8; ++j)
            for (int k = 0; k < PCR_NMOM; ++k) acc[j][k] = 0.0;
        for (int b2 = 0; b2 < grids[l]; ++b2)
            for (int k = 0; k < PCR_NMOM; ++k) acc[b2 & 7][k] += slab[(size_t)b2 * PCR_NMOM + k];
        for (int k = 0; k < PCR_NMOM; ++k)
            m[k] += ((acc[0][k] + acc[1][k]) + (acc[2][k] + acc[3][k])) + ((acc[4][k] + acc[5][k]) + (acc[6][k] + acc[7][k]));
    }
    // hand the sum over the way the single-lane pass does (the caller reads h_pinned, or copies d_moments)
    if (d_moments == ctx->h_pinned) memcpy(ctx->h_pinned, m, sizeof(m));
    else PCR_HIP(ctx, hipMemcpyAsync(d_moments, m, sizeof(m), hipMemcpyHostToDevice, ctx->stream));
    if (d_moments != ctx->h_pinned) PCR_HIP(ctx, pcr_sync(ctx->stream));
    return PCR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Whole ICP loop with the host out of the iteration (Registration/main.py:107-154 / icp_template.py:160-198).
// Per pass three launches: tile -> hard -> accumulate; the last block of the accumulate kernel sums the slabs in fixed
// order, solves the 3x3 Procrustes step, tests convergence and writes the next increment into the device state, which
// the next tile kernel reads.  The host enqueues a CHUNK of passes, then copies the 4.5-KB state back once; passes
// enqueued behind a stop return at their first instruction.
// Everything the one-launch pass wants initialised at the start of an ICP call, in ONE launch (three API calls -- state upload, two
// memsets -- cost a registration of one iteration, as in the batch, 10-15 us): accumulators and queue words zero, item slots all-ones,
// the loop state built from T0 (what pcr_grid_icp_loop writes into the host copy).
struct pass_T0 { double v[16]; };
__global__ void __launch_bounds__(256)
pass_init_kernel(unsigned long long* __restrict__ zero_p, unsigned int zero_n, unsigned long long* __restrict__ ones_p, unsigned long long ones_n,
                 pcr_icp_dev_state* __restrict__ st, pass_T0 T0, unsigned int* __restrict__ plog, unsigned long long* __restrict__ host_started) {
    if (host_started && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(host_started, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // "the call's first kernel runs"
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < ones_n; i += stride) ones_p[i] = ~0ull;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < zero_n; i += stride) zero_p[i] = 0ull;
    if (blockIdx.x != 0) return;
    static_assert(sizeof(pcr_icp_dev_state) % 8 == 0, "state in 8-byte words");
    unsigned long long* w = reinterpret_cast<unsigned long long*>(st);
    for (unsigned int i = threadIdx.x; i < sizeof(pcr_icp_dev_state) / 8; i += blockDim.x) w[i] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                st->x.r[3 * i + j] = T0.v[4 * i + j];
                st->R_last[3 * i + j] = T0.v[4 * i + j];
            }
            st->x.t[i] = T0.v[4 * i + 3];
            st->t_last[i] = T0.v[4 * i + 3];
        }
        for (int i = 0; i < 16; ++i) st->T_total[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 9; ++i) st->V[i] = (i % 4 == 0) ? 1.0 : 0.0;
        st->first = 1;
        if (plog) plog[3 * PCR_ICP_MAX_LOG] = (unsigned int)__builtin_amdgcn_s_memrealtime();
    }
}

// Fraction bits of the fixed-point moment accumulators.  Every moment is bounded by M = N * max(R^2, gate, 1), R = half diagonal
// of the target box + gate radius: totals stay below 2^61, one correspondence's moments below 2^51 (to_fixed).
bool pcr_pass_fixed_scale(const double lo[3], const double hi[3], long long nq, double max_d2, double* scale, double* inv_scale) {
    double r2 = 0.0;
    for (int k = 0; k < 3; ++k) r2 += 0.25 * (hi[k] - lo[k]) * (hi[k] - lo[k]);
    const double R = sqrt(r2) + sqrt(max_d2);
    const double M = (double)nq * fmax(fmax(R * R, max_d2), 1.0);
    int ex = 0;
    frexp(M, &ex);              // M < 2^ex
    int exq = 0;
    frexp(M / (double)nq, &exq);   // one correspondence's moments < 2^exq
    const int F = (61 - ex < 51 - exq) ? 61 - ex : 51 - exq;
    if (!(M > 0) || !std::isfinite(M) || F < 20) return false;
    *scale = ldexp(1.0, F);
    *inv_scale = ldexp(1.0, -F);
    return true;
}

// ICP loops of this process in flight per device (any context)
static std::atomic<int> g_loops_in_flight[64];   // per device ordinal (zero-initialised)
struct loop_guard {
    std::atomic<int>& n;
    explicit loop_guard(int device) : n(g_loops_in_flight[(unsigned int)device % 64u]) { n.fetch_add(1); }
    ~loop_guard() { n.fetch_sub(1); }
};

int pcr_grid_icp_loop(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_icp_params* params, const double T0[16],
                      pcr_icp_result* res) {
    const loop_guard in_flight(ctx->device);
    const auto h_t0 = std::chrono::steady_clock::now();   // host-side phases of the call (pcr_icp_pass_log): where a slow call spent its time
    auto h_us = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - a).count() / 1e3; };
    for (double& v : ctx->pass_host_us) v = 0.0;
    const char* const wait_s = getenv("PCR_PASS_INLINE");   // read per call: the tests switch it
    const int wait_env = wait_s ? atoi(wait_s) : -1;   // 0 / 1 force a variant; default: inline when alone on the device
    const bool gated = (params->max_d2 > 0) && std::isfinite(params->max_d2);
    int rc = pcr_cloud_morton_sort(ctx, qc, idx->cell);
    if (rc) return rc;
    const int64_t nq = qc->n;
    grid_scratch sc;
    if ((rc = grid_scratch_alloc(ctx, nq, false, ctx->d_counters + PCR_HARD_COUNTERS, &sc))) return rc;
    if ((rc = pcr_dev_alloc(ctx, 24 * (size_t)nq, &sc.prev_xyz))) {
        grid_scratch_free(ctx, &sc);
        return rc;
    }
    int grid = (int)((nq + 1023) / 1024);  // four queries per thread
    if (grid > ctx->cu_count) grid = ctx->cu_count;
    // One-kernel pass (fixed-point accumulators): needs a gate (it bounds |a'| by the target's extent) and enough fraction
    // bits.  Every moment is bounded by M = N * max(R^2, gate, 1), R = half diagonal of the target box + gate radius.
    pass_args pa{};
    bool fused = gated && nq <= PASS_MAX_NQ && getenv("PCR_ICP_NO_FUSED") == nullptr;
    if (fused) {
        double sc_f = 0, sc_i = 0;
        if (!pcr_pass_fixed_scale(idx->lo, idx->hi, nq, params->max_d2, &sc_f, &sc_i)) fused = false;   // absurd extents: keep the binary64 slabs
        else {
            pa.cap = pass_item_cap(nq);
            if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned long long) * (ACC_SETS * PCR_NMOM + PASS_SYNC_WORDS + PASS_LOG_WORDS / 2), (void**)&sc.acc)) ||
                (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * (size_t)((nq + 4 * WT_Q - 1) / (4 * WT_Q)) * 4, (void**)&sc.tile_cost)) ||
                (rc = pcr_dev_alloc(ctx, 32 * (size_t)PASS_GROUPS * pa.cap, (void**)&sc.items))) {
                grid_scratch_free(ctx, &sc);
                return rc;
            }
            pa.items = sc.items;
            pa.tile_cost = sc.tile_cost;
            pa.acc = sc.acc;
            pa.sync = sc.acc + ACC_SETS * PCR_NMOM;
            pa.plog = (unsigned int*)(sc.acc + ACC_SETS * PCR_NMOM + PASS_SYNC_WORDS);   // (zeroed with the accumulators by pass_init_kernel)
            pa.prev_xyz = (wt_xyz*)sc.prev_xyz;
            pa.scale = sc_f;
            pa.inv_scale = sc_i;
        }
    }
    pcr_icp_dev_state* d_st = nullptr;
    if ((rc = pcr_ensure_scratch(ctx, sizeof(double) * PCR_NMOM * (size_t)grid)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(pcr_icp_dev_state), (void**)&d_st))) {
        grid_scratch_free(ctx, &sc);
        return rc;
    }
    static_assert(sizeof(pcr_icp_dev_state) <= 8192, "state must fit the pinned staging buffer");
    pcr_icp_dev_state* h_st = (pcr_icp_dev_state*)ctx->h_state;
    memset(h_st, 0, sizeof(*h_st));
    pcr_xform_from_T(T0, &h_st->x);
    for (int i = 0; i < 16; ++i) h_st->T_total[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) h_st->R_last[3 * i + j] = T0[4 * i + j];
        h_st->t_last[i] = T0[4 * i + 3];
    }
    h_st->first = 1;
    for (int i = 0; i < 9; ++i) h_st->V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    pcr_icp_loop_args la;
    la.max_iter = params->max_iter; la.min_iter = params->min_iter;
    la.compat = params->mode == PCR_ICP_COMPAT_MAIN; la.r_metric = params->r_metric;
    la.r_thres = params->r_thres; la.t_thres = params->t_thres;
    volatile unsigned long long* const h_flag = (volatile unsigned long long*)((char*)ctx->h_state + HOST_FLAG_OFF);
    static const bool no_notify = getenv("PCR_ICP_NO_NOTIFY") != nullptr;   // A/B: read the state back by kernels behind the chunk instead
    if (fused && !no_notify) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, ctx->h_state, 0) == hipSuccess) pa.host_block = (unsigned long long*)dp;
        *h_flag = 0ull;   // (the stream is idle: the previous call waited for its own flag)
        h_flag[-1] = 0ull;   // "the call's first kernel runs"
    }
    hipError_t e = hipSuccess;
    ctx->pass_host_us[0] = h_us(h_t0);   // set-up: lay-out check, scratch from the arena
    if (fused) {
        pass_T0 t0;
        for (int i = 0; i < 16; ++i) t0.v[i] = T0[i];
        const unsigned long long ones_n = 4ull * PASS_GROUPS * pa.cap;
        const int ib = (int)((ones_n + 256 * 8 - 1) / (256 * 8));
        hipLaunchKernelGGL(pass_init_kernel, dim3(ib < 1 ? 1 : (ib > 1024 ? 1024 : ib)), dim3(256), 0, ctx->stream, sc.acc,
                           (unsigned int)(ACC_SETS * PCR_NMOM + PASS_SYNC_WORDS + PASS_LOG_WORDS / 2), sc.items, ones_n, d_st, t0, pa.plog,
                           pa.host_block ? pa.host_block + HOST_FLAG_OFF / 8 - 1 : (unsigned long long*)nullptr);
        e = hipGetLastError();
    } else e = hipMemcpyAsync(d_st, h_st, sizeof(*h_st), hipMemcpyHostToDevice, ctx->stream);
    pa.st = d_st;
    pa.la = la;
    pcr_xform xi;
    pcr_xform_from_T(nullptr, &xi);
    int enq = 0;
    const double prof_before = ctx->prof_ms[0] + ctx->prof_ms[1] + ctx->prof_ms[2] + ctx->prof_ms[3];
    // chunk schedule: what is known to run (min_iter) in one go, otherwise 2, 4, 8, ... (the reference's thresholds
    // usually stop after 1-3 iterations; a no-op pass costs three empty launches)
    int chunk = params->min_iter > 2 ? params->min_iter : 2;
    if (ctx->profile) chunk = 1;   // the state must be read after every pass: a pass behind a stop would log empty kernels
    while (e == hipSuccess && rc == PCR_OK && enq < params->max_iter) {
        if (chunk > params->max_iter - enq) chunk = params->max_iter - enq;
        if (chunk > 64) chunk = 64;
        for (int c = 0; c < chunk && rc == PCR_OK; ++c) {
            // from the second pass on, res_pos holds the previous pass's neighbours (same query order, same target)
            static const bool no_prev = getenv("PCR_NO_PREV") != nullptr;
            const bool use_prev = enq + c > 0 && !no_prev;
            // one launch only when it is a single generation of waves (16 per CU at <= 128 VGPRs) with the device to itself: in a
            // launch of several generations most waves may not wait for work (the later tiles need their slots), and the queue
            // would be served by the last generation alone (1 M points: 2.4 ms per pass against 0.32 ms for two launches)
            const bool one_generation = (nq + WT_Q - 1) / WT_Q <= 16ll * ctx->cu_count;
            const int inline_queue = wait_env >= 0 ? wait_env : ((one_generation && !ctx->shared_device && in_flight.n.load(std::memory_order_relaxed) == 1) ? 1 : 0);
            if (fused) {
                if (ctx->profile) pcr_prof_mark(ctx, 0);
                if (ctx->d_debug) hipMemsetAsync(ctx->d_debug, 0, sizeof(unsigned long long) * (inline_queue ? ((1 << 16) + 8 * (size_t)((nq + 63) / 64)) : ((size_t)1 << 19)), ctx->stream);
                const int wblocks = (int)((nq + 4 * WT_Q - 1) / (4 * WT_Q));
                pa.pass_id = (unsigned int)(enq + c);
                pa.notify = c == chunk - 1 ? 1u : 0u;
                // (the busy cap only where the drain launch exists and logs its item counts: two-launch passes)
                const unsigned int cap_n = wtile_point_cap(ctx, nq), cap_busy = (!inline_queue && cap_n > (unsigned int)(WT_PR * WT_ROUNDS_SMALL)) ? wtile_busy_cap() : cap_n;
                hipLaunchKernelGGL(grid_pass_kernel, dim3(wblocks), dim3(256), 0, ctx->stream, (const pcr_grid_view*)idx->d_view, idx->view, qc->d, (long long)nq,
                                   params->max_d2, wtile_xcd_remap(), cap_n, cap_busy, sc.res_pos, ctx->d_debug, use_prev ? 1 : 0, inline_queue, pa);
                if (ctx->profile) pcr_prof_mark(ctx, 1);
                if (!inline_queue) {
                    // 8 blocks of 4 waves per CU, like the stand-alone hard stage
                    const long long want = (nq + 3) / 4;
                    const int g3 = (int)(want < 8ll * ctx->cu_count ? (want < 1 ? 1 : want) : 8ll * ctx->cu_count);
                    hipLaunchKernelGGL(grid_drain_kernel, dim3(g3), dim3(256), 0, ctx->stream, idx->view, params->max_d2, sc.res_pos, ctx->d_debug, pa);
                }
                if (hipGetLastError() != hipSuccess) { rc = PCR_E_HIP; break; }
                pcr_prof_mark(ctx, 2);
            } else {
                // (ungated or absurdly large clouds) tile -> hard -> binary64 slabs: after the write-back of the tile kernel the
                // cloud already holds the transformed points
                rc = grid_search_enqueue(ctx, idx, qc->d, nq, ctx->stream, nullptr, 1, params->max_d2, gated, ctx->profile, &sc, d_st, use_prev);
                if (rc) break;
                pcr_prof_mark(ctx, 2);
                hipLaunchKernelGGL(grid_accumulate_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (const pcr_pt*)qc->d, (long long)nq, xi,
                                   0, (const unsigned int*)sc.res_pos, params->max_d2, gated ? 1 : 0, ctx->d_partials, ctx->d_counters + 64,
                                   (double*)nullptr, sc.hard_count, d_st, la, (wt_xyz*)sc.prev_xyz);
            }
            pcr_prof_mark(ctx, 3);
            pcr_prof_mark(ctx, 4);
            pcr_prof_finish(ctx);   // per-kernel HIP events (profiling only: one event sync per pass)
        }
        enq += chunk;
        if (rc) break;
        if (ctx->pass_host_us[1] == 0.0) ctx->pass_host_us[1] = h_us(h_t0) - ctx->pass_host_us[0];   // enqueueing the first chunk of passes
        const auto h_tw = std::chrono::steady_clock::now();
        if (pa.host_block) {
            // the finishing wave of the chunk's last pass (or of the pass that stopped the loop) has written state, log and flag
            bool seen = false;
            for (unsigned int spins = 0;; ++spins) {
                if (ctx->pass_host_us[4] == 0.0 && h_flag[-1] != 0ull) ctx->pass_host_us[4] = h_us(h_t0);   // the call's first kernel has started
                const unsigned long long f = __atomic_load_n((const unsigned long long*)h_flag, __ATOMIC_ACQUIRE);
                if (f >= (unsigned long long)enq || (f > 0 && h_st->stop)) { seen = true; break; }
                if ((spins & 255u) == 255u && h_us(h_tw) > 2e6) break;
                __builtin_ia32_pause();
            }
            ctx->pass_host_us[5] = h_us(h_tw);
            if (!seen) {   // never seen within two seconds: ask the runtime, read the state back the slow way
                e = pcr_sync(ctx->stream);
                if (e == hipSuccess && pcr_d2h_small(ctx, h_st, d_st, sizeof(*h_st)) != PCR_OK) e = hipErrorUnknown;
            }
        } else {
            hipEventRecord(ctx->ev3, ctx->stream);   // behind the chunk's last kernel, in front of the read-back (pass_host_us[4])
            if (pcr_d2h_small_enqueue(ctx, h_st, d_st, sizeof(*h_st)) != PCR_OK) e = hipErrorUnknown;
            if (e == hipSuccess && fused && pcr_d2h_small_enqueue(ctx, (char*)ctx->h_state + HOST_LOG_OFF, pa.plog, 4 * PASS_LOG_WORDS) != PCR_OK) e = hipErrorUnknown;
            double flag_us = 0.0;
            if (e == hipSuccess && pcr_wait_flag(ctx, &flag_us) != PCR_OK) e = hipErrorUnknown;   // (by reading memory: not through the runtime)
            ctx->pass_host_us[5] = flag_us;
            { float ms = 0; if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev3) == hipSuccess) ctx->pass_host_us[4] = 1e3 * ms; }   // call start .. last kernel of this chunk, by HIP events
        }
        ctx->pass_host_us[2] += h_us(h_tw);   // waiting for the device
        if (e != hipSuccess || h_st->stop) break;
        if (!ctx->profile) chunk *= 2;
    }
    if (e == hipSuccess) e = hipGetLastError();
    ctx->pass_log_n = 0;
    ctx->loop_dev_ms = 0.0;
    if (e == hipSuccess && rc == PCR_OK && fused) {
        const unsigned int* const lg = (const unsigned int*)((const char*)ctx->h_state + HOST_LOG_OFF);
        const int np = h_st->passes < PCR_ICP_MAX_LOG ? h_st->passes : PCR_ICP_MAX_LOG;
        unsigned int prev = lg[3 * PCR_ICP_MAX_LOG];
        for (int k = 0; k < np; ++k) {
            const unsigned int end = lg[k], ds = lg[PCR_ICP_MAX_LOG + k];
            ctx->pass_log[0][k] = 0.01 * (double)(unsigned int)((ds ? ds : end) - prev);     // tile launch (one-launch pass: the whole pass), us
            ctx->pass_log[1][k] = ds ? 0.01 * (double)(unsigned int)(end - ds) : 0.0;       // drain launch, us
            ctx->pass_log[2][k] = (double)lg[2 * PCR_ICP_MAX_LOG + k];                      // queue items (two-launch passes)
            prev = end;
        }
        ctx->pass_log_n = np;
        // the loop's duration on the device, from the kernels' own clock: first kernel of the call .. end of its last pass
        ctx->loop_dev_ms = np > 0 ? 1e-5 * (double)(unsigned int)(lg[np - 1] - lg[3 * PCR_ICP_MAX_LOG]) : 0.0;
    }
    grid_scratch_free(ctx, &sc);
    pcr_dev_free(ctx, d_st, sizeof(pcr_icp_dev_state));
    ctx->pass_host_us[3] = h_us(h_t0);   // the whole loop on the host
    if (rc) return rc;
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PCR_E_HIP; }
    if (h_st->status == PCR_E_HIP) { ctx->last_error = "ICP pass: a wave gave up waiting on the work queue"; return PCR_E_HIP; }
    double T_cur[16];
    pcr::T_from_xform(h_st->x, T_cur);
    if (!la.compat && h_st->status == PCR_OK && !h_st->converged && h_st->it == params->max_iter && params->max_iter > 0) {
        // icp_template.py:195-198: a non-converged last iteration still updates src_points and homo_mat_total
        if ((rc = pcr_cloud_transform(ctx, qc, T_cur))) return rc;
        pcr::T_mul4(T_cur, h_st->T_total, h_st->T_total);
    }
    res->iters = h_st->it;
    res->status = h_st->status;
    res->n_assoc = h_st->n_assoc;
    res->cost = h_st->cost;
    res->mean_d2 = h_st->mean_d2;
    for (int i = 0; i < h_st->it && i < PCR_ICP_MAX_LOG; ++i) { res->r_diff[i] = h_st->r_diff[i]; res->t_diff[i] = h_st->t_diff[i]; }
    res->nn_launches = h_st->passes;
    // per-pass kernel time (HIP events around the pass's launches): only measured while pcr_profile_enable is on
    res->nn_kernel_ms = ctx->profile ? (ctx->prof_ms[0] + ctx->prof_ms[1] + ctx->prof_ms[2] + ctx->prof_ms[3]) - prof_before : 0.0;
    memcpy(res->T_total, h_st->T_total, sizeof(double) * 16);
    if (la.compat) memcpy(res->T, T_cur, sizeof(T_cur));
    else memcpy(res->T, h_st->T_total, sizeof(double) * 16);
    return PCR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Fused batch of registrations (Registration/main.py:190-216: the loop over scan pairs): the same tile / queue / finish code as
// above, one launch per STAGE for every pair of the batch.  A wave finds its pair through tile_pair[], reads the pair's grid
// view, source slot and loop state through the (wave-uniform) descriptor pointer, and adds into the pair's own accumulator
// sets; queue items carry the GLOBAL source slot, from which the serving wave finds the pair again.  Always the two-launch
// ("throughput") variant of the pass -- many pairs share the device by construction -- followed by one wave per pair for the
// Procrustes step: what a wave adds to the accumulators never depends on who serves what, so every pair's result is
// bit-identical to pcr_icp on that pair alone.
__device__ static inline const pcr_batch_pair* batch_pair_ptr(const pcr_batch_pass_args& B, unsigned int p) {
    return as_global(B.pairs) + p;
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCR_WT_WAVES, 8)))
batch_pass_kernel(pcr_batch_pass_args B, int xcd_remap, unsigned int pcap, unsigned int pass_id, int use_prev) {
    __shared__ pass_lds s_lds[4];
    // (the wave index is the same in every lane: through readfirstlane everything derived from it -- the tile, its queue group, half a
    // dozen pointers -- lives in scalar registers instead of occupying a vector register pair each)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned int tile_g = (unsigned int)__builtin_amdgcn_readfirstlane((int)(wtile_block(xcd_remap) * 4 + wave));
    const unsigned int p = (unsigned int)__builtin_amdgcn_readfirstlane((int)as_global(B.tile_pair)[tile_g]);
    if (p >= (unsigned int)B.n_pairs) return;   // (slot of a pair that was rejected on the host)
    const pcr_batch_pair* const bp = batch_pair_ptr(B, p);
    const unsigned long long q_off = bp->q_off;
    const long long nq = bp->nq;
    const unsigned int tile = tile_g - (unsigned int)(q_off / WT_Q);
    pcr_pt* const q = B.q + q_off;
    unsigned int* const res_pos = B.res_pos + q_off;
    wt_xyz* const prev_xyz = (wt_xyz*)B.prev_xyz + q_off;
    const pcr_icp_dev_state* const st = as_global(B.st) + p;
    wt_pre P;
    wtile_preload(tile, lane, q, nq, res_pos, use_prev ? prev_xyz : nullptr, P);
    const unsigned int last_cost = use_prev ? B.tile_cost[tile_g] : 0u;
    const pcr_xform x = st->x;
    if (st->stop) return;
    {
        const unsigned int lc = (unsigned int)__builtin_amdgcn_readfirstlane((int)last_cost);
        const unsigned int c = lc & 0xffffu, had_open = lc >> 16;
        if (c > 3 * WT_PR) __builtin_amdgcn_s_setprio(3);
        else if (c > 2 * WT_PR) __builtin_amdgcn_s_setprio(2);
        else if (c > WT_PR || had_open) __builtin_amdgcn_s_setprio(1);
    }
    pass_lds* L = &s_lds[wave];
    const unsigned int n_waves = gridDim.x * 4;
    const unsigned int g = tile_g % PASS_GROUPS;
    const unsigned int n_groups = n_waves < (unsigned int)PASS_GROUPS ? n_waves : (unsigned int)PASS_GROUPS;
    const double max_d2 = B.max_d2;
    wt_state S;
    wtile_search(bp->gv, &L->t, tile, lane, P, q, nq, x, 1, 1, max_d2, 1, pcap, res_pos, nullptr, nullptr, use_prev ? prev_xyz : nullptr, S);
    wave_sync();
    __builtin_amdgcn_s_setprio(0);
    {
        const unsigned int n_open = (unsigned int)__popcll(__ballot(S.open && lane < WT_Q));
        if (lane == 0) B.tile_cost[tile_g] = (S.staged < 0xffffu ? S.staged : 0xffffu) | (n_open << 16);
    }
    const bool proven = lane < WT_Q && S.won != POS_NONE;
    pcr_pt nb;
    nb.x = nb.y = nb.z = 0.0; nb.id = 0;
    const bool nb_known = proven && S.won == P.seed_pos;
    if (proven && !nb_known) nb = as_global(bp->gv.pts)[S.won];
    // ---- open queries: the k-th one reserves a slot in group (g + k) mod groups (one vector atomic), then stores its item
    const bool unres = S.open && lane < WT_Q;
    const unsigned long long um = __ballot(unres);
    if (um) {
        const unsigned int rank = (unsigned int)__popcll(um & ((1ull << lane) - 1ull));
        const unsigned int tg = (g + rank) % n_groups;
        unsigned long long slot_w = 0;
        if (unres) {
            slot_w = __hip_atomic_fetch_add(B.sync + (size_t)PASS_SYNC_STRIDE * tg, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long* it = B.items + ((size_t)tg * B.cap + (size_t)(slot_w & Q_MASK)) * 4;
            const bool cand = !S.clamped && S.cand_pos != POS_NONE;
            it[0] = S.ax == S.ax ? (unsigned long long)__double_as_longlong(S.ax) : 0x7ff8000000000000ull;
            it[1] = S.ay == S.ay ? (unsigned long long)__double_as_longlong(S.ay) : 0x7ff8000000000000ull;
            it[2] = S.az == S.az ? (unsigned long long)__double_as_longlong(S.az) : 0x7ff8000000000000ull;
            it[3] = (unsigned long long)(unsigned int)(q_off + (unsigned long long)S.qi) | ((unsigned long long)__float_as_uint(cand ? S.bound2 : INFINITY) << 32);
        }
    }
    // ---- moments of the proven queries: rounded to fixed point one by one, summed as integers, 19 atomics per wave
    {
        double m[PCR_NMOM];
#pragma unroll
        for (int k = 0; k < PCR_NMOM; ++k) m[k] = 0.0;
        if (proven) {
            if (nb_known) { nb.x = P.seed_b.x; nb.y = P.seed_b.y; nb.z = P.seed_b.z; }
            else prev_xyz[S.qi] = wt_xyz{nb.x, nb.y, nb.z};
            moments_add(m, bp->gv.origin, S.ax, S.ay, S.az, nb, max_d2, 1);
        }
        const double scale = bp->scale;
        if (lane < WT_Q) {
#pragma unroll
            for (int k = 0; k < PCR_NMOM - 1; ++k) L->xll[lane * (PCR_NMOM - 1) + k] = to_fixed(m[k], scale);
        }
        wave_sync();
        long long tot = 0;
        if (lane < PCR_NMOM - 1) {
#pragma unroll 8
            for (int j = 0; j < WT_Q; ++j) tot += L->xll[j * (PCR_NMOM - 1) + lane];
        }
        acc_fixed_add_ll(B.acc + (size_t)p * ACC_SETS * PCR_NMOM, tile, lane, tot);
    }
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PCR_WT_WAVES, 8)))
batch_drain_kernel(pcr_batch_pass_args B) {
    __shared__ pass_lds s_lds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    pass_lds* L = &s_lds[wave];
    const unsigned int l_cnt = lane < PASS_GROUPS ? (unsigned int)(B.sync[(size_t)PASS_SYNC_STRIDE * lane] & Q_MASK) : 0u;
    unsigned int total = 0;
    const unsigned int l_exc = wave_excl_scan_u32(l_cnt, lane, &total);
    const unsigned int n_waves = gridDim.x * 4;
    for (unsigned int w_i = blockIdx.x * 4 + wave; w_i < total; w_i += n_waves) {
        const int grp = (int)__ffsll((long long)__ballot(lane < PASS_GROUPS && l_exc <= w_i && w_i < l_exc + l_cnt)) - 1;
        const unsigned int idx = w_i - (unsigned int)__builtin_amdgcn_readlane((int)l_exc, grp);
        unsigned long long* it = B.items + ((size_t)grp * B.cap + idx) * 4;
        unsigned long long w = ITEM_NONE;
        if (lane < 4) {
            w = it[lane];
            it[lane] = ITEM_NONE;   // the slot is clean for the next pass
        }
        const unsigned int slot = (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)w, 3);   // global source slot of the query
        const unsigned int p = (unsigned int)__builtin_amdgcn_readfirstlane((int)as_global(B.tile_pair)[slot / WT_Q]);
        const pcr_batch_pair* const bp = batch_pair_ptr(B, p);
        pass_args A{};
        A.acc = B.acc + (size_t)p * ACC_SETS * PCR_NMOM;
        A.prev_xyz = (wt_xyz*)B.prev_xyz;
        A.scale = bp->scale;
        pass_serve_item(bp->gv, L, lane, w, B.max_d2, A, B.res_pos, nullptr, w_i, 0ull);
    }
}

// one wave per pair: totals of the pair's accumulators -> Procrustes step -> convergence test -> the pair's loop state
__global__ void __launch_bounds__(256)
batch_finish_kernel(pcr_batch_pass_args B, unsigned int pass_id) {
    __shared__ pass_lds s_lds[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && wave == 0 && lane < PASS_GROUPS) B.sync[(size_t)PASS_SYNC_STRIDE * lane] = 0ull;   // queue words: clean for the next pass
    const unsigned int p = (unsigned int)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + wave));
    if (p >= (unsigned int)B.n_pairs) return;
    pcr_icp_dev_state* const st = B.st + p;
    if (as_global(st)->stop) return;   // stopped before this pass: its tiles did nothing
    const pcr_batch_pair* const bp = batch_pair_ptr(B, p);
    if (bp->nq <= 0) return;
    pass_lds* L = &s_lds[wave];
    pass_args A{};
    A.acc = B.acc + (size_t)p * ACC_SETS * PCR_NMOM;
    A.st = st;
    A.la = B.la;
    A.inv_scale = bp->inv_scale;
    unsigned long long* const root = B.sync + (size_t)PASS_SYNC_STRIDE * PASS_GROUPS;   // (root + 16: the error word, never set here)
    pass_finish(bp->gv, L, lane, A, root, nullptr);
    wave_sync();
    if (lane == 0 && !reinterpret_cast<const pcr_icp_dev_state*>(L->xch + 32)->stop) atomicAdd(B.running + pass_id, 1u);
}

// everything a batch wants initialised, in one launch: item slots all-ones, accumulators / queue words / running counts zero,
// every pair's loop state built from its T0 (what pcr_grid_icp_loop writes for one pair)
__global__ void __launch_bounds__(256)
batch_init_kernel(unsigned long long* __restrict__ zero_p, unsigned long long zero_n, unsigned long long* __restrict__ ones_p, unsigned long long ones_n,
                  unsigned int* __restrict__ running, pcr_icp_dev_state* __restrict__ st, int n_pairs, const double* __restrict__ T0) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long i = t0; i < ones_n; i += stride) ones_p[i] = ~0ull;
    for (unsigned long long i = t0; i < zero_n; i += stride) zero_p[i] = 0ull;
    for (unsigned long long i = t0; i < PCR_ICP_MAX_LOG + 1; i += stride) running[i] = 0u;
    for (int p = blockIdx.x; p < n_pairs; p += gridDim.x) {
        pcr_icp_dev_state* s = st + p;
        unsigned long long* w = reinterpret_cast<unsigned long long*>(s);
        for (unsigned int i = threadIdx.x; i < sizeof(pcr_icp_dev_state) / 8; i += blockDim.x) w[i] = 0ull;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double* T = T0 + 16 * (size_t)p;
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) {
                    s->x.r[3 * i + j] = T[4 * i + j];
                    s->R_last[3 * i + j] = T[4 * i + j];
                }
                s->x.t[i] = T[4 * i + 3];
                s->t_last[i] = T[4 * i + 3];
            }
            for (int i = 0; i < 16; ++i) s->T_total[i] = (i % 5 == 0) ? 1.0 : 0.0;
            for (int i = 0; i < 9; ++i) s->V[i] = (i % 4 == 0) ? 1.0 : 0.0;
            s->first = 1;
        }
        __syncthreads();
    }
}

void pcr_grid_batch_scratch_bytes(unsigned int n_tiles, int n_pairs, size_t* items_bytes, size_t* acc_sync_bytes, size_t* sync_word, unsigned int* cap) {
    *sync_word = (size_t)n_pairs * ACC_SETS * PCR_NMOM;   // the queue words follow the accumulators
    const unsigned int c = pass_item_cap((long long)n_tiles * WT_Q);
    *cap = c;
    *items_bytes = 32 * (size_t)PASS_GROUPS * c;
    *acc_sync_bytes = sizeof(unsigned long long) * ((size_t)n_pairs * ACC_SETS * PCR_NMOM + PASS_SYNC_WORDS);
}

int pcr_grid_batch_init(pcr_ctx* ctx, const pcr_batch_pass_args* a, const double* d_T0) {
    // a->acc and a->sync are one allocation: [n_pairs][ACC_SETS][PCR_NMOM] accumulators, then the queue words
    const unsigned long long zero_n = (unsigned long long)a->n_pairs * ACC_SETS * PCR_NMOM + PASS_SYNC_WORDS;
    const unsigned long long ones_n = 4ull * PASS_GROUPS * a->cap;
    long long ib = (long long)((ones_n + 256 * 8 - 1) / (256 * 8));
    if (ib < a->n_pairs) ib = a->n_pairs;
    hipLaunchKernelGGL(batch_init_kernel, dim3((unsigned int)(ib < 1 ? 1 : (ib > 2048 ? 2048 : ib))), dim3(256), 0, ctx->stream, a->acc, zero_n, a->items, ones_n,
                       a->running, a->st, a->n_pairs, d_T0);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int pcr_grid_batch_pass(pcr_ctx* ctx, const pcr_batch_pass_args* a, unsigned int pass_id) {
    if (a->n_tiles == 0) return PCR_OK;
    static const int rounds_env = getenv("PCR_BATCH_WT_ROUNDS") ? atoi(getenv("PCR_BATCH_WT_ROUNDS")) : 0;
    const unsigned int pcap = (unsigned int)WT_PR * (rounds_env > 0 ? rounds_env : WT_ROUNDS_LARGE);
    static const bool no_prev = getenv("PCR_NO_PREV") != nullptr;
    pcr_prof_mark(ctx, 0);   // (pcr_profile_enable: HIP events around the three launches; slots 0..2 of pcr_profile_read)
    hipLaunchKernelGGL(batch_pass_kernel, dim3(a->n_tiles / 4), dim3(256), 0, ctx->stream, *a, wtile_xcd_remap(), pcap, pass_id, (pass_id > 0 && !no_prev) ? 1 : 0);
    pcr_prof_mark(ctx, 1);
    const long long want = (long long)a->n_tiles * WT_Q / 4;
    const int g3 = (int)(want < 8ll * ctx->cu_count ? (want < 1 ? 1 : want) : 8ll * ctx->cu_count);
    hipLaunchKernelGGL(batch_drain_kernel, dim3(g3), dim3(256), 0, ctx->stream, *a);
    pcr_prof_mark(ctx, 2);
    hipLaunchKernelGGL(batch_finish_kernel, dim3((a->n_pairs + 3) / 4), dim3(256), 0, ctx->stream, *a, pass_id);
    pcr_prof_mark(ctx, 3);
    pcr_prof_mark(ctx, 4);
    PCR_HIP(ctx, hipGetLastError());
    pcr_prof_finish(ctx);
    return PCR_OK;
}
