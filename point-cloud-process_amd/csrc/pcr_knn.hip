// Batched k-NN and radius queries on the voxel-hash grid: the device side of
// kdtree_knn_search / octree_knn_search (Kdtree_Octree/lesson2/kdtree.py:141-172,
// octree.py:262-306) and kdtree_radius_search / octree_radius_search(_fast)
// (kdtree.py:176-208, octree.py:166-259).  One wave per query walks the nested cell
// hierarchy top-down, pruning cells whose box is farther than the current bound.
//   k-NN   k rounds; round r finds the nearest point that is lexicographically greater in
//          (d2, index) than the result of round r-1 -- ascending distances, ties by index,
//          exactly the order of a stable sort of all distances.
//   radius bound fixed to r; a count pass, then a fill pass into caller-provided offsets.
// Distances are reported as sqrt((dx*dx+dy*dy)+dz*dz) in binary64 (result_set.py stores
// np.linalg.norm values).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "pcr_grid_dev.h"

constexpr int KN_STACK = 192;
constexpr unsigned int KN_SCAN_T = 128;

struct kn_entry {
    unsigned int start, end;
    unsigned int x, y, z;
    int level;
};

__device__ static inline double kn_sq_pos(double v) {
    v = fmax(v, 0.0);
    return v * v;
}

__device__ static inline double kn_box_dist2(const pcr_grid_view& gv, int level, double cell, unsigned int X, unsigned int Y, unsigned int Z,
                                             double ax, double ay, double az) {
    const int bl = (int)(PCR_COORD_BIAS >> (2 * level));
    const double slack = cell * 1e-9;
    const double x0 = gv.lo[0] + (double)((int)X - bl) * cell;
    const double y0 = gv.lo[1] + (double)((int)Y - bl) * cell;
    const double z0 = gv.lo[2] + (double)((int)Z - bl) * cell;
    const double dx = kn_sq_pos(fmax(x0 - ax, ax - (x0 + cell)) - slack);
    const double dy = kn_sq_pos(fmax(y0 - ay, ay - (y0 + cell)) - slack);
    const double dz = kn_sq_pos(fmax(z0 - az, az - (z0 + cell)) - slack);
    return (dx + dy) + dz;
}

// Pruned descent from the root cells.  `scan(start, end, bound2)` is called (wave-uniformly) for
// every cell that has to be read; it may lower bound2.
template <class Scan>
__device__ static inline void kn_descend(const pcr_grid_view& gv, double ax, double ay, double az, double& bound2, kn_entry* stack, int lane,
                                         Scan& scan) {
    const int top = gv.levels - 1;
    int sp = 0;
    {
        const double cell = gv.cell0 * (double)(1ll << (2 * top));
        const int b0 = (int)(PCR_COORD_BIAS >> (2 * top));
        const unsigned int X = b0 + (lane & 1), Y = b0 + ((lane >> 1) & 1), Z = b0 + ((lane >> 2) & 1);
        unsigned int s = 0, e = 0;
        bool valid = lane < 8;
        double bdist = 0.0;
        if (valid) {
            bdist = kn_box_dist2(gv, top, cell, X, Y, Z, ax, ay, az);
            valid = bdist <= bound2 && lookup_cell(gv.table[top], gv.mask[top], X, Y, Z, &s, &e);
        }
        const unsigned long long m_far = __ballot(valid && bdist > 0.0), m_near = __ballot(valid && !(bdist > 0.0));
        const unsigned long long below = (1ull << lane) - 1ull;
        int slot = -1;
        if (valid && bdist > 0.0) slot = __popcll(m_far & below);
        else if (valid) slot = __popcll(m_far) + __popcll(m_near & below);
        if (slot >= 0) stack[slot] = kn_entry{s, e, X, Y, Z, top};
        sp = __popcll(m_far) + __popcll(m_near);
    }
    while (sp > 0) {
        --sp;
        const kn_entry en = stack[sp];
        const double cell = gv.cell0 * (double)(1ll << (2 * en.level));
        if (kn_box_dist2(gv, en.level, cell, en.x, en.y, en.z, ax, ay, az) > bound2) continue;
        const unsigned int cnt = en.end - en.start;
        const bool room = sp + 64 <= KN_STACK;
        if (en.level == 0 || cnt <= KN_SCAN_T || !room) {
            scan(en.start, en.end, bound2);
        } else {
            const int cl = en.level - 1;
            const unsigned int X = en.x * 4u + (lane & 3), Y = en.y * 4u + ((lane >> 2) & 3), Z = en.z * 4u + (lane >> 4);
            const double bdist = kn_box_dist2(gv, cl, cell * 0.25, X, Y, Z, ax, ay, az);
            unsigned int s = 0, e = 0;
            const bool valid = bdist <= bound2 && lookup_cell(gv.table[cl], gv.mask[cl], X, Y, Z, &s, &e);
            const unsigned long long m_far = __ballot(valid && bdist > 0.0), m_near = __ballot(valid && !(bdist > 0.0));
            const unsigned long long below = (1ull << lane) - 1ull;
            int slot = -1;
            if (valid && bdist > 0.0) slot = __popcll(m_far & below);
            else if (valid) slot = __popcll(m_far) + __popcll(m_near & below);
            if (slot >= 0) stack[sp + slot] = kn_entry{s, e, X, Y, Z, cl};
            sp += __popcll(m_far) + __popcll(m_near);
        }
    }
}

struct knn_scan {
    const pcr_pt* pts;
    double ax, ay, az;
    double floor_d2;     // results must be lexicographically greater than (floor_d2, floor_id)
    long long floor_id;
    double bd2;
    long long bid;
    int lane;
    __device__ void operator()(unsigned int s, unsigned int e, double& bound2) {
        for (unsigned int j = s + lane; j < e; j += 64) {
            const pcr_pt b = pts[j];
            const double d2 = dist2(ax, ay, az, b);
            const bool above = d2 > floor_d2 || (d2 == floor_d2 && b.id > floor_id);
            if (above && better(d2, b.id, bd2, bid)) { bd2 = d2; bid = b.id; }
        }
        double m = bd2;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmin(m, __shfl_xor(m, off, 64));
        bound2 = fmin(bound2, m);
    }
};

__global__ void __launch_bounds__(256)
knn_kernel(pcr_grid_view gv, const double* __restrict__ queries, long long nq, int k, int* __restrict__ idx_out, double* __restrict__ dist_out,
           const int* __restrict__ redo_list, const unsigned int* __restrict__ redo_count) {
    __shared__ kn_entry s_stack[4][KN_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long qi = (long long)blockIdx.x * 4 + wave;
    if (redo_list) {  // second stage of the batched search: only the queries the block scan could not prove
        if (qi >= (long long)*redo_count) return;
        qi = redo_list[qi];
    }
    if (qi >= nq) return;
    knn_scan sc;
    sc.pts = gv.pts;
    sc.ax = queries[3 * qi];
    sc.ay = queries[3 * qi + 1];
    sc.az = queries[3 * qi + 2];
    sc.lane = lane;
    sc.floor_d2 = -1.0;
    sc.floor_id = -1;
    for (int r = 0; r < k; ++r) {
        sc.bd2 = DBL_MAX;
        sc.bid = 0x7fffffffffffffffll;
        double bound2 = DBL_MAX;
        kn_descend(gv, sc.ax, sc.ay, sc.az, bound2, s_stack[wave], lane, sc);
        double bd2 = sc.bd2;
        long long bid = sc.bid;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double od2 = __shfl_xor(bd2, off, 64);
            const long long oid = __shfl_xor(bid, off, 64);
            if (better(od2, oid, bd2, bid)) { bd2 = od2; bid = oid; }
        }
        const bool found = bd2 < DBL_MAX;
        if (lane == 0) {
            // unfilled slots keep KNNResultSet's initial (1e10, 0) (result_set.py:19-22)
            idx_out[qi * k + r] = found ? (int)bid : 0;
            dist_out[qi * k + r] = found ? sqrt(bd2) : 1e10;
        }
        if (!found) {
            if (lane == 0)
                for (int rr = r + 1; rr < k; ++rr) { idx_out[qi * k + rr] = 0; dist_out[qi * k + rr] = 1e10; }
            break;
        }
        sc.floor_d2 = bd2;
        sc.floor_id = bid;
    }
}

struct radius_scan {
    const pcr_pt* pts;
    double ax, ay, az, radius;
    int lane;
    unsigned int count;        // wave-uniform running count
    int* idx_out;              // null in the count pass
    double* dist_out;
    long long base;
    unsigned int cap = 0xffffffffu;   // entries this query may write (single-call variant: what lies beyond is counted only)
    __device__ void operator()(unsigned int s, unsigned int e, double& bound2) {
        (void)bound2;
        for (unsigned int j0 = s; j0 < e; j0 += 64) {
            const unsigned int j = j0 + lane;
            bool hit = false;
            double d = 0;
            long long id = 0;
            if (j < e) {
                const pcr_pt b = pts[j];
                d = sqrt(dist2(ax, ay, az, b));
                hit = !(d > radius);  // RadiusNNResultSet.add_point rejects dist > radius (result_set.py:80)
                id = b.id;
            }
            const unsigned long long m = __ballot(hit);
            if (idx_out && hit) {
                const unsigned int k = count + __popcll(m & ((1ull << lane) - 1ull));
                if (k < cap) {
                    idx_out[base + k] = (int)id;
                    dist_out[base + k] = d;
                }
            }
            count += __popcll(m);
        }
    }
};

__global__ void __launch_bounds__(256)
radius_kernel(pcr_grid_view gv, const double* __restrict__ queries, long long nq, double radius, long long* __restrict__ counts_out,
              const long long* __restrict__ offsets, int* __restrict__ idx_out, double* __restrict__ dist_out) {
    __shared__ kn_entry s_stack[4][KN_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long qi = (long long)blockIdx.x * 4 + wave;
    if (qi >= nq) return;
    radius_scan sc;
    sc.pts = gv.pts;
    sc.ax = queries[3 * qi];
    sc.ay = queries[3 * qi + 1];
    sc.az = queries[3 * qi + 2];
    sc.radius = radius;
    sc.lane = lane;
    sc.count = 0;
    sc.idx_out = offsets ? idx_out : nullptr;
    sc.dist_out = dist_out;
    sc.base = offsets ? offsets[qi] : 0;
    double bound2 = radius * radius * (1.0 + 1e-12);  // conservative cell pruning; membership is tested on sqrt(d2)
    kn_descend(gv, sc.ax, sc.ay, sc.az, bound2, s_stack[wave], lane, sc);
    if (lane == 0 && !offsets) counts_out[qi] = sc.count;
}



// Batched k-NN, first stage (k <= 16): ONE LANE per query scans the 3x3x3 block of cells around it and keeps the k best
// (d2, index) pairs in registers, sorted.  The block covers every point within the distance from the query to the
// block's nearest face (>= one cell), so the result is exact when the k-th distance does not exceed that; level 0 is
// tried first, then level 1 (cells 4x wider), then -- in sparse surroundings only -- levels 2 and 3.  Queries it cannot prove (sparse surroundings, fewer than k points in
// reach, coordinates outside the grid) go to a list for the wave-per-query descent above.  On a KITTI scan the block
// scan settles > 95 % of the queries at ~1/40 of the descent's cost per query.
#ifndef PCR_KNN_LV
#define PCR_KNN_LV 3
#endif
template <int K>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K <= 8 ? 4 : 2, 8)))   // (k <= 8: 129 VGPRs without the hint, one too many for four waves per SIMD)
knn_block_kernel(pcr_grid_view gv, const double* __restrict__ queries, long long nq, int k, int* __restrict__ idx_out,
                 double* __restrict__ dist_out, int* __restrict__ redo_list, unsigned int* __restrict__ redo_count,
                 const int* __restrict__ todo /* query ids to do (what the wave tiles left), or null: all */, const unsigned int* __restrict__ todo_count,
                 int lv_cap /* highest level this launch may climb to */,
                 double* __restrict__ redo_kth /* (or null) next to every redo id: the squared k-th distance this scan found, DBL_MAX = fewer than k points */) {
    const long long ti = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n_do = (todo && todo_count) ? (long long)*todo_count : nq;   // (a list without a count: all nq entries, e.g. the curve order)
    const long long qi = ti < n_do ? (todo ? (long long)todo[ti] : ti) : nq;
    bool proven = false;
    double kth_seen = DBL_MAX;
    double bd[K];
    long long bi[K];
    if (qi < nq) {
        const double ax = queries[3 * qi], ay = queries[3 * qi + 1], az = queries[3 * qi + 2];
        bool clamped = false;
        const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
        // levels 0, 1 and -- only where the surroundings are sparse (few points met one level down) -- 2 and 3
        const int max_level = min(gv.levels - 1 < PCR_KNN_LV ? gv.levels - 1 : PCR_KNN_LV, lv_cap);
        unsigned int met = 0;
        for (int level = 0; level <= max_level && !clamped && !proven && (level < 2 || met <= 192u); ++level) {
            met = 0;
#pragma unroll
            for (int j = 0; j < K; ++j) { bd[j] = DBL_MAX; bi[j] = 0x7fffffffffffffffll; }
            const int X0 = cx >> (2 * level), Y0 = cy >> (2 * level), Z0 = cz >> (2 * level);
            const int lim = (int)(PCR_COORD_MAX >> (2 * level));
            for (int c3 = 0; c3 < 9; ++c3) {
              const int Y = Y0 + (c3 % 3) - 1, Z = Z0 + (c3 / 3) - 1;
              if (Y < 0 || Z < 0 || Y > lim || Z > lim) continue;
              unsigned int s3[3] = {0, 0, 0}, e3[3] = {0, 0, 0};
              const unsigned int found = lookup_cell3(gv.table[level], gv.mask[level], (unsigned int)X0, (unsigned int)Y, (unsigned int)Z, (unsigned int)lim, s3, e3);
#pragma unroll
              for (int cx3 = 0; cx3 < 3; ++cx3) {
                if (!((found >> cx3) & 1u)) continue;
                const unsigned int s = s3[cx3], e = e3[cx3];
                met += e - s;
                // four records per trip, requested together: one thread walking a cell record by record is a chain of dependent
                // loads (2.2 ms for 120 000 queries, k = 8, at two waves per SIMD)
                for (unsigned int j0 = s; j0 < e; j0 += 4) {
                    pcr_pt rec[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {   // (assigned on every path: conditionally unassigned records become loop-carried registers)
                        rec[u] = pcr_pt{0.0, 0.0, 0.0, 0};
                        if (j0 + u < e) rec[u] = gv.pts[j0 + u];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (j0 + u >= e) break;
                        double d2 = dist2(ax, ay, az, rec[u]);
                        long long id = rec[u].id;
                        if (!better(d2, id, bd[K - 1], bi[K - 1])) continue;
                        // insertion into the sorted list with constant indices: the displaced element travels down
#pragma unroll
                        for (int t = 0; t < K; ++t) {
                            if (better(d2, id, bd[t], bi[t])) {
                                const double td = bd[t]; const long long ti = bi[t];
                                bd[t] = d2; bi[t] = id;
                                d2 = td; id = ti;
                            }
                        }
                    }
                }
              }
            }
            // radius the block certainly covers: distance to its nearest face
            const double cell = gv.cell0 * (double)(1ll << (2 * level));
            const int bl = (int)(PCR_COORD_BIAS >> (2 * level));
            const double a[3] = {ax, ay, az};
            const int C0[3] = {X0, Y0, Z0};
            double cover = DBL_MAX;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double f_lo = gv.lo[d] + (double)(C0[d] - 1 - bl) * cell, f_hi = gv.lo[d] + (double)(C0[d] + 2 - bl) * cell;
                cover = fmin(cover, fmin(a[d] - f_lo, f_hi - a[d]));
            }
            cover = fmax(cover - cell * 1e-9, 0.0);
            // the k-th neighbour (1-based k) must exist and lie inside the covered ball
            double kth = DBL_MAX;
#pragma unroll
            for (int j = 0; j < K; ++j) kth = (j == k - 1) ? bd[j] : kth;
            proven = kth <= cover * cover;
            kth_seen = kth;   // k points this close exist: the ball that holds the answer
        }
        if (proven) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (j < k) {
                    idx_out[qi * k + j] = (int)bi[j];
                    dist_out[qi * k + j] = sqrt(bd[j]);
                }
            }
        }
    }
    const bool redo = qi < nq && !proven;
    const unsigned long long m = __ballot(redo);
    if (m) {
        const int lane = threadIdx.x & 63;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(redo_count, (unsigned int)__popcll(m));
        base = __shfl(base, 0, 64);
        if (redo) {
            const unsigned int at = base + __popcll(m & ((1ull << lane) - 1ull));
            redo_list[at] = (int)qi;
            if (redo_kth) redo_kth[at] = kth_seen;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Batched k-NN (k <= 16), second stage: ONE QUERY PER WAVE (LPQ = 64 lanes per query; the kernel is written for any power of two,
// 64 / LPQ queries sharing a box).  The wave stages the cells of the query's box into its LDS slice, every lane scans its share
// of the staged points keeping its k best in registers, and the lanes' lists meet by shuffle exchanges.  Pass 0: the query's own
// level-0 cell and one ring.  A query whose k-th distance lies inside the ball the staged box certainly covers is proven; one
// that found k points knows a ball that holds its answer, and the next pass stages the box of that ball -- in cells of the
// finest level that keeps the box small --; one that still lacks k points asks for rings of 4, 16, 64 ... cells.
// Measured on one box (PCR_KNN_NO_TILES=1 = round 2's path: lane-per-query scan at every level, then the wave-per-query
// descent), 120 000-point scan: 120 000 other queries, k = 8: 1.76 against 3.05 ms; 20 000 off-surface queries: 1.5 against
// 9.7 ms (k = 8), 2.8 against 17 (k = 16) -- nothing reaches the descent any more, whose k full descents per query took
// milliseconds for a handful of far-out queries --; the scan against itself: a tie (1.09 / 1.14 ms at k = 8).
// Tiles of SEVERAL queries in curve order (64 x 1 lane, then 16 x 4 -> 4 x 16 -> 1 x 64 in three stages) were built first: the
// one-lane version lost outright (6.5 against 3.0 ms), the three-stage one won on other-cloud queries (1.56 ms) but lost 2 x on
// small k against itself (0.92 against 0.51 ms at k = 3: a sort and three launches in front of work the first scan does in 0.5 ms).
constexpr int KT_PTS = 384;         // points staged per round
constexpr int KT_LIST = 512;        // occupied cells of a tile's box
constexpr int KT_BOX = 2048;        // cells of a tile's box (occupied or not)

struct knn_tile_lds {
    double x[KT_PTS], y[KT_PTS], z[KT_PTS];
    int id[KT_PTS];
    unsigned int c_start[KT_LIST], c_pre[KT_LIST + 1];
};
__device__ static inline void kt_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ static inline int kt_wave_min(int v) { return wave_min_i32(v); }   // (DPP network: pcr_grid_dev.h)
__device__ static inline int kt_wave_max(int v) { return wave_max_i32(v); }

// LPQ lanes per query (64 / LPQ queries per wave), PASSES passes of at most ROUNDS staging rounds.  `order`: the query ids to do, n
// of them (n = *count_p when count_p is given: the list the first stage left).
template <int K, int LPQ, int PASSES, int ROUNDS>
__device__ static void knn_tile_one(const pcr_grid_view& gv, knn_tile_lds* L, const int lane, const long long tile, const long long nq,
                                    const double* __restrict__ queries, const unsigned int* __restrict__ order, int k, int* __restrict__ idx_out,
                                    double* __restrict__ dist_out, int* __restrict__ redo_list, unsigned int* __restrict__ redo_count,
                                    const double* __restrict__ seed_kth) {
    constexpr int QPT = 64 / LPQ;
    const int sub = lane & (LPQ - 1);
    const long long slot = tile * QPT + (lane / LPQ);
    const bool valid = slot < nq;
    const long long qi = valid ? (long long)order[slot] : 0;
    double ax = 0, ay = 0, az = 0;
    bool clamped = false;
    int c0[3] = {0, 0, 0};
    if (valid) {
        ax = queries[3 * qi]; ay = queries[3 * qi + 1]; az = queries[3 * qi + 2];
        c0[0] = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        c0[1] = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        c0[2] = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
    }
    bool open = valid && !clamped;     // still to be proven (the four lanes of a query agree)
    bool proven = false;
    double bd[K];
    long long bi[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = DBL_MAX; bi[j] = 0x7fffffffffffffffll; }
    const int top = gv.levels - 1;
    // `seed_kth` (the first stage's lists): that scan has been through the query's own cell and one ring -- pass 0 here would stage the
    // same 27 cells to learn the k-th distance it already knew.  Start with the ball of that distance (or, with fewer than k points
    // found, with the ring of 4 cells).
    const int first_pass = seed_kth ? 1 : 0;
    const double seed = (seed_kth && valid) ? seed_kth[slot] : DBL_MAX;
    for (int pass = first_pass; pass < PASSES; ++pass) {
        if (!__ballot(open)) break;
        int lo0[3], hi0[3];   // this query's box in level-0 cell coordinates
        {
            double kth = DBL_MAX;
#pragma unroll
            for (int j = 0; j < K; ++j) kth = (j == k - 1) ? bd[j] : kth;
            if (pass == first_pass && seed_kth) kth = seed;
            const double a[3] = {ax, ay, az};
            if (pass > 0 && kth < DBL_MAX) {
                const double r = sqrt(kth) * (1.0 + 1e-9) + gv.cell0 * 1e-6;
                bool cl = false;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    lo0[d] = cell_coord(a[d] - r, gv.lo[d], gv.inv_cell0, &cl);
                    hi0[d] = cell_coord(a[d] + r, gv.lo[d], gv.inv_cell0, &cl);
                }
            } else {
                const int ring = 1 << (2 * pass);
#pragma unroll
                for (int d = 0; d < 3; ++d) { lo0[d] = max(c0[d] - ring, 0); hi0[d] = min(c0[d] + ring, (int)PCR_COORD_MAX); }
            }
        }
        int w0[3], w1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            w0[d] = kt_wave_min(open ? lo0[d] : 0x7fffffff);
            w1[d] = kt_wave_max(open ? hi0[d] : 0);
        }
        int level = 0;
        long long ncell = 0;
        for (; level <= top; ++level) {
            ncell = (long long)((w1[0] >> (2 * level)) - (w0[0] >> (2 * level)) + 1) * ((w1[1] >> (2 * level)) - (w0[1] >> (2 * level)) + 1) *
                    ((w1[2] >> (2 * level)) - (w0[2] >> (2 * level)) + 1);
            if (ncell <= KT_BOX) break;
        }
        if (level > top) break;   // the tile's queries are spread too far for a common box (a jump of the curve)
        const int x0 = w0[0] >> (2 * level), y0 = w0[1] >> (2 * level), z0 = w0[2] >> (2 * level);
        const int dx = (w1[0] >> (2 * level)) - x0 + 1, dy = (w1[1] >> (2 * level)) - y0 + 1;
        const int b1[3] = {w1[0] >> (2 * level), w1[1] >> (2 * level), w1[2] >> (2 * level)};
        // ---- directory: occupied cells of the box and the exclusive prefix of their point counts
        unsigned int n_list = 0, n_pts = 0;
        bool too_much = false;
        for (int base = 0; base < (int)ncell; base += 64) {
            const int c = base + lane;
            unsigned int cs = 0, ce = 0;
            bool found = false;
            if (c < (int)ncell) {
                const int X = x0 + c % dx, Y = y0 + (c / dx) % dy, Z = z0 + c / (dx * dy);
                found = lookup_cell(gv.table[level], gv.mask[level], (unsigned int)X, (unsigned int)Y, (unsigned int)Z, &cs, &ce);
            }
            const unsigned long long fm = __ballot(found);
            const unsigned int cnt = found ? ce - cs : 0u, inc = wave_incl_scan_add(cnt);
            const unsigned int li = n_list + (unsigned int)__popcll(fm & ((1ull << lane) - 1ull));
            if (found && li < KT_LIST) { L->c_start[li] = cs; L->c_pre[li] = n_pts + inc - cnt; }
            n_list += (unsigned int)__popcll(fm);
            n_pts += (unsigned int)__builtin_amdgcn_readlane((int)inc, 63);
            if (n_list > KT_LIST) { too_much = true; break; }
        }
        if (too_much || n_pts > (unsigned int)(ROUNDS * KT_PTS)) break;   // too much for this stage
        if (lane == 0) L->c_pre[n_list] = n_pts;
#pragma unroll
        for (int j = 0; j < K; ++j) { bd[j] = DBL_MAX; bi[j] = 0x7fffffffffffffffll; }
        // ---- rounds: stage up to KT_PTS points; every lane scans a quarter of them for its query
        for (unsigned int r0 = 0; r0 < n_pts; r0 += KT_PTS) {
            const unsigned int cnt = min(n_pts - r0, (unsigned int)KT_PTS);
            kt_wave_sync();   // the directory (first round) / the last scan's reads are done
            for (unsigned int t = lane; t < cnt; t += 64) {
                const unsigned int j = r0 + t;
                unsigned int lo = 0, hi = n_list - 1;   // last cell whose prefix <= j
                while (lo < hi) {
                    const unsigned int mid = (lo + hi + 1) >> 1;
                    if (L->c_pre[mid] <= j) lo = mid;
                    else hi = mid - 1;
                }
                const pcr_pt rec = gv.pts[L->c_start[lo] + (j - L->c_pre[lo])];
                L->x[t] = rec.x; L->y[t] = rec.y; L->z[t] = rec.z; L->id[t] = (int)rec.id;
            }
            kt_wave_sync();
            if (open) {
                for (unsigned int t = sub; t < cnt; t += LPQ) {
                    const double ex = ax - L->x[t], ey = ay - L->y[t], ez = az - L->z[t];
                    double d2 = (ex * ex + ey * ey) + ez * ez;
                    long long id = (long long)L->id[t];
                    if (!better(d2, id, bd[K - 1], bi[K - 1])) continue;
#pragma unroll
                    for (int u = 0; u < K; ++u) {
                        if (better(d2, id, bd[u], bi[u])) {
                            const double td = bd[u]; const long long ti = bi[u];
                            bd[u] = d2; bi[u] = id;
                            d2 = td; id = ti;
                        }
                    }
                }
            }
        }
        kt_wave_sync();
        // ---- the lanes' lists of a query meet.  One query per wave: k times over, the smallest HEAD of the 64 sorted lists (DPP minimum,
        // ties to the lowest id) is the next entry of the merged list and leaves its lane's list -- ~70 instructions per entry.  (The
        // butterfly below inserts every entry of the partner's list, six rounds of K x K compare-and-swaps and 4 K ds_bpermute: 12 000
        // instructions per pass at K = 16, most of the wave-per-query stage's time.)
        if (LPQ == 64) {
            double md[K];
            long long mi[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                md[j] = DBL_MAX; mi[j] = 0x7fffffffffffffffll;
                const double m = j < k ? wave_min_f64(bd[0]) : DBL_MAX;
                if (m < DBL_MAX) {   // (wave-uniform; DBL_MAX: the lists are exhausted)
                    unsigned long long cm = __ballot(bd[0] == m);
                    int wl = (int)__ffsll((long long)cm) - 1;
                    cm &= cm - 1;
                    long long wid = readlane_i64(bi[0], wl);
                    while (cm) {   // equal distances: the lowest id
                        const int l = (int)__ffsll((long long)cm) - 1;
                        cm &= cm - 1;
                        const long long id_l = readlane_i64(bi[0], l);
                        if (id_l < wid) { wid = id_l; wl = l; }
                    }
                    md[j] = m; mi[j] = wid;
                    if (lane == wl) {
#pragma unroll
                        for (int u = 0; u + 1 < K; ++u) { bd[u] = bd[u + 1]; bi[u] = bi[u + 1]; }
                        bd[K - 1] = DBL_MAX; bi[K - 1] = 0x7fffffffffffffffll;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < K; ++j) { bd[j] = md[j]; bi[j] = mi[j]; }
        }
#pragma unroll
        for (int xm = 1; xm < (LPQ == 64 ? 1 : LPQ); xm <<= 1) {
            double od[K];
            long long oi[K];
#pragma unroll
            for (int j = 0; j < K; ++j) { od[j] = __shfl_xor(bd[j], xm, 64); oi[j] = __shfl_xor(bi[j], xm, 64); }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                double d2 = od[j];
                long long id = oi[j];
                if (!better(d2, id, bd[K - 1], bi[K - 1])) continue;   // (the partner's list is sorted: the rest is no better)
#pragma unroll
                for (int u = 0; u < K; ++u) {
                    if (better(d2, id, bd[u], bi[u])) {
                        const double td = bd[u]; const long long ti = bi[u];
                        bd[u] = d2; bi[u] = id;
                        d2 = td; id = ti;
                    }
                }
            }
        }
        // ---- radius the staged box certainly covers: distance to its nearest face
        if (open) {
            const double cell = gv.cell0 * (double)(1ll << (2 * level));
            const int bl = (int)(PCR_COORD_BIAS >> (2 * level));
            const double a[3] = {ax, ay, az};
            const int b0[3] = {x0, y0, z0};
            double cover = DBL_MAX;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double f_lo = gv.lo[d] + (double)(b0[d] - bl) * cell, f_hi = gv.lo[d] + (double)(b1[d] + 1 - bl) * cell;
                cover = fmin(cover, fmin(a[d] - f_lo, f_hi - a[d]));
            }
            cover = fmax(cover - cell * 1e-9, 0.0);
            double kth = DBL_MAX;
#pragma unroll
            for (int j = 0; j < K; ++j) kth = (j == k - 1) ? bd[j] : kth;
            if (kth <= cover * cover) {
                proven = true;
                open = false;
                if (sub == 0) {
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (j < k) {
                            idx_out[qi * k + j] = (int)bi[j];
                            dist_out[qi * k + j] = sqrt(bd[j]);
                        }
                    }
                }
            }
        }
    }
    const bool redo = valid && !proven && sub == 0;
    const unsigned long long m = __ballot(redo);
    if (m) {
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(redo_count, (unsigned int)__popcll(m));
        base = __shfl(base, 0, 64);
        if (redo) redo_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)qi;
    }
}

// a fixed grid of waves strides over the tiles: their number is only known on the device (the list the first stage left), and one
// block per four POSSIBLE tiles was 30 000 blocks that read the count and left (0.1 ms of a 0.5-ms call)
template <int K, int LPQ, int PASSES, int ROUNDS>
__global__ void __launch_bounds__(256)
knn_tile_kernel(pcr_grid_view gv, const double* __restrict__ queries, const unsigned int* __restrict__ order, long long n_static, const unsigned int* __restrict__ count_p,
                int k, int* __restrict__ idx_out, double* __restrict__ dist_out, int* __restrict__ redo_list, unsigned int* __restrict__ redo_count,
                const double* __restrict__ seed_kth /* (or null) per entry of `order`: what the first stage knows about the k-th distance */) {
    __shared__ knn_tile_lds s_lds[4];
    constexpr int QPT = 64 / LPQ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nq = count_p ? (long long)*count_p : n_static;
    const long long n_tiles = (nq + QPT - 1) / QPT;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long long)gridDim.x * 4) {
        knn_tile_one<K, LPQ, PASSES, ROUNDS>(gv, &s_lds[wave], lane, tile, nq, queries, order, k, idx_out, dist_out, redo_list, redo_count, seed_kth);
        kt_wave_sync();
    }
}

// A handful of queries in ONE launch (the reference's API is one query per call, kdtree.py:176-208, octree.py:166-259): every
// query's neighbours go, unordered, straight into a pinned, device-mapped block of `cap` entries per query (what lies beyond is
// only counted); the host orders them.  No count pass, no offsets, no device scratch, no copy.
__global__ void __launch_bounds__(256)
radius_small_kernel(pcr_grid_view gv, const double* __restrict__ queries, int nq, double radius, unsigned int cap, unsigned int* __restrict__ counts_out,
                    int* __restrict__ idx_out, double* __restrict__ dist_out) {
    __shared__ kn_entry s_stack[4][KN_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qi = (int)blockIdx.x * 4 + wave;
    if (qi >= nq) return;
    radius_scan sc;
    sc.pts = gv.pts;
    sc.ax = queries[3 * qi];
    sc.ay = queries[3 * qi + 1];
    sc.az = queries[3 * qi + 2];
    sc.radius = radius;
    sc.lane = lane;
    sc.count = 0;
    sc.idx_out = idx_out;
    sc.dist_out = dist_out;
    sc.base = (long long)qi * cap;
    sc.cap = cap;
    double bound2 = radius * radius * (1.0 + 1e-12);
    kn_descend(gv, sc.ax, sc.ay, sc.az, bound2, s_stack[wave], lane, sc);
    if (lane == 0) counts_out[qi] = sc.count;
}

// pinned, device-mapped landing block for the few-query paths (grown on demand, kept with the context)
static int small_block(pcr_ctx* ctx, size_t bytes, char** host, char** dev) {
    if (ctx->h_big_bytes < bytes) {
        if (ctx->h_big) hipHostFree(ctx->h_big);
        ctx->h_big = nullptr;
        ctx->h_big_bytes = 0;
        const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        if (hipHostMalloc(&ctx->h_big, want, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); ctx->h_big = nullptr; return PCR_E_NOMEM; }
        ctx->h_big_bytes = want;
    }
    void* dp = nullptr;
    PCR_HIP(ctx, hipHostGetDevicePointer(&dp, ctx->h_big, 0));
    *host = (char*)ctx->h_big;
    *dev = (char*)dp;
    return PCR_OK;
}

extern "C" {

int pcr_radius_small(pcr_ctx* ctx, const pcr_index* index, const double* queries, int q, double radius, int64_t cap, int64_t* counts_out,
                     int32_t* idx_out, double* dist_out) {
    if (!ctx || !index || !queries || !counts_out || !idx_out || !dist_out || !(radius >= 0) || q < 1 || q > 64 || cap < 1 || cap > (1 << 20)) return PCR_E_INVALID;
    if (index->kind != PCR_INDEX_GRID) return PCR_E_UNSUPPORTED;
    hipSetDevice(ctx->device);
    // landing block: queries | counts | indices | distances
    const size_t o_cnt = 24 * (size_t)q, o_idx = (o_cnt + 4 * (size_t)q + 63) & ~(size_t)63, o_d = (o_idx + 4 * (size_t)q * cap + 63) & ~(size_t)63;
    const size_t bytes = o_d + 8 * (size_t)q * cap;
    char *h = nullptr, *d = nullptr;
    int rc = small_block(ctx, bytes, &h, &d);
    if (rc) return rc;
    memcpy(h, queries, 24 * (size_t)q);
    hipLaunchKernelGGL(radius_small_kernel, dim3((unsigned)((q + 3) / 4)), dim3(256), 0, ctx->stream, index->view, (const double*)d, q, radius, (unsigned int)cap,
                       (unsigned int*)(d + o_cnt), (int*)(d + o_idx), (double*)(d + o_d));
    PCR_HIP(ctx, hipGetLastError());
    if ((rc = pcr_wait_flag(ctx, nullptr))) return rc;
    const unsigned int* cnt = (const unsigned int*)(h + o_cnt);
    bool over = false;
    for (int i = 0; i < q; ++i) { counts_out[i] = cnt[i]; over = over || cnt[i] > (unsigned int)cap; }
    if (over) return PCR_E_UNSUPPORTED;   // counts_out holds the true counts: the two-pass pcr_radius takes such queries
    // ascending distance, ties by index (result_set.py:78-84 keeps insertion order; the reference's lists are sorted by the caller)
    std::vector<std::pair<double, int>> tmp;
    int64_t out = 0;
    for (int i = 0; i < q; ++i) {
        const int* ii = (const int*)(h + o_idx) + (size_t)i * cap;
        const double* dd = (const double*)(h + o_d) + (size_t)i * cap;
        tmp.resize(cnt[i]);
        for (unsigned int k = 0; k < cnt[i]; ++k) tmp[k] = std::make_pair(dd[k], ii[k]);
        std::sort(tmp.begin(), tmp.end());
        for (unsigned int k = 0; k < cnt[i]; ++k) { dist_out[out + k] = tmp[k].first; idx_out[out + k] = tmp[k].second; }
        out += cnt[i];
    }
    return PCR_OK;
}

int pcr_knn(pcr_ctx* ctx, const pcr_index* index, const double* queries, int64_t q, int k, int32_t* idx_out, double* dist_out) {
    if (!ctx || !index || !queries || !idx_out || !dist_out || k <= 0) return PCR_E_INVALID;
    if (q <= 0) return PCR_OK;
    if (index->kind != PCR_INDEX_GRID) return PCR_E_UNSUPPORTED;
    hipSetDevice(ctx->device);
    int rc;
    static const bool no_small = getenv("PCR_KNN_NO_SMALL") != nullptr;
    if (q <= 16 && k <= 16 && !no_small) {
        // A handful of queries (the reference's API is one per call, kdtree.py:141-172, octree.py:262-306): one query per wave with all
        // 64 lanes on the query's own box (knn_tile_kernel, the second stage of the batched path), queries read from and results
        // written to a pinned, device-mapped block -- one launch, no copies, no device scratch; what it cannot prove (clamped
        // coordinates, fewer than k points) goes through the descent kernel, as in the batched path.
        const size_t o_ord = 24 * (size_t)q, o_cnt = o_ord + 4 * (size_t)q, o_redo = o_cnt + 64, o_idx = (o_redo + 4 * (size_t)q + 63) & ~(size_t)63;
        const size_t o_d = (o_idx + 4 * (size_t)q * k + 63) & ~(size_t)63, bytes = o_d + 8 * (size_t)q * k;
        char *h = nullptr, *d = nullptr;
        if ((rc = small_block(ctx, bytes, &h, &d))) return rc;
        memcpy(h, queries, 24 * (size_t)q);
        for (int64_t i = 0; i < q; ++i) ((unsigned int*)(h + o_ord))[i] = (unsigned int)i;
        *(unsigned int*)(h + o_cnt) = 0u;
        const unsigned gw = (unsigned)((q + 3) / 4);
        if (k <= 8)
            hipLaunchKernelGGL((knn_tile_kernel<8, 64, 7, 512>), dim3(gw), dim3(256), 0, ctx->stream, index->view, (const double*)d, (const unsigned int*)(d + o_ord), (long long)q,
                               (const unsigned int*)nullptr, k, (int*)(d + o_idx), (double*)(d + o_d), (int*)(d + o_redo), (unsigned int*)(d + o_cnt), (const double*)nullptr);
        else
            hipLaunchKernelGGL((knn_tile_kernel<16, 64, 7, 512>), dim3(gw), dim3(256), 0, ctx->stream, index->view, (const double*)d, (const unsigned int*)(d + o_ord), (long long)q,
                               (const unsigned int*)nullptr, k, (int*)(d + o_idx), (double*)(d + o_d), (int*)(d + o_redo), (unsigned int*)(d + o_cnt), (const double*)nullptr);
        PCR_HIP(ctx, hipGetLastError());
        if ((rc = pcr_wait_flag(ctx, nullptr))) return rc;
        if (*(volatile unsigned int*)(h + o_cnt) != 0u) {
            hipLaunchKernelGGL(knn_kernel, dim3(gw), dim3(256), 0, ctx->stream, index->view, (const double*)d, (long long)q, k, (int*)(d + o_idx), (double*)(d + o_d),
                               (const int*)(d + o_redo), (const unsigned int*)(d + o_cnt));
            PCR_HIP(ctx, hipGetLastError());
            if ((rc = pcr_wait_flag(ctx, nullptr))) return rc;
        }
        memcpy(idx_out, h + o_idx, 4 * (size_t)q * k);
        memcpy(dist_out, h + o_d, 8 * (size_t)q * k);
        return PCR_OK;
    }
    pcr_dev_block b_q(ctx), b_idx(ctx), b_dist(ctx), b_redo(ctx);   // (back to the arena on every return path)
    if ((rc = b_q.alloc(sizeof(double) * 3 * q)) || (rc = b_idx.alloc(sizeof(int) * q * k)) || (rc = b_dist.alloc(sizeof(double) * q * k))) return rc;
    const double* d_q = b_q.as<const double>();
    int* d_idx = b_idx.as<int>();
    double* d_dist = b_dist.as<double>();
    PCR_HIP(ctx, hipMemcpyAsync(b_q.p, queries, sizeof(double) * 3 * q, hipMemcpyHostToDevice, ctx->stream));
    static const bool no_block = getenv("PCR_KNN_NO_BLOCK") != nullptr;
    static const bool no_tiles = getenv("PCR_KNN_NO_TILES") != nullptr;   // A/B: the lane-per-query kernel at every level, then the descent (round 2)
    if (k <= 16 && q >= 256 && !no_block) {
        // Batched path, three stages, each handing what it cannot prove to the next through a list:
        //   1. the lane-per-query scan at the finest level(s) only -- a query's own 27 cells: where it is fast and proves the easy majority;
        //   2. ONE QUERY PER WAVE for what it leaves: all 64 lanes scan the query's own box -- its cells and a ring, then the ball its
        //      k-th distance so far defines, or rings of 4, 16, 64 ... cells while it lacks k points (knn_tile_kernel);
        //   3. the lane-per-query scan at every level and the wave-per-query descent for what is left (clamped coordinates).
        unsigned int* d_cnt_a = ctx->d_counters + 125;   // queries stage 2 left
        unsigned int* d_cnt_b = ctx->d_counters + 126;   // queries the full block scan left (-> descent)
        unsigned int* d_cnt_d = ctx->d_counters + 127;   // queries stage 1 left
        pcr_dev_block b_redo2(ctx), b_redo4(ctx), b_kth(ctx);
        if ((rc = b_redo.alloc(sizeof(int) * q)) || (rc = b_redo2.alloc(sizeof(int) * q)) || (rc = b_redo4.alloc(sizeof(int) * q)) || (rc = b_kth.alloc(sizeof(double) * q))) return rc;
        static const bool no_seed = getenv("PCR_KNN_NO_SEED") != nullptr;   // A/B: the second stage starts over at the query's own 27 cells
        double* d_kth = no_seed ? nullptr : b_kth.as<double>();
        int *d_redo_a = b_redo.as<int>(), *d_redo_b = b_redo2.as<int>(), *d_redo_d = b_redo4.as<int>();
        PCR_HIP(ctx, hipMemsetAsync(d_cnt_a, 0, 3 * sizeof(unsigned int), ctx->stream));
        const unsigned gb = (unsigned)((q + 255) / 256);
        if (!no_tiles) {
            const int lv_first = k > 8 ? 1 : 0;   // (k = 16, a scan against itself: 44 000 of 120 000 queries are left at level 0)
            unsigned gw = (unsigned)((q + 3) / 4);
            if (gw > 8u * (unsigned)ctx->cu_count) gw = 8u * (unsigned)ctx->cu_count;   // (a fixed grid strides over the list: its length is only known on the device)
            if (k <= 8) {
                hipLaunchKernelGGL(knn_block_kernel<8>, dim3(gb), dim3(256), 0, ctx->stream, index->view, d_q, (long long)q, k, d_idx, d_dist, d_redo_d, d_cnt_d,
                                   (const int*)nullptr, (const unsigned int*)nullptr, lv_first, d_kth);
                hipLaunchKernelGGL((knn_tile_kernel<8, 64, 7, 512>), dim3(gw), dim3(256), 0, ctx->stream, index->view, d_q, (const unsigned int*)d_redo_d, (long long)q,
                                   (const unsigned int*)d_cnt_d, k, d_idx, d_dist, d_redo_a, d_cnt_a, (const double*)d_kth);
            } else {
                hipLaunchKernelGGL(knn_block_kernel<16>, dim3(gb), dim3(256), 0, ctx->stream, index->view, d_q, (long long)q, k, d_idx, d_dist, d_redo_d, d_cnt_d,
                                   (const int*)nullptr, (const unsigned int*)nullptr, lv_first, d_kth);
                hipLaunchKernelGGL((knn_tile_kernel<16, 64, 7, 512>), dim3(gw), dim3(256), 0, ctx->stream, index->view, d_q, (const unsigned int*)d_redo_d, (long long)q,
                                   (const unsigned int*)d_cnt_d, k, d_idx, d_dist, d_redo_a, d_cnt_a, (const double*)d_kth);
            }
        }
        const int* todo = no_tiles ? nullptr : d_redo_a;
        if (k <= 8)
            hipLaunchKernelGGL(knn_block_kernel<8>, dim3(gb), dim3(256), 0, ctx->stream, index->view, d_q, (long long)q, k, d_idx, d_dist, d_redo_b, d_cnt_b, todo,
                               (const unsigned int*)d_cnt_a, PCR_KNN_LV, (double*)nullptr);
        else
            hipLaunchKernelGGL(knn_block_kernel<16>, dim3(gb), dim3(256), 0, ctx->stream, index->view, d_q, (long long)q, k, d_idx, d_dist, d_redo_b, d_cnt_b, todo,
                               (const unsigned int*)d_cnt_a, PCR_KNN_LV, (double*)nullptr);
        unsigned int n_redo[3] = {0, 0, 0};
        { const int rc_n = pcr_d2h_small(ctx, n_redo, d_cnt_a, 3 * sizeof(unsigned int)); if (rc_n) return rc_n; }   // (synchronises; no copy engine)
        static const bool dbg = getenv("PCR_KNN_DEBUG") != nullptr;
        if (dbg) fprintf(stderr, "pcr_knn: %lld queries, k = %d: %u left by the first scan, %u by the wave-per-query boxes, %u to the descent\n", (long long)q, k, n_redo[2], n_redo[0], n_redo[1]);
        if (n_redo[1])
            hipLaunchKernelGGL(knn_kernel, dim3((n_redo[1] + 3) / 4), dim3(256), 0, ctx->stream, index->view, d_q, (long long)q, k, d_idx, d_dist, (const int*)d_redo_b,
                               (const unsigned int*)d_cnt_b);
    } else {
        hipLaunchKernelGGL(knn_kernel, dim3((unsigned)((q + 3) / 4)), dim3(256), 0, ctx->stream, index->view, d_q, (long long)q, k, d_idx, d_dist, (const int*)nullptr,
                           (const unsigned int*)nullptr);
    }
    PCR_HIP(ctx, hipGetLastError());
    if ((rc = pcr_d2h_staged(ctx, idx_out, d_idx, sizeof(int) * (size_t)q * k))) return rc;
    if ((rc = pcr_d2h_staged(ctx, dist_out, d_dist, sizeof(double) * (size_t)q * k))) return rc;
    PCR_HIP(ctx, pcr_sync(ctx->stream));
    return PCR_OK;
}

int pcr_radius(pcr_ctx* ctx, const pcr_index* index, const double* queries, int64_t q, double radius, int64_t* counts_out,
               const int64_t* offsets, int32_t* idx_out, double* dist_out) {
    if (!ctx || !index || !queries || !(radius >= 0)) return PCR_E_INVALID;
    if (!offsets && !counts_out) return PCR_E_INVALID;
    if (offsets && (!idx_out || !dist_out)) return PCR_E_INVALID;
    if (q <= 0) return PCR_OK;
    if (index->kind != PCR_INDEX_GRID) return PCR_E_UNSUPPORTED;
    hipSetDevice(ctx->device);
    // (every scratch block goes back to the arena on every return path)
    pcr_dev_block d_q(ctx), d_counts(ctx), d_offs(ctx), d_idx(ctx), d_dist(ctx), d_idx2(ctx), d_dist2(ctx), d_tmp(ctx);
    int rc;
    if ((rc = d_q.alloc(sizeof(double) * 3 * q))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(d_q.p, queries, sizeof(double) * 3 * q, hipMemcpyHostToDevice, ctx->stream));
    const unsigned grid = (unsigned)((q + 3) / 4);
    if (!offsets) {
        if ((rc = d_counts.alloc(sizeof(long long) * q))) return rc;
        hipLaunchKernelGGL(radius_kernel, dim3(grid), dim3(256), 0, ctx->stream, index->view, d_q.as<const double>(), (long long)q, radius,
                           d_counts.as<long long>(), (const long long*)nullptr, (int*)nullptr, (double*)nullptr);
        PCR_HIP(ctx, hipGetLastError());
        PCR_HIP(ctx, hipMemcpyAsync(counts_out, d_counts.p, sizeof(long long) * q, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, pcr_sync(ctx->stream));
        return PCR_OK;
    }
    const int64_t total = offsets[q];
    if (total < 0) return PCR_E_INVALID;
    const bool rt_on = getenv("PCR_RADIUS_TIMING") != nullptr;
    auto rt_now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double rt[6] = {rt_now(), 0, 0, 0, 0, 0};
    // the segmented sorts below index with 32 bits: more than 2^32 - 1 neighbours in one call (it fits in 288 GB) is refused
    // rather than silently truncated
    if (total > 0xffffffffll) { ctx->last_error = "pcr_radius: more than 2^32 - 1 neighbours in one call; split the queries"; return PCR_E_UNSUPPORTED; }
    if ((rc = d_offs.alloc(sizeof(long long) * (q + 1))) || (rc = d_idx.alloc(sizeof(int) * (total + 1))) || (rc = d_dist.alloc(sizeof(double) * (total + 1)))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(d_offs.p, offsets, sizeof(long long) * (q + 1), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(radius_kernel, dim3(grid), dim3(256), 0, ctx->stream, index->view, d_q.as<const double>(), (long long)q, radius,
                       (long long*)nullptr, d_offs.as<const long long>(), d_idx.as<int>(), d_dist.as<double>());
    PCR_HIP(ctx, hipGetLastError());
    // Ascending distance (ties by index) inside every query's segment, on the device: two stable segmented radix sorts (by
    // index, then by distance).  The host used to sort the 15.7 M neighbours of 20 000 queries on 16 threads after a pageable
    // 190-MB copy; now the lists arrive ordered, through a pinned double buffer.
    if (total > 0) {
        if ((rc = d_idx2.alloc(sizeof(int) * (total + 1))) || (rc = d_dist2.alloc(sizeof(double) * (total + 1)))) return rc;
        int *i1 = d_idx.as<int>(), *i2 = d_idx2.as<int>();
        double *x1 = d_dist.as<double>(), *x2 = d_dist2.as<double>();
        long long* offs = d_offs.as<long long>();
        size_t tb1 = 0, tb2 = 0;
        PCR_HIP(ctx, rocprim::segmented_radix_sort_pairs(nullptr, tb1, i1, i2, x1, x2, (unsigned int)total, (unsigned int)q, offs, offs + 1, 0, 32, ctx->stream));
        PCR_HIP(ctx, rocprim::segmented_radix_sort_pairs(nullptr, tb2, x2, x1, i2, i1, (unsigned int)total, (unsigned int)q, offs, offs + 1, 0, 64, ctx->stream));
        const size_t tb = tb1 > tb2 ? tb1 : tb2;
        if ((rc = d_tmp.alloc(tb > 0 ? tb : 16))) return rc;
        PCR_HIP(ctx, rocprim::segmented_radix_sort_pairs(d_tmp.p, tb1, i1, i2, x1, x2, (unsigned int)total, (unsigned int)q, offs, offs + 1, 0, 32, ctx->stream));
        PCR_HIP(ctx, rocprim::segmented_radix_sort_pairs(d_tmp.p, tb2, x2, x1, i2, i1, (unsigned int)total, (unsigned int)q, offs, offs + 1, 0, 64, ctx->stream));
        if (rt_on) { pcr_sync(ctx->stream); rt[1] = rt_now(); }
        if ((rc = pcr_d2h_staged(ctx, idx_out, i1, sizeof(int) * (size_t)total))) return rc;
        if (rt_on) rt[2] = rt_now();
        if ((rc = pcr_d2h_staged(ctx, dist_out, x1, sizeof(double) * (size_t)total))) return rc;
        if (rt_on) { rt[3] = rt_now(); fprintf(stderr, "pcr_radius: kernels + sorts %.2f ms, indices out %.2f ms (%.1f MB), distances out %.2f ms (%.1f MB)\n", rt[1] - rt[0], rt[2] - rt[1], 4e-6 * total, rt[3] - rt[2], 8e-6 * total); }
    }
    PCR_HIP(ctx, pcr_sync(ctx->stream));
    return PCR_OK;
}


// DBSCAN.fit of Cluster_dbscan/dbscan.py:10-36 -- a consumer of the radius query.  All N radius queries
// run on the device (count pass + fill pass of radius_kernel); the labelling loop is the reference's own sequential
// traversal, run on the host over the neighbour lists, so labels match the reference's including its quirks: seeds are
// popped from the END of the index list, a seed needs >= min_pts neighbours (self included) but a reached point only
// expands with > min_pts, and a point first popped as a noise seed is never relabelled.
int pcr_dbscan(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, int min_pts, int32_t* labels_out, int32_t* n_clusters_out) {
    if (!ctx || !cloud || !labels_out || !(radius >= 0)) return PCR_E_INVALID;
    const int64_t n = cloud->n;
    if (n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    std::vector<double> xyz(3 * (size_t)n);
    int rc = pcr_cloud_download_f64(ctx, cloud, xyz.data());
    if (rc) return rc;
    pcr_index* index = nullptr;
    if ((rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, radius > 0 ? radius : 0.0, &index))) return rc;
    double* d_q = nullptr;
    long long *d_counts = nullptr, *d_offs = nullptr;
    int* d_idx = nullptr;
    double* d_dist = nullptr;
    std::vector<long long> counts((size_t)n), offs((size_t)n + 1, 0);
    std::vector<int> nbr;
    const unsigned grid = (unsigned)((n + 3) / 4);
    do {
        if ((rc = pcr_dev_alloc(ctx, sizeof(double) * 3 * n, (void**)&d_q))) break;
        if ((rc = pcr_dev_alloc(ctx, sizeof(long long) * n, (void**)&d_counts))) break;
        if ((rc = pcr_dev_alloc(ctx, sizeof(long long) * (n + 1), (void**)&d_offs))) break;
        hipMemcpyAsync(d_q, xyz.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream);
        hipLaunchKernelGGL(radius_kernel, dim3(grid), dim3(256), 0, ctx->stream, index->view, (const double*)d_q, (long long)n, radius, d_counts,
                           (const long long*)nullptr, (int*)nullptr, (double*)nullptr);
        hipMemcpyAsync(counts.data(), d_counts, sizeof(long long) * n, hipMemcpyDeviceToHost, ctx->stream);
        if (pcr_sync(ctx->stream) != hipSuccess) { rc = PCR_E_HIP; break; }
        for (int64_t i = 0; i < n; ++i) offs[(size_t)i + 1] = offs[(size_t)i] + counts[(size_t)i];
        const long long total = offs[(size_t)n];
        if ((rc = pcr_dev_alloc(ctx, sizeof(int) * (total + 1), (void**)&d_idx))) break;
        if ((rc = pcr_dev_alloc(ctx, sizeof(double) * (total + 1), (void**)&d_dist))) break;
        hipMemcpyAsync(d_offs, offs.data(), sizeof(long long) * (n + 1), hipMemcpyHostToDevice, ctx->stream);
        hipLaunchKernelGGL(radius_kernel, dim3(grid), dim3(256), 0, ctx->stream, index->view, (const double*)d_q, (long long)n, radius,
                           (long long*)nullptr, (const long long*)d_offs, d_idx, d_dist);
        nbr.resize((size_t)total + 1);
        hipMemcpyAsync(nbr.data(), d_idx, sizeof(int) * total, hipMemcpyDeviceToHost, ctx->stream);
        if (pcr_sync(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) { rc = PCR_E_HIP; break; }
        pcr_dev_free(ctx, d_idx, sizeof(int) * (total + 1));
        pcr_dev_free(ctx, d_dist, sizeof(double) * (total + 1));
        d_idx = nullptr; d_dist = nullptr;
    } while (0);
    if (d_q) pcr_dev_free(ctx, d_q, sizeof(double) * 3 * n);
    if (d_counts) pcr_dev_free(ctx, d_counts, sizeof(long long) * n);
    if (d_offs) pcr_dev_free(ctx, d_offs, sizeof(long long) * (n + 1));
    pcr_index_free(ctx, index);
    if (rc) return rc;
    // dbscan.py:17-34
    std::vector<char> visited((size_t)n, 0);
    std::vector<int> stack;
    int label = -1;
    for (int64_t i = 0; i < n; ++i) labels_out[i] = -1;
    for (int64_t ind = n - 1; ind >= 0; --ind) {           // unvisited.pop()
        if (visited[(size_t)ind]) continue;
        visited[(size_t)ind] = 1;
        if (counts[(size_t)ind] < min_pts) continue;       // noise (for good: it left `unvisited`)
        ++label;
        labels_out[ind] = label;
        stack.assign(nbr.begin() + offs[(size_t)ind], nbr.begin() + offs[(size_t)ind + 1]);
        while (!stack.empty()) {
            const int cur = stack.back();
            stack.pop_back();
            if (visited[(size_t)cur]) continue;            // `if cur_ind in unvisited`
            visited[(size_t)cur] = 1;
            labels_out[cur] = label;
            if (counts[(size_t)cur] > min_pts)             // strict, unlike the seed test
                stack.insert(stack.end(), nbr.begin() + offs[(size_t)cur], nbr.begin() + offs[(size_t)cur + 1]);
        }
    }
    if (n_clusters_out) *n_clusters_out = label + 1;
    return PCR_OK;
}

}  // extern "C"
