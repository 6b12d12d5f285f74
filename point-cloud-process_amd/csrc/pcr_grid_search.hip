// Exact 1-NN search over the multi-level voxel-hash grid (see pcr_grid.hip for the layout).
//
// Queries are processed in TILES of 64 consecutive records of the Morton-sorted query cloud
// (a rigid transform keeps a tile spatially compact, so the cloud is sorted once per pair).
//
//   tile    one 256-thread block per tile: bounding box of the tile's cells (+1 ring) at the
//           finest level whose box has <= 512 cells; every thread looks up cells of the box in
//           the hash table (2-3 lookups per query instead of 27); the points of the occupied
//           cells are staged ONCE into LDS with coalesced reads; then every query is compared
//           with every staged point (LDS broadcast reads, no global traffic, no divergence).
//           A query is resolved when its bound ball (best distance so far, or the gate) lies
//           inside the staged box.  Measured alternatives, all exact, all slower on the 120k
//           KITTI-shaped pair: 8 lanes/query over the 27 cells with lane-owned cells (117 us),
//           flattened directory (174 us), prune-then-visit (88-180 us) -- dependent
//           lookup->scan chains and uncoalesced 32-B reads dominate there.
//   ring 1  (per query, 8 lanes) only for tiles whose box is too large to stage.
//   ring 2  queries whose bound ball fits in their own 5x5x5 block: the 98 shell cells.
//   hard    everything else, one wave per query: pruned top-down descent of the nested cell
//           hierarchy (cells are contiguous runs of the Morton-sorted target at every level).
// Work lists between stages are per-block segments (no global append counter: a single
// returning atomic word saturates at ~88 appends/us on this chip).
// Pruning only ever skips a cell whose box distance exceeds a bound that is itself >= the
// final answer, so every stage returns the exact nearest neighbour (lowest index on ties).
#include <cmath>
#include <cstdlib>
#include "pcr_grid_dev.h"

constexpr int TQ = 64;                      // queries per tile (= slots per block in the work lists)
constexpr int T_MAXC = 1024;                // cells in a tile box
constexpr int T_CPT = T_MAXC / 256;         // cells looked up per thread
constexpr int T_PMAX = 512;                 // points staged per round
constexpr unsigned int T_PCAP = 4 * T_PMAX;   // tiles with more candidate points than this go per-query
constexpr int SG = 8;                       // lanes per query in rings 1 and 2
constexpr unsigned int HARD_SCAN_T = 192;   // the hard stage scans cells up to this size, descends into bigger ones
constexpr int HARD_STACK = 160;
constexpr long long ID_NONE = 0x7fffffffffffffffll;

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

__device__ static inline void scan_range(const pcr_pt* __restrict__ pts, unsigned int s, unsigned int e, unsigned int step, double ax,
                                         double ay, double az, double& bd2, long long& bid, unsigned int& bpos) {
    for (unsigned int j = s; j < e; j += step) {
        const pcr_pt b = pts[j];
        const double d2 = dist2(ax, ay, az, b);
        if (better(d2, b.id, bd2, bid)) { bd2 = d2; bid = b.id; bpos = j; }
    }
}

__device__ static inline double vmin(double a, double b) {  // plain v_min_f64 (fmin() adds two canonicalising v_max)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ static inline double sq_pos(double v) {
    v = fmax(v, 0.0);
    return v * v;
}

// per-block work lists: list[block * TQ + k], k < counts[block * 3 + which]
enum { L_RING1 = 0, L_RING2 = 1, L_HARD = 2 };

__device__ static inline void push_item(work_item* __restrict__ list, unsigned int* s_count, double ax, double ay, double az,
                                        double bd2, unsigned int bpos, unsigned int qi) {
    const unsigned int w = blockIdx.x * TQ + atomicAdd(s_count, 1u);  // LDS atomic
    work_item it;
    it.ax = ax; it.ay = ay; it.az = az;
    it.best_d2 = bd2;
    it.best_pos = bpos;
    it.qi = qi;
    list[w] = it;
}

// -------------------------------------------------------------------- tile
struct tile_smem {
    double px[T_PMAX + 4], py[T_PMAX + 4], pz[T_PMAX + 4];  // staged candidates (SoA: broadcast reads), padded to a multiple of 4
    int pid[T_PMAX];
    unsigned int ppos[T_PMAX];
    unsigned int c_start[T_MAXC];
    unsigned int c_off[T_MAXC + 1];  // exclusive prefix of the cell counts
    double m_d2[4][TQ];
    int m_id[4][TQ];
    unsigned int m_pos[4][TQ];
    unsigned int scan_tmp[256];
    int box_lo[3], dims[3];
    int level, ncell;
    unsigned int total;
    unsigned int counts[3];
};

__global__ void __launch_bounds__(256)
grid_tile_kernel(pcr_grid_view gv, pcr_pt* __restrict__ q, long long nq, pcr_xform x, int has_x, int write_back, double max_d2,
                 int gated, unsigned int* __restrict__ res_pos, double* __restrict__ res_d2, work_item* __restrict__ list1,
                 work_item* __restrict__ list_a, work_item* __restrict__ list_b, unsigned int* __restrict__ counts /* [block][3] */,
                 unsigned long long* __restrict__ dbg) {
    __shared__ tile_smem sm;
    const unsigned long long t_start = dbg ? __builtin_amdgcn_s_memtime() : 0;
    unsigned long long t_ph[5] = {0, 0, 0, 0, 0};
#define PH_STAMP(i) do { if (dbg) t_ph[i] = __builtin_amdgcn_s_memtime() - t_start; } while (0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long qi = (long long)blockIdx.x * TQ + lane;  // every wave holds the same 64 queries
    const bool qvalid = qi < nq;
    if (tid < 3) sm.counts[tid] = 0;
    // ---- load + transform the tile's queries (wave 0 writes back)
    double ax = 0, ay = 0, az = 0;
    bool clamped = false;
    int cx = 0, cy = 0, cz = 0;
    if (qvalid) {
        pcr_pt p = q[qi];
        ax = p.x; ay = p.y; az = p.z;
        if (has_x) {
            xform_apply(x, p, &ax, &ay, &az);
            if (write_back && wave == 0) {
                p.x = ax; p.y = ay; p.z = az;
                q[qi] = p;
            }
        }
        cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
    }
    const bool in_box = qvalid && !clamped;
    // ---- bounding box of the tile's level-0 cells (wave 0), then the level whose box fits T_MAXC cells
    if (wave == 0) {
        int mn[3] = {in_box ? cx : 0x7fffffff, in_box ? cy : 0x7fffffff, in_box ? cz : 0x7fffffff};
        int mx[3] = {in_box ? cx : -1, in_box ? cy : -1, in_box ? cz : -1};
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                mn[k] = min(mn[k], __shfl_xor(mn[k], off, 64));
                mx[k] = max(mx[k], __shfl_xor(mx[k], off, 64));
            }
        }
        if (lane == 0) {
            int level = -1;
            if (mx[0] >= 0) {
                for (int l = 0; l < gv.levels; ++l) {
                    const long long d0 = (mx[0] >> (2 * l)) - (mn[0] >> (2 * l)) + 3, d1 = (mx[1] >> (2 * l)) - (mn[1] >> (2 * l)) + 3,
                                    d2 = (mx[2] >> (2 * l)) - (mn[2] >> (2 * l)) + 3;
                    if (d0 * d1 * d2 <= T_MAXC) {
                        level = l;
                        sm.box_lo[0] = (mn[0] >> (2 * l)) - 1; sm.box_lo[1] = (mn[1] >> (2 * l)) - 1; sm.box_lo[2] = (mn[2] >> (2 * l)) - 1;
                        sm.dims[0] = (int)d0; sm.dims[1] = (int)d1; sm.dims[2] = (int)d2;
                        sm.ncell = (int)(d0 * d1 * d2);
                        break;
                    }
                }
            }
            sm.level = level;
        }
    }
    __syncthreads();
    PH_STAMP(0);
    const int level = sm.level;
    double bd2 = DBL_MAX;
    long long bid = ID_NONE;
    unsigned int bpos = POS_NONE;
    bool staged = false;
    if (level >= 0) {
        // ---- cell directory: every thread looks up cells of the box
        const int ncell = sm.ncell;
        const int d0 = sm.dims[0], d1 = sm.dims[1];
        const int lim = (int)(PCR_COORD_MAX >> (2 * level));
        // thread t looks up cells t, t+256, ... (independent 64-B bucket reads)
        unsigned int my_cnt[T_CPT];
#pragma unroll
        for (int r = 0; r < T_CPT; ++r) {
            const int c = tid + r * 256;
            my_cnt[r] = 0;
            if (c < ncell) {
                const int ix = c % d0, iy = (c / d0) % d1, iz = c / (d0 * d1);
                const int X = sm.box_lo[0] + ix, Y = sm.box_lo[1] + iy, Z = sm.box_lo[2] + iz;
                unsigned int s = 0, e = 0;
                if (X >= 0 && Y >= 0 && Z >= 0 && X <= lim && Y <= lim && Z <= lim)
                    lookup_cell(gv.table[level], gv.mask[level], (unsigned int)X, (unsigned int)Y, (unsigned int)Z, &s, &e);
                sm.c_start[c] = s;
                my_cnt[r] = e - s;
            }
        }
        PH_STAMP(1);
        // block exclusive scan in cell order (c = r*256 + tid): per-r wave scans + one barrier
        {
            unsigned int inc[T_CPT];
#pragma unroll
            for (int r = 0; r < T_CPT; ++r) inc[r] = my_cnt[r];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
                for (int r = 0; r < T_CPT; ++r) {
                    const unsigned int o = __shfl_up(inc[r], off, 64);
                    if (lane >= off) inc[r] += o;
                }
            }
            if (lane == 63) {
#pragma unroll
                for (int r = 0; r < T_CPT; ++r) sm.scan_tmp[r * 4 + wave] = inc[r];
            }
            __syncthreads();
            unsigned int run = 0;  // sum of all cells before (r, wave 0)
#pragma unroll
            for (int r = 0; r < T_CPT; ++r) {
                unsigned int base = run;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const unsigned int t = sm.scan_tmp[r * 4 + w];
                    if (w < wave) base += t;
                    run += t;
                }
                const int c = tid + r * 256;
                if (c < ncell) sm.c_off[c] = base + inc[r] - my_cnt[r];
            }
            if (tid == 0) { sm.c_off[ncell] = run; sm.total = run; }
            __syncthreads();
        }
        PH_STAMP(2);
        const unsigned int total = sm.total;
        if (total <= T_PCAP) {
            staged = true;
            // ---- rounds: stage up to T_PMAX points into LDS, compare every query with every staged point
            for (unsigned int base = 0; base < total; base += T_PMAX) {
                const unsigned int wend = min(total, base + T_PMAX);
                const unsigned int cnt = wend - base;
                // staging: thread t copies staged points t, t+256, ...; the owning cell is found by binary
                // search in the prefix array, so all global reads of a round are independent
                for (unsigned int f = base + tid; f < wend; f += 256) {
                    int lo = 0, hi = ncell - 1;
                    while (lo < hi) {
                        const int mid = (lo + hi + 1) >> 1;
                        if (sm.c_off[mid] <= f) lo = mid;
                        else hi = mid - 1;
                    }
                    const unsigned int j = sm.c_start[lo] + (f - sm.c_off[lo]);
                    const pcr_pt b = gv.pts[j];
                    const unsigned int k = f - base;
                    sm.px[k] = b.x; sm.py[k] = b.y; sm.pz[k] = b.z;
                    sm.pid[k] = (int)b.id;
                    sm.ppos[k] = j;
                }
                PH_STAMP(3);
                if (tid < 4) { sm.px[cnt + tid] = 1e300; sm.py[cnt + tid] = 0.0; sm.pz[cnt + tid] = 0.0; }  // padding of the last group of 4
                __syncthreads();
                // evaluation: wave w takes groups of 4 staged points; lane = query.  Branch-free: strict <
                // keeps the first of equal distances; exact ties (duplicate targets) are flagged and resolved
                // to the lowest original index in the rare fix-up below.
                double rd2 = DBL_MAX;  // best of this round
                int rk = -1;
                bool tie = false;
                for (unsigned int k0 = wave * 4; k0 < cnt; k0 += 16) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const double dx = ax - sm.px[k0 + u], dy = ay - sm.py[k0 + u], dz = az - sm.pz[k0 + u];
                        const double d = (dx * dx + dy * dy) + dz * dz;
                        const bool lt = d < rd2;
                        tie = lt ? false : (tie | (d == rd2));
                        rd2 = lt ? d : rd2;
                        rk = lt ? (int)(k0 + u) : rk;
                    }
                }
                if (rk >= 0) {
                    long long rid = sm.pid[rk];
                    if (tie) {  // some other staged point of this wave's share is exactly as far: lowest id wins
                        for (unsigned int k = wave * 4; k < cnt; k += 16)
                            for (int u = 0; u < 4 && k + u < cnt; ++u) {
                                const double dx = ax - sm.px[k + u], dy = ay - sm.py[k + u], dz = az - sm.pz[k + u];
                                if ((dx * dx + dy * dy) + dz * dz == rd2 && sm.pid[k + u] < rid) { rid = sm.pid[k + u]; rk = (int)(k + u); }
                            }
                    }
                    if (better(rd2, rid, bd2, bid)) { bd2 = rd2; bid = rid; bpos = sm.ppos[rk]; }
                }
                __syncthreads();
            }
        }
    }
    PH_STAMP(4);
    // ---- merge the four waves' results per query
    sm.m_d2[wave][lane] = bd2;
    sm.m_id[wave][lane] = (bid == ID_NONE) ? 0x7fffffff : (int)bid;
    sm.m_pos[wave][lane] = bpos;
    __syncthreads();
    if (wave == 0 && qvalid) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const double od2 = sm.m_d2[w][lane];
            const long long oid = sm.m_id[w][lane];
            if (sm.m_pos[w][lane] != POS_NONE && better(od2, oid, bd2, bid)) { bd2 = od2; bid = oid; bpos = sm.m_pos[w][lane]; }
        }
        if (clamped) {
            push_item(list_b, &sm.counts[L_HARD], ax, ay, az, DBL_MAX, POS_NONE, (unsigned int)qi);
        } else if (!staged) {
            push_item(list_b, &sm.counts[L_HARD], ax, ay, az, DBL_MAX, POS_NONE, (unsigned int)qi);
        } else {
            // distance from the query to the boundary of the staged box
            const double cell = gv.cell0 * (double)(1ll << (2 * level));
            const int bl = (int)(PCR_COORD_BIAS >> (2 * level));
            double db = DBL_MAX;
            const double a[3] = {ax, ay, az};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double blo = gv.lo[k] + (double)(sm.box_lo[k] - bl) * cell;
                const double bhi = gv.lo[k] + (double)(sm.box_lo[k] + sm.dims[k] - bl) * cell;
                db = fmin(db, fmin(a[k] - blo, bhi - a[k]));
            }
            db = fmax(db - cell * 1e-9, 0.0);
            const double bound2 = gated ? fmin(bd2, max_d2) : bd2;
            const double safe0 = gv.cell0 * (1.0 - 1e-9);
            if (bound2 <= db * db) {  // the bound ball lies inside the staged box: exact
                res_pos[qi] = bpos;
                if (res_d2) res_d2[qi] = bd2;
            } else if (bound2 <= 4.0 * safe0 * safe0) {
                push_item(list_b, &sm.counts[L_HARD], ax, ay, az, bd2, bpos, (unsigned int)qi);
            } else {
                push_item(list_b, &sm.counts[L_HARD], ax, ay, az, bd2, bpos, (unsigned int)qi);
            }
        }
    }
    __syncthreads();
    if (tid < 3) counts[blockIdx.x * 3 + tid] = sm.counts[tid];
    if (dbg && tid == 0) {
        dbg[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memtime() - t_start;
        for (int i = 0; i < 5; ++i) dbg[(1 << 16) + blockIdx.x * 8 + i] = t_ph[i];
        dbg[blockIdx.x * 4 + 1] = (level >= 0) ? sm.total : 0xffffffffu;
        dbg[blockIdx.x * 4 + 2] = ((unsigned long long)(level + 1) << 32) | (unsigned)sm.ncell;
        dbg[blockIdx.x * 4 + 3] = ((unsigned long long)sm.counts[0] << 40) | ((unsigned long long)sm.counts[1] << 20) | sm.counts[2];
    }
}

// ------------------------------------------------------------- ring 1 / 2
// Per-axis squared distance from the query to the slab of cells at offset d (|d| <= 2),
// shrunk by a rounding slack so that pruning stays conservative.
struct axis_d2 {
    double m2, m1, p1, p2;  // d = -2, -1, +1, +2   (d = 0 -> 0)
    __device__ inline double at(int d) const { return d == 0 ? 0.0 : d == -1 ? m1 : d == 1 ? p1 : d == -2 ? m2 : p2; }
};

__device__ static inline axis_d2 make_axis(double f, double cell) {
    const double slack = cell * 1e-9;
    axis_d2 a;
    a.m1 = sq_pos(f - slack);
    a.p1 = sq_pos(cell - f - slack);
    a.m2 = sq_pos(f + cell - slack);
    a.p2 = sq_pos(2.0 * cell - f - slack);
    return a;
}

// Ring 1 for the queries of tiles that could not be staged: own cell by the whole group, then
// lane-owned neighbour cells pruned against the bound.  Appends to the block's ring-2 / hard lists.
__global__ void __launch_bounds__(256)
grid_ring1_list_kernel(pcr_grid_view gv, const work_item* __restrict__ list1, double max_d2, int gated, unsigned int* __restrict__ res_pos,
                       double* __restrict__ res_d2, work_item* __restrict__ list_a, work_item* __restrict__ list_b,
                       unsigned int* __restrict__ counts) {
    constexpr int G = SG;
    __shared__ unsigned int s_counts[3];
    if (threadIdx.x < 3) s_counts[threadIdx.x] = counts[blockIdx.x * 3 + threadIdx.x];
    __syncthreads();
    const unsigned int count = s_counts[L_RING1];
    __syncthreads();
    const int gl = threadIdx.x % G;
    const pcr_cell_slot* __restrict__ tab = gv.table[0];
    const unsigned int mask = gv.mask[0];
    const double cell = gv.cell0;
    for (unsigned int g = threadIdx.x / G; g < count; g += 256 / G) {
        const work_item it = list1[(size_t)blockIdx.x * TQ + g];
        const double ax = it.ax, ay = it.ay, az = it.az;
        bool clamped = false;
        const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
        double bd2 = DBL_MAX;
        long long bid = ID_NONE;
        unsigned int bpos = POS_NONE;
        {
            unsigned int s, e;
            if (lookup_cell(tab, mask, (unsigned int)cx, (unsigned int)cy, (unsigned int)cz, &s, &e))
                scan_range(gv.pts, s + gl, e, G, ax, ay, az, bd2, bid, bpos);
            group_best<G>(bd2, bid, bpos);
        }
        double bound2 = gated ? fmin(bd2, max_d2) : bd2;
        const axis_d2 dx2 = make_axis((ax - gv.lo[0]) - (double)(cx - (int)PCR_COORD_BIAS) * cell, cell);
        const axis_d2 dy2 = make_axis((ay - gv.lo[1]) - (double)(cy - (int)PCR_COORD_BIAS) * cell, cell);
        const axis_d2 dz2 = make_axis((az - gv.lo[2]) - (double)(cz - (int)PCR_COORD_BIAS) * cell, cell);
#pragma unroll
        for (int i = 0; i < (27 + G - 1) / G; ++i) {
            const int n = gl + i * G;
            if (n < 27 && n != 13) {
                const int dx = n % 3 - 1, dy = (n / 3) % 3 - 1, dz = n / 9 - 1;
                const unsigned int nx = (unsigned int)(cx + dx), ny = (unsigned int)(cy + dy), nz = (unsigned int)(cz + dz);
                if ((dx2.at(dx) + dy2.at(dy)) + dz2.at(dz) <= bound2 && nx <= (unsigned int)PCR_COORD_MAX &&
                    ny <= (unsigned int)PCR_COORD_MAX && nz <= (unsigned int)PCR_COORD_MAX) {
                    unsigned int s, e;
                    if (lookup_cell(tab, mask, nx, ny, nz, &s, &e)) scan_range(gv.pts, s, e, 1, ax, ay, az, bd2, bid, bpos);
                }
            }
        }
        group_best<G>(bd2, bid, bpos);
        if (gl == 0) {
            bound2 = gated ? fmin(bd2, max_d2) : bd2;
            const double safe = cell * (1.0 - 1e-9);
            if (bound2 <= safe * safe) {
                res_pos[it.qi] = bpos;
                if (res_d2) res_d2[it.qi] = bd2;
            } else if (bound2 <= 4.0 * safe * safe) {
                push_item(list_a, &s_counts[L_RING2], ax, ay, az, bd2, bpos, it.qi);
            } else {
                push_item(list_b, &s_counts[L_HARD], ax, ay, az, bd2, bpos, it.qi);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) counts[blockIdx.x * 3 + threadIdx.x] = s_counts[threadIdx.x];
}

// Exclusive prefix of the per-block ring-2 and hard counts, so that the (few, unevenly spread)
// items can be dealt evenly to the waves of the next two kernels: offs[which][b], which = 0 ring 2, 1 hard.
__global__ void __launch_bounds__(1024)
grid_prefix_kernel(const unsigned int* __restrict__ counts, int nblocks, unsigned int* __restrict__ offs) {
    __shared__ unsigned int s_w[2][16];
    __shared__ unsigned int s_carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 2) s_carry[tid] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int b = b0 + tid;
        const unsigned int c0 = b < nblocks ? counts[b * 3 + L_RING2] : 0, c1 = b < nblocks ? counts[b * 3 + L_HARD] : 0;
        unsigned int i0 = c0, i1 = c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o0 = __shfl_up(i0, off, 64), o1 = __shfl_up(i1, off, 64);
            if (lane >= off) { i0 += o0; i1 += o1; }
        }
        if (lane == 63) { s_w[0][wave] = i0; s_w[1][wave] = i1; }
        __syncthreads();
        unsigned int base0 = s_carry[0], base1 = s_carry[1], tot0 = 0, tot1 = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) { base0 += s_w[0][w]; base1 += s_w[1][w]; }
            tot0 += s_w[0][w]; tot1 += s_w[1][w];
        }
        if (b < nblocks) { offs[b] = base0 + i0 - c0; offs[(nblocks + 1) + b] = base1 + i1 - c1; }
        __syncthreads();
        if (tid == 0) { s_carry[0] += tot0; s_carry[1] += tot1; }
        __syncthreads();
    }
    if (tid == 0) { offs[nblocks] = s_carry[0]; offs[(nblocks + 1) + nblocks] = s_carry[1]; }
}

// compact index i -> (block, slot): largest b with offs[b] <= i
__device__ static inline work_item locate_item(const work_item* __restrict__ list, const unsigned int* __restrict__ offs, int nblocks,
                                               unsigned int i) {
    int lo = 0, hi = nblocks - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (offs[mid] <= i) lo = mid;
        else hi = mid - 1;
    }
    return list[(size_t)lo * TQ + (i - offs[lo])];
}

// Ring 2: the 98 shell cells of the query's 5x5x5 block, lane-owned, pruned against the bound.
__global__ void __launch_bounds__(256)
grid_ring2_kernel(pcr_grid_view gv, const work_item* __restrict__ list, const unsigned int* __restrict__ offs, int nblocks, double max_d2,
                  int gated, unsigned int* __restrict__ res_pos, double* __restrict__ res_d2) {
    constexpr int G = SG;
    const int gl = threadIdx.x % G;
    const unsigned int count = offs[nblocks];
    const pcr_cell_slot* __restrict__ tab = gv.table[0];
    const unsigned int mask = gv.mask[0];
    const double cell = gv.cell0;
    for (unsigned int g = blockIdx.x * (256 / G) + threadIdx.x / G; g < count; g += gridDim.x * (256 / G)) {
        const work_item it = locate_item(list, offs, nblocks, g);
        const double ax = it.ax, ay = it.ay, az = it.az;
        double bd2 = DBL_MAX;
        long long bid = ID_NONE;
        unsigned int bpos = POS_NONE;
        if (gl == 0 && it.best_pos != POS_NONE) {
            bd2 = it.best_d2;
            bpos = it.best_pos;
            bid = gv.pts[bpos].id;
        }
        const double bound2 = gated ? fmin(it.best_d2, max_d2) : it.best_d2;
        bool clamped = false;
        const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
        const axis_d2 dx2 = make_axis((ax - gv.lo[0]) - (double)(cx - (int)PCR_COORD_BIAS) * cell, cell);
        const axis_d2 dy2 = make_axis((ay - gv.lo[1]) - (double)(cy - (int)PCR_COORD_BIAS) * cell, cell);
        const axis_d2 dz2 = make_axis((az - gv.lo[2]) - (double)(cz - (int)PCR_COORD_BIAS) * cell, cell);
#pragma unroll 4
        for (int i = 0; i < (125 + G - 1) / G; ++i) {
            const int c = gl + i * G;
            const int ix = c % 5, iy = (c / 5) % 5, iz = c / 25;
            const bool shell = c < 125 && (ix == 0 || ix == 4 || iy == 0 || iy == 4 || iz == 0 || iz == 4);
            if (shell && (dx2.at(ix - 2) + dy2.at(iy - 2)) + dz2.at(iz - 2) <= bound2) {
                const unsigned int nx = (unsigned int)(cx + ix - 2), ny = (unsigned int)(cy + iy - 2), nz = (unsigned int)(cz + iz - 2);
                if (nx <= (unsigned int)PCR_COORD_MAX && ny <= (unsigned int)PCR_COORD_MAX && nz <= (unsigned int)PCR_COORD_MAX) {
                    unsigned int s, e;
                    if (lookup_cell(tab, mask, nx, ny, nz, &s, &e)) scan_range(gv.pts, s, e, 1, ax, ay, az, bd2, bid, bpos);
                }
            }
        }
        group_best<G>(bd2, bid, bpos);
        if (gl == 0) {
            res_pos[it.qi] = bpos;
            if (res_d2) res_d2[it.qi] = bd2;
        }
    }
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

__device__ static inline double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ void __launch_bounds__(256)
grid_hard_kernel(pcr_grid_view gv, const work_item* __restrict__ list, const unsigned int* __restrict__ offs, int nblocks, double max_d2,
                 int gated, unsigned int* __restrict__ res_pos, double* __restrict__ res_d2) {
    __shared__ hard_entry s_stack[4][HARD_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    hard_entry* stack = s_stack[wave];
    const unsigned int count = offs[nblocks];
    const int top = gv.levels - 1;
    for (unsigned int w = blockIdx.x * 4 + wave; w < count; w += gridDim.x * 4) {
        const work_item it = locate_item(list, offs, nblocks, w);
        const double ax = it.ax, ay = it.ay, az = it.az;
        double bd2 = DBL_MAX;
        long long bid = ID_NONE;
        unsigned int bpos = POS_NONE;
        if (lane == 0 && it.best_pos != POS_NONE) {
            bd2 = it.best_d2;
            bpos = it.best_pos;
            bid = gv.pts[bpos].id;
        }
        double bound2 = gated ? fmin(it.best_d2, max_d2) : it.best_d2;  // DBL_MAX when nothing bounds the search
        bool clamped = false;
        const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
        if (!clamped && it.best_pos == POS_NONE) {
            // nothing known yet (tile too large to stage): the query's own level-0 cell gives a first bound
            unsigned int s, e;
            if (lookup_cell(gv.table[0], gv.mask[0], (unsigned int)cx, (unsigned int)cy, (unsigned int)cz, &s, &e)) {
                scan_range(gv.pts, s + lane, e, 64, ax, ay, az, bd2, bid, bpos);
                bound2 = fmin(bound2, wave_min(bd2));
            }
        }
        // start level: the smallest one whose 3x3x3 block covers the bound ball
        int s_level = -1;
        if (!clamped && bound2 < DBL_MAX) {
            double c = gv.cell0;
            for (int l = 0; l <= top; ++l) {
                const double safe = c * (1.0 - 1e-9);
                if (safe * safe >= bound2) { s_level = l; break; }
                c *= 4.0;
            }
        }
        int sp = 0;  // wave-uniform stack pointer
        {
            // initial cells: the query's 3x3x3 block at s_level, or the <= 8 root cells that hold the whole target
            const bool roots = (s_level < 0);
            const int lvl = roots ? top : s_level;
            const double cell = gv.cell0 * (double)(1ll << (2 * lvl));
            const int b0 = (int)(PCR_COORD_BIAS >> (2 * lvl));
            int X, Y, Z;
            bool valid;
            if (roots) {
                valid = lane < 8;
                X = b0 + (lane & 1); Y = b0 + ((lane >> 1) & 1); Z = b0 + ((lane >> 2) & 1);
            } else {
                valid = lane < 27;
                X = (cx >> (2 * lvl)) + (lane % 3 - 1);
                Y = (cy >> (2 * lvl)) + ((lane / 3) % 3 - 1);
                Z = (cz >> (2 * lvl)) + (lane / 9 - 1);
                const int lim = (int)(PCR_COORD_MAX >> (2 * lvl));
                valid = valid && X >= 0 && Y >= 0 && Z >= 0 && X <= lim && Y <= lim && Z <= lim;
            }
            unsigned int s = 0, e = 0;
            double bdist = 0.0;
            if (valid) {
                bdist = box_dist2(gv, lvl, cell, (unsigned int)X, (unsigned int)Y, (unsigned int)Z, ax, ay, az);
                valid = bdist <= bound2 && lookup_cell(gv.table[lvl], gv.mask[lvl], (unsigned int)X, (unsigned int)Y, (unsigned int)Z, &s, &e);
            }
            // far cells first, the cell containing the query last (popped first)
            const unsigned long long m_far = __ballot(valid && bdist > 0.0);
            const unsigned long long m_near = __ballot(valid && !(bdist > 0.0));
            const unsigned long long below = (1ull << lane) - 1ull;
            int slot = -1;
            if (valid && bdist > 0.0) slot = __popcll(m_far & below);
            else if (valid) slot = __popcll(m_far) + __popcll(m_near & below);
            if (slot >= 0 && slot < HARD_STACK) {
                hard_entry en;
                en.start = s; en.end = e; en.x = (unsigned int)X; en.y = (unsigned int)Y; en.z = (unsigned int)Z; en.level = lvl;
                stack[slot] = en;
            }
            sp = __popcll(m_far) + __popcll(m_near);
        }
        while (sp > 0) {
            --sp;
            const hard_entry en = stack[sp];  // same address in every lane: LDS broadcast
            const double cell = gv.cell0 * (double)(1ll << (2 * en.level));
            if (box_dist2(gv, en.level, cell, en.x, en.y, en.z, ax, ay, az) > bound2) continue;
            const unsigned int cnt = en.end - en.start;
            const bool room = sp + 64 <= HARD_STACK;
            if (en.level == 0 || cnt <= HARD_SCAN_T || !room) {
                scan_range(gv.pts, en.start + lane, en.end, 64, ax, ay, az, bd2, bid, bpos);
                bound2 = fmin(bound2, wave_min(bd2));
            } else {
                // one child per lane: box test against the bound, probe, push the survivors
                const int cl = en.level - 1;
                const unsigned int X = en.x * 4u + (lane & 3), Y = en.y * 4u + ((lane >> 2) & 3), Z = en.z * 4u + (lane >> 4);
                const double bdist = box_dist2(gv, cl, cell * 0.25, X, Y, Z, ax, ay, az);
                unsigned int s = 0, e = 0;
                const bool valid = bdist <= bound2 && lookup_cell(gv.table[cl], gv.mask[cl], X, Y, Z, &s, &e);
                const unsigned long long m_far = __ballot(valid && bdist > 0.0);
                const unsigned long long m_near = __ballot(valid && !(bdist > 0.0));
                const unsigned long long below = (1ull << lane) - 1ull;
                int slot = -1;
                if (valid && bdist > 0.0) slot = __popcll(m_far & below);
                else if (valid) slot = __popcll(m_far) + __popcll(m_near & below);
                if (slot >= 0) {
                    hard_entry ch;
                    ch.start = s; ch.end = e; ch.x = X; ch.y = Y; ch.z = Z; ch.level = cl;
                    stack[sp + slot] = ch;
                }
                sp += __popcll(m_far) + __popcll(m_near);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double od2 = __shfl_xor(bd2, off, 64);
            const long long oid = __shfl_xor(bid, off, 64);
            const unsigned int opos = __shfl_xor(bpos, off, 64);
            if (better(od2, oid, bd2, bid)) { bd2 = od2; bid = oid; bpos = opos; }
        }
        if (lane == 0) {
            res_pos[it.qi] = bpos;
            if (res_d2) res_d2[it.qi] = bd2;
        }
    }
}

// --------------------------------------------------------------- epilogues
// nn1: sorted position -> original target index, gate, scatter to the query's original slot
__global__ void grid_finalize_nn1_kernel(pcr_grid_view gv, const pcr_pt* __restrict__ q, long long nq, const unsigned int* __restrict__ res_pos,
                                         const double* __restrict__ res_d2, double max_d2, int gated, int* __restrict__ idx_out,
                                         double* __restrict__ d2_out) {
    const long long qi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
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
                       const unsigned int* __restrict__ res_pos, double max_d2, int gated, double* __restrict__ partials) {
    __shared__ double s_part[4][PCR_NMOM];
    double m[PCR_NMOM];
#pragma unroll
    for (int k = 0; k < PCR_NMOM; ++k) m[k] = 0.0;
    for (long long qi = (long long)blockIdx.x * blockDim.x + threadIdx.x; qi < nq; qi += (long long)gridDim.x * blockDim.x) {
        const unsigned int pos = res_pos[qi];
        if (pos == POS_NONE) continue;
        const pcr_pt p = q[qi];
        double ax = p.x, ay = p.y, az = p.z;
        if (apply_x) xform_apply(x, p, &ax, &ay, &az);
        const pcr_pt b = gv.pts[pos];
        const double d2 = dist2(ax, ay, az, b);
        if (gated && !(d2 < max_d2)) continue;
        const double a0 = ax - gv.origin[0], a1 = ay - gv.origin[1], a2 = az - gv.origin[2];
        const double b0 = b.x - gv.origin[0], b1 = b.y - gv.origin[1], b2 = b.z - gv.origin[2];
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
#pragma unroll
    for (int k = 0; k < PCR_NMOM - 1; ++k) {
        double v = m[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        m[k] = v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < PCR_NMOM; ++k) s_part[wave][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < PCR_NMOM) {
        const double v = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
        partials[(long long)blockIdx.x * PCR_NMOM + threadIdx.x] = v;
    }
}

// out[k] = sum_b partials[b][k], fixed association order: 32 strided slices then a tree.
__global__ void __launch_bounds__(1024) reduce_partials_kernel(const double* __restrict__ partials, int nblocks,
                                                               double* __restrict__ out) {
    __shared__ double s[32][32];
    const int k = threadIdx.x & 31, slice = threadIdx.x >> 5;
    double v = 0.0;
    if (k < PCR_NMOM) {
        for (int b = slice; b < nblocks; b += 32) v += partials[(long long)b * PCR_NMOM + k];
    }
    s[slice][k] = v;
    __syncthreads();
    for (int st = 16; st > 0; st >>= 1) {
        if (slice < st) s[slice][k] += s[slice + st][k];
        __syncthreads();
    }
    if (slice == 0 && k < PCR_NMOM) out[k] = s[0][k];
}

// ------------------------------------------------------------------- host
struct grid_scratch {
    unsigned int* res_pos = nullptr;
    double* res_d2 = nullptr;
    work_item* lists = nullptr;      // [3][nblocks][TQ]
    unsigned int* counts = nullptr;  // [nblocks][3]
    unsigned int* offs = nullptr;    // [2][nblocks + 1]
    int64_t nq = 0;
    int nblocks = 0;
};

// Runs the search stages; leaves res_pos (and res_d2 when asked) on the device.
static int grid_search_launch(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_xform* x, int write_back, double max_d2,
                              bool gated, bool want_d2, grid_scratch* sc) {
    int rc;
    // tiles are runs of the Morton-sorted query cloud (sorted once; rigid motion keeps them compact)
    if ((rc = pcr_cloud_morton_sort(ctx, qc, idx->cell))) return rc;
    const int64_t nq = qc->n;
    pcr_pt* q = qc->d;
    sc->nq = nq;
    const int nblocks = (int)((nq + TQ - 1) / TQ);
    sc->nblocks = nblocks;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * nq, (void**)&sc->res_pos))) return rc;
    if (want_d2 && (rc = pcr_dev_alloc(ctx, sizeof(double) * nq, (void**)&sc->res_d2))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(work_item) * 3 * TQ * (size_t)nblocks, (void**)&sc->lists))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * 3 * (size_t)nblocks, (void**)&sc->counts))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * 2 * (size_t)(nblocks + 1), (void**)&sc->offs))) return rc;
    work_item* l1 = sc->lists;
    work_item* la = sc->lists + (size_t)TQ * nblocks;
    work_item* lb = sc->lists + 2 * (size_t)TQ * nblocks;
    pcr_xform xi;
    pcr_xform_from_T(nullptr, &xi);
    pcr_prof_mark(ctx, 0);
    hipLaunchKernelGGL(grid_tile_kernel, dim3(nblocks), dim3(256), 0, ctx->stream, idx->view, q, (long long)nq, x ? *x : xi, x ? 1 : 0,
                       write_back, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2, l1, la, lb, sc->counts, ctx->d_debug);
    pcr_prof_mark(ctx, 1);
    hipLaunchKernelGGL(grid_prefix_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const unsigned int*)sc->counts, nblocks, sc->offs);
    // fixed grids walk the compacted lists (their lengths are only known on the device)
    const int g3 = nblocks < 8 * ctx->cu_count ? nblocks : 8 * ctx->cu_count;
    hipLaunchKernelGGL(grid_hard_kernel, dim3(g3), dim3(256), 0, ctx->stream, idx->view, (const work_item*)lb,
                       (const unsigned int*)(sc->offs + nblocks + 1), nblocks, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

static void grid_scratch_free(pcr_ctx* ctx, grid_scratch* sc) {
    pcr_dev_free(ctx, sc->res_pos, sizeof(unsigned int) * sc->nq);
    if (sc->res_d2) pcr_dev_free(ctx, sc->res_d2, sizeof(double) * sc->nq);
    pcr_dev_free(ctx, sc->lists, sizeof(work_item) * 3 * TQ * (size_t)sc->nblocks);
    pcr_dev_free(ctx, sc->counts, sizeof(unsigned int) * 3 * (size_t)sc->nblocks);
    pcr_dev_free(ctx, sc->offs, sizeof(unsigned int) * 2 * (size_t)(sc->nblocks + 1));
}

int pcr_grid_nn1(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_xform* x, double max_d2, int32_t* d_idx, double* d_d2) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    grid_scratch sc;
    int rc = grid_search_launch(ctx, idx, qc, x, 0, max_d2, gated, true, &sc);
    if (rc) return rc;
    const int64_t nq = qc->n;
    const int grid = (int)((nq + 255) / 256);
    hipLaunchKernelGGL(grid_finalize_nn1_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (const pcr_pt*)qc->d, (long long)nq,
                       (const unsigned int*)sc.res_pos, (const double*)sc.res_d2, max_d2, gated ? 1 : 0, d_idx, d_d2);
    PCR_HIP(ctx, hipGetLastError());
    grid_scratch_free(ctx, &sc);
    return PCR_OK;
}

int pcr_grid_icp_pass(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_xform* x, double max_d2, int write_back,
                      double* d_moments) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    grid_scratch sc;
    int rc = grid_search_launch(ctx, idx, qc, x, write_back, max_d2, gated, false, &sc);
    if (rc) return rc;
    const int64_t nq = qc->n;
    int grid = (int)((nq + 255) / 256);
    if (grid > 4 * ctx->cu_count) grid = 4 * ctx->cu_count;
    if ((rc = pcr_ensure_scratch(ctx, sizeof(double) * PCR_NMOM * (size_t)grid))) return rc;
    // after a write-back pass the cloud already holds the transformed points
    pcr_prof_mark(ctx, 2);
    hipLaunchKernelGGL(grid_accumulate_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (const pcr_pt*)qc->d, (long long)nq, *x,
                       write_back ? 0 : 1, (const unsigned int*)sc.res_pos, max_d2, gated ? 1 : 0, ctx->d_partials);
    pcr_prof_mark(ctx, 3);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)ctx->d_partials, grid, d_moments);
    pcr_prof_mark(ctx, 4);
    PCR_HIP(ctx, hipGetLastError());
    pcr_prof_finish(ctx);
    grid_scratch_free(ctx, &sc);
    return PCR_OK;
}
