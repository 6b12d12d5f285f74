// Exact 1-NN search over the multi-level voxel-hash grid (see pcr_grid.hip for the layout).
//
// Three stages, each a kernel whose waves are full of queries of the same difficulty
// (mixing them in one kernel left most lanes idle: 88 % of the waves contained at least
// one slow query):
//   ring 1  every query, 8 lanes each: scan the query's own level-0 cell, then only those of
//           the 26 neighbour cells whose box is closer than the best distance so far (or
//           the gate).  Resolved when the bound ball fits in the 3x3x3 block.
//   ring 2  queries whose bound ball fits in the 5x5x5 block: the 98 shell cells, pruned the
//           same way.  Always resolves.
//   hard    everything else, one wave per query: pruned top-down descent of the nested cell
//           hierarchy (cells are contiguous runs of the Morton-sorted cloud at every level).
// Each stage writes res_pos[query] = position of the neighbour in the sorted target (or NONE).
// Pruning only ever skips a cell whose box distance exceeds a bound that is itself >= the
// final answer, so the result is the exact nearest neighbour (lowest index on exact ties).
#include <cmath>
#include <cstdlib>
#include "pcr_grid_dev.h"

#ifndef PCR_SG
#define PCR_SG 8
#endif
constexpr int SG = PCR_SG;                 // lanes per query in rings 1 and 2
constexpr int QPB = 256 / SG;              // queries per ring-1 block = slots per block in the work lists
constexpr unsigned int HARD_SCAN_T = 192;  // the hard stage scans cells up to this size, descends into bigger ones
constexpr int HARD_STACK = 160;

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

__device__ static inline double sq_pos(double v) {
    v = fmax(v, 0.0);
    return v * v;
}

// Work lists are sharded per ring-1 block (QPB slots each, filled from the front): a single
// global append counter saturates at ~88 returning atomics per microsecond on this chip,
// which alone cost more than the whole search.
template <int G>
__device__ static inline void push_item(work_item* __restrict__ list, unsigned int* s_count, double ax, double ay, double az,
                                        double bd2, unsigned int bpos, unsigned int qi) {
    const unsigned int w = blockIdx.x * (256 / G) + atomicAdd(s_count, 1u);  // LDS atomic
    work_item it;
    it.ax = ax; it.ay = ay; it.az = az;
    it.best_d2 = bd2;
    it.best_pos = bpos;
    it.qi = qi;
    list[w] = it;
}

// ------------------------------------------------------------------ ring 1
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

template <int G>
__device__ static inline unsigned long long group_or(unsigned long long v) {
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
}

// Ring-1 search of one query by a group of G lanes.  Cells are first box-tested by
// arithmetic only (each lane a few cells), the survivors are then visited one after the
// other by the whole group: the lookup is one broadcast load, the points are read G at a
// time from consecutive addresses, and the bound shrinks between cells.
template <int G>
__device__ static inline void ring1_body(const pcr_grid_view& gv, pcr_pt* __restrict__ q, long long nq, const pcr_xform& x, int has_x,
                                         int write_back, double max_d2, int gated, unsigned int* __restrict__ res_pos,
                                         double* __restrict__ res_d2, work_item* __restrict__ list_a, work_item* __restrict__ list_b,
                                         unsigned int* s_counts, int dbg) {
    const int gl = threadIdx.x % G;
    const long long qi = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (qi >= nq) return;  // whole groups leave together (G divides 64)
    if (dbg & 16) return;
    pcr_pt p = q[qi];
    double ax = p.x, ay = p.y, az = p.z;
    if (has_x) {
        xform_apply(x, p, &ax, &ay, &az);
        if (write_back && gl == 0) {
            p.x = ax; p.y = ay; p.z = az;
            q[qi] = p;
        }
    }
    bool clamped = false;
    const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
    const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
    const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
    double bd2 = DBL_MAX;
    long long bid = 0x7fffffffffffffffll;
    unsigned int bpos = POS_NONE;
    if (clamped) {
        if (gl == 0) push_item<G>(list_b, s_counts + 1, ax, ay, az, bd2, bpos, (unsigned int)qi);
        return;
    }
    if (dbg & 4) { if (gl == 0) res_pos[qi] = POS_NONE; return; }
    const pcr_cell_slot* __restrict__ tab = gv.table[0];
    const unsigned int mask = gv.mask[0];
    const double cell = gv.cell0;
    // the query's own cell
    {
        unsigned int s, e;
        if (lookup_cell(tab, mask, (unsigned int)cx, (unsigned int)cy, (unsigned int)cz, &s, &e) && !(dbg & 2))
            scan_range(gv.pts, s + gl, e, G, ax, ay, az, bd2, bid, bpos);
        group_best<G>(bd2, bid, bpos);
    }
    double bound2 = gated ? fmin(bd2, max_d2) : bd2;
    const axis_d2 dx2 = make_axis((ax - gv.lo[0]) - (double)(cx - (int)PCR_COORD_BIAS) * cell, cell);
    const axis_d2 dy2 = make_axis((ay - gv.lo[1]) - (double)(cy - (int)PCR_COORD_BIAS) * cell, cell);
    const axis_d2 dz2 = make_axis((az - gv.lo[2]) - (double)(cz - (int)PCR_COORD_BIAS) * cell, cell);
    // which of the 26 neighbours can hold something closer than the bound (bit n of the mask)
    unsigned long long m = 0;
    if (!(dbg & 1)) {
#pragma unroll
        for (int i = 0; i < (27 + G - 1) / G; ++i) {
            const int n = gl + i * G;
            if (n < 27 && n != 13) {
                const int dx = n % 3 - 1, dy = (n / 3) % 3 - 1, dz = n / 9 - 1;
                if ((dx2.at(dx) + dy2.at(dy)) + dz2.at(dz) <= bound2) m |= 1ull << n;
            }
        }
        m = group_or<G>(m);
    }
    while (m) {
        const int n = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int dx = n % 3 - 1, dy = (n / 3) % 3 - 1, dz = n / 9 - 1;
        if ((dx2.at(dx) + dy2.at(dy)) + dz2.at(dz) > bound2) continue;  // the bound may have shrunk meanwhile
        const unsigned int nx = (unsigned int)(cx + dx), ny = (unsigned int)(cy + dy), nz = (unsigned int)(cz + dz);
        if (nx > (unsigned int)PCR_COORD_MAX || ny > (unsigned int)PCR_COORD_MAX || nz > (unsigned int)PCR_COORD_MAX) continue;
        unsigned int s, e;
        if (!lookup_cell(tab, mask, nx, ny, nz, &s, &e)) continue;
        if (dbg & 8) continue;
        scan_range(gv.pts, s + gl, e, G, ax, ay, az, bd2, bid, bpos);
        group_best<G>(bd2, bid, bpos);
        bound2 = gated ? fmin(bd2, max_d2) : bd2;
    }
    if (gl != 0) return;
    const double safe = cell * (1.0 - 1e-9);
    if (bound2 <= safe * safe || dbg) {  // the bound ball lies inside the 3x3x3 block: exact
        res_pos[qi] = bpos;
        if (res_d2) res_d2[qi] = bd2;
    } else if (bound2 <= 4.0 * safe * safe) {
        push_item<G>(list_a, s_counts, ax, ay, az, bd2, bpos, (unsigned int)qi);
    } else {
        push_item<G>(list_b, s_counts + 1, ax, ay, az, bd2, bpos, (unsigned int)qi);
    }
}

template <int G>
__global__ void __launch_bounds__(256)
grid_ring1_kernel(pcr_grid_view gv, pcr_pt* __restrict__ q, long long nq, pcr_xform x, int has_x, int write_back, double max_d2,
                  int gated, unsigned int* __restrict__ res_pos, double* __restrict__ res_d2, work_item* __restrict__ list_a,
                  work_item* __restrict__ list_b, unsigned int* __restrict__ counts /* [block][2]: ring 2, hard */, int dbg) {
    __shared__ unsigned int s_counts[2];
    if (threadIdx.x < 2) s_counts[threadIdx.x] = 0;
    __syncthreads();
    ring1_body<G>(gv, q, nq, x, has_x, write_back, max_d2, gated, res_pos, res_d2, list_a, list_b, s_counts, dbg);
    __syncthreads();
    if (threadIdx.x < 2) counts[blockIdx.x * 2 + threadIdx.x] = s_counts[threadIdx.x];
}

// ------------------------------------------------------------------ ring 2
// Items of block b of ring 1 sit in list[b*QPB ...]; G lanes per item, same scheme as ring 1
// over the 98 shell cells of the 5x5x5 block.
template <int G, int QPB>
__global__ void __launch_bounds__(256)
grid_ring2_kernel(pcr_grid_view gv, const work_item* __restrict__ list, const unsigned int* __restrict__ count_p, double max_d2, int gated,
                  unsigned int* __restrict__ res_pos, double* __restrict__ res_d2) {
    const int gl = threadIdx.x % G;
    const unsigned int count = count_p[blockIdx.x * 2];
    const pcr_cell_slot* __restrict__ tab = gv.table[0];
    const unsigned int mask = gv.mask[0];
    const double cell = gv.cell0;
    for (unsigned int g = threadIdx.x / G; g < count; g += 256 / G) {
        const work_item it = list[(size_t)blockIdx.x * QPB + g];
        const double ax = it.ax, ay = it.ay, az = it.az;
        double bd2 = DBL_MAX;
        long long bid = 0x7fffffffffffffffll;
        unsigned int bpos = POS_NONE;
        if (it.best_pos != POS_NONE) {
            bd2 = it.best_d2;
            bpos = it.best_pos;
            bid = gv.pts[bpos].id;
        }
        double bound2 = gated ? fmin(bd2, max_d2) : bd2;
        bool clamped = false;
        const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
        const axis_d2 dx2 = make_axis((ax - gv.lo[0]) - (double)(cx - (int)PCR_COORD_BIAS) * cell, cell);
        const axis_d2 dy2 = make_axis((ay - gv.lo[1]) - (double)(cy - (int)PCR_COORD_BIAS) * cell, cell);
        const axis_d2 dz2 = make_axis((az - gv.lo[2]) - (double)(cz - (int)PCR_COORD_BIAS) * cell, cell);
        // shell cells that can hold something closer than the bound: 125 bits in two words
        unsigned long long m0 = 0, m1 = 0;
#pragma unroll
        for (int i = 0; i < (125 + G - 1) / G; ++i) {
            const int c = gl + i * G;
            const int ix = c % 5, iy = (c / 5) % 5, iz = c / 25;
            const bool shell = c < 125 && (ix == 0 || ix == 4 || iy == 0 || iy == 4 || iz == 0 || iz == 4);
            if (shell && (dx2.at(ix - 2) + dy2.at(iy - 2)) + dz2.at(iz - 2) <= bound2) {
                if (c < 64) m0 |= 1ull << c;
                else m1 |= 1ull << (c - 64);
            }
        }
        m0 = group_or<G>(m0);
        m1 = group_or<G>(m1);
        while (m0 | m1) {
            int c;
            if (m0) { c = __ffsll((long long)m0) - 1; m0 &= m0 - 1; }
            else { c = 64 + __ffsll((long long)m1) - 1; m1 &= m1 - 1; }
            const int ix = c % 5, iy = (c / 5) % 5, iz = c / 25;
            if ((dx2.at(ix - 2) + dy2.at(iy - 2)) + dz2.at(iz - 2) > bound2) continue;
            const unsigned int nx = (unsigned int)(cx + ix - 2), ny = (unsigned int)(cy + iy - 2), nz = (unsigned int)(cz + iz - 2);
            if (nx > (unsigned int)PCR_COORD_MAX || ny > (unsigned int)PCR_COORD_MAX || nz > (unsigned int)PCR_COORD_MAX) continue;
            unsigned int s, e;
            if (!lookup_cell(tab, mask, nx, ny, nz, &s, &e)) continue;
            scan_range(gv.pts, s + gl, e, G, ax, ay, az, bd2, bid, bpos);
            group_best<G>(bd2, bid, bpos);
            bound2 = gated ? fmin(bd2, max_d2) : bd2;
        }
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
grid_hard_kernel(pcr_grid_view gv, const work_item* __restrict__ list, const unsigned int* __restrict__ count_p, double max_d2, int gated,
                 unsigned int* __restrict__ res_pos, double* __restrict__ res_d2) {
    __shared__ hard_entry s_stack[4][HARD_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    hard_entry* stack = s_stack[wave];
    const unsigned int count = count_p[blockIdx.x * 2 + 1];
    const int top = gv.levels - 1;
    for (unsigned int w = wave; w < count; w += 4) {
        const work_item it = list[blockIdx.x * QPB + w];
        const double ax = it.ax, ay = it.ay, az = it.az;
        double bd2 = DBL_MAX;
        long long bid = 0x7fffffffffffffffll;
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
    work_item* list_a = nullptr;
    work_item* list_b = nullptr;
    unsigned int* counts = nullptr;  // [nblocks][2]
    int64_t nq = 0;
    int nblocks = 0;
};

// Runs the three search stages; leaves res_pos (and res_d2 when asked) on the device.
static int grid_search_launch(pcr_ctx* ctx, const pcr_index* idx, pcr_pt* q, int64_t nq, const pcr_xform* x, int write_back,
                              double max_d2, bool gated, bool want_d2, grid_scratch* sc) {
    int rc;
    sc->nq = nq;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * nq, (void**)&sc->res_pos))) return rc;
    if (want_d2 && (rc = pcr_dev_alloc(ctx, sizeof(double) * nq, (void**)&sc->res_d2))) return rc;
    const int block = 256;
    const long long threads = (long long)nq * SG;
    const int grid1 = (int)((threads + block - 1) / block);
    sc->nblocks = grid1;
    if ((rc = pcr_dev_alloc(ctx, sizeof(work_item) * QPB * (size_t)grid1, (void**)&sc->list_a))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(work_item) * QPB * (size_t)grid1, (void**)&sc->list_b))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * 2 * (size_t)grid1, (void**)&sc->counts))) return rc;
    pcr_xform xi;
    pcr_xform_from_T(nullptr, &xi);
    static const int dbg = getenv("PCR_DEBUG_MODE") ? atoi(getenv("PCR_DEBUG_MODE")) : 0;  // timing experiments only
    pcr_prof_mark(ctx, 0);
    hipLaunchKernelGGL(grid_ring1_kernel<SG>, dim3(grid1), dim3(block), 0, ctx->stream, idx->view, q, (long long)nq, x ? *x : xi, x ? 1 : 0,
                       write_back, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2, sc->list_a, sc->list_b, sc->counts, dbg);
    // later stages: fixed grids walk the device-side lists (their lengths are only known on the device)
    pcr_prof_mark(ctx, 1);
    hipLaunchKernelGGL((grid_ring2_kernel<SG, QPB>), dim3(grid1), dim3(256), 0, ctx->stream, idx->view, (const work_item*)sc->list_a,
                       (const unsigned int*)sc->counts, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2);
    hipLaunchKernelGGL(grid_hard_kernel, dim3(grid1), dim3(256), 0, ctx->stream, idx->view, (const work_item*)sc->list_b,
                       (const unsigned int*)sc->counts, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

static void grid_scratch_free(pcr_ctx* ctx, grid_scratch* sc) {
    pcr_dev_free(ctx, sc->res_pos, sizeof(unsigned int) * sc->nq);
    if (sc->res_d2) pcr_dev_free(ctx, sc->res_d2, sizeof(double) * sc->nq);
    pcr_dev_free(ctx, sc->list_a, sizeof(work_item) * QPB * (size_t)sc->nblocks);
    pcr_dev_free(ctx, sc->list_b, sizeof(work_item) * QPB * (size_t)sc->nblocks);
    pcr_dev_free(ctx, sc->counts, sizeof(unsigned int) * 2 * (size_t)sc->nblocks);
}

int pcr_grid_nn1(pcr_ctx* ctx, const pcr_index* idx, const pcr_pt* q, int64_t nq, const pcr_xform* x, double max_d2,
                 int32_t* d_idx, double* d_d2) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    grid_scratch sc;
    int rc = grid_search_launch(ctx, idx, (pcr_pt*)q, nq, x, 0, max_d2, gated, true, &sc);
    if (rc) return rc;
    const int grid = (int)((nq + 255) / 256);
    hipLaunchKernelGGL(grid_finalize_nn1_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, q, (long long)nq,
                       (const unsigned int*)sc.res_pos, (const double*)sc.res_d2, max_d2, gated ? 1 : 0, d_idx, d_d2);
    PCR_HIP(ctx, hipGetLastError());
    grid_scratch_free(ctx, &sc);
    return PCR_OK;
}

int pcr_grid_icp_pass(pcr_ctx* ctx, const pcr_index* idx, pcr_pt* q, int64_t nq, const pcr_xform* x, double max_d2,
                      int write_back, double* d_moments) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    grid_scratch sc;
    int rc = grid_search_launch(ctx, idx, q, nq, x, write_back, max_d2, gated, false, &sc);
    if (rc) return rc;
    int grid = (int)((nq + 255) / 256);
    if (grid > 4 * ctx->cu_count) grid = 4 * ctx->cu_count;
    if ((rc = pcr_ensure_scratch(ctx, sizeof(double) * PCR_NMOM * (size_t)grid))) return rc;
    // after a write-back pass q already holds the transformed points
    pcr_prof_mark(ctx, 2);
    hipLaunchKernelGGL(grid_accumulate_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (const pcr_pt*)q, (long long)nq, *x,
                       write_back ? 0 : 1, (const unsigned int*)sc.res_pos, max_d2, gated ? 1 : 0, ctx->d_partials);
    pcr_prof_mark(ctx, 3);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)ctx->d_partials, grid, d_moments);
    pcr_prof_mark(ctx, 4);
    PCR_HIP(ctx, hipGetLastError());
    pcr_prof_finish(ctx);
    grid_scratch_free(ctx, &sc);
    return PCR_OK;
}
