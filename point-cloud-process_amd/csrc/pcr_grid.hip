// Multi-level voxel-hash grid over the target cloud + exact 1-NN search kernels.
//
// Layout in HBM (per target cloud, built once per pair; stands in for
// o3d.geometry.KDTreeFlann(target), Registration/main.py:105):
//   * sorted[n]       32-B records {x,y,z,id}, ordered by the 63-bit Morton key of the
//                     level-0 cell (cell0 metres).  Level l has cells of cell0*4^l and
//                     its cell id is key >> 6l, so EVERY level's cell is one contiguous
//                     run of `sorted` (nested octree property of the Morton order).
//   * table[l][cap_l] open-addressing hash: level-l cell id -> [start,end) in `sorted`
//                     (16-B slots, load factor <= 0.5; ~1 MB at level 0 for a KITTI scan,
//                     i.e. L2-resident).
// Search (per query, G lanes cooperate): at level l examine the 3x3x3 cells around
// the query; any point outside that block is >= cell_l away, so a best distance
// <= cell_l is exact.  Otherwise go one level up (cells 4x larger).  With a gate
// (max_d2) the climb stops at the first level whose cell covers the gate radius.
// The voxel binning is the same floor((p-min)/leaf) used by voxel_filter.py:30-32.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include "pcr_internal.h"

// ----------------------------------------------------------------- helpers
__host__ __device__ static inline unsigned long long spread21(unsigned long long x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__host__ __device__ static inline unsigned long long mix64(unsigned long long k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// level-0 integer cell coordinate (biased, clamped).  *clamped is set when the
// point lies outside the representable range (then only a linear scan is exact).
__device__ static inline long long cell_coord(double v, double lo, double inv, bool* clamped) {
    double f = floor((v - lo) * inv);
    // |f| beyond 2^20 cells: clamp (also catches NaN/inf via the comparisons below)
    if (!(f >= -(double)(PCR_COORD_BIAS)) || !(f <= (double)(PCR_COORD_MAX - PCR_COORD_BIAS))) {
        *clamped = true;
        return f > 0 ? PCR_COORD_MAX : 0;
    }
    return (long long)f + PCR_COORD_BIAS;
}

// ------------------------------------------------------------ build kernels
__global__ void bbox_partial_kernel(const pcr_pt* __restrict__ pts, long long n, double* __restrict__ part) {
    __shared__ double s[6][256];
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        pcr_pt p = pts[i];
        mn[0] = fmin(mn[0], p.x); mx[0] = fmax(mx[0], p.x);
        mn[1] = fmin(mn[1], p.y); mx[1] = fmax(mx[1], p.y);
        mn[2] = fmin(mn[2], p.z); mx[2] = fmax(mx[2], p.z);
    }
    for (int k = 0; k < 3; ++k) { s[k][threadIdx.x] = mn[k]; s[3 + k][threadIdx.x] = mx[k]; }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            for (int k = 0; k < 3; ++k) {
                s[k][threadIdx.x] = fmin(s[k][threadIdx.x], s[k][threadIdx.x + st]);
                s[3 + k][threadIdx.x] = fmax(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + st]);
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = s[threadIdx.x][0];
}

__global__ void morton_keys_kernel(const pcr_pt* __restrict__ pts, long long n, double lox, double loy, double loz,
                                   double inv, unsigned long long* __restrict__ keys, unsigned int* __restrict__ vals) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pcr_pt p = pts[i];
    bool cl = false;
    unsigned long long cx = (unsigned long long)cell_coord(p.x, lox, inv, &cl);
    unsigned long long cy = (unsigned long long)cell_coord(p.y, loy, inv, &cl);
    unsigned long long cz = (unsigned long long)cell_coord(p.z, loz, inv, &cl);
    keys[i] = spread21(cx) | (spread21(cy) << 1) | (spread21(cz) << 2);
    vals[i] = (unsigned int)i;
}

__global__ void gather_sorted_kernel(const pcr_pt* __restrict__ pts, const unsigned int* __restrict__ perm, long long n,
                                     pcr_pt* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = pts[perm[i]];
}

__global__ void count_cells_kernel(const unsigned long long* __restrict__ keys, long long n, int levels,
                                   unsigned int* __restrict__ counts) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = keys[i];
    unsigned long long kp = (i > 0) ? keys[i - 1] : 0;
    for (int l = 0; l < levels; ++l) {
        bool start = (i == 0) || ((k >> (6 * l)) != (kp >> (6 * l)));
        // hipcc folds these per-lane adds into one atomic per wave
        if (start) atomicAdd(&counts[l], 1u);
    }
}

struct pcr_tables {
    pcr_cell_slot* t[PCR_MAX_LEVELS];
    unsigned int mask[PCR_MAX_LEVELS];
};

__host__ __device__ static inline unsigned int compact21(unsigned long long x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffffull;
    return (unsigned int)x;
}

__device__ static inline unsigned int slot_find_or_insert(pcr_cell_slot* tab, unsigned int mask, unsigned long long key, unsigned int h) {
    h &= mask;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
        unsigned long long old = atomicCAS(&tab[h].key, PCR_EMPTY_KEY, key);
        if (old == PCR_EMPTY_KEY || old == key) return h;
        h = (h + 1) & mask;
    }
    return 0xffffffffu;  // table full: cannot happen at load factor <= 0.5
}

__global__ void insert_cells_kernel(const unsigned long long* __restrict__ keys, long long n, int levels, pcr_tables tabs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long k = keys[i];
    unsigned long long kp = (i > 0) ? keys[i - 1] : 0;
    unsigned long long kn = (i + 1 < n) ? keys[i + 1] : 0;
    for (int l = 0; l < levels; ++l) {
        unsigned long long ck = k >> (6 * l);
        bool start = (i == 0) || (ck != (kp >> (6 * l)));
        bool end = (i + 1 == n) || (ck != (kn >> (6 * l)));
        if (start || end) {
            // table key = packed level-l cell coordinates (cheap to form at query time)
            const unsigned int X = compact21(ck), Y = compact21(ck >> 1), Z = compact21(ck >> 2);
            const unsigned long long pk = (unsigned long long)X | ((unsigned long long)Y << 21) | ((unsigned long long)Z << 42);
            unsigned int h = slot_find_or_insert(tabs.t[l], tabs.mask[l], pk, (X * 73856093u) ^ (Y * 19349663u) ^ (Z * 83492791u));
            if (h != 0xffffffffu) {
                if (start) tabs.t[l][h].start = (unsigned int)i;
                if (end) tabs.t[l][h].end = (unsigned int)(i + 1);
            }
        }
    }
}

// ----------------------------------------------------------- search kernels
// Phase 1 (G lanes per query): the 3x3x3 level-0 cells around the query.  Resolves the
//   query when the best distance is <= cell0 (nothing outside the block can be closer) or
//   when the block already covers the gate ball.  Everything else goes to a work list.
// Phase 2 (one wave per listed query): pruned top-down descent of the nested cell
//   hierarchy.  A cell is scanned cooperatively when it is small, otherwise its 64
//   children (one per lane) are box-tested against the current bound and probed.
// Both write res_pos[query] = position of the neighbour in `sorted` (or NONE).
constexpr unsigned int POS_NONE = 0xffffffffu;
constexpr int P1_G = 8;             // lanes per query in phase 1
constexpr int P1_NC = (27 + P1_G - 1) / P1_G;
constexpr unsigned int P1_CELL_CAP = 96;   // a level-0 cell with more points than this is left to phase 2
constexpr unsigned int P2_SCAN_T = 192;    // phase 2 scans cells up to this size, descends into bigger ones
constexpr int P2_STACK = 160;

struct __attribute__((aligned(16))) work_item {
    double ax, ay, az;
    double best_d2;
    unsigned int best_pos;
    unsigned int qi;
};

__device__ static inline bool better(double d2, long long id, double bd2, long long bid) {
    return d2 < bd2 || (d2 == bd2 && id < bid);
}

// Exact squared distance, evaluated exactly like the host oracle:
// (dx*dx + dy*dy) + dz*dz with each operation rounded (no FMA; the TU is built
// with -ffp-contract=off).
__device__ static inline double dist2(double ax, double ay, double az, const pcr_pt& b) {
    double dx = ax - b.x, dy = ay - b.y, dz = az - b.z;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ static inline unsigned int cell_hash(unsigned int x, unsigned int y, unsigned int z) {
    return (x * 73856093u) ^ (y * 19349663u) ^ (z * 83492791u);
}

__device__ static inline unsigned long long cell_pack(unsigned int x, unsigned int y, unsigned int z) {
    return (unsigned long long)x | ((unsigned long long)y << 21) | ((unsigned long long)z << 42);
}

__device__ static inline bool lookup_cell(const pcr_cell_slot* __restrict__ tab, unsigned int mask, unsigned int x, unsigned int y,
                                          unsigned int z, unsigned int* s, unsigned int* e) {
    const unsigned long long key = cell_pack(x, y, z);
    unsigned int h = cell_hash(x, y, z) & mask;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
        pcr_cell_slot sl = tab[h];
        if (sl.key == key) { *s = sl.start; *e = sl.end; return true; }
        if (sl.key == PCR_EMPTY_KEY) return false;
        h = (h + 1) & mask;
    }
    return false;
}

__device__ static inline void xform_apply(const pcr_xform& x, const pcr_pt& p, double* ax, double* ay, double* az) {
    *ax = ((x.r[0] * p.x + x.r[1] * p.y) + x.r[2] * p.z) + x.t[0];
    *ay = ((x.r[3] * p.x + x.r[4] * p.y) + x.r[5] * p.z) + x.t[1];
    *az = ((x.r[6] * p.x + x.r[7] * p.y) + x.r[8] * p.z) + x.t[2];
}

__global__ void __launch_bounds__(256)
grid_phase1_kernel(pcr_grid_view gv, pcr_pt* __restrict__ q, long long nq, pcr_xform x, int has_x, int write_back, double max_d2,
                   int gated, unsigned int* __restrict__ res_pos, double* __restrict__ res_d2, work_item* __restrict__ work,
                   unsigned int* __restrict__ work_count) {
    constexpr int G = P1_G;
    const int gl = threadIdx.x % G;
    const long long qi = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (qi >= nq) return;  // whole groups leave together (G divides 64)
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
    const long long cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
    const long long cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
    const long long cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
    double bd2 = DBL_MAX;
    long long bid = 0x7fffffffffffffffll;
    unsigned int bpos = POS_NONE;
    bool hard = clamped;
    if (!clamped) {
        const pcr_cell_slot* __restrict__ tab = gv.table[0];
        const unsigned int mask = gv.mask[0];
        unsigned int cs[P1_NC], ce[P1_NC];
        // step A: all of this lane's cell lookups (independent loads in flight together)
#pragma unroll
        for (int i = 0; i < P1_NC; ++i) {
            const int c = gl + i * G;
            cs[i] = 0; ce[i] = 0;
            if (c < 27) {
                const long long nx = cx + (c % 3 - 1), ny = cy + ((c / 3) % 3 - 1), nz = cz + (c / 9 - 1);
                if (nx >= 0 && ny >= 0 && nz >= 0 && nx <= PCR_COORD_MAX && ny <= PCR_COORD_MAX && nz <= PCR_COORD_MAX) {
                    unsigned int s, e;
                    if (lookup_cell(tab, mask, (unsigned int)nx, (unsigned int)ny, (unsigned int)nz, &s, &e)) { cs[i] = s; ce[i] = e; }
                }
            }
        }
        // step B: scan the owned cells
#pragma unroll
        for (int i = 0; i < P1_NC; ++i) {
            if (ce[i] - cs[i] > P1_CELL_CAP) { hard = true; continue; }
            for (unsigned int j = cs[i]; j < ce[i]; ++j) {
                const pcr_pt b = gv.pts[j];
                const double d2 = dist2(ax, ay, az, b);
                if (better(d2, b.id, bd2, bid)) { bd2 = d2; bid = b.id; bpos = j; }
            }
        }
    }
    // group reduction (G aligned lanes)
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
        const double od2 = __shfl_xor(bd2, off, 64);
        const long long oid = __shfl_xor(bid, off, 64);
        const unsigned int opos = __shfl_xor(bpos, off, 64);
        const int ohard = __shfl_xor((int)hard, off, 64);
        if (better(od2, oid, bd2, bid)) { bd2 = od2; bid = oid; bpos = opos; }
        hard = hard || (ohard != 0);
    }
    if (gl != 0) return;
    const double safe = gv.cell0 * (1.0 - 1e-9);
    bool resolved = false;
    if (!hard) {
        if (bd2 <= safe * safe) resolved = true;                      // nothing outside the block can be closer
        else if (gated && safe * safe >= max_d2) resolved = true;     // the block covers the gate ball
    }
    if (resolved) {
        res_pos[qi] = bpos;
        if (res_d2) res_d2[qi] = bd2;
    } else {
        const unsigned int w = atomicAdd(work_count, 1u);
        work_item it;
        it.ax = ax; it.ay = ay; it.az = az;
        it.best_d2 = bd2;
        it.best_pos = bpos;
        it.qi = (unsigned int)qi;
        work[w] = it;
    }
}

struct p2_entry {
    unsigned int start, end;
    unsigned int x, y, z;
    int level;
};

__device__ static inline double box_dist2(const pcr_grid_view& gv, int level, double cell, unsigned int X, unsigned int Y, unsigned int Z,
                                          double ax, double ay, double az) {
    const long long bl = PCR_COORD_BIAS >> (2 * level);
    const double slack = cell * 1e-9;
    const double x0 = gv.lo[0] + (double)((long long)X - bl) * cell;
    const double y0 = gv.lo[1] + (double)((long long)Y - bl) * cell;
    const double z0 = gv.lo[2] + (double)((long long)Z - bl) * cell;
    double dx = fmax(fmax(x0 - ax, ax - (x0 + cell)), 0.0);
    double dy = fmax(fmax(y0 - ay, ay - (y0 + cell)), 0.0);
    double dz = fmax(fmax(z0 - az, az - (z0 + cell)), 0.0);
    dx = fmax(dx - slack, 0.0);
    dy = fmax(dy - slack, 0.0);
    dz = fmax(dz - slack, 0.0);
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ static inline double wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ void __launch_bounds__(256)
grid_phase2_kernel(pcr_grid_view gv, const work_item* __restrict__ work, const unsigned int* __restrict__ work_count, double max_d2,
                   int gated, unsigned int* __restrict__ res_pos, double* __restrict__ res_d2) {
    __shared__ p2_entry s_stack[4][P2_STACK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    p2_entry* stack = s_stack[wave];
    const unsigned int count = *work_count;
    const unsigned int n_waves = gridDim.x * 4;
    double cells[PCR_MAX_LEVELS];
    {
        double c = gv.cell0;
#pragma unroll
        for (int l = 0; l < PCR_MAX_LEVELS; ++l) { cells[l] = c; c *= 4.0; }
    }
    const int top = gv.levels - 1;
    for (unsigned int w = blockIdx.x * 4 + wave; w < count; w += n_waves) {
        const work_item it = work[w];
        const double ax = it.ax, ay = it.ay, az = it.az;
        double bd2 = DBL_MAX;
        long long bid = 0x7fffffffffffffffll;
        unsigned int bpos = POS_NONE;
        if (lane == 0 && it.best_pos != POS_NONE) {
            bd2 = it.best_d2;
            bpos = it.best_pos;
            bid = gv.pts[bpos].id;
        }
        double bound2 = it.best_d2;  // DBL_MAX when nothing was found
        if (gated) bound2 = fmin(bound2, max_d2);
        bool clamped = false;
        const long long cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
        const long long cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
        const long long cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
        // start level: the smallest one whose 3x3x3 block covers the bound ball
        int s_level = -1;
        if (!clamped && bound2 < DBL_MAX) {
            for (int l = 0; l <= top; ++l) {
                const double safe = cells[l] * (1.0 - 1e-9);
                if (safe * safe >= bound2) { s_level = l; break; }
            }
        }
        int sp = 0;  // wave-uniform stack pointer
        {
            // initial cells: the query's 3x3x3 block at s_level, or the <= 8 root cells that hold the whole target
            const bool roots = (s_level < 0);
            const int lvl = roots ? top : s_level;
            const long long b0 = PCR_COORD_BIAS >> (2 * lvl);
            long long X = 0, Y = 0, Z = 0;
            bool valid;
            if (roots) {
                valid = lane < 8;
                X = b0 + (lane & 1); Y = b0 + ((lane >> 1) & 1); Z = b0 + ((lane >> 2) & 1);
            } else {
                valid = lane < 27;
                X = (cx >> (2 * lvl)) + (lane % 3 - 1);
                Y = (cy >> (2 * lvl)) + ((lane / 3) % 3 - 1);
                Z = (cz >> (2 * lvl)) + (lane / 9 - 1);
                const long long lim = PCR_COORD_MAX >> (2 * lvl);
                valid = valid && X >= 0 && Y >= 0 && Z >= 0 && X <= lim && Y <= lim && Z <= lim;
            }
            unsigned int s = 0, e = 0;
            double bdist = 0.0;
            if (valid) {
                bdist = box_dist2(gv, lvl, cells[lvl], (unsigned int)X, (unsigned int)Y, (unsigned int)Z, ax, ay, az);
                valid = bdist <= bound2 && lookup_cell(gv.table[lvl], gv.mask[lvl], (unsigned int)X, (unsigned int)Y, (unsigned int)Z, &s, &e);
            }
            // far cells first, the cell containing the query last (popped first)
            const unsigned long long m_far = __ballot(valid && bdist > 0.0);
            const unsigned long long m_near = __ballot(valid && !(bdist > 0.0));
            const unsigned long long below = (1ull << lane) - 1ull;
            int slot = -1;
            if (valid && bdist > 0.0) slot = __popcll(m_far & below);
            else if (valid) slot = __popcll(m_far) + __popcll(m_near & below);
            if (slot >= 0 && slot < P2_STACK) {
                p2_entry en;
                en.start = s; en.end = e; en.x = (unsigned int)X; en.y = (unsigned int)Y; en.z = (unsigned int)Z; en.level = lvl;
                stack[slot] = en;
            }
            sp = __popcll(m_far) + __popcll(m_near);
        }
        while (sp > 0) {
            --sp;
            const p2_entry en = stack[sp];  // same address in every lane: LDS broadcast
            const double cell = cells[en.level];
            if (box_dist2(gv, en.level, cell, en.x, en.y, en.z, ax, ay, az) > bound2) continue;
            const unsigned int cnt = en.end - en.start;
            const bool room = sp + 64 <= P2_STACK;
            if (en.level == 0 || cnt <= P2_SCAN_T || !room) {
                for (unsigned int j = en.start + lane; j < en.end; j += 64) {
                    const pcr_pt b = gv.pts[j];
                    const double d2 = dist2(ax, ay, az, b);
                    if (better(d2, b.id, bd2, bid)) { bd2 = d2; bid = b.id; bpos = j; }
                }
                bound2 = fmin(bound2, wave_min(bd2));
            } else {
                const int cl = en.level - 1;
                const unsigned int X = en.x * 4u + (lane & 3), Y = en.y * 4u + ((lane >> 2) & 3), Z = en.z * 4u + (lane >> 4);
                const double bdist = box_dist2(gv, cl, cells[cl], X, Y, Z, ax, ay, az);
                unsigned int s = 0, e = 0;
                const bool valid = bdist <= bound2 && lookup_cell(gv.table[cl], gv.mask[cl], X, Y, Z, &s, &e);
                const unsigned long long m_far = __ballot(valid && bdist > 0.0);
                const unsigned long long m_near = __ballot(valid && !(bdist > 0.0));
                const unsigned long long below = (1ull << lane) - 1ull;
                int slot = -1;
                if (valid && bdist > 0.0) slot = __popcll(m_far & below);
                else if (valid) slot = __popcll(m_far) + __popcll(m_near & below);
                if (slot >= 0) {
                    p2_entry ch;
                    ch.start = s; ch.end = e; ch.x = X; ch.y = Y; ch.z = Z; ch.level = cl;
                    stack[sp + slot] = ch;
                }
                sp += __popcll(m_far) + __popcll(m_near);
            }
        }
        // final wave reduction with the lowest-index tie rule
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

// nn1 epilogue: sorted position -> original target index, gate, scatter to the query's original slot
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

// ICP epilogue: gate + Procrustes moments about gv.origin.  One partial slab of PCR_NMOM
// doubles per block, summed in fixed order by reduce_partials_kernel (bitwise reproducible).
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
        double v = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
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
static int next_pow2(unsigned int v) {
    unsigned int p = 1;
    while (p < v) p <<= 1;
    return (int)p;
}

int pcr_bbox(pcr_ctx* ctx, const pcr_pt* pts, long long n, double lo[3], double hi[3]) {
    const int grid_n = (int)((n + 255) / 256);
    int nb = grid_n < 256 ? grid_n : 256;
    double* d_part = nullptr;
    int rc = pcr_dev_alloc(ctx, sizeof(double) * 6 * nb, (void**)&d_part);
    if (rc) return rc;
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(nb), dim3(256), 0, ctx->stream, pts, n, d_part);
    std::vector<double> h_part(6 * nb);
    PCR_HIP(ctx, hipMemcpyAsync(h_part.data(), d_part, sizeof(double) * 6 * nb, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pcr_dev_free(ctx, d_part, sizeof(double) * 6 * nb);
    for (int k = 0; k < 3; ++k) { lo[k] = DBL_MAX; hi[k] = -DBL_MAX; }
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < 3; ++k) {
            lo[k] = fmin(lo[k], h_part[6 * b + k]);
            hi[k] = fmax(hi[k], h_part[6 * b + 3 + k]);
        }
    for (int k = 0; k < 3; ++k)
        if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) { ctx->last_error = "non-finite coordinates in cloud"; return PCR_E_INVALID; }
    return PCR_OK;
}

// idx->lo/hi must already hold the target's bounding box (pcr_index_build computes it).
int pcr_grid_build(pcr_ctx* ctx, const pcr_cloud* tgt, double cell, pcr_index* idx) {
    const long long n = tgt->n;
    const int block = 256;
    const int grid_n = (int)((n + block - 1) / block);
    int rc;
    double lo[3] = {idx->lo[0], idx->lo[1], idx->lo[2]}, hi[3] = {idx->hi[0], idx->hi[1], idx->hi[2]};
    double ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    double emax = fmax(ext[0], fmax(ext[1], ext[2]));
    if (!(cell > 0)) {
        // surface-like clouds: ~N points over the two largest extents; aim at a few points per cell
        double e[3] = {ext[0], ext[1], ext[2]};
        if (e[0] < e[1]) std::swap(e[0], e[1]);
        if (e[1] < e[2]) std::swap(e[1], e[2]);
        if (e[0] < e[1]) std::swap(e[0], e[1]);
        double area = e[0] * e[1];
        cell = (area > 0) ? 0.55 * sqrt(area / (double)n) : (emax > 0 ? emax / 64.0 : 1.0);
    }
    // keep every coordinate inside 2^18 level-0 cells of lo, and cell strictly positive
    double min_cell = emax / 262144.0;
    if (cell < min_cell) cell = min_cell;
    if (!(cell > 0)) cell = 1.0;
    int levels = 1;
    {
        double c = cell;
        while (c < emax * (1.0 + 1e-9) && levels < PCR_MAX_LEVELS) { c *= 4.0; ++levels; }
    }
    idx->cell = cell;
    // ---- keys, sort, gather
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    unsigned int *d_vals = nullptr, *d_vals2 = nullptr;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned long long) * n, (void**)&d_keys))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned long long) * n, (void**)&d_keys2))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_vals))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_vals2))) return rc;
    const double inv = 1.0 / cell;
    hipLaunchKernelGGL(morton_keys_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const pcr_pt*)tgt->d, n, lo[0], lo[1], lo[2],
                       inv, d_keys, d_vals);
    size_t temp_bytes = 0;
    PCR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0, 63, ctx->stream));
    void* d_temp = nullptr;
    if ((rc = pcr_dev_alloc(ctx, temp_bytes, &d_temp))) return rc;
    PCR_HIP(ctx, rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0, 63, ctx->stream));
    if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * n, (void**)&idx->sorted))) return rc;
    hipLaunchKernelGGL(gather_sorted_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const pcr_pt*)tgt->d,
                       (const unsigned int*)d_vals2, n, idx->sorted);
    // ---- per-level tables
    unsigned int* d_counts = ctx->d_counters;  // 16 words at offset 0
    PCR_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(unsigned int) * 16, ctx->stream));
    hipLaunchKernelGGL(count_cells_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const unsigned long long*)d_keys2, n, levels,
                       d_counts);
    unsigned int h_counts[16];
    PCR_HIP(ctx, hipMemcpyAsync(h_counts, d_counts, sizeof(h_counts), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pcr_tables tabs;
    memset(&tabs, 0, sizeof(tabs));
    for (int l = 0; l < levels; ++l) {
        unsigned int cap = (unsigned int)next_pow2(h_counts[l] * 2 + 2);
        if (cap < 16) cap = 16;
        idx->caps[l] = cap;
        if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_cell_slot) * cap, (void**)&idx->tables[l]))) return rc;
        PCR_HIP(ctx, hipMemsetAsync(idx->tables[l], 0xff, sizeof(pcr_cell_slot) * cap, ctx->stream));
        tabs.t[l] = idx->tables[l];
        tabs.mask[l] = cap - 1;
    }
    hipLaunchKernelGGL(insert_cells_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const unsigned long long*)d_keys2, n, levels, tabs);
    PCR_HIP(ctx, hipGetLastError());
    pcr_dev_free(ctx, d_temp, temp_bytes);
    pcr_dev_free(ctx, d_keys, sizeof(unsigned long long) * n);
    pcr_dev_free(ctx, d_keys2, sizeof(unsigned long long) * n);
    pcr_dev_free(ctx, d_vals, sizeof(unsigned int) * n);
    pcr_dev_free(ctx, d_vals2, sizeof(unsigned int) * n);
    // ---- view
    pcr_grid_view& v = idx->view;
    memset(&v, 0, sizeof(v));
    v.pts = idx->sorted;
    v.n = n;
    v.levels = levels;
    v.cell0 = cell;
    v.inv_cell0 = inv;
    for (int k = 0; k < 3; ++k) {
        v.lo[k] = lo[k];
        v.origin[k] = 0.5 * (lo[k] + hi[k]);
    }
    for (int l = 0; l < levels; ++l) {
        v.table[l] = idx->tables[l];
        v.mask[l] = idx->caps[l] - 1;
    }
    return PCR_OK;
}

void pcr_grid_free(pcr_ctx* ctx, pcr_index* idx) {
    pcr_dev_free(ctx, idx->sorted, sizeof(pcr_pt) * idx->n);
    idx->sorted = nullptr;
    for (int l = 0; l < PCR_MAX_LEVELS; ++l) {
        if (idx->tables[l]) pcr_dev_free(ctx, idx->tables[l], sizeof(pcr_cell_slot) * idx->caps[l]);
        idx->tables[l] = nullptr;
    }
}

// Runs phase 1 + phase 2; leaves res_pos (and res_d2 when asked) on the device.
struct grid_scratch {
    unsigned int* res_pos = nullptr;
    double* res_d2 = nullptr;
    work_item* work = nullptr;
    unsigned int* work_count = nullptr;
    int64_t nq = 0;
};

static int grid_search_launch(pcr_ctx* ctx, const pcr_index* idx, pcr_pt* q, int64_t nq, const pcr_xform* x, int write_back,
                              double max_d2, bool gated, bool want_d2, grid_scratch* sc) {
    int rc;
    sc->nq = nq;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * nq, (void**)&sc->res_pos))) return rc;
    if (want_d2 && (rc = pcr_dev_alloc(ctx, sizeof(double) * nq, (void**)&sc->res_d2))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(work_item) * nq, (void**)&sc->work))) return rc;
    sc->work_count = ctx->d_counters + 32;  // one word of the zeroed scratch page
    PCR_HIP(ctx, hipMemsetAsync(sc->work_count, 0, sizeof(unsigned int), ctx->stream));
    pcr_xform xi;
    pcr_xform_from_T(nullptr, &xi);
    const int block = 256;
    const long long threads = (long long)nq * P1_G;
    const int grid1 = (int)((threads + block - 1) / block);
    pcr_prof_mark(ctx, 0);
    hipLaunchKernelGGL(grid_phase1_kernel, dim3(grid1), dim3(block), 0, ctx->stream, idx->view, q, (long long)nq, x ? *x : xi, x ? 1 : 0,
                       write_back, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2, sc->work, sc->work_count);
    // phase 2: a fixed grid of waves walks the work list (its length is only known on the device)
    long long waves = nq < 8ll * 4 * ctx->cu_count ? nq : 8ll * 4 * ctx->cu_count;
    int grid2 = (int)((waves + 3) / 4);
    if (grid2 < 1) grid2 = 1;
    pcr_prof_mark(ctx, 1);
    hipLaunchKernelGGL(grid_phase2_kernel, dim3(grid2), dim3(256), 0, ctx->stream, idx->view, (const work_item*)sc->work,
                       (const unsigned int*)sc->work_count, max_d2, gated ? 1 : 0, sc->res_pos, sc->res_d2);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

static void grid_scratch_free(pcr_ctx* ctx, grid_scratch* sc) {
    pcr_dev_free(ctx, sc->res_pos, sizeof(unsigned int) * sc->nq);
    if (sc->res_d2) pcr_dev_free(ctx, sc->res_d2, sizeof(double) * sc->nq);
    pcr_dev_free(ctx, sc->work, sizeof(work_item) * sc->nq);
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
