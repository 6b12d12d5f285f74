// Brute-force exact 1-NN as an MFMA-tiled contraction (config 2 of BASELINE.json).
//
// For target b and query a (both taken about the target's bbox centre o):
//     |a-b|^2 = |a'|^2 + ( |b'|^2 - 2 b'.a' )          a' = a-o, b' = b-o
// The bracket is a K=4 contraction  [-2b'x, -2b'y, -2b'z, |b'|^2 + bias] . [a'x, a'y, a'z, 1]
// and |a'|^2 enters through the C operand of the MFMA (its layout is the D layout: one query column per
// lane), so one v_mfma_f64_16x16x4_f64 per 16 targets x 16 queries yields  v = d^2 + bias  directly.
// binary64 is used because the expanded form cancels: KITTI coordinates reach 80 m (|b'|^2 ~ 6e3) while
// neighbour spacing is ~3 cm (d^2 ~ 1e-3); binary32 (ulp(6e3) = 5e-4) cannot rank neighbours there and
// gfx950 has no xf32.
//
// EXACTNESS.  The sweep is only a filter; what it returns is proven, not assumed:
//   * |v - (d^2 + bias)| <= E_q := 2^-48 (|a'_q| + Rt)^2   (Rt = half diagonal of the target box): at most 22
//     roundings of quantities bounded by (|a'| + |b'|)^2; bias = 2^-38 Rt^2 keeps v > 0.
//   * tiles are folded in GROUPS of BR_GRP = 8; a group's minimum carries the group index in its low 16 mantissa bits
//     (|packed - v| < 2^-36 v, relative to d^2 + bias now, not to |a'|^2), and per (query, row group) the
//     smallest AND the second-smallest packed group minimum are kept.
//   * every row of the winning group is re-evaluated in the direct form (dx*dx+dy*dy)+dz*dz, so the reported
//     d^2 is bit-identical to the grid path and ties inside a group resolve to the lowest index.
//   * if second - smallest <= 2 E_q + 2^-34 second, another group may hold the true neighbour (near-tie,
//     duplicate targets, far-away clouds): the query is flagged -- unless that cell cannot beat the exact
//     distance already in hand -- and re-done by brute_exact_kernel, a plain direct-form sweep.
// Generic scans flag nothing (the band is ~1e-11 m^2 at 100 m); lattices and duplicated clouds flag a lot
// and pay the VALU price, but every answer is the exact nearest neighbour, lowest index on ties.
//
// HBM layout: mfma_a[tile][64] doubles -- the A operand of tile `tile` in lane order (lane l holds
// A[i = l&15][k = l>>4]) so one wave loads a tile with a single coalesced 512-B read.  3.84 MB for 120k
// targets: L2/MALL resident and streamed by every wave.
#include <cfloat>
#include <cmath>
#include "pcr_internal.h"

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int BR_NT = 4;          // query tiles (of 16) per wave -> 64 queries per wave
constexpr int BR_WAVES = 4;       // waves per block
constexpr int BR_QPB = BR_NT * 16 * BR_WAVES;  // queries per block = 256
#ifndef PCR_BR_GRP
#define PCR_BR_GRP 8
#endif
constexpr int BR_PF = 4;          // tiles per prefetch block
constexpr int BR_GRP = PCR_BR_GRP;  // tiles per argmin group (multiple of BR_PF): one tracker update per group
constexpr int BR_PAD = BR_GRP + BR_PF;  // never-winning padding tiles behind the last real one (unconditional prefetch)
static_assert(BR_GRP % BR_PF == 0, "argmin group = whole prefetch blocks");
constexpr double BR_BIAS_REL = 3.6379788070917130e-12;   // 2^-38 (x Rt^2)
constexpr double BR_ERR_REL = 3.5527136788005009e-15;    // 2^-48 (x (|a'| + Rt)^2)
constexpr double BR_TRUNC_REL = 5.8207660913467407e-11;  // 2^-34

__global__ void brute_prep_kernel(const pcr_pt* __restrict__ pts, long long n, long long n_tiles, double ox, double oy, double oz,
                                  double bias, double* __restrict__ mfma_a) {
    long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (n_tiles + BR_PAD) * 64) return;
    long long t = g >> 6;
    int l = (int)(g & 63);
    int i = l & 15, k = l >> 4;
    long long j = t * 16 + i;
    double v;
    if (j < n) {
        pcr_pt p = pts[j];
        double bx = p.x - ox, by = p.y - oy, bz = p.z - oz;
        // `bias` keeps every contraction result positive (the expanded form of a coincident pair is off by up to E_q):
        // the epilogue packs the group index into the low mantissa bits and relies on the ordering of positive doubles
        v = (k == 0) ? -2.0 * bx : (k == 1) ? -2.0 * by : (k == 2) ? -2.0 * bz : (((bx * bx + by * by) + bz * bz) + bias);
    } else {
        v = (k == 3) ? 1e300 : 0.0;  // padding rows can never win
    }
    mfma_a[g] = v;
}

__global__ void brute_unpermute_kernel(const pcr_pt* __restrict__ in, long long n, pcr_pt* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pcr_pt p = in[i];
    out[p.id] = p;
}

struct brute_cand {  // per (target split, query): exact best candidate of that split
    double d2;
    double pamb;     // smallest packed minimum among this query's AMBIGUOUS (row group) cells of the split; DBL_MAX = none
    int id;
    int pad;
};

struct brute_res {   // per query, after the splits are merged (and after the exact fallback)
    double d2;
    int id;          // target row, -1 = none
    int pad;
};

// plain v_min_f64 / v_max_f64: this file is compiled with -fno-honor-nans so fmin() does not canonicalise its inputs
// (an inline-asm v_min_f64 is not an option here: hipcc inserts no MFMA->VALU wait states for asm operands).
__device__ static inline double bvmin(double a, double b) { return fmin(a, b); }
__device__ static inline double bvmax(double a, double b) { return fmax(a, b); }

constexpr unsigned int BR_CODE_MASK = 0xffffu;
__device__ static inline double pack_code(double m, unsigned int code) {
    return __hiloint2double(__double2hiint(m), (int)(((unsigned int)__double2loint(m) & ~BR_CODE_MASK) | code));
}

__device__ static inline double dist2_pt(double ax, double ay, double az, const pcr_pt& b) {
    double dx = ax - b.x, dy = ay - b.y, dz = az - b.z;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ static inline void brute_xform(const pcr_xform& x, const pcr_pt& p, double* ax, double* ay, double* az) {
    *ax = ((x.r[0] * p.x + x.r[1] * p.y) + x.r[2] * p.z) + x.t[0];
    *ay = ((x.r[3] * p.x + x.r[4] * p.y) + x.r[5] * p.z) + x.t[1];
    *az = ((x.r[6] * p.x + x.r[7] * p.y) + x.r[8] * p.z) + x.t[2];
}

__global__ void __launch_bounds__(256, 2)
brute_nn_kernel(const double* __restrict__ mfma_a, const pcr_pt* __restrict__ tgt, long long n_tgt, long long n_tiles, int n_splits,
                const pcr_pt* __restrict__ q, long long nq, pcr_xform x, int has_x, double ox, double oy, double oz, double rt,
                double bias, brute_cand* __restrict__ cand /* [n_splits][nq] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long qbase = (long long)blockIdx.x * BR_QPB + wave * (BR_NT * 16);
    const int split = blockIdx.y;
    // this lane's own query (one query per lane, 64 per wave)
    const long long qi = qbase + lane;
    double ax = 0, ay = 0, az = 0;
    if (qi < nq) {
        pcr_pt p = q[qi];
        ax = p.x; ay = p.y; az = p.z;
        if (has_x) brute_xform(x, p, &ax, &ay, &az);
    }
    // B operands: tile tt covers queries 16*tt .. 16*tt+15 of this wave; lane l supplies B[k = l>>4][j = l&15].
    // C operands: |a'_j|^2 of the lane's own output column j = l&15 (the C layout is the D layout).
    const int kk = lane >> 4, jj = lane & 15;
    double bq[BR_NT];
    double qn[BR_NT];
#pragma unroll
    for (int tt = 0; tt < BR_NT; ++tt) {
        int src = tt * 16 + jj;
        double sx = __shfl(ax, src, 64) - ox, sy = __shfl(ay, src, 64) - oy, sz = __shfl(az, src, 64) - oz;
        bq[tt] = (kk == 0) ? sx : (kk == 1) ? sy : (kk == 2) ? sz : 1.0;
        qn[tt] = (sx * sx + sy * sy) + sz * sz;
    }
    double best[BR_NT], sec[BR_NT];
    int bgrp[BR_NT];
#pragma unroll
    for (int tt = 0; tt < BR_NT; ++tt) { best[tt] = DBL_MAX; sec[tt] = DBL_MAX; bgrp[tt] = -1; }

    const long long per = (n_tiles + n_splits - 1) / n_splits;
    const long long t0 = (long long)split * per;
    const long long t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    if (t0 < t1) {
        // Software pipeline.  (1) A operands are fetched BR_PF tiles ahead (one tile is only 4 x 64 MFMA cycles of
        // work, less than an L2 round trip).  (2) The 4 MFMAs of tile t are issued BEFORE the min epilogue of tile
        // t-1, so the VALU work runs in the shadow of the matrix pipe.  The epilogue uses plain v_min_f64 (fmin()
        // would add two canonicalising v_max_f64 per call).  Because of the one-tile lag, argmin group G of a split
        // covers tiles t0 + BR_GRP*G - 1 .. t0 + BR_GRP*G + BR_GRP - 2 (clipped at t0); the last tile forms a group of its own.
        // mfma_a is padded with BR_PAD never-winning tiles, so every load below is unconditional: the compiler can
        // then wait with a counted vmcnt(BR_PF) for the tile it needs instead of draining the prefetch.
        double a_cur[BR_PF], a_nxt[BR_PF];
#pragma unroll
        for (int i = 0; i < BR_PF; ++i) a_cur[i] = mfma_a[(t0 + i) * 64 + lane];
        const v4f64 never = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX};
        v4f64 acc[BR_NT], cq[BR_NT];
#pragma unroll
        for (int tt = 0; tt < BR_NT; ++tt) {
            acc[tt] = never;
            cq[tt] = v4f64{qn[tt], qn[tt], qn[tt], qn[tt]};
        }
        unsigned int grp = 0;
        for (long long tg = t0; tg < t1; tg += BR_GRP, ++grp) {
            double gm[BR_NT];
#pragma unroll
            for (int h = 0; h < BR_GRP / BR_PF; ++h) {
                const long long tb = tg + h * BR_PF;
#pragma unroll
                for (int i = 0; i < BR_PF; ++i) a_nxt[i] = mfma_a[(tb + BR_PF + i) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);  // keep the prefetch up here: hipcc otherwise sinks it next to its first use
#pragma unroll
                for (int i = 0; i < BR_PF; ++i) {
                    // tiles past t1 inside the last group belong to the next split or are padding: harmless to look at
                    v4f64 cur[BR_NT];
#pragma unroll
                    for (int tt = 0; tt < BR_NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], cq[tt], 0, 0, 0);
#pragma unroll
                    for (int tt = 0; tt < BR_NT; ++tt) {
                        const double m = bvmin(bvmin(acc[tt][0], acc[tt][1]), bvmin(acc[tt][2], acc[tt][3]));
                        gm[tt] = (h == 0 && i == 0) ? m : bvmin(gm[tt], m);
                    }
#pragma unroll
                    for (int tt = 0; tt < BR_NT; ++tt) acc[tt] = cur[tt];
                }
#pragma unroll
                for (int i = 0; i < BR_PF; ++i) a_cur[i] = a_nxt[i];
            }
            // one tracker update per group: smallest and second-smallest packed group minimum
#pragma unroll
            for (int tt = 0; tt < BR_NT; ++tt) {
                const double p = pack_code(gm[tt], grp & BR_CODE_MASK);
                sec[tt] = bvmin(sec[tt], bvmax(best[tt], p));
                best[tt] = bvmin(best[tt], p);
            }
        }
#pragma unroll
        for (int tt = 0; tt < BR_NT; ++tt) {  // the last tile is a group of its own; then unpack the winning group
            const double m = bvmin(bvmin(acc[tt][0], acc[tt][1]), bvmin(acc[tt][2], acc[tt][3]));
            const double p = pack_code(m, grp & BR_CODE_MASK);
            sec[tt] = bvmin(sec[tt], bvmax(best[tt], p));
            best[tt] = bvmin(best[tt], p);
            bgrp[tt] = best[tt] < 1e299 ? (int)((unsigned int)__double2loint(best[tt]) & BR_CODE_MASK) : -1;
        }
    }
    // exact re-evaluation of the winning group's rows (BR_GRP tiles x this lane's 4 rows), then merge the 4 row groups
    const int rg = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < BR_NT; ++tt) {
        int src = tt * 16 + jj;
        double qx = __shfl(ax, src, 64), qy = __shfl(ay, src, 64), qz = __shfl(az, src, 64);
        double bd2 = DBL_MAX;
        int bid = 0x7fffffff;
        double pamb = DBL_MAX;
        if (bgrp[tt] >= 0) {
            const long long tl = t0 + (long long)BR_GRP * bgrp[tt] - 1;
            for (int u = 0; u < BR_GRP; ++u) {
                const long long tile = tl + u;
                if (tile < t0) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    long long j = tile * 16 + rg + 4 * r;
                    if (j < n_tgt) {
                        pcr_pt b = tgt[j];
                        double d2 = dist2_pt(qx, qy, qz, b);
                        if (d2 < bd2 || (d2 == bd2 && (int)j < bid)) { bd2 = d2; bid = (int)j; }
                    }
                }
            }
            // could another group of this (query, row group) cell hold the true neighbour?
            const double rq = sqrt(qn[tt]) + rt;
            const double band = 2.0 * BR_ERR_REL * rq * rq + BR_TRUNC_REL * sec[tt];
            if (sec[tt] - best[tt] <= band) pamb = best[tt];
        }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            double od2 = __shfl_xor(bd2, off, 64);
            int oid = __shfl_xor(bid, off, 64);
            double opa = __shfl_xor(pamb, off, 64);
            if (od2 < bd2 || (od2 == bd2 && oid < bid)) { bd2 = od2; bid = oid; }
            pamb = bvmin(pamb, opa);
        }
        long long qo = qbase + tt * 16 + jj;
        if (rg == 0 && qo < nq) {
            brute_cand c;
            c.d2 = bd2;
            c.pamb = pamb;
            c.id = (bd2 == DBL_MAX) ? -1 : bid;
            c.pad = 0;
            cand[(long long)split * nq + qo] = c;
        }
    }
}

// merge the per-split candidates of every query; queries some ambiguous cell could still improve go to the flagged
// list (one atomic per wave).  need_exact_far = 0 (ICP): a cell that cannot beat the gate is irrelevant as well.
__global__ void __launch_bounds__(256)
brute_merge_kernel(const brute_cand* __restrict__ cand, int n_splits, const pcr_pt* __restrict__ q, long long nq, pcr_xform x, int has_x,
                   double ox, double oy, double oz, double rt, double bias, double max_d2, int gate_bounds, brute_res* __restrict__ res,
                   unsigned int* __restrict__ flag_list, unsigned int* __restrict__ flag_count) {
    const long long qi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool flagged = false;
    if (qi < nq) {
        double bd2 = DBL_MAX, pamb = DBL_MAX;
        int bid = -1;
        for (int s = 0; s < n_splits; ++s) {
            const brute_cand c = cand[(long long)s * nq + qi];
            if (c.id >= 0 && (c.d2 < bd2 || (c.d2 == bd2 && c.id < bid))) { bd2 = c.d2; bid = c.id; }
            pamb = bvmin(pamb, c.pamb);
        }
        if (pamb < DBL_MAX) {
            // an ambiguous cell with packed minimum P holds no point closer than P - E - trunc - bias
            const pcr_pt p = q[qi];
            double ax = p.x, ay = p.y, az = p.z;
            if (has_x) brute_xform(x, p, &ax, &ay, &az);
            const double sx = ax - ox, sy = ay - oy, sz = az - oz;
            const double rq = sqrt((sx * sx + sy * sy) + sz * sz) + rt;
            const double lower = pamb - BR_ERR_REL * rq * rq - BR_TRUNC_REL * pamb - bias;
            const double bound = gate_bounds ? fmin(bd2, max_d2) : bd2;
            flagged = lower <= bound;
        }
        brute_res r;
        r.d2 = bd2; r.id = bid; r.pad = 0;
        res[qi] = r;
    }
    const unsigned long long m = __ballot(flagged);
    if (m) {
        const int lane = threadIdx.x & 63;
        unsigned int base = 0;
        if (lane == 0) base = atomicAdd(flag_count, (unsigned int)__popcll(m));
        base = __shfl(base, 0, 64);
        if (flagged) flag_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned int)qi;
    }
}

// Exact fallback: direct-form sweep of ALL targets for the flagged queries, BR_XQ queries per block pass.
constexpr int BR_XQ = 8;
__global__ void __launch_bounds__(256)
brute_exact_kernel(const pcr_pt* __restrict__ tgt, long long n_tgt, const pcr_pt* __restrict__ q, pcr_xform x, int has_x,
                   const unsigned int* __restrict__ flag_list, const unsigned int* __restrict__ flag_count, brute_res* __restrict__ res) {
    __shared__ double s_d2[4][BR_XQ];
    __shared__ int s_id[4][BR_XQ];
    const unsigned int count = *flag_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned int c0 = blockIdx.x * BR_XQ; c0 < count; c0 += gridDim.x * BR_XQ) {
        double ax[BR_XQ], ay[BR_XQ], az[BR_XQ], bd2[BR_XQ];
        int bid[BR_XQ];
#pragma unroll
        for (int u = 0; u < BR_XQ; ++u) {
            const unsigned int c = c0 + u < count ? c0 + u : count - 1;  // a short last chunk repeats its last query
            const pcr_pt p = q[flag_list[c]];
            ax[u] = p.x; ay[u] = p.y; az[u] = p.z;
            if (has_x) brute_xform(x, p, &ax[u], &ay[u], &az[u]);
            bd2[u] = DBL_MAX;
            bid[u] = 0x7fffffff;
        }
        for (long long j = threadIdx.x; j < n_tgt; j += 256) {  // ascending j per thread: strict < keeps the lowest index
            const pcr_pt b = tgt[j];
#pragma unroll
            for (int u = 0; u < BR_XQ; ++u) {
                const double d2 = dist2_pt(ax[u], ay[u], az[u], b);
                if (d2 < bd2[u]) { bd2[u] = d2; bid[u] = (int)j; }
            }
        }
#pragma unroll
        for (int u = 0; u < BR_XQ; ++u) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double od2 = __shfl_xor(bd2[u], off, 64);
                const int oid = __shfl_xor(bid[u], off, 64);
                if (od2 < bd2[u] || (od2 == bd2[u] && oid < bid[u])) { bd2[u] = od2; bid[u] = oid; }
            }
            if (lane == 0) { s_d2[wave][u] = bd2[u]; s_id[wave][u] = bid[u]; }
        }
        __syncthreads();
        if (threadIdx.x < BR_XQ && c0 + threadIdx.x < count) {
            double d = s_d2[0][threadIdx.x];
            int id = s_id[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const double od = s_d2[w][threadIdx.x];
                const int oi = s_id[w][threadIdx.x];
                if (od < d || (od == d && oi < id)) { d = od; id = oi; }
            }
            brute_res r;
            r.d2 = d; r.id = (d == DBL_MAX) ? -1 : id; r.pad = 0;
            res[flag_list[c0 + threadIdx.x]] = r;
        }
        __syncthreads();
    }
}

// MODE 0: write idx/d2 (nn1); MODE 1: fused ICP accumulate (+ in-place transform).  Block 0 re-arms the flag counter.
template <int MODE>
__global__ void __launch_bounds__(256)
brute_final_kernel(const brute_res* __restrict__ res, pcr_pt* __restrict__ q, long long nq, pcr_xform x, int has_x,
                   const pcr_pt* __restrict__ tgt, double max_d2, int gated, int write_back, double ox, double oy, double oz,
                   int* __restrict__ idx_out, double* __restrict__ d2_out, double* __restrict__ partials,
                   unsigned int* __restrict__ flag_count, unsigned int* __restrict__ flag_seen) {
    __shared__ double s_part[4][PCR_NMOM];
    long long qi = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (qi == 0) {
        *flag_seen = *flag_count;  // diagnostics: how many queries took the exact fallback in this pass
        *flag_count = 0;           // next search starts with an empty list (stream-ordered)
    }
    double m[PCR_NMOM];
#pragma unroll
    for (int k = 0; k < PCR_NMOM; ++k) m[k] = 0.0;
    if (qi < nq) {
        const brute_res r = res[qi];
        double bd2 = r.d2;
        int bid = r.id;
        if (bid >= 0 && gated && !(bd2 < max_d2)) bid = -1;
        pcr_pt p = q[qi];
        if (MODE == 0) {
            idx_out[p.id] = bid;
            d2_out[p.id] = (bd2 == DBL_MAX) ? INFINITY : bd2;
        } else {
            double ax = p.x, ay = p.y, az = p.z;
            if (has_x) brute_xform(x, p, &ax, &ay, &az);
            if (write_back) {
                p.x = ax; p.y = ay; p.z = az;
                q[qi] = p;
            }
            if (bid >= 0) {
                pcr_pt b = tgt[bid];
                double a0 = ax - ox, a1 = ay - oy, a2 = az - oz;
                double b0 = b.x - ox, b1 = b.y - oy, b2 = b.z - oz;
                m[0] = 1.0;
                m[1] = a0; m[2] = a1; m[3] = a2;
                m[4] = b0; m[5] = b1; m[6] = b2;
                m[7] = b0 * a0; m[8] = b0 * a1; m[9] = b0 * a2;
                m[10] = b1 * a0; m[11] = b1 * a1; m[12] = b1 * a2;
                m[13] = b2 * a0; m[14] = b2 * a1; m[15] = b2 * a2;
                m[16] = (a0 * a0 + a1 * a1) + a2 * a2;
                m[17] = (b0 * b0 + b1 * b1) + b2 * b2;
                m[18] = bd2;
            }
        }
    }
    if (MODE == 1) {
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
}

__global__ void __launch_bounds__(1024) brute_reduce_partials_kernel(const double* __restrict__ partials, int nblocks,
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
int pcr_brute_build(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_index* idx) {
    const long long n = tgt->n;
    idx->n_tiles = (n + 15) / 16;
    int rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(double) * 64 * (idx->n_tiles + BR_PAD), (void**)&idx->mfma_a))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * n, (void**)&idx->plain))) return rc;
    // row order (record id == position), whatever order the cloud currently has on the device
    hipLaunchKernelGGL(brute_unpermute_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const pcr_pt*)tgt->d, n, idx->plain);
    const long long threads = (idx->n_tiles + BR_PAD) * 64;
    double r2 = 0.0;
    for (int k = 0; k < 3; ++k) r2 += 0.25 * (idx->hi[k] - idx->lo[k]) * (idx->hi[k] - idx->lo[k]);
    idx->brute_rt = sqrt(r2);                                  // half diagonal of the target box
    idx->brute_bias = (r2 > 0 ? r2 : 1.0) * BR_BIAS_REL;       // Rt^2 * 2^-38 > E_q of any query that can come within 3 Rt
    const double bias = idx->brute_bias;
    hipLaunchKernelGGL(brute_prep_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, (const pcr_pt*)idx->plain, n,
                       (long long)idx->n_tiles, idx->view.origin[0], idx->view.origin[1], idx->view.origin[2], bias, idx->mfma_a);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

void pcr_brute_free(pcr_ctx* ctx, pcr_index* idx) {
    pcr_dev_free(ctx, idx->mfma_a, sizeof(double) * 64 * (idx->n_tiles + BR_PAD));
    pcr_dev_free(ctx, idx->plain, sizeof(pcr_pt) * idx->n);
    idx->mfma_a = nullptr;
    idx->plain = nullptr;
}

static int brute_splits(pcr_ctx* ctx, long long nq, long long n_tiles) {
    long long qblocks = (nq + BR_QPB - 1) / BR_QPB;
    long long want = 8ll * ctx->cu_count;  // >= 8 blocks of 4 waves per CU in the grid
    long long s = (want + qblocks - 1) / qblocks;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    if (s > n_tiles) s = n_tiles;
    const long long min_s = (n_tiles + 65535) / 65536;  // the group index inside a split must fit the 16 packed bits
    if (s < min_s) s = min_s;
    return (int)s;
}

struct brute_scratch {
    brute_cand* cand = nullptr;
    brute_res* res = nullptr;
    unsigned int* flag_list = nullptr;
    int splits = 0;
    int64_t nq = 0;
};

static void brute_scratch_free(pcr_ctx* ctx, brute_scratch* sc) {
    if (sc->cand) pcr_dev_free(ctx, sc->cand, sizeof(brute_cand) * (size_t)sc->splits * sc->nq);
    if (sc->res) pcr_dev_free(ctx, sc->res, sizeof(brute_res) * (size_t)sc->nq);
    if (sc->flag_list) pcr_dev_free(ctx, sc->flag_list, sizeof(unsigned int) * (size_t)sc->nq);
    sc->cand = nullptr; sc->res = nullptr; sc->flag_list = nullptr;
}

constexpr int BR_FLAG_COUNT_WORD = 120, BR_FLAG_SEEN_WORD = 121;  // words of ctx->d_counters

// sweep -> merge -> exact fallback; leaves the per-query exact result in sc->res (query order)
static int brute_search(pcr_ctx* ctx, const pcr_index* idx, const pcr_pt* q, int64_t nq, const pcr_xform* x, double max_d2, bool gate_bounds,
                        brute_scratch* sc) {
    sc->splits = brute_splits(ctx, nq, idx->n_tiles);
    sc->nq = nq;
    int rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(brute_cand) * (size_t)sc->splits * nq, (void**)&sc->cand)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(brute_res) * (size_t)nq, (void**)&sc->res)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * (size_t)nq, (void**)&sc->flag_list))) {
        brute_scratch_free(ctx, sc);
        return rc;
    }
    pcr_xform xi;
    pcr_xform_from_T(nullptr, &xi);
    const pcr_xform xx = x ? *x : xi;
    const int has_x = x ? 1 : 0;
    const double ox = idx->view.origin[0], oy = idx->view.origin[1], oz = idx->view.origin[2];
    unsigned int* flag_count = ctx->d_counters + BR_FLAG_COUNT_WORD;  // zero at context creation, re-armed by brute_final_kernel
    dim3 grid((unsigned)((nq + BR_QPB - 1) / BR_QPB), (unsigned)sc->splits);
    pcr_prof_mark(ctx, 0);
    hipLaunchKernelGGL(brute_nn_kernel, grid, dim3(256), 0, ctx->stream, (const double*)idx->mfma_a, (const pcr_pt*)idx->plain,
                       (long long)idx->n, (long long)idx->n_tiles, sc->splits, q, (long long)nq, xx, has_x, ox, oy, oz, idx->brute_rt,
                       idx->brute_bias, sc->cand);
    pcr_prof_mark(ctx, 1);
    const int g1 = (int)((nq + 255) / 256);
    hipLaunchKernelGGL(brute_merge_kernel, dim3(g1), dim3(256), 0, ctx->stream, (const brute_cand*)sc->cand, sc->splits, q, (long long)nq, xx,
                       has_x, ox, oy, oz, idx->brute_rt, idx->brute_bias, max_d2, gate_bounds ? 1 : 0, sc->res, sc->flag_list, flag_count);
    // fixed grid; blocks beyond the (device-side) count leave at once
    const long long want = (nq + BR_XQ - 1) / BR_XQ;
    const int g2 = (int)(want < 4ll * ctx->cu_count ? (want < 1 ? 1 : want) : 4ll * ctx->cu_count);
    hipLaunchKernelGGL(brute_exact_kernel, dim3(g2), dim3(256), 0, ctx->stream, (const pcr_pt*)idx->plain, (long long)idx->n, q, xx, has_x,
                       (const unsigned int*)sc->flag_list, (const unsigned int*)flag_count, sc->res);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int pcr_brute_nn1(pcr_ctx* ctx, const pcr_index* idx, const pcr_pt* q, int64_t nq, const pcr_xform* x, double max_d2,
                  int32_t* d_idx, double* d_d2) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    brute_scratch sc;
    // the nn1 API reports the exact neighbour distance of gated-out queries as well: the gate does not bound the search
    int rc = brute_search(ctx, idx, q, nq, x, max_d2, false, &sc);
    if (rc) return rc;
    pcr_xform xi;
    pcr_xform_from_T(nullptr, &xi);
    const int grid = (int)((nq + 255) / 256);
    hipLaunchKernelGGL(brute_final_kernel<0>, dim3(grid), dim3(256), 0, ctx->stream, (const brute_res*)sc.res, (pcr_pt*)q, (long long)nq,
                       x ? *x : xi, x ? 1 : 0, (const pcr_pt*)idx->plain, max_d2, gated ? 1 : 0, 0, idx->view.origin[0], idx->view.origin[1],
                       idx->view.origin[2], d_idx, d_d2, (double*)nullptr, ctx->d_counters + BR_FLAG_COUNT_WORD,
                       ctx->d_counters + BR_FLAG_SEEN_WORD);
    PCR_HIP(ctx, hipGetLastError());
    brute_scratch_free(ctx, &sc);
    return PCR_OK;
}

int pcr_brute_icp_pass(pcr_ctx* ctx, const pcr_index* idx, pcr_pt* q, int64_t nq, const pcr_xform* x, double max_d2, int write_back,
                       double* d_moments) {
    const bool gated = (max_d2 > 0) && std::isfinite(max_d2);
    brute_scratch sc;
    int rc = brute_search(ctx, idx, q, nq, x, max_d2, gated, &sc);
    if (rc) return rc;
    const int grid = (int)((nq + 255) / 256);
    if ((rc = pcr_ensure_scratch(ctx, sizeof(double) * PCR_NMOM * (size_t)grid))) {
        brute_scratch_free(ctx, &sc);
        return rc;
    }
    pcr_prof_mark(ctx, 2);
    hipLaunchKernelGGL(brute_final_kernel<1>, dim3(grid), dim3(256), 0, ctx->stream, (const brute_res*)sc.res, q, (long long)nq, *x, 1,
                       (const pcr_pt*)idx->plain, max_d2, gated ? 1 : 0, write_back, idx->view.origin[0], idx->view.origin[1],
                       idx->view.origin[2], (int*)nullptr, (double*)nullptr, ctx->d_partials, ctx->d_counters + BR_FLAG_COUNT_WORD,
                       ctx->d_counters + BR_FLAG_SEEN_WORD);
    pcr_prof_mark(ctx, 3);
    hipLaunchKernelGGL(brute_reduce_partials_kernel, dim3(1), dim3(1024), 0, ctx->stream, (const double*)ctx->d_partials, grid, d_moments);
    pcr_prof_mark(ctx, 4);
    PCR_HIP(ctx, hipGetLastError());
    pcr_prof_finish(ctx);
    brute_scratch_free(ctx, &sc);
    return PCR_OK;
}

// diagnostics (tests, bench): queries the last brute-force search sent to the exact fallback
int pcr_brute_last_fallback(pcr_ctx* ctx, unsigned int* out) {
    return pcr_d2h_small(ctx, out, ctx->d_counters + BR_FLAG_SEEN_WORD, sizeof(unsigned int));
}
