// Device-side helpers shared by the grid build and search kernels.
#pragma once
#include <cfloat>
#include "pcr_internal.h"

constexpr unsigned int POS_NONE = 0xffffffffu;

struct __attribute__((aligned(16))) work_item {  // a query handed from one search stage to the next
    double ax, ay, az;   // transformed query
    double best_d2;      // best squared distance so far (DBL_MAX = none)
    unsigned int best_pos;
    unsigned int qi;
};

__host__ __device__ static inline unsigned long long spread21(unsigned long long x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

__host__ __device__ static inline unsigned int compact21(unsigned long long x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffffull;
    return (unsigned int)x;
}

// level-0 integer cell coordinate (biased by 2^20, clamped to 21 bits).  *clamped is set
// when the point lies outside the representable range (then only a full descent is exact).
__device__ static inline int cell_coord(double v, double lo, double inv, bool* clamped) {
    const double f = floor((v - lo) * inv);
    if (!(f >= -(double)(PCR_COORD_BIAS)) || !(f <= (double)(PCR_COORD_MAX - PCR_COORD_BIAS))) {  // also NaN
        *clamped = true;
        return f > 0 ? (int)PCR_COORD_MAX : 0;
    }
    return (int)f + (int)PCR_COORD_BIAS;
}

// Hash of a cell's integer coordinates (well mixed: the classic x*p1 ^ y*p2 ^ z*p3 gave probe
// chains of up to 65 slots on a KITTI scan).
__host__ __device__ static inline unsigned int cell_hash(unsigned int x, unsigned int y, unsigned int z) {
    unsigned int h = x * 0x9E3779B1u;
    h = (h << 15) | (h >> 17);
    h ^= y * 0x85EBCA77u;
    h = (h << 13) | (h >> 19);
    h ^= z * 0xC2B2AE3Du;
    h ^= h >> 16;
    h *= 0x7FEB352Du;
    h ^= h >> 15;
    h *= 0x846CA68Bu;
    h ^= h >> 16;
    return h;
}

__device__ static inline unsigned long long cell_pack(unsigned int x, unsigned int y, unsigned int z) {
    return (unsigned long long)x | ((unsigned long long)y << 21) | ((unsigned long long)z << 42);
}

// The table is made of 64-byte buckets of 4 slots, filled from slot 0 (load factor <= 0.25): a
// lookup is one cache line in almost every case, and an empty slot in the bucket proves absence.
// `mask` = number of buckets - 1.
// The bucket is fetched as four 16-byte GLOBAL loads issued together and pinned before the first compare: written
// field by field through the generic pointer of the grid view, the compiler fetched the keys first and the (start, end)
// of the matching slot in a second, dependent round trip (flat loads, each followed by a full wait).
typedef unsigned int pcr_u4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) pcr_u4* pcr_gu4p;
__device__ static inline unsigned long long u4_key(const pcr_u4& v) { return (unsigned long long)v.x | ((unsigned long long)v.y << 32); }

__device__ static inline bool lookup_cell(const pcr_cell_slot* __restrict__ tab, unsigned int mask, unsigned int x, unsigned int y,
                                          unsigned int z, unsigned int* s, unsigned int* e) {
    static_assert(sizeof(pcr_cell_slot) == 16, "one slot = one 16-byte load");
    const unsigned long long key = cell_pack(x, y, z);
    unsigned int b = cell_hash(x, y, z) & mask;
    const pcr_gu4p base = (pcr_gu4p)(const void*)tab;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
        const pcr_gu4p bk = base + (size_t)b * 4;
        pcr_u4 s0 = bk[0], s1 = bk[1], s2 = bk[2], s3 = bk[3];
        asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));   // all four here, now
        if (u4_key(s0) == key) { *s = s0.z; *e = s0.w; return true; }
        if (u4_key(s1) == key) { *s = s1.z; *e = s1.w; return true; }
        if (u4_key(s2) == key) { *s = s2.z; *e = s2.w; return true; }
        if (u4_key(s3) == key) { *s = s3.z; *e = s3.w; return true; }
        if (u4_key(s3) == PCR_EMPTY_KEY) return false;  // buckets fill from slot 0: a free last slot means the bucket never overflowed
        b = (b + 1) & mask;
    }
    return false;
}

// Three cells (x - 1, x, x + 1; same y, z) at once: the twelve loads of their three buckets are in flight together (a thread
// that walks the 27 cells around a point one lookup after the other pays 27 dependent round trips).  found bit i = cell x - 1 + i.
__device__ static inline unsigned int lookup_cell3(const pcr_cell_slot* __restrict__ tab, unsigned int mask, unsigned int x, unsigned int y,
                                                   unsigned int z, unsigned int lim, unsigned int s[3], unsigned int e[3]) {
    const pcr_gu4p base = (pcr_gu4p)(const void*)tab;
    pcr_u4 v[3][4];
    bool in[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const unsigned int xi = x + (unsigned int)i - 1u;   // x = 0: wraps, caught by the range test
        in[i] = xi <= lim;
        const pcr_gu4p bk = base + (size_t)(cell_hash(in[i] ? xi : x, y, z) & mask) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[i][k] = bk[k];
    }
    asm volatile("" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[0][2]), "+v"(v[0][3]));
    asm volatile("" : "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[1][2]), "+v"(v[1][3]));
    asm volatile("" : "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[2][2]), "+v"(v[2][3]));
    unsigned int found = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (!in[i]) continue;
        const unsigned int xi = x + (unsigned int)i - 1u;
        const unsigned long long key = cell_pack(xi, y, z);
        bool hit = false, settled = false;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (!hit && u4_key(v[i][k]) == key) { s[i] = v[i][k].z; e[i] = v[i][k].w; hit = true; }
        settled = hit || u4_key(v[i][3]) == PCR_EMPTY_KEY;   // a free last slot: the bucket never overflowed
        if (!settled) hit = lookup_cell(tab, mask, xi, y, z, &s[i], &e[i]);   // (rare) follow the probe chain
        if (hit) found |= 1u << i;
    }
    return found;
}

// 2x2x2-block lookup: one 32-byte read in almost every case (both halves requested together, see lookup_cell)
__device__ static inline bool lookup_block(const pcr_block_slot* __restrict__ tab, unsigned int mask, unsigned int bx, unsigned int by,
                                           unsigned int bz, pcr_block_slot* out) {
    static_assert(sizeof(pcr_block_slot) == 32, "one slot = two 16-byte loads");
    const unsigned long long key = cell_pack(bx, by, bz);
    unsigned int b = cell_hash(bx, by, bz) & mask;
    const pcr_gu4p base = (pcr_gu4p)(const void*)tab;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
        pcr_u4 lo = base[(size_t)b * 2], hi = base[(size_t)b * 2 + 1];
        asm volatile("" : "+v"(lo), "+v"(hi));
        const unsigned long long k = u4_key(lo);
        if (k == key) {
            pcr_u4 w[2] = {lo, hi};
            __builtin_memcpy(out, w, 32);
            return true;
        }
        if (k == PCR_EMPTY_KEY) return false;
        b = (b + 1) & mask;
    }
    return false;
}

__device__ static inline bool better(double d2, long long id, double bd2, long long bid) {
    return d2 < bd2 || (d2 == bd2 && id < bid);
}

// Exact squared distance, evaluated exactly like the host check:
// (dx*dx + dy*dy) + dz*dz with each operation rounded (no FMA: -ffp-contract=off).
__device__ static inline double dist2(double ax, double ay, double az, const pcr_pt& b) {
    const double dx = ax - b.x, dy = ay - b.y, dz = az - b.z;
    return (dx * dx + dy * dy) + dz * dz;
}

__device__ static inline void xform_apply(const pcr_xform& x, const pcr_pt& p, double* ax, double* ay, double* az) {
    *ax = ((x.r[0] * p.x + x.r[1] * p.y) + x.r[2] * p.z) + x.t[0];
    *ay = ((x.r[3] * p.x + x.r[4] * p.y) + x.r[5] * p.z) + x.t[1];
    *az = ((x.r[6] * p.x + x.r[7] * p.y) + x.r[8] * p.z) + x.t[2];
}

// ---------------------------------------------------------------- wave64 reductions on the DPP network
// row_shr 1 / 2 / 4 / 8, then row_bcast15 / row_bcast31: six VALU instructions of a few cycles each, no LDS traffic (a __shfl_up / __shfl_xor
// chain is six DEPENDENT ds_bpermute round trips).  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ static inline unsigned int dpp_u32(unsigned int identity, unsigned int v) {
    return (unsigned int)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ static inline unsigned int wave_incl_scan_add(unsigned int v) {
    v += dpp_u32<0x111, 0xf>(0u, v);
    v += dpp_u32<0x112, 0xf>(0u, v);
    v += dpp_u32<0x114, 0xf>(0u, v);
    v += dpp_u32<0x118, 0xf>(0u, v);
    v += dpp_u32<0x142, 0xa>(0u, v);   // row_bcast15 into rows 1 and 3
    v += dpp_u32<0x143, 0xc>(0u, v);   // row_bcast31 into rows 2 and 3
    return v;
}
__device__ static inline unsigned int wave_excl_scan_u32(unsigned int v, int lane, unsigned int* total) {
    const unsigned int inc = wave_incl_scan_add(v);
    *total = __builtin_amdgcn_readlane((int)inc, 63);
    return inc - v;
}
// minimum / maximum over the wave in every lane (lanes without a source keep their own value; lane 63 ends with the result)
__device__ static inline int wave_min_i32(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ static inline int wave_max_i32(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ static inline double vmin(double a, double b) {  // plain v_min_f64 (fmin() adds two canonicalising v_max)
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// binary64 minimum over the wave in every lane: ~20 instructions of a few cycles each -- the __shfl_xor butterfly was six DEPENDENT
// ds_bpermute round trips (~0.3 us) in the middle of every step of a descent.  Lanes without a source keep their own value (old = self).
template <int CTRL, int ROW_MASK>
__device__ static inline double dpp_min_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return vmin(v, __hiloint2double(hi, lo));
}
__device__ static inline double wave_min_f64(double v) {
    v = dpp_min_f64<0x111, 0xf>(v);
    v = dpp_min_f64<0x112, 0xf>(v);
    v = dpp_min_f64<0x114, 0xf>(v);
    v = dpp_min_f64<0x118, 0xf>(v);
    v = dpp_min_f64<0x142, 0xa>(v);
    v = dpp_min_f64<0x143, 0xc>(v);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ static inline long long readlane_i64(long long v, int l) {   // l: wave-uniform
    return ((long long)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)v, l);
}

