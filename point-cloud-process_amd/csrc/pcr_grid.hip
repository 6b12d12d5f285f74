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
#include "pcr_grid_dev.h"
#include "pcr_sort.h"

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




// ------------------------------------------------------------ set-up path without fills, copies and table scans
// Morton keys as sorted by the set-up path: only the key bits that vary over the cloud (pcr_morton_end_bit), in 32 bits when
// they fit -- the (key, position) sort moves a third less -- else in 64.  full_key() puts the constant bias bits back.
constexpr unsigned long long MORTON_BIAS3 = 7ull << 60;   // spread21(PCR_COORD_BIAS) on x, y and z
template <typename K>
__device__ static inline unsigned long long full_key(K k) { return (unsigned long long)k | MORTON_BIAS3; }

template <typename K>
__global__ void __launch_bounds__(256)
morton_keys_var_kernel(const pcr_pt* __restrict__ pts, long long n, double lox, double loy, double loz, double inv, unsigned long long mask,
                       K* __restrict__ keys, unsigned int* __restrict__ vals) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pcr_pt p = pts[i];
    bool cl = false;
    const unsigned long long cx = (unsigned long long)cell_coord(p.x, lox, inv, &cl);
    const unsigned long long cy = (unsigned long long)cell_coord(p.y, loy, inv, &cl);
    const unsigned long long cz = (unsigned long long)cell_coord(p.z, loz, inv, &cl);
    keys[i] = (K)((spread21(cx) | (spread21(cy) << 1) | (spread21(cz) << 2)) & mask);
    vals[i] = (unsigned int)i;
}

// Records into sorted order AND, for an index build (levels > 0), the number of cells of every level = run starts of key >> 6l:
// one atomic per block and level; the block that finishes last hands the counts to the host (pinned, device-mapped words) and
// leaves the counters zero for the next build -- no memset, no copy.
template <typename K>
__global__ void __launch_bounds__(256)
gather_count_kernel(const pcr_pt* __restrict__ pts, const unsigned int* __restrict__ perm, const K* __restrict__ keys, long long n, int levels,
                    pcr_pt* __restrict__ out, unsigned int* __restrict__ counts, unsigned int* __restrict__ host_counts) {
    __shared__ unsigned int s_cnt[PCR_MAX_LEVELS];
    __shared__ int s_last;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (levels > 0 && threadIdx.x < PCR_MAX_LEVELS) s_cnt[threadIdx.x] = 0;
    if (i < n) out[i] = pts[perm[i]];
    if (levels <= 0) return;
    __syncthreads();
    const unsigned long long k = i < n ? full_key(keys[i]) : 0ull, kp = (i < n && i > 0) ? full_key(keys[i - 1]) : 0ull;
    for (int l = 0; l < levels; ++l) {
        const bool start = i < n && (i == 0 || (k >> (6 * l)) != (kp >> (6 * l)));
        const unsigned long long b = __ballot(start);
        if (b && (threadIdx.x & 63) == 0) atomicAdd(&s_cnt[l], (unsigned int)__popcll(b));
    }
    __syncthreads();
    // 64 counters per level (block & 63): a million points are 3 900 blocks, and atomics on ONE word are served at ~90 per
    // microsecond (0.41 ms of 0.78 for a 1 M-point index with one counter per level)
    // (no fence: device-scope atomics are performed at the coherence point -- an agent-scope release is an L2 write-back on this
    // chip; the adds are acknowledged (vmcnt) before the same wave takes the ticket)
    if ((int)threadIdx.x < levels && s_cnt[threadIdx.x])
        __hip_atomic_fetch_add(&counts[threadIdx.x * 64 + (blockIdx.x & 63u)], s_cnt[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&counts[PCR_MAX_LEVELS * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;   // ticket
    __syncthreads();
    if (!s_last) return;
    for (int l = threadIdx.x >> 6; l < PCR_MAX_LEVELS; l += 4) {   // wave w sums levels w, w + 4, w + 8
        unsigned int v = __hip_atomic_load(&counts[l * 64 + (threadIdx.x & 63)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        counts[l * 64 + (threadIdx.x & 63)] = 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        if ((threadIdx.x & 63) == 0) host_counts[l] = v;
    }
    if (threadIdx.x == 0) counts[PCR_MAX_LEVELS * 64] = 0;
}

// all tables of an index in one launch: cell slots all-ones, block slots {free key, start = ~0, flags = 0, counts = 0}
__global__ void __launch_bounds__(256)
init_pools_kernel(pcr_cell_slot* __restrict__ cell_pool, unsigned long long n_cells, pcr_block_slot* __restrict__ block_pool, unsigned long long n_blocks) {
    typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    u2* cp = reinterpret_cast<u2*>(cell_pool);
    for (unsigned long long i = t0; i < n_cells; i += stride) cp[i] = u2{~0ull, ~0ull};
    u2* bp = reinterpret_cast<u2*>(block_pool);
    for (unsigned long long i = t0; i < 2 * n_blocks; i += stride) bp[i] = (i & 1) ? u2{0ull, 0ull} : u2{PCR_EMPTY_KEY, 0x00000000ffffffffull};
}



struct pcr_tables {
    pcr_cell_slot* t[PCR_MAX_LEVELS];
    unsigned int mask[PCR_MAX_LEVELS];
};

// buckets of 4 slots, filled from slot 0; mask = number of buckets - 1
__device__ static inline unsigned int slot_find_or_insert(pcr_cell_slot* tab, unsigned int mask, unsigned long long key, unsigned int h) {
    unsigned int b = h & mask;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
        for (unsigned int k = 0; k < 4; ++k) {
            const unsigned int slot = b * 4 + k;
            unsigned long long old = atomicCAS(&tab[slot].key, PCR_EMPTY_KEY, key);
            if (old == PCR_EMPTY_KEY || old == key) return slot;
        }
        b = (b + 1) & mask;
    }
    return 0xffffffffu;  // table full: cannot happen at load factor <= 0.25
}


struct pcr_btables {
    pcr_block_slot* t[PCR_MAX_LEVELS];
    unsigned int mask[PCR_MAX_LEVELS];
    unsigned int cap[PCR_MAX_LEVELS];
};




template <typename K>
__global__ void __launch_bounds__(256)
insert_cells_var_kernel(const K* __restrict__ keys, long long n, int levels, pcr_tables tabs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = full_key(keys[i]);
    const unsigned long long kp = i > 0 ? full_key(keys[i - 1]) : 0ull, kn = i + 1 < n ? full_key(keys[i + 1]) : 0ull;
    for (int l = 0; l < levels; ++l) {
        const unsigned long long ck = k >> (6 * l);
        const bool start = (i == 0) || (ck != (kp >> (6 * l)));
        const bool end = (i + 1 == n) || (ck != (kn >> (6 * l)));
        if (start || end) {
            const unsigned int X = compact21(ck), Y = compact21(ck >> 1), Z = compact21(ck >> 2);
            const unsigned long long pk = (unsigned long long)X | ((unsigned long long)Y << 21) | ((unsigned long long)Z << 42);
            const unsigned int h = slot_find_or_insert(tabs.t[l], tabs.mask[l], pk, cell_hash(X, Y, Z));
            if (h != 0xffffffffu) {
                if (start) tabs.t[l][h].start = (unsigned int)i;
                if (end) tabs.t[l][h].end = (unsigned int)(i + 1);
            }
        }
    }
}

// The thread at the first point of a cell's run looks its (now complete) slot up and registers the cell in its 2x2x2 block:
// work proportional to the cells, coalesced key reads (walking every slot of every table instead -- four fifths of them
// empty -- took 18.5 us at 120 000 points).  Thread 0 also stores the device copy of the view.
template <typename K>
__global__ void __launch_bounds__(256)
insert_blocks_var_kernel(const K* __restrict__ keys, long long n, int levels, pcr_tables tabs, pcr_btables bt, pcr_grid_view v, pcr_grid_view* __restrict__ d_view) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *d_view = v;
    if (i >= n) return;
    const unsigned long long k = full_key(keys[i]);
    const unsigned long long kp = i > 0 ? full_key(keys[i - 1]) : 0ull;
    for (int l = 0; l < levels; ++l) {
        const unsigned long long ck = k >> (6 * l);
        if (i != 0 && ck == (kp >> (6 * l))) break;   // not a run start here: not one on any coarser level either
        const unsigned int X = compact21(ck), Y = compact21(ck >> 1), Z = compact21(ck >> 2);
        unsigned int cs = 0, ce = 0;
        if (!lookup_cell(tabs.t[l], tabs.mask[l], X, Y, Z, &cs, &ce)) continue;
        const unsigned int BX = X >> 1, BY = Y >> 1, BZ = Z >> 1;
        const int child = (int)((X & 1) | ((Y & 1) << 1) | ((Z & 1) << 2));
        const unsigned long long bk = cell_pack(BX, BY, BZ);
        unsigned int b = cell_hash(BX, BY, BZ) & bt.mask[l];
        for (unsigned int probe = 0; probe <= bt.mask[l]; ++probe) {
            const unsigned long long old = atomicCAS(&bt.t[l][b].key, PCR_EMPTY_KEY, bk);
            if (old == PCR_EMPTY_KEY || old == bk) break;
            b = (b + 1) & bt.mask[l];
        }
        const unsigned int cnt = ce - cs;
        if (cnt >= 0xffffu) atomicOr(&bt.t[l][b].flags, 1u);
        bt.t[l][b].cnt[child] = (unsigned short)(cnt >= 0xffffu ? 0xffffu : cnt);
        atomicMin(&bt.t[l][b].start, cs);
    }
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
    PCR_HIP(ctx, pcr_sync(ctx->stream));
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

// Morton keys of a cloud whose coordinates span `ext` metres from its min corner: every cell coordinate is BIAS + k with
// 0 <= k <= ext / cell, so only the low 3 * bits(k_max) key bits differ between points (the bias bit and the zeros below it
// are the same for all): the radix sort only needs those (30 of 63 bits on a KITTI scan: half the passes).
int pcr_morton_end_bit(const double lo[3], const double hi[3], double inv) {
    double kmax = 0.0;
    for (int k = 0; k < 3; ++k) kmax = fmax(kmax, floor((hi[k] - lo[k]) * inv) + 2.0);   // +1 rounding slack, +1 for "count"
    int nb = 1;
    while (nb < PCR_COORD_BITS - 1 && (double)(1ll << nb) <= kmax) ++nb;
    const int end = 3 * nb;
    return end > 63 ? 63 : end;
}

int pcr_cloud_bbox(pcr_ctx* ctx, const pcr_cloud* c, double lo[3], double hi[3]) {
    if (c->has_bbox) {
        for (int k = 0; k < 3; ++k) { lo[k] = c->lo[k]; hi[k] = c->hi[k]; }
        return PCR_OK;
    }
    return pcr_bbox(ctx, c->d, c->n, lo, hi);
}

// Level-0 cell size and number of levels of the grid over a target with bounding box [lo, hi] and n points (cell_in > 0: the
// caller's choice, still clamped).
void pcr_grid_plan(const double lo[3], const double hi[3], long long n, double cell, double* cell_out, int* levels_out) {
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
    *cell_out = cell;
    *levels_out = levels;
}

// Keys of the varying Morton bits -> stable sort of (key, position) -> records in sorted order (+ cell counts of `levels` levels
// in ctx->h_pinned[0..levels), valid after the next stream synchronisation).  The sorted keys stay in *keys_out until sort_scratch_free.
struct sort_scratch {
    void *keys = nullptr, *keys2 = nullptr, *temp = nullptr;
    unsigned int *vals = nullptr, *vals2 = nullptr;
    size_t key_bytes = 0, temp_bytes = 0;
    long long n = 0;
};
static void sort_scratch_free(pcr_ctx* ctx, sort_scratch* sc) {
    if (sc->temp) pcr_dev_free(ctx, sc->temp, sc->temp_bytes);
    if (sc->keys) pcr_dev_free(ctx, sc->keys, sc->key_bytes);
    if (sc->keys2) pcr_dev_free(ctx, sc->keys2, sc->key_bytes);
    if (sc->vals) pcr_dev_free(ctx, sc->vals, sizeof(unsigned int) * sc->n);
    if (sc->vals2) pcr_dev_free(ctx, sc->vals2, sizeof(unsigned int) * sc->n);
    *sc = sort_scratch();
}
template <typename K>
static int morton_sort_records(pcr_ctx* ctx, const pcr_pt* in, long long n, const double lo[3], double inv, int end_bit, int levels, pcr_pt* out,
                               sort_scratch* sc) {
    int rc;
    sc->n = n;
    sc->key_bytes = sizeof(K) * (size_t)n;
    if ((rc = pcr_dev_alloc(ctx, sc->key_bytes, &sc->keys)) || (rc = pcr_dev_alloc(ctx, sc->key_bytes, &sc->keys2)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&sc->vals)) || (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&sc->vals2)))
        return rc;
    K *keys = (K*)sc->keys, *keys2 = (K*)sc->keys2;
    const int grid_n = (int)((n + 255) / 256);
    const unsigned long long mask = end_bit >= 64 ? ~0ull : (1ull << end_bit) - 1ull;
    hipLaunchKernelGGL(morton_keys_var_kernel<K>, dim3(grid_n), dim3(256), 0, ctx->stream, in, n, lo[0], lo[1], lo[2], inv, mask, keys, sc->vals);
    PCR_HIP(ctx, pcr_sort_pairs(nullptr, sc->temp_bytes, keys, keys2, sc->vals, sc->vals2, (size_t)n, (unsigned int)end_bit, ctx->stream));
    if ((rc = pcr_dev_alloc(ctx, sc->temp_bytes, &sc->temp))) return rc;
    PCR_HIP(ctx, pcr_sort_pairs(sc->temp, sc->temp_bytes, keys, keys2, sc->vals, sc->vals2, (size_t)n, (unsigned int)end_bit, ctx->stream));
    unsigned int* h_counts_dev = nullptr;
    if (levels > 0) PCR_HIP(ctx, hipHostGetDevicePointer((void**)&h_counts_dev, ctx->h_pinned, 0));
    hipLaunchKernelGGL(gather_count_kernel<K>, dim3(grid_n), dim3(256), 0, ctx->stream, in, (const unsigned int*)sc->vals2, (const K*)keys2, n, levels, out,
                       ctx->d_cell_counts, h_counts_dev);   // zero between builds (the last block leaves them so)
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

template <typename K>
static int grid_build_tables(pcr_ctx* ctx, pcr_index* idx, const sort_scratch* sc, int levels, const unsigned int* h_counts, const double lo[3], const double hi[3],
                             double cell) {
    const long long n = idx->n;
    int rc;
    // every level's tables in two allocations (cells, 2x2x2 blocks): one launch initialises them all
    size_t cell_slots = 0, block_slots = 0;
    for (int l = 0; l < levels; ++l) {
        unsigned int cap = (unsigned int)next_pow2(h_counts[l] * 4 + 4);   // slots; load factor <= 0.25
        if (cap < 16) cap = 16;
        unsigned int bcap = (unsigned int)next_pow2(h_counts[l] * 2 + 4);  // blocks <= cells: load factor <= 0.5, usually ~0.15
        if (bcap < 16) bcap = 16;
        idx->caps[l] = cap;
        idx->bcaps[l] = bcap;
        cell_slots += cap;
        block_slots += bcap;
    }
    idx->cell_pool_bytes = sizeof(pcr_cell_slot) * cell_slots;
    idx->block_pool_bytes = sizeof(pcr_block_slot) * block_slots;
    if ((rc = pcr_dev_alloc(ctx, idx->cell_pool_bytes, (void**)&idx->cell_pool)) || (rc = pcr_dev_alloc(ctx, idx->block_pool_bytes, (void**)&idx->block_pool)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(pcr_grid_view), (void**)&idx->d_view)))
        return rc;
    pcr_tables tabs;
    pcr_btables bt;
    memset(&tabs, 0, sizeof(tabs));
    memset(&bt, 0, sizeof(bt));
    pcr_grid_view& v = idx->view;
    memset(&v, 0, sizeof(v));
    v.pts = idx->sorted;
    v.n = n;
    v.levels = levels;
    v.cell0 = cell;
    v.inv_cell0 = 1.0 / cell;
    for (int k = 0; k < 3; ++k) {
        v.lo[k] = lo[k];
        v.origin[k] = 0.5 * (lo[k] + hi[k]);
    }
    size_t co = 0, bo = 0;
    for (int l = 0; l < levels; ++l) {
        idx->tables[l] = idx->cell_pool + co;
        idx->btables[l] = idx->block_pool + bo;
        co += idx->caps[l];
        bo += idx->bcaps[l];
        tabs.t[l] = idx->tables[l];
        tabs.mask[l] = idx->caps[l] / 4 - 1;   // buckets of 4 slots
        bt.t[l] = idx->btables[l];
        bt.mask[l] = idx->bcaps[l] - 1;
        bt.cap[l] = idx->bcaps[l];
        v.table[l] = idx->tables[l];
        v.mask[l] = tabs.mask[l];
        v.btable[l] = idx->btables[l];
        v.bmask[l] = bt.mask[l];
    }
    const int grid_n = (int)((n + 255) / 256);
    const unsigned long long words = cell_slots + 2 * block_slots;
    int gi = (int)((words + 256 * 4 - 1) / (256 * 4));
    if (gi > 4 * ctx->cu_count) gi = 4 * ctx->cu_count;
    hipLaunchKernelGGL(init_pools_kernel, dim3(gi < 1 ? 1 : gi), dim3(256), 0, ctx->stream, idx->cell_pool, (unsigned long long)cell_slots, idx->block_pool,
                       (unsigned long long)block_slots);
    hipLaunchKernelGGL(insert_cells_var_kernel<K>, dim3(grid_n), dim3(256), 0, ctx->stream, (const K*)sc->keys2, n, levels, tabs);
    // (the device copy of the view -- the search kernels read the few fields a wave needs through a pointer, by scalar loads,
    // instead of carrying its 400 bytes in kernel-argument SGPRs -- is stored by thread 0 of this launch)
    hipLaunchKernelGGL(insert_blocks_var_kernel<K>, dim3(grid_n), dim3(256), 0, ctx->stream, (const K*)sc->keys2, n, levels, tabs, bt, idx->view, idx->d_view);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// idx->lo/hi must already hold the target's bounding box (pcr_index_build computes it).
// Launches: keys, the sort's, gather + cell counts, [one synchronisation: the counts size the tables], tables' init, cells, blocks.
int pcr_grid_build(pcr_ctx* ctx, const pcr_cloud* tgt, double cell, pcr_index* idx) {
    const long long n = tgt->n;
    int rc;
    double lo[3] = {idx->lo[0], idx->lo[1], idx->lo[2]}, hi[3] = {idx->hi[0], idx->hi[1], idx->hi[2]};
    int levels = 1;
    pcr_grid_plan(lo, hi, n, cell, &cell, &levels);
    idx->cell = cell;
    const double inv = 1.0 / cell;
    const int end_bit = pcr_morton_end_bit(lo, hi, inv);
    if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * n, (void**)&idx->sorted))) return rc;
    sort_scratch sc;
    const bool k32 = end_bit <= 32;
    rc = k32 ? morton_sort_records<unsigned int>(ctx, tgt->d, n, lo, inv, end_bit, levels, idx->sorted, &sc)
             : morton_sort_records<unsigned long long>(ctx, tgt->d, n, lo, inv, end_bit, levels, idx->sorted, &sc);
    if (rc == PCR_OK && pcr_sync(ctx->stream) != hipSuccess) { ctx->last_error = "hipStreamSynchronize (index build)"; rc = PCR_E_HIP; }
    if (rc == PCR_OK) {
        unsigned int h_counts[PCR_MAX_LEVELS];
        memcpy(h_counts, ctx->h_pinned, sizeof(h_counts));
        rc = k32 ? grid_build_tables<unsigned int>(ctx, idx, &sc, levels, h_counts, lo, hi, cell)
                 : grid_build_tables<unsigned long long>(ctx, idx, &sc, levels, h_counts, lo, hi, cell);
    }
    sort_scratch_free(ctx, &sc);   // (stream-ordered with the launches above)
    return rc;
}

int pcr_cloud_morton_sort(pcr_ctx* ctx, pcr_cloud* c, double cell) {
    if (c->morton_sorted || c->n < 2) { c->morton_sorted = true; return PCR_OK; }
    const long long n = c->n;
    double lo[3], hi[3];
    int rc = pcr_cloud_bbox(ctx, c, lo, hi);
    if (rc) return rc;
    const double emax = fmax(hi[0] - lo[0], fmax(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(cell > 0)) cell = emax > 0 ? emax / 1024.0 : 1.0;
    if (cell < emax / 262144.0) cell = emax / 262144.0;
    pcr_pt* d_out = nullptr;
    if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * n, (void**)&d_out))) return rc;
    const int end_bit = pcr_morton_end_bit(lo, hi, 1.0 / cell);
    sort_scratch sc;
    rc = end_bit <= 32 ? morton_sort_records<unsigned int>(ctx, c->d, n, lo, 1.0 / cell, end_bit, 0, d_out, &sc)
                       : morton_sort_records<unsigned long long>(ctx, c->d, n, lo, 1.0 / cell, end_bit, 0, d_out, &sc);
    sort_scratch_free(ctx, &sc);
    if (rc) { pcr_dev_free(ctx, d_out, sizeof(pcr_pt) * n); return rc; }
    pcr_dev_free(ctx, c->d, sizeof(pcr_pt) * n);
    c->d = d_out;
    c->morton_sorted = true;
    return PCR_OK;
}

void pcr_grid_free(pcr_ctx* ctx, pcr_index* idx) {
    if (idx->d_view) pcr_dev_free(ctx, idx->d_view, sizeof(pcr_grid_view));
    idx->d_view = nullptr;
    if (idx->sorted) pcr_dev_free(ctx, idx->sorted, sizeof(pcr_pt) * idx->n);
    idx->sorted = nullptr;
    if (idx->cell_pool) pcr_dev_free(ctx, idx->cell_pool, idx->cell_pool_bytes);
    if (idx->block_pool) pcr_dev_free(ctx, idx->block_pool, idx->block_pool_bytes);
    idx->cell_pool = nullptr;
    idx->block_pool = nullptr;
    for (int l = 0; l < PCR_MAX_LEVELS; ++l) idx->tables[l] = nullptr, idx->btables[l] = nullptr;   // (pointers into the pools)
}
