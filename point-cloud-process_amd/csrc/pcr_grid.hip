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

__global__ void store_view_kernel(pcr_grid_view v, pcr_grid_view* __restrict__ out) { *out = v; }

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
            unsigned int h = slot_find_or_insert(tabs.t[l], tabs.mask[l], pk, cell_hash(X, Y, Z));
            if (h != 0xffffffffu) {
                if (start) tabs.t[l][h].start = (unsigned int)i;
                if (end) tabs.t[l][h].end = (unsigned int)(i + 1);
            }
        }
    }
}

struct pcr_btables {
    pcr_block_slot* t[PCR_MAX_LEVELS];
    unsigned int mask[PCR_MAX_LEVELS];
    unsigned int cap[PCR_MAX_LEVELS];
};

__global__ void init_blocks_kernel(pcr_btables bt, int levels) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int l = 0; l < levels; ++l)
        if (i < bt.cap[l]) {
            pcr_block_slot e;
            e.key = PCR_EMPTY_KEY; e.start = 0xffffffffu; e.flags = 0;
            for (int k = 0; k < 8; ++k) e.cnt[k] = 0;
            bt.t[l][i] = e;
        }
}

// one thread per slot of the cell tables: the cell registers itself in its 2x2x2 block
__global__ void insert_blocks_kernel(pcr_tables tabs, pcr_btables bt, int levels) {
    const unsigned int i = blockIdx.x * blockDim.x + threadIdx.x;
    for (int l = 0; l < levels; ++l) {
        if (i >= (tabs.mask[l] + 1) * 4) continue;
        const pcr_cell_slot c = tabs.t[l][i];
        if (c.key == PCR_EMPTY_KEY) continue;
        const unsigned int X = (unsigned int)(c.key & 0x1fffffull), Y = (unsigned int)((c.key >> 21) & 0x1fffffull), Z = (unsigned int)((c.key >> 42) & 0x1fffffull);
        const unsigned int BX = X >> 1, BY = Y >> 1, BZ = Z >> 1;
        const int child = (int)((X & 1) | ((Y & 1) << 1) | ((Z & 1) << 2));
        const unsigned long long bk = cell_pack(BX, BY, BZ);
        unsigned int b = cell_hash(BX, BY, BZ) & bt.mask[l];
        for (unsigned int probe = 0; probe <= bt.mask[l]; ++probe) {
            const unsigned long long old = atomicCAS(&bt.t[l][b].key, PCR_EMPTY_KEY, bk);
            if (old == PCR_EMPTY_KEY || old == bk) break;
            b = (b + 1) & bt.mask[l];
        }
        const unsigned int cnt = c.end - c.start;
        if (cnt >= 0xffffu) atomicOr(&bt.t[l][b].flags, 1u);
        bt.t[l][b].cnt[child] = (unsigned short)(cnt >= 0xffffu ? 0xffffu : cnt);
        atomicMin(&bt.t[l][b].start, c.start);
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

// idx->lo/hi must already hold the target's bounding box (pcr_index_build computes it).
int pcr_grid_build(pcr_ctx* ctx, const pcr_cloud* tgt, double cell, pcr_index* idx) {
    const long long n = tgt->n;
    const int block = 256;
    const int grid_n = (int)((n + block - 1) / block);
    int rc;
    double lo[3] = {idx->lo[0], idx->lo[1], idx->lo[2]}, hi[3] = {idx->hi[0], idx->hi[1], idx->hi[2]};
    int levels = 1;
    pcr_grid_plan(lo, hi, n, cell, &cell, &levels);
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
    const int end_bit = pcr_morton_end_bit(lo, hi, inv);
    PCR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0, end_bit, ctx->stream));
    void* d_temp = nullptr;
    if ((rc = pcr_dev_alloc(ctx, temp_bytes, &d_temp))) return rc;
    PCR_HIP(ctx, rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0, end_bit, ctx->stream));
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
        unsigned int cap = (unsigned int)next_pow2(h_counts[l] * 4 + 4);  // slots; load factor <= 0.25
        if (cap < 16) cap = 16;
        idx->caps[l] = cap;
        if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_cell_slot) * cap, (void**)&idx->tables[l]))) return rc;
        PCR_HIP(ctx, hipMemsetAsync(idx->tables[l], 0xff, sizeof(pcr_cell_slot) * cap, ctx->stream));
        tabs.t[l] = idx->tables[l];
        tabs.mask[l] = cap / 4 - 1;  // buckets of 4 slots
    }
    hipLaunchKernelGGL(insert_cells_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const unsigned long long*)d_keys2, n, levels, tabs);
    PCR_HIP(ctx, hipGetLastError());
    // ---- 2x2x2-block tables (one line per 8 cells for the tile stage's directory)
    pcr_btables bt;
    memset(&bt, 0, sizeof(bt));
    unsigned int max_cap = 16, max_bcap = 16;
    for (int l = 0; l < levels; ++l) {
        unsigned int bcap = (unsigned int)next_pow2(h_counts[l] * 2 + 4);  // blocks <= cells: load factor <= 0.5, usually ~0.15
        if (bcap < 16) bcap = 16;
        idx->bcaps[l] = bcap;
        if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_block_slot) * bcap, (void**)&idx->btables[l]))) return rc;
        bt.t[l] = idx->btables[l];
        bt.mask[l] = bcap - 1;
        bt.cap[l] = bcap;
        if (bcap > max_bcap) max_bcap = bcap;
        if (idx->caps[l] > max_cap) max_cap = idx->caps[l];
    }
    hipLaunchKernelGGL(init_blocks_kernel, dim3((max_bcap + 255) / 256), dim3(256), 0, ctx->stream, bt, levels);
    hipLaunchKernelGGL(insert_blocks_kernel, dim3((max_cap + 255) / 256), dim3(256), 0, ctx->stream, tabs, bt, levels);
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
        v.mask[l] = idx->caps[l] / 4 - 1;  // buckets of 4 slots
        v.btable[l] = idx->btables[l];
        v.bmask[l] = idx->bcaps[l] - 1;
    }
    // device copy of the view: the search kernels read it through a pointer (scalar loads of the few fields a wave
    // needs) instead of carrying its 400 bytes in kernel-argument SGPRs
    // (written by a one-thread kernel that takes the view as its argument: no host copy, no synchronisation)
    if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_grid_view), (void**)&idx->d_view))) return rc;
    hipLaunchKernelGGL(store_view_kernel, dim3(1), dim3(1), 0, ctx->stream, idx->view, idx->d_view);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int pcr_cloud_morton_sort(pcr_ctx* ctx, pcr_cloud* c, double cell) {
    if (c->morton_sorted || c->n < 2) { c->morton_sorted = true; return PCR_OK; }
    const long long n = c->n;
    const int block = 256;
    const int grid_n = (int)((n + block - 1) / block);
    double lo[3], hi[3];
    int rc = pcr_cloud_bbox(ctx, c, lo, hi);
    if (rc) return rc;
    const double emax = fmax(hi[0] - lo[0], fmax(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(cell > 0)) cell = emax > 0 ? emax / 1024.0 : 1.0;
    if (cell < emax / 262144.0) cell = emax / 262144.0;
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    unsigned int *d_vals = nullptr, *d_vals2 = nullptr;
    pcr_pt* d_out = nullptr;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned long long) * n, (void**)&d_keys))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned long long) * n, (void**)&d_keys2))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_vals))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_vals2))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * n, (void**)&d_out))) return rc;
    hipLaunchKernelGGL(morton_keys_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const pcr_pt*)c->d, n, lo[0], lo[1], lo[2],
                       1.0 / cell, d_keys, d_vals);
    size_t temp_bytes = 0;
    const int end_bit = pcr_morton_end_bit(lo, hi, 1.0 / cell);
    PCR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0, end_bit, ctx->stream));
    void* d_temp = nullptr;
    if ((rc = pcr_dev_alloc(ctx, temp_bytes, &d_temp))) return rc;
    PCR_HIP(ctx, rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_vals, d_vals2, (size_t)n, 0, end_bit, ctx->stream));
    hipLaunchKernelGGL(gather_sorted_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const pcr_pt*)c->d, (const unsigned int*)d_vals2, n,
                       d_out);
    PCR_HIP(ctx, hipGetLastError());
    pcr_dev_free(ctx, d_temp, temp_bytes);
    pcr_dev_free(ctx, d_keys, sizeof(unsigned long long) * n);
    pcr_dev_free(ctx, d_keys2, sizeof(unsigned long long) * n);
    pcr_dev_free(ctx, d_vals, sizeof(unsigned int) * n);
    pcr_dev_free(ctx, d_vals2, sizeof(unsigned int) * n);
    pcr_dev_free(ctx, c->d, sizeof(pcr_pt) * n);
    c->d = d_out;
    c->morton_sorted = true;
    return PCR_OK;
}

void pcr_grid_free(pcr_ctx* ctx, pcr_index* idx) {
    if (idx->d_view) pcr_dev_free(ctx, idx->d_view, sizeof(pcr_grid_view));
    idx->d_view = nullptr;
    pcr_dev_free(ctx, idx->sorted, sizeof(pcr_pt) * idx->n);
    idx->sorted = nullptr;
    for (int l = 0; l < PCR_MAX_LEVELS; ++l) {
        if (idx->tables[l]) pcr_dev_free(ctx, idx->tables[l], sizeof(pcr_cell_slot) * idx->caps[l]);
        idx->tables[l] = nullptr;
        if (idx->btables[l]) pcr_dev_free(ctx, idx->btables[l], sizeof(pcr_block_slot) * idx->bcaps[l]);
        idx->btables[l] = nullptr;
    }
}

