// Fused batch of registrations: ONE launch per stage for all pairs of a sub-batch
// (the loop over scan pairs of Registration/main.py:190-216; per pair it does what
// pcr_cloud_upload_f32 x 2 -> pcr_index_build -> pcr_icp do, main.py:105-156).
//
//   host     every cloud's xyz columns are packed into one pinned buffer (12 B per point) while its bounding box is taken;
//            grid parameters (cell, levels, Morton bits, fixed-point scale) per pair: the same functions the per-pair path uses
//   H2D      one copy of the packed coordinates + one of the descriptors
//   keys     (cloud id << mbits | Morton key of the cloud's own curve) for every point of every cloud: sources first, then
//            targets, every cloud in a slot of whole 256-record blocks
//   sort     ONE rocPRIM radix sort over mbits + log2(clouds) bits
//   gather   sorted 32-B records {x, y, z (binary64), caller's row}
//   grids    run starts of every level counted per (target, level) -> table capacities and pool offsets planned ON THE DEVICE
//            (no size read-back) -> tables initialised -> cells and 2x2x2 blocks inserted, all targets per launch
//   ICP      pcr_grid_batch_pass: tiles -> queue -> one Procrustes wave per pair, until every pair has stopped
//   D2H      the loop states
// About 15 launches + 3 x iterations per SUB-BATCH instead of ~46 launches and 3 synchronisations per PAIR.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <chrono>
#include <thread>
#include <sched.h>
#include <pthread.h>
#include <cctype>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "pcr_internal.h"
#include "pcr_grid_dev.h"
#include "pcr_icp_step.h"
#include "pcr_sort.h"

constexpr int SLOT = 256;                 // records per block of the set-up kernels; cloud slots are whole blocks
constexpr unsigned int MIN_CAP = 256;     // smallest table of the pools: every block of 256 pool slots belongs to one table
constexpr unsigned long long MORTON_BIAS3 = 7ull << 60;   // spread21(PCR_COORD_BIAS) on x, y and z

struct batch_cloud {          // one per cloud of a sub-batch: sources [0, m), targets [m, 2m)
    unsigned long long off;   // first record slot (multiple of SLOT)
    long long n;              // points that take part in the fused stages (0: the pair is left to the per-pair path)
    long long n_copy;         // points the key kernel brings over to the device (>= n: a scan later sub-batches share is brought over
                              // even when its own pair is not taken)
    double lo[3];             // origin of the cloud's Morton curve (its bounding box's min corner)
    double inv;               // 1 / cell of the curve
    const float* src;         // the cloud's packed coordinates (12 B per point) as the key kernel reads them: this sub-batch's mapped
                              // staging block (over PCIe), or the device copy an earlier sub-batch of the call made of the same scan
};
struct batch_target {         // per pair: what the plan kernel needs to write the pair's descriptor
    double cell, hi[3];
    int levels;
    int pad;
    double scale, inv_scale;
};
struct batch_plan {           // device: totals and prefix offsets written by the plan kernel
    unsigned long long used_cells, used_blocks;   // slots of the two pools in use
    unsigned int overflow;
    unsigned int pad;
};

__device__ inline unsigned int next_pow2_dev(unsigned int v) {
    return v <= 1 ? 1u : 1u << (32 - __clz((int)(v - 1)));
}

// cloud of a block of SLOT records: clouds' slots are whole blocks, so the answer is uniform over the block
__device__ inline int cloud_of_block(const batch_cloud* __restrict__ cl, int n_clouds, unsigned long long first) {
    __shared__ int s_c;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_clouds - 1;   // last cloud with off <= first
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cl[mid].off <= first) lo = mid;
            else hi = mid - 1;
        }
        s_c = lo;
    }
    __syncthreads();
    return s_c;
}

// Every cloud's packed coordinates are read through its own pointer: pinned, device-mapped host memory, read over PCIe by this
// kernel (no copy engine: copies of several contexts queue behind each other in the runtime, and sometimes for milliseconds), or
// -- a scan that an earlier sub-batch of the call already brought over (Registration/reg_result.txt: 342 pairs over 504 scans) --
// that sub-batch's device copy.  `xyz_dev` receives this sub-batch's device copy, which the gather reads later.
__global__ void __launch_bounds__(SLOT)
batch_keys_kernel(float* xyz_dev, const batch_cloud* __restrict__ cl, int n_clouds, int mbits, unsigned int n_blocks,
                  unsigned long long* __restrict__ keys, unsigned int* __restrict__ vals) {
    // A FEW blocks stride over the slots: the kernel is bound by PCIe (56 GB/s: a few thousand loads in flight saturate it), and
    // launched one block per 256 records its waiting waves fill every wave slot of the chip -- the sort, grid and pass kernels of the
    // other sub-batches' streams, which should run underneath these reads, then queue behind them.
    const unsigned long long mask = (1ull << mbits) - 1ull;
    for (unsigned int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const unsigned long long i = (unsigned long long)blk * SLOT + threadIdx.x;
        const int c = cloud_of_block(cl, n_clouds, (unsigned long long)blk * SLOT);
        const batch_cloud C = cl[c];
        unsigned long long k = mask;   // padding of the slot: behind every point of its cloud (the sort is stable)
        if (i - C.off < (unsigned long long)C.n_copy) {
            const float* const p = C.src + 3 * (i - C.off);
            const float x = p[0], y = p[1], z = p[2];
            if (xyz_dev + 3 * i != p) { xyz_dev[3 * i] = x; xyz_dev[3 * i + 1] = y; xyz_dev[3 * i + 2] = z; }
            if (i - C.off < (unsigned long long)C.n) {
            bool clamped = false;
            const unsigned long long cx = (unsigned long long)cell_coord((double)x, C.lo[0], C.inv, &clamped);
            const unsigned long long cy = (unsigned long long)cell_coord((double)y, C.lo[1], C.inv, &clamped);
            const unsigned long long cz = (unsigned long long)cell_coord((double)z, C.lo[2], C.inv, &clamped);
            k = (spread21(cx) | (spread21(cy) << 1) | (spread21(cz) << 2)) & mask;
            }
        }
        keys[i] = ((unsigned long long)c << mbits) | k;
        vals[i] = (unsigned int)i;
        __syncthreads();   // (cloud_of_block's shared word is reused by the next round)
    }
}

__global__ void __launch_bounds__(SLOT)
batch_gather_kernel(const float* __restrict__ xyz, const unsigned int* __restrict__ perm, const batch_cloud* __restrict__ cl, int n_clouds,
                    pcr_pt* __restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * SLOT + threadIdx.x;
    const int c = cloud_of_block(cl, n_clouds, (unsigned long long)blockIdx.x * SLOT);
    const unsigned long long off = cl[c].off;
    pcr_pt o;
    o.x = o.y = o.z = 0.0; o.id = 0;
    if (i - off < (unsigned long long)cl[c].n) {
        const unsigned int src = perm[i];
        const float* p = xyz + 3 * (unsigned long long)src;
        o.x = (double)p[0];
        o.y = (double)p[1];
        o.z = (double)p[2];
        o.id = (long long)(src - off);   // the caller's row
    }
    out[i] = o;
}

// 8-byte words between device memory and mapped host memory (descriptors in, loop states and flags out), zeros behind `n_copy`
__global__ void __launch_bounds__(256)
batch_words_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, unsigned long long n_copy, unsigned long long n_total) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_total; i += stride) dst[i] = i < n_copy ? src[i] : 0ull;
}
__global__ void batch_flag_kernel(const unsigned int* __restrict__ running, unsigned int* __restrict__ host_word) { *host_word = *running; }

// run starts of every level, per target: counts[t][l] (one atomic per wave and level)
__global__ void __launch_bounds__(SLOT)
batch_count_kernel(const unsigned long long* __restrict__ keys, const batch_cloud* __restrict__ cl, int n_clouds, int m, const batch_target* __restrict__ tg,
                   unsigned long long first_slot, int mbits, unsigned int* __restrict__ counts) {
    const unsigned long long i = first_slot + (unsigned long long)blockIdx.x * SLOT + threadIdx.x;
    const int c = cloud_of_block(cl, n_clouds, first_slot + (unsigned long long)blockIdx.x * SLOT);
    const int t = c - m;
    const unsigned long long li = i - cl[c].off;
    const bool valid = li < (unsigned long long)cl[c].n;
    const unsigned long long mask = (1ull << mbits) - 1ull;
    const unsigned long long k = valid ? (keys[i] & mask) : 0ull, kp = (valid && li > 0) ? (keys[i - 1] & mask) : 0ull;
    const int levels = tg[t].levels;
    for (int l = 0; l < levels; ++l) {
        const bool start = valid && (li == 0 || (k >> (6 * l)) != (kp >> (6 * l)));
        const unsigned long long b = __ballot(start);
        if (b && (threadIdx.x & 63) == 0) atomicAdd(&counts[t * PCR_MAX_LEVELS + l], (unsigned int)__popcll(b));
    }
}

// ONE block: capacities of every (target, level) table from the counts, exclusive prefix over all of them = offsets into the two
// pools, the pairs' descriptors (grid view, source slot, fixed-point scale), the tile -> pair map's inputs.
__global__ void __launch_bounds__(256)
batch_plan_kernel(const unsigned int* __restrict__ counts, const batch_cloud* __restrict__ cl, const batch_target* __restrict__ tg, int m,
                  const pcr_pt* __restrict__ pts, pcr_cell_slot* __restrict__ cell_pool, unsigned long long cell_pool_slots,
                  pcr_block_slot* __restrict__ block_pool, unsigned long long block_pool_slots, pcr_batch_pair* __restrict__ pairs,
                  unsigned long long* __restrict__ cell_off, batch_plan* __restrict__ plan) {
    __shared__ unsigned long long s_c[256], s_b[256];
    const int n_e = m * PCR_MAX_LEVELS;
    unsigned long long run_c = 0, run_b = 0;   // pool slots before this chunk of 256 entries
    for (int base = 0; base < n_e; base += 256) {
        const int e = base + (int)threadIdx.x;
        unsigned int cap = 0, bcap = 0;
        if (e < n_e) {
            const int t = e / PCR_MAX_LEVELS, l = e % PCR_MAX_LEVELS;
            if (l < tg[t].levels) {
                const unsigned int cnt = counts[e];
                cap = next_pow2_dev(cnt * 4 + 4);    // load factor <= 0.25 (pcr_grid_build)
                if (cap < MIN_CAP) cap = MIN_CAP;
                bcap = next_pow2_dev(cnt * 2 + 4);
                if (bcap < MIN_CAP) bcap = MIN_CAP;
            }
        }
        s_c[threadIdx.x] = cap;
        s_b[threadIdx.x] = bcap;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {   // inclusive scan
            unsigned long long a = 0, b = 0;
            if ((int)threadIdx.x >= d) { a = s_c[threadIdx.x - d]; b = s_b[threadIdx.x - d]; }
            __syncthreads();
            s_c[threadIdx.x] += a;
            s_b[threadIdx.x] += b;
            __syncthreads();
        }
        const unsigned long long oc = run_c + s_c[threadIdx.x] - cap, ob = run_b + s_b[threadIdx.x] - bcap;
        if (e < n_e) {
            const int t = e / PCR_MAX_LEVELS, l = e % PCR_MAX_LEVELS;
            cell_off[e] = oc;
            pcr_grid_view* gv = &pairs[t].gv;
            gv->table[l] = cap ? cell_pool + oc : nullptr;
            gv->mask[l] = cap ? cap / 4 - 1 : 0;
            gv->btable[l] = bcap ? block_pool + ob : nullptr;
            gv->bmask[l] = bcap ? bcap - 1 : 0;
        }
        run_c += s_c[255];
        run_b += s_b[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        cell_off[n_e] = run_c;
        plan->used_cells = run_c;
        plan->used_blocks = run_b;
        plan->overflow = (run_c > cell_pool_slots || run_b > block_pool_slots) ? 1u : 0u;
    }
    for (int t = threadIdx.x; t < m; t += 256) {
        const batch_cloud S = cl[t], T = cl[m + t];
        pcr_grid_view* gv = &pairs[t].gv;
        gv->pts = pts + T.off;
        gv->n = T.n;
        gv->levels = tg[t].levels;
        gv->cell0 = tg[t].cell;
        gv->inv_cell0 = T.inv;
        for (int k = 0; k < 3; ++k) {
            gv->lo[k] = T.lo[k];
            gv->origin[k] = 0.5 * (T.lo[k] + tg[t].hi[k]);
        }
        pairs[t].q_off = S.off;
        pairs[t].nq = S.n;
        pairs[t].scale = tg[t].scale;
        pairs[t].inv_scale = tg[t].inv_scale;
    }
}

__global__ void __launch_bounds__(256)
batch_init_tables_kernel(pcr_cell_slot* __restrict__ cell_pool, pcr_block_slot* __restrict__ block_pool, const batch_plan* __restrict__ plan,
                         const pcr_batch_pair* __restrict__ pairs, int m, unsigned int* __restrict__ tile_pair, unsigned int n_tiles) {
    if (plan->overflow) return;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long nc = plan->used_cells, nb = plan->used_blocks;
    typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
    u2* cp = reinterpret_cast<u2*>(cell_pool);
    for (unsigned long long i = t0; i < nc; i += stride) cp[i] = u2{~0ull, ~0ull};
    u2* bp = reinterpret_cast<u2*>(block_pool);
    for (unsigned long long i = t0; i < 2 * nb; i += stride)
        bp[i] = (i & 1) ? u2{0ull, 0ull} : u2{PCR_EMPTY_KEY, 0x00000000ffffffffull};   // key | start = ~0, flags = 0 | cnt[8] = 0
    // tile -> pair: the sources' slots are whole tiles, in pair order
    for (unsigned long long tl = t0; tl < n_tiles; tl += stride) {
        const unsigned long long rec = tl * 32ull;
        int lo = 0, hi = m - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pairs[mid].q_off <= rec) lo = mid;
            else hi = mid - 1;
        }
        tile_pair[tl] = (unsigned int)lo;
    }
}

__device__ inline unsigned int batch_slot_find_or_insert(pcr_cell_slot* tab, unsigned int mask, unsigned long long key, unsigned int h) {
    unsigned int b = h & mask;
    for (unsigned int probe = 0; probe <= mask; ++probe) {
        for (unsigned int k = 0; k < 4; ++k) {
            const unsigned int slot = b * 4 + k;
            const unsigned long long old = atomicCAS(&tab[slot].key, PCR_EMPTY_KEY, key);
            if (old == PCR_EMPTY_KEY || old == key) return slot;
        }
        b = (b + 1) & mask;
    }
    return 0xffffffffu;
}

__global__ void __launch_bounds__(SLOT)
batch_insert_cells_kernel(const unsigned long long* __restrict__ keys, const batch_cloud* __restrict__ cl, int n_clouds, int m,
                          const pcr_batch_pair* __restrict__ pairs, unsigned long long first_slot, int mbits, const batch_plan* __restrict__ plan) {
    if (plan->overflow) return;
    const unsigned long long i = first_slot + (unsigned long long)blockIdx.x * SLOT + threadIdx.x;
    const int c = cloud_of_block(cl, n_clouds, first_slot + (unsigned long long)blockIdx.x * SLOT);
    const int t = c - m;
    const unsigned long long li = i - cl[c].off, n = (unsigned long long)cl[c].n;
    if (li >= n) return;
    const unsigned long long mask = (1ull << mbits) - 1ull;
    const unsigned long long k = (keys[i] & mask) | MORTON_BIAS3;
    const unsigned long long kp = li > 0 ? (keys[i - 1] & mask) | MORTON_BIAS3 : 0ull;
    const unsigned long long kn = li + 1 < n ? (keys[i + 1] & mask) | MORTON_BIAS3 : 0ull;
    const pcr_grid_view* gv = &pairs[t].gv;
    const int levels = gv->levels;
    for (int l = 0; l < levels; ++l) {
        const unsigned long long ck = k >> (6 * l);
        const bool start = (li == 0) || (ck != (kp >> (6 * l)));
        const bool end = (li + 1 == n) || (ck != (kn >> (6 * l)));
        if (start || end) {
            const unsigned int X = compact21(ck), Y = compact21(ck >> 1), Z = compact21(ck >> 2);
            pcr_cell_slot* tab = const_cast<pcr_cell_slot*>(gv->table[l]);
            const unsigned int h = batch_slot_find_or_insert(tab, gv->mask[l], cell_pack(X, Y, Z), cell_hash(X, Y, Z));
            if (h != 0xffffffffu) {
                if (start) tab[h].start = (unsigned int)li;
                if (end) tab[h].end = (unsigned int)(li + 1);
            }
        }
    }
}

// The thread at the first point of a cell's run looks its (now complete) slot up and registers the cell in its 2x2x2 block: work
// proportional to the cells, coalesced key reads.  (One thread per slot of the cell POOL -- four fifths of them empty, a binary
// search per block of 256 to find the slot's table -- took 0.62 of the 5.1 ms of a 256-pair batch.)
__global__ void __launch_bounds__(SLOT)
batch_insert_blocks_kernel(const unsigned long long* __restrict__ keys, const batch_cloud* __restrict__ cl, int n_clouds, int m,
                           const pcr_batch_pair* __restrict__ pairs, unsigned long long first_slot, int mbits, const batch_plan* __restrict__ plan) {
    if (plan->overflow) return;
    const unsigned long long i = first_slot + (unsigned long long)blockIdx.x * SLOT + threadIdx.x;
    const int c = cloud_of_block(cl, n_clouds, first_slot + (unsigned long long)blockIdx.x * SLOT);
    const int t = c - m;
    const unsigned long long li = i - cl[c].off, n = (unsigned long long)cl[c].n;
    if (li >= n) return;
    const unsigned long long mask = (1ull << mbits) - 1ull;
    const unsigned long long k = (keys[i] & mask) | MORTON_BIAS3;
    const unsigned long long kp = li > 0 ? (keys[i - 1] & mask) | MORTON_BIAS3 : 0ull;
    const pcr_grid_view* gv = &pairs[t].gv;
    const int levels = gv->levels;
    for (int l = 0; l < levels; ++l) {
        const unsigned long long ck = k >> (6 * l);
        if (li != 0 && ck == (kp >> (6 * l))) break;   // not a run start here: not one on any coarser level either
        const unsigned int X = compact21(ck), Y = compact21(ck >> 1), Z = compact21(ck >> 2);
        unsigned int cs = 0, ce = 0;
        if (!lookup_cell(gv->table[l], gv->mask[l], X, Y, Z, &cs, &ce)) continue;
        pcr_block_slot* bt = const_cast<pcr_block_slot*>(gv->btable[l]);
        const unsigned int bmask = gv->bmask[l];
        const unsigned int BX = X >> 1, BY = Y >> 1, BZ = Z >> 1;
        const int child = (int)((X & 1) | ((Y & 1) << 1) | ((Z & 1) << 2));
        const unsigned long long bk = cell_pack(BX, BY, BZ);
        unsigned int b = cell_hash(BX, BY, BZ) & bmask;
        for (unsigned int probe = 0; probe <= bmask; ++probe) {
            const unsigned long long old = atomicCAS(&bt[b].key, PCR_EMPTY_KEY, bk);
            if (old == PCR_EMPTY_KEY || old == bk) break;
            b = (b + 1) & bmask;
        }
        const unsigned int cnt = ce - cs;
        if (cnt >= 0xffffu) atomicOr(&bt[b].flags, 1u);
        bt[b].cnt[child] = (unsigned short)(cnt >= 0xffffu ? 0xffffu : cnt);
        atomicMin(&bt[b].start, cs);
    }
}

namespace {

unsigned int next_pow2_host(unsigned long long v) {
    unsigned long long p = 1;
    while (p < v) p <<= 1;
    return (unsigned int)p;
}

using dev_block = pcr_dev_block;

// One scan of the call's table as the sub-batches see it: the sub-batch that uses it FIRST (its owner: fixed before any thread
// starts, so that a sub-batch only ever waits for earlier ones) packs it, takes its box and brings it over; later sub-batches read
// the owner's device copy instead of packing and uploading the scan again.
struct cloud_entry {
    std::atomic<int> box_ready{0};    // lo / hi / bad are valid
    std::atomic<int> resident{0};     // dev / ev are valid: the owner's key kernel has been enqueued
    float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool bad = false;                 // non-finite coordinates
    const float* dev = nullptr;       // packed coordinates inside the owner's device block
    hipEvent_t ev = nullptr;          // recorded behind the owner's key kernel
    hipStream_t ev_stream = nullptr;
    int owner = -1;                   // sub-batch
    bool shared_later = false;        // some later sub-batch reads it
};
// what a batch call works on: the scans, the pairs as rows of the scan table, their initial transforms
struct batch_call {
    const pcr_cloud_ref* clouds = nullptr;
    int64_t n_clouds = 0;
    const pcr_pair_ref* pairs = nullptr;
    int64_t n_pairs = 0;
    std::unique_ptr<cloud_entry[]> cache;
    // device blocks whose contents later sub-batches read: released when the call ends (ctx, pointer, bytes)
    std::mutex keep_mu;
    struct kept { pcr_ctx* ctx; void* p; size_t bytes; };
    std::vector<kept> keep;
    const double* T0_of(int64_t i) const { return pairs[i].T0; }
};

// the per-pair path (what pcr_icp_batch did for every pair before the fused stages; still used for what they cannot take)
int run_one_pair(pcr_ctx* ctx, const batch_call& call, int64_t i, const pcr_icp_params* params, pcr_icp_result* res) {
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const pcr_pair_ref& R = call.pairs[i];
    if (R.src < 0 || R.src >= call.n_clouds || R.tgt < 0 || R.tgt >= call.n_clouds) return PCR_E_INVALID;
    const pcr_cloud_ref &S = call.clouds[R.src], &T = call.clouds[R.tgt];
    pcr_cloud *src = nullptr, *tgt = nullptr;
    pcr_index* index = nullptr;
    int rc = pcr_cloud_upload_f32(ctx, S.xyz, S.n, S.stride, &src);
    if (rc == PCR_OK) rc = pcr_cloud_upload_f32(ctx, T.xyz, T.n, T.stride, &tgt);
    if (rc == PCR_OK) rc = pcr_index_build(ctx, tgt, PCR_INDEX_GRID, 0.0, &index);
    if (rc == PCR_OK) rc = pcr_icp(ctx, src, index, params, R.T0 ? R.T0 : eye, res);
    if (index) pcr_index_free(ctx, index);
    if (tgt) pcr_cloud_free(ctx, tgt);
    if (src) pcr_cloud_free(ctx, src);
    return rc;
}

struct batch_timing {
    std::atomic<long long> ns[6];   // stage, set-up enqueue, ICP (incl. waiting for the set-up), read-back, fallbacks, sub-batches
};

// pinned host memory of a context, grown on demand (hipHostMalloc pins pages under a process-wide lock: kept across batches)
int ensure_pinned(pcr_ctx* ctx, size_t bytes) {
    if (ctx->h_stage_bytes >= bytes && ctx->h_stage) return PCR_OK;
    pcr_sync(ctx->stream);
    if (ctx->h_stage) hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->h_stage_bytes = 0;
    const size_t want = bytes + bytes / 4;
    if (hipHostMalloc(&ctx->h_stage, want, hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        ctx->last_error = "hipHostMalloc(batch staging)";
        return PCR_E_NOMEM;
    }
    ctx->h_stage_bytes = want;
    return PCR_OK;
}

// One sub-batch in flight on one context: begin() packs the clouds, enqueues the whole set-up and the first chunk of ICP passes
// and returns WITHOUT waiting; end() waits, enqueues further chunks while pairs are still iterating, reads the loop states back
// and fills the results.  A worker thread keeps two jobs going (on a context and its companion): it packs the next sub-batch
// while the device works on the current one.  status[i] < 0 marks pairs that must take the per-pair path instead
// (PCR_E_UNSUPPORTED as a private "not taken" mark) or that are invalid; results/status of the others are final after end().
struct batch_job {
    pcr_ctx* ctx;
    batch_call* call = nullptr;
    int index = 0;                 // which sub-batch of the call
    std::vector<int64_t> ids;
    std::vector<char> room;        // per slot: it has room in the staging block (see layout())
    std::vector<int> alias;        // per slot: -1 = this slot packs its scan; >= 0: the slot of this sub-batch that does; -2: an earlier sub-batch's copy
    int m = 0;
    const pcr_icp_params* params = nullptr;
    pcr_icp_result* results = nullptr;
    int32_t* status = nullptr;
    batch_timing* tm = nullptr;
    bool active = false;           // begin() enqueued work that end() must collect
    std::vector<char> take;
    std::vector<char> bad_cloud;   // non-finite coordinates met while packing (one flag per cloud: clouds are packed by different threads)
    std::vector<batch_cloud> cl;
    std::vector<batch_target> tg;
    std::vector<double> hi_all;    // upper corners of the clouds' bounding boxes
    int n_clouds = 0;
    unsigned long long slots = 0, src_slots = 0;
    size_t xyz_bytes = 0, off_cl = 0, off_tg = 0, off_T0 = 0;
    bool nothing = false;          // layout() found nothing for the fused stages
    bool keep_in = false;          // later sub-batches read this one's device copy of the coordinates: it outlives release()
    hipEvent_t own_event = nullptr;
    char *hp = nullptr, *hp_dev = nullptr;
    size_t off_st = 0, off_plan = 0;
    pcr_batch_pass_args a{};
    batch_plan* d_plan = nullptr;
    unsigned int *d_running = nullptr, *h_run = nullptr, *h_run_dev = nullptr;
    int enq = 0, chunk = 0;
    std::chrono::steady_clock::time_point t_begin, t_staged, t_enq;
    dev_block d_in, d_pts, d_keys, d_keys2, d_vals, d_vals2, d_tmp, d_cells, d_blocks, d_small, d_res, d_prev, d_cost, d_items, d_acc, d_st, d_tp;
    explicit batch_job(pcr_ctx* c)
        : ctx(c), d_in(c), d_pts(c), d_keys(c), d_keys2(c), d_vals(c), d_vals2(c), d_tmp(c), d_cells(c), d_blocks(c), d_small(c), d_res(c), d_prev(c), d_cost(c),
          d_items(c), d_acc(c), d_st(c), d_tp(c) {}
    void release() {   // scratch back to the arena (stream-ordered with what was enqueued)
        if (keep_in && d_in.p) {   // (the call releases it when every sub-batch is done)
            std::lock_guard<std::mutex> g(call->keep_mu);
            call->keep.push_back({ctx, d_in.p, d_in.bytes});
            d_in.p = nullptr;
            d_in.bytes = 0;
            keep_in = false;
        }
        for (dev_block* b : {&d_in, &d_pts, &d_keys, &d_keys2, &d_vals, &d_vals2, &d_tmp, &d_cells, &d_blocks, &d_small, &d_res, &d_prev, &d_cost, &d_items, &d_acc, &d_st, &d_tp}) b->free_now();
        active = false;
    }
    static std::chrono::steady_clock::time_point now() { return std::chrono::steady_clock::now(); }
    static long long ns(std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
        return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(y - x).count();
    }
    // begin() = layout() -> pack_cloud(c) for every cloud c in [0, 2m) (any thread, in any order) -> launch()
    int layout(batch_call* call_, int index_, const int64_t* ids_, int m_, const pcr_icp_params* params_, pcr_icp_result* results_, int32_t* status_, batch_timing* tm_);
    void pack_cloud(int c);
    int launch();
    const pcr_cloud_ref* ref_of(int c) const {   // scan of slot c (nullptr: the pair names no valid scan)
        const pcr_pair_ref& R = call->pairs[ids[c % m]];
        const int64_t g = c < m ? R.src : R.tgt;
        return (g >= 0 && g < call->n_clouds) ? &call->clouds[g] : nullptr;
    }
    int64_t scan_of(int c) const { const pcr_pair_ref& R = call->pairs[ids[c % m]]; return c < m ? R.src : R.tgt; }
    int enqueue_chunk();
    int end();
};

int batch_job::layout(batch_call* call_, int index_, const int64_t* ids_, int m_, const pcr_icp_params* params_, pcr_icp_result* results_, int32_t* status_,
                      batch_timing* tm_) {
    call = call_; index = index_; m = m_; params = params_; results = results_; status = status_; tm = tm_;
    ids.assign(ids_, ids_ + m_);
    active = false;
    nothing = false;
    t_begin = now();
    hipSetDevice(ctx->device);
    n_clouds = 2 * m;
    // ---- slots: sources first (a source slot index is a query index of the pass kernels), then targets
    cl.assign(n_clouds, batch_cloud{});
    tg.assign(m, batch_target{});
    take.assign(m, 1);
    bad_cloud.assign(n_clouds, 0);
    hi_all.assign(3 * (size_t)n_clouds, 0.0);
    slots = 0;
    auto usable = [&](int c) {
        const pcr_cloud_ref* R = ref_of(c);
        return R && R->n > 0 && R->stride >= 3 && R->xyz && R->n <= (1 << 24);
    };
    for (int c = 0; c < n_clouds; ++c)
        if (!usable(c)) take[c % m] = 0;   // empty / invalid / very large clouds: per-pair path decides
    // Who packs a slot's scan: an earlier sub-batch (the scan's owner: alias -2), another slot of this one (alias = that slot), or
    // the slot itself (alias -1).  A slot has ROOM in the staging block when its pair is taken -- or when it is the one that must bring
    // a scan over for later sub-batches although its own pair is not (room, but no points in the fused stages).
    alias.assign(n_clouds, -1);
    room.assign(n_clouds, 0);
    {
        std::vector<std::pair<int64_t, int>> packer;   // (scan, slot) of this sub-batch's packers
        auto find = [&](int64_t g) { for (auto& e : packer) if (e.first == g) return e.second; return -1; };
        for (int c = 0; c < n_clouds; ++c) {
            if (!take[c % m]) continue;
            room[c] = 1;
            const int64_t g = scan_of(c);
            if (call->cache[g].owner != index) { alias[c] = -2; continue; }
            const int d = find(g);
            if (d >= 0) alias[c] = d;
            else packer.emplace_back(g, c);
        }
        for (int c = 0; c < n_clouds; ++c) {
            if (take[c % m] || !usable(c)) continue;
            const int64_t g = scan_of(c);
            const cloud_entry& E = call->cache[g];
            if (E.owner != index || !E.shared_later || find(g) >= 0) continue;
            room[c] = 1;
            packer.emplace_back(g, c);
        }
    }
    for (int c = 0; c < n_clouds; ++c) {
        batch_cloud& C = cl[c];
        C.off = slots;
        C.n = take[c % m] ? ref_of(c)->n : 0;
        C.n_copy = room[c] ? ref_of(c)->n : 0;
        C.src = nullptr;
        slots += (unsigned long long)((C.n_copy + SLOT - 1) / SLOT) * SLOT;
    }
    src_slots = cl[m].off;
    if (slots == 0 || slots >= (1ull << 31)) {
        for (int k = 0; k < m; ++k) status[ids[k]] = PCR_E_UNSUPPORTED;
        nothing = true;
        return PCR_OK;
    }
    // ---- pinned staging: packed xyz | clouds | targets | T0s ; read-back: states
    xyz_bytes = (size_t)slots * 12;
    off_cl = (xyz_bytes + 255) & ~(size_t)255;
    off_tg = off_cl + ((sizeof(batch_cloud) * n_clouds + 255) & ~(size_t)255);
    off_T0 = off_tg + ((sizeof(batch_target) * m + 255) & ~(size_t)255);
    off_st = off_T0 + ((128 * (size_t)m + 255) & ~(size_t)255);
    off_plan = off_st + ((sizeof(pcr_icp_dev_state) * (size_t)m + 255) & ~(size_t)255);
    const size_t pinned_bytes = off_plan + 256;
    const int rc = ensure_pinned(ctx, pinned_bytes);
    if (rc) return rc;
    hp = (char*)ctx->h_stage;
    return PCR_OK;
}

// xyz columns of cloud c -> the staging block, its exact box on the way (float -> double is exact: the box the per-pair path takes).
// Only the slot that owns the scan packs it; the box goes to the call's table, where every other use of the scan finds it.
void batch_job::pack_cloud(int c) {
    float* const h_xyz = (float*)hp;
    if (!room[c]) return;
    cloud_entry& E = call->cache[scan_of(c)];
    if (alias[c] == -1) {
        const pcr_cloud_ref& R = *ref_of(c);
        const int64_t n = R.n, stride = R.stride;
        const float* ptr = R.xyz;
        float* dst = h_xyz + 3 * cl[c].off;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        bool finite = true;
        for (int64_t i = 0; i < n; ++i) {   // (float -> double is exact: the box equals the one the per-pair path takes in binary64)
            const float* p = ptr + i * stride;
            const float x = p[0], y = p[1], z = p[2];
            dst[3 * i + 0] = x; dst[3 * i + 1] = y; dst[3 * i + 2] = z;
            finite = finite && std::isfinite(x) && std::isfinite(y) && std::isfinite(z);
            lo[0] = x < lo[0] ? x : lo[0]; hi[0] = x > hi[0] ? x : hi[0];
            lo[1] = y < lo[1] ? y : lo[1]; hi[1] = y > hi[1] ? y : hi[1];
            lo[2] = z < lo[2] ? z : lo[2]; hi[2] = z > hi[2] ? z : hi[2];
        }
        for (int a = 0; a < 3; ++a) { E.lo[a] = lo[a]; E.hi[a] = hi[a]; }
        E.bad = !finite;
        E.box_ready.store(1, std::memory_order_release);
    } else {
        // packed by another slot (this sub-batch's or an earlier one's, whose packing tasks were handed out before this one)
        while (!E.box_ready.load(std::memory_order_acquire)) std::this_thread::yield();
    }
    if (E.bad) { bad_cloud[c] = 1; return; }   // (the per-pair path reports it)
    for (int a = 0; a < 3; ++a) { cl[c].lo[a] = (double)E.lo[a]; hi_all[3 * (size_t)c + a] = (double)E.hi[a]; }
}

int batch_job::launch() {
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    const auto t_packed = now();
    hipSetDevice(ctx->device);
    int rc;
    float* const h_xyz = (float*)hp;
    for (int k = 0; k < m; ++k)
        if (bad_cloud[k] || bad_cloud[m + k]) take[k] = 0;
    // grid parameters per pair: the functions the per-pair path uses, on the same boxes
    int mbits = 3;
    for (int k = 0; k < m; ++k) {
        if (!take[k]) continue;
        batch_cloud &S = cl[k], &T = cl[m + k];
        const double *s_hi = &hi_all[3 * (size_t)k], *t_hi = &hi_all[3 * (size_t)(m + k)];
        for (int a = 0; a < 3; ++a) tg[k].hi[a] = t_hi[a];
        double cell = 0;
        int levels = 1;
        pcr_grid_plan(T.lo, t_hi, T.n, 0.0, &cell, &levels);
        tg[k].cell = cell;
        tg[k].levels = levels;
        T.inv = 1.0 / cell;
        // the source's curve (pcr_cloud_morton_sort): the target's cell, clamped to the source's own extent
        const double emax_s = fmax(s_hi[0] - S.lo[0], fmax(s_hi[1] - S.lo[1], s_hi[2] - S.lo[2]));
        double cell_s = cell;
        if (cell_s < emax_s / 262144.0) cell_s = emax_s / 262144.0;
        S.inv = 1.0 / cell_s;
        double sc = 0, isc = 0;
        if (!pcr_pass_fixed_scale(T.lo, t_hi, S.n, params->max_d2, &sc, &isc)) { take[k] = 0; continue; }   // absurd extents: per-pair path (binary64 slabs)
        tg[k].scale = sc;
        tg[k].inv_scale = isc;
        const int bt = pcr_morton_end_bit(T.lo, t_hi, T.inv), bs = pcr_morton_end_bit(S.lo, s_hi, S.inv);
        mbits = bt > mbits ? bt : mbits;
        mbits = bs > mbits ? bs : mbits;
    }
    int cbits = 1;
    while ((1 << cbits) < n_clouds) ++cbits;
    bool any = false;
    for (int k = 0; k < m; ++k) {
        if (!take[k]) {
            status[ids[k]] = PCR_E_UNSUPPORTED;
            cl[k].n = 0;       // (its slots stay, empty: no tile of it does anything; a shared scan is still brought over: n_copy)
            cl[m + k].n = 0;
            tg[k].levels = 0;
            tg[k].cell = 1.0; tg[k].scale = tg[k].inv_scale = 1.0;
            for (int a = 0; a < 3; ++a) { tg[k].hi[a] = 0; cl[k].lo[a] = cl[m + k].lo[a] = 0; }
            cl[k].inv = cl[m + k].inv = 1.0;
        } else any = true;
    }
    bool must_upload = false;   // a scan later sub-batches wait for is brought over even when no pair of this sub-batch is taken
    for (int c = 0; c < n_clouds; ++c) must_upload = must_upload || (room[c] && alias[c] == -1 && call->cache[scan_of(c)].shared_later);
    if ((!any && !must_upload) || mbits + cbits > 63) {
        for (int k = 0; k < m; ++k) status[ids[k]] = PCR_E_UNSUPPORTED;
        if (must_upload) return PCR_E_UNSUPPORTED;   // (cannot happen: mbits + cbits <= 63 for clouds of <= 2^24 points in <= 1024 slots)
        return PCR_OK;
    }
    // table pools: rigorous upper bounds of what the plan kernel will hand out (cells of level l <= min(n, cells of the box))
    unsigned long long cell_pool_slots = 0, block_pool_slots = 0;
    for (int k = 0; k < m; ++k) {
        if (!take[k]) continue;
        const batch_cloud& T = cl[m + k];
        for (int l = 0; l < tg[k].levels; ++l) {
            double box = 1.0;
            for (int a = 0; a < 3; ++a) {
                const double k0 = floor((tg[k].hi[a] - T.lo[a]) * T.inv) + 2.0;
                box *= floor(ldexp(k0, -2 * l)) + 1.0;
            }
            const unsigned long long bound = box < (double)T.n ? (unsigned long long)box : (unsigned long long)T.n;
            unsigned int cap = next_pow2_host(bound * 4 + 4), bcap = next_pow2_host(bound * 2 + 4);
            cell_pool_slots += cap < MIN_CAP ? MIN_CAP : cap;
            block_pool_slots += bcap < MIN_CAP ? MIN_CAP : bcap;
        }
    }
    memcpy(hp + off_tg, tg.data(), sizeof(batch_target) * m);
    double* const h_T0 = (double*)(hp + off_T0);
    for (int k = 0; k < m; ++k) memcpy(h_T0 + 16 * (size_t)k, call->pairs[ids[k]].T0 ? call->pairs[ids[k]].T0 : eye, 128);
    t_staged = t_packed;
    // ---- device memory
    const unsigned int n_tiles = (unsigned int)(src_slots / 32);
    size_t items_bytes = 0, acc_bytes = 0, sync_word = 0;
    unsigned int cap = 0;
    pcr_grid_batch_scratch_bytes(n_tiles, m, &items_bytes, &acc_bytes, &sync_word, &cap);
    const int n_e = m * PCR_MAX_LEVELS;
    size_t temp_bytes = 0;
    {
        unsigned long long* kn = nullptr;
        unsigned int* vn = nullptr;
        if (rocprim::radix_sort_pairs(nullptr, temp_bytes, kn, kn, vn, vn, (size_t)slots, 0, (unsigned int)(mbits + cbits), ctx->stream) != hipSuccess) return PCR_E_HIP;
    }
    // small things in one block: clouds | targets | T0s | counts | cell offsets | plan | pairs | running
    const size_t s_cl = 0, s_tg = off_tg - off_cl, s_T0 = off_T0 - off_cl, s_in_end = off_st - off_cl;   // (same layout as the pinned block)
    const size_t s_counts = s_in_end, s_coff = s_counts + ((4 * (size_t)n_e + 255) & ~(size_t)255), s_plan = s_coff + ((8 * ((size_t)n_e + 1) + 255) & ~(size_t)255);
    const size_t s_pairs = s_plan + 256, s_run = s_pairs + ((sizeof(pcr_batch_pair) * m + 255) & ~(size_t)255), s_end = s_run + ((4 * (PCR_ICP_MAX_LOG + 1) + 255) & ~(size_t)255);
    if ((rc = d_in.alloc(xyz_bytes)) || (rc = d_pts.alloc(sizeof(pcr_pt) * (size_t)slots)) || (rc = d_keys.alloc(8 * (size_t)slots)) ||
        (rc = d_keys2.alloc(8 * (size_t)slots)) || (rc = d_vals.alloc(4 * (size_t)slots)) || (rc = d_vals2.alloc(4 * (size_t)slots)) ||
        (rc = d_tmp.alloc(temp_bytes)) || (rc = d_cells.alloc(sizeof(pcr_cell_slot) * (size_t)cell_pool_slots)) ||
        (rc = d_blocks.alloc(sizeof(pcr_block_slot) * (size_t)block_pool_slots)) || (rc = d_small.alloc(s_end)) || (rc = d_res.alloc(4 * (size_t)src_slots)) ||
        (rc = d_prev.alloc(24 * (size_t)src_slots)) || (rc = d_cost.alloc(4 * (size_t)n_tiles)) || (rc = d_items.alloc(items_bytes)) ||
        (rc = d_acc.alloc(acc_bytes)) || (rc = d_st.alloc(sizeof(pcr_icp_dev_state) * (size_t)m)) || (rc = d_tp.alloc(4 * (size_t)n_tiles)))
        return rc;
    char* const ds = d_small.as<char>();
    const batch_cloud* d_cl = (const batch_cloud*)(ds + s_cl);
    const batch_target* d_tg = (const batch_target*)(ds + s_tg);
    const double* d_T0 = (const double*)(ds + s_T0);
    unsigned int* d_counts = (unsigned int*)(ds + s_counts);
    unsigned long long* d_coff = (unsigned long long*)(ds + s_coff);
    d_plan = (batch_plan*)(ds + s_plan);
    pcr_batch_pair* d_pairs = (pcr_batch_pair*)(ds + s_pairs);
    d_running = (unsigned int*)(ds + s_run);
    hipStream_t st = ctx->stream;
    PCR_HIP(ctx, hipEventRecord(ctx->ev0, st));
    // ---- uploads, keys, sort, records
    // everything crosses PCIe through kernels that read / write the pinned, device-mapped staging block: no copy engine
    hp_dev = nullptr;
    PCR_HIP(ctx, hipHostGetDevicePointer((void**)&hp_dev, hp, 0));
    static const bool use_dma = getenv("PCR_BATCH_DMA") != nullptr;   // A/B: the packed coordinates by the copy engine instead
    // where the key kernel finds every slot's coordinates: this sub-batch's staging block, or -- a scan an earlier sub-batch of the
    // call owns -- that sub-batch's device copy, behind its key kernel (an event on its stream; its launch is waited for here: the
    // owner is always an EARLIER sub-batch, whose tasks were handed out before this one's, so the wait cannot form a cycle)
    bool owns_shared = false;
    for (int c = 0; c < n_clouds; ++c) {
        if (!room[c]) continue;
        const float* const own_base = use_dma ? d_in.as<float>() : (const float*)hp_dev;
        if (alias[c] == -1) { cl[c].src = own_base + 3 * cl[c].off; owns_shared = owns_shared || call->cache[scan_of(c)].shared_later; }
        else if (alias[c] >= 0) cl[c].src = own_base + 3 * cl[alias[c]].off;
        else if (!take[c % m]) cl[c].n_copy = 0;   // (un-taken since the lay-out, and somebody else's scan: nothing to bring over)
        else {
            cloud_entry& E = call->cache[scan_of(c)];
            while (!E.resident.load(std::memory_order_acquire)) std::this_thread::yield();
            if (E.ev_stream != st) PCR_HIP(ctx, hipStreamWaitEvent(st, E.ev, 0));
            cl[c].src = E.dev;
        }
    }
    memcpy(hp + off_cl, cl.data(), sizeof(batch_cloud) * n_clouds);
    hipLaunchKernelGGL(batch_words_kernel, dim3(8), dim3(256), 0, st, (const unsigned long long*)(hp_dev + off_cl), (unsigned long long*)ds,
                       (unsigned long long)(s_in_end / 8), (unsigned long long)((s_plan + 256) / 8));   // descriptors; counts, offsets and plan zeroed
    const unsigned int blocks_all = (unsigned int)(slots / SLOT), blocks_src = (unsigned int)(src_slots / SLOT), blocks_tgt = blocks_all - blocks_src;
    if (use_dma) PCR_HIP(ctx, hipMemcpyAsync(d_in.p, h_xyz, xyz_bytes, hipMemcpyHostToDevice, st));
    static const int keys_blocks_env = getenv("PCR_BATCH_KEYS_BLOCKS") ? atoi(getenv("PCR_BATCH_KEYS_BLOCKS")) : 0;
    unsigned int keys_grid = keys_blocks_env > 0 ? (unsigned int)keys_blocks_env : 64u;   // (256 pairs, ms per batch: one block per 256 records 6.4-6.8, 1024 blocks 6.1-6.3, 128: 5.9, 64: 5.4, 32: 5.7, 16: 5.9-6.2)
    if (keys_grid > blocks_all) keys_grid = blocks_all;
    hipLaunchKernelGGL(batch_keys_kernel, dim3(keys_grid), dim3(SLOT), 0, st, d_in.as<float>(), d_cl, n_clouds,
                       mbits, blocks_all, d_keys.as<unsigned long long>(), d_vals.as<unsigned int>());
    if (owns_shared) {
        // later sub-batches read this one's device copy of the scans they share with it
        hipEvent_t ev = nullptr;
        PCR_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        PCR_HIP(ctx, hipEventRecord(ev, st));
        own_event = ev;
        keep_in = true;
        for (int c = 0; c < n_clouds; ++c) {
            if (!room[c] || alias[c] != -1) continue;
            cloud_entry& E = call->cache[scan_of(c)];
            if (!E.shared_later) continue;
            E.dev = d_in.as<float>() + 3 * cl[c].off;
            E.ev = ev;
            E.ev_stream = st;
            E.resident.store(1, std::memory_order_release);
        }
    }
    // (rocPRIM's default configuration = merge sort at this size.  Onesweep is faster for one sort of 1.3 M pairs alone -- 126 against
    // 190 us -- but not with eight sub-batches' sorts in flight together, and it adds a dozen 17-us fills per sort.)
    PCR_HIP(ctx, rocprim::radix_sort_pairs(d_tmp.p, temp_bytes, d_keys.as<unsigned long long>(), d_keys2.as<unsigned long long>(), d_vals.as<unsigned int>(),
                                           d_vals2.as<unsigned int>(), (size_t)slots, 0, (unsigned int)(mbits + cbits), st));
    hipLaunchKernelGGL(batch_gather_kernel, dim3(blocks_all), dim3(SLOT), 0, st, d_in.as<float>(), d_vals2.as<unsigned int>(), d_cl, n_clouds, d_pts.as<pcr_pt>());
    // ---- grids of all targets
    if (blocks_tgt)
        hipLaunchKernelGGL(batch_count_kernel, dim3(blocks_tgt), dim3(SLOT), 0, st, d_keys2.as<unsigned long long>(), d_cl, n_clouds, m, d_tg, src_slots, mbits, d_counts);
    hipLaunchKernelGGL(batch_plan_kernel, dim3(1), dim3(256), 0, st, d_counts, d_cl, d_tg, m, d_pts.as<pcr_pt>(), d_cells.as<pcr_cell_slot>(), cell_pool_slots,
                       d_blocks.as<pcr_block_slot>(), block_pool_slots, d_pairs, d_coff, d_plan);
    hipLaunchKernelGGL(batch_init_tables_kernel, dim3(4 * ctx->cu_count), dim3(256), 0, st, d_cells.as<pcr_cell_slot>(), d_blocks.as<pcr_block_slot>(), d_plan, d_pairs, m,
                       d_tp.as<unsigned int>(), n_tiles);
    if (blocks_tgt)
        hipLaunchKernelGGL(batch_insert_cells_kernel, dim3(blocks_tgt), dim3(SLOT), 0, st, d_keys2.as<unsigned long long>(), d_cl, n_clouds, m, d_pairs, src_slots, mbits,
                           d_plan);
    if (blocks_tgt)
        hipLaunchKernelGGL(batch_insert_blocks_kernel, dim3(blocks_tgt), dim3(SLOT), 0, st, d_keys2.as<unsigned long long>(), d_cl, n_clouds, m, d_pairs, src_slots, mbits,
                           d_plan);
    PCR_HIP(ctx, hipGetLastError());
    // ---- ICP
    a = pcr_batch_pass_args{};
    a.pairs = d_pairs;
    a.n_pairs = m;
    a.tile_pair = d_tp.as<unsigned int>();
    a.n_tiles = n_tiles;
    a.q = d_pts.as<pcr_pt>();
    a.res_pos = d_res.as<unsigned int>();
    a.prev_xyz = d_prev.p;
    a.tile_cost = d_cost.as<unsigned int>();
    a.items = d_items.as<unsigned long long>();
    a.acc = d_acc.as<unsigned long long>();
    a.sync = a.acc + sync_word;
    a.cap = cap;
    a.st = d_st.as<pcr_icp_dev_state>();
    a.running = d_running;
    a.la.max_iter = params->max_iter; a.la.min_iter = params->min_iter;
    a.la.compat = params->mode == PCR_ICP_COMPAT_MAIN; a.la.r_metric = params->r_metric;
    a.la.r_thres = params->r_thres; a.la.t_thres = params->t_thres;
    a.max_d2 = params->max_d2;
    if ((rc = pcr_grid_batch_init(ctx, &a, d_T0))) return rc;
    PCR_HIP(ctx, hipEventRecord(ctx->ev2, st));
    t_enq = now();
    enq = 0;
    chunk = params->min_iter > 2 ? params->min_iter : 2;
    if (ctx->profile) chunk = 1;   // per-launch HIP events: a pass behind the last stop would log empty kernels
    h_run = (unsigned int*)ctx->h_pinned;   // (pinned, device-mapped, coherent)
    h_run_dev = nullptr;
    PCR_HIP(ctx, hipHostGetDevicePointer((void**)&h_run_dev, h_run, 0));
    // the first chunk of passes goes out with the set-up: begin() returns with everything enqueued and nothing waited for
    if (params->max_iter > 0) {
        if ((rc = enqueue_chunk())) return rc;
    }
    active = true;
    return PCR_OK;
}

int batch_job::enqueue_chunk() {
    if (chunk > params->max_iter - enq) chunk = params->max_iter - enq;
    if (chunk > 64) chunk = 64;
    int rc;
    for (int c = 0; c < chunk; ++c)
        if ((rc = pcr_grid_batch_pass(ctx, &a, (unsigned int)(enq + c)))) return rc;
    enq += chunk;
    hipLaunchKernelGGL(batch_flag_kernel, dim3(1), dim3(1), 0, ctx->stream, (const unsigned int*)(d_running + (enq - 1)), h_run_dev);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int batch_job::end() {
    if (!active) return PCR_OK;
    const auto t_wait = now();
    hipSetDevice(ctx->device);
    hipStream_t st = ctx->stream;
    int rc;
    for (;;) {
        PCR_HIP(ctx, pcr_sync(st));
        if (params->max_iter <= 0 || *h_run == 0 || enq >= params->max_iter) break;
        if (!ctx->profile) chunk *= 2;
        if ((rc = enqueue_chunk())) return rc;
    }
    PCR_HIP(ctx, hipEventRecord(ctx->ev1, st));
    const auto t_icp = now();
    pcr_icp_dev_state* const h_st = (pcr_icp_dev_state*)(hp + off_st);
    const batch_plan* const h_plan = (const batch_plan*)(hp + off_plan);
    static_assert(sizeof(pcr_icp_dev_state) % 8 == 0 && sizeof(batch_plan) % 8 == 0, "copied as 8-byte words");
    {
        const unsigned long long nw = sizeof(pcr_icp_dev_state) / 8 * (unsigned long long)m;
        hipLaunchKernelGGL(batch_words_kernel, dim3(64), dim3(256), 0, st, (const unsigned long long*)a.st, (unsigned long long*)(hp_dev + off_st), nw, nw);
        hipLaunchKernelGGL(batch_words_kernel, dim3(1), dim3(64), 0, st, (const unsigned long long*)d_plan, (unsigned long long*)(hp_dev + off_plan),
                           (unsigned long long)(sizeof(batch_plan) / 8), (unsigned long long)(sizeof(batch_plan) / 8));
    }
    PCR_HIP(ctx, hipGetLastError());
    PCR_HIP(ctx, pcr_sync(st));
    if (h_plan->overflow) { ctx->last_error = "batch: table pool bound exceeded"; return PCR_E_HIP; }
    float icp_ms = 0;
    hipEventElapsedTime(&icp_ms, ctx->ev2, ctx->ev1);
    int n_take = 0;
    for (int k = 0; k < m; ++k) n_take += take[k] ? 1 : 0;
    for (int k = 0; k < m; ++k) {
        if (!take[k]) continue;
        const pcr_icp_dev_state& S = h_st[k];
        pcr_icp_result* res = &results[ids[k]];
        memset(res, 0, sizeof(*res));
        double T_cur[16], T_total[16];
        pcr::T_from_xform(S.x, T_cur);
        memcpy(T_total, S.T_total, sizeof(T_total));
        if (!a.la.compat && S.status == PCR_OK && !S.converged && S.it == params->max_iter && params->max_iter > 0)
            pcr::T_mul4(T_cur, T_total, T_total);   // icp_template.py:195-198: a non-converged last iteration still updates homo_mat_total
        res->iters = S.it;
        res->status = S.status;
        res->n_assoc = S.n_assoc;
        res->cost = S.cost;
        res->mean_d2 = S.mean_d2;
        for (int i = 0; i < S.it && i < PCR_ICP_MAX_LOG; ++i) { res->r_diff[i] = S.r_diff[i]; res->t_diff[i] = S.t_diff[i]; }
        res->nn_launches = S.passes;
        res->device_ms = icp_ms / (double)n_take;   // the pair's share of the sub-batch's loop
        memcpy(res->T_total, T_total, sizeof(T_total));
        memcpy(res->T, a.la.compat ? T_cur : T_total, sizeof(T_cur));
        status[ids[k]] = S.status;
    }
    if (tm) {
        const auto t_end = now();
        tm->ns[0] += ns(t_begin, t_staged);
        tm->ns[1] += ns(t_staged, t_enq);
        tm->ns[2] += ns(t_wait, t_icp);
        tm->ns[3] += ns(t_icp, t_end);
        tm->ns[5] += 1;
    }
    release();
    return PCR_OK;
}

// CPUs of the NUMA node the device hangs off (sysfs: /sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<N>/cpulist):
// the packing threads of a rank read 245 MB of 24-byte records and write 123 MB of pinned staging memory per 256-pair batch, and with
// eight ranks of a node doing that at once, memory on the far socket halves it.  false: unknown (no pinning).
bool device_numa_cpus(int device, cpu_set_t* set) {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (char* c = bdf; *c; ++c) *c = (char)tolower((unsigned char)*c);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE* f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    const int got = fscanf(f, "%d", &node);
    fclose(f);
    if (got != 1 || node < 0) return false;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const size_t nr = fread(list, 1, sizeof(list) - 1, f);
    fclose(f);
    if (nr == 0) return false;
    CPU_ZERO(set);
    int n_set = 0;
    for (char* p = list; *p;) {   // "0-63,128-191"
        char* end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); p = end; }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) { CPU_SET((int)c, set); ++n_set; }
        while (*p == ',' || *p == '\n' || *p == ' ') ++p;
    }
    // only inside what the process may use at all
    cpu_set_t mine;
    if (sched_getaffinity(0, sizeof(mine), &mine) == 0) {
        int both = 0;
        for (int c = 0; c < CPU_SETSIZE; ++c) {
            if (CPU_ISSET(c, set) && !CPU_ISSET(c, &mine)) CPU_CLR(c, set);
            if (CPU_ISSET(c, set)) ++both;
        }
        n_set = both;
    }
    return n_set > 0;
}

// threads of a batch call: created one by one, so that a refused thread (EAGAIN under a process limit) degrades the pool instead of
// letting std::system_error escape through the C ABI; with none, the caller runs the work itself.  `cpus` (may be null): the
// spawned threads are bound to that set (the caller's own thread keeps its affinity).
template <typename F>
void run_pool(int n_workers, F&& worker, const cpu_set_t* cpus = nullptr) {
    std::vector<std::thread> pool;
    for (int c = 1; c < n_workers; ++c) {
        try {
            pool.emplace_back([&worker, cpus] {
                if (cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), cpus);
                worker();
            });
        }
        catch (...) { break; }
    }
    worker();
    for (auto& t : pool) t.join();
}

// the fused batch (or, with PCR_BATCH_PER_PAIR=1 / ungated runs, the per-pair path) over the call's pairs
int batch_run(pcr_ctx* const* ctxs, int n_ctx, batch_call& call, const pcr_icp_params* params, pcr_icp_result* results, int32_t* status_out) {
    const int64_t n_pairs = call.n_pairs;
    // several worker contexts keep the device busy together: two-launch ICP passes for what takes the per-pair path
    int was_shared[64];
    for (int c = 0; c < n_ctx && c < 64; ++c) {
        was_shared[c] = ctxs[c]->shared_device;
        if (n_ctx > 1) ctxs[c]->shared_device = 1;
    }
    const char* per_pair_s = getenv("PCR_BATCH_PER_PAIR");   // read per call: the tests switch it
    const bool gated = (params->max_d2 > 0) && std::isfinite(params->max_d2);
    const bool fused = gated && !(per_pair_s && atoi(per_pair_s) != 0);
    // Sub-batches.  The device does a sub-batch's stages the more efficiently the bigger it is (256 pairs x 20 000 points as ONE
    // sub-batch: 5.1 ms of kernels, 2.2 of them the key kernel's PCIe reads; as eight of 32 on eight streams: 7.0-7.5 ms wall), but
    // a sub-batch packed by one thread keeps the device waiting (12 ms for 256 pairs).  So the n_ctx threads pack ONE sub-batch
    // TOGETHER (a cloud each, from a shared counter); the thread that packs its last cloud launches it on the sub-batch's own
    // context and waits for it, while the others already pack the next one -- whose key kernel then reads over PCIe while the
    // first one's sort, grids and passes run.
    int64_t sub = 1;
    const char* sub_s = getenv("PCR_BATCH_SUB");
    if (fused) {
        sub = sub_s ? atoll(sub_s) : (n_ctx > 1 ? 64 : n_pairs);   // (256 pairs on 8 threads, ms: 32 or 64 per sub-batch 4.8, 128: 5.6, 256: 7.0)
        if (sub < 1) sub = 1;
        if (sub > 512) sub = 512;
    }
    // (a smaller first sub-batch -- device work after a quarter of a packing time -- measured 5.4 against 4.8-5.1 ms: not adopted)
    std::vector<int64_t> bounds(1, 0);
    // PCR_BATCH_SPLIT="a,b,c": sizes of the first sub-batches (then `sub` again): experiments with a short first / last one
    if (const char* split_s = fused ? getenv("PCR_BATCH_SPLIT") : nullptr) {
        for (const char* p = split_s; *p && bounds.back() < n_pairs;) {
            char* end = nullptr;
            const long v = strtol(p, &end, 10);
            if (end == p) break;
            if (v > 0) bounds.push_back(bounds.back() + v < n_pairs ? bounds.back() + v : n_pairs);
            p = *end ? end + 1 : end;
        }
    }
    while (bounds.back() < n_pairs) bounds.push_back(bounds.back() + sub < n_pairs ? bounds.back() + sub : n_pairs);
    const int64_t n_sub = (int64_t)bounds.size() - 1;
    std::vector<int32_t> status_own;
    int32_t* status = status_out;
    if (!status) { status_own.assign((size_t)n_pairs, 0); status = status_own.data(); }
    std::atomic<int> hard_error(PCR_OK);
    const bool timing = getenv("PCR_BATCH_TIMING") != nullptr;   // read per call
    batch_timing tm;
    for (auto& v : tm.ns) v = 0;
    const auto wall0 = std::chrono::steady_clock::now();
    auto note_error = [&](int rc) {
        if (rc < 0) {
            int expected = PCR_OK;
            hard_error.compare_exchange_strong(expected, rc);
        }
    };
    // what a sub-batch leaves for the per-pair path (marked PCR_E_UNSUPPORTED by the fused stages, or the whole sub-batch when they failed)
    auto per_pair_rest = [&](pcr_ctx* ctx, int64_t lo, int64_t hi) {
        const auto tf0 = std::chrono::steady_clock::now();
        for (int64_t i = lo; i < hi; ++i) {
            if (status[i] != PCR_E_UNSUPPORTED) continue;
            memset(&results[i], 0, sizeof(results[i]));
            status[i] = run_one_pair(ctx, call, i, params, &results[i]);
            note_error(status[i]);
        }
        if (timing) tm.ns[4] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tf0).count();
    };
    for (int64_t i = 0; i < n_pairs; ++i) status[i] = PCR_E_UNSUPPORTED;
    if (!fused) {   // per-pair path: a pool of workers, one context each, pairs from a shared counter
        std::atomic<int64_t> next(0);
        std::atomic<int> next_ctx(0);
        auto worker = [&]() {
            pcr_ctx* ctx = ctxs[next_ctx.fetch_add(1)];
            hipSetDevice(ctx->device);
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n_pairs) break;
                per_pair_rest(ctx, i, i + 1);
            }
        };
        run_pool((int)(n_pairs < n_ctx ? n_pairs : n_ctx), worker);
    } else {
        // every scan's owner = the first sub-batch that uses it; shared_later = some other sub-batch uses it too
        for (int64_t j = 0; j < n_sub; ++j)
            for (int64_t i = bounds[(size_t)j]; i < bounds[(size_t)j + 1]; ++i)
                for (int h = 0; h < 2; ++h) {
                    const int64_t g = h == 0 ? call.pairs[i].src : call.pairs[i].tgt;
                    if (g < 0 || g >= call.n_clouds) continue;
                    cloud_entry& E = call.cache[g];
                    if (E.owner < 0) E.owner = (int)j;
                    else if (E.owner != (int)j) E.shared_later = true;
                }
        struct sub_state {
            std::unique_ptr<batch_job> job;
            int64_t lo = 0, hi = 0;
            std::vector<int64_t> ids;
            std::once_flag once;
            std::atomic<int> packed{0};
            std::atomic<int> finished{0};
            int rc = PCR_OK;
        };
        std::vector<sub_state> subs((size_t)n_sub);
        std::vector<int64_t> task0((size_t)n_sub + 1, 0);   // tasks = (sub-batch, cloud), sub-batch after sub-batch
        for (int64_t j = 0; j < n_sub; ++j) {
            sub_state& S = subs[(size_t)j];
            S.lo = bounds[(size_t)j];
            S.hi = bounds[(size_t)j + 1];
            for (int64_t i = S.lo; i < S.hi; ++i) S.ids.push_back(i);
            S.job.reset(new batch_job(ctxs[j % n_ctx]));   // sub-batch j runs on context j mod n_ctx, after sub-batch j - n_ctx
            task0[(size_t)j + 1] = task0[(size_t)j] + 2 * (S.hi - S.lo);
        }
        std::atomic<int64_t> next(0);
        const int64_t n_tasks = task0[(size_t)n_sub];
        auto worker = [&]() {
            for (;;) {
                const int64_t t = next.fetch_add(1);
                if (t >= n_tasks) break;
                int64_t j = (int64_t)(std::upper_bound(task0.begin(), task0.end(), t) - task0.begin()) - 1;
                sub_state& S = subs[(size_t)j];
                const int c = (int)(t - task0[(size_t)j]);
                batch_job& J = *S.job;
                std::call_once(S.once, [&] {
                    if (j >= n_ctx)   // the context (stream, staging block, arena) is still the earlier sub-batch's until that one has finished
                        while (!subs[(size_t)(j - n_ctx)].finished.load(std::memory_order_acquire)) std::this_thread::yield();
                    S.rc = J.layout(&call, (int)j, S.ids.data(), (int)S.ids.size(), params, results, status, timing ? &tm : nullptr);
                });
                const bool usable = S.rc == PCR_OK && !J.nothing;
                if (usable) J.pack_cloud(c);
                if (S.packed.fetch_add(1, std::memory_order_acq_rel) + 1 < 2 * (int)S.ids.size()) continue;
                // this thread packed the sub-batch's last cloud: it launches it, waits for it and finishes it
                hipSetDevice(J.ctx->device);
                int rc = S.rc;
                if (usable) {
                    rc = J.launch();
                    if (rc == PCR_OK) rc = J.end();
                }
                if (rc) {   // the whole sub-batch failed on the way (out of memory, HIP error): its pairs take the per-pair path
                    pcr_sync(J.ctx->stream);
                    J.release();
                    for (int64_t i = S.lo; i < S.hi; ++i) status[i] = PCR_E_UNSUPPORTED;
                }
                per_pair_rest(J.ctx, S.lo, S.hi);
                S.finished.store(1, std::memory_order_release);
            }
        };
        // Packing (6-float records -> packed coordinates in the mapped staging block: 0.75 ms per sub-batch of 64 pairs on eight
        // threads) is the critical path of a compat-mode batch -- every sub-batch's key kernel waits for it --, and the thread that
        // launches a sub-batch leaves the pool until that sub-batch is done.  Threads are not tied to contexts: more of them pack.
        // Default: two threads per context, at most 16 (a rank's share of the host on an 8-GPU node; 256 pairs on 8 contexts, C call:
        // 8 threads 5.5-5.8 ms, 12: 4.9-5.3, 16: 4.6-4.9, 24: no better), bound to the NUMA node of the device.
        const char* thr_s = getenv("PCR_BATCH_THREADS");
        int n_workers = thr_s ? atoi(thr_s) : (2 * n_ctx < 16 ? 2 * n_ctx : (n_ctx > 16 ? n_ctx : 16));
        if (n_workers < 1) n_workers = 1;
        if (n_workers > 64) n_workers = 64;
        if ((int64_t)n_workers > n_tasks) n_workers = (int)n_tasks;
        cpu_set_t node_cpus;
        static const bool no_pin = getenv("PCR_BATCH_NO_PIN") != nullptr;
        const bool pin = !no_pin && device_numa_cpus(ctxs[0]->device, &node_cpus);
        run_pool(n_workers, worker, pin ? &node_cpus : nullptr);
        // what later sub-batches read of earlier ones: back to the arenas now that every stream has been waited for
        for (auto& S : subs)
            if (S.job && S.job->own_event) hipEventDestroy(S.job->own_event);
        for (auto& k : call.keep) pcr_dev_free(k.ctx, k.p, k.bytes);
        call.keep.clear();
    }
    for (int c = 0; c < n_ctx && c < 64; ++c) ctxs[c]->shared_device = was_shared[c];
    if (timing)
        fprintf(stderr, "pcr_icp_batch: %lld pairs, %lld sub-batches of <= %lld, %d threads in %.2f ms; per sub-batch: layout + packing (all threads) %.0f us, launch %.0f us, "
                "waiting for the device + further passes %.0f us, read-back %.0f us; per-pair path %.0f us in total\n", (long long)n_pairs, (long long)tm.ns[5].load(), (long long)sub,
                n_ctx, std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - wall0).count() / 1e3,
                tm.ns[5] ? tm.ns[0] / 1e3 / tm.ns[5] : 0.0, tm.ns[5] ? tm.ns[1] / 1e3 / tm.ns[5] : 0.0, tm.ns[5] ? tm.ns[2] / 1e3 / tm.ns[5] : 0.0,
                tm.ns[5] ? tm.ns[3] / 1e3 / tm.ns[5] : 0.0, tm.ns[4] / 1e3);
    return hard_error.load();
}

int check_batch_args(pcr_ctx* const* ctxs, int n_ctx, const pcr_icp_params* params) {
    if (!ctxs || n_ctx <= 0 || !params) return PCR_E_INVALID;
    for (int c = 0; c < n_ctx; ++c)
        if (!ctxs[c]) return PCR_E_INVALID;
    if (params->max_iter > PCR_ICP_MAX_LOG) return PCR_E_TOO_MANY_ITERS;
    return PCR_OK;
}

}  // namespace

extern "C" int pcr_icp_batch(pcr_ctx* const* ctxs, int n_ctx, const pcr_pair* pairs, int64_t n_pairs, const pcr_icp_params* params,
                             pcr_icp_result* results, int32_t* status_out) {
    int rc = check_batch_args(ctxs, n_ctx, params);
    if (rc) return rc;
    if (n_pairs > 0 && (!pairs || !results)) return PCR_E_INVALID;
    if (n_pairs == 0) return PCR_OK;
    // the scan table of the call: host buffers that are the same (pointer, size, stride) are the same scan
    std::vector<pcr_cloud_ref> clouds;
    std::vector<pcr_pair_ref> refs((size_t)n_pairs);
    try {
        clouds.reserve((size_t)(2 * n_pairs));
        struct key { const float* p; int64_t n, s; };
        std::vector<std::pair<key, int>> table;   // open addressing by pointer hash
        size_t cap = 64;
        while (cap < (size_t)(4 * n_pairs)) cap <<= 1;
        table.assign(cap, {key{nullptr, 0, 0}, -1});
        auto scan = [&](const float* p, int64_t n, int64_t st) -> int {
            size_t h = ((size_t)(uintptr_t)p >> 4) * 0x9E3779B97F4A7C15ull;
            h ^= (size_t)n * 0xD1B54A32D192ED03ull + (size_t)st;
            for (size_t i = h & (cap - 1);; i = (i + 1) & (cap - 1)) {
                auto& e = table[i];
                if (e.second < 0) {
                    e.first = key{p, n, st};
                    e.second = (int)clouds.size();
                    clouds.push_back(pcr_cloud_ref{p, n, st});
                    return e.second;
                }
                if (e.first.p == p && e.first.n == n && e.first.s == st) return e.second;
            }
        };
        for (int64_t i = 0; i < n_pairs; ++i) {
            refs[(size_t)i].src = scan(pairs[i].src, pairs[i].n_src, pairs[i].stride_src);
            refs[(size_t)i].tgt = scan(pairs[i].tgt, pairs[i].n_tgt, pairs[i].stride_tgt);
            refs[(size_t)i].T0 = pairs[i].T0;
        }
    } catch (...) { return PCR_E_NOMEM; }
    batch_call call;
    call.clouds = clouds.data();
    call.n_clouds = (int64_t)clouds.size();
    call.pairs = refs.data();
    call.n_pairs = n_pairs;
    call.cache.reset(new (std::nothrow) cloud_entry[clouds.size()]);
    if (!call.cache) return PCR_E_NOMEM;
    return batch_run(ctxs, n_ctx, call, params, results, status_out);
}

extern "C" int pcr_global_default_params(double voxel_size, pcr_global_params* p) {
    if (!p || !(voxel_size > 0)) return PCR_E_INVALID;
    memset(p, 0, sizeof(*p));
    p->voxel_size = voxel_size;                 // main.py:196
    p->normal_radius = voxel_size * 2;          // main.py:38
    p->fpfh_radius = voxel_size * 5;            // main.py:43
    p->normal_max_nn = 30;                      // main.py:40
    p->fpfh_max_nn = 100;                       // main.py:46
    p->mutual_filter = 1;                       // main.py:74
    pcr_ransac_default_params(&p->ransac);
    p->ransac.max_distance = voxel_size * 1.5;  // main.py:70
    return PCR_OK;
}

// The pair loop of Registration/main.py:183-216 over a table of scans (see pcr.h).
extern "C" int pcr_register_pairs(pcr_ctx* const* ctxs, int n_ctx, const pcr_cloud_ref* clouds, int64_t n_clouds, const pcr_pair_ref* pairs, int64_t n_pairs,
                                  const pcr_global_params* global, const pcr_icp_params* icp, pcr_icp_result* results, int32_t* status_out, double* T_init_out) {
    static const double eye[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    int rc = check_batch_args(ctxs, n_ctx, icp);
    if (rc) return rc;
    if (n_pairs > 0 && (!pairs || !results || !clouds)) return PCR_E_INVALID;
    if (n_pairs == 0) return PCR_OK;
    for (int64_t i = 0; i < n_pairs; ++i)
        if (pairs[i].src < 0 || pairs[i].src >= n_clouds || pairs[i].tgt < 0 || pairs[i].tgt >= n_clouds) return PCR_E_INVALID;
    std::vector<pcr_pair_ref> refs;
    std::vector<double> T_init;
    try {
        refs.assign(pairs, pairs + n_pairs);
        T_init.assign((size_t)n_pairs * 16, 0.0);
    } catch (...) { return PCR_E_NOMEM; }
    for (int64_t i = 0; i < n_pairs; ++i) memcpy(&T_init[(size_t)i * 16], pairs[i].T0 ? pairs[i].T0 : eye, 128);
    std::atomic<int> hard_error(PCR_OK);
    // (a scan the stage cannot take -- PCR_E_EMPTY, PCR_E_UNSUPPORTED -- leaves its pairs at identity; anything else negative ends the call)
    auto note_error = [&](int e) { if (e < 0 && e != PCR_E_EMPTY && e != PCR_E_UNSUPPORTED) { int expected = PCR_OK; hard_error.compare_exchange_strong(expected, e); } };
    if (global) {
        // ---- prepare_dataset (main.py:197): every scan that a pair without a given T0 uses is preprocessed ONCE, on whichever
        // context's thread gets to it; the descriptors stay on the device for every pair the scan takes part in
        std::vector<char> need((size_t)n_clouds, 0);
        std::vector<int64_t> todo, todo_pairs;
        for (int64_t i = 0; i < n_pairs; ++i)
            if (!pairs[i].T0) { need[(size_t)pairs[i].src] = 1; need[(size_t)pairs[i].tgt] = 1; todo_pairs.push_back(i); }
        for (int64_t g = 0; g < n_clouds; ++g)
            if (need[(size_t)g]) todo.push_back(g);
        struct prep_slot { pcr_prep* prep = nullptr; pcr_ctx* ctx = nullptr; int rc = PCR_OK; };
        std::vector<prep_slot> preps((size_t)n_clouds);
        // the whole stage fused on the first context (all scans down-sampled by one sort, every later step one launch for all scans /
        // all pairs: pcr_global_init_batch); a share that does not fit that path is taken scan by scan and pair by pair below -- same results
        int fused = todo_pairs.empty() ? PCR_OK : pcr_global_init_batch(ctxs[0], clouds, n_clouds, todo.data(), (int64_t)todo.size(), pairs, todo_pairs.data(),
                                                                        (int64_t)todo_pairs.size(), global, T_init.data(), 16);
        if (fused != PCR_OK && fused != PCR_E_UNSUPPORTED) note_error(fused);
        if (fused == PCR_E_UNSUPPORTED && hard_error.load() == PCR_OK) {
        {
            std::atomic<int64_t> next(0);
            std::atomic<int> next_ctx(0);
            auto worker = [&]() {
                pcr_ctx* ctx = ctxs[next_ctx.fetch_add(1)];
                hipSetDevice(ctx->device);
                for (;;) {
                    const int64_t t = next.fetch_add(1);
                    if (t >= (int64_t)todo.size()) break;
                    const int64_t g = todo[(size_t)t];
                    prep_slot& P = preps[(size_t)g];
                    P.ctx = ctx;
                    pcr_cloud* full = nullptr;
                    P.rc = pcr_cloud_upload_f32(ctx, clouds[g].xyz, clouds[g].n, clouds[g].stride, &full);
                    if (P.rc == PCR_OK)
                        P.rc = pcr_preprocess(ctx, full, global->voxel_size, global->normal_radius, global->normal_max_nn, global->fpfh_radius, global->fpfh_max_nn, &P.prep);
                    if (full) pcr_cloud_free(ctx, full);
                }
            };
            run_pool((int)((int64_t)n_ctx < (int64_t)todo.size() ? n_ctx : (int)todo.size()), worker);
        }
        // ---- execute_global_registration (main.py:200) per pair: any context may read any scan's descriptors (same device, and
        // every pcr_preprocess has synchronised before it returned)
        {
            std::atomic<int64_t> next(0);
            std::atomic<int> next_ctx(0);
            auto worker = [&]() {
                pcr_ctx* ctx = ctxs[next_ctx.fetch_add(1)];
                hipSetDevice(ctx->device);
                for (;;) {
                    const int64_t t = next.fetch_add(1);
                    if (t >= (int64_t)todo_pairs.size()) break;
                    const int64_t i = todo_pairs[(size_t)t];
                    const prep_slot &S = preps[(size_t)pairs[i].src], &T = preps[(size_t)pairs[i].tgt];
                    if (S.rc != PCR_OK || T.rc != PCR_OK || !S.prep || !T.prep) {
                        // a scan the stage cannot take (empty, one voxel, ...): identity, like a pair without a valid hypothesis; hard errors are reported
                        note_error(S.rc);
                        note_error(T.rc);
                        continue;
                    }
                    pcr_ransac_result rr;
                    const int e = pcr_global_registration(ctx, S.prep, T.prep, &global->ransac, global->mutual_filter, &rr);
                    note_error(e);
                    if (e == PCR_OK) memcpy(&T_init[(size_t)i * 16], rr.T, 128);
                }
            };
            run_pool((int)((int64_t)n_ctx < (int64_t)todo_pairs.size() ? n_ctx : (int)todo_pairs.size()), worker);
        }
        for (auto& P : preps)
            if (P.prep) { hipSetDevice(P.ctx->device); pcr_prep_free(P.ctx, P.prep); }
        }
        for (int64_t i = 0; i < n_pairs; ++i) refs[(size_t)i].T0 = &T_init[(size_t)i * 16];
    }
    if (T_init_out) memcpy(T_init_out, T_init.data(), sizeof(double) * 16 * (size_t)n_pairs);
    if (hard_error.load() < 0) return hard_error.load();
    batch_call call;
    call.clouds = clouds;
    call.n_clouds = n_clouds;
    call.pairs = refs.data();
    call.n_pairs = n_pairs;
    call.cache.reset(new (std::nothrow) cloud_entry[(size_t)n_clouds]);
    if (!call.cache) return PCR_E_NOMEM;
    return batch_run(ctxs, n_ctx, call, icp, results, status_out);
}
