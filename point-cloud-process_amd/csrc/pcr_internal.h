// Internal types shared by the host C++ and the HIP kernels of libpcr.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "pcr.h"

#define PCR_HIDDEN __attribute__((visibility("hidden")))

// 32-byte point record: x,y,z (binary64) + original index (int64 bits).
struct __attribute__((aligned(32))) pcr_pt {
    double x, y, z;
    long long id;
};

struct pcr_xform {  // row-major 3x4: p' = R p + t
    double r[9];
    double t[3];
};

constexpr int PCR_MAX_LEVELS = 12;
constexpr int PCR_COORD_BITS = 21;
constexpr long long PCR_COORD_BIAS = 1ll << 20;  // multiple of 4^10: keeps level nesting aligned
constexpr long long PCR_COORD_MAX = (1ll << PCR_COORD_BITS) - 1;
constexpr unsigned long long PCR_EMPTY_KEY = ~0ull;

struct pcr_cell_slot {  // 16 B open-addressing slot: Morton key of a cell -> [start,end) in the sorted cloud
    unsigned long long key;
    unsigned int start;
    unsigned int end;
};

// 32-B slot of the block table: a 2x2x2 block of level-l cells.  The 8 children are consecutive runs of the sorted
// cloud (Morton order), so one slot = one cache line read answers 8 cell lookups: child c starts at
// start + sum(cnt[0..c)).  flags != 0: some child holds >= 65535 points, use the per-cell table for this block.
struct __attribute__((aligned(32))) pcr_block_slot {
    unsigned long long key;  // packed block coordinates (cell >> 1); PCR_EMPTY_KEY = free
    unsigned int start;
    unsigned int flags;
    unsigned short cnt[8];
};

// Device view of the multi-level voxel-hash grid over one target cloud.
struct pcr_grid_view {
    const pcr_pt* pts;  // target points sorted by level-0 Morton key (id = original index)
    long long n;
    int levels;
    double lo[3];    // grid origin (min corner of the target)
    double cell0;    // level-0 cell size; level l has cell0 * 4^l
    double inv_cell0;
    const pcr_cell_slot* table[PCR_MAX_LEVELS];
    unsigned int mask[PCR_MAX_LEVELS];  // capacity-1 (power of two)
    const pcr_block_slot* btable[PCR_MAX_LEVELS];  // 2x2x2-block tables (linear probing by slot)
    unsigned int bmask[PCR_MAX_LEVELS];
    double origin[3];                   // shift origin for moment accumulation (bbox centre)
};

struct pcr_cloud {
    pcr_pt* d = nullptr;
    int64_t n = 0;
    bool morton_sorted = false;  // records reordered for spatial locality (id keeps the caller's row)
    bool has_bbox = false;       // lo/hi below are the exact bounding box of the records (computed on the host while uploading;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};  // dropped as soon as the cloud is transformed)
};

struct pcr_index {
    int kind = PCR_INDEX_GRID;
    int64_t n = 0;
    double cell = 0;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    // GRID
    pcr_pt* sorted = nullptr;
    pcr_cell_slot* cell_pool = nullptr;       // every level's cell table (tables[l] point into it)
    pcr_block_slot* block_pool = nullptr;     // every level's 2x2x2-block table
    size_t cell_pool_bytes = 0, block_pool_bytes = 0;
    pcr_cell_slot* tables[PCR_MAX_LEVELS] = {nullptr};
    unsigned int caps[PCR_MAX_LEVELS] = {0};
    pcr_block_slot* btables[PCR_MAX_LEVELS] = {nullptr};
    unsigned int bcaps[PCR_MAX_LEVELS] = {0};
    pcr_grid_view view;
    pcr_grid_view* d_view = nullptr;  // device copy of `view`
    // BRUTE (f64 MFMA operand layout): tiles of 16 targets, 64 doubles per tile in lane order
    double* mfma_a = nullptr;   // [n_tiles][64]
    pcr_pt* plain = nullptr;    // targets in original order, centred copy not needed (exact recheck uses these)
    int64_t n_tiles = 0;
    double brute_rt = 0, brute_bias = 0;  // half diagonal of the target box; offset that keeps the contraction positive
};

struct pcr_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    std::string last_error;
    // grow-only caching allocator (device) keyed by size; avoids hipMalloc in the ICP loop
    struct blk { void* p; size_t sz; };
    std::vector<blk> free_list;
    std::vector<blk> live;       // blocks handed out, with their TRUE capacity (a reused block may be larger than asked for)
    std::vector<void*> arenas;   // 256-MiB hipMalloc chunks the blocks are carved from
    size_t arena_cap = 0, arena_used = 0;
    unsigned long long flag_seq = 0;   // pcr_wait_flag
    long long arena_grow_count = 0;   // hipMalloc calls for arenas since the context was created, and the host time they took
    double arena_grow_us = 0;
    // pinned host scratch for the per-iteration moment read-back
    double* h_pinned = nullptr;
    void* h_state = nullptr;              // pinned, device-mapped 8-KiB landing block of the device-resident ICP state (+ its pass log)
    void* h_small = nullptr;              // pinned, device-mapped landing block of pcr_d2h_small
    void* h_big = nullptr;                // pinned, device-mapped landing block of the few-query k-NN / radius paths (grown on demand)
    size_t h_big_bytes = 0;
    void* h_stage = nullptr;              // pinned staging buffer for uploads from pageable caller memory (grown on demand, <= 64 MiB)
    size_t h_stage_bytes = 0;
    void* h_init = nullptr;               // pinned packing block of the fused global initialisation (pcr_global_init_batch), grown on demand
    size_t h_init_bytes = 0;
    void* h_down = nullptr;               // pinned double buffer for large device-to-host results (pcr_d2h_staged), 2 x h_down_half bytes
    size_t h_down_half = 0;
    size_t h_pinned_bytes = 0;
    int shared_device = 0;                // pcr_ctx_set_shared: other contexts keep the device busy (two-launch ICP pass)
    int icp_lanes = 1;                    // runs of the source searched on separate streams per ICP pass (PCR_ICP_LANES)
    hipStream_t lane_stream[4] = {nullptr, nullptr, nullptr, nullptr};
    double* h_slabs = nullptr;            // pinned, device-mapped: per-block moment slabs of the host-sum ICP pass
    bool zero_copy = true;  // kernels write small results straight into h_pinned (PCR_NO_ZEROCOPY=1 disables)
    // device scratch for per-block partial moments
    double* d_partials = nullptr;
    size_t d_partials_bytes = 0;
    unsigned int* d_counters = nullptr;  // small zeroed scratch (tickets, flags)
    unsigned int* d_cell_counts = nullptr;  // [PCR_MAX_LEVELS][64] + ticket: run-start counters of the index build, zero between builds
    unsigned long long* d_debug = nullptr;  // diagnostics stamps (PCR_DEBUG_STAMPS=1), 1 Mi words
    // optional per-kernel profile of the ICP pass
    bool profile = false;
    hipEvent_t pev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    double prof_ms[4] = {0, 0, 0, 0};
    int prof_passes = 0;
    // per-pass log of the last device-resident ICP loop on this context (pcr_icp_pass_log): tile launch us, drain launch us, queue items
    double pass_log[3][PCR_ICP_MAX_LOG];
    int pass_log_n = 0;
    double loop_dev_ms = 0.0;   // that loop's duration by the kernels' own 100-MHz clock (0: not measured)
    double pass_host_us[6] = {0, 0, 0, 0, 0, 0};   // host phases of that call: set-up, enqueue of the first chunk, waiting for the device, whole loop; [4] HIP events: call start .. behind the last kernel
    int cu_count = 256;
    char name[256] = {0};
    int64_t hbm_bytes = 0;
};

// Waiting for the device WITHOUT sleeping on an interrupt: hipStreamSynchronize / hipEventSynchronize hand the thread to the
// runtime's blocking wait after a short spin, and on this pool's (virtualised) hosts the wake-up then comes 30-45 ms late every
// few dozen calls -- 1 M-point registrations whose kernels took their usual 15 ms were timed at 45-60 ms (the per-pass device log
// of pcr_icp_pass_log showed every kernel at its usual duration and no gap between them: DESIGN section 3.1.7).  Polling the
// stream's signal (hipStreamQuery reads it, no system call) does not depend on the interrupt; after two seconds of polling the
// blocking call takes over.
#include <chrono>
static inline hipError_t pcr_sync(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int spins = 0;; ++spins) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return hipStreamSynchronize(s);
        __builtin_ia32_pause();
    }
}
static inline hipError_t pcr_event_sync(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned int spins = 0;; ++spins) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) return hipEventSynchronize(ev);
        __builtin_ia32_pause();
    }
}

#define PCR_HIP(ctx, expr)                                                            \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) {                                                       \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);    \
            return PCR_E_HIP;                                                         \
        }                                                                             \
    } while (0)

PCR_HIDDEN int pcr_dev_alloc(pcr_ctx* ctx, size_t bytes, void** out);
PCR_HIDDEN void pcr_dev_free(pcr_ctx* ctx, void* p, size_t bytes);
// records of a cloud in caller row order (the device copy may be Morton-reordered)
PCR_HIDDEN int pcr_cloud_rows(pcr_ctx* ctx, const pcr_cloud* c, pcr_pt* d_out);

// device scratch that goes back to the context's arena when it leaves scope (every return path, error or not)
struct pcr_dev_block {
    pcr_ctx* ctx;
    void* p = nullptr;
    size_t bytes = 0;
    explicit pcr_dev_block(pcr_ctx* c) : ctx(c) {}
    pcr_dev_block(const pcr_dev_block&) = delete;
    pcr_dev_block& operator=(const pcr_dev_block&) = delete;
    int alloc(size_t b) { free_now(); bytes = b; return pcr_dev_alloc(ctx, b, &p); }
    void free_now() { if (p) pcr_dev_free(ctx, p, bytes); p = nullptr; bytes = 0; }
    ~pcr_dev_block() { free_now(); }
    template <typename T> T* as() const { return (T*)p; }
};
constexpr int PCR_MAX_LANES = 4;
constexpr int PCR_SLABS_PER_LANE = 256;
// d_counters: words 0..1023 small per-subsystem counters; from word 1024 on, 1024 words per search lane for the
// hard-list counters (32 counters, one per 128-byte line)
constexpr int PCR_HARD_COUNTERS = 1024;
constexpr size_t PCR_COUNTER_BYTES = 4 * (1024 + 1024 * 4);
PCR_HIDDEN int pcr_ctx_lanes(pcr_ctx* ctx, int lanes);  // creates the lane streams on first use
PCR_HIDDEN int pcr_ensure_scratch(pcr_ctx* ctx, size_t partial_bytes);
PCR_HIDDEN void pcr_xform_from_T(const double* T, pcr_xform* x);
// profile helpers: mark slot boundary k (0..4) on the stream; finish() syncs and accumulates
PCR_HIDDEN void pcr_prof_mark(pcr_ctx* ctx, int k);
PCR_HIDDEN void pcr_prof_finish(pcr_ctx* ctx);

// grid (pcr_grid.hip)
PCR_HIDDEN int pcr_bbox(pcr_ctx* ctx, const pcr_pt* pts, long long n, double lo[3], double hi[3]);
// bounding box of a cloud: the one remembered from the upload if still valid, otherwise a device reduction + read-back
PCR_HIDDEN int pcr_cloud_bbox(pcr_ctx* ctx, const pcr_cloud* c, double lo[3], double hi[3]);
PCR_HIDDEN int pcr_grid_build(pcr_ctx* ctx, const pcr_cloud* tgt, double cell, pcr_index* idx);
// Reorder a cloud's records along a Morton curve (cell = curve resolution); rigid transforms keep the locality.
PCR_HIDDEN int pcr_cloud_morton_sort(pcr_ctx* ctx, pcr_cloud* c, double cell);
PCR_HIDDEN void pcr_grid_free(pcr_ctx* ctx, pcr_index* idx);
// (the query cloud may be re-ordered on the device: record ids keep the caller's rows)
PCR_HIDDEN int pcr_grid_nn1(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_xform* x, double max_d2, int32_t* d_idx,
                            double* d_d2);
// One fused association+accumulate pass.  If write_back, q[i] <- x(q[i]) (in-place transform).
// d_moments receives 20 doubles: 18 moments + sum d2 + (unused).
PCR_HIDDEN int pcr_grid_icp_pass(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_xform* x, double max_d2,
                                 int write_back, double* d_moments);
// brute (pcr_brute.hip)
PCR_HIDDEN int pcr_brute_build(pcr_ctx* ctx, const pcr_cloud* tgt, pcr_index* idx);
PCR_HIDDEN void pcr_brute_free(pcr_ctx* ctx, pcr_index* idx);
PCR_HIDDEN int pcr_brute_nn1(pcr_ctx* ctx, const pcr_index* idx, const pcr_pt* q, int64_t nq, const pcr_xform* x,
                             double max_d2, int32_t* d_idx, double* d_d2);
PCR_HIDDEN int pcr_brute_last_fallback(pcr_ctx* ctx, unsigned int* out);
PCR_HIDDEN int pcr_brute_icp_pass(pcr_ctx* ctx, const pcr_index* idx, pcr_pt* q, int64_t nq, const pcr_xform* x,
                                  double max_d2, int write_back, double* d_moments);

constexpr int PCR_NMOM = 20;  // K, Sa[3], Sb[3], Sba[9], Saa, Sbb, Sd2, pad

// Device-resident state of the ICP loop (grid path): the last block of the accumulate kernel solves the Procrustes
// step, tests convergence (Registration/main.py:131-154) and leaves the next pass's transform here, so the host
// enqueues several iterations per synchronisation; passes enqueued behind a stop are no-ops.
struct __attribute__((aligned(16))) pcr_icp_dev_state {
    pcr_xform x;          // transform the NEXT pass applies = last solved increment (T_cur / T_ret of the host loop)
    double T_total[16];   // composed transform applied so far
    double R_last[9], t_last[3];
    double V[9];          // right singular vectors of the last Procrustes solve (warm start of the next one); identity at first
    double cost, mean_d2;
    long long n_assoc;
    int it;               // Procrustes solves performed
    int stop;             // no further pass may run (converged, max_iter reached, or too few associations)
    int converged;
    int status;
    int first;            // main.py:100,150: the first t_diff broadcasts (3,1)-(3,)
    int passes;           // association passes executed (the source has been transformed this many times)
    double r_diff[PCR_ICP_MAX_LOG], t_diff[PCR_ICP_MAX_LOG];
};
struct pcr_icp_loop_args {
    int max_iter, min_iter, compat, r_metric;
    double r_thres, t_thres;
};
// whole ICP loop on the device (grid index); fills res like the host loop of pcr_icp
// device -> pageable host memory through a pinned double buffer (large results: a pageable copy runs at ~4.5 GB/s)
PCR_HIDDEN int pcr_d2h_staged(pcr_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
// Open3D's voxel_down_sample of every scan of a chunk at once (pcr_voxel.hip).  d_xyz: the chunk's points, 3 x f32 each, scan behind scan
// (scans[s].first_pt .. + n_pts; mn / mx = the scan's bounding box).  Out, from the arena (the caller frees: sizes ng, ng, n_scans + 1):
// the records by row WITHIN their scan, the scan of every record, the first record of every scan (+ the total) -- the latter also on the host.
struct pcr_down_scan { unsigned int first_pt, n_pts; double mn[3], mx[3]; };
// prepare_dataset + execute_global_registration for a share of pairs at once (pcr_fpfh.hip): scans[] = rows of clouds[] that a pair of todo[] uses;
// T_init (16 doubles per row of pairs[]) gets the result of every pair with a valid hypothesis.  PCR_E_UNSUPPORTED: take the scans one by one.
PCR_HIDDEN int pcr_global_init_batch(pcr_ctx* ctx, const pcr_cloud_ref* clouds, int64_t n_clouds, const int64_t* scans, int64_t n_scans, const pcr_pair_ref* pairs,
                                     const int64_t* todo, int64_t n_todo, const pcr_global_params* g, double* T_init, int host_threads);
PCR_HIDDEN int pcr_voxel_downsample_scans(pcr_ctx* ctx, const float* d_xyz, int64_t n_pts, const pcr_down_scan* scans, int n_scans, double leaf, pcr_pt** down_out,
                                          unsigned int** vsid_out, unsigned int** scan_first_out, unsigned int* scan_first_host, int64_t* ng_out);
// small results (<= PCR_SMALL_D2H_BYTES) by a kernel writing into a pinned, device-mapped block: no copy engine (see pcr_core.hip).
// pcr_d2h_small synchronises the stream; _enqueue only launches (mapped_host_dst must be device-mapped pinned memory of the context)
constexpr size_t PCR_SMALL_D2H_BYTES = 16384;
PCR_HIDDEN int pcr_d2h_small(pcr_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes);
PCR_HIDDEN int pcr_wait_flag(pcr_ctx* ctx, double* flag_us);
PCR_HIDDEN int pcr_d2h_small_enqueue(pcr_ctx* ctx, void* mapped_host_dst, const void* dev_src, size_t bytes);
PCR_HIDDEN int pcr_grid_icp_loop(pcr_ctx* ctx, const pcr_index* idx, pcr_cloud* qc, const pcr_icp_params* params, const double T0[16],
                                 pcr_icp_result* res);

// ---------------------------------------------------------------------------------------------------------------
// Fused batch of registrations (pcr_batch.hip, Registration/main.py:190-216): one launch per STAGE for all pairs.
// The sources of all pairs live in ONE record array (pair p at [q_off, q_off + nq), Morton ordered, slots padded to
// whole blocks of 8 wave tiles), every pair has its own target grid, accumulator sets and loop state.
struct pcr_batch_pair {
    pcr_grid_view gv;            // the target's grid (pointers into the batch's pools)
    unsigned long long q_off;    // first source record = 32 x first wave tile
    long long nq;                // source points (<= slot size)
    double scale, inv_scale;     // 2^F, 2^-F of the pair's fixed-point moment accumulators
};
struct pcr_batch_pass_args {
    const pcr_batch_pair* pairs;       // device [n_pairs]
    int n_pairs;
    const unsigned int* tile_pair;     // device [n_tiles]: pair of every wave tile
    unsigned int n_tiles;              // multiple of 4
    pcr_pt* q;                         // all source records
    unsigned int* res_pos;             // [32 * n_tiles]
    void* prev_xyz;                    // [32 * n_tiles] x 24 B
    unsigned int* tile_cost;           // [n_tiles]
    unsigned long long* items;         // work queue [groups][cap][4], all-ones between passes
    unsigned long long* sync;          // queue words, zero between passes
    unsigned int cap;
    unsigned long long* acc;           // [n_pairs][sets][PCR_NMOM] fixed-point accumulators, zero between passes
    pcr_icp_dev_state* st;             // [n_pairs]
    unsigned int* running;             // [PCR_ICP_MAX_LOG + 1]: pairs still iterating after pass i
    pcr_icp_loop_args la;
    double max_d2;
};
// bytes of the queue / accumulator scratch a batch of that shape needs (items, sync + acc)
PCR_HIDDEN void pcr_grid_batch_scratch_bytes(unsigned int n_tiles, int n_pairs, size_t* items_bytes, size_t* acc_sync_bytes, size_t* sync_word, unsigned int* cap);
// states from T0 (device array [n_pairs][16]), queue and accumulators initialised: one launch
PCR_HIDDEN int pcr_grid_batch_init(pcr_ctx* ctx, const pcr_batch_pass_args* a, const double* d_T0);
// one ICP pass over every pair that has not stopped: tiles (publish) -> queue (drain) -> per-pair Procrustes step
PCR_HIDDEN int pcr_grid_batch_pass(pcr_ctx* ctx, const pcr_batch_pass_args* a, unsigned int pass_id);
// exactly the grid parameters pcr_grid_build derives from the target's box (shared so that a batch builds the same grid)
PCR_HIDDEN void pcr_grid_plan(const double lo[3], const double hi[3], long long n, double cell_in, double* cell_out, int* levels_out);
PCR_HIDDEN int pcr_morton_end_bit(const double lo[3], const double hi[3], double inv);
// fixed-point scale of the one-launch pass for a target box / source size / gate; false: cannot use the fused pass
PCR_HIDDEN bool pcr_pass_fixed_scale(const double lo[3], const double hi[3], long long nq, double max_d2, double* scale, double* inv_scale);

