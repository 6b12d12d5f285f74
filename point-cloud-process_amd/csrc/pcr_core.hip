// Context, device memory, cloud upload/download/transform, timers.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <atomic>
#include <memory>
#include "pcr_internal.h"

extern "C" {

const char* pcr_version(void) { return "pcr 0.1 (gfx950, f64)"; }

const char* pcr_strerror(int s) {
    switch (s) {
        case PCR_OK: return "ok";
        case PCR_E_TOO_FEW_ASSOC: return "ICP failed, cannot find enough associations!";
        case PCR_E_INVALID: return "invalid argument";
        case PCR_E_EMPTY: return "empty point cloud";
        case PCR_E_NOMEM: return "out of memory";
        case PCR_E_HIP: return "HIP runtime error";
        case PCR_E_NO_DEVICE: return "no HIP device (libpcr.so needs an AMD GPU; there is no CPU fallback)";
        case PCR_E_UNSUPPORTED: return "unsupported";
        case PCR_E_TOO_MANY_ITERS: return "max_iter exceeds PCR_ICP_MAX_LOG";
        default: return "unknown status";
    }
}

const char* pcr_last_error(const pcr_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int pcr_ctx_create(int device, pcr_ctx** out) {
    if (!out) return PCR_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return PCR_E_NO_DEVICE;
    if (device < 0 || device >= count) return PCR_E_INVALID;
    pcr_ctx* c = new pcr_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return PCR_E_HIP; }
    // the ICP loop waits for a 160-byte read-back every iteration: spin instead of sleeping on an interrupt
    hipSetDeviceFlags(hipDeviceScheduleSpin);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->cu_count = prop.multiProcessorCount;
        snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
        c->hbm_bytes = (int64_t)prop.totalGlobalMem;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return PCR_E_HIP; }
    hipEventCreate(&c->ev0);
    hipEventCreate(&c->ev1);
    hipEventCreate(&c->ev2);
    hipEventCreate(&c->ev3);
    c->h_pinned_bytes = 4096;
    c->zero_copy = getenv("PCR_NO_ZEROCOPY") == nullptr;
    if (const char* l = getenv("PCR_ICP_LANES")) c->icp_lanes = atoi(l) < 1 ? 1 : atoi(l);
    if (c->zero_copy && getenv("PCR_NO_HOSTSUM") == nullptr &&
        hipHostMalloc((void**)&c->h_slabs, sizeof(double) * PCR_NMOM * PCR_SLABS_PER_LANE * PCR_MAX_LANES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess)
        c->h_slabs = nullptr;
    if (hipHostMalloc((void**)&c->h_pinned, c->h_pinned_bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        delete c;
        return PCR_E_NOMEM;
    }
    // (device-mapped: small results are WRITTEN there by kernels, see pcr_d2h_small)
    if (hipHostMalloc(&c->h_state, 8192, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { delete c; return PCR_E_NOMEM; }
    if (hipHostMalloc(&c->h_small, PCR_SMALL_D2H_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { delete c; return PCR_E_NOMEM; }
    if (hipMalloc((void**)&c->d_counters, PCR_COUNTER_BYTES) != hipSuccess) { delete c; return PCR_E_NOMEM; }
    hipMemsetAsync(c->d_counters, 0, PCR_COUNTER_BYTES, c->stream);
    if (hipMalloc((void**)&c->d_cell_counts, 4 * (PCR_MAX_LEVELS * 64 + 64)) != hipSuccess) { delete c; return PCR_E_NOMEM; }
    hipMemsetAsync(c->d_cell_counts, 0, 4 * (PCR_MAX_LEVELS * 64 + 64), c->stream);
    if (getenv("PCR_DEBUG_STAMPS")) {
        hipMalloc((void**)&c->d_debug, sizeof(unsigned long long) << 20);
        hipMemsetAsync(c->d_debug, 0, sizeof(unsigned long long) << 20, c->stream);
    }
    // first arena now, not inside the first upload: a 256-MiB hipMalloc takes milliseconds under the driver's lock, and a
    // dozen fresh contexts paying it inside a timed batch cost more than the batch's kernels
    {
        void* warm = nullptr;
        if (pcr_dev_alloc(c, 256, &warm) == PCR_OK) pcr_dev_free(c, warm, 256);
        // same for the pinned upload staging buffer (hipHostMalloc pins pages under a lock: ~2 ms)
        if (hipHostMalloc(&c->h_stage, 4u << 20, hipHostMallocMapped) == hipSuccess) c->h_stage_bytes = 4u << 20;
        else c->h_stage = nullptr;
    }
    *out = c;
    return PCR_OK;
}

int pcr_ctx_destroy(pcr_ctx* c) {
    if (!c) return PCR_OK;
    hipSetDevice(c->device);
    pcr_sync(c->stream);
    for (void* a : c->arenas) hipFree(a);
    if (c->d_partials) hipFree(c->d_partials);
    if (c->d_counters) hipFree(c->d_counters);
    if (c->d_cell_counts) hipFree(c->d_cell_counts);
    if (c->h_pinned) hipHostFree(c->h_pinned);
    if (c->h_slabs) hipHostFree(c->h_slabs);
    if (c->h_state) hipHostFree(c->h_state);
    if (c->h_small) hipHostFree(c->h_small);
    if (c->h_big) hipHostFree(c->h_big);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_down) hipHostFree(c->h_down);
    if (c->h_init) hipHostFree(c->h_init);
    hipEventDestroy(c->ev0);
    hipEventDestroy(c->ev1);
    hipEventDestroy(c->ev2);
    hipEventDestroy(c->ev3);
    for (int i = 0; i < 5; ++i)
        if (c->pev[i]) hipEventDestroy(c->pev[i]);
    for (int l = 0; l < PCR_MAX_LANES; ++l)
        if (c->lane_stream[l]) hipStreamDestroy(c->lane_stream[l]);
    hipStreamDestroy(c->stream);
    delete c;
    return PCR_OK;
}

int pcr_ctx_sync(pcr_ctx* c) {
    if (!c) return PCR_E_INVALID;
    PCR_HIP(c, pcr_sync(c->stream));
    return PCR_OK;
}

int pcr_ctx_device_info(pcr_ctx* c, char* name256, int* cu, int64_t* hbm) {
    if (!c) return PCR_E_INVALID;
    if (name256) { strncpy(name256, c->name, 255); name256[255] = 0; }
    if (cu) *cu = c->cu_count;
    if (hbm) *hbm = c->hbm_bytes;
    return PCR_OK;
}

int pcr_debug_read(pcr_ctx* c, uint64_t* out, int64_t n) {
    if (!c || !out || !c->d_debug || n > (1 << 20)) return PCR_E_INVALID;
    PCR_HIP(c, pcr_sync(c->stream));
    PCR_HIP(c, hipMemcpy(out, c->d_debug, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
    return PCR_OK;
}

int pcr_profile_enable(pcr_ctx* c, int on) {
    if (!c) return PCR_E_INVALID;
    if (on && !c->pev[0])
        for (int i = 0; i < 5; ++i) PCR_HIP(c, hipEventCreate(&c->pev[i]));
    c->profile = on != 0;
    for (int i = 0; i < 4; ++i) c->prof_ms[i] = 0;
    c->prof_passes = 0;
    return PCR_OK;
}

int pcr_profile_read(pcr_ctx* c, double ms_out[4], int* passes_out) {
    if (!c || !ms_out) return PCR_E_INVALID;
    for (int i = 0; i < 4; ++i) ms_out[i] = c->prof_ms[i];
    if (passes_out) *passes_out = c->prof_passes;
    return PCR_OK;
}

}  // extern "C"

// Small device-to-host results WITHOUT the copy engine.  A hipMemcpyAsync of a few bytes or kilobytes behind the last kernel of a
// call goes through the runtime's DMA path, and on this pool that path stalls at random: 1 M-point registrations whose kernels
// had finished after their usual 13.5 ms (a HIP event behind the last kernel said so) returned after 30-57 ms, every 10th-30th
// call, because the 4.5-KB read-back of the loop state behind them took 16-44 ms (the batch path had met the
// same with its uploads in round 3).  So: a one-block kernel copies the bytes into a pinned, device-mapped block of the context,
// the host waits for the stream (polling) and copies them out.
__global__ void __launch_bounds__(256) d2h_small_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, unsigned int n) {
    if (((((size_t)src) | ((size_t)dst) | n) & 3u) == 0u) {
        const unsigned int* s4 = reinterpret_cast<const unsigned int*>(src);
        unsigned int* d4 = reinterpret_cast<unsigned int*>(dst);
        for (unsigned int i = threadIdx.x; i < n / 4; i += blockDim.x) d4[i] = s4[i];
    } else
        for (unsigned int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

// a sequence number into a mapped host word, behind everything the stream has done so far (system-scope release first): the host
// can wait for it by reading memory, without asking the runtime
__global__ void flag_kernel(unsigned long long* __restrict__ dst, unsigned long long seq) {
    __threadfence_system();
    __hip_atomic_store(dst, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Waits until everything enqueued on the context's stream so far has run: a flag kernel + polling of its mapped word (bounded), then
// the runtime's own wait, which by then returns at once unless the flag never came.  *flag_us (may be null) = microseconds until the
// flag was seen (-1: not within the bound).
int pcr_wait_flag(pcr_ctx* ctx, double* flag_us) {
    unsigned long long* const h_flag = (unsigned long long*)((char*)ctx->h_pinned + 1024);
    unsigned long long* d_flag = nullptr;
    PCR_HIP(ctx, hipHostGetDevicePointer((void**)&d_flag, h_flag, 0));
    const unsigned long long seq = ++ctx->flag_seq;
    hipLaunchKernelGGL(flag_kernel, dim3(1), dim3(1), 0, ctx->stream, d_flag, seq);
    PCR_HIP(ctx, hipGetLastError());
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (unsigned int spins = 0;; ++spins) {
        if (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
        if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
        __builtin_ia32_pause();
    }
    if (flag_us) *flag_us = seen ? std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() / 1e3 : -1.0;
    if (!seen) PCR_HIP(ctx, pcr_sync(ctx->stream));
    return PCR_OK;
}

int pcr_d2h_small_enqueue(pcr_ctx* ctx, void* mapped_host_dst, const void* dev_src, size_t bytes) {
    void* dp = nullptr;
    PCR_HIP(ctx, hipHostGetDevicePointer(&dp, mapped_host_dst, 0));
    hipLaunchKernelGGL(d2h_small_kernel, dim3(1), dim3(256), 0, ctx->stream, (const unsigned char*)dev_src, (unsigned char*)dp, (unsigned int)bytes);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

int pcr_d2h_small(pcr_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    if (bytes == 0) return PCR_OK;
    if (bytes > PCR_SMALL_D2H_BYTES || !ctx->h_small) {
        PCR_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, pcr_sync(ctx->stream));
        return PCR_OK;
    }
    const int rc = pcr_d2h_small_enqueue(ctx, ctx->h_small, dev_src, bytes);
    if (rc) return rc;
    PCR_HIP(ctx, pcr_sync(ctx->stream));
    memcpy(host_dst, ctx->h_small, bytes);
    return PCR_OK;
}

// Large device-to-host results (radius neighbour lists: 190 MB) through a pinned double buffer: the DMA of chunk i + 1 runs while
// the host copies chunk i into the caller's (pageable) array.  Small copies and contexts that cannot get the buffer take the
// plain path.  The stream is synchronised on return.
int pcr_d2h_staged(pcr_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    constexpr size_t HALF = 16u << 20;
    if (bytes < (48u << 20)) {   // (measured: 24 MB of ISS eigenvalues got slower through the double buffer, 190 MB of neighbour lists 2x faster)
        PCR_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, pcr_sync(ctx->stream));
        return PCR_OK;
    }
    if (!ctx->h_down) {
        if (hipHostMalloc(&ctx->h_down, 2 * HALF, hipHostMallocDefault) == hipSuccess) ctx->h_down_half = HALF;
        else { ctx->h_down = nullptr; (void)hipGetLastError(); }
    }
    if (!ctx->h_down) {
        PCR_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, pcr_sync(ctx->stream));
        return PCR_OK;
    }
    const size_t half = ctx->h_down_half;
    const size_t n_chunks = (bytes + half - 1) / half;
    char* const pin = (char*)ctx->h_down;
    auto chunk_bytes = [&](size_t c) { return c + 1 < n_chunks ? half : bytes - c * half; };
    // The host copy out of the pinned halves is the slower side of the pipeline (~10 GB/s per core against ~55 GB/s of DMA): W
    // persistent workers each copy their slice of every chunk as soon as the chunk has landed, and a half goes back to the DMA
    // when all W slices of the chunk it held are out.  (Three threads started per chunk, as before, left the DMA waiting.)
    constexpr int W = 8;
    std::atomic<size_t> landed{0};   // chunks whose DMA has completed
    std::atomic<int> failed{0};
    std::unique_ptr<std::atomic<int>[]> outs(new std::atomic<int>[n_chunks]);
    for (size_t c = 0; c < n_chunks; ++c) outs[c].store(0, std::memory_order_relaxed);
    auto worker = [&](int w) {
        for (size_t c = 0; c < n_chunks; ++c) {
            while (landed.load(std::memory_order_acquire) <= c) {
                if (failed.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const size_t nb = chunk_bytes(c), per = ((nb + W - 1) / W + 63) & ~(size_t)63;
            const size_t o = per * (size_t)w, len = o >= nb ? 0 : (o + per > nb ? nb - o : per);
            if (len) memcpy((char*)host_dst + c * half + o, pin + (c & 1) * half + o, len);
            outs[c].fetch_add(1, std::memory_order_release);
        }
    };
    std::thread th[W];
    int started = 0;
    try {
        for (int w = 0; w < W; ++w) { th[w] = std::thread(worker, w); ++started; }
    } catch (...) {
        // a refused thread (EAGAIN under a process limit) must not escape through the C ABI: stop the ones that run, plain copy instead
        failed.store(1, std::memory_order_relaxed);
        for (int w = 0; w < started; ++w) th[w].join();
        PCR_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, pcr_sync(ctx->stream));
        return PCR_OK;
    }
    hipError_t e = hipMemcpyAsync(pin, dev_src, chunk_bytes(0), hipMemcpyDeviceToHost, ctx->stream);
    for (size_t c = 0; c < n_chunks && e == hipSuccess; ++c) {
        e = pcr_sync(ctx->stream);   // chunk c is in its half
        if (e != hipSuccess) break;
        landed.store(c + 1, std::memory_order_release);
        if (c + 1 < n_chunks) {
            if (c >= 1)   // the other half held chunk c - 1
                while (outs[c - 1].load(std::memory_order_acquire) < W) std::this_thread::yield();
            e = hipMemcpyAsync(pin + ((c + 1) & 1) * half, (const char*)dev_src + (c + 1) * half, chunk_bytes(c + 1), hipMemcpyDeviceToHost, ctx->stream);
        }
    }
    if (e != hipSuccess) failed.store(1, std::memory_order_relaxed);
    for (int w = 0; w < W; ++w) th[w].join();
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PCR_E_HIP; }
    return PCR_OK;
}

extern "C" {

int pcr_ctx_set_shared(pcr_ctx* c, int shared) {
    if (!c) return PCR_E_INVALID;
    c->shared_device = shared ? 1 : 0;
    return PCR_OK;
}

int pcr_search_stats(pcr_ctx* c, int64_t out[4]) {
    if (!c || !out) return PCR_E_INVALID;
    hipSetDevice(c->device);
    unsigned int fb = 0;
    int rc = pcr_brute_last_fallback(c, &fb);
    if (rc) return rc;
    out[0] = fb;
    out[1] = c->arena_grow_count;
    out[2] = (int64_t)c->arena_grow_us;
    out[3] = (int64_t)c->arenas.size();
    return PCR_OK;
}

int pcr_icp_pass_log(pcr_ctx* c, int max_n, double* tile_us, double* drain_us, int64_t* items, int* n_out, double host_us[6]) {
    if (!c || !n_out || max_n < 0) return PCR_E_INVALID;
    const int n = c->pass_log_n < max_n ? c->pass_log_n : max_n;
    for (int k = 0; k < n; ++k) {
        if (tile_us) tile_us[k] = c->pass_log[0][k];
        if (drain_us) drain_us[k] = c->pass_log[1][k];
        if (items) items[k] = (int64_t)c->pass_log[2][k];
    }
    *n_out = c->pass_log_n;
    if (host_us)
        for (int k = 0; k < 6; ++k) host_us[k] = c->pass_host_us[k];
    return PCR_OK;
}

int pcr_timer_start(pcr_ctx* c) {
    if (!c) return PCR_E_INVALID;
    PCR_HIP(c, hipEventRecord(c->ev2, c->stream));
    return PCR_OK;
}

int pcr_timer_stop_ms(pcr_ctx* c, double* ms) {
    if (!c || !ms) return PCR_E_INVALID;
    PCR_HIP(c, hipEventRecord(c->ev3, c->stream));
    PCR_HIP(c, pcr_event_sync(c->ev3));
    float f = 0;
    PCR_HIP(c, hipEventElapsedTime(&f, c->ev2, c->ev3));
    *ms = f;
    return PCR_OK;
}

}  // extern "C"

void pcr_prof_mark(pcr_ctx* ctx, int k) {
    if (ctx->profile) hipEventRecord(ctx->pev[k], ctx->stream);
}

void pcr_prof_finish(pcr_ctx* ctx) {
    if (!ctx->profile) return;
    pcr_event_sync(ctx->pev[4]);
    for (int i = 0; i < 4; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->pev[i], ctx->pev[i + 1]) == hipSuccess) ctx->prof_ms[i] += ms;
    }
    ctx->prof_passes += 1;
}

// ------------------------------------------------------------------ memory
// Device memory comes from a few large arenas (256 MiB hipMalloc each, so every buffer of the path sits in
// 2-MiB-fragment mappings: with one hipMalloc per buffer the scattered 16..64-byte reads of the search kernels
// paid a TLB walk on most accesses).  Blocks are recycled through a size-matched free list; arenas are only
// released with the context.
static const size_t PCR_ARENA_BYTES = 256ull << 20;

int pcr_dev_alloc(pcr_ctx* ctx, size_t bytes, void** out) {
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~size_t(255);
    int best = -1;
    for (int i = 0; i < (int)ctx->free_list.size(); ++i) {
        size_t sz = ctx->free_list[i].sz;
        if (sz >= bytes && sz <= 2 * bytes + 4096 && (best < 0 || sz < ctx->free_list[best].sz)) best = i;
    }
    if (best >= 0) {
        *out = ctx->free_list[best].p;
        ctx->live.push_back(ctx->free_list[best]);  // remember the block's true capacity, not the request
        ctx->free_list.erase(ctx->free_list.begin() + best);
        return PCR_OK;
    }
    if (ctx->arenas.empty() || ctx->arena_used + bytes > ctx->arena_cap) {
        const size_t cap = bytes > PCR_ARENA_BYTES ? bytes : PCR_ARENA_BYTES;
        void* base = nullptr;
        const auto t_a = std::chrono::steady_clock::now();
        hipError_t e = hipMalloc(&base, cap);
        ctx->arena_grow_count += 1;   // (pcr_search_stats: a call that had to grow an arena -- milliseconds, on the host, with the stream idle -- says so)
        ctx->arena_grow_us += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_a).count() / 1e3;
        if (e != hipSuccess) {
            ctx->last_error = std::string("hipMalloc(arena): ") + hipGetErrorString(e);
            return PCR_E_NOMEM;
        }
        // what is left of the previous arena stays usable through the free list
        if (!ctx->arenas.empty() && ctx->arena_cap > ctx->arena_used)
            ctx->free_list.push_back({(char*)ctx->arenas.back() + ctx->arena_used, ctx->arena_cap - ctx->arena_used});
        ctx->arenas.push_back(base);
        ctx->arena_cap = cap;
        ctx->arena_used = 0;
    }
    *out = (char*)ctx->arenas.back() + ctx->arena_used;
    ctx->arena_used += bytes;
    ctx->live.push_back({*out, bytes});
    return PCR_OK;
}

void pcr_dev_free(pcr_ctx* ctx, void* p, size_t bytes) {
    if (!p) return;
    if (bytes == 0) bytes = 16;
    bytes = (bytes + 255) & ~size_t(255);
    // the block goes back with its true capacity (a larger recycled block would otherwise lose its tail for good)
    for (int i = (int)ctx->live.size() - 1; i >= 0; --i) {
        if (ctx->live[i].p == p) {
            bytes = ctx->live[i].sz;
            ctx->live[i] = ctx->live.back();
            ctx->live.pop_back();
            break;
        }
    }
    // Frees are stream-ordered with later allocations: every user of a block runs on ctx->stream, or on a lane stream
    // that the ICP pass synchronises before it frees (pcr_grid_icp_pass).
    ctx->free_list.push_back({p, bytes});
}

int pcr_ctx_lanes(pcr_ctx* ctx, int lanes) {
    for (int l = 0; l < lanes && l < PCR_MAX_LANES; ++l)
        if (!ctx->lane_stream[l]) PCR_HIP(ctx, hipStreamCreateWithFlags(&ctx->lane_stream[l], hipStreamNonBlocking));
    return PCR_OK;
}

int pcr_ensure_scratch(pcr_ctx* ctx, size_t partial_bytes) {
    if (ctx->d_partials_bytes >= partial_bytes) return PCR_OK;
    if (ctx->d_partials) {
        pcr_sync(ctx->stream);
        hipFree(ctx->d_partials);
        ctx->d_partials = nullptr;
        ctx->d_partials_bytes = 0;
    }
    size_t want = partial_bytes < (1u << 20) ? (1u << 20) : partial_bytes;
    PCR_HIP(ctx, hipMalloc((void**)&ctx->d_partials, want));
    ctx->d_partials_bytes = want;
    return PCR_OK;
}

void pcr_xform_from_T(const double* T, pcr_xform* x) {
    if (!T) {
        for (int i = 0; i < 9; ++i) x->r[i] = (i % 4 == 0) ? 1.0 : 0.0;
        x->t[0] = x->t[1] = x->t[2] = 0.0;
        return;
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) x->r[3 * i + j] = T[4 * i + j];
        x->t[i] = T[4 * i + 3];
    }
}

// ------------------------------------------------------------------ clouds
// order-preserving map of a binary64 value to an unsigned key (monotone over all non-NaN values; every key of a real value > 0)
__device__ static inline unsigned long long ordered_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static inline double ordered_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double v;
    memcpy(&v, &b, 8);
    return v;
}

// caller's records (S = float / double, `stride` elements apart, possibly in pinned host memory read over PCIe) -> 32-byte
// records, AND the cloud's exact bounding box on the way: six atomic maxima per block (of x, y, z and of -x, -y, -z, as ordered
// keys: zero is the identity), the block that finishes last hands them to the host (pinned, device-mapped words) and leaves
// the six words zero.  The upload ends with a stream synchronisation anyway, so the box costs the host nothing -- taking it in
// a host loop cost more than the whole transfer (195 us of 250 at 120 000 points).
template <typename S>
__global__ void __launch_bounds__(256)
expand_cloud_kernel(const S* __restrict__ in, long long n, long long stride, pcr_pt* __restrict__ out, unsigned long long* __restrict__ box,
                    unsigned long long* __restrict__ host_box) {
    __shared__ unsigned long long s_box[6];
    __shared__ int s_last;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (threadIdx.x < 6) s_box[threadIdx.x] = 0ull;
    double v[3] = {0.0, 0.0, 0.0};
    if (i < n) {
        const S* p = in + i * stride;
        pcr_pt o;
        o.x = v[0] = (double)p[0];
        o.y = v[1] = (double)p[1];
        o.z = v[2] = (double)p[2];
        o.id = i;
        out[i] = o;
    }
    __syncthreads();
    unsigned long long k[6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        k[a] = i < n ? ordered_key(v[a]) : 0ull;
        k[3 + a] = i < n ? ordered_key(-v[a]) : 0ull;
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const unsigned long long o = __shfl_xor(k[a], d, 64);
            k[a] = o > k[a] ? o : k[a];
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&s_box[a], k[a]);
    }
    __syncthreads();
    // (no fence: device-scope atomics are performed at the coherence point, and an agent-scope release is an L2 write-back on
    // this chip -- with __threadfence() here the kernel took 82 us instead of 30 at 120 000 points.  The six maxima are
    // acknowledged (vmcnt) before the same wave takes the ticket.)
    if (threadIdx.x < 6) __hip_atomic_fetch_max(&box[threadIdx.x], s_box[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(&box[6], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1ull ? 1 : 0;   // ticket
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 7) {
        const unsigned long long w = __hip_atomic_load(&box[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 6) host_box[threadIdx.x] = w;
        box[threadIdx.x] = 0ull;
    }
}

__global__ void pack_xyz_kernel(const pcr_pt* __restrict__ in, long long n, double* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pcr_pt p = in[i];
    // records may have been reordered on the device (Morton sort); id is the caller's row
    out[3 * p.id + 0] = p.x;
    out[3 * p.id + 1] = p.y;
    out[3 * p.id + 2] = p.z;
}

// p' = R p + t, evaluated as ((r0*x + r1*y) + r2*z) + t with no FMA contraction.
__global__ void transform_cloud_kernel(pcr_pt* __restrict__ pts, long long n, pcr_xform x) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pcr_pt p = pts[i];
    double nx = ((x.r[0] * p.x + x.r[1] * p.y) + x.r[2] * p.z) + x.t[0];
    double ny = ((x.r[3] * p.x + x.r[4] * p.y) + x.r[5] * p.z) + x.t[1];
    double nz = ((x.r[6] * p.x + x.r[7] * p.y) + x.r[8] * p.z) + x.t[2];
    p.x = nx;
    p.y = ny;
    p.z = nz;
    pts[i] = p;
}

template <typename S>
static int upload_impl(pcr_ctx* ctx, const S* xyz, int64_t n, int64_t stride, pcr_cloud** out) {
    if (!ctx || !out) return PCR_E_INVALID;
    *out = nullptr;
    if (n < 0 || stride < 3 || (n > 0 && !xyz)) return PCR_E_INVALID;
    if (n == 0) return PCR_E_EMPTY;
    if (n > 0x7fffffffll) return PCR_E_UNSUPPORTED;
    hipSetDevice(ctx->device);
    static const bool timing = getenv("PCR_UPLOAD_TIMING") != nullptr;   // diagnostics: microseconds per phase to stderr
    const auto t_0 = std::chrono::steady_clock::now();
    pcr_cloud* c = new pcr_cloud();
    c->n = n;
    int rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * n, (void**)&c->d);
    if (rc != PCR_OK) { delete c; return rc; }
    size_t raw_elems = (size_t)(n - 1) * stride + 3;
    // Caller buffers are pageable: the runtime stages such copies through its own pinned buffers under a lock (~4.5 GB/s in
    // aggregate, however many contexts copy at once), and even copies from pinned memory are submitted through one queue per
    // device (12 contexts uploading at once each waited 12x as long).  So: a plain memcpy into this context's own pinned,
    // device-mapped buffer, and the expand kernel reads the 12 useful bytes of every record straight from there over PCIe.
    const size_t raw_bytes = raw_elems * sizeof(S);
    const S* d_src = nullptr;     // what the expand kernel reads
    void* d_raw = nullptr;
    if (raw_bytes <= (64u << 20)) {
        if (ctx->h_stage_bytes < raw_bytes) {
            if (ctx->h_stage) hipHostFree(ctx->h_stage);
            ctx->h_stage = nullptr;
            ctx->h_stage_bytes = 0;
            size_t want = raw_bytes < (4u << 20) ? (4u << 20) : raw_bytes;
            if (hipHostMalloc(&ctx->h_stage, want, hipHostMallocMapped) == hipSuccess) ctx->h_stage_bytes = want;
            else ctx->h_stage = nullptr;
        }
        if (ctx->h_stage) {
            // (a large cloud: one thread copies ~10 GB/s out of pageable memory -- 2.5 ms for the 24 MB of a million binary64 points, more than
            // everything the device then does with them; a few threads share the copy.  Small clouds: a thread costs more than their copy.)
            const size_t PAR_BYTES = 8u << 20;
            bool copied = false;
            if (raw_bytes >= PAR_BYTES) {
                const int nt = (int)(raw_bytes / (4u << 20) < 8 ? raw_bytes / (4u << 20) : 8);
                std::vector<std::thread> pool;
                const size_t per = ((raw_bytes / nt) + 4095) & ~(size_t)4095;
                try {
                    for (int t = 1; t < nt; ++t) {
                        const size_t b0 = per * t, b1 = b0 + per < raw_bytes ? b0 + per : raw_bytes;
                        if (b0 < raw_bytes) pool.emplace_back([=]() { memcpy((char*)ctx->h_stage + b0, (const char*)xyz + b0, b1 - b0); });
                    }
                    memcpy(ctx->h_stage, xyz, per < raw_bytes ? per : raw_bytes);
                    copied = true;
                } catch (...) { copied = false; }   // (a refused thread: the plain copy below redoes it all)
                for (auto& th : pool) th.join();
            }
            if (!copied) memcpy(ctx->h_stage, xyz, raw_bytes);
            void* dp = nullptr;
            if (hipHostGetDevicePointer(&dp, ctx->h_stage, 0) == hipSuccess) d_src = (const S*)dp;
        }
    }
    if (!d_src) {   // very large clouds (or no pinned memory): device staging buffer + the runtime's own copy
        rc = pcr_dev_alloc(ctx, raw_bytes, &d_raw);
        if (rc != PCR_OK) { pcr_dev_free(ctx, c->d, sizeof(pcr_pt) * n); delete c; return rc; }
        PCR_HIP(ctx, hipMemcpyAsync(d_raw, xyz, raw_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_src = (const S*)d_raw;
    }
    const auto t_1 = std::chrono::steady_clock::now();
    int block = 256;
    int grid = (int)((n + block - 1) / block);
    // bounding box: d_counters words 16..29 (six 64-bit maxima + ticket, zero between uploads) -> h_pinned bytes 512..559
    unsigned long long* const h_box = (unsigned long long*)((char*)ctx->h_pinned + 512);
    unsigned long long* h_box_dev = nullptr;
    PCR_HIP(ctx, hipHostGetDevicePointer((void**)&h_box_dev, h_box, 0));
    hipLaunchKernelGGL(expand_cloud_kernel<S>, dim3(grid), dim3(block), 0, ctx->stream, d_src, (long long)n, (long long)stride, c->d,
                       (unsigned long long*)(ctx->d_counters + 16), h_box_dev);
    PCR_HIP(ctx, hipGetLastError());
    // the staging buffer is reused by the next upload (and an unstaged copy reads caller-owned memory): finish before returning
    PCR_HIP(ctx, pcr_sync(ctx->stream));
    {
        // (S -> double is exact, so this is the box a binary64 reduction over the records gives; a NaN or an infinity shows up as
        // a non-finite corner: no box then, the device reduction of pcr_cloud_bbox decides later, as before)
        double lo[3], hi[3];
        bool finite = true;
        for (int k = 0; k < 3; ++k) {
            hi[k] = ordered_value(h_box[k]);
            lo[k] = -ordered_value(h_box[3 + k]);
            finite = finite && std::isfinite(lo[k]) && std::isfinite(hi[k]);
        }
        if (finite) {
            c->has_bbox = true;
            for (int k = 0; k < 3; ++k) { c->lo[k] = lo[k] == 0.0 ? 0.0 : lo[k]; c->hi[k] = hi[k]; }   // (-(+0) = -0: keep +0)
        }
    }
    if (timing) {
        const auto t_2 = std::chrono::steady_clock::now();
        fprintf(stderr, "upload n=%lld stride=%lld: host copy + box %.1f us, expand kernel + sync %.1f us\n", (long long)n, (long long)stride,
                std::chrono::duration_cast<std::chrono::nanoseconds>(t_1 - t_0).count() / 1e3, std::chrono::duration_cast<std::chrono::nanoseconds>(t_2 - t_1).count() / 1e3);
    }
    if (d_raw) pcr_dev_free(ctx, d_raw, raw_bytes);
    *out = c;
    return PCR_OK;
}

__global__ void cloud_rows_kernel(const pcr_pt* __restrict__ in, long long n, pcr_pt* __restrict__ out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pcr_pt p = in[i];
    out[p.id] = p;
}

int pcr_cloud_rows(pcr_ctx* ctx, const pcr_cloud* c, pcr_pt* d_out) {
    hipLaunchKernelGGL(cloud_rows_kernel, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, ctx->stream, (const pcr_pt*)c->d, (long long)c->n, d_out);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

extern "C" {

int pcr_cloud_upload_f32(pcr_ctx* ctx, const float* xyz, int64_t n, int64_t stride, pcr_cloud** out) {
    return upload_impl<float>(ctx, xyz, n, stride, out);
}

int pcr_cloud_upload_f64(pcr_ctx* ctx, const double* xyz, int64_t n, int64_t stride, pcr_cloud** out) {
    return upload_impl<double>(ctx, xyz, n, stride, out);
}

int pcr_cloud_download_f64(pcr_ctx* ctx, const pcr_cloud* c, double* out) {
    if (!ctx || !c || !out) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    double* d_tmp = nullptr;
    int rc = pcr_dev_alloc(ctx, sizeof(double) * 3 * c->n, (void**)&d_tmp);
    if (rc != PCR_OK) return rc;
    int block = 256;
    int grid = (int)((c->n + block - 1) / block);
    hipLaunchKernelGGL(pack_xyz_kernel, dim3(grid), dim3(block), 0, ctx->stream, (const pcr_pt*)c->d, (long long)c->n, d_tmp);
    PCR_HIP(ctx, hipGetLastError());
    PCR_HIP(ctx, hipMemcpyAsync(out, d_tmp, sizeof(double) * 3 * c->n, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, pcr_sync(ctx->stream));
    pcr_dev_free(ctx, d_tmp, sizeof(double) * 3 * c->n);
    return PCR_OK;
}

int64_t pcr_cloud_size(const pcr_cloud* c) { return c ? c->n : 0; }

int pcr_cloud_free(pcr_ctx* ctx, pcr_cloud* c) {
    if (!c) return PCR_OK;
    if (!ctx) return PCR_E_INVALID;
    pcr_dev_free(ctx, c->d, sizeof(pcr_pt) * c->n);
    delete c;
    return PCR_OK;
}

int pcr_cloud_transform(pcr_ctx* ctx, pcr_cloud* c, const double T[16]) {
    if (!ctx || !c || !T) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    pcr_xform x;
    pcr_xform_from_T(T, &x);
    int block = 256;
    int grid = (int)((c->n + block - 1) / block);
    c->has_bbox = false;
    hipLaunchKernelGGL(transform_cloud_kernel, dim3(grid), dim3(block), 0, ctx->stream, c->d, (long long)c->n, x);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

}  // extern "C"
