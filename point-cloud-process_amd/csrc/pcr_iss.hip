// ISS keypoints -- Keypoint_detection_ISS/ISS.py:35-73.
//   for every point p_i: N(p_i) = points within `radius` (inclusive, p_i itself included, ISS.py:43);
//   weight of neighbour p_j = 1 / |N(p_j)| (ISS.py:49); scatter = sum w (p_j-p_i)(p_j-p_i)^T / sum w
//   (ISS.py:50-52); eigenvalues sorted descending (ISS.py:53); candidate iff l2/l1 < g21 and l3/l2 < g32
//   (ISS.py:55); non-maximum suppression by l3 with radius nms_radius, stopping once MORE than
//   max_keypoints were taken (ISS.py:59-73).
// A grid with cell = radius is built over the cloud: the neighbourhood of a point is inside the
// 3x3x3 block of its cell.  Two passes, 8 lanes per point, lane-owned cells:
//   pass 1  |N(p_i)|            (16 B point + 4 B count per point: HBM-trivial, cache-resident gathers)
//   pass 2  weighted scatter + symmetric 3x3 eigenvalues (Jacobi, binary64) -> 3 doubles per point
// (Wave tiles -- 64 consecutive points of the sorted cloud staging the cells of their common box into LDS once, the way the
// batched k-NN does -- were built and measured here: 0.94 + 1.89 ms at 1 M points, k = 38, against 0.55 + 1.17 ms: both
// passes are bound by the binary64 distance evaluations, and a common box holds 1.5-2 x the candidates of a point's own 27
// cells.)  The NMS is a short sequential host loop over the candidates in lambda_3 order.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "pcr_grid_dev.h"

constexpr int IG = 8;  // lanes per point

template <class F>
__device__ static inline void iss_visit_block(const pcr_grid_view& gv, double ax, double ay, double az, int gl, F& f) {
    bool clamped = false;
    const int cx = cell_coord(ax, gv.lo[0], gv.inv_cell0, &clamped);
    const int cy = cell_coord(ay, gv.lo[1], gv.inv_cell0, &clamped);
    const int cz = cell_coord(az, gv.lo[2], gv.inv_cell0, &clamped);
#pragma unroll
    for (int i = 0; i < (27 + IG - 1) / IG; ++i) {
        const int n = gl + i * IG;
        if (n < 27) {
            const unsigned int nx = (unsigned int)(cx + n % 3 - 1), ny = (unsigned int)(cy + (n / 3) % 3 - 1), nz = (unsigned int)(cz + n / 9 - 1);
            if (nx <= (unsigned int)PCR_COORD_MAX && ny <= (unsigned int)PCR_COORD_MAX && nz <= (unsigned int)PCR_COORD_MAX) {
                unsigned int s, e;
                if (lookup_cell(gv.table[0], gv.mask[0], nx, ny, nz, &s, &e))
                    for (unsigned int j0 = s; j0 < e; j0 += 4) {   // four records per trip, requested together
                        pcr_pt rec[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {   // (assigned on every path: conditionally unassigned records become loop-carried registers)
                            rec[u] = pcr_pt{0.0, 0.0, 0.0, 0};
                            if (j0 + u < e) rec[u] = gv.pts[j0 + u];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (j0 + u < e) f(rec[u]);
                    }
            }
        }
    }
}

__global__ void __launch_bounds__(256) iss_count_kernel(pcr_grid_view gv, long long n, double r2_in, int* __restrict__ counts /* by row id */) {
    const int gl = threadIdx.x % IG;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / IG;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    int c = 0;
    auto f = [&](const pcr_pt& b) { c += dist2(p.x, p.y, p.z, b) <= r2_in; };
    iss_visit_block(gv, p.x, p.y, p.z, gl, f);
#pragma unroll
    for (int off = IG / 2; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (gl == 0) counts[p.id] = c;
}

// a point that passed the ratio tests: lambda_3, caller row, position in the index's sorted order (its coordinates are gv.pts[pos])
struct __attribute__((aligned(16))) iss_cand {
    double l3;
    int id;
    unsigned int pos;
};
// ISS.py:59-61 visits the candidates by descending lambda_3, ties in input order
struct iss_cand_before {
    __host__ __device__ bool operator()(const iss_cand& a, const iss_cand& b) const { return a.l3 > b.l3 || (a.l3 == b.l3 && a.id < b.id); }
};

// eigenvalues of a symmetric 3x3 matrix (cyclic Jacobi), descending
__device__ static inline void sym3_eigenvalues(double a00, double a01, double a02, double a11, double a12, double a22, double ev[3]) {
    double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * diag) break;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            const int r = 3 - p - q;
            const double app = A[p][p], aqq = A[q][q], arp = A[r][p], arq = A[r][q];
            A[p][p] = app - t * apq;
            A[q][q] = aqq + t * apq;
            A[p][q] = A[q][p] = 0.0;
            A[r][p] = A[p][r] = c * arp - s * arq;
            A[r][q] = A[q][r] = s * arp + c * arq;
        }
    }
    double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
    if (e0 < e1) { double t = e0; e0 = e1; e1 = t; }
    if (e1 < e2) { double t = e1; e1 = e2; e2 = t; }
    if (e0 < e1) { double t = e0; e0 = e1; e1 = t; }
    ev[0] = e0; ev[1] = e1; ev[2] = e2;
}

__global__ void __launch_bounds__(256) iss_cov_kernel(pcr_grid_view gv, long long n, double r2_in, const int* __restrict__ counts,
                                                      double* __restrict__ lambdas /* by row id, (n,3) */, double gamma21, double gamma32,
                                                      int* __restrict__ cand /* row ids passing the ratio tests, any order */,
                                                      unsigned int* __restrict__ cand_count, struct iss_cand* __restrict__ cand_rec /* or the records */) {
    const int gl = threadIdx.x % IG;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / IG;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    double m[7] = {0, 0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz, denom
    auto f = [&](const pcr_pt& b) {
        const double dx = b.x - p.x, dy = b.y - p.y, dz = b.z - p.z;
        if ((dx * dx + dy * dy) + dz * dz <= r2_in) {
            const double w = 1.0 / (double)counts[b.id];
            m[0] += w * (dx * dx); m[1] += w * (dx * dy); m[2] += w * (dx * dz);
            m[3] += w * (dy * dy); m[4] += w * (dy * dz); m[5] += w * (dz * dz);
            m[6] += w;
        }
    };
    iss_visit_block(gv, p.x, p.y, p.z, gl, f);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double v = m[k];
#pragma unroll
        for (int off = IG / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        m[k] = v;
    }
    if (gl == 0) {
        double ev[3];
        const double inv = 1.0 / m[6];  // the point itself is a neighbour: denom > 0
        sym3_eigenvalues(m[0] * inv, m[1] * inv, m[2] * inv, m[3] * inv, m[4] * inv, m[5] * inv, ev);
        lambdas[3 * p.id + 0] = ev[0];
        lambdas[3 * p.id + 1] = ev[1];
        lambdas[3 * p.id + 2] = ev[2];
        // ISS.py:55-57: candidate when both eigenvalue ratios pass (NaN ratios of degenerate neighbourhoods fail, like the
        // reference's comparisons do); the host orders the (unordered) list, so the append order does not matter
        if ((cand || cand_rec) && ev[1] / ev[0] < gamma21 && ev[2] / ev[1] < gamma32) {
            const unsigned int slot = atomicAdd(cand_count, 1u);
            if (cand_rec) cand_rec[slot] = iss_cand{ev[2], (int)p.id, (unsigned int)i};
            else cand[slot] = (int)p.id;
        }
    }
}


// Non-maximum suppression (ISS.py:59-73) on the device, ONE block: the candidates come sorted (descending lambda_3, ties by row);
// the reference keeps a candidate unless an already kept keypoint lies within nms_radius, and stops once MORE than max_keypoints
// were taken.  256 candidates per round: every thread tests its candidate against the keypoints kept so far (LDS); then, in
// order, the first one still alive is kept and the rest of the round is tested against it, until nobody is left alive.
constexpr int ISS_NMS_MAX = 1024;   // keypoints the block can hold in LDS (more: the host loop below)
__global__ void __launch_bounds__(256)
iss_nms_kernel(const pcr_pt* __restrict__ pts, const iss_cand* __restrict__ cand, unsigned int n_cand, double nms_radius, int max_keypoints,
               int* __restrict__ keypoints_out /* mapped host memory */, int* __restrict__ n_out) {
    __shared__ double kx[ISS_NMS_MAX + 1], ky[ISS_NMS_MAX + 1], kz[ISS_NMS_MAX + 1];
    __shared__ int s_first, s_taken;
    if (threadIdx.x == 0) s_taken = 0;
    __syncthreads();
    for (unsigned int base = 0; base < n_cand; base += 256) {
        const unsigned int j = base + threadIdx.x;
        bool alive = j < n_cand;
        double x = 0, y = 0, z = 0;
        int id = 0;
        if (alive) {
            const iss_cand c = cand[j];
            const pcr_pt p = pts[c.pos];
            x = p.x; y = p.y; z = p.z; id = c.id;
        }
        int checked = 0;   // keypoints this thread has tested its candidate against
        for (;;) {
            const int taken = s_taken;
            for (; alive && checked < taken; ++checked) {
                const double dx = kx[checked] - x, dy = ky[checked] - y, dz = kz[checked] - z;
                const double d = sqrt((dx * dx + dy * dy) + dz * dz);   // the expression of the radius query (pcr_knn.hip)
                if (!(d > nms_radius)) alive = false;
            }
            checked = taken;
            __syncthreads();
            if (threadIdx.x == 0) s_first = 0x7fffffff;
            __syncthreads();
            if (alive) atomicMin(&s_first, (int)threadIdx.x);
            __syncthreads();
            const int first = s_first;
            if (first == 0x7fffffff) break;          // nobody of this round is left: next round
            if ((int)threadIdx.x == first) {
                kx[taken] = x; ky[taken] = y; kz[taken] = z;
                keypoints_out[taken] = id;
                s_taken = taken + 1;
                alive = false;
            }
            __syncthreads();
            if (s_taken > max_keypoints) break;      // ISS.py:72-73: stops once MORE than iss_count were taken
        }
        __syncthreads();
        if (s_taken > max_keypoints) break;
    }
    if (threadIdx.x == 0) *n_out = s_taken;
}

__global__ void iss_cand_l3_kernel(const double* __restrict__ lam, const int* __restrict__ cand, unsigned int m, double* __restrict__ out) {
    const unsigned int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) out[j] = lam[3 * (size_t)cand[j] + 2];
}

__global__ void iss_gather_kernel(const pcr_pt* __restrict__ rows, const int* __restrict__ ids, int m, pcr_pt* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) out[j] = rows[ids[j]];
}

extern "C" int pcr_iss(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, double gamma21, double gamma32, double nms_radius,
                       int max_keypoints, double* lambdas_out, int32_t* counts_out, int32_t* keypoints_out, int* n_keypoints_out) {
    if (!ctx || !cloud || !(radius > 0)) return PCR_E_INVALID;
    if (!lambdas_out && !(keypoints_out && n_keypoints_out)) return PCR_E_INVALID;   // nothing asked for
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    const int64_t n = cloud->n;
    pcr_index* idx = nullptr;
    int rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, radius * (1.0 + 1e-9), &idx);  // block of 27 cells covers the ball with rounding slack
    if (rc) return rc;
    // the index may have coarsened the cell (huge extents): the 3x3x3 block then still covers the radius
    if (idx->cell < radius * (1.0 - 1e-12)) { pcr_index_free(ctx, idx); idx = nullptr; return PCR_E_UNSUPPORTED; }
    // (every scratch block and the index go back on every return path: ADVICE r3)
    struct index_guard { pcr_ctx* c; pcr_index*& i; ~index_guard() { if (i) { pcr_index_free(c, i); i = nullptr; } } } idx_guard{ctx, idx};
    pcr_dev_block b_counts(ctx), b_lam(ctx);
    if ((rc = b_counts.alloc(sizeof(int) * n)) || (rc = b_lam.alloc(sizeof(double) * 3 * n))) return rc;
    int* const d_counts = b_counts.as<int>();
    double* const d_lam = b_lam.as<double>();
    const unsigned grid = (unsigned)((n * IG + 255) / 256);
    const bool want_kp = keypoints_out && n_keypoints_out;
    static const bool host_nms = getenv("PCR_ISS_HOST_NMS") != nullptr;   // A/B: the host loop over a heap of the candidates
    const bool dev_nms = want_kp && !host_nms && max_keypoints >= 0 && max_keypoints < ISS_NMS_MAX;
    int* d_cand = nullptr;
    pcr_dev_block b_rec(ctx), b_rec2(ctx), b_tmp(ctx);
    unsigned int* d_cand_count = ctx->d_counters + 124;
    if (want_kp) {
        if (dev_nms) { if ((rc = b_rec.alloc(sizeof(iss_cand) * n))) return rc; }
        else if ((rc = pcr_dev_alloc(ctx, sizeof(int) * n, (void**)&d_cand))) return rc;
        PCR_HIP(ctx, hipMemsetAsync(d_cand_count, 0, sizeof(unsigned int), ctx->stream));
    }
    // "within radius" is `not (sqrt(d2) > radius)` (the radius query's expression, pcr_knn.hip).  sqrt is monotone and correctly
    // rounded, so that is `d2 <= r2_in` with r2_in the LARGEST binary64 whose square root does not exceed the radius: same
    // decisions bit for bit, no square root per candidate (it was most of the VALU work of both passes).
    double r2_in = radius * radius;
    while (sqrt(r2_in) > radius) r2_in = nextafter(r2_in, 0.0);
    while (sqrt(nextafter(r2_in, INFINITY)) <= radius) r2_in = nextafter(r2_in, INFINITY);
    hipLaunchKernelGGL(iss_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (long long)n, r2_in, d_counts);
    hipLaunchKernelGGL(iss_cov_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (long long)n, r2_in, (const int*)d_counts, d_lam, gamma21,
                       gamma32, d_cand, d_cand_count, b_rec.as<iss_cand>());
    PCR_HIP(ctx, hipGetLastError());
    // (24 MB + 4 MB at 1 M points: through the pinned double buffer, a pageable copy runs at ~4.5 GB/s.  Both are optional: a
    // caller that only wants the keypoints gets the candidates' lambda_3 in a compact list instead -- the copy was most of the
    // call's wall time)
    if (lambdas_out && (rc = pcr_d2h_staged(ctx, lambdas_out, d_lam, sizeof(double) * 3 * (size_t)n))) return rc;
    if (counts_out && (rc = pcr_d2h_staged(ctx, counts_out, d_counts, sizeof(int) * (size_t)n))) return rc;
    unsigned int n_cand = 0;
    if (want_kp) { if ((rc = pcr_d2h_small(ctx, &n_cand, d_cand_count, sizeof(unsigned int)))) return rc; }
    else PCR_HIP(ctx, pcr_sync(ctx->stream));
    b_counts.free_now();
    rc = PCR_OK;
    if (dev_nms) {
        // candidates ordered on the device (one merge sort with the reference's order as the comparison), suppression by one block,
        // the <= max_keypoints + 1 row ids written straight into pinned memory: nothing but the count and the keypoints crosses PCIe
        b_lam.free_now();
        int taken = 0;
        if (n_cand) {
            size_t tb = 0;
            iss_cand* recs = b_rec.as<iss_cand>();
            hipError_t e = rocprim::merge_sort(nullptr, tb, recs, recs, (size_t)n_cand, iss_cand_before(), ctx->stream);
            if (e == hipSuccess && (rc = b_tmp.alloc(tb > 0 ? tb : 16)) == PCR_OK) e = rocprim::merge_sort(b_tmp.p, tb, recs, recs, (size_t)n_cand, iss_cand_before(), ctx->stream);
            if (e != hipSuccess) { ctx->last_error = std::string("pcr_iss (candidate sort): ") + hipGetErrorString(e); rc = PCR_E_HIP; }
            if (rc == PCR_OK) {
                // landing block: [n_out][keypoints...] in the context's mapped small block
                int* h_out = (int*)ctx->h_small;
                int* d_out = nullptr;
                if ((size_t)(max_keypoints + 2) * sizeof(int) > PCR_SMALL_D2H_BYTES || hipHostGetDevicePointer((void**)&d_out, h_out, 0) != hipSuccess) rc = PCR_E_HIP;
                if (rc == PCR_OK) {
                    hipLaunchKernelGGL(iss_nms_kernel, dim3(1), dim3(256), 0, ctx->stream, idx->view.pts, (const iss_cand*)recs, n_cand, nms_radius, max_keypoints, d_out + 1, d_out);
                    if (hipGetLastError() != hipSuccess) rc = PCR_E_HIP;
                    if (rc == PCR_OK) rc = pcr_wait_flag(ctx, nullptr);
                    if (rc == PCR_OK) {
                        taken = h_out[0];
                        for (int k = 0; k < taken; ++k) keypoints_out[k] = h_out[1 + k];
                    }
                }
            }
        }
        *n_keypoints_out = taken;
        return rc;
    }
    std::vector<double> cand_l3(n_cand);
    if (want_kp && n_cand) {   // lambda_3 of the candidates, in list order
        double* d_l3 = nullptr;
        if ((rc = pcr_dev_alloc(ctx, sizeof(double) * n_cand, (void**)&d_l3)) == PCR_OK) {
            hipLaunchKernelGGL(iss_cand_l3_kernel, dim3((n_cand + 255) / 256), dim3(256), 0, ctx->stream, (const double*)d_lam, (const int*)d_cand, n_cand, d_l3);
            hipError_t e = hipMemcpyAsync(cand_l3.data(), d_l3, sizeof(double) * n_cand, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = pcr_sync(ctx->stream);
            if (e != hipSuccess) { ctx->last_error = std::string("pcr_iss: ") + hipGetErrorString(e); rc = PCR_E_HIP; }
            pcr_dev_free(ctx, d_l3, sizeof(double) * n_cand);
        }
    }
    b_lam.free_now();
    if (rc) { if (d_cand) pcr_dev_free(ctx, d_cand, sizeof(int) * n); return rc; }
    if (want_kp) {
        // Non-maximum suppression (ISS.py:59-73).  The reference walks the candidates in descending lambda_3 (stable: ties
        // in input order) and, for each one still alive, keeps it and removes everything within nms_radius of it.  A
        // candidate is therefore dropped exactly when an ALREADY KEPT keypoint lies within nms_radius: at most
        // max_keypoints + 1 distance checks per candidate on the host, no radius queries; and only the head of the order is
        // ever visited, so the candidates sit in a heap instead of being sorted.
        struct cand_t { int id; double l3; };
        std::vector<cand_t> cand(n_cand);
        if (n_cand) {
            std::vector<int> ids(n_cand);
            hipError_t e = hipMemcpyAsync(ids.data(), d_cand, sizeof(int) * n_cand, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = pcr_sync(ctx->stream);
            if (e != hipSuccess) { ctx->last_error = std::string("pcr_iss: ") + hipGetErrorString(e); rc = PCR_E_HIP; }
            for (unsigned int j = 0; j < n_cand; ++j) cand[j] = cand_t{ids[j], cand_l3[j]};
        }
        auto later = [&](const cand_t& a, const cand_t& b) {  // heap order: a comes AFTER b
            return a.l3 < b.l3 || (a.l3 == b.l3 && a.id > b.id);
        };
        std::make_heap(cand.begin(), cand.end(), later);
        // coordinates of the visited candidates only: fetched in small batches by row id
        pcr_pt *d_rows = nullptr, *d_batch = nullptr;
        int* d_ids = nullptr;
        constexpr size_t BATCH = 256;
        if ((rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * n, (void**)&d_rows)) == PCR_OK) rc = pcr_cloud_rows(ctx, cloud, d_rows);
        if (rc == PCR_OK) rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * BATCH, (void**)&d_batch);
        if (rc == PCR_OK) rc = pcr_dev_alloc(ctx, sizeof(int) * BATCH, (void**)&d_ids);
        std::vector<pcr_pt> kept;
        int taken = 0;
        size_t heap_n = cand.size();
        while (rc == PCR_OK && heap_n > 0) {
            // pop the next batch of candidates in order and read their records with one copy each (they are few)
            const size_t batch = std::min<size_t>(heap_n, BATCH);
            std::vector<int> ids(batch);
            for (size_t j = 0; j < batch; ++j) {
                std::pop_heap(cand.begin(), cand.begin() + heap_n, later);
                ids[j] = cand[--heap_n].id;
            }
            std::vector<pcr_pt> recs(batch);
            // (errors leave through the clean-up below: the scratch blocks and the index go back on every path)
            hipError_t e = hipMemcpyAsync(d_ids, ids.data(), sizeof(int) * batch, hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(iss_gather_kernel, dim3(1), dim3((unsigned)BATCH), 0, ctx->stream, (const pcr_pt*)d_rows, (const int*)d_ids, (int)batch, d_batch);
                e = hipMemcpyAsync(recs.data(), d_batch, sizeof(pcr_pt) * batch, hipMemcpyDeviceToHost, ctx->stream);
            }
            if (e == hipSuccess) e = pcr_sync(ctx->stream);
            if (e != hipSuccess) { ctx->last_error = std::string("pcr_iss (suppression): ") + hipGetErrorString(e); rc = PCR_E_HIP; break; }
            bool done = false;
            for (size_t j = 0; j < batch && !done; ++j) {
                const pcr_pt& c = recs[j];
                bool alive = true;
                for (const pcr_pt& k : kept) {
                    const double dx = k.x - c.x, dy = k.y - c.y, dz = k.z - c.z;
                    const double d = sqrt((dx * dx + dy * dy) + dz * dz);   // the expression of the radius query (pcr_knn.hip)
                    if (!(d > nms_radius)) { alive = false; break; }
                }
                if (!alive) continue;
                kept.push_back(c);
                keypoints_out[taken++] = ids[j];
                if (taken > max_keypoints) done = true;  // ISS.py:72-73: stops once MORE than iss_count were taken
            }
            if (done) break;
        }
        if (d_rows) pcr_dev_free(ctx, d_rows, sizeof(pcr_pt) * n);
        if (d_batch) pcr_dev_free(ctx, d_batch, sizeof(pcr_pt) * BATCH);
        if (d_ids) pcr_dev_free(ctx, d_ids, sizeof(int) * BATCH);
        pcr_dev_free(ctx, d_cand, sizeof(int) * n);
        *n_keypoints_out = taken;
    }
    return rc;
}
