// PCA of a cloud and per-point normals from the k nearest neighbours --
// Pca_and_Voxel_filter/pca_normal.py:10-36 (PCA) and :85-90 (normal = eigenvector of the smallest
// eigenvalue of the covariance of the point's k = 5 nearest neighbours, the point itself included).
//
//   pcr_pca      mean, then centred second moments (np.cov: divisor N-1), fixed-order reduction on the
//                device; the symmetric 3x3 eigen-problem is solved on the host (Jacobi), eigenvalues
//                descending like PCA(sort=True).
//   pcr_normals  one lane per point: the 3x3x3 level-0 cells around the point are scanned keeping the
//                k best (d2, index) in registers; the result is exact when the k-th distance is <= the
//                cell size (nothing outside the block can be closer); the few other points are redone
//                by the exact descent of pcr_knn.  Covariance of the k neighbours (divisor k-1) and its
//                Jacobi eigen-decomposition on the device.
// Eigenvectors are defined up to sign (LAPACK's sign in the reference is implementation-defined).
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>
#include "pcr_grid_dev.h"
#include "pcr_linalg.h"

constexpr int NK_MAX = 16;

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi): eigenvalues descending, V columns = eigenvectors
__host__ __device__ static inline void sym3_eig(const double S[6] /* xx xy xz yy yz zz */, double ev[3], double V[9]) {
    double A[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
    double Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * diag) break;
        for (int pq = 0; pq < 3; ++pq) {
            const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2, r = 3 - p - q;
            const double apq = A[p][q];
            if (apq == 0.0) continue;
            const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            const double arp = A[r][p], arq = A[r][q];
            A[p][p] -= t * apq;
            A[q][q] += t * apq;
            A[p][q] = A[q][p] = 0.0;
            A[r][p] = A[p][r] = c * arp - s * arq;
            A[r][q] = A[q][r] = s * arp + c * arq;
            for (int i = 0; i < 3; ++i) {
                const double qp = Q[i][p], qq = Q[i][q];
                Q[i][p] = c * qp - s * qq;
                Q[i][q] = s * qp + c * qq;
            }
        }
    }
    int o[3] = {0, 1, 2};
    double e[3] = {A[0][0], A[1][1], A[2][2]};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (e[o[j]] < e[o[j + 1]]) { const int tt = o[j]; o[j] = o[j + 1]; o[j + 1] = tt; }
    for (int c = 0; c < 3; ++c) {
        ev[c] = e[o[c]];
        for (int i = 0; i < 3; ++i) V[3 * i + c] = Q[i][o[c]];
    }
}

// ------------------------------------------------------------------- PCA
__global__ void __launch_bounds__(256) pca_sum_kernel(const pcr_pt* __restrict__ pts, long long n, double mx, double my, double mz, int second,
                                                       double* __restrict__ partials /* [grid][8] */) {
    __shared__ double s[4][8];
    double m[6] = {0, 0, 0, 0, 0, 0};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const pcr_pt p = pts[i];
        if (!second) {
            m[0] += p.x; m[1] += p.y; m[2] += p.z;
        } else {
            const double x = p.x - mx, y = p.y - my, z = p.z - mz;
            m[0] += x * x; m[1] += x * y; m[2] += x * z; m[3] += y * y; m[4] += y * z; m[5] += z * z;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double v = m[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        m[k] = v;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < 6; ++k) s[wave][k] = m[k];
    __syncthreads();
    if (threadIdx.x < 6) partials[blockIdx.x * 8 + threadIdx.x] = (s[0][threadIdx.x] + s[1][threadIdx.x]) + (s[2][threadIdx.x] + s[3][threadIdx.x]);
}

// --------------------------------------------------------------- normals
template <int K>
__global__ void __launch_bounds__(256)
normals_kernel(pcr_grid_view gv, long long n, double* __restrict__ normals /* by row id (n,3) */, double* __restrict__ evals /* (n,3) or null */,
               int* __restrict__ nbr_out /* (n,K) by row id, or null */, unsigned int* __restrict__ redo_list, unsigned int* __restrict__ redo_count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    double bd[K];
    long long bi[K];
    unsigned int bp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { bd[k] = DBL_MAX; bi[k] = 0x7fffffffffffffffll; bp[k] = POS_NONE; }
    bool clamped = false;
    const int cx = cell_coord(p.x, gv.lo[0], gv.inv_cell0, &clamped);
    const int cy = cell_coord(p.y, gv.lo[1], gv.inv_cell0, &clamped);
    const int cz = cell_coord(p.z, gv.lo[2], gv.inv_cell0, &clamped);
    for (int c = 0; c < 27; ++c) {
        const unsigned int nx = (unsigned int)(cx + c % 3 - 1), ny = (unsigned int)(cy + (c / 3) % 3 - 1), nz = (unsigned int)(cz + c / 9 - 1);
        if (nx > (unsigned int)PCR_COORD_MAX || ny > (unsigned int)PCR_COORD_MAX || nz > (unsigned int)PCR_COORD_MAX) continue;
        unsigned int s, e;
        if (!lookup_cell(gv.table[0], gv.mask[0], nx, ny, nz, &s, &e)) continue;
        for (unsigned int j0 = s; j0 < e; j0 += 4) {   // four records per trip, requested together
            pcr_pt rec[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {   // (assigned on every path: conditionally unassigned records become loop-carried registers)
                rec[u] = pcr_pt{0.0, 0.0, 0.0, 0};
                if (j0 + u < e) rec[u] = gv.pts[j0 + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u >= e) break;
                double d2 = dist2(p.x, p.y, p.z, rec[u]);
                long long id = rec[u].id;
                unsigned int pos = j0 + u;
                if (!better(d2, id, bd[K - 1], bi[K - 1])) continue;
                // insertion into the sorted top-K (registers, fully unrolled bubble)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (better(d2, id, bd[k], bi[k])) {
                        const double td = bd[k]; const long long ti = bi[k]; const unsigned int tp = bp[k];
                        bd[k] = d2; bi[k] = id; bp[k] = pos;
                        d2 = td; id = ti; pos = tp;
                    }
                }
            }
        }
    }
    const double safe = gv.cell0 * (1.0 - 1e-9);
    const bool exact = !clamped && bp[K - 1] != POS_NONE && bd[K - 1] <= safe * safe;
    if (!exact && n > K) {  // (with n <= K every point is a neighbour: nothing can be missing)
        redo_list[atomicAdd(redo_count, 1u)] = (unsigned int)p.id;
        return;
    }
    // covariance of the neighbours (np.cov: mean removed, divisor count-1) and its smallest eigenvector
    int cnt = 0;
    double mx = 0, my = 0, mz = 0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (bp[k] != POS_NONE) { const pcr_pt b = gv.pts[bp[k]]; mx += b.x; my += b.y; mz += b.z; ++cnt; }
    mx /= cnt; my /= cnt; mz /= cnt;
    double S[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < K; ++k)
        if (bp[k] != POS_NONE) {
            const pcr_pt b = gv.pts[bp[k]];
            const double x = b.x - mx, y = b.y - my, z = b.z - mz;
            S[0] += x * x; S[1] += x * y; S[2] += x * z; S[3] += y * y; S[4] += y * z; S[5] += z * z;
        }
    const double inv = cnt > 1 ? 1.0 / (cnt - 1) : 0.0;
#pragma unroll
    for (int k = 0; k < 6; ++k) S[k] *= inv;
    double ev[3], V[9];
    sym3_eig(S, ev, V);
    normals[3 * p.id + 0] = V[2]; normals[3 * p.id + 1] = V[5]; normals[3 * p.id + 2] = V[8];
    if (evals) { evals[3 * p.id + 0] = ev[0]; evals[3 * p.id + 1] = ev[1]; evals[3 * p.id + 2] = ev[2]; }
    if (nbr_out) {
#pragma unroll
        for (int k = 0; k < K; ++k) nbr_out[(long long)p.id * K + k] = bp[k] != POS_NONE ? (int)bi[k] : -1;
    }
}

extern "C" {

int pcr_pca(pcr_ctx* ctx, const pcr_cloud* cloud, double eigvals_out[3], double eigvecs_out[9], double mean_out[3]) {
    if (!ctx || !cloud || !eigvals_out || !eigvecs_out) return PCR_E_INVALID;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    const long long n = cloud->n;
    int grid = (int)((n + 1023) / 1024);
    if (grid > 256) grid = 256;
    if (grid < 1) grid = 1;
    double* d_part = nullptr;
    int rc = pcr_dev_alloc(ctx, sizeof(double) * 8 * grid, (void**)&d_part);
    if (rc) return rc;
    std::vector<double> h(8 * grid);
    double mean[3] = {0, 0, 0}, S[6] = {0, 0, 0, 0, 0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        hipLaunchKernelGGL(pca_sum_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const pcr_pt*)cloud->d, n, mean[0], mean[1], mean[2], pass, d_part);
        PCR_HIP(ctx, hipGetLastError());
        PCR_HIP(ctx, hipMemcpyAsync(h.data(), d_part, sizeof(double) * 8 * grid, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, pcr_sync(ctx->stream));
        if (pass == 0) {
            for (int b = 0; b < grid; ++b)
                for (int k = 0; k < 3; ++k) mean[k] += h[8 * b + k];
            for (int k = 0; k < 3; ++k) mean[k] /= (double)n;
        } else {
            for (int b = 0; b < grid; ++b)
                for (int k = 0; k < 6; ++k) S[k] += h[8 * b + k];
        }
    }
    pcr_dev_free(ctx, d_part, sizeof(double) * 8 * grid);
    const double inv = n > 1 ? 1.0 / (double)(n - 1) : NAN;  // np.cov of a single observation is nan too
    for (int k = 0; k < 6; ++k) S[k] *= inv;
    sym3_eig(S, eigvals_out, eigvecs_out);
    if (mean_out)
        for (int k = 0; k < 3; ++k) mean_out[k] = mean[k];
    return PCR_OK;
}

int pcr_normals(pcr_ctx* ctx, const pcr_cloud* cloud, int k, double* normals_out, double* eigvals_out, int32_t* neighbours_out) {
    if (!ctx || !cloud || !normals_out || k < 2) return PCR_E_INVALID;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    if (k > NK_MAX) return PCR_E_UNSUPPORTED;
    hipSetDevice(ctx->device);
    const int64_t n = cloud->n;
    // cell such that a sphere of one cell radius holds ~2k points of a surface-like cloud (n points over
    // the two largest extents of the bounding box): then the 3x3x3 block answers almost every point.
    double lo[3], hi[3];
    int rc = pcr_cloud_bbox(ctx, cloud, lo, hi);
    if (rc) return rc;
    double e[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    if (e[0] < e[1]) std::swap(e[0], e[1]);
    if (e[1] < e[2]) std::swap(e[1], e[2]);
    if (e[0] < e[1]) std::swap(e[0], e[1]);
    double factor = 1.8;
    if (const char* f = getenv("PCR_NORMALS_CELL_FACTOR")) factor = atof(f);
    const double cell = (e[0] * e[1] > 0) ? factor * sqrt((double)k / 5.0 * e[0] * e[1] / (double)n) : 0.0;
    pcr_index* idx = nullptr;
    rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, cell, &idx);
    if (rc) return rc;
    if (getenv("PCR_NORMALS_DEBUG")) fprintf(stderr, "pcr_normals: n=%lld k=%d cell=%g\n", (long long)n, k, cell);
    double *d_nrm = nullptr, *d_ev = nullptr;
    int* d_nbr = nullptr;
    unsigned int* d_redo = nullptr;
    if ((rc = pcr_dev_alloc(ctx, sizeof(double) * 3 * n, (void**)&d_nrm))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(double) * 3 * n, (void**)&d_ev))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(int) * (size_t)k * n, (void**)&d_nbr))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_redo))) return rc;
    unsigned int* d_count = ctx->d_counters + 112;
    PCR_HIP(ctx, hipMemsetAsync(d_count, 0, sizeof(unsigned int), ctx->stream));
    const unsigned grid = (unsigned)((n + 255) / 256);
#define PCR_NK(K) case K: hipLaunchKernelGGL(normals_kernel<K>, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (long long)n, d_nrm, d_ev, d_nbr, d_redo, d_count); break;
    switch (k) {
        PCR_NK(2) PCR_NK(3) PCR_NK(4) PCR_NK(5) PCR_NK(6) PCR_NK(7) PCR_NK(8) PCR_NK(9) PCR_NK(10) PCR_NK(11) PCR_NK(12) PCR_NK(13) PCR_NK(14) PCR_NK(15) PCR_NK(16)
    }
#undef PCR_NK
    PCR_HIP(ctx, hipGetLastError());
    // eigenvalues and neighbour lists cross PCIe only when the caller asked for them, and then straight into the caller's arrays
    // (they went through two host vectors before: a page-faulting allocation and a second copy of 5 MB per 120 000 points)
    unsigned int n_redo = 0;
    PCR_HIP(ctx, hipMemcpyAsync(normals_out, d_nrm, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    if (eigvals_out) PCR_HIP(ctx, hipMemcpyAsync(eigvals_out, d_ev, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    if (neighbours_out) PCR_HIP(ctx, hipMemcpyAsync(neighbours_out, d_nbr, sizeof(int) * (size_t)k * n, hipMemcpyDeviceToHost, ctx->stream));
    { const int rc_n = pcr_d2h_small(ctx, &n_redo, d_count, sizeof(unsigned int)); if (rc_n) return rc_n; }   // (synchronises)
    rc = PCR_OK;
    if (getenv("PCR_NORMALS_DEBUG")) fprintf(stderr, "pcr_normals: redo=%u\n", n_redo);
    if (n_redo > 0) {
        // points whose k-th neighbour lies beyond the 3x3x3 block: exact k-NN by the pruned descent, PCA on the host
        std::vector<unsigned int> redo(n_redo);
        PCR_HIP(ctx, hipMemcpy(redo.data(), d_redo, sizeof(unsigned int) * n_redo, hipMemcpyDeviceToHost));
        std::vector<double> xyz(3 * (size_t)n);
        rc = pcr_cloud_download_f64(ctx, cloud, xyz.data());
        std::vector<double> q(3 * (size_t)n_redo);
        std::vector<long long> rows(n_redo);
        for (unsigned int r = 0; r < n_redo; ++r) {  // redo entries are caller row ids
            rows[r] = redo[r];
            for (int c = 0; c < 3; ++c) q[3 * (size_t)r + c] = xyz[3 * (size_t)redo[r] + c];
        }
        std::vector<int32_t> kidx((size_t)k * n_redo);
        std::vector<double> kdist((size_t)k * n_redo);
        if (rc == PCR_OK) rc = pcr_knn(ctx, idx, q.data(), n_redo, k, kidx.data(), kdist.data());
        for (unsigned int r = 0; r < n_redo && rc == PCR_OK; ++r) {
            int cnt = 0;
            double m[3] = {0, 0, 0};
            for (int j = 0; j < k; ++j) {
                if (kdist[(size_t)r * k + j] >= 1e10) continue;
                const int id = kidx[(size_t)r * k + j];
                for (int c = 0; c < 3; ++c) m[c] += xyz[3 * (size_t)id + c];
                ++cnt;
            }
            for (int c = 0; c < 3; ++c) m[c] /= cnt;
            double S[6] = {0, 0, 0, 0, 0, 0};
            for (int j = 0; j < k; ++j) {
                if (kdist[(size_t)r * k + j] >= 1e10) { if (neighbours_out) neighbours_out[(size_t)rows[r] * k + j] = -1; continue; }
                const int id = kidx[(size_t)r * k + j];
                if (neighbours_out) neighbours_out[(size_t)rows[r] * k + j] = id;
                const double x = xyz[3 * (size_t)id] - m[0], y = xyz[3 * (size_t)id + 1] - m[1], z = xyz[3 * (size_t)id + 2] - m[2];
                S[0] += x * x; S[1] += x * y; S[2] += x * z; S[3] += y * y; S[4] += y * z; S[5] += z * z;
            }
            const double inv = cnt > 1 ? 1.0 / (cnt - 1) : 0.0;
            for (int c = 0; c < 6; ++c) S[c] *= inv;
            double e3[3], V[9];
            sym3_eig(S, e3, V);
            for (int c = 0; c < 3; ++c) {
                normals_out[3 * (size_t)rows[r] + c] = V[3 * c + 2];
                if (eigvals_out) eigvals_out[3 * (size_t)rows[r] + c] = e3[c];
            }
        }
    }
    pcr_dev_free(ctx, d_nrm, sizeof(double) * 3 * n);
    pcr_dev_free(ctx, d_ev, sizeof(double) * 3 * n);
    pcr_dev_free(ctx, d_nbr, sizeof(int) * (size_t)k * n);
    pcr_dev_free(ctx, d_redo, sizeof(unsigned int) * n);
    pcr_index_free(ctx, idx);
    return rc;
}

}  // extern "C"
