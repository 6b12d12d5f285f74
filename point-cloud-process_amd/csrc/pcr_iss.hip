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
// The NMS is a short sequential host loop over radius queries (<= max_keypoints + 1 of them).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <numeric>
#include <vector>
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
                    for (unsigned int j = s; j < e; ++j) f(gv.pts[j]);
            }
        }
    }
}

__global__ void __launch_bounds__(256) iss_count_kernel(pcr_grid_view gv, long long n, double radius, int* __restrict__ counts /* by row id */) {
    const int gl = threadIdx.x % IG;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / IG;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    int c = 0;
    auto f = [&](const pcr_pt& b) {
        const double d = sqrt(dist2(p.x, p.y, p.z, b));
        c += !(d > radius);
    };
    iss_visit_block(gv, p.x, p.y, p.z, gl, f);
#pragma unroll
    for (int off = IG / 2; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (gl == 0) counts[p.id] = c;
}

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

__global__ void __launch_bounds__(256) iss_cov_kernel(pcr_grid_view gv, long long n, double radius, const int* __restrict__ counts,
                                                      double* __restrict__ lambdas /* by row id, (n,3) */) {
    const int gl = threadIdx.x % IG;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) / IG;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    double m[7] = {0, 0, 0, 0, 0, 0, 0};  // xx xy xz yy yz zz, denom
    auto f = [&](const pcr_pt& b) {
        const double dx = b.x - p.x, dy = b.y - p.y, dz = b.z - p.z;
        const double d = sqrt((dx * dx + dy * dy) + dz * dz);
        if (!(d > radius)) {
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
    }
}

extern "C" int pcr_iss(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, double gamma21, double gamma32, double nms_radius,
                       int max_keypoints, double* lambdas_out, int32_t* counts_out, int32_t* keypoints_out, int* n_keypoints_out) {
    if (!ctx || !cloud || !lambdas_out || !(radius > 0)) return PCR_E_INVALID;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    const int64_t n = cloud->n;
    pcr_index* idx = nullptr;
    int rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, radius * (1.0 + 1e-9), &idx);  // block of 27 cells covers the ball with rounding slack
    if (rc) return rc;
    // the index may have coarsened the cell (huge extents): the 3x3x3 block then still covers the radius
    if (idx->cell < radius * (1.0 - 1e-12)) { pcr_index_free(ctx, idx); return PCR_E_UNSUPPORTED; }
    int* d_counts = nullptr;
    double* d_lam = nullptr;
    if ((rc = pcr_dev_alloc(ctx, sizeof(int) * n, (void**)&d_counts))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(double) * 3 * n, (void**)&d_lam))) return rc;
    const unsigned grid = (unsigned)((n * IG + 255) / 256);
    hipLaunchKernelGGL(iss_count_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (long long)n, radius, d_counts);
    hipLaunchKernelGGL(iss_cov_kernel, dim3(grid), dim3(256), 0, ctx->stream, idx->view, (long long)n, radius, (const int*)d_counts, d_lam);
    PCR_HIP(ctx, hipGetLastError());
    PCR_HIP(ctx, hipMemcpyAsync(lambdas_out, d_lam, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<int32_t> counts_host;
    int32_t* cdst = counts_out;
    if (!cdst) { counts_host.resize(n); cdst = counts_host.data(); }
    PCR_HIP(ctx, hipMemcpyAsync(cdst, d_counts, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pcr_dev_free(ctx, d_counts, sizeof(int) * n);
    pcr_dev_free(ctx, d_lam, sizeof(double) * 3 * n);
    rc = PCR_OK;
    if (keypoints_out && n_keypoints_out) {
        // candidates (ISS.py:55-57), sorted by lambda3 descending, stable (ISS.py:59)
        std::vector<int> cand;
        for (int64_t i = 0; i < n; ++i) {
            const double l1 = lambdas_out[3 * i], l2 = lambdas_out[3 * i + 1], l3 = lambdas_out[3 * i + 2];
            if (l2 / l1 < gamma21 && l3 / l2 < gamma32) cand.push_back((int)i);
        }
        std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return lambdas_out[3 * a + 2] > lambdas_out[3 * b + 2]; });
        std::vector<char> alive(n, 0);
        for (int c : cand) alive[c] = 1;
        // point coordinates by row id for the NMS queries
        std::vector<double> xyz(3 * n);
        rc = pcr_cloud_download_f64(ctx, cloud, xyz.data());
        int taken = 0;
        for (size_t ci = 0; ci < cand.size() && rc == PCR_OK; ++ci) {
            const int id = cand[ci];
            if (!alive[id]) continue;
            keypoints_out[taken++] = id;
            int64_t cnt = 0, offs[2] = {0, 0};
            rc = pcr_radius(ctx, idx, &xyz[3 * (size_t)id], 1, nms_radius, &cnt, nullptr, nullptr, nullptr);
            if (rc) break;
            offs[1] = cnt;
            std::vector<int32_t> nb((size_t)cnt + 1);
            std::vector<double> nd((size_t)cnt + 1);
            rc = pcr_radius(ctx, idx, &xyz[3 * (size_t)id], 1, nms_radius, nullptr, offs, nb.data(), nd.data());
            if (rc) break;
            for (int64_t j = 0; j < cnt; ++j) alive[nb[(size_t)j]] = 0;
            if (taken > max_keypoints) break;  // ISS.py:72-73: stops once MORE than iss_count were taken
        }
        *n_keypoints_out = taken;
    }
    pcr_index_free(ctx, idx);
    return rc;
}
