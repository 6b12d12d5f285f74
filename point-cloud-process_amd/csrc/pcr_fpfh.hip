// Global initialisation in front of ICP -- the Open3D calls of Registration/main.py:33-84 and the template
// surface icp_template.py:20-41,56-110 (find_matchings, ransac_init):
//   pcr_normals_hybrid   estimate_normals(KDTreeSearchParamHybrid(radius, max_nn))      main.py:39-40
//   pcr_fpfh             compute_fpfh_feature(pcd, KDTreeSearchParamHybrid(radius, max_nn)) main.py:44-46
//   pcr_feature_match    nearest neighbour in feature space (find_matchings / the matching inside
//                        registration_ransac_based_on_feature_matching)                  main.py:73, icp_template.py:20-41
//   pcr_ransac           3-point RANSAC with edge-length and distance checkers            main.py:73-83, icp_template.py:88-110
// Open3D is a third-party dependency that is absent here and unpinned in the reference: the algorithms below
// follow its published behaviour (FPFH of Rusu et al. 2009 as implemented by Open3D >= 0.12: 3 x 11 bins,
// increments 100/(k-1), neighbour SPFH weighted by 1/d^2 and renormalised to 100 per sub-histogram).
// "Parity unpinned": no output of the reference exists for this stage (its RANSAC is randomised).
//
// Hybrid neighbourhood = the up-to-max_nn nearest points with d^2 < radius^2, ordered by (d^2, row).  One
// 64-lane wave per point: the 3x3x3 block of a grid with cell = radius is scanned, candidates inside the
// sphere are compacted into LDS (ballot + prefix), bitonic-sorted, truncated.  More than NB_CAP candidates
// inside the sphere: the radius is first bisected down to a value that keeps between max_nn and NB_CAP.
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "pcr_grid_dev.h"
#include "pcr_linalg.h"

constexpr int NB_CAP = 1024;

struct __attribute__((aligned(16))) nb_entry {
    double d2;
    unsigned int pos;  // position in the index's sorted order
    unsigned int id;   // caller row
};

__device__ static inline bool nb_less(const nb_entry& a, const nb_entry& b) { return a.d2 < b.d2 || (a.d2 == b.d2 && a.id < b.id); }

// blockDim.x == 64.  Returns the neighbour count (<= max_nn), entries sorted in nb[0..count); -1 = cannot bound the set.
// The 27 cells are looked up by 27 lanes AT ONCE (one lane walking them one after the other paid 27 dependent round trips per
// point), the non-empty ones become a flat list of ranges (cell_s / cell_o: start and exclusive point offset) and the 64 lanes
// stride over the concatenation, so a scan is ceil(points / 64) round trips whatever the cells' sizes.
struct hybrid_lds {
    nb_entry nb[NB_CAP];
    unsigned int cell_s[28], cell_o[28];
};
__device__ static int gather_hybrid(const pcr_grid_view& gv, double qx, double qy, double qz, double r2, int max_nn, hybrid_lds* L) {
    nb_entry* const nb = L->nb;
    const int lane = threadIdx.x;
    bool clamped = false;
    const int cx = cell_coord(qx, gv.lo[0], gv.inv_cell0, &clamped);
    const int cy = cell_coord(qy, gv.lo[1], gv.inv_cell0, &clamped);
    const int cz = cell_coord(qz, gv.lo[2], gv.inv_cell0, &clamped);
    unsigned int s = 0, e = 0;
    bool has = false;
    if (gv.levels == 0) {
        // no grid (a cloud of a few thousand points, see brute_view): the whole cloud is the one "cell".  The sphere test, the order
        // (d^2, row) and the cut at max_nn make the list -- the same list whatever superset of the sphere was scanned.
        has = lane == 0;
        e = (unsigned int)gv.n;
    } else if (lane < 27) {
        const unsigned int nx = (unsigned int)(cx + lane % 3 - 1), ny = (unsigned int)(cy + (lane / 3) % 3 - 1), nz = (unsigned int)(cz + lane / 9 - 1);
        if (nx <= (unsigned int)PCR_COORD_MAX && ny <= (unsigned int)PCR_COORD_MAX && nz <= (unsigned int)PCR_COORD_MAX)
            has = lookup_cell(gv.table[0], gv.mask[0], nx, ny, nz, &s, &e);
    }
    const unsigned long long m_has = __ballot(has);
    const int n_cells = __popcll(m_has);
    unsigned int inc = has ? e - s : 0u;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {   // lanes 0..26 hold the counts
        const unsigned int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    const unsigned int total = __shfl(inc, 31, 64);
    if (has) {
        const int r = __popcll(m_has & ((1ull << lane) - 1ull));
        L->cell_s[r] = s;
        L->cell_o[r] = inc - (e - s);
    }
    __syncthreads();
    auto scan = [&](double T, bool store) -> int {
        int found = 0;
        for (unsigned int t0 = 0; t0 < total; t0 += 64) {
            const unsigned int t = t0 + lane;
            bool keep = false;
            nb_entry en;
            en.d2 = 0.0; en.pos = 0; en.id = 0;
            if (t < total) {
                int lo = 0, hi = n_cells - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (L->cell_o[mid] <= t) lo = mid;
                    else hi = mid - 1;
                }
                const unsigned int j = L->cell_s[lo] + (t - L->cell_o[lo]);
                const pcr_pt b = gv.pts[j];
                en.d2 = dist2(qx, qy, qz, b);
                en.pos = j;
                en.id = (unsigned int)b.id;
                keep = en.d2 < T;
            }
            const unsigned long long m = __ballot(keep);
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            if (store && keep && found + rank < NB_CAP) nb[found + rank] = en;
            found += __popcll(m);
        }
        return found;
    };
    int cnt = scan(r2, true);
    if (cnt > NB_CAP) {
        unsigned long long lo = 0, hi = (unsigned long long)__double_as_longlong(r2);
        bool found = false;
        double T = r2;
        for (int it = 0; it < 70 && hi - lo > 1; ++it) {
            const unsigned long long mid = lo + (hi - lo) / 2;
            T = __longlong_as_double((long long)mid);
            const int c = scan(T, false);
            if (c > NB_CAP) hi = mid;
            else if (c < max_nn) lo = mid;
            else { found = true; break; }
        }
        if (!found) return -1;
        __syncthreads();
        cnt = scan(T, true);
    }
    int P = 64;
    while (P < cnt) P <<= 1;
    for (int i = cnt + lane; i < P; i += 64) { nb[i].d2 = DBL_MAX; nb[i].pos = POS_NONE; nb[i].id = 0xffffffffu; }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < P; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const nb_entry a = nb[i], b = nb[l];
                    const bool up = (i & k) == 0;
                    if (up ? nb_less(b, a) : nb_less(a, b)) { nb[i] = b; nb[l] = a; }
                }
            }
            __syncthreads();
        }
    return cnt < max_nn ? cnt : max_nn;
}

__device__ static inline double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// symmetric 3x3 Jacobi: eigenvector of the smallest eigenvalue
__device__ static void smallest_eigvec(const double S[6], double n[3]) {
    double A[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
    double Q[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * diag) break;
#pragma unroll
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
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double qp = Q[i][p], qq = Q[i][q];
                Q[i][p] = c * qp - s * qq;
                Q[i][q] = s * qp + c * qq;
            }
        }
    }
    // (selects on values read into scalars first, not Q[i][m] with a run-time m: a dynamically indexed local array lives in scratch
    // memory -- 112 bytes per lane for this kernel)
    const double a0 = A[0][0], a1 = A[1][1], a2 = A[2][2];
    const bool m1 = a1 < a0;
    const double am = m1 ? a1 : a0;
    const bool m2 = a2 < am;
    const double q00 = Q[0][0], q01 = Q[0][1], q02 = Q[0][2], q10 = Q[1][0], q11 = Q[1][1], q12 = Q[1][2], q20 = Q[2][0], q21 = Q[2][1], q22 = Q[2][2];
    n[0] = m2 ? q02 : (m1 ? q01 : q00);
    n[1] = m2 ? q12 : (m1 ? q11 : q10);
    n[2] = m2 ? q22 : (m1 ? q21 : q20);
}

// ------------------------------------------------------------ hybrid normals
__global__ void __launch_bounds__(64) hybrid_normals_kernel(pcr_grid_view gv, long long n, double r2, int max_nn, int orient, double vx, double vy,
                                                            double vz, double* __restrict__ normals /* (n,3) by row */, int* __restrict__ fail) {
    __shared__ hybrid_lds s_L;
    nb_entry* const nb = s_L.nb;
    const long long i = blockIdx.x;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    const int cnt = gather_hybrid(gv, p.x, p.y, p.z, r2, max_nn, &s_L);
    if (cnt < 0) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
    double nrm[3] = {0.0, 0.0, 1.0};  // Open3D's value for neighbourhoods of fewer than 3 points
    if (cnt >= 3) {
        // cumulants about the query point (Open3D accumulates raw coordinates; centring first is the same
        // covariance with less cancellation)
        double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = threadIdx.x; k < cnt; k += 64) {
            const pcr_pt b = gv.pts[nb[k].pos];
            const double x = b.x - p.x, y = b.y - p.y, z = b.z - p.z;
            c[0] += x; c[1] += y; c[2] += z;
            c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) c[k] = wave_sum(c[k]) / (double)cnt;
        const double S[6] = {c[3] - c[0] * c[0], c[4] - c[0] * c[1], c[5] - c[0] * c[2], c[6] - c[1] * c[1], c[7] - c[1] * c[2], c[8] - c[2] * c[2]};
        smallest_eigvec(S, nrm);
        const double len = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
        if (len == 0.0 || !(len == len)) { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; }
        else if (orient) {
            const double d = nrm[0] * (vx - p.x) + nrm[1] * (vy - p.y) + nrm[2] * (vz - p.z);
            if (d < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
        }
    }
    const double n0 = nrm[0], n1 = nrm[1], n2 = nrm[2];
    if (threadIdx.x < 3) normals[3 * p.id + threadIdx.x] = threadIdx.x == 0 ? n0 : (threadIdx.x == 1 ? n1 : n2);
}

// ---------------------------------------------------------------------- SPFH
// Darboux-frame pair features (Open3D ComputePairFeatures): f0 = atan2 angle, f1 = v.n2, f2 = n1.d/|d|
__device__ static inline bool pair_features(const double p1[3], const double n1[3], const double p2[3], const double n2[3], double f[3]) {
    double d[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    const double len = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
    f[0] = f[1] = f[2] = 0.0;
    if (len == 0.0) return true;  // zero vector still lands in bins (5, 5, 5), like Open3D's Zero() return
    const double a1 = ((n1[0] * d[0] + n1[1] * d[1]) + n1[2] * d[2]) / len;
    const double a2 = ((n2[0] * d[0] + n2[1] * d[1]) + n2[2] * d[2]) / len;
    double u[3], w2[3];
    if (fabs(a1) < fabs(a2)) {  // acos(|a1|) > acos(|a2|): the frame is anchored at the point whose normal is closer to the line
        u[0] = n2[0]; u[1] = n2[1]; u[2] = n2[2];
        w2[0] = n1[0]; w2[1] = n1[1]; w2[2] = n1[2];
        d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2];
        f[2] = -a2;
    } else {
        u[0] = n1[0]; u[1] = n1[1]; u[2] = n1[2];
        w2[0] = n2[0]; w2[1] = n2[1]; w2[2] = n2[2];
        f[2] = a1;
    }
    double v[3] = {d[1] * u[2] - d[2] * u[1], d[2] * u[0] - d[0] * u[2], d[0] * u[1] - d[1] * u[0]};
    const double vn = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    if (vn == 0.0) { f[2] = 0.0; return true; }
    v[0] /= vn; v[1] /= vn; v[2] /= vn;
    const double w[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
    f[1] = (v[0] * w2[0] + v[1] * w2[1]) + v[2] * w2[2];
    f[0] = atan2((w[0] * w2[0] + w[1] * w2[1]) + w[2] * w2[2], (u[0] * w2[0] + u[1] * w2[1]) + u[2] * w2[2]);
    return true;
}

__device__ static inline int clamp_bin(double x) {
    int h = (int)floor(x);
    return h < 0 ? 0 : (h > 10 ? 10 : h);
}

__global__ void __launch_bounds__(64)
spfh_kernel(pcr_grid_view gv, long long n, double r2, int max_nn, const double* __restrict__ normals /* by row */, double* __restrict__ spfh /* (n,33) by row */,
            unsigned int* __restrict__ nb_id /* (n,max_nn) by row */, double* __restrict__ nb_d2, int* __restrict__ nb_cnt, int* __restrict__ fail) {
    __shared__ hybrid_lds s_L;
    nb_entry* const nb = s_L.nb;
    __shared__ int hist[33];
    const long long i = blockIdx.x;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    const int cnt = gather_hybrid(gv, p.x, p.y, p.z, r2, max_nn, &s_L);  // ends with a barrier
    if (cnt < 0) { if (threadIdx.x == 0) atomicAdd(fail, 1); return; }
    const double p1[3] = {p.x, p.y, p.z};
    const double n1[3] = {normals[3 * p.id], normals[3 * p.id + 1], normals[3 * p.id + 2]};
    for (int k = threadIdx.x; k < cnt; k += 64) {
        const nb_entry en = nb[k];
        nb_id[(long long)p.id * max_nn + k] = en.id;
        nb_d2[(long long)p.id * max_nn + k] = en.d2;
        if (k == 0) continue;  // the query point itself (or a duplicate of it)
        const pcr_pt b = gv.pts[en.pos];
        const double p2[3] = {b.x, b.y, b.z};
        const double n2[3] = {normals[3 * (long long)en.id], normals[3 * (long long)en.id + 1], normals[3 * (long long)en.id + 2]};
        double f[3];
        pair_features(p1, n1, p2, n2, f);
        atomicAdd(&hist[clamp_bin(11.0 * (f[0] + M_PI) / (2.0 * M_PI))], 1);
        atomicAdd(&hist[11 + clamp_bin(11.0 * (f[1] + 1.0) * 0.5)], 1);
        atomicAdd(&hist[22 + clamp_bin(11.0 * (f[2] + 1.0) * 0.5)], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) nb_cnt[p.id] = cnt;
    if (threadIdx.x < 33) {
        double v = 0.0;
        if (cnt > 1) {
            const double incr = 100.0 / (double)(cnt - 1);
            for (int c = 0; c < hist[threadIdx.x]; ++c) v += incr;  // repeated addition, like the reference library
        }
        spfh[33 * (long long)p.id + threadIdx.x] = v;
    }
}

// ---------------------------------------------------------------------- FPFH
// One wave per point, lane = histogram bin (33 of 64).  The neighbour list (row, d^2) goes through LDS first, so the SPFH rows
// of four neighbours are requested together instead of one dependent chain id -> row per neighbour; every term is spfh / d^2
// (a true division, as in Open3D); the three renormalising sums are taken over the lanes of each 11-bin block at the end.
__global__ void __launch_bounds__(64)
fpfh_kernel(long long n, int max_nn, const double* __restrict__ spfh, const unsigned int* __restrict__ nb_id, const double* __restrict__ nb_d2,
            const int* __restrict__ nb_cnt, double* __restrict__ fpfh /* (n,33) by row */) {
    __shared__ unsigned int s_id[NB_CAP];
    __shared__ double s_d2[NB_CAP];
    __shared__ double s_acc[33];
    const long long i = blockIdx.x;
    if (i >= n) return;
    const int cnt = nb_cnt[i];
    const int lane = threadIdx.x;
    for (int k = lane; k < cnt; k += 64) { s_id[k] = nb_id[i * max_nn + k]; s_d2[k] = nb_d2[i * max_nn + k]; }
    __syncthreads();
    double acc = 0.0;
    if (cnt > 1 && lane < 33) {
        int k = 1;
        for (; k + 4 <= cnt; k += 4) {
            double v[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { d[u] = s_d2[k + u]; v[u] = spfh[33 * (long long)s_id[k + u] + lane]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (d[u] != 0.0) acc += v[u] / d[u];
        }
        for (; k < cnt; ++k) {
            const double d = s_d2[k];
            if (d != 0.0) acc += spfh[33 * (long long)s_id[k] + lane] / d;
        }
        s_acc[lane] = acc;
    }
    __syncthreads();
    if (lane < 33) {
        double v = 0.0;
        if (cnt > 1) {
            const int g = lane / 11;
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < 11; ++j) sum += s_acc[11 * g + j];
            v = acc * (sum != 0.0 ? 100.0 / sum : 0.0) + spfh[33 * i + lane];
        }
        fpfh[33 * i + lane] = v;
    }
}

// ----------------------------------------------------------- feature matching
// One thread per query row, target rows staged through LDS in tiles; squared L2 summed over the
// dimensions in order; ties to the lowest target row.
constexpr int FM_TILE = 32;
// grid = (query blocks, target splits): block (bx, by) scans targets [by * per, (by + 1) * per); a small merge kernel
// takes the minimum over the splits (ascending split order + strict comparison keeps the lowest row on ties).
template <int DIM>
__global__ void __launch_bounds__(256) feature_match_kernel(const double* __restrict__ A, long long na, const double* __restrict__ B, long long nb, int dim_rt,
                                                             long long per, int* __restrict__ idx_out, double* __restrict__ d2_out) {
    extern __shared__ double tile[];  // FM_TILE * dim
    const int dim = DIM > 0 ? DIM : dim_rt;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < na;
    double a[DIM > 0 ? DIM : 1];
    if (DIM > 0 && live) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) a[k] = A[i * DIM + k];
    }
    double best = DBL_MAX;
    int bidx = -1;
    const long long tb = (long long)blockIdx.y * per, te = (tb + per < nb) ? tb + per : nb;
    for (long long t0 = tb; t0 < te; t0 += FM_TILE) {
        const int rows = (int)((te - t0) < FM_TILE ? (te - t0) : FM_TILE);
        __syncthreads();
        for (int e = threadIdx.x; e < rows * dim; e += blockDim.x) tile[e] = B[t0 * dim + e];
        __syncthreads();
        if (!live) continue;
        for (int r = 0; r < rows; ++r) {
            double s = 0.0;
            if (DIM > 0) {
#pragma unroll
                for (int k = 0; k < DIM; ++k) { const double d = a[k] - tile[r * DIM + k]; s += d * d; }
            } else {
                for (int k = 0; k < dim; ++k) { const double d = A[i * dim + k] - tile[r * dim + k]; s += d * d; }
            }
            if (s < best) { best = s; bidx = (int)(t0 + r); }
        }
    }
    if (live) { idx_out[(long long)blockIdx.y * na + i] = bidx; d2_out[(long long)blockIdx.y * na + i] = best; }
}

__global__ void feature_match_merge_kernel(const int* __restrict__ cidx, const double* __restrict__ cd2, long long na, int splits,
                                           int* __restrict__ idx_out, double* __restrict__ d2_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na) return;
    double best = DBL_MAX;
    int bidx = -1;
    for (int sp = 0; sp < splits; ++sp) {
        const double d = cd2[(long long)sp * na + i];
        const int j = cidx[(long long)sp * na + i];
        if (j >= 0 && d < best) { best = d; bidx = j; }
    }
    idx_out[i] = bidx;
    d2_out[i] = best;
}

// --------------------------------------------------------------------- RANSAC
// The loop of registration_ransac_based_on_feature_matching (main.py:73-83) / ransac_init (icp_template.py:88-110) stays on the
// device: a batch of hypotheses is evaluated side by side (one wave each), then ONE wave walks the batch in iteration order --
// running best, confidence-based exit -- exactly as the sequential loop would, and leaves the loop state in device memory; the
// batches behind a stop return at once.  The host enqueues every batch and synchronises once.
struct ransac_state {
    double best_fit, best_rmse;
    double bestT[12];
    long long best_itr, exit_itr, done, n_valid;
    int stop, m;
    int pad[2];
};
struct ransac_args {
    const pcr_pt* src;  // by row (id == position)
    const pcr_pt* tgt;
    const int* corr;    // (m,2)
    int first_iter, n_iter;
    unsigned long long seed;
    double edge_sim;    // <= 0: checker off
    double max_dist;    // inlier threshold and distance checker
    int check_distance;
    int max_iteration;
    double confidence;
};

__host__ __device__ static inline unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// one wave per hypothesis; out: inl[h] (-1 = rejected by a checker), err2[h], T[h][12]
__global__ void __launch_bounds__(64) ransac_kernel(ransac_args a, const ransac_state* __restrict__ st, int* __restrict__ inl, double* __restrict__ err2,
                                                    double* __restrict__ Tout) {
    const int h = blockIdx.x;
    if (h >= a.n_iter || st->stop) return;
    const int m = st->m;
    const unsigned long long itr = (unsigned long long)(a.first_iter + h);
    if ((long long)itr >= st->exit_itr) return;   // (the walk never looks at it)
    double s[3][3], t[3][3];
    for (int j = 0; j < 3; ++j) {
        const unsigned int c = (unsigned int)(mix64(a.seed ^ mix64(itr * 3 + j)) % (unsigned long long)m);
        const pcr_pt ps = a.src[a.corr[2 * c]], pt = a.tgt[a.corr[2 * c + 1]];
        s[j][0] = ps.x; s[j][1] = ps.y; s[j][2] = ps.z;
        t[j][0] = pt.x; t[j][1] = pt.y; t[j][2] = pt.z;
    }
    bool ok = true;
    if (a.edge_sim > 0) {
        for (int i = 0; i < 3 && ok; ++i)
            for (int j = i + 1; j < 3; ++j) {
                const double ds = sqrt(((s[i][0] - s[j][0]) * (s[i][0] - s[j][0]) + (s[i][1] - s[j][1]) * (s[i][1] - s[j][1])) + (s[i][2] - s[j][2]) * (s[i][2] - s[j][2]));
                const double dt = sqrt(((t[i][0] - t[j][0]) * (t[i][0] - t[j][0]) + (t[i][1] - t[j][1]) * (t[i][1] - t[j][1])) + (t[i][2] - t[j][2]) * (t[i][2] - t[j][2]));
                if (ds < dt * a.edge_sim || dt < ds * a.edge_sim) { ok = false; break; }
            }
    }
    double R[9], tr[3];
    if (ok) {
        // Kabsch on the three pairs (procrustes_transformation, icp_template.py:43-54; proper rotation for the rank-2 case)
        double mo[18];
        for (int k = 0; k < 18; ++k) mo[k] = 0.0;
        const double org[3] = {s[0][0], s[0][1], s[0][2]};
        mo[0] = 3.0;
        for (int j = 0; j < 3; ++j) {
            const double ax = s[j][0] - org[0], ay = s[j][1] - org[1], az = s[j][2] - org[2];
            const double bx = t[j][0] - org[0], by = t[j][1] - org[1], bz = t[j][2] - org[2];
            mo[1] += ax; mo[2] += ay; mo[3] += az;
            mo[4] += bx; mo[5] += by; mo[6] += bz;
            mo[7] += bx * ax; mo[8] += bx * ay; mo[9] += bx * az;
            mo[10] += by * ax; mo[11] += by * ay; mo[12] += by * az;
            mo[13] += bz * ax; mo[14] += bz * ay; mo[15] += bz * az;
            mo[16] += (ax * ax + ay * ay) + az * az;
            mo[17] += (bx * bx + by * by) + bz * bz;
        }
        pcr::kabsch_from_moments(mo, org, R, tr, nullptr);
        for (int k = 0; k < 9; ++k) ok = ok && (R[k] == R[k]);
        if (ok && a.check_distance) {
            for (int j = 0; j < 3; ++j) {
                const double x = ((R[0] * s[j][0] + R[1] * s[j][1]) + R[2] * s[j][2]) + tr[0] - t[j][0];
                const double y = ((R[3] * s[j][0] + R[4] * s[j][1]) + R[5] * s[j][2]) + tr[1] - t[j][1];
                const double z = ((R[6] * s[j][0] + R[7] * s[j][1]) + R[8] * s[j][2]) + tr[2] - t[j][2];
                if (sqrt((x * x + y * y) + z * z) > a.max_dist) ok = false;
            }
        }
    }
    if (!ok) {
        if (threadIdx.x == 0) { inl[h] = -1; err2[h] = 0.0; }
        return;
    }
    int good = 0;
    double e2 = 0.0;
    for (int c = threadIdx.x; c < m; c += 64) {
        const pcr_pt ps = a.src[a.corr[2 * c]], pt = a.tgt[a.corr[2 * c + 1]];
        const double x = ((R[0] * ps.x + R[1] * ps.y) + R[2] * ps.z) + tr[0] - pt.x;
        const double y = ((R[3] * ps.x + R[4] * ps.y) + R[5] * ps.z) + tr[1] - pt.y;
        const double z = ((R[6] * ps.x + R[7] * ps.y) + R[8] * ps.z) + tr[2] - pt.z;
        const double dis = sqrt((x * x + y * y) + z * z);
        if (dis < a.max_dist) { ++good; e2 += dis * dis; }
    }
    e2 = wave_sum(e2);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) good += __shfl_xor(good, off, 64);
    if (threadIdx.x == 0) {
        inl[h] = good;
        err2[h] = e2;
        for (int k = 0; k < 9; ++k) Tout[12 * (long long)h + k] = R[k];
        for (int k = 0; k < 3; ++k) Tout[12 * (long long)h + 9 + k] = tr[k];
    }
}

// ONE wave: the sequential loop over the batch -- iteration order, running best (IsBetterRANSACThan: higher fitness, or equal
// fitness and lower rmse; the initial best is (0, 0)), exit_itr = min(exit_itr, ceil(log(1 - confidence) / log(1 - fitness^3)))
// after every improvement, stop at the first iteration >= exit_itr -- 64 iterations at a time: inside a chunk the NEXT improvement
// is the first lane that beats the current best (the order is a strict weak order, so nobody in front of it can beat the new best).
__global__ void __launch_bounds__(64) ransac_walk_kernel(ransac_args a, ransac_state* __restrict__ st, const int* __restrict__ inl, const double* __restrict__ err2,
                                                         const double* __restrict__ Tout) {
    if (st->stop) return;
    const int lane = threadIdx.x;
    const int m = st->m;
    double best_fit = st->best_fit, best_rmse = st->best_rmse;
    long long best_itr = st->best_itr, exit_itr = st->exit_itr, n_valid = st->n_valid;
    int best_h = -1;
    bool stop = false;
    for (int base = 0; base < a.n_iter && !stop; base += 64) {
        const int h = base + lane;
        const long long itr = (long long)a.first_iter + h;
        bool active = h < a.n_iter && itr < exit_itr;
        int good = -1;
        double e2 = 0.0;
        if (active) { good = inl[h]; e2 = err2[h]; }
        const bool valid = good >= 0;
        const double fit = valid ? (double)good / (double)m : 0.0;
        const double rmse = good > 0 ? sqrt(e2 / (double)good) : 0.0;
        for (;;) {
            const bool cand = active && valid && (fit > best_fit || (fit == best_fit && rmse < best_rmse));
            const unsigned long long mk = __ballot(cand);
            if (!mk) break;
            const int l = (int)__ffsll((long long)mk) - 1;
            best_fit = __shfl(fit, l, 64);
            best_rmse = __shfl(rmse, l, 64);
            best_itr = (long long)a.first_iter + base + l;
            best_h = base + l;
            const double x = 1.0 - pow(best_fit, 3.0);
            const double k = x <= 0.0 ? 0.0 : log(1.0 - a.confidence) / log(x);
            if (k < (double)a.max_iteration) { const long long ke = (long long)ceil(k); if (ke < exit_itr) exit_itr = ke; }
            if (lane > l) active = active && itr < exit_itr;   // what comes after the improvement sees the new exit
        }
        n_valid += __popcll(__ballot(active && valid));
        if ((long long)a.first_iter + base + 64 >= exit_itr) stop = true;   // the next chunk starts at or behind exit_itr
    }
    const long long end = (long long)a.first_iter + a.n_iter;
    if (best_h >= 0 && lane < 12) st->bestT[lane] = Tout[12 * (long long)best_h + lane];
    if (lane == 0) {
        st->best_fit = best_fit; st->best_rmse = best_rmse; st->best_itr = best_itr; st->exit_itr = exit_itr; st->n_valid = n_valid;
        st->done = end < exit_itr ? end : exit_itr;
        if (end >= exit_itr) st->stop = 1;
    }
}

__global__ void ransac_init_kernel(ransac_state* st, const int* m_p, int max_iteration) {
    if (threadIdx.x != 0) return;
    ransac_state z;
    z.best_fit = 0.0; z.best_rmse = 0.0;
    for (int k = 0; k < 12; ++k) z.bestT[k] = 0.0;
    z.best_itr = -1; z.exit_itr = max_iteration; z.done = 0; z.n_valid = 0;
    z.m = *m_p;
    z.stop = z.m < 3 ? 1 : 0;
    z.pad[0] = z.pad[1] = 0;
    *st = z;
}

// correspondence set of registration_ransac_based_on_feature_matching: (i, ij[i]) for every source row, kept when mutual
// (ji[ij[i]] == i); when fewer than `min_mutual` survive, Open3D falls back to the one-way set.  ONE block, rows in order.
__global__ void __launch_bounds__(256) corr_build_kernel(const int* __restrict__ ij, const int* __restrict__ ji, int na, int mutual, int min_mutual,
                                                         int* __restrict__ corr, int* __restrict__ m_out) {
    __shared__ int s_w[4], s_total, s_use;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    int mine = 0;
    if (mutual)
        for (int i = threadIdx.x; i < na; i += 256) mine += (ij[i] >= 0 && ji[ij[i]] == i) ? 1 : 0;
    if (mutual) atomicAdd(&s_total, mine);
    __syncthreads();
    if (threadIdx.x == 0) s_use = (mutual && s_total >= min_mutual) ? 1 : 0;
    __syncthreads();
    const int use_mutual = s_use;
    int base = 0;
    for (int i0 = 0; i0 < na; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool keep = i < na && ij[i] >= 0 && (!use_mutual || ji[ij[i]] == i);
        const unsigned long long mk = __ballot(keep);
        if (lane == 0) s_w[wave] = __popcll(mk);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += s_w[w];
        if (keep) {
            const int r = off + __popcll(mk & ((1ull << lane) - 1ull));
            corr[2 * r] = i;
            corr[2 * r + 1] = ij[i];
        }
        base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *m_out = base;
}

// ------------------------------------------------------------------ host side
namespace {
using dev_buf = pcr_dev_block;

int* fail_word(pcr_ctx* ctx) { return (int*)(ctx->d_counters + 116); }

// A cloud of a few thousand points (what the 2 m down-sample of main.py:35 leaves of a scan: 300 - 1 500 points) is searched without
// an index: two grid builds per scan -- one per radius, ~20 launches each -- cost several times what the neighbourhoods themselves
// cost, and a wave reads 4 096 records in 64 trips.  The "view" of such a cloud: its records in row order, levels = 0.
#ifndef PCR_HYBRID_BRUTE_MAX
#define PCR_HYBRID_BRUTE_MAX 4096
#endif
bool brute_view(const pcr_cloud* cloud, pcr_grid_view* v) {
    static const bool off = getenv("PCR_HYBRID_GRID") != nullptr;   // A/B: always build the grid
    if (off || cloud->n > PCR_HYBRID_BRUTE_MAX || cloud->morton_sorted) return false;
    memset(v, 0, sizeof(*v));
    v->pts = cloud->d;
    v->n = cloud->n;
    v->levels = 0;
    v->cell0 = 1.0; v->inv_cell0 = 1.0;
    return true;
}

// normals of a device cloud into d_normals (n,3 by row); enqueued, not waited for (the grid build inside synchronises once);
// a neighbourhood that cannot be bounded bumps the context's fail word
int hybrid_normals_device(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, int max_nn, int orient, const double* viewpoint, double* d_normals) {
    pcr_index* idx = nullptr;
    pcr_grid_view view;
    if (!brute_view(cloud, &view)) {
        int rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, radius, &idx);
        if (rc) return rc;
        if (!(idx->view.cell0 >= radius)) { pcr_index_free(ctx, idx); ctx->last_error = "radius too small for the cloud's extent"; return PCR_E_UNSUPPORTED; }
        view = idx->view;
    }
    const double v[3] = {viewpoint ? viewpoint[0] : 0.0, viewpoint ? viewpoint[1] : 0.0, viewpoint ? viewpoint[2] : 0.0};
    hipLaunchKernelGGL(hybrid_normals_kernel, dim3((unsigned)cloud->n), dim3(64), 0, ctx->stream, view, (long long)cloud->n, radius * radius, max_nn,
                       orient, v[0], v[1], v[2], d_normals, fail_word(ctx));
    const hipError_t e = hipGetLastError();
    if (idx) pcr_index_free(ctx, idx);   // (stream-ordered)
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PCR_E_HIP; }
    return PCR_OK;
}

// FPFH of a device cloud from device normals into d_out (n,33 by row); enqueued, not waited for
int fpfh_device(pcr_ctx* ctx, const pcr_cloud* cloud, const double* d_normals, double radius, int max_nn, double* d_out) {
    const long long n = cloud->n;
    dev_buf spfh(ctx), nbid(ctx), nbd2(ctx), nbcnt(ctx);
    int rc;
    if ((rc = spfh.alloc(sizeof(double) * 33 * n))) return rc;
    if ((rc = nbid.alloc(sizeof(unsigned int) * (size_t)max_nn * n))) return rc;
    if ((rc = nbd2.alloc(sizeof(double) * (size_t)max_nn * n))) return rc;
    if ((rc = nbcnt.alloc(sizeof(int) * n))) return rc;
    pcr_index* idx = nullptr;
    pcr_grid_view view;
    if (!brute_view(cloud, &view)) {
        rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, radius, &idx);
        if (rc) return rc;
        if (!(idx->view.cell0 >= radius)) { pcr_index_free(ctx, idx); ctx->last_error = "radius too small for the cloud's extent"; return PCR_E_UNSUPPORTED; }
        view = idx->view;
    }
    hipLaunchKernelGGL(spfh_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, view, n, radius * radius, max_nn, d_normals,
                       spfh.as<double>(), nbid.as<unsigned int>(), nbd2.as<double>(), nbcnt.as<int>(), fail_word(ctx));
    hipLaunchKernelGGL(fpfh_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, n, max_nn, (const double*)spfh.as<double>(),
                       (const unsigned int*)nbid.as<unsigned int>(), (const double*)nbd2.as<double>(), (const int*)nbcnt.as<int>(), d_out);
    const hipError_t e = hipGetLastError();
    if (idx) pcr_index_free(ctx, idx);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PCR_E_HIP; }
    return PCR_OK;   // (scratch goes back to the arena stream-ordered)
}

int read_fail(pcr_ctx* ctx) {   // synchronises
    int fail = 0;
    const int rc = pcr_d2h_small(ctx, &fail, fail_word(ctx), sizeof(int));   // (synchronises)
    if (rc) return rc;
    if (fail) hipMemsetAsync(fail_word(ctx), 0, sizeof(int), ctx->stream);
    if (fail) { ctx->last_error = "more than 1024 equidistant neighbours"; return PCR_E_UNSUPPORTED; }
    return PCR_OK;
}

// nearest row of B (nb,dim) for every row of A (na,dim), both on the device; enqueued, not waited for
int feature_match_device(pcr_ctx* ctx, const double* dA, long long na, const double* dB, long long nb, int dim, int* d_idx, double* d_d2) {
    const unsigned grid = (unsigned)((na + 255) / 256);
    const size_t lds = sizeof(double) * FM_TILE * dim;
    // enough blocks to fill the chip: split the targets when there are few query blocks
    int splits = (int)((4ll * ctx->cu_count + grid - 1) / grid);
    const long long max_splits = (nb + FM_TILE - 1) / FM_TILE;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
    if (splits > 256) splits = 256;
    const long long per = ((nb + splits - 1) / splits + FM_TILE - 1) / FM_TILE * FM_TILE;
    splits = (int)((nb + per - 1) / per);
    dev_buf ci(ctx), cd(ctx);
    int rc;
    if ((rc = ci.alloc(sizeof(int) * na * splits))) return rc;
    if ((rc = cd.alloc(sizeof(double) * na * splits))) return rc;
    if (dim == 33)
        hipLaunchKernelGGL(feature_match_kernel<33>, dim3(grid, splits), dim3(256), lds, ctx->stream, dA, na, dB, nb, dim, per, ci.as<int>(), cd.as<double>());
    else
        hipLaunchKernelGGL(feature_match_kernel<0>, dim3(grid, splits), dim3(256), lds, ctx->stream, dA, na, dB, nb, dim, per, ci.as<int>(), cd.as<double>());
    hipLaunchKernelGGL(feature_match_merge_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int*)ci.as<int>(), (const double*)cd.as<double>(), na, splits, d_idx, d_d2);
    PCR_HIP(ctx, hipGetLastError());
    return PCR_OK;
}

// the whole RANSAC loop over the device correspondence set (d_corr, *d_m); clouds by row.  One synchronisation.
int ransac_device(pcr_ctx* ctx, const pcr_pt* d_src, const pcr_pt* d_tgt, const int* d_corr, const int* d_m, const pcr_ransac_params* prm, pcr_ransac_result* res) {
    memset(res, 0, sizeof(*res));
    for (int k = 0; k < 4; ++k) res->T[5 * k] = 1.0;
    const int BATCH = 16384, FIRST = 4096;   // (most registrations exit within the first thousand iterations)
    dev_buf dinl(ctx), derr(ctx), dT(ctx), dst(ctx);
    int rc;
    if ((rc = dinl.alloc(sizeof(int) * BATCH))) return rc;
    if ((rc = derr.alloc(sizeof(double) * BATCH))) return rc;
    if ((rc = dT.alloc(sizeof(double) * 12 * BATCH))) return rc;
    if ((rc = dst.alloc(sizeof(ransac_state)))) return rc;
    ransac_state* const st = dst.as<ransac_state>();
    hipLaunchKernelGGL(ransac_init_kernel, dim3(1), dim3(64), 0, ctx->stream, st, d_m, prm->max_iteration);
    ransac_args a;
    a.src = d_src; a.tgt = d_tgt; a.corr = d_corr;
    a.seed = prm->seed; a.edge_sim = prm->edge_similarity; a.max_dist = prm->max_distance; a.check_distance = prm->check_distance;
    a.max_iteration = prm->max_iteration; a.confidence = prm->confidence;
    // the first two batches (20 480 iterations) go out together -- most registrations exit within the first thousand --, then
    // the state is looked at; what is left of the budget follows in one go (batches behind a stop return at once)
    ransac_state h;
    long long done = 0;
    for (int round = 0; done < prm->max_iteration; ++round) {
        for (int b = 0; done < prm->max_iteration && (round > 0 || b < 2); ++b) {
            const long long want = done == 0 ? FIRST : BATCH;
            const int nb = (int)((prm->max_iteration - done) < want ? (prm->max_iteration - done) : want);
            a.first_iter = (int)done; a.n_iter = nb;
            hipLaunchKernelGGL(ransac_kernel, dim3(nb), dim3(64), 0, ctx->stream, a, (const ransac_state*)st, dinl.as<int>(), derr.as<double>(), dT.as<double>());
            hipLaunchKernelGGL(ransac_walk_kernel, dim3(1), dim3(64), 0, ctx->stream, a, st, (const int*)dinl.as<int>(), (const double*)derr.as<double>(), (const double*)dT.as<double>());
            done += nb;
        }
        PCR_HIP(ctx, hipGetLastError());
        if ((rc = pcr_d2h_small(ctx, &h, st, sizeof(h)))) return rc;   // (synchronises)
        if (h.stop) break;
    }
    if (h.m < 3) return PCR_E_TOO_FEW_ASSOC;
    res->iterations = (int)h.done;
    res->n_valid = (int)h.n_valid;
    res->best_iteration = (int)h.best_itr;
    res->corr_fitness = h.best_fit;
    res->corr_rmse = h.best_rmse;
    res->reserved_i = h.m;   // size of the correspondence set that was sampled
    if (h.best_itr < 0) return PCR_E_TOO_FEW_ASSOC;  // no hypothesis passed the checkers: identity
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) res->T[4 * i + j] = h.bestT[3 * i + j];
        res->T[4 * i + 3] = h.bestT[9 + i];
    }
    return PCR_OK;
}
}  // namespace

// preprocess_point_cloud's result (Registration/main.py:33-47), resident on the device: the down-sampled cloud (records in row
// order: id == position), its normals (n,3) and its FPFH descriptors (n,33)
struct pcr_prep {
    pcr_cloud* down = nullptr;
    double* normals = nullptr;
    double* fpfh = nullptr;
    int64_t n = 0;
};

extern "C" {

int pcr_prep_free(pcr_ctx* ctx, pcr_prep* p) {
    if (!p) return PCR_OK;
    if (!ctx) return PCR_E_INVALID;
    if (p->normals) pcr_dev_free(ctx, p->normals, sizeof(double) * 3 * p->n);
    if (p->fpfh) pcr_dev_free(ctx, p->fpfh, sizeof(double) * 33 * p->n);
    if (p->down) pcr_cloud_free(ctx, p->down);
    delete p;
    return PCR_OK;
}

int64_t pcr_prep_size(const pcr_prep* p) { return p ? p->n : 0; }
const pcr_cloud* pcr_prep_cloud(const pcr_prep* p) { return p ? p->down : nullptr; }

int pcr_preprocess(pcr_ctx* ctx, const pcr_cloud* cloud, double voxel_size, double normal_radius, int normal_max_nn, double fpfh_radius, int fpfh_max_nn,
                   pcr_prep** out) {
    if (!ctx || !cloud || !out || !(voxel_size > 0) || !(normal_radius > 0) || !(fpfh_radius > 0) || normal_max_nn < 1 || normal_max_nn > NB_CAP ||
        fpfh_max_nn < 2 || fpfh_max_nn > NB_CAP)
        return PCR_E_INVALID;
    *out = nullptr;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    pcr_prep* p = new pcr_prep();
    int rc = pcr_voxel_filter_cloud(ctx, cloud, voxel_size, 2, 0, &p->down);   // mode 2 = Open3D's voxel_down_sample (main.py:35)
    if (rc == PCR_OK) {
        p->n = p->down->n;
        rc = pcr_dev_alloc(ctx, sizeof(double) * 3 * p->n, (void**)&p->normals);
    }
    if (rc == PCR_OK) rc = pcr_dev_alloc(ctx, sizeof(double) * 33 * p->n, (void**)&p->fpfh);
    if (rc == PCR_OK) rc = hybrid_normals_device(ctx, p->down, normal_radius, normal_max_nn, 1, nullptr, p->normals);
    if (rc == PCR_OK) rc = fpfh_device(ctx, p->down, p->normals, fpfh_radius, fpfh_max_nn, p->fpfh);
    if (rc == PCR_OK) rc = read_fail(ctx);
    if (rc != PCR_OK) { pcr_sync(ctx->stream); pcr_prep_free(ctx, p); return rc; }
    *out = p;
    return PCR_OK;
}

int pcr_prep_download(pcr_ctx* ctx, const pcr_prep* p, double* points, double* normals, double* features) {
    if (!ctx || !p) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    if (points) { const int rc = pcr_cloud_download_f64(ctx, p->down, points); if (rc) return rc; }
    if (normals) PCR_HIP(ctx, hipMemcpyAsync(normals, p->normals, sizeof(double) * 3 * p->n, hipMemcpyDeviceToHost, ctx->stream));
    if (features) PCR_HIP(ctx, hipMemcpyAsync(features, p->fpfh, sizeof(double) * 33 * p->n, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, pcr_sync(ctx->stream));
    return PCR_OK;
}

int pcr_global_registration(pcr_ctx* ctx, const pcr_prep* source, const pcr_prep* target, const pcr_ransac_params* prm, int mutual_filter,
                            pcr_ransac_result* res) {
    if (!ctx || !source || !target || !prm || !res || prm->max_iteration < 1 || !(prm->max_distance > 0)) return PCR_E_INVALID;
    if (source->n <= 0 || target->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    const long long na = source->n, nb = target->n;
    dev_buf ij(ctx), ji(ctx), dab(ctx), dba(ctx), corr(ctx);
    int rc;
    if ((rc = ij.alloc(sizeof(int) * na)) || (rc = ji.alloc(sizeof(int) * nb)) || (rc = dab.alloc(sizeof(double) * na)) || (rc = dba.alloc(sizeof(double) * nb)) ||
        (rc = corr.alloc(sizeof(int) * 2 * na + 16)))
        return rc;
    if ((rc = feature_match_device(ctx, source->fpfh, na, target->fpfh, nb, 33, ij.as<int>(), dab.as<double>()))) return rc;
    if (mutual_filter && (rc = feature_match_device(ctx, target->fpfh, nb, source->fpfh, na, 33, ji.as<int>(), dba.as<double>()))) return rc;
    int* const d_m = corr.as<int>() + 2 * na;
    hipLaunchKernelGGL(corr_build_kernel, dim3(1), dim3(256), 0, ctx->stream, (const int*)ij.as<int>(), (const int*)ji.as<int>(), (int)na, mutual_filter ? 1 : 0, 9,
                       corr.as<int>(), d_m);
    // the sampled records by ROW: the down-sampled clouds are written in row order, but a caller that has used one as the query
    // cloud of a search since (pcr_nn1 lays its queries out along the index's curve, in place) has re-ordered it
    const pcr_pt *s_rows = source->down->d, *t_rows = target->down->d;
    dev_buf s_tmp(ctx), t_tmp(ctx);
    if (source->down->morton_sorted) {
        if ((rc = s_tmp.alloc(sizeof(pcr_pt) * na)) || (rc = pcr_cloud_rows(ctx, source->down, s_tmp.as<pcr_pt>()))) return rc;
        s_rows = s_tmp.as<pcr_pt>();
    }
    if (target->down->morton_sorted) {
        if ((rc = t_tmp.alloc(sizeof(pcr_pt) * nb)) || (rc = pcr_cloud_rows(ctx, target->down, t_tmp.as<pcr_pt>()))) return rc;
        t_rows = t_tmp.as<pcr_pt>();
    }
    return ransac_device(ctx, s_rows, t_rows, corr.as<int>(), d_m, prm, res);
}

int pcr_normals_hybrid(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, int max_nn, int orient, const double viewpoint[3], double* normals_out) {
    if (!ctx || !cloud || !normals_out || !(radius > 0) || max_nn < 1 || max_nn > NB_CAP) return PCR_E_INVALID;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    dev_buf nrm(ctx);
    int rc = nrm.alloc(sizeof(double) * 3 * cloud->n);
    if (rc) return rc;
    rc = hybrid_normals_device(ctx, cloud, radius, max_nn, orient, viewpoint, nrm.as<double>());
    if (rc) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(normals_out, nrm.p, sizeof(double) * 3 * cloud->n, hipMemcpyDeviceToHost, ctx->stream));
    return read_fail(ctx);
}

int pcr_fpfh(pcr_ctx* ctx, const pcr_cloud* cloud, const double* normals, double radius, int max_nn, double* features_out) {
    if (!ctx || !cloud || !normals || !features_out || !(radius > 0) || max_nn < 2 || max_nn > NB_CAP) return PCR_E_INVALID;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    const long long n = cloud->n;
    dev_buf nrm(ctx), out(ctx);
    int rc;
    if ((rc = nrm.alloc(sizeof(double) * 3 * n))) return rc;
    if ((rc = out.alloc(sizeof(double) * 33 * n))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(nrm.p, normals, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = fpfh_device(ctx, cloud, nrm.as<double>(), radius, max_nn, out.as<double>()))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(features_out, out.p, sizeof(double) * 33 * n, hipMemcpyDeviceToHost, ctx->stream));
    return read_fail(ctx);
}

int pcr_feature_match(pcr_ctx* ctx, const double* queries, int64_t nq, const double* targets, int64_t nt, int dim, int32_t* idx_out, double* d2_out) {
    if (!ctx || !queries || !targets || !idx_out || dim < 1 || dim > 512) return PCR_E_INVALID;
    if (nq <= 0 || nt <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    dev_buf a(ctx), b(ctx), di(ctx), dd(ctx);
    int rc;
    if ((rc = a.alloc(sizeof(double) * dim * nq))) return rc;
    if ((rc = b.alloc(sizeof(double) * dim * nt))) return rc;
    if ((rc = di.alloc(sizeof(int) * nq))) return rc;
    if ((rc = dd.alloc(sizeof(double) * nq))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(a.p, queries, sizeof(double) * dim * nq, hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(b.p, targets, sizeof(double) * dim * nt, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = feature_match_device(ctx, a.as<double>(), nq, b.as<double>(), nt, dim, di.as<int>(), dd.as<double>()))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(idx_out, di.p, sizeof(int) * nq, hipMemcpyDeviceToHost, ctx->stream));
    if (d2_out) PCR_HIP(ctx, hipMemcpyAsync(d2_out, dd.p, sizeof(double) * nq, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, pcr_sync(ctx->stream));
    return PCR_OK;
}

int pcr_ransac_default_params(pcr_ransac_params* p) {
    if (!p) return PCR_E_INVALID;
    memset(p, 0, sizeof(*p));
    p->max_iteration = 100000;   // RANSACConvergenceCriteria(100000, 0.999), main.py:83
    p->confidence = 0.999;
    p->max_distance = 3.0;       // voxel_size * 1.5 with voxel_size = 2.0, main.py:70,197
    p->edge_similarity = 0.9;    // CorrespondenceCheckerBasedOnEdgeLength(0.9), main.py:78-79
    p->check_distance = 1;       // CorrespondenceCheckerBasedOnDistance, main.py:80-81
    p->seed = 0;
    return PCR_OK;
}

int pcr_ransac(pcr_ctx* ctx, const pcr_cloud* source, const pcr_cloud* target, const int32_t* corr, int64_t m, const pcr_ransac_params* prm,
               pcr_ransac_result* res) {
    if (!ctx || !source || !target || !corr || !prm || !res || prm->max_iteration < 1 || !(prm->max_distance > 0)) return PCR_E_INVALID;
    memset(res, 0, sizeof(*res));
    for (int k = 0; k < 4; ++k) res->T[5 * k] = 1.0;
    if (m < 3) return PCR_E_TOO_FEW_ASSOC;
    if (m > 0x7fffffffll) return PCR_E_UNSUPPORTED;
    for (int64_t c = 0; c < m; ++c)
        if (corr[2 * c] < 0 || corr[2 * c] >= source->n || corr[2 * c + 1] < 0 || corr[2 * c + 1] >= target->n) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    // clouds in caller row order
    dev_buf s(ctx), t(ctx), dc(ctx);
    int rc;
    if ((rc = s.alloc(sizeof(pcr_pt) * source->n))) return rc;
    if ((rc = t.alloc(sizeof(pcr_pt) * target->n))) return rc;
    if ((rc = dc.alloc(sizeof(int) * 2 * m + 16))) return rc;
    if ((rc = pcr_cloud_rows(ctx, source, s.as<pcr_pt>()))) return rc;
    if ((rc = pcr_cloud_rows(ctx, target, t.as<pcr_pt>()))) return rc;
    const int m32 = (int)m;
    PCR_HIP(ctx, hipMemcpyAsync(dc.p, corr, sizeof(int) * 2 * m, hipMemcpyHostToDevice, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(dc.as<int>() + 2 * m, &m32, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return ransac_device(ctx, s.as<pcr_pt>(), t.as<pcr_pt>(), dc.as<int>(), dc.as<int>() + 2 * m, prm, res);
}

}  // extern "C"
