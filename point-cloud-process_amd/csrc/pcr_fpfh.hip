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
__device__ static int gather_hybrid(const pcr_grid_view& gv, double qx, double qy, double qz, double r2, int max_nn, nb_entry* nb) {
    const int lane = threadIdx.x;
    bool clamped = false;
    const int cx = cell_coord(qx, gv.lo[0], gv.inv_cell0, &clamped);
    const int cy = cell_coord(qy, gv.lo[1], gv.inv_cell0, &clamped);
    const int cz = cell_coord(qz, gv.lo[2], gv.inv_cell0, &clamped);
    auto scan = [&](double T, bool store) -> int {
        int total = 0;
        for (int c = 0; c < 27; ++c) {
            const unsigned int nx = (unsigned int)(cx + c % 3 - 1), ny = (unsigned int)(cy + (c / 3) % 3 - 1), nz = (unsigned int)(cz + c / 9 - 1);
            if (nx > (unsigned int)PCR_COORD_MAX || ny > (unsigned int)PCR_COORD_MAX || nz > (unsigned int)PCR_COORD_MAX) continue;
            unsigned int s, e;
            if (!lookup_cell(gv.table[0], gv.mask[0], nx, ny, nz, &s, &e)) continue;
            for (unsigned int base = s; base < e; base += 64) {
                const unsigned int j = base + lane;
                bool keep = false;
                nb_entry en;
                if (j < e) {
                    const pcr_pt b = gv.pts[j];
                    en.d2 = dist2(qx, qy, qz, b);
                    en.pos = j;
                    en.id = (unsigned int)b.id;
                    keep = en.d2 < T;
                }
                const unsigned long long m = __ballot(keep);
                const int rank = __popcll(m & ((1ull << lane) - 1ull));
                if (store && keep && total + rank < NB_CAP) nb[total + rank] = en;
                total += __popcll(m);
            }
        }
        return total;
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
    __shared__ nb_entry nb[NB_CAP];
    const long long i = blockIdx.x;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    const int cnt = gather_hybrid(gv, p.x, p.y, p.z, r2, max_nn, nb);
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
    __shared__ nb_entry nb[NB_CAP];
    __shared__ int hist[33];
    const long long i = blockIdx.x;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    const int cnt = gather_hybrid(gv, p.x, p.y, p.z, r2, max_nn, nb);  // ends with a barrier
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
__global__ void __launch_bounds__(64)
fpfh_kernel(long long n, int max_nn, const double* __restrict__ spfh, const unsigned int* __restrict__ nb_id, const double* __restrict__ nb_d2,
            const int* __restrict__ nb_cnt, double* __restrict__ fpfh /* (n,33) by row */) {
    __shared__ double gsum[3];
    const long long i = blockIdx.x;
    if (i >= n) return;
    const int cnt = nb_cnt[i];
    const int lane = threadIdx.x;
    double acc = 0.0;
    if (cnt > 1) {
        if (lane < 33) {
            for (int k = 1; k < cnt; ++k) {
                const double d2 = nb_d2[i * max_nn + k];
                if (d2 == 0.0) continue;
                acc += spfh[33 * (long long)nb_id[i * max_nn + k] + lane] / d2;
            }
        } else if (lane < 36) {
            const int g = lane - 33;
            double s = 0.0;
            for (int k = 1; k < cnt; ++k) {
                const double d2 = nb_d2[i * max_nn + k];
                if (d2 == 0.0) continue;
                const double* row = spfh + 33 * (long long)nb_id[i * max_nn + k] + 11 * g;
                for (int j = 0; j < 11; ++j) s += row[j] / d2;
            }
            gsum[g] = s != 0.0 ? 100.0 / s : 0.0;
        }
    }
    __syncthreads();
    if (lane < 33) {
        double v = 0.0;
        if (cnt > 1) v = acc * gsum[lane / 11] + spfh[33 * i + lane];
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
struct ransac_args {
    const pcr_pt* src;  // by row (id == position)
    const pcr_pt* tgt;
    const int* corr;    // (m,2)
    int m;
    int first_iter, n_iter;
    unsigned long long seed;
    double edge_sim;    // <= 0: checker off
    double max_dist;    // inlier threshold and distance checker
    int check_distance;
};

__host__ __device__ static inline unsigned long long mix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// one wave per hypothesis; out: inl[h] (-1 = rejected by a checker), err2[h], T[h][12]
__global__ void __launch_bounds__(64) ransac_kernel(ransac_args a, int* __restrict__ inl, double* __restrict__ err2, double* __restrict__ Tout) {
    const int h = blockIdx.x;
    if (h >= a.n_iter) return;
    const unsigned long long itr = (unsigned long long)(a.first_iter + h);
    double s[3][3], t[3][3];
    for (int j = 0; j < 3; ++j) {
        const unsigned int c = (unsigned int)(mix64(a.seed ^ mix64(itr * 3 + j)) % (unsigned long long)a.m);
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
        double m[18];
        for (int k = 0; k < 18; ++k) m[k] = 0.0;
        const double org[3] = {s[0][0], s[0][1], s[0][2]};
        m[0] = 3.0;
        for (int j = 0; j < 3; ++j) {
            const double ax = s[j][0] - org[0], ay = s[j][1] - org[1], az = s[j][2] - org[2];
            const double bx = t[j][0] - org[0], by = t[j][1] - org[1], bz = t[j][2] - org[2];
            m[1] += ax; m[2] += ay; m[3] += az;
            m[4] += bx; m[5] += by; m[6] += bz;
            m[7] += bx * ax; m[8] += bx * ay; m[9] += bx * az;
            m[10] += by * ax; m[11] += by * ay; m[12] += by * az;
            m[13] += bz * ax; m[14] += bz * ay; m[15] += bz * az;
            m[16] += (ax * ax + ay * ay) + az * az;
            m[17] += (bx * bx + by * by) + bz * bz;
        }
        pcr::kabsch_from_moments(m, org, R, tr, nullptr);
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
    for (int c = threadIdx.x; c < a.m; c += 64) {
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

// ------------------------------------------------------------------ host side
namespace {
struct dev_buf {
    pcr_ctx* ctx;
    void* p = nullptr;
    size_t bytes = 0;
    explicit dev_buf(pcr_ctx* c) : ctx(c) {}
    int alloc(size_t b) { bytes = b; return pcr_dev_alloc(ctx, b, &p); }
    ~dev_buf() { if (p) pcr_dev_free(ctx, p, bytes); }
    template <class T> T* as() { return (T*)p; }
};
}  // namespace

static int hybrid_normals_device(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, int max_nn, int orient, const double* viewpoint, double* d_normals) {
    pcr_index* idx = nullptr;
    int rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, radius, &idx);
    if (rc) return rc;
    if (!(idx->view.cell0 >= radius)) { pcr_index_free(ctx, idx); ctx->last_error = "radius too small for the cloud's extent"; return PCR_E_UNSUPPORTED; }
    int* d_fail = (int*)(ctx->d_counters + 116);
    hipMemsetAsync(d_fail, 0, sizeof(int), ctx->stream);
    const double v[3] = {viewpoint ? viewpoint[0] : 0.0, viewpoint ? viewpoint[1] : 0.0, viewpoint ? viewpoint[2] : 0.0};
    hipLaunchKernelGGL(hybrid_normals_kernel, dim3((unsigned)cloud->n), dim3(64), 0, ctx->stream, idx->view, (long long)cloud->n, radius * radius, max_nn,
                       orient, v[0], v[1], v[2], d_normals, d_fail);
    int fail = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    pcr_index_free(ctx, idx);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PCR_E_HIP; }
    if (fail) { ctx->last_error = "more than 1024 equidistant neighbours"; return PCR_E_UNSUPPORTED; }
    return PCR_OK;
}

extern "C" {

int pcr_normals_hybrid(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, int max_nn, int orient, const double viewpoint[3], double* normals_out) {
    if (!ctx || !cloud || !normals_out || !(radius > 0) || max_nn < 1 || max_nn > NB_CAP) return PCR_E_INVALID;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    dev_buf nrm(ctx);
    int rc = nrm.alloc(sizeof(double) * 3 * cloud->n);
    if (rc) return rc;
    rc = hybrid_normals_device(ctx, cloud, radius, max_nn, orient, viewpoint, nrm.as<double>());
    if (rc) return rc;
    PCR_HIP(ctx, hipMemcpy(normals_out, nrm.p, sizeof(double) * 3 * cloud->n, hipMemcpyDeviceToHost));
    return PCR_OK;
}

int pcr_fpfh(pcr_ctx* ctx, const pcr_cloud* cloud, const double* normals, double radius, int max_nn, double* features_out) {
    if (!ctx || !cloud || !normals || !features_out || !(radius > 0) || max_nn < 2 || max_nn > NB_CAP) return PCR_E_INVALID;
    if (cloud->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    const long long n = cloud->n;
    dev_buf nrm(ctx), spfh(ctx), out(ctx), nbid(ctx), nbd2(ctx), nbcnt(ctx);
    int rc;
    if ((rc = nrm.alloc(sizeof(double) * 3 * n))) return rc;
    if ((rc = spfh.alloc(sizeof(double) * 33 * n))) return rc;
    if ((rc = out.alloc(sizeof(double) * 33 * n))) return rc;
    if ((rc = nbid.alloc(sizeof(unsigned int) * (size_t)max_nn * n))) return rc;
    if ((rc = nbd2.alloc(sizeof(double) * (size_t)max_nn * n))) return rc;
    if ((rc = nbcnt.alloc(sizeof(int) * n))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(nrm.p, normals, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream));
    pcr_index* idx = nullptr;
    rc = pcr_index_build(ctx, cloud, PCR_INDEX_GRID, radius, &idx);
    if (rc) return rc;
    if (!(idx->view.cell0 >= radius)) { pcr_index_free(ctx, idx); ctx->last_error = "radius too small for the cloud's extent"; return PCR_E_UNSUPPORTED; }
    int* d_fail = (int*)(ctx->d_counters + 116);
    hipMemsetAsync(d_fail, 0, sizeof(int), ctx->stream);
    hipLaunchKernelGGL(spfh_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, idx->view, n, radius * radius, max_nn, (const double*)nrm.as<double>(),
                       spfh.as<double>(), nbid.as<unsigned int>(), nbd2.as<double>(), nbcnt.as<int>(), d_fail);
    hipLaunchKernelGGL(fpfh_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, n, max_nn, (const double*)spfh.as<double>(),
                       (const unsigned int*)nbid.as<unsigned int>(), (const double*)nbd2.as<double>(), (const int*)nbcnt.as<int>(), out.as<double>());
    int fail = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&fail, d_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(features_out, out.p, sizeof(double) * 33 * n, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    pcr_index_free(ctx, idx);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return PCR_E_HIP; }
    if (fail) { ctx->last_error = "more than 1024 equidistant neighbours"; return PCR_E_UNSUPPORTED; }
    return PCR_OK;
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
    const unsigned grid = (unsigned)((nq + 255) / 256);
    const size_t lds = sizeof(double) * FM_TILE * dim;
    // enough blocks to fill the chip: split the targets when there are few query blocks
    int splits = (int)((4ll * ctx->cu_count + grid - 1) / grid);
    const long long max_splits = (nt + FM_TILE - 1) / FM_TILE;
    if (splits > max_splits) splits = (int)max_splits;
    if (splits < 1) splits = 1;
    if (splits > 256) splits = 256;
    const long long per = ((nt + splits - 1) / splits + FM_TILE - 1) / FM_TILE * FM_TILE;
    splits = (int)((nt + per - 1) / per);
    dev_buf ci(ctx), cd(ctx);
    if ((rc = ci.alloc(sizeof(int) * nq * splits))) return rc;
    if ((rc = cd.alloc(sizeof(double) * nq * splits))) return rc;
    if (dim == 33)
        hipLaunchKernelGGL(feature_match_kernel<33>, dim3(grid, splits), dim3(256), lds, ctx->stream, (const double*)a.as<double>(), (long long)nq,
                           (const double*)b.as<double>(), (long long)nt, dim, per, ci.as<int>(), cd.as<double>());
    else
        hipLaunchKernelGGL(feature_match_kernel<0>, dim3(grid, splits), dim3(256), lds, ctx->stream, (const double*)a.as<double>(), (long long)nq,
                           (const double*)b.as<double>(), (long long)nt, dim, per, ci.as<int>(), cd.as<double>());
    hipLaunchKernelGGL(feature_match_merge_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const int*)ci.as<int>(), (const double*)cd.as<double>(),
                       (long long)nq, splits, di.as<int>(), dd.as<double>());
    PCR_HIP(ctx, hipGetLastError());
    PCR_HIP(ctx, hipMemcpyAsync(idx_out, di.p, sizeof(int) * nq, hipMemcpyDeviceToHost, ctx->stream));
    if (d2_out) PCR_HIP(ctx, hipMemcpyAsync(d2_out, dd.p, sizeof(double) * nq, hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
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
    for (int64_t c = 0; c < m; ++c)
        if (corr[2 * c] < 0 || corr[2 * c] >= source->n || corr[2 * c + 1] < 0 || corr[2 * c + 1] >= target->n) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    // clouds in caller row order
    dev_buf s(ctx), t(ctx), dc(ctx), dinl(ctx), derr(ctx), dT(ctx);
    const int BATCH = 16384;
    int rc;
    if ((rc = s.alloc(sizeof(pcr_pt) * source->n))) return rc;
    if ((rc = t.alloc(sizeof(pcr_pt) * target->n))) return rc;
    if ((rc = dc.alloc(sizeof(int) * 2 * m))) return rc;
    if ((rc = dinl.alloc(sizeof(int) * BATCH))) return rc;
    if ((rc = derr.alloc(sizeof(double) * BATCH))) return rc;
    if ((rc = dT.alloc(sizeof(double) * 12 * BATCH))) return rc;
    if ((rc = pcr_cloud_rows(ctx, source, s.as<pcr_pt>()))) return rc;
    if ((rc = pcr_cloud_rows(ctx, target, t.as<pcr_pt>()))) return rc;
    PCR_HIP(ctx, hipMemcpyAsync(dc.p, corr, sizeof(int) * 2 * m, hipMemcpyHostToDevice, ctx->stream));
    std::vector<int> inl(BATCH);
    std::vector<double> err(BATCH);
    double best_fit = 0.0, best_rmse = 0.0;
    long long best_itr = -1;
    double bestT[12];
    long long exit_itr = prm->max_iteration;
    long long done = 0, n_valid = 0;
    bool stop = false;
    while (!stop && done < exit_itr) {
        const int nb = (int)((exit_itr - done) < BATCH ? (exit_itr - done) : BATCH);
        ransac_args a;
        a.src = s.as<pcr_pt>(); a.tgt = t.as<pcr_pt>(); a.corr = dc.as<int>(); a.m = (int)m;
        a.first_iter = (int)done; a.n_iter = nb; a.seed = prm->seed;
        a.edge_sim = prm->edge_similarity; a.max_dist = prm->max_distance; a.check_distance = prm->check_distance;
        hipLaunchKernelGGL(ransac_kernel, dim3(nb), dim3(64), 0, ctx->stream, a, dinl.as<int>(), derr.as<double>(), dT.as<double>());
        PCR_HIP(ctx, hipGetLastError());
        PCR_HIP(ctx, hipMemcpyAsync(inl.data(), dinl.p, sizeof(int) * nb, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipMemcpyAsync(err.data(), derr.p, sizeof(double) * nb, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // the sequential loop of the reference library: iteration order, running best, early exit
        for (int h = 0; h < nb; ++h) {
            const long long itr = done + h;
            if (itr >= exit_itr) { stop = true; break; }
            if (inl[h] < 0) continue;
            ++n_valid;
            const double fit = (double)inl[h] / (double)m;
            const double rmse = inl[h] > 0 ? sqrt(err[h] / (double)inl[h]) : 0.0;
            if (fit > best_fit || (fit == best_fit && rmse < best_rmse)) {  // IsBetterRANSACThan; the initial best is (0, 0)
                best_fit = fit; best_rmse = rmse; best_itr = itr;
                PCR_HIP(ctx, hipMemcpy(bestT, dT.as<double>() + 12 * (size_t)h, sizeof(bestT), hipMemcpyDeviceToHost));
                const double x = 1.0 - pow(fit, 3.0);
                const double k = x <= 0.0 ? 0.0 : log(1.0 - prm->confidence) / log(x);
                if (k < (double)prm->max_iteration) { const long long ke = (long long)ceil(k); if (ke < exit_itr) exit_itr = ke; }
            }
        }
        done += nb;
    }
    res->iterations = (int)(done < exit_itr ? done : exit_itr);
    res->n_valid = (int)n_valid;
    res->best_iteration = (int)best_itr;
    res->corr_fitness = best_fit;
    res->corr_rmse = best_rmse;
    if (best_itr < 0) return PCR_E_TOO_FEW_ASSOC;  // no hypothesis passed the checkers: identity
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) res->T[4 * i + j] = bestT[3 * i + j];
        res->T[4 * i + 3] = bestT[9 + i];
    }
    return PCR_OK;
}

}  // extern "C"
