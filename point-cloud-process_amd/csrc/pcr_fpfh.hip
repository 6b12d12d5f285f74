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
#include <chrono>
#include <atomic>
#include <thread>
#include "pcr_grid_dev.h"
#include "pcr_linalg.h"

constexpr int NB_CAP = 1024;
// clouds of up to this many points are searched without an index (brute_view; the fused initialisation takes only such scans)
#ifndef PCR_HYBRID_BRUTE_MAX
#define PCR_HYBRID_BRUTE_MAX 4096
#endif

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
// CAP < NB_CAP: a block with a small candidate array (more blocks per CU); a sphere that holds more returns -2 and the point is done
// again by the block with the full array.
template <int CAP>
struct hybrid_lds_t {
    nb_entry nb[CAP];
    unsigned int cell_s[28], cell_o[28];
};
typedef hybrid_lds_t<NB_CAP> hybrid_lds;
template <int CAP>
__device__ static int gather_hybrid(const pcr_grid_view& gv, double qx, double qy, double qz, double r2, int max_nn, hybrid_lds_t<CAP>* L) {
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
            if (store && keep && found + rank < CAP) nb[found + rank] = en;
            found += __popcll(m);
        }
        return found;
    };
    int cnt = scan(r2, true);
    if (CAP < NB_CAP && cnt > CAP) return -2;
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
// returns false when the point has to be done again with the full candidate array
// the normal of a point from the covariance S of its neighbourhood (cnt < 3: Open3D's (0, 0, 1)); the same arithmetic whoever runs it
__device__ static inline void normal_from_cov(const double S[6], int cnt, const pcr_pt& p, int orient, double vx, double vy, double vz, double nrm[3]) {
    nrm[0] = 0.0; nrm[1] = 0.0; nrm[2] = 1.0;
    if (cnt < 3) return;
    smallest_eigvec(S, nrm);
    const double len = sqrt(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
    if (len == 0.0 || !(len == len)) { nrm[0] = 0; nrm[1] = 0; nrm[2] = 1; }
    else if (orient) {
        const double d = nrm[0] * (vx - p.x) + nrm[1] * (vy - p.y) + nrm[2] * (vz - p.z);
        if (d < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
    }
}
// `cov` (or null): the covariance and the count go to cov[7 * row .. + 7) and the normal is left to normals_finish_kernel -- one THREAD
// per point there: the Jacobi sweeps are ~1 000 dependent instructions, 40 % of this kernel's when a whole wave runs them for one point
template <int CAP>
__device__ static bool normals_body(const pcr_grid_view& gv, const pcr_pt& p, double r2, int max_nn, int orient, double vx, double vy, double vz,
                                    double* __restrict__ normals /* (n,3) by row */, int* __restrict__ fail, hybrid_lds_t<CAP>* L, double* __restrict__ cov = nullptr) {
    nb_entry* const nb = L->nb;
    const int cnt = gather_hybrid<CAP>(gv, p.x, p.y, p.z, r2, max_nn, L);
    if (cnt == -2) return false;
    if (cnt < 0) {
        if (threadIdx.x == 0) { atomicAdd(fail, 1); if (cov) cov[7 * p.id + 6] = 0.0; }
        return true;
    }
    double S[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
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
        S[0] = c[3] - c[0] * c[0]; S[1] = c[4] - c[0] * c[1]; S[2] = c[5] - c[0] * c[2];
        S[3] = c[6] - c[1] * c[1]; S[4] = c[7] - c[1] * c[2]; S[5] = c[8] - c[2] * c[2];
    }
    if (cov) {
        if (threadIdx.x < 7) {
            const double v = threadIdx.x == 0 ? S[0] : threadIdx.x == 1 ? S[1] : threadIdx.x == 2 ? S[2] : threadIdx.x == 3 ? S[3] : threadIdx.x == 4 ? S[4] : threadIdx.x == 5 ? S[5] : (double)cnt;
            cov[7 * p.id + threadIdx.x] = v;
        }
        return true;
    }
    double nrm[3];
    normal_from_cov(S, cnt, p, orient, vx, vy, vz, nrm);
    const double n0 = nrm[0], n1 = nrm[1], n2 = nrm[2];
    if (threadIdx.x < 3) normals[3 * p.id + threadIdx.x] = threadIdx.x == 0 ? n0 : (threadIdx.x == 1 ? n1 : n2);
    return true;
}

__global__ void __launch_bounds__(64) hybrid_normals_kernel(pcr_grid_view gv, long long n, double r2, int max_nn, int orient, double vx, double vy,
                                                            double vz, double* __restrict__ normals /* (n,3) by row */, int* __restrict__ fail) {
    __shared__ hybrid_lds s_L;
    const long long i = blockIdx.x;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    normals_body<NB_CAP>(gv, p, r2, max_nn, orient, vx, vy, vz, normals, fail, &s_L);
}

// The down-sampled scans of a chunk, one behind the other (pcr_voxel_downsample_scans): record v belongs to scan vsid[v], whose records
// are [scan_first[s], scan_first[s + 1]) with id = row within the scan.  A block's "grid view" is its own scan, searched without an index.
struct scans_view {
    const pcr_pt* down;
    const unsigned int* vsid;
    const unsigned int* scan_first;
};
__device__ static inline unsigned int scans_block_view(const scans_view& V, long long i, pcr_grid_view* gv) {
    const unsigned int s = V.vsid[i], base = V.scan_first[s];
    gv->pts = V.down + base;
    gv->n = (long long)(V.scan_first[s + 1] - base);
    gv->levels = 0;
    gv->lo[0] = gv->lo[1] = gv->lo[2] = 0.0;
    gv->cell0 = 1.0; gv->inv_cell0 = 1.0;
    return base;
}
// Two launches per stage: every point with a SMALL candidate array (16 blocks and more per CU instead of 9: a block is a chain of round
// trips and LDS sorts, its throughput is how many run side by side), then the few whose sphere holds more, from the list the first left
// (`todo` / `todo_count`; a fixed grid strides over it -- its length is only known on the device).
template <int CAP>
__global__ void __launch_bounds__(64) normals_scans_kernel(scans_view V, long long ng, double r2, int max_nn, double* __restrict__ normals /* (ng,3) */, int* __restrict__ fail,
                                                           const unsigned int* __restrict__ todo, const unsigned int* __restrict__ todo_count, unsigned int* __restrict__ redo,
                                                           unsigned int* __restrict__ redo_count, double* __restrict__ cov /* (ng,7): covariance + count; the normals follow in normals_finish_kernel */) {
    __shared__ hybrid_lds_t<CAP> s_L;
    const long long n_do = todo ? (long long)*todo_count : ng;
    for (long long t = blockIdx.x; t < n_do; t += gridDim.x) {
        const long long i = todo ? (long long)todo[t] : t;
        pcr_grid_view gv;
        const unsigned int base = scans_block_view(V, i, &gv);
        const pcr_pt p = V.down[i];
        const bool done = normals_body<CAP>(gv, p, r2, max_nn, 1, 0.0, 0.0, 0.0, normals + 3 * (size_t)base, fail, &s_L, cov + 7 * (size_t)base);
        if (!done && threadIdx.x == 0) redo[atomicAdd(redo_count, 1u)] = (unsigned int)i;
        __syncthreads();   // (the candidate array is reused by the next point)
    }
}
// one thread per down-sampled point of the chunk: covariance -> normal (towards the origin, as pcr_preprocess asks)
__global__ void __launch_bounds__(256) normals_finish_kernel(scans_view V, long long ng, const double* __restrict__ cov, double* __restrict__ normals) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ng) return;
    const pcr_pt p = V.down[i];   // (record base + r is row r: cov and normals of point i sit at i)
    double S[6], nrm[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) S[k] = cov[7 * i + k];
    normal_from_cov(S, (int)cov[7 * i + 6], p, 1, 0.0, 0.0, 0.0, nrm);
    normals[3 * i] = nrm[0]; normals[3 * i + 1] = nrm[1]; normals[3 * i + 2] = nrm[2];
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

template <int CAP>
__device__ static bool spfh_body(const pcr_grid_view& gv, const pcr_pt& p, double r2, int max_nn, const double* __restrict__ normals /* by row */,
                                 double* __restrict__ spfh /* (n,33) by row */, unsigned int* __restrict__ nb_id /* (n,max_nn) by row */, double* __restrict__ nb_d2,
                                 int* __restrict__ nb_cnt, int* __restrict__ fail, hybrid_lds_t<CAP>* L, int* hist /* LDS, 33 */) {
    nb_entry* const nb = L->nb;
    if (threadIdx.x < 33) hist[threadIdx.x] = 0;
    const int cnt = gather_hybrid<CAP>(gv, p.x, p.y, p.z, r2, max_nn, L);  // ends with a barrier
    if (cnt == -2) return false;
    if (cnt < 0) { if (threadIdx.x == 0) atomicAdd(fail, 1); return true; }
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
    return true;
}

__global__ void __launch_bounds__(64)
spfh_kernel(pcr_grid_view gv, long long n, double r2, int max_nn, const double* __restrict__ normals /* by row */, double* __restrict__ spfh /* (n,33) by row */,
            unsigned int* __restrict__ nb_id /* (n,max_nn) by row */, double* __restrict__ nb_d2, int* __restrict__ nb_cnt, int* __restrict__ fail) {
    __shared__ hybrid_lds s_L;
    __shared__ int hist[33];
    const long long i = blockIdx.x;
    if (i >= n) return;
    const pcr_pt p = gv.pts[i];
    spfh_body<NB_CAP>(gv, p, r2, max_nn, normals, spfh, nb_id, nb_d2, nb_cnt, fail, &s_L, hist);
}

template <int CAP>
__global__ void __launch_bounds__(64)
spfh_scans_kernel(scans_view V, long long ng, double r2, int max_nn, const double* __restrict__ normals, double* __restrict__ spfh, unsigned int* __restrict__ nb_id,
                  double* __restrict__ nb_d2, int* __restrict__ nb_cnt, int* __restrict__ fail, const unsigned int* __restrict__ todo,
                  const unsigned int* __restrict__ todo_count, unsigned int* __restrict__ redo, unsigned int* __restrict__ redo_count) {
    __shared__ hybrid_lds_t<CAP> s_L;
    __shared__ int hist[33];
    const long long n_do = todo ? (long long)*todo_count : ng;
    for (long long t = blockIdx.x; t < n_do; t += gridDim.x) {
        const long long i = todo ? (long long)todo[t] : t;
        pcr_grid_view gv;
        const size_t base = scans_block_view(V, i, &gv);
        const pcr_pt p = V.down[i];
        const bool done = spfh_body<CAP>(gv, p, r2, max_nn, normals + 3 * base, spfh + 33 * base, nb_id + base * (size_t)max_nn, nb_d2 + base * (size_t)max_nn, nb_cnt + base,
                                         fail, &s_L, hist);
        if (!done && threadIdx.x == 0) redo[atomicAdd(redo_count, 1u)] = (unsigned int)i;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------- FPFH
// One wave per point, lane = histogram bin (33 of 64).  The neighbour list (row, d^2) goes through LDS first, so the SPFH rows
// of four neighbours are requested together instead of one dependent chain id -> row per neighbour; every term is spfh / d^2
// (a true division, as in Open3D); the three renormalising sums are taken over the lanes of each 11-bin block at the end.
__device__ static void fpfh_body(const long long i, int max_nn, const double* __restrict__ spfh, const unsigned int* __restrict__ nb_id, const double* __restrict__ nb_d2,
                                 const int* __restrict__ nb_cnt, double* __restrict__ fpfh /* (n,33) by row */, unsigned int* s_id, double* s_d2, double* s_acc) {
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

__global__ void __launch_bounds__(64)
fpfh_kernel(long long n, int max_nn, const double* __restrict__ spfh, const unsigned int* __restrict__ nb_id, const double* __restrict__ nb_d2,
            const int* __restrict__ nb_cnt, double* __restrict__ fpfh /* (n,33) by row */) {
    __shared__ unsigned int s_id[NB_CAP];
    __shared__ double s_d2[NB_CAP];
    __shared__ double s_acc[33];
    const long long i = blockIdx.x;
    if (i >= n) return;
    fpfh_body(i, max_nn, spfh, nb_id, nb_d2, nb_cnt, fpfh, s_id, s_d2, s_acc);
}

// (the records of a scan sit in row order: record base + r is row r)
template <int CAP>   // >= max_nn (a list never holds more)
__global__ void __launch_bounds__(64)
fpfh_scans_kernel(scans_view V, long long ng, int max_nn, const double* __restrict__ spfh, const unsigned int* __restrict__ nb_id, const double* __restrict__ nb_d2,
                  const int* __restrict__ nb_cnt, double* __restrict__ fpfh) {
    __shared__ unsigned int s_id[CAP];
    __shared__ double s_d2[CAP];
    __shared__ double s_acc[33];
    const long long i = blockIdx.x;
    if (i >= ng) return;
    const size_t base = V.scan_first[V.vsid[i]];
    fpfh_body(i - (long long)base, max_nn, spfh + 33 * base, nb_id + base * (size_t)max_nn, nb_d2 + base * (size_t)max_nn, nb_cnt + base, fpfh + 33 * base, s_id, s_d2, s_acc);
}

// ----------------------------------------------------------- feature matching
// One thread per query row, target rows staged through LDS in tiles; squared L2 summed over the
// dimensions in order; ties to the lowest target row.
constexpr int FM_TILE = 32;
// grid = (query blocks, target splits): block (bx, by) scans targets [by * per, (by + 1) * per); a small merge kernel
// takes the minimum over the splits (ascending split order + strict comparison keeps the lowest row on ties).
template <int DIM>
__device__ static void feature_match_body(const double* __restrict__ A, long long na, const double* __restrict__ B, long long nb, int dim_rt, long long per,
                                          int* __restrict__ idx_out, double* __restrict__ d2_out, const unsigned int bx, const unsigned int by, double* tile /* LDS: FM_TILE * dim */) {
    const int dim = DIM > 0 ? DIM : dim_rt;
    const long long i = (long long)bx * blockDim.x + threadIdx.x;
    const bool live = i < na;
    double a[DIM > 0 ? DIM : 1];
    if (DIM > 0 && live) {
#pragma unroll
        for (int k = 0; k < DIM; ++k) a[k] = A[i * DIM + k];
    }
    double best = DBL_MAX;
    int bidx = -1;
    const long long tb = (long long)by * per, te = (tb + per < nb) ? tb + per : nb;
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
    if (live) { idx_out[(long long)by * na + i] = bidx; d2_out[(long long)by * na + i] = best; }
}
template <int DIM>
__global__ void __launch_bounds__(256) feature_match_kernel(const double* __restrict__ A, long long na, const double* __restrict__ B, long long nb, int dim_rt,
                                                             long long per, int* __restrict__ idx_out, double* __restrict__ d2_out) {
    extern __shared__ double tile[];  // FM_TILE * dim
    feature_match_body<DIM>(A, na, B, nb, dim_rt, per, idx_out, d2_out, blockIdx.x, blockIdx.y, tile);
}

__device__ static void feature_match_merge_body(const int* __restrict__ cidx, const double* __restrict__ cd2, long long na, int splits, int* __restrict__ idx_out,
                                                double* __restrict__ d2_out, const unsigned int bx) {
    const long long i = (long long)bx * blockDim.x + threadIdx.x;
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
__global__ void feature_match_merge_kernel(const int* __restrict__ cidx, const double* __restrict__ cd2, long long na, int splits,
                                           int* __restrict__ idx_out, double* __restrict__ d2_out) {
    feature_match_merge_body(cidx, cd2, na, splits, idx_out, d2_out, blockIdx.x);
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

// out: inl[h] (-1 = rejected by a checker), err2[h], T[h][12]
// 64 hypotheses per wave: every LANE draws its own sample and runs the checkers and the Kabsch step on it (the same scalar code the whole
// wave used to run 64 times over for one hypothesis); the few that pass are then scored one after the other by all 64 lanes together.
// Same arithmetic per hypothesis as ever: the batch's results do not depend on how hypotheses are dealt to waves.
__device__ static void ransac_eval(const ransac_args& a, const ransac_state* __restrict__ st, int* __restrict__ inl, double* __restrict__ err2,
                                   double* __restrict__ Tout, const int h_base) {
    if (h_base >= a.n_iter || st->stop) return;
    const int lane = threadIdx.x;
    const int h = h_base + lane;
    const int m = st->m;
    const unsigned long long itr = (unsigned long long)(a.first_iter + h);
    // (a hypothesis at or behind exit_itr is never looked at by the walk)
    const bool mine = h < a.n_iter && (long long)itr < st->exit_itr;
    bool ok = mine;
    double R[9], tr[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = 0.0;
    tr[0] = tr[1] = tr[2] = 0.0;
    if (mine) {
        double s[3][3], t[3][3];
        for (int j = 0; j < 3; ++j) {
            const unsigned int c = (unsigned int)(mix64(a.seed ^ mix64(itr * 3 + j)) % (unsigned long long)m);
            const pcr_pt ps = a.src[a.corr[2 * c]], pt = a.tgt[a.corr[2 * c + 1]];
            s[j][0] = ps.x; s[j][1] = ps.y; s[j][2] = ps.z;
            t[j][0] = pt.x; t[j][1] = pt.y; t[j][2] = pt.z;
        }
        if (a.edge_sim > 0) {
            for (int i = 0; i < 3 && ok; ++i)
                for (int j = i + 1; j < 3; ++j) {
                    const double ds = sqrt(((s[i][0] - s[j][0]) * (s[i][0] - s[j][0]) + (s[i][1] - s[j][1]) * (s[i][1] - s[j][1])) + (s[i][2] - s[j][2]) * (s[i][2] - s[j][2]));
                    const double dt = sqrt(((t[i][0] - t[j][0]) * (t[i][0] - t[j][0]) + (t[i][1] - t[j][1]) * (t[i][1] - t[j][1])) + (t[i][2] - t[j][2]) * (t[i][2] - t[j][2]));
                    if (ds < dt * a.edge_sim || dt < ds * a.edge_sim) { ok = false; break; }
                }
        }
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
        if (!ok) { inl[h] = -1; err2[h] = 0.0; }
    }
    // ---- the survivors, one after the other, scored by the whole wave
    unsigned long long mk = __ballot(ok);
    while (mk) {
        const int l = (int)__ffsll((long long)mk) - 1;
        mk &= mk - 1;
        double Rl[9], tl[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rl[k] = __shfl(R[k], l, 64);
#pragma unroll
        for (int k = 0; k < 3; ++k) tl[k] = __shfl(tr[k], l, 64);
        int good = 0;
        double e2 = 0.0;
        for (int c = lane; c < m; c += 64) {
            const pcr_pt ps = a.src[a.corr[2 * c]], pt = a.tgt[a.corr[2 * c + 1]];
            const double x = ((Rl[0] * ps.x + Rl[1] * ps.y) + Rl[2] * ps.z) + tl[0] - pt.x;
            const double y = ((Rl[3] * ps.x + Rl[4] * ps.y) + Rl[5] * ps.z) + tl[1] - pt.y;
            const double z = ((Rl[6] * ps.x + Rl[7] * ps.y) + Rl[8] * ps.z) + tl[2] - pt.z;
            const double dis = sqrt((x * x + y * y) + z * z);
            if (dis < a.max_dist) { ++good; e2 += dis * dis; }
        }
        e2 = wave_sum(e2);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) good += __shfl_xor(good, off, 64);
        if (lane == 0) {
            const int hl = h_base + l;
            inl[hl] = good;
            err2[hl] = e2;
            for (int k = 0; k < 9; ++k) Tout[12 * (long long)hl + k] = Rl[k];
            for (int k = 0; k < 3; ++k) Tout[12 * (long long)hl + 9 + k] = tl[k];
        }
    }
}

__global__ void __launch_bounds__(64) ransac_kernel(ransac_args a, const ransac_state* __restrict__ st, int* __restrict__ inl, double* __restrict__ err2,
                                                    double* __restrict__ Tout) {
    ransac_eval(a, st, inl, err2, Tout, 64 * (int)blockIdx.x);
}

// ONE wave: the sequential loop over the batch -- iteration order, running best (IsBetterRANSACThan: higher fitness, or equal
// fitness and lower rmse; the initial best is (0, 0)), exit_itr = min(exit_itr, ceil(log(1 - confidence) / log(1 - fitness^3)))
// after every improvement, stop at the first iteration >= exit_itr -- 64 iterations at a time: inside a chunk the NEXT improvement
// is the first lane that beats the current best (the order is a strict weak order, so nobody in front of it can beat the new best).
__device__ static void ransac_walk(const ransac_args& a, ransac_state* __restrict__ st, const int* __restrict__ inl, const double* __restrict__ err2,
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

__global__ void __launch_bounds__(64) ransac_walk_kernel(ransac_args a, ransac_state* __restrict__ st, const int* __restrict__ inl, const double* __restrict__ err2,
                                                         const double* __restrict__ Tout) {
    ransac_walk(a, st, inl, err2, Tout);
}

__device__ static void ransac_init(ransac_state* st, const int* m_p, int max_iteration) {
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

__global__ void ransac_init_kernel(ransac_state* st, const int* m_p, int max_iteration) { ransac_init(st, m_p, max_iteration); }

// correspondence set of registration_ransac_based_on_feature_matching: (i, ij[i]) for every source row, kept when mutual
// (ji[ij[i]] == i); when fewer than `min_mutual` survive, Open3D falls back to the one-way set.  ONE block, rows in order.
struct corr_lds { int w[4], total, use; };
__device__ static void corr_build_body(const int* __restrict__ ij, const int* __restrict__ ji, int na, int mutual, int min_mutual, int* __restrict__ corr,
                                       int* __restrict__ m_out, corr_lds* L) {
    int* const s_w = L->w;
    int& s_total = L->total;
    int& s_use = L->use;
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
__global__ void __launch_bounds__(256) corr_build_kernel(const int* __restrict__ ij, const int* __restrict__ ji, int na, int mutual, int min_mutual,
                                                         int* __restrict__ corr, int* __restrict__ m_out) {
    __shared__ corr_lds s_L;
    corr_build_body(ij, ji, na, mutual, min_mutual, corr, m_out, &s_L);
}

// ------------------------------------------------------------------------------------------------ every pair of a share at once
// The same stages with one launch for ALL pairs (a job = one pair; blockIdx.y / .z picks it): a pair alone is ten launches of a few
// microseconds of work each, and a share of the reference's pair loop (main.py:190-216) is hundreds of pairs.
struct init_job {
    const pcr_pt *src, *tgt;          // down-sampled records by row
    const double *fa, *fb;            // FPFH (n,33)
    int na, nb;
    int *ij, *ji;                     // nearest target row of every source row, and the other way
    double *dab, *dba;
    int *ci_ab, *ci_ba;               // per-split candidates (splits x n)
    double *cd_ab, *cd_ba;
    int *corr, *m;                    // (na,2) + count
    ransac_state* st;
    int* inl;                         // hypothesis scratch of a batch
    double *err2, *Tout;
    unsigned long long seed;
    // matrix-core operands of both descriptor sets (mfma_ops_kernel): as the sweep's targets, as its queries; squared norms; largest norm
    const double *ta, *tb, *qa, *qb, *n2a, *n2b, *mxa, *mxb;
    const unsigned int *mra, *mrb;    // row of the smallest norm (lowest row on ties) of either set: the answer for an all-zero query
};
constexpr int JOB_SPLITS = 8;
__device__ static inline long long job_per(long long nb) { return ((nb + JOB_SPLITS - 1) / JOB_SPLITS + FM_TILE - 1) / FM_TILE * FM_TILE; }

// ---- feature matching on the matrix cores, exact.
// |a - b|^2 = |a|^2 + (|b|^2 - 2 a.b): the bracket is a K = 36 contraction [-2 b_0 .. -2 b_32, |b|^2, 0, 0] . [a_0 .. a_32, 1, 0, 0] -- nine
// v_mfma_f64_16x16x4_f64 per 16 targets x 16 queries -- and |a|^2 does not move a query's argmin.  The sweep is only a FILTER: per query and
// lane the running minimum m of the bracket and the targets within tau of it are kept (the true winner is within tau of the running minimum
// when it is met: the minimum only falls); tau = 2^-40 (|a| + max |b|)^2 bounds twice the difference between the bracket + |a|^2 and the
// reference sum ((a_0 - b_0)^2 + ...) + ... of 33 rounded terms (< 80 roundings of quantities <= (|a| + |b|)^2 on either side).  The kept
// targets -- one, unless descriptors are (nearly) equidistant -- are then evaluated in the reference form, in index order: same index, same
// d^2, ties to the lowest row, as feature_match_body.  A lane that met more than two candidates evaluates its whole share directly.
// Operand layout (mfma_ops_kernel): ops[tile][step 0..8][lane] with lane l <-> (row = 16 tile + (l & 15), k = 4 step + (l >> 4)), the layout
// both the A operand (rows = targets) and the B operand (columns = queries) of the instruction use: one coalesced 512-byte read per step.
typedef double fm_v4 __attribute__((ext_vector_type(4)));
constexpr int FM_STEPS = 9;          // 36 = 33 dimensions + the norm / one slot + 2 zeros
constexpr int FM_NT = 4;             // query tiles of 16 per wave
constexpr double FM_TAU_REL = 9.094947017729282e-13;   // 2^-40
// Identical descriptors are common -- the points of a scan whose neighbourhoods hold a single other point all get the same one: groups of
// 50 - 70 rows in a 1 000-row scan -- and a query that is one of them ties with every target that is: dozens of exact evaluations behind the
// sweep.  A row that repeats an EARLIER row of its scan can never be the answer (same distance, higher index): it is taken out of the
// sweep's targets (its operand row becomes a padding row).  One block per scan: a hash table in LDS keeps the lowest row of every hash
// tag; a row whose tag's lowest row is an earlier one compares itself with that row bit by bit.
__global__ void __launch_bounds__(1024) dup_rows_kernel(const double* __restrict__ fpfh /* (ng,33) */, const unsigned int* __restrict__ scan_first,
                                                        unsigned char* __restrict__ dup /* (ng): 1 = repeats an earlier row of its scan */) {
    // open-addressing table in LDS: hash tag (high 32 bits) << 32 | lowest row seen with that tag; all ones = free
    constexpr unsigned int SLOTS = 2 * PCR_HYBRID_BRUTE_MAX;
    static_assert(SLOTS * 8 <= 65536 && (SLOTS & (SLOTS - 1)) == 0, "the table fits the block's LDS");
    __shared__ unsigned long long tab[SLOTS];
    const unsigned int base = scan_first[blockIdx.x], n = scan_first[blockIdx.x + 1] - base;   // (n <= PCR_HYBRID_BRUTE_MAX: checked by the caller)
    for (unsigned int i = threadIdx.x; i < SLOTS; i += 1024) tab[i] = ~0ull;
    __syncthreads();
    auto row_hash = [&](unsigned int j) {
        const unsigned long long* x = reinterpret_cast<const unsigned long long*>(fpfh + 33 * (size_t)(base + j));
        unsigned long long h = 0x9E3779B97F4A7C15ull;
        for (int k = 0; k < 33; ++k) { h ^= x[k]; h *= 0xD1B54A32D192ED03ull; h ^= h >> 29; }
        return h;
    };
    for (unsigned int j = threadIdx.x; j < n; j += 1024) {
        const unsigned long long h = row_hash(j), mine = (h & 0xffffffff00000000ull) | j;
        for (unsigned int slot = (unsigned int)h & (SLOTS - 1);; slot = (slot + 1) & (SLOTS - 1)) {
            const unsigned long long cur = tab[slot];
            if (cur == ~0ull) {
                if (atomicCAS(&tab[slot], ~0ull, mine) == ~0ull) break;
                --slot;   // somebody took it first: look at it again
                continue;
            }
            if ((cur >> 32) == (h >> 32)) { atomicMin(&tab[slot], mine); break; }   // same tag: the lowest row stays
        }
    }
    __syncthreads();
    for (unsigned int j = threadIdx.x; j < n; j += 1024) {
        const unsigned long long h = row_hash(j);
        unsigned int rep = j;
        for (unsigned int slot = (unsigned int)h & (SLOTS - 1);; slot = (slot + 1) & (SLOTS - 1)) {
            const unsigned long long cur = tab[slot];
            if (cur == ~0ull) break;
            if ((cur >> 32) == (h >> 32)) { rep = (unsigned int)cur; break; }
        }
        unsigned char d = 0;
        if (rep < j) {   // the same tag: the rows themselves decide (a colliding tag leaves the row a target: harmless)
            const unsigned long long* xi = reinterpret_cast<const unsigned long long*>(fpfh + 33 * (size_t)(base + rep));
            const unsigned long long* xj = reinterpret_cast<const unsigned long long*>(fpfh + 33 * (size_t)(base + j));
            bool same = true;
            for (int k = 0; k < 33; ++k) same = same && xi[k] == xj[k];
            d = same ? 1 : 0;
        }
        dup[base + j] = d;
    }
}

__global__ void __launch_bounds__(64) mfma_ops_kernel(const double* __restrict__ fpfh /* (ng,33) */, const unsigned int* __restrict__ scan_first, int n_scans,
                                                      const unsigned int* __restrict__ tile_first /* [n_scans + 1] */, const unsigned char* __restrict__ dup,
                                                      double* __restrict__ op_t, double* __restrict__ op_q,
                                                      double* __restrict__ norm2 /* (ng) */, unsigned long long* __restrict__ max_norm2 /* [n_scans], bits of a double */) {
    const unsigned int tile = blockIdx.x;
    const int lane = threadIdx.x;
    int lo = 0, hi = n_scans - 1;   // the scan that owns this tile
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tile_first[mid] <= tile) lo = mid;
        else hi = mid - 1;
    }
    const unsigned int base = scan_first[lo], n = scan_first[lo + 1] - base, lt = tile - tile_first[lo];
    const unsigned int row = 16u * lt + (unsigned int)(lane & 15);
    const bool live = row < n;
    const double* x = fpfh + 33 * (size_t)(base + row);
    double nn = 0.0;
    if (live)
        for (int k = 0; k < 33; ++k) nn += x[k] * x[k];
    if (live && lane < 16) {
        norm2[base + row] = nn;
        atomicMax(max_norm2 + lo, (unsigned long long)__double_as_longlong(nn));   // (non-negative doubles order like their bit patterns)
    }
#pragma unroll
    for (int st = 0; st < FM_STEPS; ++st) {
        const int k = 4 * st + (lane >> 4);
        double vt, vq;
        if (live) { vt = k < 33 ? -2.0 * x[k] : (k == 33 ? nn : 0.0); vq = k < 33 ? x[k] : (k == 33 ? 1.0 : 0.0); }
        else { vt = k == 33 ? 1e300 : 0.0; vq = 0.0; }   // a padding row never wins as a target, and is nobody's query
        if (live && dup[base + row]) vt = k == 33 ? 1e300 : 0.0;   // a repeated row: still a query, never a target
        op_t[((size_t)tile * FM_STEPS + st) * 64 + lane] = vt;
        op_q[((size_t)tile * FM_STEPS + st) * 64 + lane] = vq;
    }
}

// per scan: the row with the smallest squared norm, lowest row on ties (one block per scan)
__global__ void __launch_bounds__(256) min_norm_row_kernel(const double* __restrict__ norm2, const unsigned int* __restrict__ scan_first, unsigned int* __restrict__ min_row) {
    __shared__ double s_v[256];
    __shared__ unsigned int s_r[256];
    const unsigned int base = scan_first[blockIdx.x], n = scan_first[blockIdx.x + 1] - base;
    double v = DBL_MAX;
    unsigned int r = 0xffffffffu;
    for (unsigned int i = threadIdx.x; i < n; i += 256) {
        const double x = norm2[base + i];
        if (x < v) { v = x; r = i; }   // (ascending rows per thread: the first one met stays on ties)
    }
    s_v[threadIdx.x] = v; s_r[threadIdx.x] = r;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const double ov = s_v[threadIdx.x + off];
            const unsigned int orow = s_r[threadIdx.x + off];
            if (ov < s_v[threadIdx.x] || (ov == s_v[threadIdx.x] && orow < s_r[threadIdx.x])) { s_v[threadIdx.x] = ov; s_r[threadIdx.x] = orow; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) min_row[blockIdx.x] = s_r[0];
}

__device__ static inline double fm_exact(const double* __restrict__ a, const double* __restrict__ b) {   // feature_match_body's sum, term by term
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 33; ++k) { const double d = a[k] - b[k]; s += d * d; }
    return s;
}

__global__ void __launch_bounds__(256) feature_match_mfma_jobs_kernel(const init_job* __restrict__ jobs, int mutual, int splits /* target splits = gridDim.y */) {
    const init_job J = jobs[blockIdx.z >> 1];
    const bool back = (blockIdx.z & 1) != 0;
    if (back && !mutual) return;
    const int na = back ? J.nb : J.na, nb = back ? J.na : J.nb;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qt0 = ((int)blockIdx.x * 4 + wave) * FM_NT;   // this wave's first query tile
    if (qt0 * 16 >= na) return;
    const double* const A = back ? J.fb : J.fa;     // queries, reference layout
    const double* const B = back ? J.fa : J.fb;     // targets
    const double* const opq = back ? J.qb : J.qa;
    const double* const opt = back ? J.ta : J.tb;
    const double* const n2q = back ? J.n2b : J.n2a;
    const double* const n2t = back ? J.n2a : J.n2b;
    const unsigned int* const mrt = back ? J.mra : J.mrb;
    const double bmax = sqrt(*(back ? J.mxa : J.mxb));
    int* const idx_out = back ? J.ci_ba : J.ci_ab;
    double* const d2_out = back ? J.cd_ba : J.cd_ab;
    const int q_tiles = (na + 15) >> 4;
    double bq[FM_NT][FM_STEPS];
    double tau[FM_NT], m[FM_NT], hi[FM_NT];   // hi = m + tau, refreshed when m moves
    int c0[FM_NT], c1[FM_NT], cnt[FM_NT];
    bool zq[FM_NT];   // an all-zero query (the descriptor of an isolated point): its distance to target j is |b_j|^2 -- term for term the sum
                      // mfma_ops_kernel stored -- and it ties with every all-zero target: answered from min_norm_row_kernel's table
#pragma unroll
    for (int tt = 0; tt < FM_NT; ++tt) {
        const bool tile_ok = qt0 + tt < q_tiles;
#pragma unroll
        for (int st = 0; st < FM_STEPS; ++st) bq[tt][st] = tile_ok ? opq[((size_t)(qt0 + tt) * FM_STEPS + st) * 64 + lane] : 0.0;
        const int qi = (qt0 + tt) * 16 + (lane & 15);
        const double qn = qi < na ? sqrt(n2q[qi]) : 0.0;
        zq[tt] = splits == 1 && qi < na && qn == 0.0;
        tau[tt] = FM_TAU_REL * (qn + bmax) * (qn + bmax);
        m[tt] = DBL_MAX; hi[tt] = DBL_MAX; c0[tt] = c1[tt] = -1; cnt[tt] = 0;
    }
    // (one wave sweeps a whole split of the targets -- all of them by default: the exact evaluation behind the sweep is per (query, split))
    const long long per = ((nb + splits - 1) / splits + 15) / 16 * 16;
    const int tb = (int)((long long)blockIdx.y * per), te = (int)(tb + per < nb ? tb + per : nb);
    const int t_end = (te + 15) >> 4;
    double a_next[FM_STEPS];
#pragma unroll
    for (int st = 0; st < FM_STEPS; ++st) a_next[st] = (tb >> 4) < t_end ? opt[((size_t)(tb >> 4) * FM_STEPS + st) * 64 + lane] : 0.0;
    for (int t = tb >> 4; t < t_end; ++t) {
        double a[FM_STEPS];
#pragma unroll
        for (int st = 0; st < FM_STEPS; ++st) a[st] = a_next[st];
        if (t + 1 < t_end) {   // the next tile's operands are on their way while this one is multiplied
#pragma unroll
            for (int st = 0; st < FM_STEPS; ++st) a_next[st] = opt[((size_t)(t + 1) * FM_STEPS + st) * 64 + lane];
        }
        fm_v4 acc[FM_NT];
#pragma unroll
        for (int tt = 0; tt < FM_NT; ++tt) acc[tt] = fm_v4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int st = 0; st < FM_STEPS; ++st)
#pragma unroll
            for (int tt = 0; tt < FM_NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st], bq[tt][st], acc[tt], 0, 0, 0);
        const int j0 = t * 16 + (lane >> 4);   // this lane's four target rows of the tile: j0, j0 + 4, j0 + 8, j0 + 12 (the D layout of the instruction)
        // (almost every tile holds nothing near a lane's running minimum: one comparison of the four values' minimum against m + tau decides)
#pragma unroll
        for (int tt = 0; tt < FM_NT; ++tt) {
            const double vm = vmin(vmin(acc[tt][0], acc[tt][1]), vmin(acc[tt][2], acc[tt][3]));
            if (vm <= hi[tt]) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = acc[tt][r];
                    const int j = j0 + 4 * r;
                    if (v < m[tt] - tau[tt]) { m[tt] = v; c0[tt] = j; cnt[tt] = 1; }
                    else if (v <= m[tt] + tau[tt]) {
                        if (cnt[tt] == 1) c1[tt] = j;
                        cnt[tt] = cnt[tt] < 3 ? cnt[tt] + 1 : 3;
                        m[tt] = v < m[tt] ? v : m[tt];
                    }
                }
                hi[tt] = m[tt] + tau[tt];
            }
        }
    }
    // ---- the candidates in the reference form; the four lanes of a query meet
#pragma unroll
    for (int tt = 0; tt < FM_NT; ++tt) {
        const int qi = (qt0 + tt) * 16 + (lane & 15);
        double best = DBL_MAX;
        int bj = -1;
        // only a lane whose share's minimum is within tau of the query's minimum over all four shares can hold the winner (usually one of four)
        double gm = m[tt];
        gm = vmin(gm, __shfl_xor(gm, 16, 64));
        gm = vmin(gm, __shfl_xor(gm, 32, 64));
        const bool need = qi < na && !zq[tt] && cnt[tt] >= 1 && m[tt] <= gm + tau[tt];
        if (!need) cnt[tt] = 0;
        if (zq[tt] && (lane >> 4) == 0) { bj = (int)*mrt; best = n2t[bj]; }
        if (need) {
            const double* const aq = A + 33 * (size_t)qi;
            if (cnt[tt] <= 2) {
                if (cnt[tt] >= 1 && c0[tt] < te) { best = fm_exact(aq, B + 33 * (size_t)c0[tt]); bj = c0[tt]; }
                if (cnt[tt] == 2 && c1[tt] < te) {
                    const double d = fm_exact(aq, B + 33 * (size_t)c1[tt]);
                    if (d < best) { best = d; bj = c1[tt]; }
                }
            }   // (cnt = 3: more than two candidates -- the second sweep below)
        }
        // (Nearly) equidistant descriptors are not rare: an isolated point has an all-zero descriptor, and every scan has a few -- a query
        // that is one ties with all of the target's.  Such a lane knows its share's final minimum now: the wave multiplies once more and the
        // lane evaluates, in index order, exactly the rows within tau of it (its whole share directly was 132 evaluations of 66 loads).
        if (__any(qi < na && cnt[tt] == 3)) {
            const bool mine = qi < na && cnt[tt] == 3;
            const double* const aq = A + 33 * (size_t)(qi < na ? qi : 0);
            double a2n[FM_STEPS];   // (operands one tile ahead, as in the first sweep: nine dependent reads per tile were 0.7 ms for ONE such wave)
#pragma unroll
            for (int st = 0; st < FM_STEPS; ++st) a2n[st] = (tb >> 4) < t_end ? opt[((size_t)(tb >> 4) * FM_STEPS + st) * 64 + lane] : 0.0;
            for (int t = tb >> 4; t < t_end; ++t) {
                double a2[FM_STEPS];
#pragma unroll
                for (int st = 0; st < FM_STEPS; ++st) a2[st] = a2n[st];
                if (t + 1 < t_end) {
#pragma unroll
                    for (int st = 0; st < FM_STEPS; ++st) a2n[st] = opt[((size_t)(t + 1) * FM_STEPS + st) * 64 + lane];
                }
                fm_v4 acc2 = fm_v4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int st = 0; st < FM_STEPS; ++st) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[st], bq[tt][st], acc2, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = t * 16 + (lane >> 4) + 4 * r;
                    if (mine && j < te && acc2[r] <= m[tt] + tau[tt]) {
                        const double d = fm_exact(aq, B + 33 * (size_t)j);
                        if (d < best) { best = d; bj = j; }
                    }
                }
            }
        }
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            const double od = __shfl_xor(best, off, 64);
            const int oj = __shfl_xor(bj, off, 64);
            if (oj >= 0 && (bj < 0 || od < best || (od == best && oj < bj))) { best = od; bj = oj; }
        }
        if (qi < na && (lane >> 4) == 0) { idx_out[(long long)blockIdx.y * na + qi] = bj; d2_out[(long long)blockIdx.y * na + qi] = best; }
    }
}

__global__ void __launch_bounds__(256) feature_match_jobs_kernel(const init_job* __restrict__ jobs, int mutual) {
    __shared__ double tile[FM_TILE * 33];
    const init_job J = jobs[blockIdx.z >> 1];
    const bool back = (blockIdx.z & 1) != 0;
    if (back && !mutual) return;
    const long long na = back ? J.nb : J.na, nb = back ? J.na : J.nb;
    if ((long long)blockIdx.x * 256 >= na) return;   // (the whole block)
    feature_match_body<33>(back ? J.fb : J.fa, na, back ? J.fa : J.fb, nb, 33, job_per(nb), back ? J.ci_ba : J.ci_ab, back ? J.cd_ba : J.cd_ab, blockIdx.x, blockIdx.y, tile);
}
__global__ void __launch_bounds__(256) feature_match_merge_jobs_kernel(const init_job* __restrict__ jobs, int mutual, int splits) {
    const init_job J = jobs[blockIdx.y >> 1];
    const bool back = (blockIdx.y & 1) != 0;
    if (back && !mutual) return;
    const long long na = back ? J.nb : J.na;
    feature_match_merge_body(back ? J.ci_ba : J.ci_ab, back ? J.cd_ba : J.cd_ab, na, splits, back ? J.ji : J.ij, back ? J.dba : J.dab, blockIdx.x);
}
__global__ void __launch_bounds__(256) corr_build_jobs_kernel(const init_job* __restrict__ jobs, int mutual, int min_mutual, int max_iteration) {
    __shared__ corr_lds s_L;
    const init_job J = jobs[blockIdx.x];
    corr_build_body(J.ij, J.ji, J.na, mutual, min_mutual, J.corr, J.m, &s_L);
    __syncthreads();   // (the count is written by thread 0, which also initialises the loop state)
    ransac_init(J.st, J.m, max_iteration);
}
struct ransac_common { double edge_sim, max_dist, confidence; int check_distance, max_iteration, first_iter, n_iter; };
__device__ static inline ransac_args job_args(const init_job& J, const ransac_common& c) {
    ransac_args a;
    a.src = J.src; a.tgt = J.tgt; a.corr = J.corr;
    a.first_iter = c.first_iter; a.n_iter = c.n_iter;
    a.seed = J.seed; a.edge_sim = c.edge_sim; a.max_dist = c.max_dist; a.check_distance = c.check_distance;
    a.max_iteration = c.max_iteration; a.confidence = c.confidence;
    return a;
}
// active: the jobs still running (indices into jobs), or null = all
__global__ void __launch_bounds__(64) ransac_jobs_kernel(const init_job* __restrict__ jobs, const int* __restrict__ active, ransac_common c) {
    const init_job J = jobs[active ? active[blockIdx.y] : blockIdx.y];
    const ransac_args a = job_args(J, c);
    ransac_eval(a, J.st, J.inl, J.err2, J.Tout, 64 * (int)blockIdx.x);
}
__global__ void __launch_bounds__(64) ransac_walk_jobs_kernel(const init_job* __restrict__ jobs, const int* __restrict__ active, ransac_common c) {
    const init_job J = jobs[active ? active[blockIdx.x] : blockIdx.x];
    const ransac_args a = job_args(J, c);
    ransac_walk(a, J.st, J.inl, J.err2, J.Tout);
}

// ------------------------------------------------------------------ host side
namespace {
using dev_buf = pcr_dev_block;

int* fail_word(pcr_ctx* ctx) { return (int*)(ctx->d_counters + 116); }

// A cloud of a few thousand points (what the 2 m down-sample of main.py:35 leaves of a scan: 300 - 1 500 points) is searched without
// an index: two grid builds per scan -- one per radius, ~20 launches each -- cost several times what the neighbourhoods themselves
// cost, and a wave reads 4 096 records in 64 trips.  The "view" of such a cloud: its records in row order, levels = 0.
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
            hipLaunchKernelGGL(ransac_kernel, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, a, (const ransac_state*)st, dinl.as<int>(), derr.as<double>(), dT.as<double>());
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

}  // extern "C"

// ------------------------------------------------------------------------------------------------ the pair loop's initialisation, fused
// prepare_dataset + execute_global_registration (Registration/main.py:197-203) for a whole share of pairs on ONE context: the scans
// are packed into pinned memory by a few host threads and copied once; ONE sort down-samples all of them (pcr_voxel_downsample_scans);
// normals, SPFH and FPFH are one launch each over every down-sampled point of the chunk (a block's search space is its own scan); matching,
// correspondence sets and the RANSAC loop run for all pairs side by side.  Same arithmetic, same order, same seeds as pcr_preprocess +
// pcr_global_registration pair by pair: the results are bit for bit those (tests/test_gpu_global_init.py).
// Returns PCR_E_UNSUPPORTED when the share does not fit this path (a down-sampled scan above PCR_HYBRID_BRUTE_MAX points, a neighbourhood
// that cannot be bounded, extents too large for the packed key): the caller then takes the scans one by one.
namespace {
struct scan_chunk {
    pcr_pt* down = nullptr;
    double* fpfh = nullptr;
    unsigned int* vsid = nullptr;
    unsigned int* scan_first = nullptr;
    int64_t ng = 0;
    int n_scans = 0;
    std::vector<unsigned int> first;   // host copy of scan_first
    // matrix-core operands of the descriptors (feature_match_mfma_jobs_kernel): per scan ceil(n / 16) tiles from tile_first[scan]
    double *op_t = nullptr, *op_q = nullptr, *norm2 = nullptr;
    unsigned long long* max_norm2 = nullptr;
    unsigned int* min_row = nullptr;
    std::vector<unsigned int> tile_first;
    size_t tiles = 0;
};
struct scan_slot { int chunk = -1, local = 0; };

template <typename F>
void parallel_for(int64_t n, int threads, F&& fn) {
    if (threads > (int)n) threads = (int)n;
    if (threads <= 1) { for (int64_t i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int64_t> next(0);
    auto worker = [&]() { for (;;) { const int64_t i = next.fetch_add(1); if (i >= n) break; fn(i); } };
    std::vector<std::thread> pool;
    try { for (int t = 1; t < threads; ++t) pool.emplace_back(worker); } catch (...) {}   // (a refused thread: the others do its share)
    worker();
    for (auto& th : pool) th.join();
}
}  // namespace

int pcr_global_init_batch(pcr_ctx* ctx, const pcr_cloud_ref* clouds, int64_t n_clouds, const int64_t* scans, int64_t n_scans, const pcr_pair_ref* pairs,
                          const int64_t* todo, int64_t n_todo, const pcr_global_params* g, double* T_init, int host_threads) {
    if (!ctx || !clouds || !scans || !pairs || !todo || !g || !T_init || n_scans < 1 || n_todo < 1) return PCR_E_INVALID;
    if (!(g->voxel_size > 0) || !(g->normal_radius > 0) || !(g->fpfh_radius > 0) || g->normal_max_nn < 1 || g->normal_max_nn > NB_CAP || g->fpfh_max_nn < 2 ||
        g->fpfh_max_nn > NB_CAP || g->ransac.max_iteration < 1 || !(g->ransac.max_distance > 0))
        return PCR_E_INVALID;
    if (getenv("PCR_INIT_PER_SCAN") != nullptr) return PCR_E_UNSUPPORTED;   // A/B and tests: the scans one by one (read per call)
    hipSetDevice(ctx->device);
    static const bool timing = getenv("PCR_INIT_TIMING") != nullptr;   // diagnostics: milliseconds per stage to stderr (synchronises after every stage)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        hipStreamSynchronize(ctx->stream);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "pcr_global_init_batch: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    const bool use_mfma = getenv("PCR_INIT_MATCH_VALU") == nullptr;   // A/B and tests: the matching of the pair stage on the vector ALUs (read per call)
    constexpr int64_t CHUNK_PTS = 16ll << 20;
    constexpr int CHUNK_SCANS = 2048;
    std::vector<scan_slot> slot((size_t)n_clouds);
    std::vector<scan_chunk> chunks;
    int rc = PCR_OK;
    auto release = [&]() {
        for (auto& c : chunks) {
            if (c.down) pcr_dev_free(ctx, c.down, sizeof(pcr_pt) * (size_t)(c.ng > 0 ? c.ng : 1));
            if (c.vsid) pcr_dev_free(ctx, c.vsid, 4 * (size_t)(c.ng > 0 ? c.ng : 1));
            if (c.scan_first) pcr_dev_free(ctx, c.scan_first, 4 * (size_t)(c.n_scans + 1));
            if (c.fpfh) pcr_dev_free(ctx, c.fpfh, sizeof(double) * 33 * (size_t)(c.ng > 0 ? c.ng : 1));
            if (c.op_t) pcr_dev_free(ctx, c.op_t, 8 * 64 * (size_t)FM_STEPS * (c.tiles ? c.tiles : 1));
            if (c.op_q) pcr_dev_free(ctx, c.op_q, 8 * 64 * (size_t)FM_STEPS * (c.tiles ? c.tiles : 1));
            if (c.norm2) pcr_dev_free(ctx, c.norm2, 8 * (size_t)(c.ng > 0 ? c.ng : 1));
            if (c.max_norm2) pcr_dev_free(ctx, c.max_norm2, 8 * (size_t)(c.n_scans + 1));
            if (c.min_row) pcr_dev_free(ctx, c.min_row, 4 * (size_t)(c.n_scans + 1));
        }
        chunks.clear();
    };
    // ---------------------------------------------------------------- the scans, chunk by chunk
    // (a share of more than ~2 M points is cut into four or more chunks so that the host threads pack chunk k + 1 into the other half of
    // the pinned block while the device works on chunk k)
    struct chunk_plan { std::vector<pcr_down_scan> ds; std::vector<int64_t> who; unsigned int at = 0; };
    std::vector<chunk_plan> plans;
    {
        int64_t total = 0;
        for (int64_t s = 0; s < n_scans; ++s) {
            const pcr_cloud_ref& C = clouds[scans[s]];
            if (C.n < 0 || C.n > 0x7fffffffll || C.stride < 3 || (C.n > 0 && !C.xyz)) return PCR_E_INVALID;
            total += C.n;
        }
        int64_t chunk_pts = total / 4;
        if (chunk_pts < (2ll << 20)) chunk_pts = 2ll << 20;
        if (chunk_pts > CHUNK_PTS) chunk_pts = CHUNK_PTS;
        chunk_plan cur;
        int in_chunk = 0;
        for (int64_t s = 0; s < n_scans; ++s) {
            const int64_t n = clouds[scans[s]].n;
            if (in_chunk > 0 && (in_chunk >= CHUNK_SCANS || (int64_t)cur.at + n > chunk_pts)) {
                if (!cur.ds.empty()) plans.push_back(std::move(cur));
                cur = chunk_plan();
                in_chunk = 0;
            }
            ++in_chunk;
            if (n == 0) continue;   // (an empty scan stays without a slot: its pairs keep the identity, as when pcr_cloud_upload_f32 says PCR_E_EMPTY)
            pcr_down_scan d;
            d.first_pt = cur.at; d.n_pts = (unsigned int)n;
            cur.at += (unsigned int)n;
            cur.ds.push_back(d);
            cur.who.push_back(scans[s]);
        }
        if (!cur.ds.empty()) plans.push_back(std::move(cur));
    }
    size_t half = 0;
    for (auto& P : plans) half = 12 * (size_t)P.at > half ? 12 * (size_t)P.at : half;
    half = (half + 4095) & ~(size_t)4095;
    if (!plans.empty() && ctx->h_init_bytes < 2 * half) {
        if (ctx->h_init) hipHostFree(ctx->h_init);
        ctx->h_init = nullptr; ctx->h_init_bytes = 0;
        if (hipHostMalloc(&ctx->h_init, 2 * half, hipHostMallocDefault) != hipSuccess) {   // no pinned block of that size: the scans go one by one
            (void)hipGetLastError();
            ctx->h_init = nullptr;
            return PCR_E_UNSUPPORTED;
        }
        ctx->h_init_bytes = 2 * half;
    }
    auto pack = [&](size_t k) {
        chunk_plan& P = plans[k];
        float* const h_xyz = (float*)((char*)ctx->h_init + (k & 1) * half);
        parallel_for((int64_t)P.ds.size(), host_threads, [&](int64_t q) {
            const pcr_cloud_ref& C = clouds[P.who[(size_t)q]];
            float* const dst = h_xyz + 3 * (size_t)P.ds[(size_t)q].first_pt;
            const float* const src = C.xyz;
            const size_t st = (size_t)C.stride;
            float lo[12], hi[12];   // four points a trip: twelve independent minima / maxima
            for (int j = 0; j < 12; ++j) lo[j] = hi[j] = src[j % 3];
            int64_t i = 0;
            for (; i + 4 <= C.n; i += 4) {
                float v[12];
                for (int u = 0; u < 4; ++u)
                    for (int d = 0; d < 3; ++d) v[3 * u + d] = src[(size_t)(i + u) * st + d];
                for (int j = 0; j < 12; ++j) {
                    dst[3 * i + j] = v[j];
                    lo[j] = v[j] < lo[j] ? v[j] : lo[j];
                    hi[j] = v[j] > hi[j] ? v[j] : hi[j];
                }
            }
            for (; i < C.n; ++i)
                for (int d = 0; d < 3; ++d) {
                    const float v = src[(size_t)i * st + d];
                    dst[3 * i + d] = v;
                    lo[d] = v < lo[d] ? v : lo[d];
                    hi[d] = v > hi[d] ? v : hi[d];
                }
            for (int d = 0; d < 3; ++d) {
                float a = lo[d], b = hi[d];
                for (int u = 1; u < 4; ++u) { a = lo[3 * u + d] < a ? lo[3 * u + d] : a; b = hi[3 * u + d] > b ? hi[3 * u + d] : b; }
                P.ds[(size_t)q].mn[d] = (double)a; P.ds[(size_t)q].mx[d] = (double)b;
            }
        });
    };
    if (!plans.empty()) pack(0);
    for (size_t k = 0; k < plans.size() && rc == PCR_OK; ++k) {
        chunk_plan& P = plans[k];
        std::vector<pcr_down_scan>& ds = P.ds;
        std::vector<int64_t>& who = P.who;
        const unsigned int at = P.at;
        const size_t bytes = 12 * (size_t)at;
        float* const h_xyz = (float*)((char*)ctx->h_init + (k & 1) * half);
        // the next chunk is packed (other half of the block) while the device works on this one
        std::thread packer;
        struct joiner { std::thread& t; ~joiner() { if (t.joinable()) t.join(); } } join_packer{packer};
        if (k + 1 < plans.size()) {
            try { packer = std::thread(pack, k + 1); } catch (...) { pack(k + 1); }
        }
        lap("pack (host threads)");
        pcr_dev_block b_xyz(ctx);
        if ((rc = b_xyz.alloc(bytes))) break;
        if (hipMemcpyAsync(b_xyz.p, h_xyz, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = PCR_E_HIP; break; }
        lap("copy to the device");
        chunks.emplace_back();
        scan_chunk& c = chunks.back();
        c.n_scans = (int)ds.size();
        c.first.assign(ds.size() + 1, 0u);
        rc = pcr_voxel_downsample_scans(ctx, (const float*)b_xyz.p, (int64_t)at, ds.data(), c.n_scans, g->voxel_size, &c.down, &c.vsid, &c.scan_first, c.first.data(), &c.ng);   // (synchronises)
        if (rc) break;
        lap("down-sample (all scans)");
        for (int k = 0; k < c.n_scans; ++k) {
            if (c.first[(size_t)k + 1] - c.first[(size_t)k] > (unsigned int)PCR_HYBRID_BRUTE_MAX) { rc = PCR_E_UNSUPPORTED; break; }
            slot[(size_t)who[(size_t)k]].chunk = (int)chunks.size() - 1;
            slot[(size_t)who[(size_t)k]].local = k;
        }
        if (rc) break;
        const size_t ng = (size_t)c.ng;
        pcr_dev_block b_nrm(ctx), b_spfh(ctx), b_id(ctx), b_d2(ctx), b_cnt(ctx);
        if ((rc = pcr_dev_alloc(ctx, sizeof(double) * 33 * ng, (void**)&c.fpfh)) || (rc = b_nrm.alloc(sizeof(double) * 3 * ng)) || (rc = b_spfh.alloc(sizeof(double) * 33 * ng)) ||
            (rc = b_id.alloc(sizeof(unsigned int) * (size_t)g->fpfh_max_nn * ng)) || (rc = b_d2.alloc(sizeof(double) * (size_t)g->fpfh_max_nn * ng)) || (rc = b_cnt.alloc(sizeof(int) * ng)))
            break;
        const scans_view V{c.down, c.vsid, c.scan_first};
        pcr_dev_block b_redo(ctx), b_cov(ctx);
        if ((rc = b_redo.alloc(4 * ng)) || (rc = b_cov.alloc(sizeof(double) * 7 * ng))) break;
        unsigned int* const redo = b_redo.as<unsigned int>();
        unsigned int* const redo_n = ctx->d_counters + 117;   // [0]: normals, [1]: SPFH (zero between calls)
        const unsigned fixed = (unsigned)(ng < (size_t)(16 * ctx->cu_count) ? ng : (size_t)(16 * ctx->cu_count));
        const unsigned int* const none = nullptr;
        if (hipMemsetAsync(redo_n, 0, 8, ctx->stream) != hipSuccess) { rc = PCR_E_HIP; break; }
        hipLaunchKernelGGL(normals_scans_kernel<128>, dim3((unsigned)ng), dim3(64), 0, ctx->stream, V, (long long)ng, g->normal_radius * g->normal_radius, g->normal_max_nn,
                           b_nrm.as<double>(), fail_word(ctx), none, none, redo, redo_n, b_cov.as<double>());
        hipLaunchKernelGGL(normals_scans_kernel<NB_CAP>, dim3(fixed), dim3(64), 0, ctx->stream, V, (long long)ng, g->normal_radius * g->normal_radius, g->normal_max_nn,
                           b_nrm.as<double>(), fail_word(ctx), (const unsigned int*)redo, (const unsigned int*)redo_n, (unsigned int*)nullptr, (unsigned int*)nullptr, b_cov.as<double>());
        hipLaunchKernelGGL(normals_finish_kernel, dim3((unsigned)((ng + 255) / 256)), dim3(256), 0, ctx->stream, V, (long long)ng, (const double*)b_cov.as<double>(), b_nrm.as<double>());
        // (the SPFH list is written behind the normals' one: both launches of a stage are done before the next stage's first)
        hipLaunchKernelGGL(spfh_scans_kernel<256>, dim3((unsigned)ng), dim3(64), 0, ctx->stream, V, (long long)ng, g->fpfh_radius * g->fpfh_radius, g->fpfh_max_nn,
                           (const double*)b_nrm.as<double>(), b_spfh.as<double>(), b_id.as<unsigned int>(), b_d2.as<double>(), b_cnt.as<int>(), fail_word(ctx), none, none, redo,
                           redo_n + 1);
        hipLaunchKernelGGL(spfh_scans_kernel<NB_CAP>, dim3(fixed), dim3(64), 0, ctx->stream, V, (long long)ng, g->fpfh_radius * g->fpfh_radius, g->fpfh_max_nn,
                           (const double*)b_nrm.as<double>(), b_spfh.as<double>(), b_id.as<unsigned int>(), b_d2.as<double>(), b_cnt.as<int>(), fail_word(ctx),
                           (const unsigned int*)redo, (const unsigned int*)(redo_n + 1), (unsigned int*)nullptr, (unsigned int*)nullptr);
        if (g->fpfh_max_nn <= 128)
            hipLaunchKernelGGL(fpfh_scans_kernel<128>, dim3((unsigned)ng), dim3(64), 0, ctx->stream, V, (long long)ng, g->fpfh_max_nn, (const double*)b_spfh.as<double>(),
                               (const unsigned int*)b_id.as<unsigned int>(), (const double*)b_d2.as<double>(), (const int*)b_cnt.as<int>(), c.fpfh);
        else
            hipLaunchKernelGGL(fpfh_scans_kernel<NB_CAP>, dim3((unsigned)ng), dim3(64), 0, ctx->stream, V, (long long)ng, g->fpfh_max_nn, (const double*)b_spfh.as<double>(),
                               (const unsigned int*)b_id.as<unsigned int>(), (const double*)b_d2.as<double>(), (const int*)b_cnt.as<int>(), c.fpfh);
        if (use_mfma) {
            c.tile_first.assign((size_t)c.n_scans + 1, 0u);
            for (int k = 0; k < c.n_scans; ++k) c.tile_first[(size_t)k + 1] = c.tile_first[(size_t)k] + (c.first[(size_t)k + 1] - c.first[(size_t)k] + 15u) / 16u;
            c.tiles = c.tile_first[(size_t)c.n_scans];
            pcr_dev_block b_tf(ctx), b_dup(ctx);
            if ((rc = b_dup.alloc(ng ? ng : 1))) break;
            const size_t op_bytes = 8 * 64 * (size_t)FM_STEPS * (c.tiles ? c.tiles : 1);
            if ((rc = pcr_dev_alloc(ctx, op_bytes, (void**)&c.op_t)) || (rc = pcr_dev_alloc(ctx, op_bytes, (void**)&c.op_q)) || (rc = pcr_dev_alloc(ctx, 8 * (ng ? ng : 1), (void**)&c.norm2)) ||
                (rc = pcr_dev_alloc(ctx, 8 * (size_t)(c.n_scans + 1), (void**)&c.max_norm2)) || (rc = pcr_dev_alloc(ctx, 4 * (size_t)(c.n_scans + 1), (void**)&c.min_row)) ||
                (rc = b_tf.alloc(4 * (size_t)(c.n_scans + 1))))
                break;
            if (hipMemcpyAsync(b_tf.p, c.tile_first.data(), 4 * (size_t)(c.n_scans + 1), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipMemsetAsync(c.max_norm2, 0, 8 * (size_t)(c.n_scans + 1), ctx->stream) != hipSuccess) { rc = PCR_E_HIP; break; }
            hipLaunchKernelGGL(dup_rows_kernel, dim3((unsigned)c.n_scans), dim3(1024), 0, ctx->stream, (const double*)c.fpfh, (const unsigned int*)c.scan_first, b_dup.as<unsigned char>());
            if (c.tiles)
                hipLaunchKernelGGL(mfma_ops_kernel, dim3((unsigned)c.tiles), dim3(64), 0, ctx->stream, (const double*)c.fpfh, (const unsigned int*)c.scan_first, c.n_scans,
                                   (const unsigned int*)b_tf.as<unsigned int>(), (const unsigned char*)b_dup.as<unsigned char>(), c.op_t, c.op_q, c.norm2, c.max_norm2);
            hipLaunchKernelGGL(min_norm_row_kernel, dim3((unsigned)c.n_scans), dim3(256), 0, ctx->stream, (const double*)c.norm2, (const unsigned int*)c.scan_first, c.min_row);
            if (hipGetLastError() != hipSuccess) { rc = PCR_E_HIP; break; }
            rc = read_fail(ctx);   // (synchronises: the chunk's scratch -- and the tile table -- and the pinned block are free for the next chunk)
            lap("normals + SPFH + FPFH");
            continue;
        }
        if (hipGetLastError() != hipSuccess) { rc = PCR_E_HIP; break; }
        rc = read_fail(ctx);   // (synchronises: the chunk's scratch and the pinned block are free for the next chunk)
        lap("normals + SPFH + FPFH");
    }
    if (rc) { pcr_sync(ctx->stream); release(); return rc; }
    // ---------------------------------------------------------------- the pairs
    const int mutual = g->mutual_filter ? 1 : 0;
    constexpr int PAIR_CHUNK = 512, FIRST = 4096, BATCH = 16384;
    for (int64_t p0 = 0; p0 < n_todo && rc == PCR_OK; p0 += PAIR_CHUNK) {
        const int64_t p1 = p0 + PAIR_CHUNK < n_todo ? p0 + PAIR_CHUNK : n_todo;
        std::vector<init_job> jobs;
        std::vector<int64_t> job_pair;
        size_t n_i = 0, n_d = 0;   // ints / doubles of the pool
        int max_n = 1;
        for (int64_t t = p0; t < p1; ++t) {
            const pcr_pair_ref& P = pairs[todo[t]];
            const scan_slot &S = slot[(size_t)P.src], &T = slot[(size_t)P.tgt];
            if (S.chunk < 0 || T.chunk < 0) continue;   // an empty scan: identity
            const scan_chunk &cs = chunks[(size_t)S.chunk], &ct = chunks[(size_t)T.chunk];
            init_job J;
            memset(&J, 0, sizeof(J));
            const unsigned int fs = cs.first[(size_t)S.local], ft = ct.first[(size_t)T.local];
            J.na = (int)(cs.first[(size_t)S.local + 1] - fs);
            J.nb = (int)(ct.first[(size_t)T.local + 1] - ft);
            if (J.na <= 0 || J.nb <= 0) continue;
            J.src = cs.down + fs; J.tgt = ct.down + ft;
            J.fa = cs.fpfh + 33 * (size_t)fs; J.fb = ct.fpfh + 33 * (size_t)ft;
            if (use_mfma) {
                const size_t ts = (size_t)cs.tile_first[(size_t)S.local] * FM_STEPS * 64, tt = (size_t)ct.tile_first[(size_t)T.local] * FM_STEPS * 64;
                J.ta = cs.op_t + ts; J.qa = cs.op_q + ts; J.n2a = cs.norm2 + fs; J.mxa = (const double*)(cs.max_norm2 + S.local); J.mra = cs.min_row + S.local;
                J.tb = ct.op_t + tt; J.qb = ct.op_q + tt; J.n2b = ct.norm2 + ft; J.mxb = (const double*)(ct.max_norm2 + T.local); J.mrb = ct.min_row + T.local;
            }
            J.seed = g->ransac.seed;
            // pool offsets (resolved below): ij na | ji nb | ci_ab S*na | ci_ba S*nb | corr 2 na + 4 | inl BATCH   (ints)
            //                                dab na | dba nb | cd_ab S*na | cd_ba S*nb | err2 BATCH | Tout 12 BATCH | state   (doubles)
            n_i += (size_t)(1 + JOB_SPLITS + 2) * J.na + (size_t)(1 + JOB_SPLITS) * J.nb + 4 + BATCH;
            n_d += (size_t)(1 + JOB_SPLITS) * (J.na + J.nb) + 13 * (size_t)BATCH;
            if (J.na > max_n) max_n = J.na;
            if (J.nb > max_n) max_n = J.nb;
            jobs.push_back(J);
            job_pair.push_back(todo[t]);
        }
        const int nj = (int)jobs.size();
        if (nj == 0) continue;
        pcr_dev_block b_i(ctx), b_d(ctx), b_jobs(ctx), b_act(ctx), b_st(ctx);
        if ((rc = b_i.alloc(4 * n_i)) || (rc = b_d.alloc(8 * n_d)) || (rc = b_jobs.alloc(sizeof(init_job) * nj)) || (rc = b_act.alloc(4 * (size_t)nj)) ||
            (rc = b_st.alloc(sizeof(ransac_state) * (size_t)nj)))
            break;
        {
            int* pi = b_i.as<int>();
            double* pd = b_d.as<double>();
            ransac_state* ps = b_st.as<ransac_state>();
            for (auto& J : jobs) {
                J.st = ps++;
                J.ij = pi; pi += J.na;
                J.ji = pi; pi += J.nb;
                J.ci_ab = pi; pi += (size_t)JOB_SPLITS * J.na;
                J.ci_ba = pi; pi += (size_t)JOB_SPLITS * J.nb;
                J.corr = pi; J.m = pi + 2 * (size_t)J.na; pi += 2 * (size_t)J.na + 4;
                J.inl = pi; pi += BATCH;
                J.dab = pd; pd += J.na;
                J.dba = pd; pd += J.nb;
                J.cd_ab = pd; pd += (size_t)JOB_SPLITS * J.na;
                J.cd_ba = pd; pd += (size_t)JOB_SPLITS * J.nb;
                J.err2 = pd; pd += BATCH;
                J.Tout = pd; pd += 12 * (size_t)BATCH;
            }
        }
        if (hipMemcpyAsync(b_jobs.p, jobs.data(), sizeof(init_job) * nj, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = PCR_E_HIP; break; }
        const init_job* d_jobs = b_jobs.as<const init_job>();
        const unsigned qb = (unsigned)((max_n + 255) / 256);
        // (matrix-core sweep: one split -- 4 waves per 256 queries and direction -- unless a pair or two are all there is)
        const int mfma_splits = (long long)nj * 2 * qb * 4 >= (long long)ctx->cu_count ? 1 : (JOB_SPLITS < 4 ? JOB_SPLITS : 4);
        if (use_mfma) hipLaunchKernelGGL(feature_match_mfma_jobs_kernel, dim3(qb, mfma_splits, 2 * nj), dim3(256), 0, ctx->stream, d_jobs, mutual, mfma_splits);
        else hipLaunchKernelGGL(feature_match_jobs_kernel, dim3(qb, JOB_SPLITS, 2 * nj), dim3(256), 0, ctx->stream, d_jobs, mutual);
        hipLaunchKernelGGL(feature_match_merge_jobs_kernel, dim3(qb, 2 * nj), dim3(256), 0, ctx->stream, d_jobs, mutual, use_mfma ? mfma_splits : JOB_SPLITS);
        hipLaunchKernelGGL(corr_build_jobs_kernel, dim3(nj), dim3(256), 0, ctx->stream, d_jobs, mutual, 9, g->ransac.max_iteration);
        lap("matching + correspondences");
        ransac_common rcmn;
        rcmn.edge_sim = g->ransac.edge_similarity; rcmn.max_dist = g->ransac.max_distance; rcmn.confidence = g->ransac.confidence;
        rcmn.check_distance = g->ransac.check_distance; rcmn.max_iteration = g->ransac.max_iteration;
        std::vector<ransac_state> st((size_t)nj);
        std::vector<int> active;
        long long done = 0;
        // the first 4 096 iterations for every pair (most registrations exit within the first thousand), then 16 384 at a time for
        // the pairs that are still running
        for (int round = 0; done < g->ransac.max_iteration && rc == PCR_OK; ++round) {
            const long long want = round == 0 ? FIRST : BATCH;
            const int nb = (int)((g->ransac.max_iteration - done) < want ? (g->ransac.max_iteration - done) : want);
            rcmn.first_iter = (int)done; rcmn.n_iter = nb;
            const int n_run = round == 0 ? nj : (int)active.size();
            const int* d_act = round == 0 ? nullptr : b_act.as<const int>();
            if (round > 0 && hipMemcpyAsync(b_act.p, active.data(), 4 * (size_t)n_run, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { rc = PCR_E_HIP; break; }
            hipLaunchKernelGGL(ransac_jobs_kernel, dim3((nb + 63) / 64, n_run), dim3(64), 0, ctx->stream, d_jobs, d_act, rcmn);
            hipLaunchKernelGGL(ransac_walk_jobs_kernel, dim3(n_run), dim3(64), 0, ctx->stream, d_jobs, d_act, rcmn);
            if (hipGetLastError() != hipSuccess) { rc = PCR_E_HIP; break; }
            done += nb;
            if ((rc = pcr_d2h_staged(ctx, st.data(), b_st.p, sizeof(ransac_state) * (size_t)nj))) break;   // every job's state, one read (synchronises)
            std::vector<int> next;
            for (int j = 0; j < nj; ++j)
                if (!st[(size_t)j].stop) next.push_back(j);
            active.swap(next);
            lap("RANSAC round");
            if (active.empty()) break;
        }
        if (rc) break;
        for (int j = 0; j < nj; ++j) {
            const ransac_state& h = st[(size_t)j];
            if (h.m < 3 || h.best_itr < 0) continue;   // PCR_E_TOO_FEW_ASSOC pair by pair: identity
            double* T = T_init + 16 * (size_t)job_pair[(size_t)j];
            for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
            for (int a = 0; a < 3; ++a) {
                for (int b = 0; b < 3; ++b) T[4 * a + b] = h.bestT[3 * a + b];
                T[4 * a + 3] = h.bestT[9 + a];
            }
        }
    }
    pcr_sync(ctx->stream);
    release();
    return rc;
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
