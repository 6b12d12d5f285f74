/*
 * TEST INFRASTRUCTURE -- plain-C restatement of the reference's hot path (second, independent
 * checker next to oracle/oracle_np.py).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; nothing under point-cloud-process_amd/ does.
 * Pinned against the reference's own outputs: tests/test_oracle_golden.py runs every function
 * below against tests/golden/*.npz (produced by oracle/ref_harness.py from the reference code).
 * Paths cited are relative to /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* NumPy / Python float floor-division (npy_divmod) -- Pca_and_Voxel_filter/voxel_filter.py:22-24 */
static double floor_divide(double a, double b) {
    if (b == 0.0) return a / b;
    double mod = fmod(a, b), div = (a - mod) / b;
    if (mod != 0.0 && ((b < 0) != (mod < 0))) { mod += b; div -= 1.0; }
    if (div != 0.0) {
        double f = floor(div);
        if (div - f > 0.5) f += 1.0;
        return f;
    }
    return copysign(0.0, a / b);
}

/* voxel_filter.py:20-33: per-point key h (binary64) and D[3]; xyz is (n,3) row-major */
void oc_voxel_keys(const double* xyz, int64_t n, double leaf, double* h, double* D) {
    double mn[3] = {xyz[0], xyz[1], xyz[2]}, mx[3] = {xyz[0], xyz[1], xyz[2]};
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            double v = xyz[3 * i + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    for (int k = 0; k < 3; ++k) D[k] = floor_divide(mx[k] - mn[k], leaf);
    for (int64_t i = 0; i < n; ++i) {
        double hx = floor((xyz[3 * i] - mn[0]) / leaf);
        double hy = floor((xyz[3 * i + 1] - mn[1]) / leaf);
        double hz = floor((xyz[3 * i + 2] - mn[2]) / leaf);
        h[i] = (hx + hy * D[0]) + (hz * D[0]) * D[1];
    }
}

/* exact 1-NN by exhaustive search, squared distance in the direct form (main.py:116-121);
 * ties -> lowest index.  O(n*m): small cases only. */
void oc_nn1(const double* q, int64_t nq, const double* t, int64_t nt, int64_t* idx, double* d2) {
    for (int64_t i = 0; i < nq; ++i) {
        double best = INFINITY;
        int64_t bj = -1;
        for (int64_t j = 0; j < nt; ++j) {
            double dx = q[3 * i] - t[3 * j], dy = q[3 * i + 1] - t[3 * j + 1], dz = q[3 * i + 2] - t[3 * j + 2];
            double d = (dx * dx + dy * dy) + dz * dz;
            if (d < best) { best = d; bj = j; }
        }
        idx[i] = bj;
        d2[i] = best;
    }
}

/* one-sided Jacobi SVD of a 3x3 (row-major): H = U diag(s) V^T */
static void svd3(const double* H, double* U, double* s, double* V) {
    double A[9];
    memcpy(A, H, sizeof(A));
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0);
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) { al += A[3*i+p]*A[3*i+p]; be += A[3*i+q]*A[3*i+q]; ga += A[3*i+p]*A[3*i+q]; }
                if (ga == 0 || fabs(ga) <= 1e-17 * sqrt(al * be)) continue;
                if (fabs(ga) / sqrt(al * be) > off) off = fabs(ga) / sqrt(al * be);
                double z = (be - al) / (2 * ga), tt = (z >= 0 ? 1.0 : -1.0) / (fabs(z) + sqrt(1 + z * z));
                double c = 1 / sqrt(1 + tt * tt), sn = c * tt;
                for (int i = 0; i < 3; ++i) {
                    double ap = A[3*i+p], aq = A[3*i+q], vp = V[3*i+p], vq = V[3*i+q];
                    A[3*i+p] = c*ap - sn*aq; A[3*i+q] = sn*ap + c*aq;
                    V[3*i+p] = c*vp - sn*vq; V[3*i+q] = sn*vp + c*vq;
                }
            }
        if (off < 1e-16) break;
    }
    for (int j = 0; j < 3; ++j) {
        s[j] = sqrt(A[j]*A[j] + A[3+j]*A[3+j] + A[6+j]*A[6+j]);
        for (int i = 0; i < 3; ++i) U[3*i+j] = s[j] > 0 ? A[3*i+j] / s[j] : 0.0;
    }
    /* rank-2 input (e.g. three points): complete U to a right-handed basis */
    double smax = fmax(s[0], fmax(s[1], s[2]));
    for (int b = 0; b < 3; ++b)
        if (s[b] <= 1e-15 * smax) {
            int p = (b + 1) % 3, q = (b + 2) % 3;
            U[b]     = U[3+p]*U[6+q] - U[6+p]*U[3+q];
            U[3 + b] = U[6+p]*U[q]   - U[p]*U[6+q];
            U[6 + b] = U[p]*U[3+q]   - U[3+p]*U[q];
        }
}

/* Procrustes, Registration/main.py:131-141 with A - mean(A) in place of the dense centring matrix
 * (SURVEY 0.3).  A, B are (3,K) row-major.  R = U V^T without reflection fix. */
void oc_procrustes(const double* A, const double* B, int64_t K, double* R, double* t, double* cost) {
    double ma[3] = {0, 0, 0}, mb[3] = {0, 0, 0}, H[9] = {0};
    for (int c = 0; c < 3; ++c) {
        for (int64_t i = 0; i < K; ++i) { ma[c] += A[c * K + i]; mb[c] += B[c * K + i]; }
        ma[c] /= (double)K; mb[c] /= (double)K;
    }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int64_t i = 0; i < K; ++i) s += (B[r * K + i] - mb[r]) * (A[c * K + i] - ma[c]);
            H[3 * r + c] = s;
        }
    double U[9], s[3], V[9];
    svd3(H, U, s, V);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3*i+j] = U[3*i]*V[3*j] + U[3*i+1]*V[3*j+1] + U[3*i+2]*V[3*j+2];
    for (int r = 0; r < 3; ++r) {
        double s2 = 0;
        for (int64_t i = 0; i < K; ++i) s2 += B[r*K+i] - (R[3*r]*A[i] + R[3*r+1]*A[K+i] + R[3*r+2]*A[2*K+i]);
        t[r] = s2 / (double)K;
    }
    double c2 = 0;
    for (int64_t i = 0; i < K; ++i)
        for (int r = 0; r < 3; ++r) {
            double e = B[r*K+i] - (R[3*r]*A[i] + R[3*r+1]*A[K+i] + R[3*r+2]*A[2*K+i] + t[r]);
            c2 += e * e;
        }
    *cost = sqrt(c2);
}

/* icp_point2point, Registration/main.py:97-156 (appendix A of SURVEY.md).  src is (n,3) and is
 * transformed in place like main.py:110.  T0/T_out row-major 4x4.  Returns the number of solves;
 * *failed = 1 when fewer than 3 associations were found (main.py:125-127). */
int oc_icp_point2point(double* src, int64_t n, const double* tgt, int64_t m, const double* T0, double* T_out, int* failed) {
    double T[16], R_last[9], t_last[3];
    memcpy(T, T0, sizeof(T));
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R_last[3*i+j] = T0[4*i+j]; t_last[i] = T0[4*i+3]; }
    int64_t* idx = malloc(sizeof(int64_t) * n);
    double* d2 = malloc(sizeof(double) * n);
    double* A = malloc(sizeof(double) * 3 * n);
    double* B = malloc(sizeof(double) * 3 * n);
    int iters = 0, first = 1;
    *failed = 0;
    for (int it = 0; it < 100; ++it) {                      /* main.py:98 */
        for (int64_t i = 0; i < n; ++i) {                   /* main.py:110 */
            double x = src[3*i], y = src[3*i+1], z = src[3*i+2];
            for (int r = 0; r < 3; ++r) src[3*i+r] = ((T[4*r]*x + T[4*r+1]*y) + T[4*r+2]*z) + T[4*r+3];
        }
        oc_nn1(src, n, tgt, m, idx, d2);
        int64_t K = 0;
        for (int64_t i = 0; i < n; ++i) if (d2[i] < 5.0) ++K;   /* main.py:103,119 */
        if (K < 3) { *failed = 1; break; }
        int64_t k = 0;
        for (int64_t i = 0; i < n; ++i)
            if (d2[i] < 5.0) {
                for (int c = 0; c < 3; ++c) { A[c*K+k] = src[3*i+c]; B[c*K+k] = tgt[3*idx[i]+c]; }
                ++k;
            }
        double R[9], t[3], cost;
        oc_procrustes(A, B, K, R, t, &cost);
        ++iters;
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[4*r+c] = R[3*r+c]; T[4*r+3] = t[r]; }
        T[12] = T[13] = T[14] = 0; T[15] = 1;
        double rd = 0, td = 0;
        for (int i = 0; i < 9; ++i) rd += (R[i] - R_last[i]) * (R[i] - R_last[i]);
        if (first) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) td += (t[i]-t_last[j])*(t[i]-t_last[j]); }   /* main.py:100,150 broadcast */
        else       { for (int i = 0; i < 3; ++i) td += (t[i]-t_last[i])*(t[i]-t_last[i]); }
        first = 0;
        memcpy(R_last, R, sizeof(R_last)); memcpy(t_last, t, sizeof(t_last));
        if (sqrt(rd) <= 0.5 && sqrt(td) <= 0.5) break;       /* main.py:101-102,153 */
    }
    memcpy(T_out, T, sizeof(T));
    free(idx); free(d2); free(A); free(B);
    return iters;
}

/* rotmat2quaternion / homo2tq, main.py:158-174: out = tx,ty,tz,qw,qx,qy,qz */
void oc_homo2tq(const double* T, double* out) {
    double m00=T[0], m01=T[1], m02=T[2], m10=T[4], m11=T[5], m12=T[6], m20=T[8], m21=T[9], m22=T[10];
    double qw = sqrt(fmax(0.0, m00 + m11 + m22 + 1)) / 2;
    double qx = sqrt(fmax(0.0, 1 + m00 - m11 - m22)) / 2;
    double qy = sqrt(fmax(0.0, 1 - m00 + m11 - m22)) / 2;
    double qz = sqrt(fmax(0.0, 1 - m00 - m11 + m22)) / 2;
    if (qx * (m21 - m12) < 0) qx = -qx;
    if (qy * (m02 - m20) < 0) qy = -qy;
    if (qz * (m10 - m01) < 0) qz = -qz;
    out[0] = T[3]; out[1] = T[7]; out[2] = T[11]; out[3] = qw; out[4] = qx; out[5] = qy; out[6] = qz;
}
