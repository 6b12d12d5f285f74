// 3x3 linear algebra for the Procrustes step (binary64), usable from host code and from kernels.
#pragma once
#include <cmath>
#include <cstring>
#include <hip/hip_runtime.h>

namespace pcr {

// One-sided Jacobi SVD of a 3x3 matrix (row-major): H = U diag(s) V^T.
// Columns of U for zero singular values are completed to an orthonormal basis.
__host__ __device__ inline void svd3(const double H[9], double U[9], double s[3], double V[9]) {
    double A[9];
    for (int i = 0; i < 9; ++i) A[i] = H[i];
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p) {
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 3; ++i) {
                    alpha += A[3 * i + p] * A[3 * i + p];
                    beta += A[3 * i + q] * A[3 * i + q];
                    gamma += A[3 * i + p] * A[3 * i + q];
                }
                if (gamma == 0.0) continue;
                double lim = 1e-17 * sqrt(alpha * beta);
                if (fabs(gamma) <= lim) continue;
                off = fmax(off, fabs(gamma) / sqrt(alpha * beta));
                double zeta = (beta - alpha) / (2.0 * gamma);
                double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < 3; ++i) {
                    double ap = A[3 * i + p], aq = A[3 * i + q];
                    A[3 * i + p] = c * ap - sn * aq;
                    A[3 * i + q] = sn * ap + c * aq;
                    double vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = c * vp - sn * vq;
                    V[3 * i + q] = sn * vp + c * vq;
                }
            }
        }
        if (off < 1e-16) break;
    }
    double nrm[3];
    double nmax = 0;
    for (int j = 0; j < 3; ++j) {
        nrm[j] = sqrt(A[j] * A[j] + A[3 + j] * A[3 + j] + A[6 + j] * A[6 + j]);
        nmax = fmax(nmax, nrm[j]);
    }
    bool ok[3];
    for (int j = 0; j < 3; ++j) {
        s[j] = nrm[j];
        ok[j] = nrm[j] > 1e-300 && nrm[j] > 1e-15 * nmax;
        if (ok[j])
            for (int i = 0; i < 3; ++i) U[3 * i + j] = A[3 * i + j] / nrm[j];
    }
    // complete U for (numerically) zero singular values
    int nbad = (!ok[0]) + (!ok[1]) + (!ok[2]);
    if (nbad == 1) {
        int b = !ok[0] ? 0 : (!ok[1] ? 1 : 2);
        int p = (b + 1) % 3, q = (b + 2) % 3;
        double c0 = U[3 * 1 + p] * U[3 * 2 + q] - U[3 * 2 + p] * U[3 * 1 + q];
        double c1 = U[3 * 2 + p] * U[3 * 0 + q] - U[3 * 0 + p] * U[3 * 2 + q];
        double c2 = U[3 * 0 + p] * U[3 * 1 + q] - U[3 * 1 + p] * U[3 * 0 + q];
        U[b] = c0; U[3 + b] = c1; U[6 + b] = c2;
    } else if (nbad >= 2) {
        // rank <= 1: build any orthonormal completion
        double u0[3] = {1, 0, 0};
        int g = -1;
        for (int j = 0; j < 3; ++j) if (ok[j]) g = j;
        if (g >= 0) { u0[0] = U[g]; u0[1] = U[3 + g]; u0[2] = U[6 + g]; }
        int ax = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0 : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
        double e[3] = {0, 0, 0};
        e[ax] = 1.0;
        double v1[3] = {u0[1] * e[2] - u0[2] * e[1], u0[2] * e[0] - u0[0] * e[2], u0[0] * e[1] - u0[1] * e[0]};
        double n1 = sqrt(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2]);
        for (int i = 0; i < 3; ++i) v1[i] /= n1;
        double v2[3] = {u0[1] * v1[2] - u0[2] * v1[1], u0[2] * v1[0] - u0[0] * v1[2], u0[0] * v1[1] - u0[1] * v1[0]};
        int cols[3], nc = 0;
        if (g < 0) g = 0;
        cols[nc++] = g;
        for (int j = 0; j < 3; ++j) if (j != g) cols[nc++] = j;
        const double* vecs[3] = {u0, v1, v2};
        for (int c = 0; c < 3; ++c)
            for (int i = 0; i < 3; ++i) U[3 * i + cols[c]] = vecs[c][i];
    }
}

__host__ __device__ inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Procrustes from raw moments taken about `origin`:
//   m = {K, Sa[3], Sb[3], Sba[9] (b_i a_j), Saa, Sbb}
// R = U V^T of H = sum (b-bbar)(a-abar)^T (no reflection fix, like Registration/main.py:137-139),
// t = bbar - R abar, cost = ||B - (R A + t)||_F (main.py:140-141).
__host__ __device__ inline void kabsch_from_moments(const double m[18], const double origin[3], double R[9], double t[3], double* cost) {
    const double K = m[0];
    double abar[3] = {m[1] / K, m[2] / K, m[3] / K};
    double bbar[3] = {m[4] / K, m[5] / K, m[6] / K};
    double H[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) H[3 * i + j] = m[7 + 3 * i + j] - K * bbar[i] * abar[j];
    double U[9], s[3], V[9];
    svd3(H, U, s, V);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] = U[3 * i] * V[3 * j] + U[3 * i + 1] * V[3 * j + 1] + U[3 * i + 2] * V[3 * j + 2];
    // means in world coordinates
    double aw[3] = {abar[0] + origin[0], abar[1] + origin[1], abar[2] + origin[2]};
    double bw[3] = {bbar[0] + origin[0], bbar[1] + origin[1], bbar[2] + origin[2]};
    for (int i = 0; i < 3; ++i) t[i] = bw[i] - (R[3 * i] * aw[0] + R[3 * i + 1] * aw[1] + R[3 * i + 2] * aw[2]);
    if (cost) {
        double saa = m[16] - K * (abar[0] * abar[0] + abar[1] * abar[1] + abar[2] * abar[2]);
        double sbb = m[17] - K * (bbar[0] * bbar[0] + bbar[1] * bbar[1] + bbar[2] * bbar[2]);
        double tr = 0;  // trace(R^T H) = sum_ij R_ij H_ij
        for (int i = 0; i < 9; ++i) tr += R[i] * H[i];
        double c2 = sbb + saa - 2.0 * tr;
        *cost = c2 > 0 ? sqrt(c2) : 0.0;
    }
}

}  // namespace pcr
