// 3x3 linear algebra for the Procrustes step (binary64), usable from host code and from kernels.
// The library is built with -ffp-contract=off (the searches' proofs count roundings); the functions of this header and of
// pcr_icp_step.h opt back in (#pragma clang fp contract(fast)): on the device the step is ONE lane's chain of ~1 000 dependent
// binary64 operations at the end of every ICP pass, a quarter of them multiply-add pairs -- fused they shorten the pass by 0.7 us
// of 37 (same-box A/B).  Every device kernel compiles the same expressions the same way (the fused batch stays bitwise equal
// to the per-pair path: tested), and nothing here is compared bitwise with the host build of the same functions (goldens: 1e-9).
#pragma once
#include <cmath>
#include <cstring>
#include <hip/hip_runtime.h>

namespace pcr {

// One-sided Jacobi SVD of a 3x3 matrix (row-major): H = U diag(s) V^T.
// Columns of U for zero singular values are completed to an orthonormal basis.
// One Jacobi rotation of columns P, Q.  Written for the single device thread that solves the ICP step: binary64
// divisions and square roots are long dependent sequences there, so the rotation uses two reciprocal square roots
// and the convergence tests compare squares instead of dividing.
template <int P, int Q>
__host__ __device__ inline void svd3_rotate(double A[9], double V[9], bool& rotated) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    const double alpha = (A[P] * A[P] + A[3 + P] * A[3 + P]) + A[6 + P] * A[6 + P];
    const double beta = (A[Q] * A[Q] + A[3 + Q] * A[3 + Q]) + A[6 + Q] * A[6 + Q];
    const double gamma = (A[P] * A[Q] + A[3 + P] * A[3 + Q]) + A[6 + P] * A[6 + Q];
    if (gamma == 0.0) return;
    const double ab = alpha * beta, g2 = gamma * gamma;
    if (g2 <= 1e-34 * ab) return;           // |gamma| <= 1e-17 sqrt(alpha beta): orthogonal to working precision
    // Rotation that zeroes gamma, from the double angle: cos 2t = |d| / h, sin 2t = g / h with d = beta - alpha, g = 2 |gamma|,
    // h = sqrt(d^2 + g^2); c^2 = (1 + cos 2t) / 2, s = sin 2t / (2 c).  Two reciprocal square roots and no division (the
    // textbook zeta / t / c form is three divisions and three square roots; each is a ~12-instruction dependent chain for
    // the single device lane that solves the ICP step).
    const double d = beta - alpha, g = 2.0 * fabs(gamma);
    const bool pos = (d == 0.0) || ((d > 0.0) == (gamma > 0.0));
    const double h2 = d * d + g * g;
#if defined(__HIP_DEVICE_COMPILE__)
    const double rh = rsqrt(h2);
#else
    const double rh = 1.0 / sqrt(h2);
#endif
    const double c2 = 0.5 + 0.5 * (fabs(d) * rh);       // in [1/2, 1]
#if defined(__HIP_DEVICE_COMPILE__)
    const double rc = rsqrt(c2);
#else
    const double rc = 1.0 / sqrt(c2);
#endif
    const double c = c2 * rc;
    const double s_abs = 0.5 * (g * rh) * rc;
    const double sn = pos ? s_abs : -s_abs;
    // Another sweep is needed unless what this sweep leaves behind is below working precision.  A rotation by sine s changes
    // the other two inner products by at most |s| times their size, so after a sweep whose inner products were all below
    // eps (relative) and whose sines were all below 1e-8 the residue is <= eps * 1e-8: eps < 1e-8 needs no further sweep.
    // (Close singular values give large angles for tiny gamma: the sine test catches them.)
    if (g2 >= 1e-16 * ab || (g2 >= 1e-32 * ab && s_abs * s_abs >= 1e-16)) rotated = true;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double ap = A[3 * i + P], aq = A[3 * i + Q];
        A[3 * i + P] = c * ap - sn * aq;
        A[3 * i + Q] = sn * ap + c * aq;
        const double vp = V[3 * i + P], vq = V[3 * i + Q];
        V[3 * i + P] = c * vp - sn * vq;
        V[3 * i + Q] = sn * vp + c * vq;
    }
}

// Completion of U for (numerically) zero singular values.  Written with constant array indices and run-time SELECTS only: as three
// template instantiations behind an if / else chain the compiler merged their stores into one tail with a variable index, U went to
// scratch memory (64 bytes per lane), and a kernel that uses scratch pays for it at its first dispatch after kernels that do not:
// 20 % of the waves of an ICP call's first pass started 6-8 us late (the whole chip waits for the scratch set-up of one rare branch).
// column `bad` of U := cross product of the other two columns
__host__ __device__ inline void svd3_complete_one(double U[9], const int bad) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    // p, q = the other two columns in cyclic order: bad = 0 -> (1, 2), 1 -> (2, 0), 2 -> (0, 1)
    // (all nine elements are read into scalars FIRST: "b ? U[1] : U[2]" written on the array becomes ONE load with a selected address)
    const double u0 = U[0], u1 = U[1], u2 = U[2], u3 = U[3], u4 = U[4], u5 = U[5], u6 = U[6], u7 = U[7], u8 = U[8];
    const bool b0 = bad == 0, b1 = bad == 1, b2 = bad == 2;
    const double p0 = b0 ? u1 : (b1 ? u2 : u0), p1 = b0 ? u4 : (b1 ? u5 : u3), p2 = b0 ? u7 : (b1 ? u8 : u6);
    const double q0 = b0 ? u2 : (b1 ? u0 : u1), q1 = b0 ? u5 : (b1 ? u3 : u4), q2 = b0 ? u8 : (b1 ? u6 : u7);
    const double c0 = p1 * q2 - p2 * q1, c1 = p2 * q0 - p0 * q2, c2 = p0 * q1 - p1 * q0;
    U[0] = b0 ? c0 : u0; U[1] = b1 ? c0 : u1; U[2] = b2 ? c0 : u2;
    U[3] = b0 ? c1 : u3; U[4] = b1 ? c1 : u4; U[5] = b2 ? c1 : u5;
    U[6] = b0 ? c2 : u6; U[7] = b1 ? c2 : u7; U[8] = b2 ? c2 : u8;
}

// rank <= 1: column `good` is the only usable one (or none is: have == false): build any orthonormal completion
__host__ __device__ inline void svd3_complete_two(double U[9], const int good, const bool have) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    const double u0 = U[0], u1 = U[1], u2 = U[2], u3 = U[3], u4 = U[4], u5 = U[5], u6 = U[6], u7 = U[7], u8 = U[8];
    const bool g0 = good == 0, g1 = good == 1, g2 = good == 2;
    const double a0 = have ? (g0 ? u0 : (g1 ? u1 : u2)) : 1.0;
    const double a1 = have ? (g0 ? u3 : (g1 ? u4 : u5)) : 0.0;
    const double a2 = have ? (g0 ? u6 : (g1 ? u7 : u8)) : 0.0;
    const int ax = (fabs(a0) <= fabs(a1) && fabs(a0) <= fabs(a2)) ? 0 : (fabs(a1) <= fabs(a2) ? 1 : 2);
    const double e0 = ax == 0 ? 1.0 : 0.0, e1 = ax == 1 ? 1.0 : 0.0, e2 = ax == 2 ? 1.0 : 0.0;
    double v10 = a1 * e2 - a2 * e1, v11 = a2 * e0 - a0 * e2, v12 = a0 * e1 - a1 * e0;
    const double n1 = sqrt(v10 * v10 + v11 * v11 + v12 * v12);
    v10 /= n1; v11 /= n1; v12 /= n1;
    const double v20 = a1 * v12 - a2 * v11, v21 = a2 * v10 - a0 * v12, v22 = a0 * v11 - a1 * v10;
    // columns in the order `good`, then the remaining two ascending: the first remaining column is 1 if good == 0, else 0
    const bool f0 = !g0, f1 = g0;   // column 0 / column 1 is the FIRST remaining column
    U[0] = g0 ? a0 : (f0 ? v10 : v20); U[1] = g1 ? a0 : (f1 ? v10 : v20); U[2] = g2 ? a0 : v20;
    U[3] = g0 ? a1 : (f0 ? v11 : v21); U[4] = g1 ? a1 : (f1 ? v11 : v21); U[5] = g2 ? a1 : v21;
    U[6] = g0 ? a2 : (f0 ? v12 : v22); U[7] = g1 ? a2 : (f1 ? v12 : v22); U[8] = g2 ? a2 : v22;
}

// V0 (optional): an orthogonal starting basis, e.g. the V of a nearby matrix (consecutive ICP iterations): the sweeps
// then start from A = H V0, which is already close to orthogonal columns, and usually one or two sweeps remain.
// Any orthogonal V0 yields a valid decomposition; R = U V^T does not depend on it beyond rounding.
__host__ __device__ inline void svd3(const double H[9], double U[9], double s[3], double V[9], const double* V0 = nullptr) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    double A[9];
    if (V0) {
        for (int i = 0; i < 9; ++i) V[i] = V0[i];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) A[3 * i + j] = H[3 * i] * V0[j] + H[3 * i + 1] * V0[3 + j] + H[3 * i + 2] * V0[6 + j];
    } else {
        for (int i = 0; i < 9; ++i) A[i] = H[i];
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    // the three column pairs are spelled out with constant indices: with a (p, q) loop the arrays are indexed dynamically
    // and live in scratch memory on the device -- ~20 dependent memory round trips per rotation in the single thread that
    // solves the ICP step
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        svd3_rotate<0, 1>(A, V, rotated);
        svd3_rotate<0, 2>(A, V, rotated);
        svd3_rotate<1, 2>(A, V, rotated);
        if (!rotated) break;
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
        if (ok[j]) {
            const double inv = 1.0 / nrm[j];
            for (int i = 0; i < 3; ++i) U[3 * i + j] = A[3 * i + j] * inv;
        }
    }
    // complete U for (numerically) zero singular values (constant indices only: see svd3_rotate)
    const int nbad = (!ok[0]) + (!ok[1]) + (!ok[2]);
    if (nbad == 1) svd3_complete_one(U, !ok[0] ? 0 : (!ok[1] ? 1 : 2));
    else if (nbad >= 2) svd3_complete_two(U, ok[2] ? 2 : (ok[1] ? 1 : 0), ok[2] || ok[1] || ok[0]);
}

__host__ __device__ inline void mat3_mul(const double A[9], const double B[9], double C[9]) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// Procrustes from raw moments taken about `origin`:
//   m = {K, Sa[3], Sb[3], Sba[9] (b_i a_j), Saa, Sbb}
// R = U V^T of H = sum (b-bbar)(a-abar)^T (no reflection fix, like Registration/main.py:137-139),
// t = bbar - R abar, cost = ||B - (R A + t)||_F (main.py:140-141).
__host__ __device__ inline void kabsch_from_moments(const double m[18], const double origin[3], double R[9], double t[3], double* cost,
                                                    double* V_io = nullptr) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    const double K = m[0], invK = 1.0 / K;
    double abar[3] = {m[1] * invK, m[2] * invK, m[3] * invK};
    double bbar[3] = {m[4] * invK, m[5] * invK, m[6] * invK};
    double H[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) H[3 * i + j] = m[7 + 3 * i + j] - K * bbar[i] * abar[j];
    double U[9], s[3], V[9];
    svd3(H, U, s, V, V_io);
    if (V_io)
        for (int i = 0; i < 9; ++i) V_io[i] = V[i];
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
