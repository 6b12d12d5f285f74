// One step of the ICP driver, shared by the host loop (pcr_icp.hip: brute-force index, profiling) and the device
// loop (last block of grid_accumulate_kernel): compose the transform just applied, Procrustes from the moments,
// convergence test and bookkeeping -- Registration/main.py:125-154 (COMPAT) / icp_template.py:166-198 (TOTAL).
#pragma once
#include <cstddef>
#include "pcr_internal.h"
#include "pcr_linalg.h"

namespace pcr {

__host__ __device__ inline void T_from_xform(const pcr_xform& x, double T[16]) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = x.r[3 * i + j];
        T[4 * i + 3] = x.t[i];
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

__host__ __device__ inline void T_mul4(const double A[16], const double B[16], double C[16]) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h  // C may alias A or B
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            r[4 * i + j] = s;
        }
    for (int i = 0; i < 16; ++i) C[i] = r[i];
}

// `m` = the 20 accumulated moments of the pass that just applied st->x to the source.  Only the head of *st (everything
// before the logs) is touched; the per-iteration logs go to r_log / t_log, so that a caller may work on a copy of the head.
constexpr size_t ICP_STATE_HEAD_BYTES = offsetof(pcr_icp_dev_state, r_diff);
__host__ __device__ inline void icp_step(pcr_icp_dev_state* st, const double* m, const double origin[3], const pcr_icp_loop_args& la,
                                         double* r_log, double* t_log) {
#pragma clang fp contract(fast)   // see the note at the top of pcr_linalg.h
    double T_cur[16];
    T_from_xform(st->x, T_cur);
    T_mul4(T_cur, st->T_total, st->T_total);
    st->passes += 1;
    const long long K = (long long)llround(m[0]);
    st->n_assoc = K;
    st->mean_d2 = K > 0 ? m[18] / (double)K : 0.0;
    if (K < 3) {  // main.py:125-127
        st->status = PCR_E_TOO_FEW_ASSOC;
        st->stop = 1;
        return;
    }
    double R[9], t[3], cost;
    kabsch_from_moments(m, origin, R, t, &cost, st->V);   // warm start from the previous iteration's right singular vectors
    st->cost = cost;
    const int it = st->it;
    st->it = it + 1;
    // convergence (main.py:149-154)
    double r_diff;
    if (la.r_metric == PCR_RMETRIC_GEODESIC) {
        double tr = 0;
        for (int i = 0; i < 9; ++i) tr += R[i] * st->R_last[i];  // trace(R_last^T R)
        double c = (tr - 1.0) * 0.5;
        c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
        r_diff = acos(c);
    } else {
        double s = 0;
        for (int i = 0; i < 9; ++i) s += (R[i] - st->R_last[i]) * (R[i] - st->R_last[i]);
        r_diff = sqrt(s);
    }
    double t_diff;
    if (la.compat && st->first) {
        // main.py:100,150: t is (3,1), t_last is (3,) -> broadcast to 3x3, Frobenius norm
        double s = 0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) s += (t[i] - st->t_last[j]) * (t[i] - st->t_last[j]);
        t_diff = sqrt(s);
    } else {
        double s = 0;
        for (int i = 0; i < 3; ++i) s += (t[i] - st->t_last[i]) * (t[i] - st->t_last[i]);
        t_diff = sqrt(s);
    }
    st->first = 0;
    r_log[it] = r_diff;
    t_log[it] = t_diff;
    for (int i = 0; i < 9; ++i) { st->R_last[i] = R[i]; st->x.r[i] = R[i]; }
    for (int i = 0; i < 3; ++i) { st->t_last[i] = t[i]; st->x.t[i] = t[i]; }
    const bool converged = (r_diff <= la.r_thres && t_diff <= la.t_thres) && (it + 1 >= la.min_iter);
    if (converged) { st->converged = 1; st->stop = 1; }
    if (it + 1 >= la.max_iter) st->stop = 1;
}

}  // namespace pcr
