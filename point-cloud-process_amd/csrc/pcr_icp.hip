// Host side of the registration loop: index build/dispatch, the ICP iteration
// driver (Registration/main.py:97-156 and icp_template.py:128-200 semantics),
// the 3x3 Procrustes solve and the pose utilities.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include "pcr_internal.h"
#include "pcr_linalg.h"

extern "C" {

int pcr_index_build(pcr_ctx* ctx, const pcr_cloud* target, int kind, double cell, pcr_index** out) {
    if (!ctx || !target || !out) return PCR_E_INVALID;
    *out = nullptr;
    if (target->n <= 0) return PCR_E_EMPTY;
    if (kind != PCR_INDEX_GRID && kind != PCR_INDEX_BRUTE) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    pcr_index* idx = new pcr_index();
    idx->kind = kind;
    idx->n = target->n;
    memset(&idx->view, 0, sizeof(idx->view));
    int rc = pcr_bbox(ctx, target->d, target->n, idx->lo, idx->hi);
    if (rc == PCR_OK) {
        for (int k = 0; k < 3; ++k) idx->view.origin[k] = 0.5 * (idx->lo[k] + idx->hi[k]);
        rc = (kind == PCR_INDEX_GRID) ? pcr_grid_build(ctx, target, cell, idx) : pcr_brute_build(ctx, target, idx);
    }
    if (rc != PCR_OK) {
        pcr_index_free(ctx, idx);
        return rc;
    }
    *out = idx;
    return PCR_OK;
}

int pcr_index_free(pcr_ctx* ctx, pcr_index* idx) {
    if (!idx) return PCR_OK;
    if (!ctx) return PCR_E_INVALID;
    if (idx->kind == PCR_INDEX_GRID) pcr_grid_free(ctx, idx);
    else pcr_brute_free(ctx, idx);
    delete idx;
    return PCR_OK;
}

int pcr_index_kind(const pcr_index* idx) { return idx ? idx->kind : -1; }
double pcr_index_cell(const pcr_index* idx) { return idx ? idx->cell : 0.0; }
int64_t pcr_index_size(const pcr_index* idx) { return idx ? idx->n : 0; }

int pcr_nn1(pcr_ctx* ctx, const pcr_index* index, const pcr_cloud* queries, const double* T, double max_d2, int32_t* idx_out,
            double* d2_out) {
    if (!ctx || !index || !queries || !idx_out || !d2_out) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    const int64_t nq = queries->n;
    int32_t* d_idx = nullptr;
    double* d_d2 = nullptr;
    int rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(int32_t) * nq, (void**)&d_idx))) return rc;
    if ((rc = pcr_dev_alloc(ctx, sizeof(double) * nq, (void**)&d_d2))) return rc;
    pcr_xform x;
    if (T) pcr_xform_from_T(T, &x);
    if (index->kind == PCR_INDEX_GRID) rc = pcr_grid_nn1(ctx, index, const_cast<pcr_cloud*>(queries), T ? &x : nullptr, max_d2, d_idx, d_d2);
    else rc = pcr_brute_nn1(ctx, index, queries->d, nq, T ? &x : nullptr, max_d2, d_idx, d_d2);
    if (rc == PCR_OK) {
        PCR_HIP(ctx, hipMemcpyAsync(idx_out, d_idx, sizeof(int32_t) * nq, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipMemcpyAsync(d2_out, d_d2, sizeof(double) * nq, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    pcr_dev_free(ctx, d_idx, sizeof(int32_t) * nq);
    pcr_dev_free(ctx, d_d2, sizeof(double) * nq);
    return rc;
}

void pcr_icp_default_params(pcr_icp_params* p) {
    if (!p) return;
    memset(p, 0, sizeof(*p));
    p->max_iter = 100;   // Registration/main.py:98
    p->r_thres = 0.5;    // main.py:101
    p->t_thres = 0.5;    // main.py:102
    p->max_d2 = 5.0;     // main.py:103 (squared distance)
    p->mode = PCR_ICP_COMPAT_MAIN;
    p->r_metric = PCR_RMETRIC_FROBENIUS;
    p->min_iter = 0;
}

}  // extern "C"

static int icp_pass(pcr_ctx* ctx, const pcr_index* index, pcr_cloud* qc, const pcr_xform* x, double max_d2, int write_back,
                    double* d_mom) {
    if (index->kind == PCR_INDEX_GRID) return pcr_grid_icp_pass(ctx, index, qc, x, max_d2, write_back, d_mom);
    return pcr_brute_icp_pass(ctx, index, qc->d, qc->n, x, max_d2, write_back, d_mom);
}

static void T_from_Rt(const double R[9], const double t[3], double T[16]) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = t[i];
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

static void T_mul(const double A[16], const double B[16], double C[16]) {
    double r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            r[4 * i + j] = s;
        }
    memcpy(C, r, sizeof(r));
}

extern "C" {

int pcr_icp_moments(pcr_ctx* ctx, const pcr_cloud* source, const pcr_index* index, const double* T, double max_d2,
                    double moments_out[18], double origin_out[3], double* sum_d2_out) {
    if (!ctx || !source || !index || !moments_out) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    pcr_xform x;
    pcr_xform_from_T(T, &x);
    double* d_mom = nullptr;
    int rc = pcr_dev_alloc(ctx, sizeof(double) * PCR_NMOM, (void**)&d_mom);
    if (rc) return rc;
    rc = icp_pass(ctx, index, const_cast<pcr_cloud*>(source), &x, max_d2, 0, d_mom);
    if (rc == PCR_OK) {
        PCR_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, d_mom, sizeof(double) * PCR_NMOM, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(moments_out, ctx->h_pinned, sizeof(double) * 18);
        if (sum_d2_out) *sum_d2_out = ctx->h_pinned[18];
        if (origin_out)
            for (int k = 0; k < 3; ++k) origin_out[k] = index->view.origin[k];
    }
    pcr_dev_free(ctx, d_mom, sizeof(double) * PCR_NMOM);
    return rc;
}

int pcr_icp(pcr_ctx* ctx, pcr_cloud* source, const pcr_index* index, const pcr_icp_params* params, const double T0[16],
            pcr_icp_result* res) {
    if (!ctx || !source || !index || !params || !T0 || !res) return PCR_E_INVALID;
    if (params->max_iter > PCR_ICP_MAX_LOG) return PCR_E_TOO_MANY_ITERS;
    if (source->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    memset(res, 0, sizeof(*res));
    const bool compat = params->mode == PCR_ICP_COMPAT_MAIN;
    double* d_mom = nullptr;
    int rc = pcr_dev_alloc(ctx, sizeof(double) * PCR_NMOM, (void**)&d_mom);
    if (rc) return rc;

    double T_cur[16];   // transform to apply at the top of the next pass
    double T_ret[16];   // what COMPAT returns: T0, then the last increment
    double T_total[16]; // composed transform actually applied/solved so far
    memcpy(T_cur, T0, sizeof(T_cur));
    memcpy(T_ret, T0, sizeof(T_ret));
    for (int i = 0; i < 16; ++i) T_total[i] = (i % 5 == 0) ? 1.0 : 0.0;
    double R_last[9], t_last[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R_last[3 * i + j] = T0[4 * i + j];
        t_last[i] = T0[4 * i + 3];
    }
    bool first = true;
    bool pending = true;  // T_cur not yet applied to the source
    int status = PCR_OK;
    PCR_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    double nn_ms = 0;
    int launches = 0;
    for (int it = 0; it < params->max_iter; ++it) {
        pcr_xform x;
        pcr_xform_from_T(T_cur, &x);
        // main.py:110 / icp_template.py:195: the source is transformed in place, fused into the pass
        if (ctx->profile) PCR_HIP(ctx, hipEventRecord(ctx->ev2, ctx->stream));
        // zero-copy read-back: the last block of the pass writes the 160 bytes of moments straight into pinned,
        // device-mapped host memory (no copy command in the stream, one wait per iteration)
        rc = icp_pass(ctx, index, source, &x, params->max_d2, 1, ctx->zero_copy ? ctx->h_pinned : d_mom);
        if (rc) break;
        if (ctx->profile) PCR_HIP(ctx, hipEventRecord(ctx->ev3, ctx->stream));
        if (!ctx->zero_copy) PCR_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, d_mom, sizeof(double) * PCR_NMOM, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        float ms = 0;
        if (ctx->profile) hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3);  // per-pass kernel time only while profiling
        nn_ms += ms;
        ++launches;
        T_mul(T_cur, T_total, T_total);
        pending = false;
        const double* m = ctx->h_pinned;
        const int64_t K = (int64_t)llround(m[0]);
        res->n_assoc = K;
        res->mean_d2 = K > 0 ? m[18] / (double)K : 0.0;
        if (K < 3) {  // main.py:125-127
            status = PCR_E_TOO_FEW_ASSOC;
            break;
        }
        double R[9], t[3], cost;
        pcr::kabsch_from_moments(m, index->view.origin, R, t, &cost);
        res->cost = cost;
        res->iters = it + 1;
        // convergence (main.py:149-154)
        double r_diff;
        if (params->r_metric == PCR_RMETRIC_GEODESIC) {
            double tr = 0;
            for (int i = 0; i < 9; ++i) tr += R[i] * R_last[i];  // trace(R_last^T R)
            double c = (tr - 1.0) * 0.5;
            c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
            r_diff = acos(c);
        } else {
            double s = 0;
            for (int i = 0; i < 9; ++i) s += (R[i] - R_last[i]) * (R[i] - R_last[i]);
            r_diff = sqrt(s);
        }
        double t_diff;
        if (compat && first) {
            // main.py:100,150: t is (3,1), t_last is (3,) -> broadcast to 3x3, Frobenius norm
            double s = 0;
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) s += (t[i] - t_last[j]) * (t[i] - t_last[j]);
            t_diff = sqrt(s);
        } else {
            double s = 0;
            for (int i = 0; i < 3; ++i) s += (t[i] - t_last[i]) * (t[i] - t_last[i]);
            t_diff = sqrt(s);
        }
        first = false;
        res->r_diff[it] = r_diff;
        res->t_diff[it] = t_diff;
        memcpy(R_last, R, sizeof(R_last));
        memcpy(t_last, t, sizeof(t_last));
        T_from_Rt(R, t, T_cur);
        memcpy(T_ret, T_cur, sizeof(T_ret));
        pending = true;
        const bool converged = (r_diff <= params->r_thres && t_diff <= params->t_thres) && (it + 1 >= params->min_iter);
        if (converged) break;
        if (!compat && it + 1 == params->max_iter) {
            // icp_template.py:195-198: a non-converged last iteration still updates src_points and homo_mat_total
            rc = pcr_cloud_transform(ctx, source, T_cur);
            if (rc) break;
            T_mul(T_cur, T_total, T_total);
            pending = false;
        }
    }
    (void)pending;
    PCR_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    PCR_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float loop_ms = 0;
    hipEventElapsedTime(&loop_ms, ctx->ev0, ctx->ev1);
    pcr_dev_free(ctx, d_mom, sizeof(double) * PCR_NMOM);
    if (rc) return rc;
    res->status = status;
    res->device_ms = loop_ms;
    res->nn_kernel_ms = nn_ms;
    res->nn_launches = launches;
    memcpy(res->T_total, T_total, sizeof(T_total));
    if (compat) memcpy(res->T, T_ret, sizeof(T_ret));
    else memcpy(res->T, T_total, sizeof(T_total));
    return status;
}

int pcr_procrustes(const double* A, const double* B, int64_t k, double R_out[9], double t_out[3], double* cost_out) {
    if (!A || !B || !R_out || !t_out) return PCR_E_INVALID;
    if (k <= 0) return PCR_E_EMPTY;
    // moments about the first target point (keeps the sums well conditioned)
    double o[3] = {B[0], B[k], B[2 * k]};
    double m[18];
    for (int i = 0; i < 18; ++i) m[i] = 0;
    m[0] = (double)k;
    for (int64_t i = 0; i < k; ++i) {
        double a[3] = {A[i] - o[0], A[k + i] - o[1], A[2 * k + i] - o[2]};
        double b[3] = {B[i] - o[0], B[k + i] - o[1], B[2 * k + i] - o[2]};
        for (int c = 0; c < 3; ++c) { m[1 + c] += a[c]; m[4 + c] += b[c]; }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) m[7 + 3 * r + c] += b[r] * a[c];
        m[16] += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        m[17] += b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    }
    pcr::kabsch_from_moments(m, o, R_out, t_out, cost_out);
    return PCR_OK;
}

static double copysign_ref(double v, double s) {  // main.py:176-180
    if (v * s < 0) v *= -1;
    return v;
}

int pcr_homo2tq(const double T[16], double out[7]) {
    if (!T || !out) return PCR_E_INVALID;
    const double m00 = T[0], m01 = T[1], m02 = T[2], m10 = T[4], m11 = T[5], m12 = T[6], m20 = T[8], m21 = T[9], m22 = T[10];
    double trace = m00 + m11 + m22;
    double qw = sqrt(fmax(0.0, trace + 1)) / 2;
    double qx = sqrt(fmax(0.0, 1 + m00 - m11 - m22)) / 2;
    double qy = sqrt(fmax(0.0, 1 - m00 + m11 - m22)) / 2;
    double qz = sqrt(fmax(0.0, 1 - m00 - m11 + m22)) / 2;
    qx = copysign_ref(qx, m21 - m12);
    qy = copysign_ref(qy, m02 - m20);
    qz = copysign_ref(qz, m10 - m01);
    out[0] = T[3]; out[1] = T[7]; out[2] = T[11];
    out[3] = qw; out[4] = qx; out[5] = qy; out[6] = qz;
    return PCR_OK;
}

}  // extern "C"

extern "C" int pcr_cloud_prepare(pcr_ctx* ctx, pcr_cloud* cloud, const pcr_index* index) {
    if (!ctx || !cloud || !index) return PCR_E_INVALID;
    hipSetDevice(ctx->device);
    if (index->kind != PCR_INDEX_GRID) return PCR_OK;  // the brute-force sweep does not care about record order
    return pcr_cloud_morton_sort(ctx, cloud, index->cell);
}
