// Host side of the registration loop: index build/dispatch, the ICP iteration
// driver (Registration/main.py:97-156 and icp_template.py:128-200 semantics),
// the 3x3 Procrustes solve and the pose utilities.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>
#include "pcr_internal.h"
#include "pcr_linalg.h"
#include "pcr_icp_step.h"

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
    int rc = pcr_cloud_bbox(ctx, target, idx->lo, idx->hi);
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
        PCR_HIP(ctx, pcr_sync(ctx->stream));
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
        PCR_HIP(ctx, pcr_sync(ctx->stream));
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
    // the passes transform the source in place: lay it out first (the Morton sort may still use the box remembered from the
    // upload), then forget that box
    if (index->kind == PCR_INDEX_GRID) {
        const int rs = pcr_cloud_morton_sort(ctx, source, index->cell);
        if (rs) return rs;
    }
    source->has_bbox = false;
    const bool compat = params->mode == PCR_ICP_COMPAT_MAIN;
    PCR_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    int rc = PCR_OK;
    static const bool host_loop_env = getenv("PCR_ICP_HOSTLOOP") != nullptr;
    if (index->kind == PCR_INDEX_GRID && ctx->icp_lanes == 1 && !host_loop_env) {
        // grid index: the whole loop runs on the device, the host only enqueues passes (pcr_grid_search.hip)
        rc = pcr_grid_icp_loop(ctx, index, source, params, T0, res);
    } else {
        // host loop (brute-force index, per-kernel profiling, search lanes): one synchronisation per iteration
        double* d_mom = nullptr;
        if ((rc = pcr_dev_alloc(ctx, sizeof(double) * PCR_NMOM, (void**)&d_mom))) return rc;
        pcr_icp_dev_state st;
        memset(&st, 0, sizeof(st));
        pcr_xform_from_T(T0, &st.x);   // transform to apply at the top of the next pass; COMPAT returns it (T0, then the last increment)
        for (int i = 0; i < 16; ++i) st.T_total[i] = (i % 5 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) st.R_last[3 * i + j] = T0[4 * i + j];
            st.t_last[i] = T0[4 * i + 3];
        }
        st.first = 1;
        for (int i = 0; i < 9; ++i) st.V[i] = (i % 4 == 0) ? 1.0 : 0.0;
        pcr_icp_loop_args la;
        la.max_iter = params->max_iter; la.min_iter = params->min_iter; la.compat = compat ? 1 : 0; la.r_metric = params->r_metric;
        la.r_thres = params->r_thres; la.t_thres = params->t_thres;
        double nn_ms = 0;
        for (int it = 0; it < params->max_iter && !st.stop; ++it) {
            // main.py:110 / icp_template.py:195: the source is transformed in place, fused into the pass
            if (ctx->profile) PCR_HIP(ctx, hipEventRecord(ctx->ev2, ctx->stream));
            // zero-copy read-back: the pass writes the 160 bytes of moments straight into pinned, device-mapped host memory
            rc = icp_pass(ctx, index, source, &st.x, params->max_d2, 1, ctx->zero_copy ? ctx->h_pinned : d_mom);
            if (rc) break;
            if (ctx->profile) PCR_HIP(ctx, hipEventRecord(ctx->ev3, ctx->stream));
            if (!ctx->zero_copy) PCR_HIP(ctx, hipMemcpyAsync(ctx->h_pinned, d_mom, sizeof(double) * PCR_NMOM, hipMemcpyDeviceToHost, ctx->stream));
            PCR_HIP(ctx, pcr_sync(ctx->stream));
            float ms = 0;
            if (ctx->profile) hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3);  // per-pass kernel time only while profiling
            nn_ms += ms;
            pcr::icp_step(&st, ctx->h_pinned, index->view.origin, la, st.r_diff, st.t_diff);
        }
        pcr_dev_free(ctx, d_mom, sizeof(double) * PCR_NMOM);
        if (rc) return rc;
        double T_cur[16];
        pcr::T_from_xform(st.x, T_cur);
        if (!compat && st.status == PCR_OK && !st.converged && st.it == params->max_iter && params->max_iter > 0) {
            // icp_template.py:195-198: a non-converged last iteration still updates src_points and homo_mat_total
            if ((rc = pcr_cloud_transform(ctx, source, T_cur))) return rc;
            pcr::T_mul4(T_cur, st.T_total, st.T_total);
        }
        res->iters = st.it;
        res->status = st.status;
        res->n_assoc = st.n_assoc;
        res->cost = st.cost;
        res->mean_d2 = st.mean_d2;
        for (int i = 0; i < st.it; ++i) { res->r_diff[i] = st.r_diff[i]; res->t_diff[i] = st.t_diff[i]; }
        res->nn_kernel_ms = nn_ms;
        res->nn_launches = st.passes;
        memcpy(res->T_total, st.T_total, sizeof(st.T_total));
        if (compat) memcpy(res->T, T_cur, sizeof(T_cur));
        else memcpy(res->T, st.T_total, sizeof(st.T_total));
    }
    if (rc) return rc;
    if (index->kind == PCR_INDEX_GRID && ctx->loop_dev_ms > 0.0) {
        // device-resident loop: its duration by the kernels' own clock (first kernel of the call .. end of the last pass).  No event
        // behind the loop: on this pool a small operation behind the last big kernel starts 16-45 ms late every 10th-30th call
        // (DESIGN section 3.1.7), and the call has its result already.
        res->device_ms = ctx->loop_dev_ms;
        return res->status;
    }
    PCR_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    PCR_HIP(ctx, pcr_event_sync(ctx->ev1));
    float loop_ms = 0;
    hipEventElapsedTime(&loop_ms, ctx->ev0, ctx->ev1);
    res->device_ms = loop_ms;
    return res->status;
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
