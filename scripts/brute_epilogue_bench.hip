// Microbenchmark: inner loop of the brute-force f64 MFMA sweep with different min/argmin epilogues.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-honor-nans scripts/brute_epilogue_bench.hip -o scripts/bin/brute_epi
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int NT = 4, PF = 4;

template <int V>
__global__ void __launch_bounds__(256, 2) k(const double* __restrict__ A, long long n_tiles, int n_splits, double* __restrict__ out, int* __restrict__ iout) {
    const int lane = threadIdx.x & 63;
    const int split = blockIdx.y;
    double bq[NT];
    for (int tt = 0; tt < NT; ++tt) bq[tt] = 0.001 * (lane + 64 * tt + blockIdx.x);
    const long long per = (n_tiles + n_splits - 1) / n_splits;
    const long long t0 = split * per, t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    const v4f64 zero = {0, 0, 0, 0};
    double best[NT]; int btile[NT]; unsigned int bhi[NT], bse[NT];
    for (int tt = 0; tt < NT; ++tt) { bhi[tt] = 0xffffffffu; bse[tt] = 0xffffffffu; }
    v4f64 best4[NT];
    for (int tt = 0; tt < NT; ++tt) { best[tt] = DBL_MAX; btile[tt] = -1; best4[tt] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX}; }
    double a_cur[PF], a_nxt[PF];
    for (int i = 0; i < PF; ++i) a_cur[i] = A[(t0 + i) * 64 + lane];
    v4f64 acc[NT];
    for (int tt = 0; tt < NT; ++tt) acc[tt] = {DBL_MAX, DBL_MAX, DBL_MAX, DBL_MAX};
    for (long long tb = t0; tb < t1; tb += PF) {
        for (int i = 0; i < PF; ++i) a_nxt[i] = A[(tb + PF + i) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            if (V == 0) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], acc[tt], 0, 0, 0);
            } else if (V == 1) {
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const double m = fmin(fmin(acc[tt][0], acc[tt][1]), fmin(acc[tt][2], acc[tt][3]));
                    const bool lt = m < best[tt];
                    best[tt] = lt ? m : best[tt];
                    btile[tt] = lt ? (int)(tb + i - 1) : btile[tt];
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            } else if (V == 2) {  // 4 separate minima, no index
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) best4[tt][r] = fmin(best4[tt][r], acc[tt][r]);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            } else if (V == 3) {  // interleaved: mfma(tt) then epilogue(tt) of the previous tile
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
                    const double m = fmin(fmin(acc[tt][0], acc[tt][1]), fmin(acc[tt][2], acc[tt][3]));
                    const bool lt = m < best[tt];
                    best[tt] = lt ? m : best[tt];
                    btile[tt] = lt ? (int)(tb + i - 1) : btile[tt];
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            } else if (V == 4) {  // tile index packed into the low 16 mantissa bits, 4 separate minima
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
                const unsigned int code = (unsigned int)(tb + i - 1 - t0) & 0xffffu;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        unsigned long long b = (unsigned long long)__double_as_longlong(acc[tt][r]);
                        unsigned int lo = ((unsigned int)b & 0xffff0000u) | code;
                        const double v = __longlong_as_double((long long)((b & 0xffffffff00000000ull) | lo));
                        best4[tt][r] = fmin(best4[tt][r], v);
                    }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            } else if (V == 5) {  // min of the 4 rows, then ONE packed-index min (3 + 1 mins, 1 and_or)
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
                const unsigned int code = (unsigned int)(tb + i - 1 - t0) & 0xffffu;
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const double m = fmin(fmin(acc[tt][0], acc[tt][1]), fmin(acc[tt][2], acc[tt][3]));
                    unsigned long long b = (unsigned long long)__double_as_longlong(m);
                    unsigned int lo = ((unsigned int)b & 0xffff0000u) | code;
                    best[tt] = fmin(best[tt], __longlong_as_double((long long)((b & 0xffffffff00000000ull) | lo)));
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            } else if (V == 6) {  // all-integer epilogue on the 64-bit patterns (values biased positive)
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    unsigned long long u0 = (unsigned long long)__double_as_longlong(acc[tt][0]), u1 = (unsigned long long)__double_as_longlong(acc[tt][1]);
                    unsigned long long u2 = (unsigned long long)__double_as_longlong(acc[tt][2]), u3 = (unsigned long long)__double_as_longlong(acc[tt][3]);
                    u0 = u0 < u1 ? u0 : u1;
                    u2 = u2 < u3 ? u2 : u3;
                    u0 = u0 < u2 ? u0 : u2;
                    const unsigned long long b = (unsigned long long)__double_as_longlong(best[tt]);
                    const bool lt = u0 < b;
                    best[tt] = __longlong_as_double((long long)(lt ? u0 : b));
                    btile[tt] = lt ? (int)(tb + i - 1) : btile[tt];
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            } else if (V == 7) {  // high dwords only: 3 v_min_u32 + cmp + 2 cndmask
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const unsigned int h0 = (unsigned int)__double2hiint(acc[tt][0]), h1 = (unsigned int)__double2hiint(acc[tt][1]);
                    const unsigned int h2 = (unsigned int)__double2hiint(acc[tt][2]), h3 = (unsigned int)__double2hiint(acc[tt][3]);
                    const unsigned int m = min(min(h0, h1), min(h2, h3));
                    const bool lt = m < bhi[tt];
                    bhi[tt] = lt ? m : bhi[tt];
                    btile[tt] = lt ? (int)(tb + i - 1) : btile[tt];
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            } else if (V == 8) {  // high dwords: min + second-min (filter) + tile
                v4f64 cur[NT];
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) cur[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[i], bq[tt], zero, 0, 0, 0);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const unsigned int h0 = (unsigned int)__double2hiint(acc[tt][0]), h1 = (unsigned int)__double2hiint(acc[tt][1]);
                    const unsigned int h2 = (unsigned int)__double2hiint(acc[tt][2]), h3 = (unsigned int)__double2hiint(acc[tt][3]);
                    const unsigned int lo01 = min(h0, h1), hi01 = max(h0, h1), lo23 = min(h2, h3), hi23 = max(h2, h3);
                    const unsigned int m = min(lo01, lo23);
                    const unsigned int s4 = min(max(lo01, lo23), min(hi01, hi23));
                    bse[tt] = min(min(bse[tt], s4), max(bhi[tt], m));
                    const bool lt = m < bhi[tt];
                    bhi[tt] = lt ? m : bhi[tt];
                    btile[tt] = lt ? (int)(tb + i - 1) : btile[tt];
                }
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) acc[tt] = cur[tt];
            }
        }
        for (int i = 0; i < PF; ++i) a_cur[i] = a_nxt[i];
    }
    double s = 0; int si = 0;
    for (int tt = 0; tt < NT; ++tt) {
        s += best[tt] == DBL_MAX ? 0 : best[tt];
        si += btile[tt] + (int)bhi[tt] + (int)bse[tt];
        for (int r = 0; r < 4; ++r) s += (acc[tt][r] == DBL_MAX ? 0 : acc[tt][r]) + (best4[tt][r] == DBL_MAX ? 0 : best4[tt][r]);
    }
    out[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = s;
    iout[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x] = si;
}

template <int V>
void run(const double* dA, long long n_tiles, int nq_blocks, int n_splits, double* dout, int* diout) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<V>, dim3(nq_blocks, n_splits), dim3(256), 0, 0, dA, n_tiles, n_splits, dout, diout);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<V>, dim3(nq_blocks, n_splits), dim3(256), 0, 0, dA, n_tiles, n_splits, dout, diout);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = 2.0 * 4 * 16 * 16 * (double)n_tiles * (nq_blocks * 16.0);  // per MFMA 2*16*16*4; NT*4 waves... = n_tiles * (queries/16) MFMAs
    printf("variant %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 78.6)\n", V, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
}

int main() {
    const long long n = 120000, n_tiles = (n + 15) / 16;
    const int nq_blocks = (120000 + 255) / 256, n_splits = 8;
    std::vector<double> hA((n_tiles + 16) * 64);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = (double)((i * 2654435761u) % 1000) * 0.01;
    double* dA; double* dout; int* diout;
    hipMalloc(&dA, hA.size() * 8);
    hipMalloc(&dout, (size_t)nq_blocks * n_splits * 256 * 8);
    hipMalloc(&diout, (size_t)nq_blocks * n_splits * 256 * 4);
    hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
    run<0>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<1>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<2>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<3>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<4>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<5>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<6>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<7>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    run<8>(dA, n_tiles, nq_blocks, n_splits, dout, diout);
    return 0;
}
