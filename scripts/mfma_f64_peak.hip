// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (and v_fma_f64 / v_min_f64) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_f64_peak.hip -o gpurun_out/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void mfma_loop(double* out, unsigned long long* cyc, int iters) {
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0.0, 0.0, 0.0, 0.0};
    double a = threadIdx.x * 1e-3 + 1.0, b = threadIdx.x * 2e-3 + 0.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

__global__ void fma64_loop(double* out, unsigned long long* cyc, int iters) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    double a = 1.0000001, b = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
    int cus = 256;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    cus = p.multiProcessorCount;
    const int iters = 20000;
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, sizeof(double) * cus * 8 * 256 * 4);
    hipMalloc(&cyc, sizeof(unsigned long long) * cus * 8 * 4 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int wpc = 4; wpc <= 8; wpc += 4) {  // waves per CU: 4 = one per SIMD, 8 = two per SIMD
        int blocks = cus * (wpc / 4);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += v;
        mean /= h.size();
        double n_mfma = (double)iters * 4;
        double flops = (double)blocks * 4 * n_mfma * 2048.0;
        printf("f64 MFMA 16x16x4: %d waves/CU: %.1f cycles per MFMA per wave (s_memtime), %.3f ms, %.1f TFLOP/s chip-wide\n", wpc,
               mean / n_mfma, ms, flops / (ms * 1e-3) / 1e12);
    }
    for (int wpc = 4; wpc <= 16; wpc += 12) {
        int blocks = cus * (wpc / 4);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(fma64_loop, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += v;
        mean /= h.size();
        double n = (double)iters * 8;
        printf("v_fma_f64: %d waves/CU: %.2f cycles per instr per wave, %.3f ms, %.1f TFLOP/s chip-wide\n", wpc, mean / n, ms,
               (double)blocks * 256 * n * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
