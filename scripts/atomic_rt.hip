// Round trip of a returning atomic add by scope (one wave, dependent chain), and of the poll pattern of the ICP pass
// (agent-scope load of a word another XCD wrote).  MI355X: device-scope atomics are performed at the memory side
// (the L2 of an XCD is not coherent with the other seven), workgroup-scope ones in the XCD's own L2.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE>
__global__ void chain(unsigned long long* p, unsigned long long* out, int n) {
    unsigned long long v = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) v += __hip_atomic_fetch_add(p + (v & 1), 1ull, __ATOMIC_RELAXED, SCOPE);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
template <int SCOPE>
__global__ void chain_load(unsigned long long* p, unsigned long long* out, int n) {
    unsigned long long v = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) v += __hip_atomic_load(p + (v & 1), __ATOMIC_RELAXED, SCOPE);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = v; }
}
int main() {
    unsigned long long *p, *o, h[2];
    hipMalloc(&p, 4096); hipMalloc(&o, 64); hipMemset(p, 0, 4096);
    const int n = 2000;
#define RUN(K, NAME) do { hipMemset(p, 0, 4096); hipLaunchKernelGGL(K, dim3(1), dim3(64), 0, 0, p, o, n); hipMemcpy(h, o, 16, hipMemcpyDeviceToHost); \
        printf("%-44s %7.1f ns per dependent operation\n", NAME, 10.0 * (double)h[0] / n); } while (0)
    for (int rep = 0; rep < 2; ++rep) {
        RUN(chain<__HIP_MEMORY_SCOPE_AGENT>, "returning add, agent scope");
        RUN(chain<__HIP_MEMORY_SCOPE_WORKGROUP>, "returning add, workgroup scope");
        RUN(chain<__HIP_MEMORY_SCOPE_WAVEFRONT>, "returning add, wavefront scope");
        RUN(chain_load<__HIP_MEMORY_SCOPE_AGENT>, "atomic load, agent scope");
        RUN(chain_load<__HIP_MEMORY_SCOPE_WORKGROUP>, "atomic load, workgroup scope");
    }
    return 0;
}
