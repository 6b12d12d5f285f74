// Where does global_load_lds_dwordx3 put lane l's 12 bytes?  (prints the LDS image of one wave-instruction)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const unsigned int* src, unsigned int* out) {
    __shared__ unsigned int lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    typedef __attribute__((address_space(1))) const void* gp_t;
    typedef __attribute__((address_space(3))) void* lp_t;
    __builtin_amdgcn_global_load_lds((gp_t)(src + 3 * threadIdx.x), (lp_t)(lds + 8), 12, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    unsigned int h[512], *d, *o;
    for (int i = 0; i < 512; ++i) h[i] = i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(h));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 192; ++i) ok &= h[8 + i] == (unsigned)i;
    printf("contiguous 12-byte lanes at the given base: %s\n", ok ? "yes" : "NO");
    for (int i = 0; i < 40; ++i) printf("%x ", h[i]);
    printf("\n... "); for (int i = 190; i < 270; ++i) printf("%x ", h[i]);
    printf("\n");
    return ok ? 0 : 1;
}
