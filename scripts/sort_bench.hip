// microbenchmark: rocPRIM sorts of (Morton key, index) pairs at the sizes of the set-up path (20k, 120k, 1M)
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
template <typename K, typename F>
static float time_it(F f, hipStream_t st, int reps = 20) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps * 1e3f;
}
template <typename K>
static void run(size_t n, int bits) {
    hipStream_t st; hipStreamCreate(&st);
    std::vector<K> h(n); std::mt19937_64 g(1);
    for (auto& v : h) v = (K)(g() & ((1ull << bits) - 1));
    K *k1, *k2; unsigned int *v1, *v2;
    hipMalloc(&k1, n * sizeof(K)); hipMalloc(&k2, n * sizeof(K)); hipMalloc(&v1, n * 4); hipMalloc(&v2, n * 4);
    hipMemcpy(k1, h.data(), n * sizeof(K), hipMemcpyHostToDevice);
    size_t tb = 0; void* tmp = nullptr;
    rocprim::radix_sort_pairs(nullptr, tb, k1, k2, v1, v2, n, 0, bits, st);
    size_t tb2 = 0;
    rocprim::merge_sort(nullptr, tb2, k1, k2, v1, v2, n, rocprim::less<K>(), st);
    if (tb2 > tb) tb = tb2;
    hipMalloc(&tmp, tb + 256);
    float t_def = time_it<K>([&] { rocprim::radix_sort_pairs(tmp, tb, k1, k2, v1, v2, n, 0, bits, st); }, st);
    using cfg = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 0>;   // merge-sort limit 0: always onesweep
    size_t tb3 = 0;
    rocprim::radix_sort_pairs<cfg>(nullptr, tb3, k1, k2, v1, v2, n, 0, bits, st);
    void* tmp3 = nullptr; hipMalloc(&tmp3, tb3 + 256);
    float t_one = time_it<K>([&] { rocprim::radix_sort_pairs<cfg>(tmp3, tb3, k1, k2, v1, v2, n, 0, bits, st); }, st);
    float t_mrg = time_it<K>([&] { rocprim::merge_sort(tmp, tb, k1, k2, v1, v2, n, rocprim::less<K>(), st); }, st);
    printf("n=%8zu key=%zuB bits=%2d : radix_sort_pairs default %7.1f us | onesweep forced %7.1f us | merge_sort %7.1f us\n", n, sizeof(K), bits, t_def, t_one, t_mrg);
    hipFree(k1); hipFree(k2); hipFree(v1); hipFree(v2); hipFree(tmp); hipFree(tmp3);
}
int main() {
    for (size_t n : {20000ul, 120000ul, 1000000ul, 10240000ul}) {
        run<unsigned long long>(n, 30); run<unsigned int>(n, 30); run<unsigned int>(n, 24); run<unsigned long long>(n, 39);
    }
    return 0;
}
