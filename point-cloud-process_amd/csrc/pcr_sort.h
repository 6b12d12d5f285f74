// (key, value) sorts of the set-up paths: rocPRIM's radix sort with the merge-sort cross-over moved.  Its default configuration
// merge-sorts up to ~1 M pairs; measured on MI355X (scripts/sort_bench.hip, 30-bit keys, us): 120 000 pairs merge 48-58 /
// onesweep 103-118; 1 M pairs merge 160-190 / onesweep 103-134; 10 M pairs 400-550 either way.  So: merge sort below 400 000
// pairs, onesweep above.
#pragma once
#include <rocprim/rocprim.hpp>

using pcr_sort_config = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 400000>;

template <typename K, typename V>
static inline hipError_t pcr_sort_pairs(void* temp, size_t& temp_bytes, K* keys_in, K* keys_out, V* vals_in, V* vals_out, size_t n, unsigned int end_bit,
                                        hipStream_t stream) {
    return rocprim::radix_sort_pairs<pcr_sort_config>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream);
}
