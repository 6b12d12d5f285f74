// Entry points not implemented yet return PCR_E_UNSUPPORTED (they never fall back to a CPU path).
#include "pcr_internal.h"
extern "C" {
int pcr_iss(pcr_ctx*, const pcr_cloud*, double, double, double, double, int, double*, int32_t*, int32_t*, int*) { return PCR_E_UNSUPPORTED; }
int pcr_knn(pcr_ctx*, const pcr_index*, const double*, int64_t, int, int32_t*, double*) { return PCR_E_UNSUPPORTED; }
int pcr_radius(pcr_ctx*, const pcr_index*, const double*, int64_t, double, int64_t*, const int64_t*, int32_t*, double*) { return PCR_E_UNSUPPORTED; }
int pcr_voxel_keys(pcr_ctx*, const double*, int64_t, double, double*, double*) { return PCR_E_UNSUPPORTED; }
int pcr_voxel_filter(pcr_ctx*, const double*, int64_t, double, int, uint64_t, double*, int64_t*) { return PCR_E_UNSUPPORTED; }
int pcr_voxel_filter_cloud(pcr_ctx*, const pcr_cloud*, double, int, uint64_t, pcr_cloud**) { return PCR_E_UNSUPPORTED; }
}
