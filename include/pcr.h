/*
 * pcr.h -- C ABI of the MI355X-native point-cloud registration hot path.
 *
 * One shared library (libpcr.so, hipcc --offload-arch=gfx950) exports exactly
 * these symbols.  Plain C: opaque handles, caller-owned host buffers, integer
 * status codes, no exceptions across the boundary.  Every entry point names
 * the reference interface (file:line under /root/reference) it stands in for.
 *
 * Threading: a pcr_ctx owns one device and one HIP stream and is not
 * thread-safe; distinct contexts are independent.  Host arrays are only read
 * (or written) during the call.
 *
 * Arithmetic: the whole path computes in IEEE binary64 like the reference
 * (Open3D points are double; NumPy float64), with FMA contraction disabled in
 * device code so squared distances are bit-identical to
 * (dx*dx + dy*dy) + dz*dz evaluated on the host.
 */
#ifndef PCR_H
#define PCR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCR_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- status */
enum {
    PCR_OK = 0,
    /* soft status: result is valid.  Registration/main.py:125-127 prints
     * "ICP failed, cannot find enough associations!" and returns the current
     * transformation; so do we. */
    PCR_E_TOO_FEW_ASSOC = 1,
    PCR_E_INVALID = -1,     /* bad argument                                  */
    PCR_E_EMPTY = -2,       /* empty cloud (reference: IndexError/ValueError) */
    PCR_E_NOMEM = -3,
    PCR_E_HIP = -4,         /* HIP runtime error; see pcr_last_error()        */
    PCR_E_NO_DEVICE = -5,
    PCR_E_UNSUPPORTED = -6,
    PCR_E_TOO_MANY_ITERS = -7
};

typedef struct pcr_ctx pcr_ctx;
typedef struct pcr_cloud pcr_cloud;
typedef struct pcr_index pcr_index;

PCR_API const char* pcr_strerror(int status);
PCR_API const char* pcr_last_error(const pcr_ctx* ctx); /* last HIP error text */
PCR_API const char* pcr_version(void);

/* --------------------------------------------------------------- context */
PCR_API int pcr_ctx_create(int device, pcr_ctx** out);
PCR_API int pcr_ctx_destroy(pcr_ctx* ctx);
PCR_API int pcr_ctx_sync(pcr_ctx* ctx);
/* device facts for the bench: name (<=255 chars), CU count, HBM bytes */
PCR_API int pcr_ctx_device_info(pcr_ctx* ctx, char* name256, int* cu_count, int64_t* hbm_bytes);

/* ---------------------------------------------------------------- clouds
 * Device-resident (n,3) float64 cloud stored as 32-byte records {x,y,z,id}.
 * Stands in for o3d.geometry.PointCloud.points (Registration/main.py:52-56)
 * and the (N,3) ndarrays of Kdtree_Octree/lesson2 and voxel_filter.py:19.
 * float32 input (the .bin readers, main.py:10-17) is widened exactly.       */
PCR_API int pcr_cloud_upload_f32(pcr_ctx* ctx, const float* xyz, int64_t n, int64_t stride_floats, pcr_cloud** out);
PCR_API int pcr_cloud_upload_f64(pcr_ctx* ctx, const double* xyz, int64_t n, int64_t stride_doubles, pcr_cloud** out);
PCR_API int pcr_cloud_download_f64(pcr_ctx* ctx, const pcr_cloud* cloud, double* xyz_out /* n*3 */);
PCR_API int64_t pcr_cloud_size(const pcr_cloud* cloud);
PCR_API int pcr_cloud_free(pcr_ctx* ctx, pcr_cloud* cloud);
/* Optional: lay the cloud out for queries against `index` now (Morton order of its records; row
 * ids are kept, downloads are unaffected).  pcr_nn1 / pcr_icp do this themselves on first use. */
PCR_API int pcr_cloud_prepare(pcr_ctx* ctx, pcr_cloud* cloud, const pcr_index* index);
/* PointCloud.transform(T) in place (Registration/main.py:110), T row-major 4x4 */
PCR_API int pcr_cloud_transform(pcr_ctx* ctx, pcr_cloud* cloud, const double T[16]);

/* ----------------------------------------------------------- target index
 * Stands in for o3d.geometry.KDTreeFlann(target) (Registration/main.py:105)
 * and kdtree_construction / octree_construction
 * (Kdtree_Octree/lesson2/kdtree.py:119-137, octree.py:310-328).             */
enum { PCR_INDEX_GRID = 0, PCR_INDEX_BRUTE = 1 };
/* cell <= 0: choose the finest cell size from the cloud's extent and size */
PCR_API int pcr_index_build(pcr_ctx* ctx, const pcr_cloud* target, int kind, double cell, pcr_index** out);
PCR_API int pcr_index_free(pcr_ctx* ctx, pcr_index* index);
PCR_API int pcr_index_kind(const pcr_index* index);
PCR_API double pcr_index_cell(const pcr_index* index);
PCR_API int64_t pcr_index_size(const pcr_index* index);

/* Exact 1-NN of every query point (optionally transformed by T first) in the
 * indexed cloud: the body of the association loop Registration/main.py:116-121
 * == find_associations, icp_template.py:113-126.  idx_out[i] = target index or
 * -1 when the nearest neighbour is not closer than max_d2 (strict <, on the
 * SQUARED distance like main.py:119); max_d2 <= 0 or +inf disables the gate.
 * d2_out[i] = exact squared distance (binary64) of the reported neighbour.   */
PCR_API int pcr_nn1(pcr_ctx* ctx, const pcr_index* index, const pcr_cloud* queries, const double* T /* 16 or NULL */,
                    double max_d2, int32_t* idx_out, double* d2_out);

/* Batched k-NN / radius queries: kdtree_knn_search / octree_knn_search
 * (kdtree.py:141-172, octree.py:262-306) and *_radius_search
 * (kdtree.py:176-208, octree.py:166-259), Q queries per call.
 * knn: idx/dist are (Q,k), ascending distance, EUCLIDEAN (not squared) like
 * result_set.py; unfilled slots (k > n) hold index 0 / distance 1e10 like
 * KNNResultSet.__init__ (result_set.py:19-22).
 * radius: two-pass.  Call with idx == NULL to get counts[Q] (neighbours with
 * distance <= r, inclusive like result_set.py:80); then with offsets[Q+1]
 * (exclusive prefix sums of counts) to fill idx/dist, ascending distance.    */
PCR_API int pcr_knn(pcr_ctx* ctx, const pcr_index* index, const double* queries_xyz, int64_t q, int k,
                    int32_t* idx_out, double* dist_out);
PCR_API int pcr_radius(pcr_ctx* ctx, const pcr_index* index, const double* queries_xyz, int64_t q, double radius,
                       int64_t* counts_out, const int64_t* offsets, int32_t* idx_out, double* dist_out);

/* A handful of radius queries (q <= 64) in one call and one launch -- the reference's API is one query per call (kdtree.py:176-208,
 * octree.py:166-259): counts_out[q] = neighbours of every query; idx_out / dist_out receive the lists back to back (query i at the
 * sum of the counts before it), ascending distance, ties by index; they must hold q * cap entries.  A query with more than `cap`
 * neighbours makes the call return PCR_E_UNSUPPORTED with counts_out filled: take the two-pass pcr_radius then.                  */
PCR_API int pcr_radius_small(pcr_ctx* ctx, const pcr_index* index, const double* queries_xyz, int q, double radius, int64_t cap,
                             int64_t* counts_out, int32_t* idx_out, double* dist_out);

/* -------------------------------------------------------------------- ICP */
enum { PCR_ICP_COMPAT_MAIN = 0, /* Registration/main.py:97-156 semantics, returns LAST increment */
       PCR_ICP_TOTAL = 1        /* icp_template.py:128-200 semantics, returns composed transform */ };
enum { PCR_RMETRIC_FROBENIUS = 0, PCR_RMETRIC_GEODESIC = 1 };
#define PCR_ICP_MAX_LOG 256

typedef struct pcr_icp_params {
    int32_t max_iter;   /* main.py:98  -> 100 */
    double r_thres;     /* main.py:101 -> 0.5 */
    double t_thres;     /* main.py:102 -> 0.5 */
    double max_d2;      /* main.py:103 -> 5 (threshold on SQUARED distance) */
    int32_t mode;       /* PCR_ICP_COMPAT_MAIN | PCR_ICP_TOTAL */
    int32_t r_metric;   /* PCR_RMETRIC_* (template hint icp_template.py:184) */
    int32_t min_iter;   /* bench only: never break before this many iterations (0 = reference behaviour) */
    int32_t reserved;
} pcr_icp_params;

typedef struct pcr_icp_result {
    double T[16];          /* row-major 4x4: last increment (COMPAT) or composed (TOTAL) */
    double T_total[16];    /* composed transform in both modes                          */
    int32_t iters;         /* Procrustes solves performed                               */
    int32_t status;        /* PCR_OK or PCR_E_TOO_FEW_ASSOC                             */
    int64_t n_assoc;       /* associations of the last solved iteration                 */
    double cost;           /* ||B - (R A + t)||_F of the last solve (main.py:141)       */
    double mean_d2;        /* mean squared NN distance of the last association pass     */
    double r_diff[PCR_ICP_MAX_LOG]; /* log["R_diff"], icp_template.py:189 */
    double t_diff[PCR_ICP_MAX_LOG]; /* log["t_diff"], icp_template.py:190 */
    double device_ms;      /* duration of the whole loop on the device: the kernels' own 100-MHz clock, first kernel .. end of the last pass (device-resident loop), HIP events otherwise */
    double nn_kernel_ms;   /* sum of HIP-event times of the pass kernels; 0 unless pcr_profile_enable(ctx, 1) (host loop: always) */
    int32_t nn_launches;   /* launches of the correspondence kernel                     */
    int32_t reserved;
} pcr_icp_result;

PCR_API void pcr_icp_default_params(pcr_icp_params* p);
/* icp_point2point(source, target, transformation) (main.py:97) / ICP (icp_template.py:128).
 * `source` is updated in place exactly like main.py:110 mutates it (COMPAT) or
 * icp_template.py:195-196 (TOTAL).                                            */
PCR_API int pcr_icp(pcr_ctx* ctx, pcr_cloud* source, const pcr_index* target_index, const pcr_icp_params* params,
                    const double T0[16], pcr_icp_result* result);

/* Batched scan-pair registration: the loop Registration/main.py:190-216 (read pair, register, keep the pose) for many
 * independent pairs at once.  pairs[i] are caller-owned host buffers (records of `stride` values, x,y,z first: stride 6 is
 * the registration_dataset .bin record of main.py:10-17, stride 4 the KITTI record); T0 may be NULL (identity).
 * The n_ctx contexts (one device, one HIP stream each) are driven by n_ctx native worker threads that take pairs from a
 * shared counter: upload both clouds, build the grid index, run pcr_icp, free -- several pairs in flight per GPU, no
 * interpreter in the loop.  results[i] belongs to pairs[i] whatever thread ran it; status_out[i] is pcr_icp's return
 * value for that pair.  Returns the first hard error (< 0) or PCR_OK.  The contexts must not be used concurrently
 * by the caller. */
typedef struct pcr_pair {
    const float* src;     /* n_src records of stride_src floats */
    int64_t n_src;
    int64_t stride_src;
    const float* tgt;
    int64_t n_tgt;
    int64_t stride_tgt;
    const double* T0;     /* 16 doubles, row-major, or NULL */
} pcr_pair;
PCR_API int pcr_icp_batch(pcr_ctx* const* ctxs, int n_ctx, const pcr_pair* pairs, int64_t n_pairs, const pcr_icp_params* params,
                          pcr_icp_result* results, int32_t* status_out);

/* One association + accumulation pass (no solve): the 18 moments the Procrustes
 * step needs: {K, Sa[3], Sb[3], Sba[9] (row-major b_i*a_j), Saa, Sbb}, taken about
 * `origin_out[3]`.  Lets tests check the fused kernel against the oracle.      */
PCR_API int pcr_icp_moments(pcr_ctx* ctx, const pcr_cloud* source, const pcr_index* target_index, const double* T,
                            double max_d2, double moments_out[18], double origin_out[3], double* sum_d2_out);
/* procrustes_transformation(A, B) (icp_template.py:43-54, main.py:131-141) on host
 * arrays A,B laid out (3,K) row-major; R_out[9], t_out[3].                     */
PCR_API int pcr_procrustes(const double* A, const double* B, int64_t k, double R_out[9], double t_out[3], double* cost_out);
/* rotmat2quaternion / homo2tq (main.py:158-174): out = tx,ty,tz,qw,qx,qy,qz */
PCR_API int pcr_homo2tq(const double T[16], double out7[7]);

/* ----------------------------------------------------------- voxel filter
 * voxel_filter(point_cloud, leaf_size, type) (Pca_and_Voxel_filter/voxel_filter.py:10-68).
 * pcr_voxel_keys: per-point key h (float64, bit-exact, voxel_filter.py:20-33) and D[3].
 * pcr_voxel_filter: mode 0 = "centroid", 1 = "random" (explicit seed).  Output
 * rows = occupied voxels - 1 (the reference never emits its last group,
 * voxel_filter.py:42-51); out must hold n*3 doubles.
 * mode 2 = Open3D's voxel_down_sample as called at Registration/main.py:35:
 * origin min - leaf/2, every occupied voxel emitted, centroid = running sum in
 * input order / count; rows ordered by voxel key (Open3D's order is that of
 * its hash map and is not part of its contract).                             */
PCR_API int pcr_voxel_keys(pcr_ctx* ctx, const double* xyz, int64_t n, double leaf, double* h_out, double D_out[3]);
PCR_API int pcr_voxel_filter(pcr_ctx* ctx, const double* xyz, int64_t n, double leaf, int mode, uint64_t seed,
                             double* out_xyz, int64_t* n_out);
/* device-resident variant used by the downsample -> ICP pipeline (config 3) */
PCR_API int pcr_voxel_filter_cloud(pcr_ctx* ctx, const pcr_cloud* in, double leaf, int mode, uint64_t seed, pcr_cloud** out);

/* --------------------------------------------------------------------- ISS
 * Keypoint_detection_ISS/ISS.py:35-73.  lambdas_out (n,3) descending eigenvalues
 * of the weighted scatter; counts_out[n] = |N(p_i)| (inclusive radius, self
 * included).  keypoints_out holds up to max_keypoints + 1 indices after NMS
 * (ISS.py:72-73 stops once MORE than iss_count were taken).  lambdas_out and
 * counts_out may be NULL (they are 28 bytes per point over PCIe); at least
 * one of lambdas_out / keypoints_out must be asked for.                       */
PCR_API int pcr_iss(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, double gamma21, double gamma32, double nms_radius,
                    int max_keypoints, double* lambdas_out, int32_t* counts_out, int32_t* keypoints_out, int* n_keypoints_out);

/* ---------------------------------------------------------- PCA / normals
 * pcr_pca: Pca_and_Voxel_filter/pca_normal.py:10-36 PCA(data, sort=True):
 * eigen-decomposition of np.cov of the cloud (divisor n-1); eigvals_out
 * descending, eigvecs_out row-major 3x3 whose COLUMNS are the eigenvectors
 * (sign arbitrary, like LAPACK's in the reference).  mean_out may be NULL.
 * pcr_normals: pca_normal.py:85-90 -- for every point the eigenvector of the
 * smallest eigenvalue of the covariance of its k nearest neighbours (the point
 * itself included, like search_knn_vector_3d on a cloud point); 2 <= k <= 16.
 * normals_out (n,3) by caller row; eigvals_out (n,3) descending and
 * neighbours_out (n,k) ascending (distance, index) may be NULL.             */
PCR_API int pcr_pca(pcr_ctx* ctx, const pcr_cloud* cloud, double eigvals_out[3], double eigvecs_out[9], double mean_out[3]);
PCR_API int pcr_normals(pcr_ctx* ctx, const pcr_cloud* cloud, int k, double* normals_out, double* eigvals_out,
                        int32_t* neighbours_out);

/* ------------------------------------------- global initialisation (next row)
 * The Open3D stage in front of ICP, Registration/main.py:33-84, and the template
 * surface icp_template.py:20-41,56-110.  Open3D is absent and unpinned in the
 * reference; these follow its published behaviour ("parity unpinned").
 * pcr_normals_hybrid: estimate_normals(KDTreeSearchParamHybrid(radius, max_nn))
 *   (main.py:39-40): neighbourhood = the <= max_nn nearest points with
 *   d^2 < radius^2; fewer than 3 -> (0,0,1).  orient != 0 flips every normal
 *   toward viewpoint[3] (NULL = origin); orient == 0 leaves the solver's sign.
 * pcr_fpfh: compute_fpfh_feature (main.py:44-46); normals (n,3) by row;
 *   features_out (n,33) row-major (= Open3D's Feature.data (33,n) column-major).
 * pcr_feature_match: nearest target row in feature space for every query row
 *   (find_matchings, icp_template.py:20-41); squared L2, ties to the lowest row.
 * pcr_ransac: registration_ransac_based_on_feature_matching's loop
 *   (main.py:73-83) / ransac_init's loop (icp_template.py:88-110) over a given
 *   correspondence set corr (m,2) of (source row, target row): 3 samples,
 *   edge-length and distance checkers, Kabsch, inliers counted over corr,
 *   running best in iteration order with the confidence-based early exit.    */
typedef struct pcr_ransac_params {
    int32_t max_iteration;
    int32_t check_distance;
    double confidence;
    double max_distance;
    double edge_similarity; /* <= 0 switches the edge-length checker off */
    uint64_t seed;
    double reserved[4];
} pcr_ransac_params;
typedef struct pcr_ransac_result {
    double T[16];
    int32_t iterations, n_valid, best_iteration, reserved_i;
    double corr_fitness, corr_rmse;
    double reserved[4];
} pcr_ransac_result;
PCR_API int pcr_normals_hybrid(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, int max_nn, int orient, const double viewpoint[3],
                               double* normals_out);
PCR_API int pcr_fpfh(pcr_ctx* ctx, const pcr_cloud* cloud, const double* normals, double radius, int max_nn, double* features_out);
PCR_API int pcr_feature_match(pcr_ctx* ctx, const double* queries, int64_t nq, const double* targets, int64_t nt, int dim,
                              int32_t* idx_out, double* d2_out);
PCR_API int pcr_ransac_default_params(pcr_ransac_params* p);
PCR_API int pcr_ransac(pcr_ctx* ctx, const pcr_cloud* source, const pcr_cloud* target, const int32_t* corr, int64_t m,
                       const pcr_ransac_params* params, pcr_ransac_result* result);

/* Device-resident variant of the same stage -- what the pair loop main.py:190-216 runs per pair, without a host round trip
 * between its steps.
 * pcr_preprocess = preprocess_point_cloud(pcd, voxel_size) (main.py:33-47): voxel_down_sample(voxel_size) (mode 2 of
 *   pcr_voxel_filter), estimate_normals(Hybrid(normal_radius, normal_max_nn)) oriented toward the origin,
 *   compute_fpfh_feature(Hybrid(fpfh_radius, fpfh_max_nn)); main.py:39,44 use 2 x / 5 x voxel_size and 30 / 100.  The result
 *   (down-sampled cloud, normals, 33-d descriptors) stays on the device in a pcr_prep, to be used for every pair the scan
 *   takes part in (342 pairs over 504 scans in Registration/reg_result.txt).  A pcr_prep belongs to the context that made it
 *   (pcr_prep_free with that context) but may be READ by any context of the same device once pcr_preprocess has returned.
 * pcr_global_registration = execute_global_registration (main.py:68-84): nearest descriptors both ways, mutual filter (falls
 *   back to the one-way set below 9 survivors, like Open3D), pcr_ransac's loop over that set; result->reserved_i = size of
 *   the correspondence set.  The final whole-cloud evaluation Open3D appends (fitness / inlier_rmse of the RegistrationResult)
 *   is not part of it: main.py:211 reads .transformation only.                                                            */
typedef struct pcr_prep pcr_prep;
PCR_API int pcr_preprocess(pcr_ctx* ctx, const pcr_cloud* cloud, double voxel_size, double normal_radius, int normal_max_nn,
                           double fpfh_radius, int fpfh_max_nn, pcr_prep** out);
PCR_API int64_t pcr_prep_size(const pcr_prep* prep);
PCR_API const pcr_cloud* pcr_prep_cloud(const pcr_prep* prep);   /* the down-sampled cloud (owned by the prep) */
/* points (n,3), normals (n,3), features (n,33) row-major; any of them may be NULL */
PCR_API int pcr_prep_download(pcr_ctx* ctx, const pcr_prep* prep, double* points, double* normals, double* features);
PCR_API int pcr_prep_free(pcr_ctx* ctx, pcr_prep* prep);
PCR_API int pcr_global_registration(pcr_ctx* ctx, const pcr_prep* source, const pcr_prep* target, const pcr_ransac_params* params,
                                    int mutual_filter, pcr_ransac_result* result);

/* The whole pair loop of Registration/main.py:183-216 over a table of scans: pairs[i] = (source scan, target scan) as indices
 * into clouds[] (the rows "trg,src" of the pair list, main.py:186-194; clouds[] = the <id>.bin files read by read_bin_velodyne,
 * main.py:10-17).  A scan that takes part in several pairs (Registration/reg_result.txt: 342 pairs over 504 scans) is uploaded
 * and, with `global`, preprocessed ONCE per call and kept on the device while the call runs.
 *   global != NULL: prepare_dataset + execute_global_registration (main.py:196-203) give the initial transform of every pair
 *     whose T0 is NULL: pcr_preprocess of every scan such a pair uses, pcr_global_registration per pair (the same ransac seed
 *     for every pair, so a pair's result does not depend on which other pairs share the call); a pair for which no hypothesis
 *     passes the checkers starts from identity.  T_init_out (n_pairs x 16, may be NULL) receives the transforms ICP started from.
 *     (Inside, the stage runs fused for the whole call on the first context -- every scan down-sampled by one sort, every later step
 *     one launch for all scans / all pairs -- and scan by scan for a share that does not fit that path: the same results bit for bit.)
 *   then icp_point2point (main.py:211) for every pair through pcr_icp_batch's fused stages.
 * results / status_out as pcr_icp_batch.                                                                                   */
typedef struct pcr_cloud_ref {
    const float* xyz;     /* n records of `stride` floats, x, y, z first */
    int64_t n;
    int64_t stride;
} pcr_cloud_ref;
typedef struct pcr_pair_ref {
    int32_t src, tgt;     /* rows of clouds[] */
    const double* T0;     /* 16 doubles, row-major, or NULL (identity, or the global registration's result) */
} pcr_pair_ref;
typedef struct pcr_global_params {
    double voxel_size;        /* main.py:196 -> 2.0 */
    double normal_radius;     /* main.py:38  -> voxel_size * 2 */
    double fpfh_radius;       /* main.py:43  -> voxel_size * 5 */
    int32_t normal_max_nn;    /* main.py:40  -> 30 */
    int32_t fpfh_max_nn;      /* main.py:46  -> 100 */
    int32_t mutual_filter;    /* main.py:74  -> 1 */
    int32_t reserved_i;
    pcr_ransac_params ransac; /* main.py:70-83 -> max_distance voxel_size * 1.5, edge 0.9, 100000 iterations / 0.999 */
} pcr_global_params;
PCR_API int pcr_global_default_params(double voxel_size, pcr_global_params* p);
PCR_API int pcr_register_pairs(pcr_ctx* const* ctxs, int n_ctx, const pcr_cloud_ref* clouds, int64_t n_clouds, const pcr_pair_ref* pairs,
                               int64_t n_pairs, const pcr_global_params* global, const pcr_icp_params* icp, pcr_icp_result* results,
                               int32_t* status_out, double* T_init_out);

/* ------------------------------------------------------------------ DBSCAN
 * DBSCAN.fit (Cluster_dbscan/dbscan.py:10-36), a consumer of the radius query:
 * labels_out[n] = cluster id per row (-1 noise), numbered in the reference's
 * discovery order (seeds taken from the END of the index list; a seed needs
 * >= min_pts neighbours, a reached point expands only with > min_pts; a point
 * first met as a noise seed stays noise).  Neighbourhood = distance <= radius,
 * the point itself included (scipy query_ball_point).                        */
PCR_API int pcr_dbscan(pcr_ctx* ctx, const pcr_cloud* cloud, double radius, int min_pts, int32_t* labels_out, int32_t* n_clusters_out);

/* ------------------------------------------------------------- timing aid
 * HIP-event stopwatch on the ctx stream, for bench.py's roofline figures.   */
PCR_API int pcr_timer_start(pcr_ctx* ctx);
PCR_API int pcr_timer_stop_ms(pcr_ctx* ctx, double* ms_out);
/* Per-kernel HIP-event profile of the ICP pass (adds one event sync per pass while on).
 * slots: 0 = the one-launch ICP pass (grid: tiles, work queue, moments, Procrustes step; in the two-launch variant and in
 *        ungated runs: the tile launch) / MFMA sweep (brute), 1 = the drain launch of the two-launch variant (ungated runs:
 *        hard stage) / merge + exact fallback (brute), 2 = separate accumulate kernel (grid, ungated runs only) / final
 *        (brute), 3 = reduce (brute).
 * ms_out[4] = summed milliseconds (each slot includes the launch gap in front of it), *passes_out = passes profiled. */
/* diagnostics: per-block {cycles, work} stamps of the last grid search stage kernels (PCR_DEBUG_STAMPS=1) */
PCR_API int pcr_debug_read(pcr_ctx* ctx, uint64_t* out, int64_t n_words);
PCR_API int pcr_profile_enable(pcr_ctx* ctx, int on);
/* diagnostics of the LAST correspondence search on this context (Registration/main.py:116-121 is the step they describe):
 * out[0] = queries the brute-force MFMA sweep could not prove and re-did with the exact direct-form sweep;
 * out[1] = device arenas (256-MiB hipMalloc blocks the context's buffers are carved from) allocated since the context was
 * created, out[2] = microseconds of host time those allocations took, out[3] = arenas held now: a call that had to grow an
 * arena pays milliseconds on the host with its stream idle, and a caller timing calls can name that.
 * Exactness never depends on these numbers; tests use out[0] to see the fallback fire.   */
PCR_API int pcr_search_stats(pcr_ctx* ctx, int64_t out[4]);

/* Per-pass log of the LAST pcr_icp call on this context that ran the device-resident loop (grid index, gated): for every association
 * pass (Registration/main.py:107-154 is one) the duration of its tile launch -- of the whole pass when it ran as one launch --, of
 * its drain launch (0 for a one-launch pass) and the queries the tiles handed to the work queue (two-launch passes), from the
 * kernels' own 100-MHz timestamps.  *n_out = passes logged (<= PCR_ICP_MAX_LOG); at most max_n entries are written.
 * host_us (may be NULL): the call on the host, microseconds -- [0] set-up before the first launch, [1] enqueueing the first chunk
 * of passes, [2] waiting for the device (all synchronisations), [3] the whole loop, [4] by HIP events: from the start of the call to
 * behind the last kernel of the last chunk (in front of the read-back of the loop state), [5] the last wait alone.                 */
PCR_API int pcr_icp_pass_log(pcr_ctx* ctx, int max_n, double* tile_us, double* drain_us, int64_t* items, int* n_out, double host_us[6]);

/* Scheduling hint: shared != 0 tells the library that other contexts keep the same device busy while this one runs ICP
 * loops (a batch worker: Registration/main.py:190-216 spread over several streams).  The ICP pass then runs as two launches
 * instead of one whose idle waves wait in place for work -- latency for one pair bought with wave slots the other pairs
 * could use.  Results are identical bit for bit either way.  pcr_icp_batch sets it for its own worker contexts; without
 * the hint the library still switches by itself while it sees more than one loop of the process in flight. */
PCR_API int pcr_ctx_set_shared(pcr_ctx* ctx, int shared);
PCR_API int pcr_profile_read(pcr_ctx* ctx, double ms_out[4], int* passes_out);

#ifdef __cplusplus
}
#endif
#endif /* PCR_H */
