// voxel_filter(point_cloud, leaf_size, type) -- Pca_and_Voxel_filter/voxel_filter.py:10-68.
//
// Semantics kept bit for bit:
//   * D = (max - min) // leaf with NumPy/Python float floor-division (fmod based), no +1
//     (voxel_filter.py:22-24), so the max-edge cells alias into their neighbours;
//   * h = hx + hy*Dx + hz*Dx*Dy in binary64, hx = floor((x - min_x) / leaf) with a true
//     division (voxel_filter.py:30-33), evaluated left to right without FMA;
//   * stable sort by h: inside a voxel the points keep their input order (voxel_filter.py:36);
//   * a group is emitted only when the next key arrives, so the group with the largest h
//     is never emitted: output rows = occupied voxels - 1 (voxel_filter.py:42-51);
//   * centroid = np.mean over the group = NumPy's pairwise summation (8 accumulators,
//     blocks of 128) divided by the count -- restated here so the result is bitwise equal.
// Pipeline on the device: bbox -> keys -> radix sort of (integer key, index) over the occupied key bits only ->
// group heads (one rocprim::select with a computed flag) -> one thread per emitted voxel; one host sync at the end.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#include <rocprim/rocprim.hpp>
#include "pcr_internal.h"

// NumPy npy_divmod floor-division for doubles
static double npy_floor_divide(double a, double b) {
    if (b == 0.0) return a / b;
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0; }
    }
    double floordiv;
    if (div != 0.0) {
        floordiv = floor(div);
        if (div - floordiv > 0.5) floordiv += 1.0;
    } else {
        floordiv = copysign(0.0, a / b);
    }
    return floordiv;
}

__global__ void voxel_keys_kernel(const pcr_pt* __restrict__ pts, long long n, double mnx, double mny, double mnz, double leaf,
                                  double Dx, double Dy, double* __restrict__ h_out /* by row id, may be null */,
                                  unsigned long long* __restrict__ key_bits, unsigned int* __restrict__ vals) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const pcr_pt p = pts[i];
    const double hx = floor((p.x - mnx) / leaf);
    const double hy = floor((p.y - mny) / leaf);
    const double hz = floor((p.z - mnz) / leaf);
    const double h = (hx + hy * Dx) + (hz * Dx) * Dy;
    if (h_out) h_out[p.id] = h;
    if (key_bits) {
        // records may be device-reordered: sort position = row id, so that ties keep INPUT order
        key_bits[p.id] = (unsigned long long)h;  // h is a non-negative integer below 2^53: exact, and only its low bits need sorting
        vals[p.id] = (unsigned int)i;
    }
}

struct head_flag {  // position i starts a group of equal keys
    const unsigned long long* keys;
    __host__ __device__ bool operator()(unsigned int i) const { return i == 0u || keys[i] != keys[i - 1u]; }
};

// NumPy pairwise_sum over a[k] = coord(pts[perm[start + k]]), k in [0, n), for the three coordinates AT ONCE: every
// record is gathered once (not once per axis) and the 8 records of an unrolled step are requested together, so a dense
// voxel costs n/8 dependent memory round trips instead of 3n.  Per axis the additions and their order are exactly
// NumPy's (8 accumulators over blocks of <= 128, then the pairwise tree), so the centroids stay bitwise equal.
struct coord_view {
    const pcr_pt* pts;
    const unsigned int* perm;
    unsigned int start;
    __device__ pcr_pt at(unsigned int k) const { return pts[perm[start + k]]; }
};

struct sum3 { double x, y, z; };

__device__ static sum3 pw_leaf(const coord_view& a, unsigned int off, unsigned int n) {
    if (n < 8) {
        sum3 res = {0.0, 0.0, 0.0};
        for (unsigned int i = 0; i < n; ++i) {
            const pcr_pt p = a.at(off + i);
            res.x += p.x; res.y += p.y; res.z += p.z;
        }
        return res;
    }
    double rx[8], ry[8], rz[8];
    {
        pcr_pt p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = a.at(off + j);
#pragma unroll
        for (int j = 0; j < 8; ++j) { rx[j] = p[j].x; ry[j] = p[j].y; rz[j] = p[j].z; }
    }
    unsigned int i = 8;
    for (; i < n - (n % 8); i += 8) {
        pcr_pt p[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = a.at(off + i + j);
#pragma unroll
        for (int j = 0; j < 8; ++j) { rx[j] += p[j].x; ry[j] += p[j].y; rz[j] += p[j].z; }
    }
    sum3 res;
    res.x = ((rx[0] + rx[1]) + (rx[2] + rx[3])) + ((rx[4] + rx[5]) + (rx[6] + rx[7]));
    res.y = ((ry[0] + ry[1]) + (ry[2] + ry[3])) + ((ry[4] + ry[5]) + (ry[6] + ry[7]));
    res.z = ((rz[0] + rz[1]) + (rz[2] + rz[3])) + ((rz[4] + rz[5]) + (rz[6] + rz[7]));
    for (; i < n; ++i) {
        const pcr_pt p = a.at(off + i);
        res.x += p.x; res.y += p.y; res.z += p.z;
    }
    return res;
}

__device__ static sum3 numpy_pairwise_sum(const coord_view& a, unsigned int n) {
    if (n <= 128) return pw_leaf(a, 0, n);
    // explicit stack for: sum(off, n) = n <= 128 ? leaf : sum(off, n2) + sum(off + n2, n - n2), n2 = (n/2) rounded down to 8
    struct frame { unsigned int off, n; int state; sum3 left; };
    frame st[40];
    int sp = 0;
    st[sp++] = {0u, n, 0, {0.0, 0.0, 0.0}};
    sum3 ret = {0.0, 0.0, 0.0};
    while (sp > 0) {
        frame& f = st[sp - 1];
        if (f.n <= 128) {
            ret = pw_leaf(a, f.off, f.n);
            --sp;
            continue;
        }
        unsigned int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.state == 0) {
            f.state = 1;
            st[sp++] = {f.off, n2, 0, {0.0, 0.0, 0.0}};
        } else if (f.state == 1) {
            f.left = ret;
            f.state = 2;
            st[sp++] = {f.off + n2, f.n - n2, 0, {0.0, 0.0, 0.0}};
        } else {
            ret.x = f.left.x + ret.x; ret.y = f.left.y + ret.y; ret.z = f.left.z + ret.z;
            --sp;
        }
    }
    return ret;
}

__device__ static inline unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// one thread per emitted voxel v in [0, n_groups - 1): the last group is never emitted (modes 0, 1); mode 2 emits all
__global__ void voxel_emit_kernel(const pcr_pt* __restrict__ pts, const unsigned int* __restrict__ perm, const unsigned int* __restrict__ heads,
                                  const unsigned int* __restrict__ n_groups_p, long long n, int mode, unsigned long long seed,
                                  pcr_pt* __restrict__ out_pts, double* __restrict__ out_xyz) {
    const unsigned int ng = *n_groups_p;
    const unsigned int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (ng == 0 || v >= (mode == 2 ? ng : ng - 1)) return;
    const unsigned int s = heads[v], e = v + 1 < ng ? heads[v + 1] : (unsigned int)n;
    const unsigned int cnt = e - s;
    double ox, oy, oz;
    if (mode == 2) {
        // Open3D voxel_down_sample: running sum in input order, then one division
        ox = oy = oz = 0.0;
        for (unsigned int k = 0; k < cnt; ++k) {
            const pcr_pt p = pts[perm[s + k]];
            ox += p.x; oy += p.y; oz += p.z;
        }
        ox /= (double)cnt; oy /= (double)cnt; oz /= (double)cnt;
    } else if (mode == 0) {
        const coord_view a{pts, perm, s};
        const sum3 t = numpy_pairwise_sum(a, cnt);
        ox = t.x / (double)cnt; oy = t.y / (double)cnt; oz = t.z / (double)cnt;
    } else {
        const unsigned int k = (unsigned int)(splitmix64(seed ^ ((unsigned long long)v * 0xD1B54A32D192ED03ull)) % cnt);
        const pcr_pt p = pts[perm[s + k]];
        ox = p.x; oy = p.y; oz = p.z;
    }
    if (out_pts) {
        pcr_pt o;
        o.x = ox; o.y = oy; o.z = oz;
        o.id = v;
        out_pts[v] = o;
    }
    if (out_xyz) {
        out_xyz[3 * (size_t)v + 0] = ox;
        out_xyz[3 * (size_t)v + 1] = oy;
        out_xyz[3 * (size_t)v + 2] = oz;
    }
}

struct voxel_work {
    double mn[3], mx[3], D[3];
    unsigned int* perm = nullptr;   // sorted position -> device record index
    unsigned int* heads = nullptr;  // group start positions
    unsigned int* n_groups = nullptr;
    int64_t n = 0;
};

// keys + sort + heads on a device cloud.  h_out_dev (by row id) optional.
static int voxel_prepare(pcr_ctx* ctx, const pcr_cloud* c, double leaf, double* h_out_dev, bool need_groups, voxel_work* w,
                         bool open3d_binning = false) {
    if (!(leaf > 0) || !std::isfinite(leaf)) return PCR_E_INVALID;
    const long long n = c->n;
    w->n = n;
    int rc = pcr_bbox(ctx, c->d, n, w->mn, w->mx);
    if (rc) return rc;
    if (open3d_binning) {
        // Open3D: voxel_min_bound = min - voxel_size/2, index = floor((p - voxel_min_bound) / voxel_size); every cell distinct
        for (int k = 0; k < 3; ++k) {
            w->mn[k] -= leaf * 0.5;
            w->D[k] = floor((w->mx[k] - w->mn[k]) / leaf) + 1.0;
        }
        if (w->D[0] * w->D[1] * w->D[2] >= 9007199254740992.0) { ctx->last_error = "voxel_size is too small"; return PCR_E_INVALID; }
    } else
    for (int k = 0; k < 3; ++k) w->D[k] = npy_floor_divide(w->mx[k] - w->mn[k], leaf);
    const int block = 256;
    const int grid_n = (int)((n + block - 1) / block);
    unsigned long long *d_keys = nullptr, *d_keys2 = nullptr;
    unsigned int* d_vals = nullptr;
    if (need_groups) {
        if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned long long) * n, (void**)&d_keys))) return rc;
        if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned long long) * n, (void**)&d_keys2))) return rc;
        if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_vals))) return rc;
        if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&w->perm))) return rc;
    }
    hipLaunchKernelGGL(voxel_keys_kernel, dim3(grid_n), dim3(block), 0, ctx->stream, (const pcr_pt*)c->d, n, w->mn[0], w->mn[1], w->mn[2],
                       leaf, w->D[0], w->D[1], h_out_dev, d_keys, d_vals);
    PCR_HIP(ctx, hipGetLastError());
    if (!need_groups) return PCR_OK;
    // keys are non-negative integers bounded by the grid: sort only the bits they can occupy (a KITTI scan at 0.2 m needs
    // 25 of the 64: half the radix passes)
    int end_bit = 63;
    {
        double bound = 0.0;
        if (open3d_binning) {
            bound = w->D[0] * w->D[1] * w->D[2];
        } else {
            const double Hx = floor((w->mx[0] - w->mn[0]) / leaf), Hy = floor((w->mx[1] - w->mn[1]) / leaf), Hz = floor((w->mx[2] - w->mn[2]) / leaf);
            bound = (Hx + Hy * w->D[0]) + (Hz * w->D[0]) * w->D[1];
        }
        if (bound >= 0.0 && bound < 4503599627370496.0) {  // 2^52
            unsigned long long b = (unsigned long long)bound + 1ull;
            end_bit = 1;
            while ((b >> end_bit) != 0ull) ++end_bit;
        }
    }
    size_t temp_bytes = 0;
    PCR_HIP(ctx, rocprim::radix_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_vals, w->perm, (size_t)n, 0, end_bit, ctx->stream));
    void* d_temp = nullptr;
    if ((rc = pcr_dev_alloc(ctx, temp_bytes, &d_temp))) return rc;
    PCR_HIP(ctx, rocprim::radix_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_vals, w->perm, (size_t)n, 0, end_bit, ctx->stream));
    if ((rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * (n + 1), (void**)&w->heads))) return rc;
    w->n_groups = ctx->d_counters + 48;
    // group heads = positions whose key differs from the previous one: one fused flag + scan + scatter (rocprim::select
    // over a counting iterator with a computed flag), instead of a flag kernel, a scan and a scatter
    const head_flag flag_op{d_keys2};
    auto positions = rocprim::counting_iterator<unsigned int>(0u);
    auto flags = rocprim::make_transform_iterator(positions, flag_op);
    size_t temp2 = 0;
    PCR_HIP(ctx, rocprim::select(nullptr, temp2, positions, flags, w->heads, w->n_groups, (size_t)n, ctx->stream));
    void* d_temp2 = nullptr;
    if ((rc = pcr_dev_alloc(ctx, temp2, &d_temp2))) return rc;
    PCR_HIP(ctx, rocprim::select(d_temp2, temp2, positions, flags, w->heads, w->n_groups, (size_t)n, ctx->stream));
    PCR_HIP(ctx, hipGetLastError());
    pcr_dev_free(ctx, d_temp, temp_bytes);
    pcr_dev_free(ctx, d_temp2, temp2);
    pcr_dev_free(ctx, d_keys, sizeof(unsigned long long) * n);
    pcr_dev_free(ctx, d_keys2, sizeof(unsigned long long) * n);
    pcr_dev_free(ctx, d_vals, sizeof(unsigned int) * n);
    return PCR_OK;
}

static void voxel_release(pcr_ctx* ctx, voxel_work* w) {
    if (w->perm) pcr_dev_free(ctx, w->perm, sizeof(unsigned int) * w->n);
    if (w->heads) pcr_dev_free(ctx, w->heads, sizeof(unsigned int) * (w->n + 1));
    w->perm = w->heads = nullptr;
}

extern "C" {

int pcr_voxel_keys(pcr_ctx* ctx, const double* xyz, int64_t n, double leaf, double* h_out, double D_out[3]) {
    if (!ctx || !xyz || !h_out) return PCR_E_INVALID;
    if (n <= 0) return PCR_E_EMPTY;
    pcr_cloud* c = nullptr;
    int rc = pcr_cloud_upload_f64(ctx, xyz, n, 3, &c);
    if (rc) return rc;
    double* d_h = nullptr;
    rc = pcr_dev_alloc(ctx, sizeof(double) * n, (void**)&d_h);
    voxel_work w;
    if (rc == PCR_OK) rc = voxel_prepare(ctx, c, leaf, d_h, false, &w);
    if (rc == PCR_OK) {
        PCR_HIP(ctx, hipMemcpyAsync(h_out, d_h, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (D_out)
            for (int k = 0; k < 3; ++k) D_out[k] = w.D[k];
    }
    if (d_h) pcr_dev_free(ctx, d_h, sizeof(double) * n);
    pcr_cloud_free(ctx, c);
    return rc;
}

static int voxel_filter_impl(pcr_ctx* ctx, const pcr_cloud* in, double leaf, int mode, uint64_t seed, pcr_pt* out_pts, double* out_xyz_dev,
                             int64_t* n_out) {
    if (mode != 0 && mode != 1 && mode != 2) return PCR_E_INVALID;
    voxel_work w;
    int rc = voxel_prepare(ctx, in, leaf, nullptr, true, &w, mode == 2);
    if (rc) { voxel_release(ctx, &w); return rc; }
    // the number of groups is only known on the device: launch for the upper bound (threads past the last emitted voxel
    // leave at once) and read the count once everything is queued -- no host round trip in the middle of the chain
    {
        const int block = 128;
        hipLaunchKernelGGL(voxel_emit_kernel, dim3((unsigned)((in->n + block - 1) / block)), dim3(block), 0, ctx->stream,
                           (const pcr_pt*)in->d, (const unsigned int*)w.perm, (const unsigned int*)w.heads, (const unsigned int*)w.n_groups,
                           (long long)in->n, mode, (unsigned long long)seed, out_pts, out_xyz_dev);
        PCR_HIP(ctx, hipGetLastError());
    }
    unsigned int ng = 0;
    PCR_HIP(ctx, hipMemcpyAsync(&ng, w.n_groups, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
    PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t rows = mode == 2 ? (int64_t)ng : (ng > 0 ? (int64_t)ng - 1 : 0);
    *n_out = rows;
    voxel_release(ctx, &w);
    return PCR_OK;
}

int pcr_voxel_filter(pcr_ctx* ctx, const double* xyz, int64_t n, double leaf, int mode, uint64_t seed, double* out_xyz, int64_t* n_out) {
    if (!ctx || !xyz || !out_xyz || !n_out) return PCR_E_INVALID;
    if (n <= 0) return PCR_E_EMPTY;
    pcr_cloud* c = nullptr;
    int rc = pcr_cloud_upload_f64(ctx, xyz, n, 3, &c);
    if (rc) return rc;
    double* d_out = nullptr;
    rc = pcr_dev_alloc(ctx, sizeof(double) * 3 * n, (void**)&d_out);
    if (rc == PCR_OK) rc = voxel_filter_impl(ctx, c, leaf, mode, seed, nullptr, d_out, n_out);
    if (rc == PCR_OK && *n_out > 0) {
        PCR_HIP(ctx, hipMemcpyAsync(out_xyz, d_out, sizeof(double) * 3 * (*n_out), hipMemcpyDeviceToHost, ctx->stream));
        PCR_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (d_out) pcr_dev_free(ctx, d_out, sizeof(double) * 3 * n);
    pcr_cloud_free(ctx, c);
    return rc;
}

int pcr_voxel_filter_cloud(pcr_ctx* ctx, const pcr_cloud* in, double leaf, int mode, uint64_t seed, pcr_cloud** out) {
    if (!ctx || !in || !out) return PCR_E_INVALID;
    *out = nullptr;
    if (in->n <= 0) return PCR_E_EMPTY;
    hipSetDevice(ctx->device);
    pcr_pt* d_pts = nullptr;
    int rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * in->n, (void**)&d_pts);
    if (rc) return rc;
    int64_t rows = 0;
    rc = voxel_filter_impl(ctx, in, leaf, mode, seed, d_pts, nullptr, &rows);
    if (rc != PCR_OK || rows == 0) {
        pcr_dev_free(ctx, d_pts, sizeof(pcr_pt) * in->n);
        return rc != PCR_OK ? rc : PCR_E_EMPTY;  // a single occupied voxel filters to nothing (voxel_filter.py:42-51)
    }
    // shrink to fit
    pcr_cloud* c = new pcr_cloud();
    c->n = rows;
    rc = pcr_dev_alloc(ctx, sizeof(pcr_pt) * rows, (void**)&c->d);
    if (rc) { delete c; pcr_dev_free(ctx, d_pts, sizeof(pcr_pt) * in->n); return rc; }
    PCR_HIP(ctx, hipMemcpyAsync(c->d, d_pts, sizeof(pcr_pt) * rows, hipMemcpyDeviceToDevice, ctx->stream));
    pcr_dev_free(ctx, d_pts, sizeof(pcr_pt) * in->n);
    *out = c;
    return PCR_OK;
}

}  // extern "C"
