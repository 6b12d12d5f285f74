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
#include "pcr_sort.h"

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

// KEY = 0: the key is the integer value of h (h is a non-negative integer below 2^52 there: exact, and only its low bits need
// sorting), in 32 bits when they fit; KEY = 1: the bit pattern of h, which orders any non-negative binary64 exactly as the
// reference's float64 argsort does (huge grids: a tiny leaf on a large extent takes h beyond 2^53, even 2^64).
template <typename K, int KEY>
__global__ void voxel_keys_kernel(const pcr_pt* __restrict__ pts, long long n, double mnx, double mny, double mnz, double leaf,
                                  double Dx, double Dy, double* __restrict__ h_out /* by row id, may be null */,
                                  K* __restrict__ key_bits, unsigned int* __restrict__ vals) {
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
        key_bits[p.id] = KEY == 0 ? (K)(unsigned long long)h : (K)(unsigned long long)__double_as_longlong(h);
        vals[p.id] = (unsigned int)i;
    }
}

template <typename K>
struct head_flag {  // position i starts a group of equal keys
    const K* keys;
    __host__ __device__ bool operator()(unsigned int i) const { return i == 0u || keys[i] != keys[i - 1u]; }
};

// NumPy pairwise_sum over a[k] = coord(pts[perm[start + k]]), k in [0, n), for the three coordinates AT ONCE: every
// record is gathered once (not once per axis) and the 8 records of an unrolled step are requested together, so a dense
// voxel costs n/8 dependent memory round trips instead of 3n.  Per axis the additions and their order are exactly
// NumPy's (8 accumulators over blocks of <= 128, then the pairwise tree), so the centroids stay bitwise equal.
struct vox_xyz { double x, y, z; };   // voxel-major copy of the coordinates (written by voxel_gather_kernel)
struct coord_view {
    const vox_xyz* xyz;   // first record of the voxel's run
    __device__ vox_xyz at(unsigned int k) const { return xyz[k]; }
};

struct sum3 { double x, y, z; };

// NumPy's pairwise_sum leaf (n <= 128) by the EIGHT lanes that share a voxel: lane j runs accumulator r[j] of NumPy's unrolled
// loop (a[j] + a[8 + j] + a[16 + j] + ..., in that order), the accumulators meet in NumPy's tree ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
// by a three-step butterfly (IEEE addition commutes, so every lane gets the same bits), the n % 8 leftovers are added in order.
// Same additions in the same order as one thread doing it alone -- the centroids stay bitwise NumPy's -- in an eighth of the
// dependent steps, with coalesced 192-byte reads.
__device__ static sum3 pw_leaf8(const coord_view& a, unsigned int off, unsigned int n, int lane8) {
    sum3 res = {0.0, 0.0, 0.0};
    if (n < 8) {
        for (unsigned int i = 0; i < n; ++i) {
            const vox_xyz p = a.at(off + i);
            res.x += p.x; res.y += p.y; res.z += p.z;
        }
        return res;
    }
    vox_xyz r = a.at(off + lane8);
    unsigned int i = 8;
    const unsigned int lim = n - (n % 8);
    // four steps' records requested together (a leaf is <= 16 steps: one dependent memory round trip per step was the leaf's
    // whole time); the additions stay in NumPy's order
    for (; i + 32 <= lim; i += 32) {
        const vox_xyz p0 = a.at(off + i + lane8), p1 = a.at(off + i + 8 + lane8), p2 = a.at(off + i + 16 + lane8), p3 = a.at(off + i + 24 + lane8);
        r.x += p0.x; r.y += p0.y; r.z += p0.z;
        r.x += p1.x; r.y += p1.y; r.z += p1.z;
        r.x += p2.x; r.y += p2.y; r.z += p2.z;
        r.x += p3.x; r.y += p3.y; r.z += p3.z;
    }
    for (; i < lim; i += 8) {
        const vox_xyz p = a.at(off + i + lane8);
        r.x += p.x; r.y += p.y; r.z += p.z;
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {
        const double ox = __shfl_xor(r.x, d, 8), oy = __shfl_xor(r.y, d, 8), oz = __shfl_xor(r.z, d, 8);
        r.x += ox; r.y += oy; r.z += oz;
    }
    res.x = r.x; res.y = r.y; res.z = r.z;
    for (; i < n; ++i) {
        const vox_xyz p = a.at(off + i);
        res.x += p.x; res.y += p.y; res.z += p.z;
    }
    return res;
}

// (off, n) and the control flow below are the same in the eight lanes of a voxel.
// The explicit stack of the recursion lives in LDS, one per voxel group of eight lanes (written by its lane 0, read by all eight): as a
// local array it was 1 616 bytes of SCRATCH per lane -- 103 KB per wave for a path only voxels of more than 128 points ever take --,
// and a kernel that uses scratch pays for it at its first dispatch after kernels that do not.
struct pw_frame { unsigned int off, n; int state; int pad; sum3 left; };
constexpr int PW_DEPTH = 28;   // n < 2^32 halves to <= 128 in 25 steps
__device__ static sum3 numpy_pairwise_sum(const coord_view& a, unsigned int off0, unsigned int n, int lane8, pw_frame* st) {
    if (n <= 128) return pw_leaf8(a, off0, n, lane8);
    // explicit stack for: sum(off, n) = n <= 128 ? leaf : sum(off, n2) + sum(off + n2, n - n2), n2 = (n/2) rounded down to 8
    int sp = 0;
    if (lane8 == 0) st[0] = pw_frame{off0, n, 0, 0, {0.0, 0.0, 0.0}};
    sp = 1;
    sum3 ret = {0.0, 0.0, 0.0};
    while (sp > 0) {
        // (LDS operations of a wave execute in order, so lane 0's store is seen by the group's other lanes without a hardware wait;
        // the fences keep the COMPILER from holding a frame in registers across the store another lane made)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const pw_frame f = st[sp - 1];
        if (f.n <= 128) {
            ret = pw_leaf8(a, f.off, f.n, lane8);
            --sp;
            continue;
        }
        unsigned int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.state == 0) {
            if (lane8 == 0) { st[sp - 1].state = 1; st[sp] = pw_frame{f.off, n2, 0, 0, {0.0, 0.0, 0.0}}; }
            ++sp;
        } else if (f.state == 1) {
            if (lane8 == 0) { st[sp - 1].left = ret; st[sp - 1].state = 2; st[sp] = pw_frame{f.off + n2, f.n - n2, 0, 0, {0.0, 0.0, 0.0}}; }
            ++sp;
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

// Coordinates into voxel-major order, one thread per POINT: the random 32-byte reads through the permutation happen here, a
// million of them in flight at once; the emit kernel then streams every voxel's run.  (One thread per voxel gathering its own
// points through the permutation -- a chain of dependent scattered reads per thread -- took 132 us at 1 M points, 18 x the
// algorithmic traffic.)  Also zeroes the emit stage's list counter (stream-ordered in front of it).
__global__ void __launch_bounds__(256)
voxel_gather_kernel(const pcr_pt* __restrict__ pts, const unsigned int* __restrict__ perm, long long n, vox_xyz* __restrict__ out, unsigned int* __restrict__ big_count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *big_count = 0u;
    if (i >= n) return;
    const pcr_pt p = pts[perm[i]];
    out[i] = vox_xyz{p.x, p.y, p.z};
}

// A populous voxel must not be walked by one lane group alone (2 m leaves on a scan: tens of thousands of points in the voxels
// next to the sensor -- 0.7 ms of a 1.2 ms down-sample at 120 000 points, 2.4 ms at 1 M): the emit kernel appends voxels above
// these sizes to a list and voxel_emit_big_kernel gives each of them a block (NumPy's pairwise recursion: independent subtrees)
// or a wave (Open3D's running sum: an LDS-pipelined chain).
constexpr unsigned int VOX_BIG_PAIRWISE = 1024;   // mode 0
constexpr unsigned int VOX_BIG_RUNNING = 64;      // mode 2
__host__ __device__ static inline unsigned int voxel_big_threshold(int mode) { return mode == 0 ? VOX_BIG_PAIRWISE : (mode == 2 ? VOX_BIG_RUNNING : 0xffffffffu); }

__device__ static inline void voxel_store(unsigned int v, double ox, double oy, double oz, pcr_pt* __restrict__ out_pts, double* __restrict__ out_xyz) {
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

// EIGHT LANES per emitted voxel v in [0, n_groups - 1): the last group is never emitted (modes 0, 1); mode 2 emits all
__global__ void __launch_bounds__(256)
voxel_emit_kernel(const vox_xyz* __restrict__ xyz, const unsigned int* __restrict__ heads, const unsigned int* __restrict__ n_groups_p, long long n, int mode,
                  unsigned long long seed, pcr_pt* __restrict__ out_pts, double* __restrict__ out_xyz, unsigned int* __restrict__ big_count,
                  unsigned int* __restrict__ big_list) {
    __shared__ pw_frame s_stack[256 / 8][PW_DEPTH];
    pw_frame* const st = s_stack[threadIdx.x >> 3];
    const unsigned int ng = *n_groups_p;
    const int lane8 = threadIdx.x & 7;
    if (ng == 0) return;
    const unsigned int n_emit = mode == 2 ? ng : ng - 1;   // (only known on the device: a fixed grid strides over the voxels)
    const unsigned int big = voxel_big_threshold(mode);
    for (unsigned int v = (blockIdx.x * blockDim.x + threadIdx.x) >> 3; v < n_emit; v += (gridDim.x * blockDim.x) >> 3) {
    const unsigned int s = heads[v], e = v + 1 < ng ? heads[v + 1] : (unsigned int)n;
    const unsigned int cnt = e - s;
    if (cnt > big) {   // (group-uniform)
        if (lane8 == 0) big_list[atomicAdd(big_count, 1u)] = v;
        continue;
    }
    double ox, oy, oz;
    if (mode == 2) {
        // Open3D voxel_down_sample: running sum in input order, then one division.  The group's lanes fetch eight records at a
        // time (one coalesced read) and every lane runs the same chain over them through the cross-lane network -- one lane
        // fetching record after record paid a memory round trip per point.
        ox = oy = oz = 0.0;
        for (unsigned int k0 = 0; k0 < cnt; k0 += 8) {
            vox_xyz p = vox_xyz{0.0, 0.0, 0.0};
            if (k0 + lane8 < cnt) p = xyz[s + k0 + lane8];
            const unsigned int m = cnt - k0 < 8u ? cnt - k0 : 8u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double px = __shfl(p.x, j, 8), py = __shfl(p.y, j, 8), pz = __shfl(p.z, j, 8);
                if ((unsigned int)j < m) { ox += px; oy += py; oz += pz; }
            }
        }
        ox /= (double)cnt; oy /= (double)cnt; oz /= (double)cnt;
    } else if (mode == 0) {
        const coord_view a{xyz + s};
        const sum3 t = numpy_pairwise_sum(a, 0u, cnt, lane8, st);   // (cnt <= 1024: one inner-loop call of np.add.reduce, from the identity + 0.0)
        ox = (0.0 + t.x) / (double)cnt; oy = (0.0 + t.y) / (double)cnt; oz = (0.0 + t.z) / (double)cnt;
    } else {
        const unsigned int k = (unsigned int)(splitmix64(seed ^ ((unsigned long long)v * 0xD1B54A32D192ED03ull)) % cnt);
        const vox_xyz p = xyz[s + k];
        ox = p.x; oy = p.y; oz = p.z;
    }
    if (lane8 != 0) continue;
    voxel_store(v, ox, oy, oz, out_pts, out_xyz);
    }
}

// The listed voxels.
//   mode 0 (np.mean = NumPy's pairwise summation): one BLOCK per voxel.  np.add.reduce hands its inner loop at most
//   np.getbufsize() = 8192 elements at a time: the result is (((0 + pw(a[0:8192])) + pw(a[8192:16384])) + ...), pw = the
//   recursion below over one such chunk (checked against NumPy: a single recursion over the whole run differs from np.mean
//   from 8193 elements on).  Within a chunk, the recursion sum(off, n) = sum(off, n2) + sum(off + n2, n - n2),
//   n2 = (n / 2) rounded down to a multiple of 8, splits into 2^D independent subtrees as long as every node above them has more
//   than 128 elements (the smallest node of a depth is its leftmost: n -> n2 is monotone); each of the block's 32 lane groups sums
//   one subtree exactly as the single group would have, and the partial sums meet in the recursion's own tree order -- a balanced
//   binary tree over the 2^D subtrees, left + right at every node -- so the result is bit for bit the sequential recursion's.
//   mode 2 (Open3D's running sum in input order): one WAVE per voxel.  The chain of additions cannot be split, but it need not
//   wait for memory: the wave copies chunks of 256 points into its LDS slice (coalesced, the next chunk in flight while the
//   current one is summed) and lanes 0..2 run the x, y and z chains side by side from there.
constexpr int VOX_RUN_CHUNK = 256;   // points per LDS chunk of the running sum (6 KB; two per wave)
constexpr int VOX_SPLIT_DEPTH = 5;   // 2^5 = the block's 32 lane groups
constexpr unsigned int NPY_BUFSIZE = 8192;   // np.getbufsize(): elements per inner-loop call of a NumPy reduction
union voxel_big_lds {
    struct { pw_frame stack[256 / 8][PW_DEPTH]; sum3 sub[1 << VOX_SPLIT_DEPTH]; } pw;
    double run[4][2][3 * VOX_RUN_CHUNK];
};
__global__ void __launch_bounds__(256)
voxel_emit_big_kernel(const vox_xyz* __restrict__ xyz, const unsigned int* __restrict__ heads, const unsigned int* __restrict__ n_groups_p, long long n, int mode,
                      pcr_pt* __restrict__ out_pts, double* __restrict__ out_xyz, const unsigned int* __restrict__ big_count,
                      const unsigned int* __restrict__ big_list) {
    __shared__ voxel_big_lds L;
    const unsigned int nb = *big_count, ng = *n_groups_p;
    if (mode == 0) {
        const int grp = threadIdx.x >> 3, lane8 = threadIdx.x & 7;
        for (unsigned int b = blockIdx.x; b < nb; b += gridDim.x) {
            const unsigned int v = big_list[b];
            const unsigned int s = heads[v], e = v + 1 < ng ? heads[v + 1] : (unsigned int)n;
            const unsigned int cnt = e - s;
            const coord_view a{xyz + s};
            sum3 S = {0.0, 0.0, 0.0};   // (thread 0) np.add.reduce: the running value starts at the identity + 0.0
            for (unsigned int c0 = 0; c0 < cnt; c0 += NPY_BUFSIZE) {
                const unsigned int clen = cnt - c0 < NPY_BUFSIZE ? cnt - c0 : NPY_BUFSIZE;
                int D = 0;
                for (unsigned int m = clen; D < VOX_SPLIT_DEPTH && m > 128u; ++D) m = (m / 2u) & ~7u;   // m = smallest node of depth D
                // (the loop leaves D = number of levels whose nodes ALL split: the test looks at depth D's smallest node before going deeper)
                if (grp < (1 << D)) {
                    unsigned int off = c0, len = clen;
                    for (int l = D - 1; l >= 0; --l) {
                        const unsigned int n2 = (len / 2u) & ~7u;
                        if ((grp >> l) & 1) { off += n2; len -= n2; }
                        else len = n2;
                    }
                    const sum3 t = numpy_pairwise_sum(a, off, len, lane8, L.pw.stack[grp]);
                    if (lane8 == 0) L.pw.sub[grp] = t;
                }
                __syncthreads();
                for (int l = D - 1; l >= 0; --l) {   // level l: node i = node 2i + node 2i+1 of level l + 1
                    sum3 t = {0.0, 0.0, 0.0};
                    const bool act = threadIdx.x < (1u << l);
                    if (act) {
                        const sum3 a0 = L.pw.sub[2 * threadIdx.x], a1 = L.pw.sub[2 * threadIdx.x + 1];
                        t.x = a0.x + a1.x; t.y = a0.y + a1.y; t.z = a0.z + a1.z;
                    }
                    __syncthreads();
                    if (act) L.pw.sub[threadIdx.x] = t;
                    __syncthreads();
                }
                if (threadIdx.x == 0) {
                    const sum3 t = L.pw.sub[0];
                    S.x += t.x; S.y += t.y; S.z += t.z;
                }
                __syncthreads();
            }
            if (threadIdx.x == 0) voxel_store(v, S.x / (double)cnt, S.y / (double)cnt, S.z / (double)cnt, out_pts, out_xyz);
        }
        return;
    }
    // mode 2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* const buf0 = L.run[wave][0];
    double* const buf1 = L.run[wave][1];
    constexpr int PER = 3 * VOX_RUN_CHUNK / 64;   // doubles per lane and chunk
    for (unsigned int b = blockIdx.x * 4 + wave; b < nb; b += gridDim.x * 4) {
        const unsigned int v = big_list[b];
        const unsigned int s = heads[v], e = v + 1 < ng ? heads[v + 1] : (unsigned int)n;
        const unsigned int cnt = e - s;
        const double* const src = reinterpret_cast<const double*>(xyz + s);
        const unsigned int total = 3u * cnt;   // doubles
        double r[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) { const unsigned int i = lane + 64 * u; r[u] = i < total ? src[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < PER; ++u) buf0[lane + 64 * u] = r[u];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double acc = 0.0;   // lane 0: x, lane 1: y, lane 2: z
        const unsigned int n_chunks = (cnt + VOX_RUN_CHUNK - 1) / VOX_RUN_CHUNK;
        for (unsigned int c = 0; c < n_chunks; ++c) {
            const bool more = c + 1 < n_chunks;
            if (more) {
                const unsigned int base = 3u * VOX_RUN_CHUNK * (c + 1);
#pragma unroll
                for (int u = 0; u < PER; ++u) { const unsigned int i = base + lane + 64 * u; r[u] = i < total ? src[i] : 0.0; }
            }
            const double* const cur = (c & 1) ? buf1 : buf0;
            const unsigned int m = cnt - c * VOX_RUN_CHUNK < (unsigned int)VOX_RUN_CHUNK ? cnt - c * VOX_RUN_CHUNK : (unsigned int)VOX_RUN_CHUNK;
            if (lane < 3) {
                unsigned int k = 0;
                for (; k + 8 <= m; k += 8) {
                    double t[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) t[u] = cur[3 * (k + u) + lane];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += t[u];
                }
                for (; k < m; ++k) acc += cur[3 * k + lane];
            }
            if (more) {
                double* const nxt = (c & 1) ? buf0 : buf1;
#pragma unroll
                for (int u = 0; u < PER; ++u) nxt[lane + 64 * u] = r[u];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        acc /= (double)cnt;
        const double ox = __shfl(acc, 0, 64), oy = __shfl(acc, 1, 64), oz = __shfl(acc, 2, 64);
        if (lane == 0) voxel_store(v, ox, oy, oz, out_pts, out_xyz);
    }
}

struct voxel_work {
    double mn[3], mx[3], D[3];
    vox_xyz* xyz = nullptr;         // coordinates in voxel-major order
    unsigned int* heads = nullptr;  // group start positions
    unsigned int* n_groups = nullptr;
    unsigned int* big_list = nullptr;   // voxels too populous for a lane group (voxel_emit_big_kernel), <= n / 65 + 1 entries
    unsigned int* big_count = nullptr;
    int64_t n = 0;
};

// sort of (key, row) + group heads + voxel-major coordinates for key type K
template <typename K, int KEY>
static int voxel_groups(pcr_ctx* ctx, const pcr_cloud* c, double leaf, int end_bit, voxel_work* w) {
    const long long n = c->n;
    const int grid_n = (int)((n + 255) / 256);
    int rc;
    K *d_keys = nullptr, *d_keys2 = nullptr;
    unsigned int *d_vals = nullptr, *d_perm = nullptr;
    void *d_temp = nullptr, *d_temp2 = nullptr;
    size_t temp_bytes = 0, temp2 = 0;
    auto release = [&]() {
        if (d_temp) pcr_dev_free(ctx, d_temp, temp_bytes);
        if (d_temp2) pcr_dev_free(ctx, d_temp2, temp2);
        if (d_keys) pcr_dev_free(ctx, d_keys, sizeof(K) * n);
        if (d_keys2) pcr_dev_free(ctx, d_keys2, sizeof(K) * n);
        if (d_vals) pcr_dev_free(ctx, d_vals, sizeof(unsigned int) * n);
        if (d_perm) pcr_dev_free(ctx, d_perm, sizeof(unsigned int) * n);
    };
    if ((rc = pcr_dev_alloc(ctx, sizeof(K) * n, (void**)&d_keys)) || (rc = pcr_dev_alloc(ctx, sizeof(K) * n, (void**)&d_keys2)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_vals)) || (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * n, (void**)&d_perm)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(vox_xyz) * n, (void**)&w->xyz)) || (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * (n + 1), (void**)&w->heads)) ||
        (rc = pcr_dev_alloc(ctx, sizeof(unsigned int) * (n / 65 + 1), (void**)&w->big_list))) {
        release();
        return rc;
    }
    hipLaunchKernelGGL((voxel_keys_kernel<K, KEY>), dim3(grid_n), dim3(256), 0, ctx->stream, (const pcr_pt*)c->d, n, w->mn[0], w->mn[1], w->mn[2], leaf, w->D[0],
                       w->D[1], (double*)nullptr, d_keys, d_vals);
    hipError_t e = pcr_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, d_vals, d_perm, (size_t)n, (unsigned int)end_bit, ctx->stream);
    if (e == hipSuccess) rc = pcr_dev_alloc(ctx, temp_bytes, &d_temp);
    if (e == hipSuccess && rc == PCR_OK) e = pcr_sort_pairs(d_temp, temp_bytes, d_keys, d_keys2, d_vals, d_perm, (size_t)n, (unsigned int)end_bit, ctx->stream);
    if (e == hipSuccess && rc == PCR_OK) {
        w->big_count = ctx->d_counters + 49;
        hipLaunchKernelGGL(voxel_gather_kernel, dim3(grid_n), dim3(256), 0, ctx->stream, (const pcr_pt*)c->d, (const unsigned int*)d_perm, n, w->xyz, w->big_count);
        // group heads = positions whose key differs from the previous one: one fused flag + scan + scatter (rocprim::select
        // over a counting iterator with a computed flag), instead of a flag kernel, a scan and a scatter
        w->n_groups = ctx->d_counters + 48;
        const head_flag<K> flag_op{d_keys2};
        auto positions = rocprim::counting_iterator<unsigned int>(0u);
        auto flags = rocprim::make_transform_iterator(positions, flag_op);
        e = rocprim::select(nullptr, temp2, positions, flags, w->heads, w->n_groups, (size_t)n, ctx->stream);
        if (e == hipSuccess) rc = pcr_dev_alloc(ctx, temp2, &d_temp2);
        if (e == hipSuccess && rc == PCR_OK) e = rocprim::select(d_temp2, temp2, positions, flags, w->heads, w->n_groups, (size_t)n, ctx->stream);
    }
    if (e == hipSuccess) e = hipGetLastError();
    release();   // (stream-ordered with the launches above)
    if (rc) return rc;
    if (e != hipSuccess) { ctx->last_error = std::string("voxel filter: ") + hipGetErrorString(e); return PCR_E_HIP; }
    return PCR_OK;
}

// keys (+ sort + heads + voxel-major coordinates when need_groups) on a device cloud.  h_out_dev (by row id) optional.
static int voxel_prepare(pcr_ctx* ctx, const pcr_cloud* c, double leaf, double* h_out_dev, bool need_groups, voxel_work* w,
                         bool open3d_binning = false) {
    if (!(leaf > 0) || !std::isfinite(leaf)) return PCR_E_INVALID;
    const long long n = c->n;
    w->n = n;
    int rc = pcr_cloud_bbox(ctx, c, w->mn, w->mx);   // (the box remembered from the upload when there is one: no reduction, no read-back)
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
    if (!need_groups) {
        hipLaunchKernelGGL((voxel_keys_kernel<unsigned long long, 0>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const pcr_pt*)c->d, n, w->mn[0],
                           w->mn[1], w->mn[2], leaf, w->D[0], w->D[1], h_out_dev, (unsigned long long*)nullptr, (unsigned int*)nullptr);
        PCR_HIP(ctx, hipGetLastError());
        return PCR_OK;
    }
    // keys are non-negative integers bounded by the grid: sort only the bits they can occupy (a KITTI scan at 0.2 m needs
    // 25 of the 64: 32-bit keys, half the radix passes).  A grid so fine that h may leave the exactly representable integers
    // (bound >= 2^52) is sorted by the bit pattern of h, all 64 bits, like the reference's float64 argsort.
    double bound = 0.0;
    if (open3d_binning) {
        bound = w->D[0] * w->D[1] * w->D[2];
    } else {
        const double Hx = floor((w->mx[0] - w->mn[0]) / leaf), Hy = floor((w->mx[1] - w->mn[1]) / leaf), Hz = floor((w->mx[2] - w->mn[2]) / leaf);
        bound = (Hx + Hy * w->D[0]) + (Hz * w->D[0]) * w->D[1];
    }
    if (!(bound >= 0.0 && bound < 4503599627370496.0)) return voxel_groups<unsigned long long, 1>(ctx, c, leaf, 64, w);   // 2^52 (also NaN / inf)
    unsigned long long b = (unsigned long long)bound + 1ull;
    int end_bit = 1;
    while ((b >> end_bit) != 0ull) ++end_bit;
    if (end_bit <= 32) return voxel_groups<unsigned int, 0>(ctx, c, leaf, end_bit, w);
    return voxel_groups<unsigned long long, 0>(ctx, c, leaf, end_bit, w);
}

static void voxel_release(pcr_ctx* ctx, voxel_work* w) {
    if (w->xyz) pcr_dev_free(ctx, w->xyz, sizeof(vox_xyz) * w->n);
    if (w->heads) pcr_dev_free(ctx, w->heads, sizeof(unsigned int) * (w->n + 1));
    if (w->big_list) pcr_dev_free(ctx, w->big_list, sizeof(unsigned int) * (w->n / 65 + 1));
    w->xyz = nullptr;
    w->heads = nullptr;
    w->big_list = nullptr;
}

// ------------------------------------------------------------------------------------------------ many scans at once
// Open3D's voxel_down_sample (mode 2 above) of EVERY scan of a chunk in one pass -- the pair loop's prepare_dataset
// (Registration/main.py:33-36,197) preprocesses hundreds of scans of 20 000 - 120 000 points, and one scan at a time is a chain of
// ~20 launches of a few microseconds each, several per pair.  The key of a point is (scan << LB) | voxel index within the scan's own
// grid (same arithmetic as voxel_keys_kernel with the scan's own min bound and extents), ONE stable sort puts every scan's voxels
// in the order the per-scan filter produces, and the emit kernels above run unchanged over all of them.
struct down_scan_dev { unsigned int first_pt, n_pts; double mnx, mny, mnz, Dx, Dy; };

__global__ void __launch_bounds__(256)
scans_keys_kernel(const float* __restrict__ xyz, long long n, const down_scan_dev* __restrict__ scans, int n_scans, double leaf, int LB,
                  unsigned long long* __restrict__ keys, unsigned int* __restrict__ vals) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_scans - 1;   // the scan that holds point i: the last one that starts at or before it
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((long long)scans[mid].first_pt <= i) lo = mid;
        else hi = mid - 1;
    }
    const down_scan_dev S = scans[lo];
    const double px = (double)xyz[3 * i], py = (double)xyz[3 * i + 1], pz = (double)xyz[3 * i + 2];
    const double hx = floor((px - S.mnx) / leaf);
    const double hy = floor((py - S.mny) / leaf);
    const double hz = floor((pz - S.mnz) / leaf);
    const double h = (hx + hy * S.Dx) + (hz * S.Dx) * S.Dy;
    keys[i] = ((unsigned long long)lo << LB) | (unsigned long long)h;
    vals[i] = (unsigned int)i;
}

__global__ void __launch_bounds__(256)
scans_gather_kernel(const float* __restrict__ xyz, const unsigned int* __restrict__ perm, long long n, vox_xyz* __restrict__ out, unsigned int* __restrict__ big_count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *big_count = 0u;
    if (i >= n) return;
    const size_t j = perm[i];
    out[i] = vox_xyz{(double)xyz[3 * j], (double)xyz[3 * j + 1], (double)xyz[3 * j + 2]};
}

// first voxel of every scan and the scan of every voxel (a scan with points has voxels: every entry of scan_first is written)
__global__ void __launch_bounds__(256)
scans_first_kernel(const unsigned long long* __restrict__ keys_sorted, const unsigned int* __restrict__ heads, const unsigned int* __restrict__ ng_p, int LB, int n_scans,
                   unsigned int* __restrict__ scan_first, unsigned int* __restrict__ vsid) {
    const unsigned int ng = *ng_p;
    for (unsigned int v = blockIdx.x * blockDim.x + threadIdx.x; v < ng; v += gridDim.x * blockDim.x) {
        const unsigned int sid = (unsigned int)(keys_sorted[heads[v]] >> LB);
        vsid[v] = sid;
        if (v == 0 || (unsigned int)(keys_sorted[heads[v - 1]] >> LB) != sid) scan_first[sid] = v;
        if (v == 0) scan_first[n_scans] = ng;
    }
}

__global__ void __launch_bounds__(256)
scans_rows_kernel(pcr_pt* __restrict__ down, const unsigned int* __restrict__ vsid, const unsigned int* __restrict__ scan_first, const unsigned int* __restrict__ ng_p) {
    const unsigned int ng = *ng_p;
    for (unsigned int v = blockIdx.x * blockDim.x + threadIdx.x; v < ng; v += gridDim.x * blockDim.x)
        down[v].id = (long long)(v - scan_first[vsid[v]]);   // the record's row within its own scan (what the per-scan filter writes)
}

int pcr_voxel_downsample_scans(pcr_ctx* ctx, const float* d_xyz, int64_t n_pts, const pcr_down_scan* scans, int n_scans, double leaf, pcr_pt** down_out,
                               unsigned int** vsid_out, unsigned int** scan_first_out, unsigned int* scan_first_host, int64_t* ng_out) {
    if (!ctx || !d_xyz || !scans || n_scans < 1 || n_pts < 1 || n_pts > 0x7fffffffll || !(leaf > 0) || !std::isfinite(leaf)) return PCR_E_INVALID;
    if ((size_t)(n_scans + 1) * sizeof(unsigned int) > PCR_SMALL_D2H_BYTES) return PCR_E_UNSUPPORTED;
    *down_out = nullptr; *vsid_out = nullptr; *scan_first_out = nullptr;
    std::vector<down_scan_dev> h_sc((size_t)n_scans);
    double bound = 1.0;
    for (int s = 0; s < n_scans; ++s) {
        down_scan_dev& d = h_sc[(size_t)s];
        d.first_pt = scans[s].first_pt; d.n_pts = scans[s].n_pts;
        double mn[3], D[3];
        for (int k = 0; k < 3; ++k) {   // (Open3D's binning, as voxel_prepare does it)
            mn[k] = scans[s].mn[k] - leaf * 0.5;
            D[k] = floor((scans[s].mx[k] - mn[k]) / leaf) + 1.0;
        }
        const double b = D[0] * D[1] * D[2];
        if (!(b >= 1.0 && b < 1099511627776.0)) return PCR_E_UNSUPPORTED;   // 2^40 (also NaN / inf): such a scan goes through the per-scan path
        if (b > bound) bound = b;
        d.mnx = mn[0]; d.mny = mn[1]; d.mnz = mn[2]; d.Dx = D[0]; d.Dy = D[1];
    }
    int LB = 1;
    while (((unsigned long long)bound >> LB) != 0ull) ++LB;
    int SB = 1;
    while (((unsigned long long)(n_scans - 1) >> SB) != 0ull) ++SB;
    const int end_bit = LB + SB;
    const long long n = n_pts;
    const int grid_n = (int)((n + 255) / 256);
    pcr_dev_block b_sc(ctx), b_keys(ctx), b_keys2(ctx), b_vals(ctx), b_perm(ctx), b_xyz(ctx), b_heads(ctx), b_big(ctx), b_temp(ctx), b_temp2(ctx), b_full(ctx);
    int rc;
    if ((rc = b_sc.alloc(sizeof(down_scan_dev) * n_scans)) || (rc = b_keys.alloc(8 * (size_t)n)) || (rc = b_keys2.alloc(8 * (size_t)n)) || (rc = b_vals.alloc(4 * (size_t)n)) ||
        (rc = b_perm.alloc(4 * (size_t)n)) || (rc = b_xyz.alloc(sizeof(vox_xyz) * (size_t)n)) || (rc = b_heads.alloc(4 * (size_t)(n + 1))) ||
        (rc = b_big.alloc(4 * (size_t)(n / 65 + 1))))
        return rc;
    PCR_HIP(ctx, hipMemcpyAsync(b_sc.p, h_sc.data(), sizeof(down_scan_dev) * n_scans, hipMemcpyHostToDevice, ctx->stream));
    unsigned long long *d_keys = b_keys.as<unsigned long long>(), *d_keys2 = b_keys2.as<unsigned long long>();
    hipLaunchKernelGGL(scans_keys_kernel, dim3(grid_n), dim3(256), 0, ctx->stream, d_xyz, n, (const down_scan_dev*)b_sc.as<down_scan_dev>(), n_scans, leaf, LB, d_keys,
                       b_vals.as<unsigned int>());
    size_t temp_bytes = 0, temp2 = 0;
    PCR_HIP(ctx, pcr_sort_pairs(nullptr, temp_bytes, d_keys, d_keys2, b_vals.as<unsigned int>(), b_perm.as<unsigned int>(), (size_t)n, (unsigned int)end_bit, ctx->stream));
    if ((rc = b_temp.alloc(temp_bytes))) return rc;
    PCR_HIP(ctx, pcr_sort_pairs(b_temp.p, temp_bytes, d_keys, d_keys2, b_vals.as<unsigned int>(), b_perm.as<unsigned int>(), (size_t)n, (unsigned int)end_bit, ctx->stream));
    unsigned int* const big_count = ctx->d_counters + 49;
    unsigned int* const n_groups = ctx->d_counters + 48;
    hipLaunchKernelGGL(scans_gather_kernel, dim3(grid_n), dim3(256), 0, ctx->stream, d_xyz, (const unsigned int*)b_perm.as<unsigned int>(), n, b_xyz.as<vox_xyz>(), big_count);
    const head_flag<unsigned long long> flag_op{d_keys2};
    auto positions = rocprim::counting_iterator<unsigned int>(0u);
    auto flags = rocprim::make_transform_iterator(positions, flag_op);
    PCR_HIP(ctx, rocprim::select(nullptr, temp2, positions, flags, b_heads.as<unsigned int>(), n_groups, (size_t)n, ctx->stream));
    if ((rc = b_temp2.alloc(temp2))) return rc;
    PCR_HIP(ctx, rocprim::select(b_temp2.p, temp2, positions, flags, b_heads.as<unsigned int>(), n_groups, (size_t)n, ctx->stream));
    // every voxel of the chunk (the count is only known on the device: the emit kernels stride over it), written into a buffer of the
    // upper bound first
    if ((rc = b_full.alloc(sizeof(pcr_pt) * (size_t)n))) return rc;
    pcr_pt* const d_full = b_full.as<pcr_pt>();
    {
        long long blocks = (n * 8 + 255) / 256;
        if (blocks > 16ll * ctx->cu_count) blocks = 16ll * ctx->cu_count;
        hipLaunchKernelGGL(voxel_emit_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (const vox_xyz*)b_xyz.as<vox_xyz>(), (const unsigned int*)b_heads.as<unsigned int>(),
                           (const unsigned int*)n_groups, n, 2, 0ull, d_full, (double*)nullptr, big_count, b_big.as<unsigned int>());
        long long big_max = (n / (voxel_big_threshold(2) + 1) + 1 + 3) / 4;
        if (big_max > 8ll * ctx->cu_count) big_max = 8ll * ctx->cu_count;
        hipLaunchKernelGGL(voxel_emit_big_kernel, dim3((unsigned)big_max), dim3(256), 0, ctx->stream, (const vox_xyz*)b_xyz.as<vox_xyz>(),
                           (const unsigned int*)b_heads.as<unsigned int>(), (const unsigned int*)n_groups, n, 2, d_full, (double*)nullptr, (const unsigned int*)big_count,
                           (const unsigned int*)b_big.as<unsigned int>());
    }
    // (blocks that go back to the arena on every error path; handed to the caller at the end)
    pcr_dev_block b_first(ctx), b_vsid(ctx), b_down(ctx), b_vs(ctx);
    if ((rc = b_first.alloc(4 * (size_t)(n_scans + 1))) || (rc = b_vsid.alloc(4 * (size_t)n))) return rc;   // (the voxel count is at most the point count)
    unsigned int* const d_first = b_first.as<unsigned int>();
    const unsigned sgrid = (unsigned)(grid_n < 4 * ctx->cu_count ? grid_n : 4 * ctx->cu_count);
    hipLaunchKernelGGL(scans_first_kernel, dim3(sgrid), dim3(256), 0, ctx->stream, (const unsigned long long*)d_keys2, (const unsigned int*)b_heads.as<unsigned int>(),
                       (const unsigned int*)n_groups, LB, n_scans, d_first, b_vsid.as<unsigned int>());
    hipLaunchKernelGGL(scans_rows_kernel, dim3(sgrid), dim3(256), 0, ctx->stream, d_full, (const unsigned int*)b_vsid.as<unsigned int>(), (const unsigned int*)d_first,
                       (const unsigned int*)n_groups);
    PCR_HIP(ctx, hipGetLastError());
    if ((rc = pcr_d2h_small(ctx, scan_first_host, d_first, 4 * (size_t)(n_scans + 1)))) return rc;   // (synchronises)
    const int64_t ng = (int64_t)scan_first_host[n_scans];
    // cut to size
    const size_t ng1 = (size_t)(ng > 0 ? ng : 1);
    if ((rc = b_down.alloc(sizeof(pcr_pt) * ng1)) || (rc = b_vs.alloc(4 * ng1))) return rc;
    pcr_pt* const d_down = b_down.as<pcr_pt>();
    unsigned int* const d_vsid = b_vs.as<unsigned int>();
    PCR_HIP(ctx, hipMemcpyAsync(d_down, d_full, sizeof(pcr_pt) * (size_t)ng, hipMemcpyDeviceToDevice, ctx->stream));
    PCR_HIP(ctx, hipMemcpyAsync(d_vsid, b_vsid.p, 4 * (size_t)ng, hipMemcpyDeviceToDevice, ctx->stream));
    b_first.p = nullptr; b_down.p = nullptr; b_vs.p = nullptr;   // the caller's now
    *down_out = d_down; *vsid_out = d_vsid; *scan_first_out = d_first; *ng_out = ng;
    return PCR_OK;
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
        PCR_HIP(ctx, pcr_sync(ctx->stream));
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
        const int block = 256;   // 32 voxels per block (8 lanes each)
        long long blocks = (in->n * 8 + block - 1) / block;
        if (blocks > 16ll * ctx->cu_count) blocks = 16ll * ctx->cu_count;
        hipLaunchKernelGGL(voxel_emit_kernel, dim3((unsigned)blocks), dim3(block), 0, ctx->stream,
                           (const vox_xyz*)w.xyz, (const unsigned int*)w.heads, (const unsigned int*)w.n_groups,
                           (long long)in->n, mode, (unsigned long long)seed, out_pts, out_xyz_dev, w.big_count, w.big_list);
        if (mode != 1) {
            // what the lane groups left on the list (usually nothing at fine leaves: the blocks read the count and leave)
            long long big_max = in->n / (voxel_big_threshold(mode) + 1) + 1;
            if (mode == 2) big_max = (big_max + 3) / 4;
            if (big_max > 8ll * ctx->cu_count) big_max = 8ll * ctx->cu_count;
            hipLaunchKernelGGL(voxel_emit_big_kernel, dim3((unsigned)big_max), dim3(256), 0, ctx->stream, (const vox_xyz*)w.xyz, (const unsigned int*)w.heads,
                               (const unsigned int*)w.n_groups, (long long)in->n, mode, out_pts, out_xyz_dev, (const unsigned int*)w.big_count,
                               (const unsigned int*)w.big_list);
        }
        PCR_HIP(ctx, hipGetLastError());
    }
    unsigned int ng = 0;
    { const int rc_n = pcr_d2h_small(ctx, &ng, w.n_groups, sizeof(unsigned int)); if (rc_n) return rc_n; }   // (synchronises)
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
        PCR_HIP(ctx, pcr_sync(ctx->stream));
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
