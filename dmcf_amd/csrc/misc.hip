// Device-wide prefix scans, reduce_subarrays_sum and the library bookkeeping entry points.
#include "common.h"

namespace dmcf {

thread_local int g_last_hip_error = 0;

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;  // 2048 elements per workgroup

template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T o = __shfl_up(v, d, kWave);
        if (lane_id() >= d) v += o;
    }
    return v;
}

// block-wide exclusive scan of one value per thread; returns the exclusive prefix, total in *total
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* total, T* smem /* >= 4 entries */) {
    const int wave = threadIdx.x >> 6;
    const T inc = wave_inclusive_scan(v);
    if (lane_id() == kWave - 1) smem[wave] = inc;
    __syncthreads();
    T carry = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kScanThreads / kWave; ++w) {
        const T s = smem[w];
        if (w < wave) carry += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return carry + inc - v;
}

template <typename TIn, typename TAcc>
__global__ __launch_bounds__(kScanThreads) void scan_block_sums(const TIn* __restrict__ in, int64_t n,
                                                                TAcc* __restrict__ sums) {
    __shared__ TAcc smem[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    TAcc s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) s += (TAcc)in[base + k];
    TAcc total;
    block_exclusive_scan<TAcc>(s, &total, smem);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of the per-block sums in place
template <typename TAcc>
__global__ __launch_bounds__(kScanThreads) void scan_sums_inplace(TAcc* __restrict__ sums, int64_t nb) {
    __shared__ TAcc smem[4];
    TAcc carry = 0;
    for (int64_t b0 = 0; b0 < nb; b0 += kScanThreads) {
        const int64_t b = b0 + threadIdx.x;
        const TAcc v = b < nb ? sums[b] : (TAcc)0;
        TAcc total;
        const TAcc ex = block_exclusive_scan<TAcc>(v, &total, smem);
        if (b < nb) sums[b] = carry + ex;
        carry += total;
    }
}

// out[i] = exclusive prefix; if WRITE_TOTAL also out[n] = grand total
template <typename TIn, typename TAcc, bool WRITE_TOTAL>
__global__ __launch_bounds__(kScanThreads) void scan_block_apply(const TIn* __restrict__ in, int64_t n,
                                                                 const TAcc* __restrict__ sums,
                                                                 TAcc* __restrict__ out) {
    __shared__ TAcc smem[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
    TAcc v[kScanItems];
    TAcc s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = base + k < n ? (TAcc)in[base + k] : (TAcc)0;
        s += v[k];
    }
    TAcc total;
    TAcc run = block_exclusive_scan<TAcc>(s, &total, smem) + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
        if (WRITE_TOTAL && base + k == n - 1) out[n] = run;
    }
}

size_t scan_tmp_bytes(int64_t n) {
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    return align_up((size_t)(nb + 1) * sizeof(int64_t), 256);
}

template <typename TIn, typename TAcc, bool WRITE_TOTAL>
static int scan_impl(const TIn* in, TAcc* out, int64_t n, void* tmp, size_t tmp_bytes, hipStream_t stream) {
    if (n < 0) return DMCF_EINVAL;
    if (n == 0) {
        if (WRITE_TOTAL) {
            if (hipMemsetAsync(out, 0, sizeof(TAcc), stream) != hipSuccess) return DMCF_ELAUNCH;
        }
        return DMCF_OK;
    }
    if (tmp_bytes < scan_tmp_bytes(n)) return DMCF_EWORKSPACE;
    const int64_t nb = (n + kScanTile - 1) / kScanTile;
    TAcc* sums = (TAcc*)tmp;
    hipLaunchKernelGGL((scan_block_sums<TIn, TAcc>), dim3((unsigned)nb), dim3(kScanThreads), 0, stream, in, n, sums);
    hipLaunchKernelGGL((scan_sums_inplace<TAcc>), dim3(1), dim3(kScanThreads), 0, stream, sums, nb);
    hipLaunchKernelGGL((scan_block_apply<TIn, TAcc, WRITE_TOTAL>), dim3((unsigned)nb), dim3(kScanThreads), 0,
                       stream, in, n, sums, out);
    return check_launch();
}

int scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, void* tmp, size_t tmp_bytes,
                       hipStream_t stream) {
    return scan_impl<uint32_t, uint32_t, false>(in, out, n, tmp, tmp_bytes, stream);
}

int scan_counts_to_row_splits(const int32_t* counts, int64_t* row_splits, int64_t n, void* tmp,
                              size_t tmp_bytes, hipStream_t stream) {
    return scan_impl<int32_t, int64_t, true>(counts, row_splits, n, tmp, tmp_bytes, stream);
}

// one wave per row for long rows would be overkill here: rows are neighbour lists (tens to a few
// thousand entries) and the op runs once per step (models/pbf_model.py:450-453); a 16-lane group per
// row keeps the loads coalesced inside a row.
// out = x W (+ bias) (+ residual) for the networks' Dense layers (dmcf_dense_forward): n ~ 1e6 rows, k and m of a few tens --
// 430 MB of rows for 2 GFLOP, memory bound; the library GEMMs took 120 - 240 us for these tall-skinny products.  A wavefront
// keeps the whole filter as B fragments of v_mfma_f32_16x16x4_f32 in registers (k / 4 x NT of them) and walks tiles of 16 rows:
// lane (r, q) loads x[row r][16 j + 4 q ..] in 16-byte pieces (64 contiguous bytes per row and instruction), the products of
// piece (j, i) pair x[.][16 j + 4 q + i] with W[16 j + 4 q + i][.] -- any bijection of the k index serves, this one makes the
// loads wide.  (One row per thread with the filter in scalar registers was tried first: every load instruction of a wave then
// touches 64 cache lines for 16 bytes each, 230 us for 32 -> 32.)  Fixed order of additions: deterministic.
typedef float dense_f32x4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(256) void dense_rows(const float* __restrict__ x, int64_t n, int k, const float* __restrict__ W, int m,
                                                  const float* __restrict__ bias, const float* __restrict__ residual,
                                                  float* __restrict__ out, int64_t ntiles) {
    const int lane = lane_id(), r = lane & 15, q = lane >> 4;
    const int kj = (k + 15) >> 4;
    float wf[4][4][NT];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int kk = 16 * j + 4 * q + i, col = 16 * t + r;
                wf[j][i][t] = (kk < k && col < m) ? W[(size_t)kk * m + col] : 0.0f;
            }
    float bs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bs[t] = (bias && 16 * t + r < m) ? bias[16 * t + r] : 0.0f;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t tile = wave; tile < ntiles; tile += nwaves) {
        const int64_t row = tile * 16 + r;
        dense_f32x4 a[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = (dense_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (j < kj && row < n && 16 * j + 4 * q < k) a[j] = *(const dense_f32x4*)(x + row * k + 16 * j + 4 * q);
        }
        dense_f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (dense_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < kj) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][i], wf[j][i][t], acc[t], 0, 0, 0);
            }
        }
        // D layout: lane (r, q) holds rows 4 q + rr, column 16 t + r
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t orow = tile * 16 + 4 * q + rr;
            if (orow >= n) continue;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = 16 * t + r;
                if (col >= m) continue;
                float v = acc[t][rr] + bs[t];
                if (residual) v += residual[orow * m + col];
                out[orow * m + col] = v;
            }
        }
    }
}

// Axis-aligned bounding box of [n, 3] points (dmcf_points_aabb): per-block extrema without atomics, then one wavefront over
// the blocks.  A NaN coordinate makes both bounds of its axis NaN (what torch.aminmax / tf.reduce_min return).
constexpr int kAabbBlocks = 1024;

__global__ __launch_bounds__(256) void aabb_partial(const float* __restrict__ pts, int64_t n, float* __restrict__ part) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool bad[3] = {false, false, false};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
            bad[a] |= v != v;
        }
    }
    __shared__ float red[3][3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float b = bad[a] ? 1.0f : 0.0f;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, kWave));
            b = fmaxf(b, __shfl_xor(b, d, kWave));
        }
        if (lane_id() == 0) {
            red[0][a][threadIdx.x >> 6] = mn[a];
            red[1][a][threadIdx.x >> 6] = mx[a];
            red[2][a][threadIdx.x >> 6] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float* o = part + (size_t)blockIdx.x * 8;
        const bool b = fmaxf(fmaxf(red[2][a][0], red[2][a][1]), fmaxf(red[2][a][2], red[2][a][3])) > 0.0f;
        o[a] = b ? NAN : fminf(fminf(red[0][a][0], red[0][a][1]), fminf(red[0][a][2], red[0][a][3]));
        o[3 + a] = b ? NAN : fmaxf(fmaxf(red[1][a][0], red[1][a][1]), fmaxf(red[1][a][2], red[1][a][3]));
    }
}

__global__ void aabb_finish(const float* __restrict__ part, int nblocks, float* __restrict__ out) {
    const int lane = lane_id();
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = INFINITY, mx = -INFINITY, b = 0.0f;
        for (int k = lane; k < nblocks; k += kWave) {
            const float lo = part[(size_t)k * 8 + a], hi = part[(size_t)k * 8 + 3 + a];
            b = fmaxf(b, lo != lo ? 1.0f : 0.0f);
            mn = fminf(mn, lo);
            mx = fmaxf(mx, hi);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, d, kWave));
            mx = fmaxf(mx, __shfl_xor(mx, d, kWave));
            b = fmaxf(b, __shfl_xor(b, d, kWave));
        }
        if (lane == 0) {
            out[a] = b > 0.0f ? NAN : mn;
            out[3 + a] = b > 0.0f ? NAN : mx;
        }
    }
}

__global__ __launch_bounds__(256) void reduce_subarrays_sum_kernel(const float* __restrict__ values,
                                                                   const int64_t* __restrict__ row_splits,
                                                                   int64_t n_rows, float* __restrict__ out) {
    const int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    if (row >= n_rows) return;
    const int64_t b = row_splits[row], e = row_splits[row + 1];
    float s = 0.0f;
    if (values) {
        for (int64_t k = b + sub; k < e; k += 16) s += values[k];
    } else {
        for (int64_t k = b + sub; k < e; k += 16) s += 1.0f;
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) s += __shfl_xor(s, d, 16);
    if (sub == 0) out[row] = s;
}

}  // namespace dmcf

extern "C" {

int dmcf_version(void) { return 20600; }  // 2.0.0: round 2 removed dmcf_cconv_geometry and the geometry field of dmcf_cconv_args; 2.1.0: filter_tile_mask; 2.2.0: DMCF_FLAG_SKIP_SELF; 2.3.0: row_length_hint (splat F); 2.4.0: dmcf_points_aabb; 2.5.0: DMCF_FLAG_FILTER_PACKED, row_length_hint = 1 (splat H); 2.6.0: dmcf_cconv_scatter_* (splat S), hashed grid_pos table (table_cells < 0)

const char* dmcf_error_string(int code) {
    switch (code) {
        case DMCF_OK: return "ok";
        case DMCF_EINVAL: return "invalid argument";
        case DMCF_EWORKSPACE: return "workspace too small";
        case DMCF_ELAUNCH: return "HIP error while enqueuing (see dmcf_last_hip_error)";
        case DMCF_EUNSUPPORTED: return "option not implemented on the HIP path";
        default: return "unknown error code";
    }
}

int dmcf_last_hip_error(void) { return dmcf::g_last_hip_error; }

int dmcf_dense_forward(const float* x, int64_t n, int32_t k, const float* W, int32_t m, const float* bias,
                       const float* residual, float* out, dmcf_stream_t stream) {
    if (n < 0 || k <= 0 || m <= 0 || (n > 0 && (!x || !W || !out))) return DMCF_EINVAL;
    if ((k & 3) || k > 64 || m > 64) return DMCF_EUNSUPPORTED;
    if ((uintptr_t)x & 15) return DMCF_EUNSUPPORTED;
    if (n == 0) return DMCF_OK;
    const int64_t ntiles = (n + 15) / 16;
    const int64_t want = (ntiles + 3) / 4;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);  // eight waves per SIMD of 256 CUs; each keeps the filter
    const int nt = (m + 15) / 16;
#define DMCF_DENSE(NTT)                                                                                                   \
    hipLaunchKernelGGL(dmcf::dense_rows<NTT>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, (int)k, W, (int)m, bias,  \
                       residual, out, ntiles)
    if (nt <= 1) DMCF_DENSE(1);
    else if (nt == 2) DMCF_DENSE(2);
    else DMCF_DENSE(4);
#undef DMCF_DENSE
    return dmcf::check_launch();
}

size_t dmcf_points_aabb_workspace_bytes(void) { return (size_t)dmcf::kAabbBlocks * 8 * sizeof(float); }

int dmcf_points_aabb(const float* points, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                     dmcf_stream_t stream) {
    if (n < 0 || !out || (n > 0 && !points)) return DMCF_EINVAL;
    if (!workspace || workspace_bytes < dmcf_points_aabb_workspace_bytes()) return DMCF_EWORKSPACE;
    const int64_t want = (n + 1023) / 1024;  // >= 4 points per thread
    const int nb = (int)(want < 1 ? 1 : (want > dmcf::kAabbBlocks ? dmcf::kAabbBlocks : want));
    hipLaunchKernelGGL(dmcf::aabb_partial, dim3(nb), dim3(256), 0, (hipStream_t)stream, points, n, (float*)workspace);
    hipLaunchKernelGGL(dmcf::aabb_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, nb, out);
    return dmcf::check_launch();
}

int dmcf_reduce_subarrays_sum(const float* values, const int64_t* row_splits, int64_t n_rows, float* out,
                              dmcf_stream_t stream) {
    if (n_rows < 0 || (n_rows > 0 && (!row_splits || !out))) return DMCF_EINVAL;
    if (n_rows == 0) return DMCF_OK;
    const int64_t threads = n_rows * 16;
    const unsigned grid = (unsigned)((threads + 255) / 256);
    hipLaunchKernelGGL(dmcf::reduce_subarrays_sum_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, values,
                       row_splits, n_rows, out);
    return dmcf::check_launch();
}

}  // extern "C"
