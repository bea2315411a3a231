// farthest_point_sample + gather_point (utils/tools/sampling.cu:125-201; the sub-sampling of get_dilated_pos when a
// multi-scale config has voxel_size: None, utils/tools/losses.py:274-282).
//
// FPS is sequential in the samples: sample j is the point farthest from samples 0..j-1.  One 1024-thread workgroup
// runs all m iterations (the reference uses one 512-thread block): every thread keeps the running minimum distance
// of its points (the first 8 per thread in registers together with their coordinates, the rest in a workspace array),
// each iteration is one pass of min-updates + an arg-max (wave shuffles, one LDS exchange, 2 barriers).
// Result contract = the reference kernel's: sample 0 is point 0; distances are squared L2 in float32,
// (dx*dx + dy*dy) + dz*dz without fused multiply-adds; among equal maxima the reference's 512-thread strided
// loop + tree reduction keeps the point with the smallest (index mod 512), then the smallest index -- reproduced here
// by comparing (distance, -(index & 511), -index), independent of this kernel's own thread count.
#include "common.h"

namespace dmcf {

constexpr int kFpsThreads = 1024;
constexpr int kFpsReg = 8;  // points per thread held in registers

struct FpsBest {
    float d;
    int i;
};

__device__ __forceinline__ bool fps_better(float d, int i, float bd, int bi) {
    if (d != bd) return d > bd;
    const int a = i & 511, b = bi & 511;
    if (a != b) return a < b;
    return i < bi;
}

__global__ __launch_bounds__(kFpsThreads) void fps_kernel(const float* __restrict__ pts, int n, int m, float* __restrict__ temp,
                                                         int32_t* __restrict__ idxs) {
    __shared__ float sd[kFpsThreads / 64];
    __shared__ int si[kFpsThreads / 64];
    __shared__ int s_old;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float rx[kFpsReg], ry[kFpsReg], rz[kFpsReg], rd[kFpsReg];
#pragma unroll
    for (int u = 0; u < kFpsReg; ++u) {
        const int k = tid + u * kFpsThreads;
        rd[u] = 1e38f;  // sampling.cu:139
        rx[u] = ry[u] = rz[u] = 0.0f;
        if (k < n) {
            rx[u] = pts[3 * k];
            ry[u] = pts[3 * k + 1];
            rz[u] = pts[3 * k + 2];
        }
    }
    for (int k = tid + kFpsReg * kFpsThreads; k < n; k += kFpsThreads) temp[k] = 1e38f;
    int old = 0;
    if (tid == 0) idxs[0] = 0;
    for (int j = 1; j < m; ++j) {
        const float x1 = pts[3 * old], y1 = pts[3 * old + 1], z1 = pts[3 * old + 2];
        float best = -1.0f;
        int besti = 0;
#pragma unroll
        for (int u = 0; u < kFpsReg; ++u) {
            const int k = tid + u * kFpsThreads;
            if (k < n) {
                const float d2 = fminf(dist2_unfused(rx[u], ry[u], rz[u], x1, y1, z1), rd[u]);
                rd[u] = d2;
                if (fps_better(d2, k, best, besti)) {
                    best = d2;
                    besti = k;
                }
            }
        }
        for (int k = tid + kFpsReg * kFpsThreads; k < n; k += kFpsThreads) {
            const float d2 = fminf(dist2_unfused(pts[3 * k], pts[3 * k + 1], pts[3 * k + 2], x1, y1, z1), temp[k]);
            temp[k] = d2;
            if (fps_better(d2, k, best, besti)) {
                best = d2;
                besti = k;
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float od = __shfl_xor(best, d, kWave);
            const int oi = __shfl_xor(besti, d, kWave);
            if (fps_better(od, oi, best, besti)) {
                best = od;
                besti = oi;
            }
        }
        if (lane == 0) {
            sd[wave] = best;
            si[wave] = besti;
        }
        __syncthreads();
        if (wave == 0) {
            float b = lane < kFpsThreads / 64 ? sd[lane] : -2.0f;
            int bi = lane < kFpsThreads / 64 ? si[lane] : 0x7fffffff;
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) {
                const float od = __shfl_xor(b, d, kWave);
                const int oi = __shfl_xor(bi, d, kWave);
                if (fps_better(od, oi, b, bi)) {
                    b = od;
                    bi = oi;
                }
            }
            if (lane == 0) {
                s_old = bi;
                idxs[j] = bi;
            }
        }
        __syncthreads();
        old = s_old;
    }
}

__global__ void gather_rows_kernel(const float* __restrict__ inp, const int32_t* __restrict__ idx, int64_t m, int ch,
                                   float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * ch) return;
    const int64_t j = e / ch;
    out[e] = inp[(int64_t)idx[j] * ch + (e - j * ch)];
}

}  // namespace dmcf

using namespace dmcf;

extern "C" {

size_t dmcf_fps_workspace_bytes(int64_t n_points) { return n_points < 0 ? 0 : align_up((size_t)(n_points > 0 ? n_points : 1) * 4, 256); }

int dmcf_farthest_point_sample(const float* points, int64_t n_points, int64_t n_samples, void* workspace, size_t workspace_bytes,
                               int32_t* sample_index, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_points < 0 || n_samples < 0 || (n_samples > 0 && (!points || !sample_index || n_points == 0))) return DMCF_EINVAL;
    if (n_points > 0x7fffffff || n_samples > 0x7fffffff) return DMCF_EUNSUPPORTED;
    if (n_samples == 0) return DMCF_OK;
    if (!workspace || workspace_bytes < dmcf_fps_workspace_bytes(n_points)) return DMCF_EWORKSPACE;
    hipLaunchKernelGGL(fps_kernel, dim3(1), dim3(kFpsThreads), 0, stream, points, (int)n_points, (int)n_samples, (float*)workspace,
                       sample_index);
    return check_launch();
}

int dmcf_gather_point(const float* inp, const int32_t* index, int64_t n_index, int channels, float* out, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n_index < 0 || channels <= 0 || (n_index > 0 && (!inp || !index || !out))) return DMCF_EINVAL;
    const int64_t total = n_index * channels;
    if (total == 0) return DMCF_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, inp, index, n_index, channels,
                       out);
    return check_launch();
}

}  // extern "C"
