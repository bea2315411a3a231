// CConv between two point sets that lie on ALIGNED REGULAR LATTICES (the coarse scales of the multi-scale models: both come
// out of grid_pos with the same centre, utils/tools/losses.py:136-181, and the spacing of one is an integer multiple of
// the other's): x_in - x_out only takes the values d * voxel, d an integer vector, and everything ml3d.ops.continuous_conv
// (utils/convolutions.py:414-431) evaluates per neighbour pair -- window, ball -> cube map, trilinear weights -- depends on
// d alone.  So
//
//     out_i = sum_d  W_d^T f_{cell(i) + d},     W_d[c][o] = window(d) * sum_{8 corners t} w_t(d) W[cell_t(d)][c][o]
//
// one [Cin x Cout] matrix per stencil offset (a few hundred to ~2000 offsets inside the radius), no neighbour search, no
// per-pair geometry: a dense product over features gathered through a cell -> point table of the input lattice.
//   lat_build_filters   one thread per element of the per-offset matrices, packed in MFMA B-fragment order
//   lat_conv_kernel     tile = 16 output points (M of v_mfma_f32_16x16x4_f32), N = 16 output channels, K = 4 input channels
//                       of one offset; the 4 waves of a workgroup split the stencil, 4 offsets in flight per wave (table
//                       lookup -> feature gather are two dependent loads), partial sums reduced through LDS.
// The offsets are nominal (d * voxel in fp32); the reference forms fl(x_in) - fl(x_out), which differs by ~1 ulp of |x|:
// about 3e-6 of the output scale at |x| ~ 5.  Pairs at exactly the radius have window 0 either way.
#include "cconv_common.h"

namespace dmcf {

struct LatParams {
    const float* Wp;          // [S][KS][NT][64]
    const int32_t* stencil;   // [S][4]: dx, dy, dz of the input cell relative to out_cell * out_step
    int S, KS, NT, cin, cout;
    const int32_t* out_cells;  // [n_out][3] (x, y, z)
    int64_t n_out;
    int out_step;
    const int32_t* table;      // [tdz][tdy][tdx]: input point index or -1
    int tmin[3], tdim[3];      // (x, y, z)
    const float* feat;
    const float* bias;
    float* out;
    int flags;
};

__global__ __launch_bounds__(256) void lat_build_filters(const float* __restrict__ W, float* __restrict__ Wp,
                                                         const int32_t* __restrict__ stencil, int S, int KS, int NT, CconvParams p,
                                                         float vx, float vy, float vz) {
    const int64_t total = (int64_t)S * KS * NT * 64;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = e;
        const int n = (int)(t & 15); t >>= 4;
        const int q = (int)(t & 3); t >>= 2;
        const int nt = (int)(t % NT); t /= NT;
        const int ks = (int)(t % KS); t /= KS;
        const int s = (int)t;
        const int c = q * KS + ks, o = nt * 16 + n;  // lane q of the A operand holds channels q * KS .. q * KS + KS - 1
        float v = 0.0f;
        if (c < p.cin && o < p.cout) {
            float x = (float)stencil[4 * s] * vx, y = (float)stencil[4 * s + 1] * vy, z = (float)stencil[4 * s + 2] * vz;
            const float d2 = (x * x + y * y) + z * z;
            const float a = window_value(p.window, d2, p.inv_r2, p.window_fac);
            filter_coords<true>(x, y, z, p);
            int bx, by, bz;
            float wx[2], wy[2], wz[2];
            axis_weights(x, p.sx, p.interp, bx, wx[0], wx[1]);
            axis_weights(y, p.sy, p.interp, by, wy[0], wy[1]);
            axis_weights(z, p.sz, p.interp, bz, wz[0], wz[1]);
            for (int iz = 0; iz < 2; ++iz)
                for (int iy = 0; iy < 2; ++iy)
                    for (int ix = 0; ix < 2; ++ix) {
                        const float w = wz[iz] * wy[iy] * wx[ix];
                        if (w == 0.0f) continue;  // also the "+1" cells that do not exist on size-1 axes
                        const int cz = min(bz + iz, p.sz - 1), cy = min(by + iy, p.sy - 1), cx = min(bx + ix, p.sx - 1);
                        v += w * W[((((int64_t)cz * p.sy + cy) * p.sx + cx) * p.cin + c) * p.cout + o];
                    }
            v *= a;
        }
        Wp[e] = v;
    }
}

constexpr int kLatTW = 2;   // 16-point tiles per wave
constexpr int kLatCH = 32;  // stencil offsets per LDS chunk of the per-offset matrices
#ifndef LAT_U
#define LAT_U 4
#endif
constexpr int kLatU = LAT_U;  // offsets whose lookup -> gather chains are in flight together (x kLatTW tiles)

// Workgroup = 4 waves x kLatTW tiles = 128 output points.  Every wave walks the WHOLE stencil for its own tiles (no
// cross-wave reduction); the per-offset matrices are staged through LDS in chunks of kLatCH offsets and shared by the
// waves.  Per offset and tile: one table lookup and one feature gather (KST consecutive channels per lane).
template <int NTT, int KST>
__global__ __launch_bounds__(256) void lat_conv_kernel(const LatParams p) {
    __shared__ float Ws[kLatCH * KST * NTT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    int cx[kLatTW], cy[kLatTW], cz[kLatTW];
    bool valid[kLatTW];
    f32x4 acc[kLatTW][NTT];
#pragma unroll
    for (int t = 0; t < kLatTW; ++t) {
        const int64_t i = ((int64_t)blockIdx.x * 4 * kLatTW + wave * kLatTW + t) * 16 + m;
        valid[t] = i < p.n_out;
        cx[t] = cy[t] = cz[t] = 0;
        if (valid[t]) {
            cx[t] = p.out_cells[3 * i] * p.out_step - p.tmin[0];
            cy[t] = p.out_cells[3 * i + 1] * p.out_step - p.tmin[1];
            cz[t] = p.out_cells[3 * i + 2] * p.out_step - p.tmin[2];
        }
#pragma unroll
        for (int n = 0; n < NTT; ++n) acc[t][n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    const float* fq = p.feat + q * KST;
    for (int s0 = 0; s0 < p.S; s0 += kLatCH) {
        const int ns = min(kLatCH, p.S - s0);
        __syncthreads();
        for (int e = threadIdx.x; e < ns * KST * NTT * 64; e += 256) {
            // chunk layout [offset][ks][n < NTT][64]; the packed array has NT (<= NTT) tiles per K step
            const int l = e & 63, n = (e >> 6) % NTT, ks = (e >> 6) / NTT % KST, so = (e >> 6) / (NTT * KST);
            Ws[e] = n < p.NT ? p.Wp[(((int64_t)(s0 + so) * KST + ks) * p.NT + n) * 64 + l] : 0.0f;
        }
        __syncthreads();
        for (int so = 0; so < ns; so += kLatU) {
            // kLatU offsets x kLatTW tiles: independent lookup -> gather chains in flight together
            int idx[kLatU][kLatTW];
#pragma unroll
            for (int u = 0; u < kLatU; ++u) {
                const int s = s0 + so + u;  // wave uniform
                const int dx = s < p.S ? p.stencil[4 * s] : 0, dy = s < p.S ? p.stencil[4 * s + 1] : 0,
                          dz = s < p.S ? p.stencil[4 * s + 2] : 0;
#pragma unroll
                for (int t = 0; t < kLatTW; ++t) {
                    const int x = cx[t] + dx, y = cy[t] + dy, z = cz[t] + dz;
                    idx[u][t] = -1;
                    if (so + u < ns && valid[t] && (unsigned)x < (unsigned)p.tdim[0] && (unsigned)y < (unsigned)p.tdim[1] &&
                        (unsigned)z < (unsigned)p.tdim[2])
                        idx[u][t] = p.table[((int64_t)z * p.tdim[1] + y) * p.tdim[0] + x];
                }
            }
            float fv[kLatU][kLatTW][KST];
#pragma unroll
            for (int u = 0; u < kLatU; ++u)
#pragma unroll
                for (int t = 0; t < kLatTW; ++t) {
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) fv[u][t][ks] = 0.0f;
                    if (idx[u][t] >= 0) {
                        const float* src = fq + (int64_t)idx[u][t] * p.cin;
                        if constexpr (KST == 2) {
                            const f32x2 v = *(const f32x2*)src;  // rows are 16-byte aligned (cin = 8), q * 2 floats in
                            fv[u][t][0] = v.x;
                            fv[u][t][1] = v.y;
                        } else {
#pragma unroll
                            for (int ks = 0; ks < KST; ++ks) fv[u][t][ks] = src[ks];
                        }
                    }
                }
#pragma unroll
            for (int u = 0; u < kLatU; ++u) {
                if (so + u < ns) {
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks)
#pragma unroll
                        for (int n = 0; n < NTT; ++n) {
                            const float w = Ws[(((so + u) * KST + ks) * NTT + n) * 64 + lane];
#pragma unroll
                            for (int t = 0; t < kLatTW; ++t) {
                                acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[u][t][ks], w, acc[t][n], 0, 0, 0);
                            }
                        }
                }
            }
        }
    }
    // D layout: lane (rows 4 (lane >> 4) + r, column lane & 15)
#pragma unroll
    for (int t = 0; t < kLatTW; ++t) {
        const int64_t i0 = ((int64_t)blockIdx.x * 4 * kLatTW + wave * kLatTW + t) * 16;
#pragma unroll
        for (int n = 0; n < NTT; ++n) {
            const int o = 16 * n + m;
            if (o >= p.cout) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t ii = i0 + 4 * q + r;
                if (ii >= p.n_out) continue;
                float v = acc[t][n][r];
                if (p.bias) v += p.bias[o];
                float* dst = p.out + ii * p.cout + o;
                if (p.flags & DMCF_FLAG_ACCUMULATE) v += *dst;
                *dst = v;
            }
        }
    }
}

static size_t lat_packed_floats(const dmcf_lattice_conv_args* a) {
    const int KS = (a->filter_dims[3] + 3) / 4, NT = (a->filter_dims[4] + 15) / 16;
    return (size_t)a->n_offsets * KS * NT * 64;
}

static int lat_validate(const dmcf_lattice_conv_args* a) {
    if (!a) return DMCF_EINVAL;
    for (int k = 0; k < 5; ++k)
        if (a->filter_dims[k] <= 0) return DMCF_EINVAL;
    if (a->n_out < 0 || a->n_offsets < 0 || a->out_step <= 0 || !(a->extent > 0.0f)) return DMCF_EINVAL;
    if (a->n_out > 0 && (!a->filters || !a->out_cells || !a->out || !a->inp_table || !a->inp_features || (a->n_offsets > 0 && !a->offsets)))
        return DMCF_EINVAL;
    for (int k = 0; k < 3; ++k)
        if (a->table_dims[k] <= 0 || !(a->voxel[k] >= 0.0f)) return DMCF_EINVAL;
    if (a->flags & (DMCF_FLAG_SYMMETRIC | DMCF_FLAG_NORMALIZE)) return DMCF_EUNSUPPORTED;
    if (a->window == DMCF_WINDOW_EXPLICIT) return DMCF_EUNSUPPORTED;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin != 4 && cin != 8) || cout > 32) return DMCF_EUNSUPPORTED;
    return DMCF_OK;
}

}  // namespace dmcf

using namespace dmcf;

extern "C" {

size_t dmcf_lattice_conv_workspace_bytes(const dmcf_lattice_conv_args* a) {
    if (lat_validate(a) != DMCF_OK) return 256;
    return 256 + align_up(lat_packed_floats(a) * sizeof(float), 256);
}

int dmcf_lattice_conv_forward(const dmcf_lattice_conv_args* a, void* workspace, size_t workspace_bytes, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = lat_validate(a);
    if (rc != DMCF_OK) return rc;
    if (a->n_out == 0) return DMCF_OK;
    if (!workspace || ((uintptr_t)workspace & 255)) return DMCF_EINVAL;
    if (workspace_bytes < dmcf_lattice_conv_workspace_bytes(a)) return DMCF_EWORKSPACE;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    const int KS = (cin + 3) / 4, NT = (cout + 15) / 16;
    CconvParams cp = {};
    cp.sz = a->filter_dims[0]; cp.sy = a->filter_dims[1]; cp.sx = a->filter_dims[2];
    cp.K = cp.sx * cp.sy * cp.sz;
    cp.cin = cin; cp.cout = cout;
    cp.inv_extent = 1.0f / a->extent;
    const float radius = 0.5f * a->extent;
    cp.inv_r2 = 1.0f / (radius * radius);
    cp.window_fac = a->window_fac;
    cp.window = a->window;
    cp.mapping = a->coordinate_mapping;
    cp.interp = a->interpolation;
    cp.flags = a->flags;
    float* packed = (float*)workspace;
    if (a->n_offsets > 0) {
        const int64_t total = (int64_t)lat_packed_floats(a);
        const unsigned g = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(lat_build_filters, dim3(g < 4096u ? g : 4096u), dim3(256), 0, stream, a->filters, packed, a->offsets,
                           (int)a->n_offsets, KS, NT, cp, a->voxel[0], a->voxel[1], a->voxel[2]);
    }
    LatParams p;
    p.Wp = packed;
    p.stencil = a->offsets;
    p.S = (int)a->n_offsets; p.KS = KS; p.NT = NT; p.cin = cin; p.cout = cout;
    p.out_cells = a->out_cells; p.n_out = a->n_out; p.out_step = a->out_step;
    p.table = a->inp_table;
    for (int k = 0; k < 3; ++k) { p.tmin[k] = a->table_min[k]; p.tdim[k] = a->table_dims[k]; }
    p.feat = a->inp_features; p.bias = a->bias; p.out = a->out; p.flags = a->flags;
    const int64_t tiles = (a->n_out + 16 * 4 * kLatTW - 1) / (16 * 4 * kLatTW);
    if (tiles > 0x7fffffff) return DMCF_EUNSUPPORTED;
    const dim3 grid((unsigned)tiles), block(256);
    if (KS == 1 && NT == 1) hipLaunchKernelGGL((lat_conv_kernel<1, 1>), grid, block, 0, stream, p);
    else if (KS == 1) hipLaunchKernelGGL((lat_conv_kernel<2, 1>), grid, block, 0, stream, p);
    else if (NT == 1) hipLaunchKernelGGL((lat_conv_kernel<1, 2>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((lat_conv_kernel<2, 2>), grid, block, 0, stream, p);
    return check_launch();
}

}  // extern "C"
