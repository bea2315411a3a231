// CConv onto a coarse LATTICE with few output channels, FILTER FIRST and INPUT STATIONARY -- splat S (round 6).
//
// The layers that gather the particles (s0, ~1.1M points, 24 / 16 channels) onto the coarsest grid_pos lattice (s2, ~150k points,
// 4 / 8 channels; models/hrnet.py:83-93, radius 0.4 = 4 lattice cells: ~1900 pairs per row, 2.9e8 pairs) spend their time in
// the splat: 8 trilinear corners x Cin multiply-adds per pair into the output point's B tile, then one small contraction.  With
// Cout <= 8 the other order is cheaper by Cin / Cout:
//
//     G_j[cell][o] = sum_c f_j[c] W[cell][c][o]                           once per INPUT point   (64 cells x Cout: 1 - 2 KB)
//     out_i[o]    += a_ij * sum_{8 corners k} w_k(ij) G_j[cell_k][o]      per pair: 8 x Cout multiply-adds, every lane a pair
//
// A gather form of this reads 8 x Cout scattered floats of G_j per pair from a 1 - 2 GB array: cache arithmetic kills it
// (splat S).  So the walk is TRANSPOSED: a wave owns an input point j, keeps G_j in LDS, and walks j's row of
// the TRANSPOSED list (the s2 -> s0 list the step builds anyway: row j = the lattice points within R of particle j), one pair
// per lane: geometry, 8 LDS reads of G_j, 8 x Cout / 2 packed multiply-adds, and an ADD INTO THE OUTPUT POINT.  Those adds are the
// problem of every scatter form -- 2.9e8 x Cout of them, and a float sum whose order depends on the schedule.  Here:
//
//   * input points are processed in BLOCKS of m x m x m lattice cells (the caller sorts them by block: `order`, `block_start`,
//     `block_cell`).  All outputs a block can reach lie in a box of D^3 lattice cells, D = m + 2 reach + 1, whose accumulators live
//     in LDS ([Cout][D^3] 64-bit integers, channel major: conflict free for distinct slots).  A block of ~64 particles adds ~16k
//     pairs into ~1000 slots and flushes each touched slot ONCE with a global atomic: 13 pairs per flushed value.
//   * sums are FIXED POINT: a contribution c is added as round(c * 2^s) into a 64-bit integer, 2^s = 2^30 / (a power-of-two bound
//     of |c|: max_j |f_j|_1 * max |W|, formed on the device by the caller, `scale`).  Integer addition is associative, so LDS
//     atomics, global atomics and any schedule give THE SAME BITS: the step stays bit reproducible with no ordering, no staging
//     and no barrier in the pair loop.  Resolution: 2^-30 of the bound per term (terms are float32: 2^-24 of themselves).
//
// Workgroup = 8 waves, one per input point at a time; 2 workgroups per CU at Cout = 4 (LDS: W 24 KB + accumulators 42.6 KB + slot
// -> output index 5.3 KB + G rows 8 KB).  Per 64-pair batch ~150 vector instructions, all 64 lanes busy (splat F: 64 matrix
// instructions + ~190 scalar + ~240 vector per batch at two waves per SIMD).
//
// Restrictions (the dispatch in dmcf_amd/utils/convolutions.py checks them, the entry point returns DMCF_EUNSUPPORTED): 4x4x4 filter,
// Cout 4 or 8, Cin <= 32, linear interpolation, align_corners, volume-preserving map, poly6 or no window (formed from the positions),
// no per-point importance, no normalisation; output points on a lattice of spacing `voxel` (cell = rint((x - x_0) / voxel)).
#include "cconv_common.h"

namespace dmcf {

constexpr int kSWaves = 8;
constexpr int kSThreads = 64 * kSWaves;

struct SctParams {
    const float* W;          // [4][4][4][cin][cout]
    const float* out_pos;    // [n_out][3]
    const float* inp_pos;    // [n_inp][3]
    const float* inp_feat;   // [n_inp][cin]
    const int32_t* t_idx;    // transposed list: row j = output indices of input point j
    const int64_t* t_rs;     // row begin (CSR row splits, or j * stride for padded rows)
    const int32_t* t_cnt;    // optional pairs per row (padded rows); NULL = CSR
    int64_t t_cap;           // entries t_idx holds
    const int32_t* order;    // [n_inp] input rows sorted by block
    const int32_t* block_start;  // [n_inp + 1]: first sorted position of block b; n_inp from the first unused entry on
    const int32_t* block_cell;   // [n_inp][3] per SORTED position: lattice cell (relative to out_pos[0]) of its block's slot box origin
    const float* scale;      // device [2]: 2^s, 2^-s
    unsigned long long* acc; // [n_out][cout] 64-bit sums (zeroed by the launch)
    int64_t n_out, n_inp;
    int cin, D;
    float inv_voxel, inv_extent, inv_r2, window_fac;
    int window;
    int* err;                // device flag: a pair fell outside its block's slot box (must stay 0)
};

__device__ __forceinline__ void lds_add_i64(unsigned long long* p, long long v) {
    __hip_atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int COUT>
__global__ __launch_bounds__(kSThreads, COUT == 4 ? 2 : 1) void cconv_sct_kernel(const SctParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = p.cin, D = p.D, NS = D * D * D;
    const int NSp = (NS + 3) & ~3;
    // LDS: accumulators [COUT][NSp] u64 | slot -> output index [NSp] | W [cin][64 * COUT] | G rows [kSWaves][64 * COUT]
    unsigned long long* Acc = (unsigned long long*)smem_raw;
    int* Sidx = (int*)(Acc + (size_t)COUT * NSp);
    float* Wl = (float*)(Sidx + NSp);
    float* Gw = Wl + cin * 64 * COUT + wave * 64 * COUT;

    // W[cell][c][o] -> Wl[c][cell * COUT + o]
    for (int e = tid; e < 64 * cin * COUT; e += kSThreads) {
        const int o = e % COUT, c = (e / COUT) % cin, cell = e / (COUT * cin);
        Wl[c * 64 * COUT + cell * COUT + o] = p.W[e];
    }
    for (int s = tid; s < COUT * NSp; s += kSThreads) Acc[s] = 0ull;
    for (int s = tid; s < NSp; s += kSThreads) Sidx[s] = -1;
    const float S = p.scale[0];
    const float ox0 = p.out_pos[0], oy0 = p.out_pos[1], oz0 = p.out_pos[2];
    CconvParams gp;  // (only what filter_coords<false> reads)
    gp.inv_extent = p.inv_extent;
    gp.sx = gp.sy = gp.sz = 4;
    __syncthreads();

    for (int64_t b = blockIdx.x; b <= p.n_inp; b += gridDim.x) {
        const int s0 = p.block_start[b];
        if (s0 >= p.n_inp) break;
        const int s1 = min(p.block_start[b + 1], (int)p.n_inp);
        const int bx0 = p.block_cell[3 * (int64_t)s0], by0 = p.block_cell[3 * (int64_t)s0 + 1], bz0 = p.block_cell[3 * (int64_t)s0 + 2];
        for (int r = s0 + wave; r < s1; r += kSWaves) {
            const int j = __builtin_amdgcn_readfirstlane(p.order[r]);
            // ---- G_j: lane = cell, COUT outputs each; features through the scalar unit (j is wave uniform)
            const float* fj = p.inp_feat + (int64_t)j * cin;
            float g[COUT];
#pragma unroll
            for (int o = 0; o < COUT; ++o) g[o] = 0.0f;
            for (int c = 0; c < cin; ++c) {
                const float f = fj[c];
                const float* wr = Wl + c * 64 * COUT + lane * COUT;
#pragma unroll
                for (int q = 0; q < COUT; q += 4) {
                    const f32x4 w = *(const f32x4*)(wr + q);
                    g[q] = fmaf(f, w.x, g[q]);
                    g[q + 1] = fmaf(f, w.y, g[q + 1]);
                    g[q + 2] = fmaf(f, w.z, g[q + 2]);
                    g[q + 3] = fmaf(f, w.w, g[q + 3]);
                }
            }
#pragma unroll
            for (int q = 0; q < COUT; q += 4) *(f32x4*)(Gw + lane * COUT + q) = (f32x4){g[q], g[q + 1], g[q + 2], g[q + 3]};
            const float px = p.inp_pos[3 * (int64_t)j], py = p.inp_pos[3 * (int64_t)j + 1], pz = p.inp_pos[3 * (int64_t)j + 2];
            const int64_t rb = p.t_rs[j];
            int cnt = p.t_cnt ? p.t_cnt[j] : (int)(p.t_rs[j + 1] - rb);
            if (rb + cnt > p.t_cap) cnt = 0;  // (a row past the buffer: the search skipped it and the caller repeats the step)
            cnt = __builtin_amdgcn_readfirstlane(cnt);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- the row's pairs, one per lane
            for (int k0 = 0; k0 < cnt; k0 += 64) {
                const int k = k0 + lane;
                const bool valid = k < cnt;
                const int i = valid ? p.t_idx[rb + k] : 0;
                const float qx = p.out_pos[3 * (int64_t)i], qy = p.out_pos[3 * (int64_t)i + 1], qz = p.out_pos[3 * (int64_t)i + 2];
                float x = px - qx, y = py - qy, z = pz - qz;
                float a = p.window == DMCF_WINDOW_NONE ? 1.0f : window_value(DMCF_WINDOW_POLY6, rel_dist2(x, y, z), p.inv_r2, p.window_fac);
                a = valid ? a * S : 0.0f;
                filter_coords<false>(x, y, z, gp);
                int bx, by, bz;
                float wx0, wx1, wy0, wy1, wz0, wz1;
                axis_weights_linear(x, 4, bx, wx0, wx1);
                axis_weights_linear(y, 4, by, wy0, wy1);
                axis_weights_linear(z, 4, bz, wz0, wz1);
                const float* gc = Gw + ((bz * 4 + by) * 4 + bx) * COUT;
                // corner weights in Open3D's product order (x-weight * y-weight) * z-weight, times the window (and 2^s)
                const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
                float acc[COUT];
#pragma unroll
                for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
#pragma unroll
                for (int zz = 0; zz < 2; ++zz) {
                    const float wz = (zz ? wz1 : wz0) * a;
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy) {
                        const float wa = (yy ? w01 : w00) * wz, wb = (yy ? w11 : w10) * wz;
                        const float* gq = gc + (zz * 16 + yy * 4) * COUT;
#pragma unroll
                        for (int q = 0; q < COUT; q += 4) {
                            const f32x4 ga = *(const f32x4*)(gq + q), gb = *(const f32x4*)(gq + COUT + q);
                            acc[q] = fmaf(wa, ga.x, fmaf(wb, gb.x, acc[q]));
                            acc[q + 1] = fmaf(wa, ga.y, fmaf(wb, gb.y, acc[q + 1]));
                            acc[q + 2] = fmaf(wa, ga.z, fmaf(wb, gb.z, acc[q + 2]));
                            acc[q + 3] = fmaf(wa, ga.w, fmaf(wb, gb.w, acc[q + 3]));
                        }
                    }
                }
                // the output point's slot in the block's box
                const int cx = (int)rintf((qx - ox0) * p.inv_voxel) - bx0, cy = (int)rintf((qy - oy0) * p.inv_voxel) - by0,
                          cz = (int)rintf((qz - oz0) * p.inv_voxel) - bz0;
                const bool inside = (unsigned)cx < (unsigned)D && (unsigned)cy < (unsigned)D && (unsigned)cz < (unsigned)D;
                if (valid && !inside) *p.err = 1;
                if (valid && inside) {
                    const int slot = (cz * D + cy) * D + cx;
                    Sidx[slot] = i;
#pragma unroll
                    for (int o = 0; o < COUT; ++o) lds_add_i64(Acc + o * NSp + slot, (long long)__float2int_rn(acc[o]));
                }
            }
            __builtin_amdgcn_wave_barrier();  // (G_j is overwritten by the wave's next row)
        }
        __syncthreads();
        // ---- flush: every touched slot once, and clear for the next block
        for (int s = tid; s < NS; s += kSThreads) {
            const int i = Sidx[s];
            if (i >= 0) {
                Sidx[s] = -1;
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const unsigned long long v = Acc[o * NSp + s];
                    Acc[o * NSp + s] = 0ull;
                    if (v) __hip_atomic_fetch_add(p.acc + (int64_t)i * COUT + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        __syncthreads();
    }
}

// out[i][o] (+)= acc[i][o] * 2^-s + bias[o]
__global__ void cconv_sct_finish(const long long* __restrict__ acc, const float* __restrict__ scale, const float* __restrict__ bias,
                                 float* __restrict__ out, int64_t n, int cout, int accumulate) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v = (float)((double)acc[e] * (double)scale[1]);
    if (bias) v += bias[e % cout];
    out[e] = accumulate ? out[e] + v : v;
}

static size_t sct_lds_bytes(int cin, int cout, int D) {
    const int NS = D * D * D, NSp = (NS + 3) & ~3;
    return (size_t)cout * NSp * 8 + (size_t)NSp * 4 + (size_t)cin * 64 * cout * 4 + (size_t)kSWaves * 64 * cout * 4;
}

}  // namespace dmcf

using namespace dmcf;

static int sct_check(const dmcf_cconv_scatter_args* a) {
    if (!a || !a->filters || !a->out_positions || !a->inp_positions || !a->inp_features || !a->t_index || !a->t_row_begin ||
        !a->order || !a->block_start || !a->block_cell || !a->scale || !a->out)
        return DMCF_EINVAL;
    if (a->n_out <= 0 || a->n_inp <= 0 || a->cin <= 0 || a->extent <= 0.0f || a->voxel <= 0.0f || a->block_cells <= 0 || a->reach <= 0)
        return DMCF_EINVAL;
    if (a->filter_dims[0] != 4 || a->filter_dims[1] != 4 || a->filter_dims[2] != 4 || (a->cout != 4 && a->cout != 8) || a->cin > 32)
        return DMCF_EUNSUPPORTED;
    if (a->window != DMCF_WINDOW_NONE && a->window != DMCF_WINDOW_POLY6) return DMCF_EUNSUPPORTED;
    if (a->flags & ~(DMCF_FLAG_ALIGN_CORNERS | DMCF_FLAG_ACCUMULATE)) return DMCF_EUNSUPPORTED;
    if (!(a->flags & DMCF_FLAG_ALIGN_CORNERS)) return DMCF_EUNSUPPORTED;
    const int D = a->block_cells + 2 * a->reach + 1;
    if (sct_lds_bytes(a->cin, a->cout, D) > 160 * 1024) return DMCF_EUNSUPPORTED;
    return DMCF_OK;
}

extern "C" size_t dmcf_cconv_scatter_workspace_bytes(const dmcf_cconv_scatter_args* a) {
    if (!a || a->n_out <= 0 || a->cout <= 0) return 0;
    return align_up((size_t)a->n_out * a->cout * 8, 256) + 256;
}

extern "C" int dmcf_cconv_scatter_forward(const dmcf_cconv_scatter_args* a, void* workspace, size_t workspace_bytes, void* stream_) {
    const int rc = sct_check(a);
    if (rc != DMCF_OK) return rc;
    if (!workspace || workspace_bytes < dmcf_cconv_scatter_workspace_bytes(a)) return DMCF_EWORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const size_t acc_bytes = align_up((size_t)a->n_out * a->cout * 8, 256);
    int* err = (int*)((char*)workspace + acc_bytes);
    if (hipMemsetAsync(workspace, 0, acc_bytes + 256, stream) != hipSuccess) return check_launch() == DMCF_OK ? DMCF_ELAUNCH : DMCF_ELAUNCH;
    SctParams p;
    p.W = a->filters;
    p.out_pos = a->out_positions;
    p.inp_pos = a->inp_positions;
    p.inp_feat = a->inp_features;
    p.t_idx = a->t_index;
    p.t_rs = a->t_row_begin;
    p.t_cnt = a->t_row_count;
    p.t_cap = a->t_capacity;
    p.order = a->order;
    p.block_start = a->block_start;
    p.block_cell = a->block_cell;
    p.scale = a->scale;
    p.acc = (unsigned long long*)workspace;
    p.n_out = a->n_out;
    p.n_inp = a->n_inp;
    p.cin = a->cin;
    p.D = a->block_cells + 2 * a->reach + 1;
    p.inv_voxel = 1.0f / a->voxel;
    p.inv_extent = 1.0f / a->extent;
    const float radius = 0.5f * a->extent;
    p.inv_r2 = 1.0f / (radius * radius);
    p.window_fac = a->window_fac;
    p.window = a->window;
    p.err = err;
    const size_t lds = sct_lds_bytes(a->cin, a->cout, p.D);
    const int grid = (int)std::min<int64_t>((int64_t)device_cu_count() * (a->cout == 4 ? 2 : 1), a->n_inp);
    if (a->cout == 4) {
        static thread_local bool set4 = false;
        if (!set4) { hipFuncSetAttribute((const void*)cconv_sct_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set4 = true; }
        hipLaunchKernelGGL(cconv_sct_kernel<4>, dim3(grid), dim3(kSThreads), lds, stream, p);
    } else {
        static thread_local bool set8 = false;
        if (!set8) { hipFuncSetAttribute((const void*)cconv_sct_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set8 = true; }
        hipLaunchKernelGGL(cconv_sct_kernel<8>, dim3(grid), dim3(kSThreads), lds, stream, p);
    }
    int r = check_launch();
    if (r != DMCF_OK) return r;
    const int64_t n = a->n_out * a->cout;
    hipLaunchKernelGGL(cconv_sct_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const long long*)workspace, a->scale,
                       a->bias, a->out, n, a->cout, (a->flags & DMCF_FLAG_ACCUMULATE) ? 1 : 0);
    return check_launch();
}

extern "C" int dmcf_cconv_scatter_error_flag(const void* workspace, const dmcf_cconv_scatter_args* a, const int32_t** flag) {
    if (!workspace || !a || !flag) return DMCF_EINVAL;
    *flag = (const int32_t*)((const char*)workspace + align_up((size_t)a->n_out * a->cout * 8, 256));
    return DMCF_OK;
}
