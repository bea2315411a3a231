// Continuous convolution (CConv) and antisymmetric CConv (ASCC) for gfx950.
//
// Replaces ml3d.ops.continuous_conv as called from the reference at utils/convolutions.py:414-431 and
// the ASCC body utils/convolutions.py:410-412,433-458 (see include/dmcf_hip.h for the contract).
//
// Formulation (same factorisation as the Open3D CPU code, which is what makes the op cheap):
//   B_i[k,c] = sum_{j in N(i)} a_ij * w_k(Lambda(x_j - x_i)) * f_j[c]        (gather + trilinear splat)
//   out_i[o] = sum_{k,c} B_i[k,c] * W[k,c,o]                                  (dense contraction)
// MI355X mapping:
//   * a workgroup (4 wavefronts) owns a tile of TM consecutive output points; B for the tile lives in
//     LDS as [TM][K*CC+1] floats (+1 pads the row stride off the 32-bank period), channel-chunked by
//     CC in {4,8} so that TM stays large enough to amortise the filter reads;
//   * splat: a wavefront walks one output point's CSR row 64 neighbours at a time.  Phase 1 has one
//     LANE PER NEIGHBOUR: index / position gather, window function, ball->cube mapping (sqrt, atan,
//     divisions) are paid once per pair per chunk at full lane utilisation.  Phase 2 re-maps lanes to
//     (corner, channel): the pair's parameters are broadcast with v_readlane (CC=8) and every lane
//     issues one ds_add_f32 into B -- 8*CC LDS adds per pair, no read-modify-write, no races between
//     the wavefronts that share a tile;
//   * contraction from LDS against the filter streamed through L1/L2, accumulators in registers
//     across channel chunks; normalisation, bias and the add_merge accumulation are fused into the
//     epilogue store;
//   * ASCC is the same kernel with pair features (f_j + f_i) and the mirrored kernel, i.e. the fused
//     single-pass form of the reference's two continuous_conv calls + batched matmul;
//   * consecutive tiles go to the same XCD (blockIdx swizzle) so neighbouring outputs share one L2.
#include "common.h"

namespace dmcf {

struct CconvParams {
    const float* W;  // [K][cin][cout] full kernel
    int sx, sy, sz, K, cin, cout;
    const float* out_pos;
    const float* inp_pos;
    const float* inp_feat;
    const float* inp_imp;
    const int32_t* idx;
    const int64_t* rs;
    const float* nval;
    int64_t n_out;
    float inv_extent, r2, window_fac;
    int window, mapping, interp, flags;
    const float* bias;
    float* out;
    int TM, KCp, ntiles, tiles_per_xcd;
};

// ---- per-pair math (float restatement of Open3D's CoordinateTransformation.h, see oracle/dmcf_oracle.c)
__device__ __forceinline__ void sphere_to_cyl(float& x, float& y, float& z) {
    const float sq_norm = x * x + y * y + z * z;
    const float norm = sqrtf(sq_norm);
    if (sq_norm < 1e-12f) {
        x = y = z = 0.0f;
    } else if (1.25f * z * z > (x * x + y * y)) {
        const float s = sqrtf(3.0f * norm / (norm + fabsf(z)));
        x *= s;
        y *= s;
        z = copysignf(norm, z);
    } else {
        const float s = norm / sqrtf(x * x + y * y);
        x *= s;
        y *= s;
        z *= 1.5f;
    }
}

__device__ __forceinline__ void cyl_to_cube(float& x, float& y) {
    const float sq_norm = x * x + y * y;
    const float norm = sqrtf(sq_norm);
    const float four_over_pi = 1.2732395447351628f;
    if (sq_norm < 1e-12f) {
        x = y = 0.0f;
    } else if (fabsf(y) <= fabsf(x)) {
        const float tmp = copysignf(norm, x);
        y = tmp * four_over_pi * atanf(y / x);
        x = tmp;
    } else {
        const float tmp = copysignf(norm, y);
        x = tmp * four_over_pi * atanf(x / y);
        y = tmp;
    }
}

__device__ __forceinline__ void filter_coords(float& x, float& y, float& z, const CconvParams& p) {
    if (p.mapping == DMCF_MAP_BALL_TO_CUBE_RADIAL) {
        const float s = 2.0f * p.inv_extent;
        x *= s; y *= s; z *= s;
        const float radius = sqrtf(x * x + y * y + z * z);
        const float abs_max = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        if (abs_max < 1e-8f) {
            x = y = z = 0.0f;
        } else {
            x *= 0.5f * radius / abs_max;
            y *= 0.5f * radius / abs_max;
            z *= 0.5f * radius / abs_max;
        }
    } else if (p.mapping == DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING) {
        const float s = 2.0f * p.inv_extent;
        x *= s; y *= s; z *= s;
        sphere_to_cyl(x, y, z);
        cyl_to_cube(x, y);
        x *= 0.5f; y *= 0.5f; z *= 0.5f;
    } else {
        x *= p.inv_extent; y *= p.inv_extent; z *= p.inv_extent;
    }
    if (p.flags & DMCF_FLAG_ALIGN_CORNERS) {
        x = (x + 0.5f) * (float)(p.sx - 1);
        y = (y + 0.5f) * (float)(p.sy - 1);
        z = (z + 0.5f) * (float)(p.sz - 1);
    } else {
        x = x * (float)p.sx + (float)(p.sx / 2);
        y = y * (float)p.sy + (float)(p.sy / 2);
        z = z * (float)p.sz + (float)(p.sz / 2);
        if (p.sx % 2 == 0) x -= 0.5f;
        if (p.sy % 2 == 0) y -= 0.5f;
        if (p.sz % 2 == 0) z -= 0.5f;
    }
    if (p.interp == DMCF_INTERP_LINEAR) {  // coordinate clamping
        x = fminf((float)(p.sx - 1), fmaxf(0.0f, x));
        y = fminf((float)(p.sy - 1), fmaxf(0.0f, y));
        z = fminf((float)(p.sz - 1), fmaxf(0.0f, z));
    }
}

// window functions of utils/tools/losses.py:8-44 on q = d^2 / R^2
__device__ __forceinline__ float window_value(int window, float v, float r2, float fac) {
    if (window == DMCF_WINDOW_NONE) return 1.0f;
    if (window == DMCF_WINDOW_EXPLICIT) return v;
    const float q = v / r2;
    switch (window) {
        case DMCF_WINDOW_POLY6: {
            const float t = 1.0f - q;
            return fac * fminf(fmaxf(t * t * t, 0.0f), 1.0f);
        }
        case DMCF_WINDOW_CUBIC: {
            const float s = sqrtf(q);
            float r = 0.0f;
            if (q <= 1.0f) r = (s <= 0.5f) ? 6.0f * (s * s * s - q) + 1.0f : 2.0f * (1.0f - s) * (1.0f - s) * (1.0f - s);
            return fac * 4.0f / 3.0f * r;
        }
        case DMCF_WINDOW_LINEAR: return fac * (1.0f - sqrtf(q));
        case DMCF_WINDOW_PEAK: return fac * (1.0f - 2.0f * sqrtf(q) + q);
        case DMCF_WINDOW_CUBIC_GRAD: {
            const float s = sqrtf(q);
            float r = 0.0f;
            if (q <= 1.0f) r = (s <= 0.5f) ? 18.0f * q - 12.0f * s : -6.0f * (1.0f - s) * (1.0f - s);
            return fac * 4.0f / 3.0f * r;
        }
    }
    return 1.0f;
}

// weight and filter cell of corner t (bit0 = x, bit1 = y, bit2 = z) for filter coordinates (x,y,z)
__device__ __forceinline__ void corner(int t, float x, float y, float z, const CconvParams& p, float& w, int& cell) {
    if (p.interp == DMCF_INTERP_NEAREST) {
        int xi = (int)roundf(x), yi = (int)roundf(y), zi = (int)roundf(z);
        xi = min(max(xi, 0), p.sx - 1);
        yi = min(max(yi, 0), p.sy - 1);
        zi = min(max(zi, 0), p.sz - 1);
        w = (t == 0) ? 1.0f : 0.0f;
        cell = (zi * p.sy + yi) * p.sx + xi;
        return;
    }
    const float xf = floorf(x), yf = floorf(y), zf = floorf(z);
    const float a = x - xf, b = y - yf, c = z - zf;
    int xi = (int)xf + (t & 1), yi = (int)yf + ((t >> 1) & 1), zi = (int)zf + ((t >> 2) & 1);
    w = ((t & 1) ? a : 1.0f - a) * ((t & 2) ? b : 1.0f - b) * ((t & 4) ? c : 1.0f - c);
    if (p.interp == DMCF_INTERP_LINEAR) {
        xi = min(xi, p.sx - 1);
        yi = min(yi, p.sy - 1);
        zi = min(zi, p.sz - 1);
    } else {  // LINEAR_BORDER: zero outside the filter array
        const bool inside = xi >= 0 && xi < p.sx && yi >= 0 && yi < p.sy && zi >= 0 && zi < p.sz;
        if (!inside) {
            w = 0.0f;
            xi = yi = zi = 0;
        }
    }
    cell = (zi * p.sy + yi) * p.sx + xi;
}

constexpr int kThreads = 256;
constexpr int kMaxAcc = 8;

template <int CC>
__global__ __launch_bounds__(kThreads) void cconv_kernel(const CconvParams p) {
    extern __shared__ float smem[];
    float* Bt = smem;                           // [TM][KCp]
    float* norm = smem + (size_t)p.TM * p.KCp;  // [TM]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware tile order: blocks b, b+8, b+16.. (same XCD) take consecutive tiles
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * p.TM;
    const int TM = p.TM, KCp = p.KCp, cin = p.cin, cout = p.cout;
    const bool symmetric = (p.flags & DMCF_FLAG_SYMMETRIC) != 0;
    constexpr int PP = 64 / (8 * CC);  // pairs per splat instruction

    // contraction thread mapping
    const int TPP = kThreads / TM;  // threads per output point (power of two, 8..64)
    const int cpt = tid / TPP, cog = tid % TPP;
    const bool ksplit = cout <= kMaxAcc;  // few outputs: split K across the point's threads instead
    float acc[kMaxAcc];
#pragma unroll
    for (int u = 0; u < kMaxAcc; ++u) acc[u] = 0.0f;

    if (tid < TM) norm[tid] = 0.0f;

    for (int c0 = 0; c0 < cin; c0 += CC) {
        for (int e = tid; e < TM * KCp; e += kThreads) Bt[e] = 0.0f;
        __syncthreads();
        // ---------------- splat ----------------
        for (int pt = wave; pt < TM; pt += kThreads / 64) {
            const int64_t i = pt0 + pt;
            if (i >= p.n_out) break;
            const int64_t rb = p.rs[i], re = p.rs[i + 1];
            const float ox = p.out_pos[3 * i], oy = p.out_pos[3 * i + 1], oz = p.out_pos[3 * i + 2];
            const int t = (lane / CC) & 7, c = lane % CC;
            const bool cvalid = (c0 + c) < cin;
            const float fi = (symmetric && cvalid) ? p.inp_feat[i * cin + c0 + c] : 0.0f;
            float* Brow = Bt + (size_t)pt * KCp;
            float nsum = 0.0f;
            for (int64_t pb = rb; pb < re; pb += 64) {
                const int npairs = (int)min((int64_t)64, re - pb);
                // phase 1: one lane per neighbour
                int j = 0;
                float a = 0.0f, x = 0.0f, y = 0.0f, z = 0.0f;
                if (lane < npairs) {
                    const int64_t pp = pb + lane;
                    j = p.idx[pp];
                    a = window_value(p.window, p.nval ? p.nval[pp] : 0.0f, p.r2, p.window_fac);
                    nsum += a;
                    if (p.inp_imp) a *= p.inp_imp[j];
                    x = p.inp_pos[3 * (int64_t)j] - ox;
                    y = p.inp_pos[3 * (int64_t)j + 1] - oy;
                    z = p.inp_pos[3 * (int64_t)j + 2] - oz;
                    filter_coords(x, y, z, p);
                }
                // phase 2: lanes = (pair slot, corner, channel)
                for (int q = 0; q < npairs; q += PP) {
                    int jq;
                    float aq, xq, yq, zq;
                    bool valid = true;
                    if constexpr (PP == 1) {
                        jq = __builtin_amdgcn_readlane(j, q);
                        aq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), q));
                        xq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), q));
                        yq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), q));
                        zq = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(z), q));
                    } else {
                        const int slot = q + lane / (8 * CC);
                        valid = slot < npairs;
                        jq = __shfl(j, slot, 64);
                        aq = __shfl(a, slot, 64);
                        xq = __shfl(x, slot, 64);
                        yq = __shfl(y, slot, 64);
                        zq = __shfl(z, slot, 64);
                    }
                    float w;
                    int cell;
                    corner(t, xq, yq, zq, p, w, cell);
                    if (valid && cvalid) {
                        float f = p.inp_feat[(int64_t)jq * cin + c0 + c];
                        if (symmetric) f += fi;
                        const float val = w * (f * aq);
                        __hip_atomic_fetch_add(&Brow[cell * CC + c], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            if (c0 == 0 && (p.flags & DMCF_FLAG_NORMALIZE)) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) nsum += __shfl_xor(nsum, d, 64);
                if (lane == 0) norm[pt] = nsum;
            }
        }
        __syncthreads();
        // ---------------- contraction of this channel chunk ----------------
        const float* Brow = Bt + (size_t)cpt * KCp;
        const int KC = p.K * CC;
        if (!ksplit) {
            for (int kc = 0; kc < KC; ++kc) {
                const int k = kc / CC, c = kc % CC;
                if (c0 + c >= cin) continue;
                const float b = Brow[kc];
                const float* Wrow = p.W + ((size_t)k * cin + c0 + c) * cout;
#pragma unroll
                for (int u = 0; u < kMaxAcc; ++u) {
                    const int o = cog + u * TPP;
                    if (o < cout) acc[u] = fmaf(b, Wrow[o], acc[u]);
                }
            }
        } else {
            for (int kc = cog; kc < KC; kc += TPP) {
                const int k = kc / CC, c = kc % CC;
                if (c0 + c >= cin) continue;
                const float b = Brow[kc];
                const float* Wrow = p.W + ((size_t)k * cin + c0 + c) * cout;
#pragma unroll
                for (int u = 0; u < kMaxAcc; ++u)
                    if (u < cout) acc[u] = fmaf(b, Wrow[u], acc[u]);
            }
        }
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    const int64_t i = pt0 + cpt;
    if (ksplit) {
#pragma unroll
        for (int u = 0; u < kMaxAcc; ++u)
            for (int d = TPP >> 1; d >= 1; d >>= 1) acc[u] += __shfl_xor(acc[u], d, 64);
    }
    if (i < p.n_out) {
        float inv_norm = 1.0f;
        if (p.flags & DMCF_FLAG_NORMALIZE) {
            const float nv = norm[cpt];
            inv_norm = nv != 0.0f ? 1.0f / nv : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < kMaxAcc; ++u) {
            const int o = ksplit ? u : cog + u * TPP;
            const bool mine = ksplit ? (cog == 0 && u < cout) : (o < cout);
            if (mine) {
                float v = acc[u];
                if (p.flags & DMCF_FLAG_NORMALIZE) v *= inv_norm;
                if (p.bias) v += p.bias[o];
                float* dst = p.out + i * cout + o;
                if (p.flags & DMCF_FLAG_ACCUMULATE) v += *dst;
                *dst = v;
            }
        }
    }
}

// full[z,y,x,c,o] = concat([-half[::-1,::-1,::-1], half], axis=sym_axis)   (utils/convolutions.py:410-412)
__global__ void mirror_kernel(const float* __restrict__ half, float* __restrict__ full, int d0, int d1, int d2,
                              int inner, int sym_axis) {
    // d0,d1,d2: FULL spatial dims (z,y,x); inner = cin*cout
    const int64_t total = (int64_t)d0 * d1 * d2 * inner;
    const int hd[3] = {sym_axis == 0 ? d0 / 2 : d0, sym_axis == 1 ? d1 / 2 : d1, sym_axis == 2 ? d2 / 2 : d2};
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int in = (int)(e % inner);
        int64_t s = e / inner;
        int c[3];
        c[2] = (int)(s % d2); s /= d2;
        c[1] = (int)(s % d1); s /= d1;
        c[0] = (int)s;
        const int h = hd[sym_axis];
        float sign = 1.0f;
        int hc[3] = {c[0], c[1], c[2]};
        if (c[sym_axis] >= h) {
            hc[sym_axis] = c[sym_axis] - h;
        } else {
            sign = -1.0f;
            for (int a = 0; a < 3; ++a) hc[a] = hd[a] - 1 - c[a];
        }
        full[e] = sign * half[(((int64_t)hc[0] * hd[1] + hc[1]) * hd[2] + hc[2]) * inner + in];
    }
}

struct LaunchCfg {
    int CC, TM;
    size_t lds;
};

static LaunchCfg choose_cfg(int K, int cin, int cout) {
    // LDS budget per workgroup: two workgroups per CU out of 160 KiB
    const size_t budget = 80 * 1024;
    LaunchCfg best = {4, 4, 0};
    double best_score = -1.0;
    const int ccs[2] = {8, 4};
    for (int ci = 0; ci < 2; ++ci) {
        const int CC = ccs[ci];
        if (CC == 8 && cin <= 4) continue;  // half of the splat lanes would idle
        for (int TM = 32; TM >= 4; TM >>= 1) {
            if (cout > kMaxAcc && (cout + (kThreads / TM) - 1) / (kThreads / TM) > kMaxAcc) continue;
            const size_t lds = ((size_t)TM * (K * CC + 1) + TM) * sizeof(float);
            if (lds > budget) continue;
            const double score = (double)TM * CC;  // filter reuse x fewer passes over the neighbour list
            if (score > best_score) {
                best_score = score;
                best = {CC, TM, lds};
            }
            break;
        }
    }
    if (best_score < 0) {
        // very large filters: one workgroup per CU, smallest tile
        best = {4, 4, ((size_t)4 * (K * 4 + 1) + 4) * sizeof(float)};
    }
    return best;
}

}  // namespace dmcf

using namespace dmcf;

extern "C" {

static int validate(const dmcf_cconv_args* a) {
    if (!a) return DMCF_EINVAL;
    for (int d = 0; d < 5; ++d)
        if (a->filter_dims[d] < 1) return DMCF_EINVAL;
    if (a->n_out < 0 || a->n_inp < 0) return DMCF_EINVAL;
    if (!(a->extent > 0.0f)) return DMCF_EINVAL;
    if (a->window < DMCF_WINDOW_NONE || a->window > DMCF_WINDOW_CUBIC_GRAD) return DMCF_EINVAL;
    if (a->coordinate_mapping < 0 || a->coordinate_mapping > 2) return DMCF_EINVAL;
    if (a->interpolation < 0 || a->interpolation > 2) return DMCF_EINVAL;
    if (a->flags & DMCF_FLAG_SYMMETRIC) {
        if (a->sym_axis < 0 || a->sym_axis > 2) return DMCF_EINVAL;
        if (a->n_inp != a->n_out) return DMCF_EINVAL;
    }
    if (a->filter_dims[4] > 64) return DMCF_EUNSUPPORTED;
    if (a->n_out > 0) {
        if (!a->filters || !a->out_positions || !a->neighbors_row_splits || !a->out) return DMCF_EINVAL;
        if (a->window != DMCF_WINDOW_NONE && !a->neighbors_value) {
            // legal only when there are no pairs at all; cannot know here, so require it
            return DMCF_EINVAL;
        }
    }
    return DMCF_OK;
}

size_t dmcf_cconv_workspace_bytes(const dmcf_cconv_args* a) {
    if (!a) return 0;
    size_t bytes = 256;
    if (a->flags & DMCF_FLAG_SYMMETRIC) {
        size_t full = 2;
        for (int d = 0; d < 5; ++d) full *= (size_t)(a->filter_dims[d] > 0 ? a->filter_dims[d] : 1);
        bytes += align_up(full * sizeof(float), 256);
    }
    return bytes;
}

int dmcf_cconv_forward(const dmcf_cconv_args* a, void* workspace, size_t workspace_bytes, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = validate(a);
    if (rc != DMCF_OK) return rc;
    if (a->n_out == 0) return DMCF_OK;
    if (workspace_bytes < dmcf_cconv_workspace_bytes(a)) return DMCF_EWORKSPACE;

    CconvParams p;
    int dz = a->filter_dims[0], dy = a->filter_dims[1], dx = a->filter_dims[2];
    p.cin = a->filter_dims[3];
    p.cout = a->filter_dims[4];
    p.W = a->filters;
    if (a->flags & DMCF_FLAG_SYMMETRIC) {
        if (!workspace || ((uintptr_t)workspace & 255)) return DMCF_EINVAL;
        if (a->sym_axis == 0) dz *= 2;
        if (a->sym_axis == 1) dy *= 2;
        if (a->sym_axis == 2) dx *= 2;
        float* full = (float*)workspace;
        const int inner = p.cin * p.cout;
        const int64_t total = (int64_t)dz * dy * dx * inner;
        const unsigned g = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(mirror_kernel, dim3(g < 4096u ? g : 4096u), dim3(256), 0, stream, a->filters, full, dz, dy,
                           dx, inner, a->sym_axis);
        p.W = full;
    }
    p.sx = dx; p.sy = dy; p.sz = dz;
    p.K = dx * dy * dz;
    p.out_pos = a->out_positions;
    p.inp_pos = a->inp_positions;
    p.inp_feat = a->inp_features;
    p.inp_imp = a->inp_importance;
    p.idx = a->neighbors_index;
    p.rs = a->neighbors_row_splits;
    p.nval = a->neighbors_value;
    p.n_out = a->n_out;
    p.inv_extent = 1.0f / a->extent;
    const float radius = 0.5f * a->extent;
    p.r2 = radius * radius;
    p.window_fac = a->window_fac;
    p.window = a->window;
    p.mapping = a->coordinate_mapping;
    p.interp = a->interpolation;
    p.flags = a->flags;
    p.bias = a->bias;
    p.out = a->out;

    const LaunchCfg cfg = choose_cfg(p.K, p.cin, p.cout);
    if (cfg.lds > 160 * 1024) return DMCF_EUNSUPPORTED;
    p.TM = cfg.TM;
    p.KCp = p.K * cfg.CC + 1;
    const int64_t ntiles = (a->n_out + cfg.TM - 1) / cfg.TM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    if (cfg.CC == 8) {
        if (cfg.lds > 64 * 1024)
            hipFuncSetAttribute((const void*)cconv_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds);
        hipLaunchKernelGGL((cconv_kernel<8>), dim3(grid), dim3(kThreads), cfg.lds, stream, p);
    } else {
        if (cfg.lds > 64 * 1024)
            hipFuncSetAttribute((const void*)cconv_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds);
        hipLaunchKernelGGL((cconv_kernel<4>), dim3(grid), dim3(kThreads), cfg.lds, stream, p);
    }
    return check_launch();
}

}  // extern "C"
