// CConv / ASCC in the DIRECT form for layers with at most 4 output channels (the ASCC output layer 32 -> 3 with its
// 6x6x6 filter, the 24 -> 4 down-sampling layer, every "-> 3" output convolution of the other models):
//
//     out_i[o] = sum_j sum_{8 corners t}  (a_ij w_t)  sum_c  f_j[c] W[cell_t, c, o]
//
// The splat + contraction factorisation of the other kernels pays for a dense B_i[K x Cin] tile per output point
// (zeroing, read-modify-write, a [K Cin] x [Cout] GEMM): for a 216-cell filter and ~30 neighbours that is twice
// the traffic of the pairs themselves, and with Cout <= 4 the contraction has nothing to amortise it
// (measured: the ASCC layer ran at 83 ps per pair and 8 channels against 22 ps for the big 4x4x4 layers).
// With Cout <= 4 the direct form is cheaper: 8 Cin Cout multiply-adds per pair and NO per-point tile.
//
// The whole (mirrored, for ASCC) filter lives in LDS as [z][y][c][x (padded)][4 outputs]; one persistent workgroup
// per CU loads it once and loops over tiles of output points.  A half-wave owns an output point; lane = input
// channel.  Per pair a lane reads the 8 corner rows W[cell_t, c, 0..3] (ds_read_b128, channel stride chosen odd
// in 16-byte units: conflict free) and does 8 + 8 Cout VALU operations; the neighbour's feature row is one
// coalesced 128-byte load per half-wave, issued four pairs ahead.  Geometry (gather, window, ball->cube map,
// trilinear weights) is computed one lane per pair for 32 pairs of each half at a time, as in cconv.hip, and
// parked in a small per-wave LDS area.  All interpolation / mapping modes are supported.
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {

constexpr int kDMaxWaves = 16;
constexpr int kDStage = 64 * 10;  // floats per wave: 64 pairs x {8 weights, base offset, index}
constexpr size_t kDLdsBudget = 160 * 1024;

struct DirectParams {
    CconvParams p;
    int sxp;          // padded x cells per (z, y, c) row: sx + pad with (sx + pad) odd
    int image_f4;     // float4 entries of the LDS image
    int nwaves;       // waves per workgroup
    int nwg;          // workgroups launched
};

// LDS image [z][y][c][x < sxp][4]: outputs beyond cout and the x padding are zero
__global__ void pack_direct(const float* __restrict__ src, f32x4* __restrict__ dst, int d0, int d1, int d2, int cin, int cout,
                            int sxp, int symmetric, int sym_axis) {
    const int64_t total = (int64_t)d0 * d1 * cin * sxp;
    const int hd[3] = {(symmetric && sym_axis == 0) ? d0 / 2 : d0, (symmetric && sym_axis == 1) ? d1 / 2 : d1,
                       (symmetric && sym_axis == 2) ? d2 / 2 : d2};
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = e;
        const int x = (int)(s % sxp); s /= sxp;
        const int c = (int)(s % cin); s /= cin;
        const int y = (int)(s % d1); s /= d1;
        const int z = (int)s;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (x < d2) {
            int c3[3] = {z, y, x};
            float sign = 1.0f;
            if (symmetric) {  // utils/convolutions.py:410-412: full kernel = concat(-reverse(k), k) along sym_axis
                const int hh = hd[sym_axis];
                if (c3[sym_axis] >= hh) {
                    c3[sym_axis] -= hh;
                } else {
                    sign = -1.0f;
                    for (int a = 0; a < 3; ++a) c3[a] = hd[a] - 1 - c3[a];
                }
            }
            const float* w = src + ((((int64_t)c3[0] * hd[1] + c3[1]) * hd[2] + c3[2]) * cin + c) * cout;
            for (int o = 0; o < cout; ++o) v[o] = sign * w[o];
        }
        dst[e] = v;
    }
}

template <int COUT, bool GENERIC>
__global__ __launch_bounds__(1024, 1) void cconv_direct_kernel(const DirectParams dp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const CconvParams& p = dp.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = dp.nwaves * 64;
    f32x4* Wl = (f32x4*)smem;
    float* stage = smem + 4 * (size_t)(dp.image_f4 + 1) + (size_t)wave * kDStage;  // +1: a zero entry past the end
    float* wst = stage;                                      // [64][8] corner weights
    int* bst = (int*)(stage + 64 * 8);                       // [64] {base offset (bytes)}
    int* jst = bst + 64;                                     // [64] neighbour index
    const int cin = p.cin;
    const bool symmetric = (p.flags & DMCF_FLAG_SYMMETRIC) != 0;

    for (int e = tid; e < dp.image_f4 + 1; e += nthreads)
        Wl[e] = e < dp.image_f4 ? ((const f32x4*)p.Wp)[e] : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    __syncthreads();

    const int h = lane >> 5, c = lane & 31;
    const bool c_ok = c < cin;
    // byte offsets of the four (y, z) corner rows of this lane's channel relative to a pair's base cell
    const int rowb = dp.sxp * 16;            // one (z, y, c) row
    const int yb = (p.sy >= 2 ? cin : 0) * rowb, zb = (p.sz >= 2 ? p.sy * cin : 0) * rowb;
    const int cb = (c_ok ? c : 0) * rowb;
    const char* Wb = (const char*)Wl;

    // XCD-aware persistent schedule: workgroups of one XCD walk neighbouring tiles
    const int per_xcd = dp.nwg / 8;
    const int wg = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
    const int pts_per_tile = 2 * dp.nwaves;
    for (int64_t tile = wg; tile < p.ntiles; tile += dp.nwg) {
        const int64_t i = tile * pts_per_tile + 2 * wave + h;
        const bool pt_valid = i < p.n_out;
        int64_t rb = 0, re = 0;
        float ox = 0.0f, oy = 0.0f, oz = 0.0f, fi = 0.0f;
        if (pt_valid) {
            rb = p.rs[i];
            re = p.cnt ? rb + p.cnt[i] : p.rs[i + 1];
            if (re > p.pair_cap) re = rb;
            ox = p.out_pos[3 * i]; oy = p.out_pos[3 * i + 1]; oz = p.out_pos[3 * i + 2];
            if (symmetric && c_ok) fi = p.inp_feat[i * cin + c];
        }
        const int cnt = (int)(re - rb);
        const int cnt_max = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
        const int nbatch = (cnt_max + 31) / 32;
        float acc[COUT];
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
        float nsum = 0.0f;

        auto load_idx = [&](int bi, int& j, float& nv, bool& valid) {
            const int64_t pp = rb + 32 * (int64_t)bi + c;
            valid = pp < re;
            j = 0;
            nv = 0.0f;
            if (valid) {
                j = p.idx[pp];
                if (p.nval) nv = p.nval[pp];
            }
        };
        auto gather = [&](int j, bool valid, float& px, float& py, float& pz) {
            px = py = pz = 0.0f;
            if (valid) {
                px = p.inp_pos[3 * (int64_t)j];
                py = p.inp_pos[3 * (int64_t)j + 1];
                pz = p.inp_pos[3 * (int64_t)j + 2];
            }
        };
        int jA, jB;
        float nvA, nvB, gx, gy, gz;
        bool vA, vB;
        load_idx(0, jA, nvA, vA);
        load_idx(1, jB, nvB, vB);
        gather(jA, vA, gx, gy, gz);
        for (int bi = 0; bi < nbatch; ++bi) {
            float nx, ny, nz;
            gather(jB, vB, nx, ny, nz);
            int jC;
            float nvC;
            bool vC;
            load_idx(bi + 2, jC, nvC, vC);
            // ---- phase 1: lane = pair (32 pairs of each half's point)
            {
                float a = 0.0f, x = 0.0f, y = 0.0f, z = 0.0f;
                if (vA) {
                    x = gx - ox;
                    y = gy - oy;
                    z = gz - oz;
                    a = window_value(p.window, p.nval ? nvA : rel_dist2(x, y, z), p.inv_r2, p.window_fac);
                    // the list holds the query points; the search this flag replaces drops every point AT the query position
                    // (frs.hip, ignore_query_point compares positions), so coincident particles go with the point itself
                    if ((p.flags & DMCF_FLAG_SKIP_SELF) && ((x == 0.0f && y == 0.0f && z == 0.0f) || jA == (int)i)) a = 0.0f;
                    nsum += a;
                    if (p.inp_imp) a *= p.inp_imp[jA];
                    filter_coords<GENERIC>(x, y, z, p);
                }
                int bx, by, bz;
                float wx0, wx1, wy0, wy1, wz0, wz1;
                if (GENERIC) {
                    axis_weights(x, p.sx, p.interp, bx, wx0, wx1);
                    axis_weights(y, p.sy, p.interp, by, wy0, wy1);
                    axis_weights(z, p.sz, p.interp, bz, wz0, wz1);
                } else {
                    axis_weights_linear(x, p.sx, bx, wx0, wx1);
                    axis_weights_linear(y, p.sy, by, wy0, wy1);
                    axis_weights_linear(z, p.sz, bz, wz0, wz1);
                }
                // corner weights in Open3D's product order (x-weight * y-weight) * z-weight, importance folded in
                const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
                const float z0 = wz0 * a, z1 = wz1 * a;
                float* wr = wst + lane * 8;
                const int wsw = (lane >> 2) & 1;  // the two float4 halves swap every 4 lanes: conflict-free stores
                *(f32x4*)(wr + 4 * wsw) = (f32x4){w00 * z0, w10 * z0, w01 * z0, w11 * z0};
                *(f32x4*)(wr + 4 * (wsw ^ 1)) = (f32x4){w00 * z1, w10 * z1, w01 * z1, w11 * z1};
                bst[lane] = ((bz * p.sy + by) * cin * dp.sxp + bx) * 16;
                jst[lane] = jA;
            }
            // ---- phase 2: lane = input channel of the half's point; the staged pairs one after the other
            int np_h = min(max(cnt - 32 * bi, 0), 32);
            const int nq = max(__builtin_amdgcn_readlane(np_h, 0), __builtin_amdgcn_readlane(np_h, 32));
            const float* wsrc = wst + (32 * h) * 8;
            const int* bsrc = bst + 32 * h;
            const int* jsrc = jst + 32 * h;
            const float* featc = p.inp_feat + (c_ok ? c : 0);
            // slots beyond a half's pair count hold zero weights and index 0: they add nothing
            float f0 = featc[(int64_t)jsrc[0] * cin], f1 = featc[(int64_t)jsrc[1] * cin];
            float f2 = featc[(int64_t)jsrc[2] * cin], f3 = featc[(int64_t)jsrc[3] * cin];
            auto pair_step = [&](int q, float fq) {
                const int sw = ((q >> 2) & 1) << 2;
                const f32x4 wa = *(const f32x4*)(wsrc + q * 8 + sw);        // corners 0..3 (z plane 0)
                const f32x4 wb = *(const f32x4*)(wsrc + q * 8 + (sw ^ 4));  // corners 4..7 (z plane 1)
                const float f = c_ok ? fq + fi : 0.0f;
                const char* a00 = Wb + bsrc[q] + cb;
                const char* a01 = a00 + yb;
                const char* a10 = a00 + zb;
                const char* a11 = a10 + yb;
                const f32x4 k0 = *(const f32x4*)a00, k1 = *(const f32x4*)(a00 + 16);
                const f32x4 k2 = *(const f32x4*)a01, k3 = *(const f32x4*)(a01 + 16);
                const f32x4 k4 = *(const f32x4*)a10, k5 = *(const f32x4*)(a10 + 16);
                const f32x4 k6 = *(const f32x4*)a11, k7 = *(const f32x4*)(a11 + 16);
                // out[o] += f * sum_t w_t k_t[o]: the corner sum first (its weights are the pair's, the same in every lane of the
                // half: packed multiply-adds take them straight out of the two loaded quads, low or high half by op_sel -- the
                // compiler's own v_pk_fma_f32 spent two v_mov per instruction on building operand pairs, 64 of 209 vector
                // instructions per four pairs), then ONE multiply-add per output with the feature: 19 instead of 40 per pair.
                f32x2 t01 = {0.0f, 0.0f}, t23 = {0.0f, 0.0f};
                const f32x2 wa01 = {wa.x, wa.y}, wa23 = {wa.z, wa.w}, wb01 = {wb.x, wb.y}, wb23 = {wb.z, wb.w};
#define DMCF_CORNER_LO(T, K, W) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(T) : "v"(K), "v"(W))
#define DMCF_CORNER_HI(T, K, W) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(T) : "v"(K), "v"(W))
                auto corner = [&](const f32x4& k, const f32x2& w, bool hi) {
                    const f32x2 kxy = {k.x, k.y}, kzw = {k.z, k.w};
                    if (COUT >= 2) {
                        if (hi) DMCF_CORNER_HI(t01, kxy, w); else DMCF_CORNER_LO(t01, kxy, w);
                    } else {
                        t01.x = fmaf(k.x, hi ? w.y : w.x, t01.x);
                    }
                    if (COUT == 4) {
                        if (hi) DMCF_CORNER_HI(t23, kzw, w); else DMCF_CORNER_LO(t23, kzw, w);
                    } else if (COUT == 3) {
                        t23.x = fmaf(k.z, hi ? w.y : w.x, t23.x);
                    }
                };
                corner(k0, wa01, false);
                corner(k1, wa01, true);
                corner(k2, wa23, false);
                corner(k3, wa23, true);
                corner(k4, wb01, false);
                corner(k5, wb01, true);
                corner(k6, wb23, false);
                corner(k7, wb23, true);
#undef DMCF_CORNER_LO
#undef DMCF_CORNER_HI
                acc[0] = fmaf(f, t01.x, acc[0]);
                if (COUT >= 2) acc[1] = fmaf(f, t01.y, acc[1]);
                if (COUT >= 3) acc[2] = fmaf(f, t23.x, acc[2]);
                if (COUT >= 4) acc[3] = fmaf(f, t23.y, acc[3]);
            };
            for (int q = 0; q < nq; q += 4) {
                // feature rows of the next four pairs (slots beyond 31 wrap: harmless duplicate loads)
                const float n0 = featc[(int64_t)jsrc[(q + 4) & 31] * cin], n1 = featc[(int64_t)jsrc[(q + 5) & 31] * cin];
                const float n2 = featc[(int64_t)jsrc[(q + 6) & 31] * cin], n3 = featc[(int64_t)jsrc[(q + 7) & 31] * cin];
                pair_step(q, f0);
                pair_step(q + 1, f1);
                pair_step(q + 2, f2);
                pair_step(q + 3, f3);
                f0 = n0; f1 = n1; f2 = n2; f3 = n3;
            }
            jA = jB; nvA = nvB; vA = vB;
            jB = jC; nvB = nvC; vB = vC;
            gx = nx; gy = ny; gz = nz;
        }
        // ---- reduce over the 32 channel lanes of the half, epilogue
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) {
#pragma unroll
            for (int o = 0; o < COUT; ++o) acc[o] += __shfl_xor(acc[o], d, 64);
            nsum += __shfl_xor(nsum, d, 64);
        }
        if (pt_valid && c < p.cout) {
            float v = acc[0];
#pragma unroll
            for (int o = 1; o < COUT; ++o) v = (c == o) ? acc[o] : v;
            if ((p.flags & DMCF_FLAG_NORMALIZE) && nsum != 0.0f) v /= nsum;
            if (p.bias) v += p.bias[c];
            float* dst = p.out + i * p.cout + c;
            if (p.flags & DMCF_FLAG_ACCUMULATE) v += *dst;
            *dst = v;
        }
    }
}

static int direct_cfg(int dz, int dy, int dx, int cin, DirectParams& dp) {
    dp.sxp = dx + ((dx & 1) ? 0 : 1);  // odd number of 16-byte cells per channel row: conflict-free ds_read_b128
    const int64_t image = (int64_t)dz * dy * cin * dp.sxp;
    if (image > (int64_t)(kDLdsBudget / 16)) return 0;
    dp.image_f4 = (int)image;
    const size_t left = kDLdsBudget - (size_t)(image + 1) * 16;
    int nw = (int)(left / (kDStage * sizeof(float)));
    nw = nw > kDMaxWaves ? kDMaxWaves : nw & ~3;
    dp.nwaves = nw;
    return nw >= 8;
}

bool cconv_direct_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "direct" forces it where it is possible, any other value disables it
    if (e && e[0] != 'd') return false;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if (cout > 4 || cin > 32) return false;
    DirectParams dp;
    if (!direct_cfg(dz, dy, dx, cin, dp)) return false;
    if (e) return true;
    // Measured on MI355X: ~75 ps per pair whatever the filter (LDS-read bound: 8 corners x Cin x 16 bytes per pair).
    // The splat kernels need 60 ps per pair for a 24-channel 4x4x4 layer but 300 ps for the 32-channel 6x6x6 ASCC
    // layer, whose dense per-point tile (216 cells) dominates: large filters come here, 4x4x4 ones stay there.
    return dz * dy * dx > 64 && cin >= 8;
}

size_t cconv_direct_packed_floats(int dz, int dy, int dx, int cin) {
    DirectParams dp;
    if (!direct_cfg(dz, dy, dx, cin, dp)) return 0;
    return (size_t)dp.image_f4 * 4;
}

int cconv_direct_launch(CconvParams p, const dmcf_cconv_args* a, int dz, int dy, int dx, void* workspace, hipStream_t stream) {
    DirectParams dp;
    if (!direct_cfg(dz, dy, dx, p.cin, dp)) return DMCF_EUNSUPPORTED;
    f32x4* image = (f32x4*)workspace;
    {
        const unsigned g = (unsigned)((dp.image_f4 + 255) / 256);
        if (!(a->flags & DMCF_FLAG_FILTER_PACKED))  // (else the workspace still holds it: dmcf_hip.h)
            hipLaunchKernelGGL(pack_direct, dim3(g < 1024u ? g : 1024u), dim3(256), 0, stream, a->filters, image, dz, dy, dx, p.cin,
                           p.cout, dp.sxp, (a->flags & DMCF_FLAG_SYMMETRIC) ? 1 : 0, a->sym_axis);
    }
    p.Wp = (const float*)image;
    const int pts = 2 * dp.nwaves;
    const int64_t ntiles = (p.n_out + pts - 1) / pts;
    if (ntiles > 0x7fffffff) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    const int ncu = device_cu_count();  // one persistent workgroup per CU
    int nwg = (ncu + 7) / 8 * 8;
    dp.nwg = nwg;
    dp.p = p;
    const size_t lds = (size_t)(dp.image_f4 + 1) * 16 + (size_t)dp.nwaves * kDStage * sizeof(float);
    const bool generic = !(a->coordinate_mapping == DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING &&
                           a->interpolation == DMCF_INTERP_LINEAR && (a->flags & DMCF_FLAG_ALIGN_CORNERS));
    const void* fn;
#define DMCF_PICK(G)                                                                                   \
    (p.cout == 1 ? (const void*)cconv_direct_kernel<1, G>                                              \
                 : (p.cout == 2 ? (const void*)cconv_direct_kernel<2, G>                               \
                                : (p.cout == 3 ? (const void*)cconv_direct_kernel<3, G> : (const void*)cconv_direct_kernel<4, G>)))
    fn = generic ? DMCF_PICK(true) : DMCF_PICK(false);
#undef DMCF_PICK
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&dp};
    e = hipLaunchKernel(fn, dim3(nwg), dim3(dp.nwaves * 64), kargs, lds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

}  // namespace dmcf
