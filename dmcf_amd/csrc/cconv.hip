// Continuous convolution (CConv) and antisymmetric CConv (ASCC) for gfx950.
//
// Replaces ml3d.ops.continuous_conv as called from the reference at utils/convolutions.py:414-431 and
// the ASCC body utils/convolutions.py:410-412,433-458 (see include/dmcf_hip.h for the contract).
//
// Formulation (same factorisation as the Open3D CPU code, which is what makes the op cheap):
//   B_i[k,c] = sum_{j in N(i)} a_ij * w_k(Lambda(x_j - x_i)) * f_j[c]        (gather + trilinear splat)
//   out_i[o] = sum_{k,c} B_i[k,c] * W[k,c,o]                                  (dense contraction)
//
// MI355X mapping (one workgroup = 8 wavefronts = a tile of 16 consecutive output points, two
// workgroups per CU so that one tile's LDS-bound splat overlaps the other's MFMA-bound contraction):
//   * B of the tile lives in LDS as [16][K*CC+4] floats, channel-chunked by CC (8, or 4 for Cin <= 4).
//   * splat, phase 1 -- ONE LANE PER NEIGHBOUR, 64 at a time: index, distance, position and the CC
//     feature floats of the neighbour are gathered with independent (batched) loads, the window
//     function and the ball->cube mapping (sqrt, atan, divisions) are evaluated once per pair at full
//     lane utilisation, and the 8 trilinear corner weights plus the importance-scaled features are
//     parked in a per-wave LDS staging area.
//   * splat, phase 2 -- lanes re-mapped to (corner, channel): per pair one v_readlane (base cell), two
//     LDS reads (corner weight, feature), one multiply and one ds_add_f32 into B.  A pair's 8 corners
//     are 8 distinct cells and a point's row of B belongs to one wave, so the adds never collide.
//   * contraction on the matrix cores in exact fp32: v_mfma_f32_16x16x4_f32, M = the 16 points,
//     N = 16 output channels per tile, K = the K*CC splat entries of the chunk split over the 8
//     waves; A fragments come from B with ds_read_b128, the filter is pre-packed (pack_filter) into
//     the B-fragment order so each lane fetches its 4 values with one coalesced 16-byte load from L2.
//     Accumulators stay in registers across channel chunks; the 8 partial tiles are reduced through
//     LDS, then normalisation, bias and the add_merge accumulation are applied in the store.
//   * ASCC is the same kernel with pair features (f_j + f_i) and the mirrored kernel, i.e. the fused
//     single-pass form of the reference's two continuous_conv calls + batched matmul.
//   * consecutive tiles go to the same XCD (blockIdx swizzle) so neighbouring outputs share one L2.
#include <stdlib.h>

#include <type_traits>

#include <stdio.h>

#include "cconv_common.h"

namespace dmcf {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int TM = 16;           // output points per workgroup (MFMA M)
constexpr int kMaxNT = 4;        // N tiles of 16 output channels (Cout <= 64)
constexpr int kWStride = 8;      // staged corner weights per pair
constexpr int kFStride = 8;      // staged features per pair
constexpr int kStage = 64 * (kWStride + kFStride + 1);  // floats per wave: weights, features, base offset per pair

// Layout of one row of B (floats): [z plane][y row][x cell][channel].  z planes are padded to "PS" floats so
// that the +z corners of a pair fall on other LDS banks than the -z corners (make_cfg picks the padding with a
// small bank model).  A rotation of odd y rows that also separates the +-y corners in the 32-bank write
// model was tried and removed: the extra v_readlane / bit-field work in the splat loop, which is co-limited
// by instruction issue and the LDS pipe, cost more (+48 % kernel time) than the conflicts it removed.

template <int CC, bool GENERIC>
__global__ __launch_bounds__(kThreads, 4) void cconv_kernel(const CconvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KCp = p.KCp, cin = p.cin, cout = p.cout, PS = p.PS;
    float* Bt = smem;                                   // [TM][KCp]
    float* norm = Bt + p.bfloats;                       // [TM]
    float* stage = norm + TM + (size_t)wave * kStage;
    float* wst = stage;                                 // [64][kWStride]
    float* fst = stage + 64 * kWStride;                 // [64][kFStride]
    int* bst = (int*)(fst + 64 * kFStride);             // [64] base cell offset of the pair (floats into its B row)
    float* deadbase = norm + TM + (size_t)kWaves * kStage;  // [kThreads][2], only if a filter axis is 1
    // XCD-aware tile order: blocks b, b+8, b+16.. (same XCD) take consecutive tiles
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * TM;
    const bool symmetric = (p.flags & DMCF_FLAG_SYMMETRIC) != 0;

    // Each wave owns two points of the tile, one per half-wave (h): rows wave and wave + 8 of B.
    const int h = lane >> 5, pl = lane & 31;
    const int pt = wave + kWaves * h;
    const int64_t i = pt0 + pt;
    const bool pt_valid = i < p.n_out;
    int64_t rb = 0, re = 0;
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (pt_valid) {
        rb = p.rs[i];
        re = p.cnt ? rb + p.cnt[i] : p.rs[i + 1];
        if (re > p.pair_cap) re = rb;
        ox = p.out_pos[3 * i]; oy = p.out_pos[3 * i + 1]; oz = p.out_pos[3 * i + 2];
    }
    const int cnt = (int)(re - rb);
    const int cnt_max = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
    const int nbatch = (cnt_max + 31) / 32;
    float* Brow = Bt + (size_t)pt * KCp;

    // phase-2 lane role inside a half-wave: corner x 4 channel groups.  The corner bits of the lane can be (x, Z, y)
    // instead of (x, y, z): a 64-bit LDS store is served in groups of 16 lanes = 4 corners against 32 banks, and with a 4-wide
    // filter the +y corner is 32 floats away (same banks, a 2-way conflict on every store) while the +z corner is one
    // padded plane away (make_cfg puts it 16 banks off): grouping (x, z) makes the stores conflict free.
    constexpr int CPL = CC / 4;  // channels per lane (2 -> 64-bit read-modify-write, 1 -> 32-bit)
    const int lt = pl >> 2, c4 = pl & 3;
    // staged-weight index (bit0 x, bit1 y, bit2 z) of this lane's corner; make_cfg picks the grouping per filter shape
    const int t = p.zgroup ? ((lt & 1) | ((lt & 4) >> 1) | ((lt & 2) << 1)) : lt;
    const int tx = (t & 1) && p.sx >= 2, ty = ((t >> 1) & 1) && p.sy >= 2, tz = ((t >> 2) & 1) && p.sz >= 2;
    const int lane_off = tz * PS + (ty * p.sx + tx) * CC + c4 * CPL;  // this corner's offset from the base cell
    // a "+1" corner along an axis of size 1 has weight 0 and would alias the base cell: such lanes stay idle
    const bool lane_live = !(((t & 1) && p.sx < 2) || ((t & 2) && p.sy < 2) || ((t & 4) && p.sz < 2));

    f32x4 acc[kMaxNT];
#pragma unroll
    for (int n = 0; n < kMaxNT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    if (tid < TM) norm[tid] = 0.0f;
    const int mi = lane & 15, mg = lane >> 4;  // MFMA roles: A row / B column index, k index

    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int c0 = chunk * CC;
        for (int e = tid * 4; e < TM * KCp; e += kThreads * 4) *(f32x4*)(Bt + e) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        __syncthreads();
        // ---------------- splat ----------------
        float fi[CC];
#pragma unroll
        for (int u = 0; u < CC; ++u)
            fi[u] = (symmetric && pt_valid && c0 + u < cin) ? p.inp_feat[i * cin + c0 + u] : 0.0f;
        float nsum = 0.0f;
        // Software pipeline over batches of 32 neighbours per half-wave: the (index, distance) loads run two
        // batches ahead and the dependent (position, feature) gathers one batch ahead of the splat that
        // consumes them, so the ~2 us index -> gather latency chain overlaps the LDS-bound phase 2.
        auto load_idx = [&](int bi, int& j, f32x4& g, int& gb, bool& valid) {
            const int64_t pp = rb + 32 * (int64_t)bi + pl;
            valid = pp < re;
            j = 0;
            gb = 0;
            g = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (valid) {
                j = p.idx[pp];
                if (p.nval) g.w = p.nval[pp];
            }
        };
        auto gather = [&](int j, bool valid, float& px, float& py, float& pz, float (&f)[CC]) {
            px = py = pz = 0.0f;
#pragma unroll
            for (int u = 0; u < CC; ++u) f[u] = 0.0f;
            if (valid) {
                const float* fp = p.inp_feat + (int64_t)j * cin + c0;
                if ((cin & 3) == 0) {  // rows are 16-byte aligned: vector gathers
#pragma unroll
                    for (int u = 0; u < CC; u += 4) {
                        if (c0 + u < cin) {
                            const f32x4 v = *(const f32x4*)(fp + u);
                            f[u] = v.x; f[u + 1] = v.y; f[u + 2] = v.z; f[u + 3] = v.w;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < CC; ++u)
                        if (c0 + u < cin) f[u] = fp[u];
                }
                px = p.inp_pos[3 * (int64_t)j];
                py = p.inp_pos[3 * (int64_t)j + 1];
                pz = p.inp_pos[3 * (int64_t)j + 2];
            }
        };
        int jA, jB, gbA, gbB;
        f32x4 gA, gB;
        bool vA, vB;
        load_idx(0, jA, gA, gbA, vA);
        load_idx(1, jB, gB, gbB, vB);
        float gx, gy, gz, gf[CC];
        gather(jA, vA, gx, gy, gz, gf);
        for (int bi = 0; bi < nbatch; ++bi) {
            // issue the loads of the following batches first
            float nx, ny, nz, nf[CC];
            gather(jB, vB, nx, ny, nz, nf);
            int jC, gbC;
            f32x4 gC;
            bool vC;
            load_idx(bi + 2, jC, gC, gbC, vC);
            // ---- phase 1: one lane per neighbour, 32 neighbours of each of the wave's two points
            int np_h = cnt - 32 * bi;  // pairs of this half in the batch (may be <= 0)
            np_h = min(max(np_h, 0), 32);
            int base = 0;
            {
                float a = 0.0f;
                int bx = 0, by = 0, bz = 0;
                float wx0 = 1.0f, wx1 = 0.0f, wy0 = 1.0f, wy1 = 0.0f, wz0 = 1.0f, wz1 = 0.0f;
                float x = 0.0f, y = 0.0f, z = 0.0f;
                if (vA) {
                    x = gx - ox;
                    y = gy - oy;
                    z = gz - oz;
                    a = window_value(p.window, p.nval ? gA.w : rel_dist2(x, y, z), p.inv_r2, p.window_fac);
                    nsum += a;
                    if (p.inp_imp) a *= p.inp_imp[jA];
                    filter_coords<GENERIC>(x, y, z, p);
                }
                if (GENERIC) {
                    axis_weights(x, p.sx, p.interp, bx, wx0, wx1);
                    axis_weights(y, p.sy, p.interp, by, wy0, wy1);
                    axis_weights(z, p.sz, p.interp, bz, wz0, wz1);
                } else {
                    axis_weights_linear(x, p.sx, bx, wx0, wx1);
                    axis_weights_linear(y, p.sy, by, wy0, wy1);
                    axis_weights_linear(z, p.sz, bz, wz0, wz1);
                }
            
                base = bz * PS + (by * p.sx + bx) * CC;
                bst[lane] = base;
                // corner weights in Open3D's product order (x-weight * y-weight) * z-weight
                const float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
                // the two float4 halves of a pair's weights swap places every 4 lanes: conflict-free b128 stores
                float* wr = wst + lane * kWStride;
                const int wsw = (lane >> 2) & 1;
                *(f32x4*)(wr + 4 * wsw) = (f32x4){w00 * wz0, w10 * wz0, w01 * wz0, w11 * wz0};
                *(f32x4*)(wr + 4 * (wsw ^ 1)) = (f32x4){w00 * wz1, w10 * wz1, w01 * wz1, w11 * wz1};
                float* fr = fst + lane * kFStride;
                if (symmetric && vA) {
#pragma unroll
                    for (int u = 0; u < CC; ++u) gf[u] += fi[u];
                }
#pragma unroll
                for (int u = 0; u < CC; u += 4)  // same half swap as the weights (CC = 8): conflict-free b128 stores
                    *(f32x4*)(fr + (CC == 8 ? (u ^ (4 * wsw)) : u)) = (f32x4){gf[u] * a, gf[u + 1] * a, gf[u + 2] * a, gf[u + 3] * a};
            }
            // The staging area is private to this wave and LDS operations of one wave are processed in
            // order, so no barrier is needed between the phases.
            // ---- phase 2: per half-wave lanes = (corner, channel group); both points advance together.
            // Plain read-modify-write instead of ds_add_f32: the LDS float atomic retires ~1 lane per
            // 3 clocks on gfx950 (measured ~200 clk per 64-lane instruction).  Race free: a row of B
            // belongs to one half-wave, the active lanes of a half hit distinct addresses (8 distinct
            // cells x channels; lanes whose "+1" cell collapses onto the base cell because that filter
            // axis has size 1 carry weight 0 and are parked on a private slot), and the two halves work
            // on two rows.  Branch-free body: slots beyond a half's pair count hold zero features
            // (phase 1 wrote f*a = 0 and a valid base cell for them), so they add 0.
            const int nq = max(__builtin_amdgcn_readlane(np_h, 0), __builtin_amdgcn_readlane(np_h, 32));
            // Groups of 8 slots, fully unrolled: the half-swap of the staging layout has period 8, so every staging
            // address is a loop-carried lane pointer + an immediate; the base cell comes from the staging area with
            // one broadcast read (it used to take 2 v_readlane + 2 v_mov + v_cndmask per pair of pairs) and the only
            // VALU work left per iteration is one address add, the multiply and the add of the read-modify-write.
            // Slots beyond a half's pair count hold zero features and a valid base cell: a group may run past nq.
            const float* wq0 = wst + (32 * h) * kWStride + t;        // slots with (q >> 2) even
            const float* wq1 = wst + (32 * h) * kWStride + (t ^ 4);  // ... odd
            const float* fq0 = fst + (32 * h) * kFStride + (CPL == 2 ? c4 * 2 : c4);
            const float* fq1 = fst + (32 * h) * kFStride + (CPL == 2 ? ((c4 * 2) ^ 4) : c4);
            const int* bq = bst + 32 * h;
            float* const brow = Brow + lane_off;
            float* const dead = deadbase + 2 * tid;
            auto splat8 = [&](auto all_live_tag) {
                constexpr bool ALL_LIVE = decltype(all_live_tag)::value;
                for (int q0 = 0; q0 < nq; q0 += 8) {
#pragma unroll
                    for (int g = 0; g < 2; ++g) {  // g = (q >> 2) & 1 selects the swapped staging halves
                        const float* wq = g ? wq1 : wq0;
                        const float* fq = g ? fq1 : fq0;
                        // staging reads of four pairs first (they never alias B), then the four dependent read-modify-writes
                        float w[4];
                        int boff[4];
                        f32x2 fv2[4];
                        float fv1[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int q = q0 + 4 * g + u;
                            w[u] = wq[q * kWStride];
                            boff[u] = bq[q];
                            if constexpr (CPL == 2)
                                fv2[u] = *(const f32x2*)(fq + q * kFStride);
                            else
                                fv1[u] = fq0[q * kFStride];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float* d = (ALL_LIVE || lane_live) ? brow + boff[u] : dead;
                            if constexpr (CPL == 2) {
                                f32x2* dst = (f32x2*)d;
                                f32x2 o = *dst;
                                o.x += w[u] * fv2[u].x;
                                o.y += w[u] * fv2[u].y;
                                *dst = o;
                            } else {
                                *d = *d + w[u] * fv1[u];
                            }
                        }
                    }
                }
            };
            if (p.sx >= 2 && p.sy >= 2 && p.sz >= 2)  // 3-D filters: every corner lane is live
                splat8(std::true_type{});
            else
                splat8(std::false_type{});
            // rotate the pipeline registers
            jA = jB; gA = gB; gbA = gbB; vA = vB;
            jB = jC; gB = gC; gbB = gbC; vB = vC;
            gx = nx; gy = ny; gz = nz;
#pragma unroll
            for (int u = 0; u < CC; ++u) gf[u] = nf[u];
        }
        if (chunk == 0 && (p.flags & DMCF_FLAG_NORMALIZE)) {
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) nsum += __shfl_xor(nsum, d, 64);
            if (pl == 0 && pt_valid) norm[pt] = nsum;
        }
        __syncthreads();
        // ---------------- contraction of this channel chunk on the matrix cores ----------------
        // out[16 x 16*NT] += B[16 x KC] * Wp_chunk[KC x 16*NT], k blocks of 16 dealt round-robin to the waves
        const float* Wc = p.Wp + (size_t)chunk * p.nblocks * (4 * p.NT * 16 * 4);
        for (int blk = wave; blk < p.nblocks; blk += kWaves) {
            const f32x4 av = *(const f32x4*)(Bt + (size_t)mi * KCp + blk * 16 + mg * 4);
            const float* wb = Wc + ((size_t)(blk * 4 + mg) * p.NT * 16 + mi) * 4;
#pragma unroll
            for (int n = 0; n < kMaxNT; ++n) {
                if (n < p.NT) {
                    const f32x4 bv = *(const f32x4*)(wb + n * 64);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[n], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---------------- cross-wave reduction + epilogue ----------------
    // D layout of 16x16x4: lane l, reg r -> row (point) 4*(l>>4)+r, column (channel) l&15
    float* red = Bt;  // [kWaves][TM][16*NT]  (B is dead now; 8*16*64*4 = 32 KiB at most)
    const int ncol = 16 * p.NT;
#pragma unroll
    for (int n = 0; n < kMaxNT; ++n) {
        if (n < p.NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                red[((size_t)wave * TM + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < TM * cout; e += kThreads) {
        const int ptt = e / cout, o = e % cout;
        const int64_t ii = pt0 + ptt;
        if (ii >= p.n_out) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += red[((size_t)w * TM + ptt) * ncol + o];
        if (p.flags & DMCF_FLAG_NORMALIZE) {
            const float nv = norm[ptt];
            if (nv != 0.0f) v /= nv;
        }
        if (p.bias) v += p.bias[o];
        float* dst = p.out + ii * cout + o;
        if (p.flags & DMCF_FLAG_ACCUMULATE) v += *dst;
        *dst = v;
    }
}

// Packs [K][cin][cout] (optionally mirrored: ASCC, utils/convolutions.py:410-412) into the B-fragment
// order of v_mfma_f32_16x16x4_f32 per channel chunk, following the padded row layout of B:
//   Wp[chunk][blk][g][n][j][q] = W[cell][c0 + cc][16 n + j],  kc' = 16 blk + 4 g + q,
//   plane z = kc' / PS, r = kc' % PS, cell = z*sx*sy + r / CC, cc = r % CC   (r < sx*sy*CC)
// zero for padding entries, c0+cc >= cin or 16n+j >= cout.
__global__ void pack_filter(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int d2, int cin,
                            int cout, int CC, int PS, int nchunks, int nblocks, int NT, int symmetric, int sym_axis) {
    const int PR = d1 * d2 * CC;
    const int64_t total = (int64_t)nchunks * nblocks * 4 * NT * 16 * 4;
    const int hd[3] = {(symmetric && sym_axis == 0) ? d0 / 2 : d0, (symmetric && sym_axis == 1) ? d1 / 2 : d1,
                       (symmetric && sym_axis == 2) ? d2 / 2 : d2};
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = e;
        const int q = (int)(s & 3); s >>= 2;
        const int j = (int)(s & 15); s >>= 4;
        const int n = (int)(s % NT); s /= NT;
        const int g = (int)(s & 3); s >>= 2;
        const int blk = (int)(s % nblocks); s /= nblocks;
        const int chunk = (int)s;
        const int kc = 16 * blk + 4 * g + q, cz = kc / PS, r = kc % PS;
        const int ci = chunk * CC + r % CC, o = 16 * n + j;
        float v = 0.0f;
        if (cz < d0 && r < PR && ci < cin && o < cout) {
            const int cp = r / CC;
            int c3[3] = {cz, cp / d2, cp % d2};
            float sign = 1.0f;
            if (symmetric) {
                const int hh = hd[sym_axis];
                if (c3[sym_axis] >= hh) {
                    c3[sym_axis] -= hh;
                } else {
                    sign = -1.0f;
                    for (int a = 0; a < 3; ++a) c3[a] = hd[a] - 1 - c3[a];
                }
            }
            v = sign * src[((((int64_t)c3[0] * hd[1] + c3[1]) * hd[2] + c3[2]) * cin + ci) * cout + o];
        }
        dst[e] = v;
    }
}

struct LaunchCfg {
    int CC, PS, KCp, nblocks, NT, nchunks, zgroup;
    size_t lds, packed_floats, bfloats;
};

static LaunchCfg make_cfg(int sx, int sy, int sz, int cin, int cout) {
    LaunchCfg c;
    c.CC = cin <= 4 ? 4 : 8;
    const int PR = sx * sy * c.CC;
    // Plane stride: the smallest padding of the z plane for which the 8 corners of a pair are conflict free
    // in the LDS bank models of the splat's accesses (CC = 8: 64-bit reads see 64 banks over the 8 corners of
    // a half-wave, 64-bit writes 32 banks over 4 corners at a time; CC = 4: 32-bit accesses, 32 banks over all
    // 8 corners), while one workgroup's B tile still fits.
    auto conflicts = [&](int PS, int zgroup) {
        int bad = 0;
        for (int by = 0; by < 2; ++by)
            for (int bx = 0; bx < 2; ++bx) {
                int off[8];
                for (int t = 0; t < 8; ++t) {
                    const int tx = (t & 1) && sx >= 2, ty = ((t >> 1) & 1) && sy >= 2, tz = ((t >> 2) & 1) && sz >= 2;
                    off[t] = tz * PS + ((by + ty) * sx + bx + tx) * c.CC;
                }
                for (int a = 0; a < 8; ++a)
                    for (int b = a + 1; b < 8; ++b) {
                        if (off[a] == off[b]) continue;  // collapsed corners are masked off in the kernel
                        const int rd = c.CC == 8 ? 64 : 32;
                        if ((off[a] % rd) / c.CC == (off[b] % rd) / c.CC) ++bad;                        // read model
                        // write model: a 16-lane store group holds the 4 corners with equal z bit, or (zgroup) equal y bit
                        const int ga = zgroup ? (a >> 1) & 1 : a >> 2, gb = zgroup ? (b >> 1) & 1 : b >> 2;
                        if (ga == gb && (off[a] % 32) / c.CC == (off[b] % 32) / c.CC) ++bad;
                    }
            }
        return bad;
    };
    const bool has_dead = sx < 2 || sy < 2 || sz < 2;
    auto lds_bytes = [&](int PS) {
        const int KCx = (sz * PS + 15) / 16 * 16;
        const size_t kcp = KCx + ((4 - KCx % 64) + 64) % 64;
        size_t bf = (size_t)TM * kcp;
        const size_t rf = (size_t)kWaves * TM * 16 * ((cout + 15) / 16);
        if (bf < rf) bf = rf;
        return (bf + TM + (size_t)kWaves * kStage + (has_dead ? 2 * kThreads : 0)) * sizeof(float);
    };
    const int PS0 = (PR + 7) / 8 * 8;
    const size_t step = lds_bytes(PS0) <= 80 * 1024 ? 80 * 1024 : 160 * 1024;  // keep the occupancy step of the unpadded tile
    int best = PS0, best_bad = 1 << 30, best_group = 0;
    for (int zgroup = 0; zgroup < 2 && best_bad > 0; ++zgroup)
        for (int k = 0; k < 12; ++k) {
            const int PS = PS0 + 8 * k;
            if (k > 0 && (sz < 2 || lds_bytes(PS) > step)) break;
            const int bad = conflicts(PS, zgroup);
            if (bad < best_bad) { best_bad = bad; best = PS; best_group = zgroup; }
            if (bad == 0) break;
        }
    c.PS = best;
    c.zgroup = best_group;
    const int KC = (sz * c.PS + 15) / 16 * 16;  // multiple of 16 (k blocks of the contraction)
    c.nblocks = KC / 16;
    // row stride == 4 (mod 64) floats: the 16 rows read by one ds_read_b128 of the contraction spread over all banks
    c.KCp = KC + ((4 - KC % 64) + 64) % 64;
    c.NT = (cout + 15) / 16;
    c.nchunks = (cin + c.CC - 1) / c.CC;
    size_t b_floats = (size_t)TM * c.KCp;
    const size_t red_floats = (size_t)kWaves * TM * 16 * c.NT;
    if (b_floats < red_floats) b_floats = red_floats;  // the reduction buffer reuses B
    c.bfloats = b_floats;
    c.lds = lds_bytes(c.PS);
    c.packed_floats = (size_t)c.nchunks * c.nblocks * 4 * c.NT * 16 * 4;
    return c;
}

}  // namespace dmcf

using namespace dmcf;

extern "C" {

static int validate(const dmcf_cconv_args* a, bool forward = true) {
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
        if (a->n_inp < a->n_out) return DMCF_EINVAL;  // out points are inp points 0..n_out-1
    }
    if (a->filter_dims[4] > 16 * kMaxNT) return DMCF_EUNSUPPORTED;
    if (a->n_out > 0) {
        if (!a->out_positions || !a->neighbors_row_splits) return DMCF_EINVAL;
        if (forward && (!a->filters || !a->out || !a->inp_features)) return DMCF_EINVAL;
        if (a->window == DMCF_WINDOW_EXPLICIT && !a->neighbors_value) return DMCF_EINVAL;
    }
    return DMCF_OK;
}

static bool specialised(const dmcf_cconv_args* a) {
    return a->coordinate_mapping == DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING && a->interpolation == DMCF_INTERP_LINEAR &&
           (a->flags & DMCF_FLAG_ALIGN_CORNERS);
}

static void full_dims(const dmcf_cconv_args* a, int& dz, int& dy, int& dx) {
    dz = a->filter_dims[0]; dy = a->filter_dims[1]; dx = a->filter_dims[2];
    if (a->flags & DMCF_FLAG_SYMMETRIC) {
        if (a->sym_axis == 0) dz *= 2;
        if (a->sym_axis == 1) dy *= 2;
        if (a->sym_axis == 2) dx *= 2;
    }
}

size_t dmcf_cconv_workspace_bytes(const dmcf_cconv_args* a) {
    if (!a || validate(a) != DMCF_OK) return 256;
    int dz, dy, dx;
    full_dims(a, dz, dy, dx);
    const LaunchCfg cfg = make_cfg(dx, dy, dz, a->filter_dims[3], a->filter_dims[4]);
    size_t floats = cfg.packed_floats;
    size_t mf = cconv_mfma_packed_floats(dz * dy * dx, a->filter_dims[3], a->filter_dims[4]);
    if (cconv_mfma_eligible(dz * dy * dx, a->filter_dims[3], a->filter_dims[4]))  // (+ the chunks' partial sums of a small launch)
        mf = align_up(mf, 64) + cconv_mfma_partial_floats(dz * dy * dx, a->filter_dims[3], a->filter_dims[4], a->n_out);
    if (floats < mf) floats = mf;
    const size_t bf = cconv_blk_packed_floats(a->filter_dims[3], a->filter_dims[4]);
    if (floats < bf) floats = bf;
    const size_t cf = cconv_cls_packed_floats(a->filter_dims[3], a->filter_dims[4]);
    if (floats < cf) floats = cf;
    const size_t df = cconv_direct_packed_floats(dz, dy, dx, a->filter_dims[3]);
    if (floats < df) floats = df;
    return 256 + align_up(floats * sizeof(float), 256);
}

int dmcf_cconv_forward(const dmcf_cconv_args* a, void* workspace, size_t workspace_bytes, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    int rc = validate(a);
    if (rc != DMCF_OK) return rc;
    if (a->n_out == 0) return DMCF_OK;
    if (!workspace || ((uintptr_t)workspace & 255)) return DMCF_EINVAL;
    if (workspace_bytes < dmcf_cconv_workspace_bytes(a)) return DMCF_EWORKSPACE;

    CconvParams p;
    p.partial = nullptr;
    p.csplit = 0;
    int dz, dy, dx;
    full_dims(a, dz, dy, dx);
    p.cin = a->filter_dims[3];
    p.cout = a->filter_dims[4];
    p.sx = dx; p.sy = dy; p.sz = dz;
    p.K = dx * dy * dz;
    p.out_pos = a->out_positions;
    p.inp_pos = a->inp_positions;
    p.inp_feat = a->inp_features;
    p.inp_imp = a->inp_importance;
    p.idx = a->neighbors_index;
    p.rs = a->neighbors_row_splits;
    p.cnt = a->neighbors_row_count;
    // (the hint covers input channels < 32 and output channels < 64: wider layers multiply every block)
    p.wmask = (a->filter_tile_mask && a->filter_dims[3] <= 32 && a->filter_dims[4] <= 64) ? a->filter_tile_mask : 0xffffffffu;
    p.nval = a->neighbors_value;
    p.n_out = a->n_out;
    p.n_inp = a->n_inp;
    p.pair_cap = a->n_pairs;
    p.inv_extent = 1.0f / a->extent;
    const float radius = 0.5f * a->extent;
    p.inv_r2 = 1.0f / (radius * radius);
    p.window_fac = a->window_fac;
    p.window = a->window;
    p.mapping = a->coordinate_mapping;
    p.interp = a->interpolation;
    p.flags = a->flags;
    p.bias = a->bias;
    p.out = a->out;
    if (cconv_direct_eligible(a, dz, dy, dx)) return cconv_direct_launch(p, a, dz, dy, dx, workspace, stream);
    if (a->flags & DMCF_FLAG_SKIP_SELF) return DMCF_EUNSUPPORTED;  // (only the direct form tests the index against the row)
    if (cconv_ws_eligible(a, dz, dy, dx)) return cconv_ws_launch(p, a, workspace, stream);
    if (cconv_pair_eligible(a, dz, dy, dx)) return cconv_pair_launch(p, a, workspace, stream);
    if (cconv_p16_eligible(a, dz, dy, dx)) return cconv_p16_launch(p, a, workspace, stream);
    if (cconv_z3_eligible(a, dz, dy, dx)) return cconv_z3_launch(p, a, workspace, stream);
    if (cconv_cls_eligible(a, dz, dy, dx)) return cconv_cls_launch(p, a, workspace, stream);
    if (cconv_blk_eligible(a, dz, dy, dx)) return cconv_blk_launch(p, a, workspace, stream);
    if (cconv_mfma_eligible(p.K, p.cin, p.cout)) return cconv_mfma_launch(p, a, dz, dy, dx, workspace, stream);
    // the generic LDS-splat kernel: its own filter packing and LDS budget (every specialised kernel above packs its own
    // layout into the same workspace and has its own limits)
    const LaunchCfg cfg = make_cfg(dx, dy, dz, p.cin, p.cout);
    if (cfg.lds > 160 * 1024) return DMCF_EUNSUPPORTED;
    {
        float* packed = (float*)workspace;
        const int64_t total = (int64_t)cfg.packed_floats;
        const unsigned g = (unsigned)((total + 255) / 256);
        if (!(a->flags & DMCF_FLAG_FILTER_PACKED))  // (else the workspace still holds it: dmcf_hip.h)
            hipLaunchKernelGGL(pack_filter, dim3(g < 2048u ? g : 2048u), dim3(256), 0, stream, a->filters, packed, dz, dy, dx,
                           p.cin, p.cout, cfg.CC, cfg.PS, cfg.nchunks, cfg.nblocks, cfg.NT,
                           (a->flags & DMCF_FLAG_SYMMETRIC) ? 1 : 0, a->sym_axis);
        p.Wp = packed;
    }
    p.KCp = cfg.KCp;
    p.PS = cfg.PS;
    p.zgroup = cfg.zgroup;
    p.nblocks = cfg.nblocks;
    p.NT = cfg.NT;
    p.nchunks = cfg.nchunks;
    p.bfloats = (int)cfg.bfloats;
    const int64_t ntiles = (a->n_out + TM - 1) / TM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    // the flag set every DMCF model uses (models/pbf_model.py:210-221) gets a specialised instantiation
    const bool generic = !specialised(a);
    const void* fn;
    if (cfg.CC == 8)
        fn = generic ? (const void*)cconv_kernel<8, true> : (const void*)cconv_kernel<8, false>;
    else
        fn = generic ? (const void*)cconv_kernel<4, true> : (const void*)cconv_kernel<4, false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return DMCF_ELAUNCH; }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(kThreads), kargs, cfg.lds, stream);
    if (e != hipSuccess) { g_last_hip_error = (int)e; return DMCF_ELAUNCH; }
    return check_launch();
}

int dmcf_cconv_kernel_name(const dmcf_cconv_args* a, char* name, size_t name_bytes) {
    if (!name || name_bytes < 2) return DMCF_EINVAL;
    int rc = validate(a, false);
    if (rc != DMCF_OK) return rc;
    int dz, dy, dx;
    full_dims(a, dz, dy, dx);
    const int cin = a->filter_dims[3], cout = a->filter_dims[4], NT = (cout + 15) / 16;
    const int ntt = NT <= 1 ? 1 : (NT <= 2 ? 2 : 4);
    const bool sym = (a->flags & DMCF_FLAG_SYMMETRIC) != 0;
    if (cconv_direct_eligible(a, dz, dy, dx))
        snprintf(name, name_bytes, "cconv_direct_kernel<%d, %s>", cout, specialised(a) ? "false" : "true");
    else if (cconv_ws_eligible(a, dz, dy, dx))
        snprintf(name, name_bytes, "cconv_ws_kernel<%d, %s>", ntt, cconv_plain(a) ? "true" : "false");
    else if (cconv_pair_eligible(a, dz, dy, dx))
        snprintf(name, name_bytes, "cconv_pair_kernel<%d, %s>", ntt, cconv_plain(a) ? "true" : "false");
    else if (cconv_p16_eligible(a, dz, dy, dx))
        snprintf(name, name_bytes, "cconv_p16_kernel<%d, %s>", ntt, cconv_plain(a) ? "true" : "false");
    else if (cconv_z3_eligible(a, dz, dy, dx))
        snprintf(name, name_bytes, "cconv_z3_kernel<%d, %s>", ntt, cconv_plain(a) ? "true" : "false");
    else if (cconv_cls_eligible(a, dz, dy, dx))
        snprintf(name, name_bytes, "cconv_cls_kernel<%d, %s, %s, %s, %s>", ntt, cin <= 8 ? "true" : "false", sym ? "true" : "false",
                 cin <= 16 ? "true" : "false", !sym && cconv_plain(a) ? "true" : "false");
    else if (cconv_blk_eligible(a, dz, dy, dx))
        snprintf(name, name_bytes, "cconv_blk_kernel<%d>", ntt);
    else if (cconv_mfma_eligible(dz * dy * dx, cin, cout))
        snprintf(name, name_bytes, "cconv_mfma_kernel");
    else
        snprintf(name, name_bytes, "cconv_kernel<%d>", make_cfg(dx, dy, dz, cin, cout).CC);
    return DMCF_OK;
}

}  // extern "C"
