// CConv / ASCC with the SPLAT ON THE MATRIX CORES (filters of up to 64 cells: 4x4x4, 8x8, 8, 4x4 ...).
//
// cconv.hip accumulates B_i[cell, c] += w * f with read-modify-writes in LDS; measured on MI355X that loop is
// co-limited by instruction issue and the LDS pipe at ~13 clocks per neighbour pair per 8 channels, while the
// fp32 matrix pipe idles.  Here the same sum is written as a small dense GEMM per output point,
//
//     B_i[cell, c] = sum_j  S_i[cell, j] * F[j, c],      S_i[cell, j] = a_ij * hat(x_ij - cx) hat(y_ij - cy) hat(z_ij - cz)
//
// where (x_ij, y_ij, z_ij) are the pair's (clamped) filter coordinates and hat(d) = max(0, 1 - |d|): the trilinear
// weights of the 8 corner cells are exactly the non-zero values of that product, every other cell gets 0.  It is
// evaluated with v_mfma_f32_16x16x4_f32: M = 16 filter cells, N = 16 channels, K = 4 neighbour pairs.  That is 8x
// more multiply-adds than the 8 corners need, but they run at the matrix rate with NO LDS traffic and no data
// dependent addressing: per 4 pairs a wave issues five ds_bpermute (the pairs' index and coordinates, which live in
// the registers of the lane that owns the pair), one coalesced feature load (4 rows x 64 B), about two dozen VALU
// operations for the hat weights and K/16 MFMAs.  B accumulates in
// registers (K/16 x 4 VGPRs) in exact fp32 (the MFMA is a k-ordered fmaf chain) and is written once per point to
// the LDS tile the contraction reads -- the contraction itself is the one of cconv.hip.
//
// Filters of several 16-cell tiles that are not 4 x 4 planes (the 8 x 8 filters of the 2-D models: four tiles of two rows each)
// would spend three quarters of their matrix instructions on tiles a pair does not touch -- its footprint is 2 x 2 cells.  ORDER:
// every batch of 64 pairs is sorted by the tiles its footprint touches (lane = pair: a mask of <= 4 bits from the 8 corner cells,
// a key "first tile, and whether a second one", 9 ballots for the ranks, 6 ds_permute for the records), so the 4 pairs of an
// instruction mostly share their tiles, and an instruction is issued only for the tiles one of its 4 pairs touches (a wave-uniform
// test on 4 ballots per batch): 1.5 - 2 of 4 on the 8 x 8 filters.  With the batch's records staged in LDS in slot order and every
// load of the batch loop unconditional (see below) the 2-D layers of the WBC-SPH scene went from 226 to 148 us per launch; what is
// left is instruction issue: ~1,100 instructions per batch of 64 pairs around the matrix instructions (two waves per SIMD).  The sum of a cell is then formed in the sorted order:
// deterministic, a different rounding than list order.
//
// One workgroup = 8 waves = a tile of 16 output points (two points per wave, one after the other), 16 channels
// per pass, two workgroups per CU.  Interpolation modes map onto the same product: 'linear' stores clamped
// coordinates, 'linear_border' unclamped ones (the hat vanishes outside the array by itself), 'nearest_neighbor'
// stores rounded coordinates (hat of an integer offset is the indicator).
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {

constexpr int kMThreads = 512;
constexpr int kMWaves = kMThreads / 64;
constexpr int MTM = 16;    // output points per workgroup (MFMA M of the contraction)
constexpr int MCH = 16;    // channels per pass (MFMA N of the splat)
constexpr int kMaxKT = 4;  // 16-cell tiles: filters of up to 64 cells
constexpr int kMMaxNT = 4;
constexpr int kMStage = 2 * 64 * 16 + 2 * 64 * 4 + 64 * 4;  // bytes per wave: two batches of records {x, y, z, a} and indices, one of tile masks

__device__ __forceinline__ float hat(float d) { return fmaxf(0.0f, 1.0f - fabsf(d)); }

// Per-pair record kept in the registers of the lane that owns the pair.  PLANE16 (4x4xN filters): the four
// plane weights a * hat(z - plane) are evaluated here, once per pair at full lane utilisation, because the fp32
// MFMA and the VALU do not overlap on this chip (measured: kernel time ~ MFMA time + VALU time) and every VALU
// instruction saved in the (cell, pair) layout -- where each value is computed 16 times over -- counts.
struct PairRec {
    float x, y, z, a;  // generic: clamped filter coordinates + importance
    float w1, w2, w3;  // PLANE16: z = a*hat(z), a = unused, w1..w3 = a*hat(z-1..3)
    unsigned long long need[kMaxKT];  // ORDER (wave uniform): the slots of the sorted batch that touch tile t
};

// GENERIC: runtime mapping / interpolation switches; PLANE16: sx*sy == 16 (the (x,y) hat product is shared by all
// tiles, z = tile index); NTT: number of 16-channel output tiles the contraction accumulates (register budget);
// KTT: number of 16-cell tiles when known at compile time (0 = runtime p.KT).
template <bool GENERIC, bool PLANE16, int NTT, int KTT>
__global__ __launch_bounds__(kMThreads, 2) void cconv_mfma_kernel(const CconvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KCp = p.KCp, cin = p.cin, cout = p.cout, KT = p.KT;
    float* Bt = smem;              // [MTM][KCp]
    float* norm = Bt + p.bfloats;  // [MTM]
    char* const stg = (char*)(norm + MTM) + wave * kMStage;  // (!PLANE16) the batch's records in slot order, see `stage`
    f32x4* const Rs = (f32x4*)stg;                 // [2][64] {x, y, z, a}
    int* const Js = (int*)(stg + 2048);            // [2][64] neighbour index
    uint32_t* const Ts = (uint32_t*)(stg + 2560);  // [64] tiles the slot's pair touches
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * MTM;
    const bool symmetric = (p.flags & DMCF_FLAG_SYMMETRIC) != 0;
    const int mi = lane & 15, mg = lane >> 4;  // MFMA roles: A row (cell) / B column (channel); k index (pair)
    constexpr bool ORDER = !GENERIC && !PLANE16 && (KTT == 0 || KTT > 1);  // see the header
    const bool order_on = ORDER && KT > 1;

    // filter cell of this lane in each 16-cell tile (cells beyond K are parked far away: weight 0)
    constexpr int NCT = PLANE16 ? 1 : kMaxKT;
    float cxs[NCT], cys[NCT], czs[NCT];
#pragma unroll
    for (int mt = 0; mt < NCT; ++mt) {
        const int cell = 16 * mt + mi;
        const int plane = p.sx * p.sy;
        const bool ok = cell < p.K;
        const int r = cell % plane;
        cxs[mt] = ok ? (float)(r % p.sx) : -100.0f;
        cys[mt] = ok ? (float)(r / p.sx) : -100.0f;
        czs[mt] = ok ? (float)(cell / plane) : -100.0f;
    }

    f32x4 acc[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // (small launches: one channel chunk per workgroup, blockIdx.y -- see cconv_mfma_launch)
    const int chunk_lo = p.csplit ? (int)blockIdx.y : 0, chunk_hi = p.csplit ? (int)blockIdx.y + 1 : p.nchunks;
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
        const int c0 = chunk * MCH;
        const bool ch_ok = c0 + mi < cin;
        // ---------------- splat on the matrix cores: two points per wave ----------------
        for (int pp = 0; pp < MTM / kMWaves; ++pp) {
            const int pt = wave + kMWaves * pp;
            const int64_t i = pt0 + pt;
            f32x4 bacc[kMaxKT];
#pragma unroll
            for (int mt = 0; mt < kMaxKT; ++mt) bacc[mt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            float nsum = 0.0f;
            if (i < p.n_out) {
                const int64_t rb = p.rs[i];
                int64_t re = p.cnt ? rb + p.cnt[i] : p.rs[i + 1];
                if (re > p.pair_cap) re = rb;
                const float ox = p.out_pos[3 * i], oy = p.out_pos[3 * i + 1], oz = p.out_pos[3 * i + 2];
                const float fi = (symmetric && ch_ok) ? p.inp_feat[i * cin + c0 + mi] : 0.0f;
                const int nb = (int)((re - rb + 63) >> 6);
                // Everything per pair lives in registers of the lane that owns the pair (phase-1 layout: lane =
                // pair); the (pair, channel) / (cell, pair) operand layouts of the MFMA fetch it with ds_bpermute.
                // Software pipeline per batch of 64 pairs: (index, d^2) loads run two batches ahead, position
                // gathers one batch ahead, the feature loads of one half batch are in flight while the MFMAs of
                // the other half batch issue.
                // Every load of the batch loop is UNCONDITIONAL (clamped indices, stand-in pointers, values masked afterwards): a load
                // inside a branch -- even a uniform one -- makes the compiler's s_waitcnt bookkeeping give up at the join and wait
                // with vmcnt(0), i.e. for the loads just issued as well, and the software pipeline below degenerates into one memory
                // round trip per quarter batch (measured on the 2-D scenes: 8,000 clocks per batch of 64 pairs, whatever the work).
                const float* const nvp = p.nval ? p.nval : (const float*)p.idx;
                const float* const impp = p.inp_imp ? p.inp_imp : p.inp_pos;
                auto ld_idx = [&](int b, int& j, float& nv, bool& v) {
                    const int64_t q = rb + 64 * (int64_t)b + lane;
                    v = q < re;
                    const int64_t qc = min(q, re - 1);  // (re > rb: the pipeline runs for non-empty rows only)
                    const int jj = p.idx[qc];
                    const float nn = nvp[qc];
                    j = v ? jj : 0;
                    nv = v ? nn : 0.0f;
                };
                auto ld_pos = [&](int j, bool v, float& x, float& y, float& z) {
                    x = p.inp_pos[3 * (int64_t)j];
                    y = p.inp_pos[3 * (int64_t)j + 1];
                    z = p.inp_pos[3 * (int64_t)j + 2];
                };
                auto geom = [&](int j, float nv, bool v, float x, float y, float z) -> PairRec {
                    float a = 0.0f;
                    const float iv = impp[j];
                    {   // (lanes without a pair carry the geometry of point 0: their weight is zeroed below)
                        x -= ox;
                        y -= oy;
                        z -= oz;
                        a = window_value(p.window, p.nval ? nv : rel_dist2(x, y, z), p.inv_r2, p.window_fac);
                        a = v ? a : 0.0f;
                        nsum += a;
                        if (p.inp_imp) a *= iv;
                        filter_coords<GENERIC>(x, y, z, p);
                        const float hx = (float)(p.sx - 1), hy = (float)(p.sy - 1), hz = (float)(p.sz - 1);
                        if (!GENERIC || p.interp == DMCF_INTERP_LINEAR) {  // coordinate clamping
                            x = fminf(hx, fmaxf(0.0f, x));
                            y = fminf(hy, fmaxf(0.0f, y));
                            z = fminf(hz, fmaxf(0.0f, z));
                        } else if (p.interp == DMCF_INTERP_NEAREST) {
                            x = fminf(hx, fmaxf(0.0f, roundf(x)));
                            y = fminf(hy, fmaxf(0.0f, roundf(y)));
                            z = fminf(hz, fmaxf(0.0f, roundf(z)));
                        } else {  // LINEAR_BORDER: no clamping; keep the values finite and small
                            if (!(x == x) || !(y == y) || !(z == z)) a = 0.0f;
                            x = fminf(hx + 2.0f, fmaxf(-2.0f, x));
                            y = fminf(hy + 2.0f, fmaxf(-2.0f, y));
                            z = fminf(hz + 2.0f, fmaxf(-2.0f, z));
                        }
                    }
                    PairRec r;
#pragma unroll
                    for (int mt = 0; mt < kMaxKT; ++mt) r.need[mt] = ~0ull;
                    r.x = x;
                    r.y = y;
                    if constexpr (PLANE16) {
                        r.z = hat(z) * a;
                        r.a = 0.0f;
                        r.w1 = hat(z - 1.0f) * a;
                        r.w2 = hat(z - 2.0f) * a;
                        r.w3 = hat(z - 3.0f) * a;
                    } else {
                        r.z = z;
                        r.a = a;  // a == 0 for lanes without a pair
                        r.w1 = r.w2 = r.w3 = 0.0f;
                    }
                    return r;
                };
                // Feature loads of one quarter batch (4 groups of 4 pairs): lane = (pair slot mg, channel mi).
                // Branch free and never touched until the MFMAs consume them (a use would make the compiler wait
                // for the load right here): out-of-range slots read row 0 / channel 0 and are masked at use.
                const int ch_safe = ch_ok ? c0 + mi : 0;
                auto issue = [&](int bj, int buf, int np, int g0, float (&f)[4]) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int q = 4 * (g0 + g) + mg;
                        // (!PLANE16: the staged index of slot q -- one broadcast read per 16 lanes instead of a ds_bpermute)
                        const int jj = PLANE16 ? __shfl(bj, q, 64) : Js[buf * 64 + q];  // 0 for slots beyond np
                        f[g] = p.inp_feat[(int64_t)jj * cin + ch_safe];
                    }
                };
                auto run = [&](const PairRec& c, int buf, int np, int g0, const float (&f)[4]) {
                    if (4 * g0 >= np) return;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int q = 4 * (g0 + g) + mg;
                        const float fv = (q < np && ch_ok) ? f[g] + fi : 0.0f;
                        f32x4 rc = {0.0f, 0.0f, 0.0f, 0.0f};
                        if constexpr (!PLANE16) rc = Rs[buf * 64 + q];  // the record of slot q: one 16-byte broadcast read
                        const float x = PLANE16 ? __shfl(c.x, q, 64) : rc.x, y = PLANE16 ? __shfl(c.y, q, 64) : rc.y;
                        if constexpr (PLANE16) {
                            const float wxy = hat(x - cxs[0]) * hat(y - cys[0]);
                            const float w0 = __shfl(c.z, q, 64);  // a * hat(z - plane): 0 for slots beyond np
                            bacc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxy * w0, fv, bacc[0], 0, 0, 0);
                            if ((KTT ? KTT : KT) > 1)
                                bacc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxy * __shfl(c.w1, q, 64), fv, bacc[1], 0, 0, 0);
                            if ((KTT ? KTT : KT) > 2)
                                bacc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxy * __shfl(c.w2, q, 64), fv, bacc[2], 0, 0, 0);
                            if ((KTT ? KTT : KT) > 3)
                                bacc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wxy * __shfl(c.w3, q, 64), fv, bacc[3], 0, 0, 0);
                        } else {
                            const float z = rc.z, a = rc.w;  // a == 0 for slots beyond np
#pragma unroll
                            for (int mt = 0; mt < kMaxKT; ++mt)
                                if (mt < (KTT ? KTT : KT) && (!ORDER || ((c.need[mt] >> (4 * (g0 + g))) & 0xfull) != 0ull))
                                    bacc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                                        (hat(x - cxs[mt]) * hat(y - cys[mt])) * (hat(z - czs[mt]) * a), fv, bacc[mt], 0, 0, 0);
                        }
                    }
                };
                // !PLANE16: the batch's records go to LDS in SLOT order -- ORDER: sorted by the tiles the pairs touch, else list order --
                // where the splat reads {x, y, z, a} of slot q with one broadcast ds_read_b128 and the feature loads read its index
                // (the four ds_bpermute per group of 4 pairs this replaces -- 128 per batch, each waited for -- were the floor of the
                // 2-D layers: 129 of 157 us per launch with the matrix instructions taken out).  need[t] = ballot of "slot touches
                // tile t".  Two buffers: the next batch is staged while the current one still has quarters to go.
                auto stage = [&](PairRec& r, int& j, bool v, int buf) {
                    if constexpr (PLANE16) return;
                    uint32_t tm = 0xfu;
                    int rank = lane;
                    if (order_on) {
                        tm = 0;
                        if (v && r.a != 0.0f) {
                            const int x0 = (int)r.x, y0 = (int)r.y, z0 = (int)r.z;  // (clamped to [0, size - 1]: never negative)
                            const int x1 = min(x0 + 1, p.sx - 1), y1 = min(y0 + 1, p.sy - 1), z1 = min(z0 + 1, p.sz - 1);
#pragma unroll
                            for (int c = 0; c < 8; ++c) {
                                const int cell = (((c & 4) ? z1 : z0) * p.sy + ((c & 2) ? y1 : y0)) * p.sx + ((c & 1) ? x1 : x0);
                                tm |= 1u << (cell >> 4);
                            }
                        }
                        // key: 2 * (first tile) + (touches a later one too); pairs without a footprint, then lanes without a pair, last
                        const int first = tm ? __builtin_ctz(tm) : 0;
                        const int key = tm ? 2 * first + ((tm >> first) > 1u ? 1 : 0) : (v ? 7 : 8);  // (the last tile has no later one: 7 is free)
                        const unsigned long long below = (1ull << lane) - 1ull;
                        int base = 0;
#pragma unroll
                        for (int k = 0; k < 9; ++k) {
                            const unsigned long long bk = __ballot(key == k);
                            if (key == k) rank = base + (int)__popcll(bk & below);
                            base += (int)__popcll(bk);
                        }
                    }
                    Rs[buf * 64 + rank] = (f32x4){r.x, r.y, r.z, r.a};
                    Js[buf * 64 + rank] = j;
                    if (order_on) Ts[rank] = tm;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (order_on) {
                        tm = Ts[lane];
#pragma unroll
                        for (int mt = 0; mt < kMaxKT; ++mt) r.need[mt] = __ballot((tm >> mt) & 1u);
                    }
                };
                if (nb > 0) {
                int j0, j1;
                float nv0, nv1, px, py, pz;
                bool v0, v1;
                ld_idx(0, j0, nv0, v0);
                ld_idx(1, j1, nv1, v1);
                ld_pos(j0, v0, px, py, pz);
                PairRec cur = geom(j0, nv0, v0, px, py, pz);
                stage(cur, j0, v0, 0);
                int curj = j0;
                int np_cur = (int)min((int64_t)64, re - rb);
                // three quarter-batch feature buffers rotate: a load has two quarters of MFMAs to land
                float fA[4], fB[4], fC[4];
                issue(curj, 0, np_cur, 0, fA);
                issue(curj, 0, np_cur, 4, fB);
                ld_pos(j1, v1, px, py, pz);
                for (int b = 0; b < nb; ++b) {
                    int j2;
                    float nv2;
                    bool v2;
                    ld_idx(b + 2, j2, nv2, v2);
                    const int bc = b & 1, bn = bc ^ 1;
                    issue(curj, bc, np_cur, 8, fC);
                    run(cur, bc, np_cur, 0, fA);
                    issue(curj, bc, np_cur, 12, fA);
                    run(cur, bc, np_cur, 4, fB);
                    PairRec nxt = geom(j1, nv1, v1, px, py, pz);
                    stage(nxt, j1, v1, bn);
                    const int nxtj = j1;
                    const int np_nxt = (int)min((int64_t)64, max((int64_t)0, re - rb - 64 * (int64_t)(b + 1)));
                    issue(nxtj, bn, np_nxt, 0, fB);
                    ld_pos(j2, v2, px, py, pz);
                    run(cur, bc, np_cur, 8, fC);
                    issue(nxtj, bn, np_nxt, 4, fC);
                    run(cur, bc, np_cur, 12, fA);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        fA[g] = fB[g];
                        fB[g] = fC[g];
                    }
                    cur = nxt;
                    curj = nxtj;
                    np_cur = np_nxt;
                    j1 = j2;
                    nv1 = nv2;
                    v1 = v2;
                }
                }  // nb > 0
            }
            // D layout of 16x16x4: lane l, reg r -> row (cell in tile) 4*(l>>4)+r, column (channel) l&15
            float* Brow = Bt + (size_t)pt * KCp;
#pragma unroll
            for (int mt = 0; mt < kMaxKT; ++mt)
                if (mt < (KTT ? KTT : KT)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) Brow[(16 * mt + 4 * mg + r) * MCH + mi] = bacc[mt][r];
                }
            if (chunk == 0 && (p.flags & DMCF_FLAG_NORMALIZE)) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) nsum += __shfl_xor(nsum, d, 64);
                if (lane == 0) norm[pt] = nsum;
            }
        }
        __syncthreads();
        // ---------------- contraction of this channel chunk on the matrix cores ----------------
        const float* Wc = p.Wp + (size_t)chunk * p.nblocks * (4 * p.NT * 16 * 4);
        for (int blk = wave; blk < p.nblocks; blk += kMWaves) {
            const f32x4 av = *(const f32x4*)(Bt + (size_t)mi * KCp + blk * 16 + mg * 4);
            const float* wb = Wc + ((size_t)(blk * 4 + mg) * p.NT * 16 + mi) * 4;
#pragma unroll
            for (int n = 0; n < NTT; ++n) {
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
    float* red = Bt;  // [kMWaves][MTM][16*NT]
    const int ncol = 16 * p.NT;
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        if (n < p.NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * MTM + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < MTM * cout; e += kMThreads) {
        const int ptt = e / cout, o = e % cout;
        const int64_t ii = pt0 + ptt;
        if (ii >= p.n_out) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kMWaves; ++w) v += red[((size_t)w * MTM + ptt) * ncol + o];
        if (p.csplit) {  // this chunk's share: cconv_mfma_sum_chunks adds the chunks in order, the bias and the value to accumulate to
            p.partial[((size_t)chunk_lo * p.n_out + ii) * cout + o] = v;
            continue;
        }
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

// out = (+=) sum over the chunks of their partial sums, in chunk order, + bias
__global__ void cconv_mfma_sum_chunks(const float* __restrict__ partial, int nchunks, int64_t n_out, int cout, const float* __restrict__ bias,
                                      float* __restrict__ out, int accumulate) {
    const int64_t total = n_out * cout;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.0f;
        for (int c = 0; c < nchunks; ++c) v += partial[(size_t)c * total + e];
        if (bias) v += bias[e % cout];
        if (accumulate) v += out[e];
        out[e] = v;
    }
}

// A launch of a few thousand output points is a few hundred workgroups, each of which walks its rows once per 16-channel chunk --
// dependent round trips, one chunk after the other: 92 us per launch for the 2,401-point 2-D scenes whatever the work (rocprofv3,
// profiles/r05_small_configs.md), times the 27 - 43 layers of a step.  With DMCF_MFMA_SPLIT=1 and at most kSplitMaxOut outputs
// every chunk gets its own workgroup (grid.y) and the chunks' partial sums meet in a small second kernel: the same products, the
// chunks added in order.  OPT-IN, because measured it is a wash: those steps are paced by the host, and the second launch costs it
// what the shorter kernel saves the GPU (WBC-SPH architecture 5.37 -> 5.17 ms per step, WaterRamps 3.15 -> 3.45).
constexpr int64_t kSplitMaxOut = 16384;

struct MfmaCfg {
    int KT, KCp, nblocks, NT, nchunks;
    size_t lds, packed_floats, bfloats;
};

static MfmaCfg mfma_cfg(int K, int cin, int cout) {
    MfmaCfg c;
    c.KT = (K + 15) / 16;
    const int KC = c.KT * 16 * MCH;  // multiple of 256
    c.nblocks = KC / 16;
    c.KCp = KC + 4;  // == 4 (mod 64): conflict-free ds_read_b128 of 16 rows in the contraction
    c.NT = (cout + 15) / 16;
    c.nchunks = (cin + MCH - 1) / MCH;
    size_t b = (size_t)MTM * c.KCp;
    const size_t r = (size_t)kMWaves * MTM * 16 * c.NT;
    if (b < r) b = r;
    c.bfloats = b;
    c.lds = (b + MTM) * sizeof(float) + (size_t)kMWaves * kMStage;
    c.packed_floats = (size_t)c.nchunks * c.nblocks * 4 * c.NT * 16 * 4;
    return c;
}

bool cconv_mfma_eligible(int K, int cin, int cout) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "lds" / "mfma": force one implementation (A/B tests)
    if (e && e[0] != 'm') return false;
    if (K > 16 * kMaxKT || cout > 16 * kMMaxNT) return false;
    if (e && e[0] == 'm') return true;
    // Measured on MI355X at 3.07e8 pairs (profiles/): the LDS splat costs ~6.9 ms per 8-channel pass (5.7 ms for a
    // 4-channel one), the matrix-core splat ~9.4 ms per 16-channel pass whatever the channel count.
    const double lds = cin <= 4 ? 5.7 : 6.9 * ((cin + 7) / 8);
    const double mfma = 9.4 * ((cin + 15) / 16);
    return mfma < lds;
}

size_t cconv_mfma_packed_floats(int K, int cin, int cout) { return mfma_cfg(K, cin, cout).packed_floats; }

size_t cconv_mfma_partial_floats(int K, int cin, int cout, int64_t n_out) {
    const int nchunks = (cin + MCH - 1) / MCH;
    const char* e = getenv("DMCF_MFMA_SPLIT");
    if (nchunks < 2 || n_out > kSplitMaxOut || !e || e[0] != '1') return 0;
    return (size_t)nchunks * (size_t)n_out * (size_t)cout;
}

int cconv_mfma_launch(CconvParams p, const dmcf_cconv_args* a, int dz, int dy, int dx, void* workspace,
                      hipStream_t stream) {
    const MfmaCfg cfg = mfma_cfg(p.K, p.cin, p.cout);
    float* packed = (float*)workspace;
    {
        const int64_t total = (int64_t)cfg.packed_floats;
        const unsigned g = (unsigned)((total + 255) / 256);
        // same packer as the LDS path with 16 channels per chunk and an unpadded plane stride
        if (!(a->flags & DMCF_FLAG_FILTER_PACKED))  // (else the workspace still holds it: dmcf_hip.h)
            hipLaunchKernelGGL(pack_filter, dim3(g < 2048u ? g : 2048u), dim3(256), 0, stream, a->filters, packed, dz, dy, dx,
                           p.cin, p.cout, MCH, dy * dx * MCH, cfg.nchunks, cfg.nblocks, cfg.NT,
                           (a->flags & DMCF_FLAG_SYMMETRIC) ? 1 : 0, a->sym_axis);
    }
    p.Wp = packed;
    p.KT = cfg.KT;
    p.KCp = cfg.KCp;
    p.nblocks = cfg.nblocks;
    p.NT = cfg.NT;
    p.nchunks = cfg.nchunks;
    p.bfloats = (int)cfg.bfloats;
    const int64_t ntiles = (p.n_out + MTM - 1) / MTM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    const size_t partial = (a->flags & DMCF_FLAG_NORMALIZE) ? 0 : cconv_mfma_partial_floats(p.K, p.cin, p.cout, p.n_out);
    p.csplit = partial ? 1 : 0;
    p.partial = partial ? packed + align_up(cfg.packed_floats, 64) : nullptr;
    const bool generic = !(a->coordinate_mapping == DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING &&
                           a->interpolation == DMCF_INTERP_LINEAR && (a->flags & DMCF_FLAG_ALIGN_CORNERS));
    const bool plane16 = !generic && dx * dy == 16;
    const int ntt = cfg.NT <= 1 ? 1 : (cfg.NT <= 2 ? 2 : 4);
    const void* fn;
#define DMCF_PICK(G, P16, KTT)                                                                          \
    (ntt == 1 ? (const void*)cconv_mfma_kernel<G, P16, 1, KTT>                                         \
              : (ntt == 2 ? (const void*)cconv_mfma_kernel<G, P16, 2, KTT> : (const void*)cconv_mfma_kernel<G, P16, 4, KTT>))
    if (generic)
        fn = DMCF_PICK(true, false, 0);
    else if (plane16 && cfg.KT == 4)
        fn = DMCF_PICK(false, true, 4);
    else if (plane16)
        fn = DMCF_PICK(false, true, 0);
    else if (cfg.KT == 4)
        fn = DMCF_PICK(false, false, 4);
    else
        fn = DMCF_PICK(false, false, 0);
#undef DMCF_PICK
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid, p.csplit ? (unsigned)p.nchunks : 1u), dim3(kMThreads), kargs, cfg.lds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    if (p.csplit) {
        const int64_t total = p.n_out * p.cout;
        hipLaunchKernelGGL(cconv_mfma_sum_chunks, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p.partial, p.nchunks, p.n_out,
                           p.cout, p.bias, p.out, (p.flags & DMCF_FLAG_ACCUMULATE) ? 1 : 0);
    }
    return check_launch();
}

}  // namespace dmcf
