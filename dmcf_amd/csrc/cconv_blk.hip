// CConv / ASCC for 4x4x4 filters with ONE NEIGHBOUR PAIR PER MATRIX INSTRUCTION (v_mfma_f32_4x4x1_16B_f32).
//
// The 16-block form of the fp32 MFMA computes 16 independent 4x4 outer products  D_b[i][j] += A_b[i] * B_b[j]
// in 8 clocks (measured: 8.9 clk/instr/SIMD, tools/ubench/mfma_blocks.hip).  With
//
//     block b = (filter row y, channel group cg),  i = filter column x,  j = channel inside the group
//     A_b[i] = hat(X - i) * hat(Y - y) * wz(plane)          (X, Y: the pair's filter coordinates)
//     B_b[j] = f[4 cg + j]
//
// one instruction adds ONE pair's contribution to one 16-cell z plane of B_i[cell, 16 channels]; the pair
// touches two planes (bz, bz+1), so two instructions per pair = 16 clocks per SIMD -- the 16x16x4 form of
// cconv_mfma.hip needs 32 and feeds them with ~6 VALU operations per instruction.  Here K = 1: everything about
// "which pair" is wave uniform.  The pairs of a 62-pair batch are ordered by plane pair (bz = 0, 1, 2) so that each
// of three inner loops has fixed accumulators; the operands come from a per-wave LDS staging area (ds_read_b64: both
// planes' A value, ds_read_b32: the feature) and the inner loop has no VMEM instruction and two VALU adds per 4
// pairs.  The staging area is filled once per 31 pairs: the lanes that own the pairs (phase 1: gather, window,
// ball->cube map as in the other kernels) push {x, y, w0, w1} to their ordered slot, two lanes per slot form the 32
// products hat*hat*wz, and the feature rows arrive by 16-byte loads issued half a batch ahead.
//
// LDS (80 KB per workgroup, two workgroups per CU): B tile [16 points][64 cells x 16 channels] without padding
// (XOR swizzle instead) + 2 KB of feature staging per wave.  While a wave splats it uses the B row of its SECOND
// point as A staging (31 pairs x 32 floats, laid out [8 chunks][31 pairs][4 floats] so that the 16-byte stores
// of consecutive lanes and the 8-byte loads of the (y, x) lanes are both conflict free) -- the first point's
// accumulators go to its own row when it is done, the second point's replace the staging.
//
// Accumulation order = the (fixed) ordered-batch order, one fmaf per pair and cell (the MFMA is an exact fp32 fma):
// deterministic.
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {

#ifndef BLK_WAVES
#define BLK_WAVES 8
#endif
constexpr int kBWaves = BLK_WAVES;
#ifndef BLK_OCC
#define BLK_OCC 4  // waves per SIMD the register budget is set for (3 waves per SIMD: 8.5 ms against 7.3 ms)
#endif
constexpr int kBThreads = 64 * kBWaves;
constexpr int BTM = 2 * kBWaves;   // output points per workgroup = rows of the B tile
constexpr int kBRows = 16;         // M of the contraction MFMA (rows >= BTM alias rows < BTM; their results are dropped)
constexpr int BCH = 16;        // channels per pass
constexpr int kRow = 1024;     // floats per B row: 64 cells x 16 channels
constexpr int kHalf = 31;      // pairs per half batch (A staging: 8 chunks x 31 pairs x 4 floats <= one B row)
constexpr int kBatch = 2 * kHalf;
constexpr int kChunk = 4 * kHalf;  // floats per A-staging chunk row
constexpr int kFst = 512;         // floats of feature staging per wave (31 pairs x 16 channels)
constexpr int kBMaxNT = 4;

__device__ __forceinline__ float hat01(float d) { return __builtin_amdgcn_fmed3f(1.0f - fabsf(d), 0.0f, 1.0f); }

__device__ __forceinline__ uint32_t lds_addr(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q;
}

// four pair slots: A at pa + 16 u bytes, feature at pf + 64 u bytes; bumps both pointers
#define BLK_LOAD4(a0, a1, a2, a3, f0, f1, f2, f3, pa, pf)                                                          \
    asm volatile(                                                                                                  \
        "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:16\n\tds_read_b64 %2, %8 offset:32\n\tds_read_b64 %3, %8 " \
        "offset:48\n\tds_read_b32 %4, %9\n\tds_read_b32 %5, %9 offset:64\n\tds_read_b32 %6, %9 offset:128\n\t"      \
        "ds_read_b32 %7, %9 offset:192"                                                                            \
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3)                   \
        : "v"(pa), "v"(pf));                                                                                     \
    pa += 64;                                                                                                      \
    pf += 256
#define BLK_WAIT(n, a0, a1, a2, a3, f0, f1, f2, f3)                                                                \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                       \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3))
#define BLK_MFMA4(a0, a1, a2, a3, f0, f1, f2, f3)                                                                  \
    lo = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, f0, lo, 0, 0, 0);                                                \
    hi = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, f0, hi, 0, 0, 0);                                                \
    lo = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, f1, lo, 0, 0, 0);                                                \
    hi = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, f1, hi, 0, 0, 0);                                                \
    lo = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.x, f2, lo, 0, 0, 0);                                                \
    hi = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.y, f2, hi, 0, 0, 0);                                                \
    lo = __builtin_amdgcn_mfma_f32_4x4x1f32(a3.x, f3, lo, 0, 0, 0);                                                \
    hi = __builtin_amdgcn_mfma_f32_4x4x1f32(a3.y, f3, hi, 0, 0, 0)

struct Compact {  // per pair, in the registers of the owner lane
    float x, y;    // clamped filter coordinates in [0, 3]
    float w0, w1;  // a * (1 - fz), a * fz for planes bz, bz + 1
};

template <int NTT>
__global__ __launch_bounds__(kBThreads, BLK_OCC) void cconv_blk_kernel(const CconvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = p.cin, cout = p.cout;
    float* Bt = smem;                                  // [BTM][kRow], column index XOR (row << 2)
    float* Fst = smem + BTM * kRow + wave * kFst;      // [31][16]
    float* Ast = Bt + (wave + kBWaves) * kRow;         // the row of this wave's second point
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * BTM;
    const bool symmetric = (p.flags & DMCF_FLAG_SYMMETRIC) != 0;

    // MFMA 4x4x1 roles: block = lane >> 2 = (y, cg), element = lane & 3
    const int by = lane >> 4, cg = (lane >> 2) & 3, q = lane & 3;
    const float* a_rd = Ast + (by * 2 + (q >> 1)) * kChunk + (q & 1) * 2;  // + 4 * slot : {A plane0, A plane1}
    const float* f_rd = Fst + cg * 4 + q;                                  // + 16 * slot
    const uint32_t a_rd_lds = lds_addr(a_rd), f_rd_lds = lds_addr(f_rd);
    // publishing roles: lane -> (slot, pair of filter rows)
    const int ps = lane & 31, yh = lane >> 5;
    // feature load roles: lane -> (pair slot in a group of 16, 4 channels)
    const int fr = lane >> 2, fc4 = lane & 3;
    // contraction roles
    const int mi = lane & 15, mg = lane >> 4;

    f32x4 acc[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int c0 = chunk * BCH;
        const bool fch_ok = c0 + 4 * fc4 < cin;
        const int fch = fch_ok ? c0 + 4 * fc4 : 0;
        for (int pp = 0; pp < BTM / kBWaves; ++pp) {
            const int pt = wave + kBWaves * pp;
            const int64_t i = pt0 + pt;
            f32x4 b0 = {0.0f, 0.0f, 0.0f, 0.0f}, b1 = b0, b2 = b0, b3 = b0;
            if (i < p.n_out) {
                const int64_t rb = p.rs[i];
                int64_t re = p.cnt ? rb + p.cnt[i] : p.rs[i + 1];
                if (re > p.pair_cap) re = rb;
                const int64_t ntot = re - rb;
                const float ox = p.out_pos[3 * i], oy = p.out_pos[3 * i + 1], oz = p.out_pos[3 * i + 2];
                f32x4 fi4 = {0.0f, 0.0f, 0.0f, 0.0f};
                if (symmetric && fch_ok) fi4 = *(const f32x4*)(p.inp_feat + i * cin + fch);
                const int nb = (int)((ntot + kBatch - 1) / kBatch);

                auto ld_idx = [&](int b, int& j, float& nv, bool& v) {
                    const int64_t o = (int64_t)kBatch * b + lane;
                    v = lane < kBatch && o < ntot;
                    j = 0;
                    nv = 0.0f;
                    if (v) {
                        j = p.idx[rb + o];
                        if (p.nval) nv = p.nval[rb + o];
                    }
                };
                auto ld_pos = [&](int j, bool v, float& x, float& y, float& z) {
                    x = y = z = 0.0f;
                    if (v) {
                        x = p.inp_pos[3 * (int64_t)j];
                        y = p.inp_pos[3 * (int64_t)j + 1];
                        z = p.inp_pos[3 * (int64_t)j + 2];
                    }
                };
                auto geom = [&](int j, float nv, bool v, float x, float y, float z, int& bz) -> Compact {
                    Compact c = {0.0f, 0.0f, 0.0f, 0.0f};
                    bz = 3;
                    if (v) {
                        x -= ox;
                        y -= oy;
                        z -= oz;
                        float a = window_value(p.window, p.nval ? nv : rel_dist2(x, y, z), p.inv_r2, p.window_fac);
                        if (p.inp_imp) a *= p.inp_imp[j];
                        filter_coords<false>(x, y, z, p);
                        c.x = fminf(3.0f, fmaxf(0.0f, x));
                        c.y = fminf(3.0f, fmaxf(0.0f, y));
                        z = fminf(3.0f, fmaxf(0.0f, z));
                        const float zf = fminf(floorf(z), 2.0f);
                        const float fz = z - zf;
                        bz = (int)zf;
                        c.w0 = a * (1.0f - fz);
                        c.w1 = a * fz;
                    }
                    return c;
                };
                // The pairs of a batch are ordered by plane pair (bz = 0, 1, 2) so that each inner loop has fixed
                // accumulators; a half = 31 consecutive slots of that order.
                struct Order {
                    int pos, c0, c1;  // slot of this lane's pair in the ordered batch (63: none); pairs with bz == 0, bz == 1
                };
                auto order = [&](int bz) -> Order {  // bz == 3: lane without a pair
                    Order o;
                    const uint64_t m0 = __ballot(bz == 0), m1 = __ballot(bz == 1), m2 = __ballot(bz == 2);
                    o.c0 = __builtin_popcountll(m0);
                    o.c1 = __builtin_popcountll(m1);
                    const uint64_t mine = bz == 0 ? m0 : (bz == 1 ? m1 : m2);
                    const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0));
                    const int pos = rank + (bz == 0 ? 0 : (bz == 1 ? o.c0 : o.c0 + o.c1));
                    o.pos = bz < 3 ? pos : 63;
                    return o;
                };
                // slots of group g inside half h of a batch of np pairs
                auto seg = [&](const Order& o, int np, int h, int (&n)[3]) {
                    const int lo = kHalf * h, hi = min(np, kHalf * (h + 1));
                    const int e0 = o.c0, e1 = o.c0 + o.c1;
                    n[0] = max(0, min(e0, hi) - lo);
                    n[1] = max(0, min(e1, hi) - max(e0, lo));
                    n[2] = max(0, hi - max(e1, lo));
                };
                // Exchange area = the feature staging (dead between two splats): the lanes that own the pairs of a half
                // PUSH {x, y, w0, w1} (for the A products) or the neighbour index (for the feature loads) to the slot
                // their pair has in the ordered batch; the publishing / loading lanes read their slot.  One LDS round
                // trip, where pulling through ds_bpermute chains (slot -> owner lane -> value) took three.
                float* xcomp = Fst;               // [31][4]
                int* xidx = (int*)(Fst + 128);    // [31]
                auto push_compact = [&](const Compact& c, int pos, int h) {
                    const int s = pos - kHalf * h;
                    if (s >= 0 && s < kHalf) *(f32x4*)(xcomp + 4 * s) = (f32x4){c.x, c.y, c.w0, c.w1};
                };
                auto push_index = [&](int j, int pos, int h) {
                    const int s = pos - kHalf * h;
                    if (s >= 0 && s < kHalf) xidx[s] = j;
                };
                auto xfence = [&]() {  // cross-lane communication through LDS: keep the compiler from reordering around it
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                };
                // 16-byte feature loads of a half with n pairs: two groups of 16 slots, lane = (slot, 4 channels).
                // Slots beyond n read row 0 and are never consumed.
                auto f_issue = [&](int n, f32x4 (&f)[2]) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int s = 16 * k + fr;
                        const int jj = s < n ? xidx[s] : 0;
                        f[k] = *(const f32x4*)(p.inp_feat + (int64_t)jj * cin + fch);
                    }
                };
                auto f_publish = [&](const f32x4 (&f)[2]) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int s = 16 * k + fr;
                        f32x4 v = f[k] + fi4;
                        if (!fch_ok) v = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                        if (s < kHalf) *(f32x4*)(Fst + s * 16 + 4 * fc4) = v;
                    }
                };
                // A staging of half h: lane (slot ps, rows 2 yh, 2 yh + 1) writes its 16 products (both planes)
                auto a_publish = [&]() {
                    const f32x4 cv = *(const f32x4*)(xcomp + 4 * (ps < kHalf ? ps : 0));
                    const float x = cv.x, y = cv.y, w0 = cv.z, w1 = cv.w;
                    float hx[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) hx[k] = hat01(x - (float)k);
                    if (ps < kHalf) {
#pragma unroll
                        for (int yl = 0; yl < 2; ++yl) {
                            const float hy = hat01(y - (float)(2 * yh + yl));
                            const float u0 = hy * w0, u1 = hy * w1;
#pragma unroll
                            for (int xh = 0; xh < 2; ++xh) {
                                const f32x4 v = {hx[2 * xh] * u0, hx[2 * xh] * u1, hx[2 * xh + 1] * u0, hx[2 * xh + 1] * u1};
                                *(f32x4*)(Ast + ((2 * yh + yl) * 2 + xh) * kChunk + 4 * ps) = v;
                            }
                        }
                    }
                };
                // inner loops: `cnt` consecutive slots into the plane pair (lo, hi).  Hand-issued LDS reads (ds_read_b64
                // for {A plane lo, A plane hi}: the compiler would merge two of them into ds_read2_b64, which runs at half
                // the rate), four pairs per group.  Keeping a second group in flight (double-buffered operands) was
                // measured and dropped: its 12 extra live registers pushed ~40 VGPRs of loop-invariant state into
                // scratch around every call (3-6 GB of spill traffic per launch) -- 8.2 ms against 7.3 ms without.
                auto run = [&](int cnt, uint32_t& pa, uint32_t& pf, f32x4& lo, f32x4& hi) {
                    const int ng = cnt >> 2;
                    for (int g = 0; g < ng; ++g) {
                        f32x2 xa0, xa1, xa2, xa3;
                        float xf0, xf1, xf2, xf3;
                        BLK_LOAD4(xa0, xa1, xa2, xa3, xf0, xf1, xf2, xf3, pa, pf);
                        BLK_WAIT(0, xa0, xa1, xa2, xa3, xf0, xf1, xf2, xf3);
                        BLK_MFMA4(xa0, xa1, xa2, xa3, xf0, xf1, xf2, xf3);
                    }
                    for (int t = 4 * ng; t < cnt; ++t) {
                        f32x2 av;
                        float fv;
                        asm volatile("ds_read_b64 %0, %2\n\tds_read_b32 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(av), "=&v"(fv)
                                     : "v"(pa), "v"(pf));
                        lo = __builtin_amdgcn_mfma_f32_4x4x1f32(av.x, fv, lo, 0, 0, 0);
                        hi = __builtin_amdgcn_mfma_f32_4x4x1f32(av.y, fv, hi, 0, 0, 0);
                        pa += 16;
                        pf += 64;
                    }
                };
                auto splat = [&](const int (&n)[3]) {
                    // the staging stores above are plain C++, the loads below hand-written asm: one compiler barrier
                    // between them (LDS itself executes a wave's operations in order)
                    asm volatile("" ::: "memory");
                    uint32_t pa = a_rd_lds, pf = f_rd_lds;
                    run(n[0], pa, pf, b0, b1);
                    run(n[1], pa, pf, b1, b2);
                    run(n[2], pa, pf, b2, b3);
                };

                int j0, j1, bzc;
                float nv0, nv1, px, py, pz;
                bool v0, v1;
                if (nb > 0) {
                    ld_idx(0, j0, nv0, v0);
                    ld_idx(1, j1, nv1, v1);
                    ld_pos(j0, v0, px, py, pz);
                    Compact cur = geom(j0, nv0, v0, px, py, pz, bzc);
                    int curj = j0;
                    Order oc = order(bzc);
                    f32x4 fA[2], fB[2];
                    push_index(curj, oc.pos, 0);
                    xfence();
                    f_issue((int)min((int64_t)kHalf, ntot), fA);
                    ld_pos(j1, v1, px, py, pz);
                    for (int b = 0; b < nb; ++b) {
                        const int np = (int)min((int64_t)kBatch, ntot - (int64_t)kBatch * b);
                        int j2;
                        float nv2;
                        bool v2;
                        int nseg[3];
                        ld_idx(b + 2, j2, nv2, v2);
                        // ---- half 0: A products of (this batch, half 0), feature loads of (this batch, half 1)
                        push_compact(cur, oc.pos, 0);
                        push_index(curj, oc.pos, 1);
                        xfence();
                        a_publish();
                        f_issue(max(np - kHalf, 0), fB);
                        xfence();
                        f_publish(fA);
                        seg(oc, np, 0, nseg);
                        splat(nseg);
                        // geometry of the next batch (its position gathers were issued one batch ago)
                        int bzn;
                        const Compact nxt = geom(j1, nv1, v1, px, py, pz, bzn);
                        const int nxtj = j1;
                        const Order on = order(bzn);
                        ld_pos(j2, v2, px, py, pz);
                        // ---- half 1: A products of (this batch, half 1), feature loads of (next batch, half 0)
                        if (np > kHalf) {
                            push_compact(cur, oc.pos, 1);
                            push_index(nxtj, on.pos, 0);
                            xfence();
                            a_publish();
                            f_issue((int)min((int64_t)kHalf, max((int64_t)0, ntot - (int64_t)kBatch * (b + 1))), fA);
                            xfence();
                            f_publish(fB);
                            seg(oc, np, 1, nseg);
                            splat(nseg);
                        }
                        cur = nxt;
                        curj = nxtj;
                        oc = on;
                        j1 = j2;
                        nv1 = nv2;
                        v1 = v2;
                    }
                }
            }
            // D layout of 4x4x1: lane (block = (y, cg), j), reg r -> cell (x = r, y, plane), channel 4 cg + j
            {
                float* Brow = Bt + pt * kRow;
                const int sw = (pt & 15) << 2;
                const int ch = cg * 4 + q;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Brow[(((0 * 16 + by * 4 + r) * 16) + ch) ^ sw] = b0[r];
                    Brow[(((1 * 16 + by * 4 + r) * 16) + ch) ^ sw] = b1[r];
                    Brow[(((2 * 16 + by * 4 + r) * 16) + ch) ^ sw] = b2[r];
                    Brow[(((3 * 16 + by * 4 + r) * 16) + ch) ^ sw] = b3[r];
                }
            }
        }
        __syncthreads();
        // ---------------- contraction of this channel chunk on the matrix cores (as cconv_mfma.hip) ----------------
        const float* Wc = p.Wp + (size_t)chunk * p.nblocks * (4 * p.NT * 16 * 4);
        for (int blk = wave; blk < p.nblocks; blk += kBWaves) {
            const f32x4 av = *(const f32x4*)(Bt + (size_t)(mi % BTM) * kRow + ((blk * 16 + mg * 4) ^ ((mi % BTM) << 2)));
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
    float* red = Bt;  // [kBWaves][kBRows][16*NT]
    const int ncol = 16 * p.NT;
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        if (n < p.NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * kBRows + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < BTM * cout; e += kBThreads) {
        const int ptt = e / cout, o = e % cout;
        const int64_t ii = pt0 + ptt;
        if (ii >= p.n_out) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kBWaves; ++w) v += red[((size_t)w * kBRows + ptt) * ncol + o];
        if (p.bias) v += p.bias[o];
        float* dst = p.out + ii * cout + o;
        if (p.flags & DMCF_FLAG_ACCUMULATE) v += *dst;
        *dst = v;
    }
}

static constexpr size_t kBlkLds = (size_t)(BTM * kRow + kBWaves * kFst) * sizeof(float);

size_t cconv_blk_packed_floats(int cin, int cout) {
    const int nchunks = (cin + BCH - 1) / BCH, NT = (cout + 15) / 16;
    return (size_t)nchunks * (kRow / 16) * 4 * NT * 16 * 4;
}

// 4x4x4 filter, the flag set every DMCF model uses, 16-byte addressable feature rows.  Measured on MI355X (16
// channels): 7.3 ms against 9.3 ms for the 16x16x4 splat at 3.07e8 pairs / 265 per output, 4.0 against 4.35 ms
// at 3.3e7 pairs / 29 per output (32 -> 32): picked for every layer with at least 12 input channels.
bool cconv_blk_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "lds" / "mfma" / "blk": force one implementation (A/B tests)
    if (e && e[0] != 'b') return false;
    if (dx != 4 || dy != 4 || dz != 4) return false;
    if (a->coordinate_mapping != DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING || a->interpolation != DMCF_INTERP_LINEAR ||
        !(a->flags & DMCF_FLAG_ALIGN_CORNERS) || (a->flags & DMCF_FLAG_NORMALIZE))
        return false;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin & 3) || cout > 16 * kBMaxNT) return false;
    if ((uintptr_t)a->inp_features & 15) return false;
    if (e) return true;
    return cin >= 12;
}

int cconv_blk_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream) {
    const int nchunks = (p.cin + BCH - 1) / BCH, NT = (p.cout + 15) / 16, nblocks = kRow / 16;
    float* packed = (float*)workspace;
    {
        const int64_t total = (int64_t)cconv_blk_packed_floats(p.cin, p.cout);
        const unsigned g = (unsigned)((total + 255) / 256);
        if (!(a->flags & DMCF_FLAG_FILTER_PACKED))  // (else the workspace still holds it: dmcf_hip.h)
            hipLaunchKernelGGL(pack_filter, dim3(g < 2048u ? g : 2048u), dim3(256), 0, stream, a->filters, packed, 4, 4, 4, p.cin,
                           p.cout, BCH, 16 * BCH, nchunks, nblocks, NT, (a->flags & DMCF_FLAG_SYMMETRIC) ? 1 : 0,
                           a->sym_axis);
    }
    p.Wp = packed;
    p.KT = 4;
    p.KCp = kRow;
    p.nblocks = nblocks;
    p.NT = NT;
    p.nchunks = nchunks;
    p.bfloats = BTM * kRow;
    const int64_t ntiles = (p.n_out + BTM - 1) / BTM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    const void* fn = NT <= 1 ? (const void*)cconv_blk_kernel<1>
                             : (NT <= 2 ? (const void*)cconv_blk_kernel<2> : (const void*)cconv_blk_kernel<4>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBlkLds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(kBThreads), kargs, kBlkLds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

}  // namespace dmcf
