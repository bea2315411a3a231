// CConv for 4x4x4 filters and at most 16 input channels: ONE NEIGHBOUR PAIR PER MATRIX INSTRUCTION at four waves per SIMD --
// splat G, v_mfma_f32_4x4x1_16B_f32 into 9 class tiles.
//
// cconv_pair.hip (splat F) showed what a pair-per-instruction splat saves around the matrix instruction -- no ordering of the
// pairs by class, no padding slots, no operand arithmetic, a third of the LDS reads -- but its 27 tiles x 4 registers leave
// room for two waves per SIMD only.  With at most 16 channels a tile can span the four filter PLANES instead of two:
//
//     block b = (plane z in 0..3, channel quad in 0..3),   row i = (y', x'),   column j = channel inside the quad
//     A[b][i] = a w_y[y'] w_x[x'] * (w_z[z - bz] if z in {bz, bz + 1} else 0),        B[b][j] = f[4 quad + j]
//
// one instruction (8 clocks) adds one pair to 4 planes x 2 x 2 cells x 16 channels, half of the products are zeros, and the
// classes are the 9 base cells (by, bx): 36 tile registers -- the register budget of splat D (cconv_cls.hip), whose workgroup
// shape this kernel keeps: 8 waves x 2 output points, two workgroups per CU, the B tile [16 points][64 cells x 16 channels] in
// 64 KB of LDS, the shared contraction.  A pair's record is its 16 products [plane][y'][x'] (8 of them zero); records and
// features are staged pair-interleaved, half a batch (32 pairs) at a time in the B row of the wave's second point, and read
// four pairs per ds_read_b128; the class bytes reach M0 through scalar instructions (tools/gen_p16_splat.py).
//
//   registers: v0 .. v75 the compiler (amdgpu_num_vgpr), v76 .. v91 the operands of 8 pairs, v92 .. v127 the class tiles
//   LDS:       80 KB per workgroup: B tile 64 KB + 2 KB per wave (feature staging); records + indices in the second point's row
//
// Accumulation order = list order inside a class, then the fixed merge order: deterministic.
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {

constexpr int kGWaves = 8;
constexpr int kGThreads = 64 * kGWaves;
constexpr int GTM = 2 * kGWaves;   // output points per workgroup = rows of the B tile
constexpr int kGRow = 1024;        // floats per B row: k' = (z * 4 + y) * 64 + channel * 4 + x
constexpr int kGRecG = 68;         // floats per record group: 16 products x 4 pairs, padded
constexpr int kGFst = 8 * 64;      // per wave outside the B tile: [8 groups][16 channels (permuted)][4 pairs]
constexpr int kGMaxNT = 4;
constexpr int kGCompilerVgprs = 38;  // (the attribute counts HALF of the unified file: v0 .. v75)

#define P16_FIXED_REGS                                                                                                     \
    "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92",   \
        "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107",     \
        "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",      \
        "v122", "v123", "v124", "v125", "v126", "v127"

__device__ __forceinline__ void gfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t glds(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q;
}

struct P16Rec {    // per pair, in the registers of its owner lane
    f32x4 lo, hi;  // a w_z[z'] w_y[y'] w_x[x'], index 2 y' + x', for z' = 0 / 1
    int bz;        // base plane: the products belong to planes bz, bz + 1
    int cls4;      // 4 * (by * 3 + bx)
};

constexpr uint32_t kGOob = 0xffffffffu;  // a byte offset no buffer holds: the load returns zeros

// PLAIN: see cconv_plain() in cconv_common.h
template <int NTT, bool PLAIN>
__global__ __launch_bounds__(kGThreads, 4) __attribute__((amdgpu_num_vgpr(kGCompilerVgprs))) void cconv_p16_kernel(const CconvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = p.cin, cout = p.cout;
    const int window = PLAIN ? (int)DMCF_WINDOW_POLY6 : p.window;
    const float* const nval = PLAIN ? nullptr : p.nval;
    const float* const imp = PLAIN ? nullptr : p.inp_imp;
    float* Bt = smem;                                    // [GTM][kGRow], 4-float groups XOR-swizzled by the row
    float* Fst = smem + GTM * kGRow + wave * kGFst;      // [8 groups][16 channels (permuted)][4 pairs]: half a batch
    float* Rec = Bt + (wave + kGWaves) * kGRow;          // [8 groups][kGRecG] in the row of this wave's second point: product
                                                         // v = plane * 4 + (y', x') of pair 4 g + t at g * kGRecG + 4 v + t
    uint32_t* Jof = (uint32_t*)(Rec + 8 * kGRecG);       // [64]: byte offset of a pair's feature row (kGOob: no pair)
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * GTM;

    // splat roles: this lane's plane (A operand) and channel (B operand, accumulator column)
    const int zl = lane >> 4, ch = lane & 15;
    // feature load roles: lane -> (pair fr of a round of 16, channels 4 fq .. 4 fq + 3)
    const int fr = lane >> 2, fq = lane & 3;
    const uint32_t rowB = (uint32_t)cin * 4u;
    const uint32_t cbyte = 4 * fq < cin ? 16u * (uint32_t)fq : kGOob;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)p.inp_feat, 0, (int)((uint32_t)p.n_inp * rowB), 0x00020000);
    // contraction roles
    const int mi = lane & 15, mg = lane >> 4;

    // The batches of the wave's two points form ONE stream (point A's batches, then point B's), as in cconv_cls.hip
    int64_t rbs[2];
    int nts[2], nbs[2];
    float oxs[2], oys[2], ozs[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int64_t i = pt0 + wave + kGWaves * pp;
        rbs[pp] = 0;
        nts[pp] = 0;
        oxs[pp] = oys[pp] = ozs[pp] = 0.0f;
        if (i < p.n_out) {
            const int64_t rb = p.rs[i];
            int64_t re = p.cnt ? rb + p.cnt[i] : p.rs[i + 1];
            if (re > p.pair_cap) re = rb;
            rbs[pp] = rb;
            nts[pp] = (int)min(re - rb, (int64_t)0x7fffffc0);
            oxs[pp] = p.out_pos[3 * i];
            oys[pp] = p.out_pos[3 * i + 1];
            ozs[pp] = p.out_pos[3 * i + 2];
        }
        nts[pp] = __builtin_amdgcn_readfirstlane(nts[pp]);
        nbs[pp] = (nts[pp] + 63) >> 6;
        oxs[pp] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oxs[pp])));
        oys[pp] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oys[pp])));
        ozs[pp] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ozs[pp])));
    }
    const int nbA = nbs[0], NB = nbs[0] + nbs[1];
    const int nt0 = nts[0], nt1 = nts[1];
    const int64_t rb0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rbs[0] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rbs[0]);
    const int64_t rb1 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rbs[1] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rbs[1]);
    // ONE buffer over both rows of the wave (they are 8 rows apart in the list; eligibility bounds every row by 2^24 entries)
    const int64_t gapB = nt1 > 0 ? rb1 - rb0 : 0;
    const bool near = gapB >= 0 && gapB + nt1 < ((int64_t)1 << 29) && nt0 < (1 << 29);
    if (!near) __builtin_trap();
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void*)(p.idx + rb0), 0, (int)(max((int64_t)nt0, gapB + nt1) * 4), 0x00020000);
    const uint32_t offB = (uint32_t)gapB * 4u;

    auto zero_tiles = [&]() {
        asm volatile(
#include "cconv_p16_zero.inc"
            ::: "memory", P16_FIXED_REGS);
    };
    zero_tiles();

    auto npairs = [&](int t) -> int {  // pairs of batch t of the stream (wave uniform)
        const bool pp = t >= nbA;
        return min(64, (pp ? nt1 : nt0) - 64 * (t - (pp ? nbA : 0)));
    };
    auto valid = [&](int t) -> bool { return t < NB && lane < npairs(t); };
    auto ld_idx = [&](int t, int& j, float& nv) {
        const bool pp = t >= nbA, ok = valid(t);
        const int o = 64 * (t - (pp ? nbA : 0)) + lane;
        j = (int)__builtin_amdgcn_raw_buffer_load_b32(rI, ok ? (uint32_t)o * 4u + (pp ? offB : 0u) : kGOob, 0, 0);
        nv = 0.0f;
        if (nval && ok) nv = nval[(pp ? rb1 : rb0) + o];
    };
    auto ld_pos = [&](int j, float& x, float& y, float& z) {  // a scalar base + one 24-bit multiply
        const float* q = (const float*)((const char*)p.inp_pos + (size_t)__umul24((uint32_t)j, 12u));
        x = q[0];
        y = q[1];
        z = q[2];
    };
    auto geom = [&](int t, int j, float nv, float x, float y, float z) -> P16Rec {
        P16Rec c;
        const bool pp = t >= nbA;
        x -= pp ? oxs[1] : oxs[0];
        y -= pp ? oys[1] : oys[0];
        z -= pp ? ozs[1] : ozs[0];
        float a = window_value(window, nval ? nv : rel_dist2(x, y, z), p.inv_r2, p.window_fac);
        if (imp) a *= imp[j];
        a = valid(t) ? a : 0.0f;  // lanes past the row's end: a pair of weight zero in class 0, features out of range
        filter_coords<false>(x, y, z, p);
        x = fminf(3.0f, fmaxf(0.0f, x));
        y = fminf(3.0f, fmaxf(0.0f, y));
        z = fminf(3.0f, fmaxf(0.0f, z));
        const float xf = fminf(floorf(x), 2.0f), yf = fminf(floorf(y), 2.0f), zf = fminf(floorf(z), 2.0f);
        const float fx = x - xf, fy = y - yf, fz = z - zf;
        c.bz = (int)zf;
        c.cls4 = valid(t) ? 4 * ((int)yf * 3 + (int)xf) : 0;
        const float a0 = a * (1.0f - fz), a1 = a * fz;
        const float y00 = (1.0f - fy) * (1.0f - fx), y01 = (1.0f - fy) * fx, y10 = fy * (1.0f - fx), y11 = fy * fx;
        c.lo = (f32x4){a0 * y00, a0 * y01, a0 * y10, a0 * y11};
        c.hi = (f32x4){a1 * y00, a1 * y01, a1 * y10, a1 * y11};
        return c;
    };
    auto push_index = [&](int t, int j) { Jof[lane] = valid(t) ? __umul24((uint32_t)j, rowB) : kGOob; };
    // the records of half h (owner lanes 32 h .. 32 h + 31) into the cleared staging: 8 products at planes bz, bz + 1
    auto push_rec = [&](int h, const P16Rec& c) {
        // clear: 8 groups x 68 floats = 136 x 16 bytes
        {
            f32x4* z4 = (f32x4*)Rec;
            z4[lane] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            z4[64 + lane] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (lane < 8) z4[128 + lane] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
        gfence();
        if ((lane >> 5) == h) {
            const int k = lane & 31;
            float* r = Rec + (k >> 2) * kGRecG + (k & 3) + 16 * c.bz;
            r[0] = c.lo.x; r[4] = c.lo.y; r[8] = c.lo.z; r[12] = c.lo.w;
            r[16] = c.hi.x; r[20] = c.hi.y; r[24] = c.hi.z; r[28] = c.hi.w;
        }
    };
    // The class bytes of the 64 pairs, four per scalar register: packed inside each quad with two DPP moves, read out of lanes
    // 0, 4, 8, ... (the splat extracts a pair's byte with scalar instructions)
    auto pack_classes = [&](int cls4, uint32_t (&c)[16]) {
        int pk = cls4 | (__builtin_amdgcn_mov_dpp(cls4, 0xb1, 0xf, 0xf, true) << 8);   // quad_perm [1, 0, 3, 2]
        pk = pk | (__builtin_amdgcn_mov_dpp(pk, 0x4e, 0xf, 0xf, true) << 16);          // quad_perm [2, 3, 0, 1]
#pragma unroll
        for (int m = 0; m < 16; ++m) c[m] = (uint32_t)__builtin_amdgcn_readlane(pk, 4 * m);
    };
    // feature rows of half h of the batch whose offsets are in Jof: two rounds of 16 pairs, lane = (pair fr, channel quad fq)
    auto f_issue = [&](int h, f32x4 (&f)[2]) {
        uint32_t jo[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) jo[r] = Jof[32 * h + 16 * r + fr];
#pragma unroll
        for (int r = 0; r < 2; ++r)
            f[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rF, __builtin_elementwise_add_sat(jo[r], cbyte), 0, 0));
    };
    // ... transposed by the stores into the pair-interleaved layout: channel c of pair 4 g + t at Fst[g * 64 + 4 c' + t] with
    // c' = 4 (c & 3) + (c >> 2) (the stores of a lane group then hit 16 banks twice: free for 4-byte stores)
    float* const wf = Fst + (fr >> 2) * 64 + 4 * fq + (fr & 3);
    auto f_publish = [&](const f32x4 (&f)[2]) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            wf[256 * r] = f[r].x;
            wf[256 * r + 16] = f[r].y;
            wf[256 * r + 32] = f[r].z;
            wf[256 * r + 48] = f[r].w;
        }
    };
    // LDS byte addresses of this lane's operands of group 0: product (plane zl, row lane & 3), channel ch
    const uint32_t a_rec = glds(Rec + 4 * (4 * zl + (lane & 3)));
    const uint32_t a_fst = glds(Fst + 4 * (4 * (ch & 3) + (ch >> 2)));
    // half h of a batch: 32 pairs at fixed staging addresses, nblk blocks of 8 (tools/gen_p16_splat.py)
    auto splat = [&](int nblk, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t c4, uint32_t c5, uint32_t c6, uint32_t c7) {
        uint32_t s0;
        __builtin_amdgcn_s_setprio(3);
        asm volatile(
#include "cconv_p16_splat.inc"
            : [s0] "=&s"(s0)
            : [pa] "v"(a_rec), [pf] "v"(a_fst), [nb] "s"(nblk), [c0] "s"(c0), [c1] "s"(c1), [c2] "s"(c2), [c3] "s"(c3),
              [c4] "s"(c4), [c5] "s"(c5), [c6] "s"(c6), [c7] "s"(c7)
            : "scc", "m0", "memory", P16_FIXED_REGS);
        __builtin_amdgcn_s_setprio(0);
    };
    // a point is done: fold its 9 tiles onto the lane's 16 cells and write them to the point's B row; clear the tiles
    auto merge = [&](int pt) {
        const uint32_t rowa = glds(Bt + pt * kGRow + zl * 256 + ((ch ^ (pt & 15)) << 2));
        gfence();  // (the staging lives in the row of the second point)
        asm volatile(
#include "cconv_p16_merge.inc"
            :: [b] "v"(rowa) : "memory", P16_FIXED_REGS);
        zero_tiles();
        gfence();
    };

    if (nbA == 0) merge(wave);
    if (NB > 0) {
        // stages: indices two batches ahead, positions one ahead; geometry + index push of batch t + 1 between the two halves of
        // batch t, its first feature loads before the second half's splat (the other three waves of the SIMD cover the rest)
        int jA, jB;
        float nvA, nvB, px, py, pz;
        uint32_t cc[16];
        f32x4 ff[2];
        P16Rec cur;
        ld_idx(0, jA, nvA);
        ld_idx(1, jB, nvB);
        ld_pos(jA, px, py, pz);
        cur = geom(0, jA, nvA, px, py, pz);
        push_index(0, jA);
        push_rec(0, cur);
        pack_classes(cur.cls4, cc);
        gfence();
        f_issue(0, ff);
        jA = jB;
        nvA = nvB;
        ld_pos(jA, px, py, pz);
        for (int t = 0; t < NB; ++t) {
            // here: Rec = the records of half 0 of batch t, cc = its classes, ff = the features of half 0 (in flight), cur = the
            // records of batch t (lanes 32 .. 63: half 1), (jA, nvA, px, py, pz) = batch t + 1
            const bool more = t + 1 < NB;
            const int np = npairs(t);
            const bool two = np > 32;
            f_publish(ff);
            if (more) ld_idx(t + 2, jB, nvB);
            if (two) f_issue(1, ff);
            gfence();
            splat((min(np, 32) + 7) >> 3, cc[0], cc[1], cc[2], cc[3], cc[4], cc[5], cc[6], cc[7]);
            P16Rec nxt;
            if (more) nxt = geom(t + 1, jA, nvA, px, py, pz);
            if (two) {
                gfence();
                push_rec(1, cur);
                f_publish(ff);
            }
            gfence();
            if (more) {
                push_index(t + 1, jA);
                jA = jB;
                nvA = nvB;
                ld_pos(jA, px, py, pz);
                gfence();
                f_issue(0, ff);
            }
            if (two) {
                gfence();
                splat((np - 32 + 7) >> 3, cc[8], cc[9], cc[10], cc[11], cc[12], cc[13], cc[14], cc[15]);
            }
            if (t == nbA - 1) merge(wave);
            if (more) {
                gfence();
                push_rec(0, nxt);
                pack_classes(nxt.cls4, cc);
                cur = nxt;
            }
        }
    }
    merge(wave + kGWaves);
    __syncthreads();

    // ---------------- contraction of the (single) channel chunk on the matrix cores: as in cconv_cls.hip ----------------
    f32x4 acc[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    {
        const int nq = (cin + 3) >> 2;
        for (int t = wave; t < 16 * nq; t += kGWaves) {
            int tq, tr;
                blk_divmod(t, nq, tq, tr);
                const int blk = tq * 4 + tr;
            const f32x4 av = *(const f32x4*)(Bt + (size_t)mi * kGRow + ((blk * 16 + mg * 4) ^ (mi << 2)));
            const float* wb = p.Wp + ((size_t)(blk * 4 + mg) * p.NT * 16 + mi) * 4;
            const uint32_t wm = p.wmask >> (4 * (tr));  // (all-zero filter blocks of a block-diagonal pair of layers: skipped)
#pragma unroll
            for (int n = 0; n < NTT; ++n) {
                if (n < p.NT && ((wm >> n) & 1)) {
                    const f32x4 bv = *(const f32x4*)(wb + n * 64);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[n], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();

    // ---------------- cross-wave reduction + epilogue ----------------
    float* red = Bt;  // [kGWaves][16][16 * NT]
    const int ncol = 16 * p.NT;
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        if (n < p.NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * 16 + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < GTM * cout; e += kGThreads) {
        const int ptt = e / cout, o = e % cout;
        const int64_t ii = pt0 + ptt;
        if (ii >= p.n_out) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kGWaves; ++w) v += red[((size_t)w * 16 + ptt) * ncol + o];
        if (p.bias) v += p.bias[o];
        float* dst = p.out + ii * cout + o;
        if (p.flags & DMCF_FLAG_ACCUMULATE) v += *dst;
        *dst = v;
    }
}

static constexpr size_t kP16Lds = (size_t)(GTM * kGRow + kGWaves * kGFst) * sizeof(float);

// Same filters and flags as cconv_z3.hip / cconv_pair.hip (no antisymmetric form); 4 .. 16 input channels.
bool cconv_p16_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "g16": force, anything else: never
    if (e && e[0] != 'g') return false;
    if (dx != 4 || dy != 4 || dz != 4) return false;
    if (a->flags & DMCF_FLAG_SYMMETRIC) return false;
    if (a->coordinate_mapping != DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING || a->interpolation != DMCF_INTERP_LINEAR ||
        !(a->flags & DMCF_FLAG_ALIGN_CORNERS) || (a->flags & DMCF_FLAG_NORMALIZE))
        return false;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin & 3) || cin > 16 || cout > 16 * kGMaxNT) return false;
    if ((uintptr_t)a->inp_features & 15) return false;
    // 24-bit multiplies form the byte offsets of feature and position rows; the buffers must stay below 2 GB
    if (a->n_inp >= (1 << 24) || a->n_inp * (int64_t)cin * 4 >= ((int64_t)1 << 31)) return false;
    if (e) return true;
    return false;
}

int cconv_p16_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream) {
    const int NT = (p.cout + 15) / 16;
    float* packed = (float*)workspace;
    const int nchunks = cconv_cls_pack(a, packed, stream);  // the B-fragment order of cconv_cls.hip (one 16-channel chunk)
    p.Wp = packed;
    p.NT = NT;
    p.nchunks = nchunks;
    const int64_t ntiles = (p.n_out + GTM - 1) / GTM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    const void* fn;
    if (cconv_plain(a))
        fn = NT <= 1 ? (const void*)cconv_p16_kernel<1, true>
                     : (NT <= 2 ? (const void*)cconv_p16_kernel<2, true> : (const void*)cconv_p16_kernel<4, true>);
    else
        fn = NT <= 1 ? (const void*)cconv_p16_kernel<1, false>
                     : (NT <= 2 ? (const void*)cconv_p16_kernel<2, false> : (const void*)cconv_p16_kernel<4, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kP16Lds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(kGThreads), kargs, kP16Lds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

}  // namespace dmcf
