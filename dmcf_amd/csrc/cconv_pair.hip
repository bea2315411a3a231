// CConv for 4x4x4 filters and up to 32 input channels: ONE NEIGHBOUR PAIR PER MATRIX INSTRUCTION with NO wasted products --
// splat F, v_mfma_f32_4x4x1_16B_f32 into 27 class tiles.
//
// A pair's trilinear footprint is 2 x 2 x 2 cells of the 4 x 4 x 4 filter; its base cell (bz, by, bx), each in 0 .. 2, is one of
// 27 CLASSES.  The 16-block form of the fp32 matrix instruction computes 16 independent 4 x 4 outer products in 8 clocks; with
//
//     block b = (z' in 0..1, channel quad in 0..7),   row i = (y', x'),   column j = channel inside the quad
//     A[b][i] = a w_z[z'] w_y[y'] w_x[x']   (the pair's 8 products),      B[b][j] = f[4 quad + j]
//
// ONE instruction adds ONE pair to its 8 cells x 32 channels, and every product is used (splat E, cconv_z3.hip, spends 32
// matrix clocks per pair on 32 cells x 32 channels, three quarters of them on zero rows; splat D, cconv_cls.hip, 8.8 per 16
// channels).  The accumulator is the tile of the pair's class: 27 tiles x 4 registers hold B_i with every cell in up to 8
// tiles, addressed relative to M0 = 4 * class (s_set_gpr_idx_on: the class is data, not control flow).  Nothing is ordered,
// nothing is padded, the A operand needs no arithmetic in the splat: per pair one scalar bit-field extract, one M0 update,
// one matrix instruction and half an LDS read (products and features are staged pair-interleaved and read four pairs at a
// time).  Measured (tools/ubench/pair_splat.hip): 12.7 - 13.4 clocks per pair and SIMD against 10.9 for the bare matrix
// instructions (17.1 with row-major features, one ds_read_b32 per pair).
//
// The price is registers: 108 for the tiles.  So a workgroup is 8 waves (two per SIMD, 256 registers each), one workgroup
// per CU, and a wave owns TWO output points of the 16-point tile, one after the other as one stream of 64-pair batches (loads
// run ahead across the point boundary); the first point's merged tile goes to the B tile when it is done: channels 0 .. 15
// to the point's own row, channels 16 .. 31 -- the second chunk of the contraction -- raw into the row of the wave's second
// point, free until that point is merged.
//
//   registers: v0 .. v115 the compiler (amdgpu_num_vgpr), v116 .. v147 operand buffers of the splat, v148 .. v255 the class
//              tiles (outside the compiler's allocation: every asm statement that touches them lists them as clobbered;
//              tests/test_fixed_registers.py)
//   LDS:       B tile [16 points][64 cells x 16 channels] 64 KB (one 16-channel chunk at a time, as in cconv_cls.hip) +
//              per wave the records [16 groups][8 products][4 pairs] (group stride 36 floats: conflict-free stores), the
//              features [16 groups][32 channels][4 pairs] and an index buffer: 148 KB
//
// Per batch: geometry (lane = pair), the 8 products and the class; feature rows by 16-byte loads, lane = (pair, channel quad),
// transposed into the pair-interleaved layout by their LDS stores (transposing with 4-byte LOADS instead costs 34 memory
// instructions per batch instead of 10, and the texture addresser then bounds the kernel -- measured, tools/ptrace.py: 920
// clocks per batch in the issue phase); class bytes packed inside each quad by two DPP moves and read out with 16 v_readlane
// (a v_readlane per pair would cost the SIMD 8 clocks each).  With two waves
// per SIMD nothing hides a round trip to memory, so every load is issued a whole splat (~1500 clocks) before its first use:
// the 32 feature loads of batch t + 1 fly during the splat of batch t, positions are requested two batches ahead, indices three.
// When a point is done its tiles are merged in registers (tools/gen_pair_splat.py: 60 adds inside the lane, 16 half-wave
// swaps, 32 row-masked adds) and handed to the shared contraction in chunks of 16 channels.
//
// Accumulation order = list order inside a class, then the fixed merge order: deterministic.
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {
// Diagnostic build (make -C dmcf_amd/csrc pair_trace -> variants/PTRACE.so, read by tools/ptrace.py): cycle stamps at the phase
// boundaries of the kernel, summed over every 16th tile.  Compiled out of the product library.
#ifdef PX_TRACE
__device__ unsigned long long g_ptrace[24];
#define PT(k) { const uint64_t now_ = __builtin_readcyclecounter(); pt[k] += now_ - plast; plast = now_; }
#else
#define PT(k)
#endif

constexpr int kPWaves = 8;
constexpr int kPThreads = 64 * kPWaves;
constexpr int PTM = 2 * kPWaves;   // output points per workgroup = rows of the B tile
constexpr int kPRow = 1024;        // floats per B row: k' = (z * 4 + y) * 64 + channel * 4 + x (16 channels)
constexpr int kPRecG = 36;         // floats per record group: 8 products x 4 pairs, padded (bank = 4 g + 4 q + t)
constexpr int kPRec = 16 * kPRecG;
constexpr int kPFst = 16 * 128;    // [16 groups][32 channels (permuted)][4 pairs]
constexpr int kPWaveF = kPRec + kPFst + 64;  // + the index buffer
constexpr int kPMaxNT = 4;
constexpr int kPCompilerVgprs = 58;  // (the attribute counts HALF of the unified file: v0 .. v115)

#define PAIR_FIXED_REGS                                                                                                    \
    "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128",      \
        "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142",      \
        "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156",      \
        "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170",      \
        "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184",      \
        "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198",      \
        "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212",      \
        "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226",      \
        "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240",      \
        "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

__device__ __forceinline__ void pfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t plds(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q;
}

__device__ __forceinline__ void pair_zero_tiles() {
    asm volatile(
        ".irp r,148,150,152,154,156,158,160,162,164,166,168,170,172,174,176,178,180,182,184,186,188,190,192,194,196,198,200,202,"
        "204,206,208,210,212,214,216,218,220,222,224,226,228,230,232,234,236,238,240,242,244,246,248,250,252,254\n\t"
        "v_mov_b64 v[\\r:\\r+1], 0\n\t.endr" ::: "memory", PAIR_FIXED_REGS);
}

struct PairRec {  // per pair, in the registers of its owner lane
    f32x4 lo, hi;  // a w_z[z'] w_y[y'] w_x[x'], index 2 y' + x', for z' = 0 / 1
    int cls4;      // 4 * ((bz * 3 + by) * 3 + bx)
};

typedef uint32_t u32x4p __attribute__((ext_vector_type(4)));
constexpr uint32_t kPOob = 0xffffffffu;  // a byte offset no buffer holds: the load returns zeros

// PLAIN: see cconv_plain() in cconv_common.h
template <int NTT, bool PLAIN>
__global__ __launch_bounds__(kPThreads, 1) __attribute__((amdgpu_num_vgpr(kPCompilerVgprs))) void cconv_pair_kernel(const CconvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = p.cin, cout = p.cout;
    const int window = PLAIN ? (int)DMCF_WINDOW_POLY6 : p.window;
    const float* const nval = PLAIN ? nullptr : p.nval;
    const float* const imp = PLAIN ? nullptr : p.inp_imp;
    float* Bt = smem;                                    // [PTM][kPRow], 4-float groups XOR-swizzled by the row
    float* Rec = smem + PTM * kPRow + wave * kPWaveF;    // [16 groups][kPRecG]: product q of pair 4 g + t at g * kPRecG + 4 q + t
    float* Fst = Rec + kPRec;                            // [16 groups][32 channels (permuted)][4 pairs]
    uint32_t* Jof = (uint32_t*)(Fst + kPFst);            // [64]: byte offset of the pair's feature row (kPOob: no pair)
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * PTM;

    // splat roles: this lane's channel (B operand, accumulator column) and plane offset z'
    const int ch = lane & 31, half = lane >> 5;
    // feature load roles: lane -> (pair fr of a round of 8, channels 4 fq .. 4 fq + 3)
    const int fr = lane >> 3, fq = lane & 7;
    const uint32_t rowB = (uint32_t)cin * 4u;
    const uint32_t cbyte = 4 * fq < cin ? 16u * (uint32_t)fq : kPOob;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)p.inp_feat, 0, (int)((uint32_t)p.n_inp * rowB), 0x00020000);
    // contraction roles
    const int mi = lane & 15, mg = lane >> 4;

    // The batches of the wave's two points form ONE stream (point A's batches, then point B's), as in cconv_cls.hip
    int64_t rbs[2];
    int nts[2], nbs[2];
    float oxs[2], oys[2], ozs[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        const int64_t i = pt0 + wave + kPWaves * pp;
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
    const int nt0 = nts[0], nt1 = nts[1];
    // ONE DENSE STREAM over the wave's two rows: positions [0, nt0) are point A's pairs, [padA, padA + nt1) point B's, padA =
    // nt0 rounded up to a block of 8 (the splat runs in blocks of 8 pairs, and a block feeds ONE point's tiles).  Batch t =
    // positions 64 t .. 64 t + 63; the batch that holds padA is split at that block: A's blocks, the merge of A, B's blocks.
    // A row of 265 pairs is 4.14 batches but 5 iterations when every point starts a batch of its own; two of them are 8.3 -> 9.
    const int padA = (nt0 + 7) & ~7;
    const int S = padA + nt1;
    const int NB = (S + 63) >> 6;
    const int tb = padA >> 6, bb = (padA & 63) >> 3;  // A's tiles are complete before block bb of batch tb
    const int64_t rb0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rbs[0] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rbs[0]);
    const int64_t rb1 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rbs[1] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rbs[1]);
    // ONE buffer over both rows of the wave (they are 8 rows apart in the list): offsets past a row's end are replaced by an
    // out-of-range one and read as index 0 (a valid point, unused)
    const int64_t gapB = nt1 > 0 ? rb1 - rb0 : 0;
    const bool near = gapB >= 0 && gapB + nt1 < ((int64_t)1 << 29) && nt0 < (1 << 29);
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.idx + rb0), 0, near ? (int)(max((int64_t)nt0, gapB + nt1) * 4) : 0, 0x00020000);
    const uint32_t offB = (uint32_t)gapB * 4u;

    pair_zero_tiles();

    // stream positions of batch t that exist (wave uniform), and whether this lane's position holds a pair
    auto npairs = [&](int t) -> int { return max(0, min(64, S - 64 * t)); };
    auto in_a = [&](int t) -> bool { return 64 * t + lane < nt0; };
    auto in_b = [&](int t) -> bool { return 64 * t + lane >= padA && 64 * t + lane < S; };
    auto valid = [&](int t) -> bool { return in_a(t) || in_b(t); };
    // (eligibility bounds n_inp -- and with it every row -- by 2^24 entries: the two rows always fit one buffer)
    if (!near) __builtin_trap();
    auto ld_idx = [&](int t, int& j, float& nv) {
        const bool a = in_a(t), b = in_b(t);
        const int sp = 64 * t + lane;
        // (row A's entry sp, or row B's entry sp - padA, which sits gapB entries behind row A's start)
        const uint32_t off = a ? (uint32_t)sp * 4u : (b ? (uint32_t)(sp - padA) * 4u + offB : kPOob);
        j = (int)__builtin_amdgcn_raw_buffer_load_b32(rI, off, 0, 0);
        nv = 0.0f;
        if (nval && (a || b)) nv = nval[a ? rb0 + sp : rb1 + (sp - padA)];
    };
    auto ld_pos = [&](int j, float& x, float& y, float& z) {  // a scalar base + one 24-bit multiply
        const float* q = (const float*)((const char*)p.inp_pos + (size_t)__umul24((uint32_t)j, 12u));
        x = q[0];
        y = q[1];
        z = q[2];
    };
    auto geom = [&](int t, int j, float nv, float x, float y, float z) -> PairRec {
        PairRec c;
#ifdef PX_NOGEOM
        c.cls4 = 4 * (lane % 27);
        c.lo = (f32x4){x, y, z, nv};
        c.hi = c.lo;
        return c;
#endif
        const bool pp = 64 * t + lane >= padA;  // (per lane: a batch may hold the end of row A and the start of row B)
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
        c.cls4 = valid(t) ? 4 * (((int)zf * 3 + (int)yf) * 3 + (int)xf) : 0;
        const float a0 = a * (1.0f - fz), a1 = a * fz;
        const float y00 = (1.0f - fy) * (1.0f - fx), y01 = (1.0f - fy) * fx, y10 = fy * (1.0f - fx), y11 = fy * fx;
        c.lo = (f32x4){a0 * y00, a0 * y01, a0 * y10, a0 * y11};
        c.hi = (f32x4){a1 * y00, a1 * y01, a1 * y10, a1 * y11};
        return c;
    };
    // byte offset of the pair's feature row (lanes without a pair: out of range, the loads return zeros)
    auto push_index = [&](int t, int j) { Jof[lane] = valid(t) ? __umul24((uint32_t)j, rowB) : kPOob; };
    auto push_rec = [&](const PairRec& c) {
#ifdef PX_NOREC
        return;
#endif
        float* r = Rec + (lane >> 2) * kPRecG + (lane & 3);
        r[0] = c.lo.x; r[4] = c.lo.y; r[8] = c.lo.z; r[12] = c.lo.w;
        r[16] = c.hi.x; r[20] = c.hi.y; r[24] = c.hi.z; r[28] = c.hi.w;
    };
    // The class bytes of the 64 pairs, four per scalar register: packed inside each quad with two DPP moves, read out of lanes
    // 0, 4, 8, ... (16 v_readlane per batch; the splat extracts a pair's byte with scalar instructions)
    int pk_cur = 0;  // the packed class bytes of the staged batch (lanes 0, 4, 8, ...: four pairs each)
    auto pack_classes = [&](int cls4, uint32_t (&c)[16]) {
#ifdef PX_NOPACK
        return;
#endif
        int pk = cls4 | (__builtin_amdgcn_mov_dpp(cls4, 0xb1, 0xf, 0xf, true) << 8);   // quad_perm [1, 0, 3, 2]
        pk = pk | (__builtin_amdgcn_mov_dpp(pk, 0x4e, 0xf, 0xf, true) << 16);          // quad_perm [2, 3, 0, 1]
        pk_cur = pk;
#pragma unroll
        for (int m = 0; m < 16; ++m) c[m] = (uint32_t)__builtin_amdgcn_readlane(pk, 4 * m);
    };
    // feature rows of the batch whose offsets are in Jof: eight rounds (four for a batch of at most 32 pairs) of 8 pairs, lane =
    // (pair fr of the round, channel quad fq), one 16-byte load each -- row-major staging [pair][32 channels]
    auto f_issue = [&](int np, f32x4 (&f)[8]) {
#ifdef PX_NOFEAT  // diagnostic builds (make -C dmcf_amd/csrc pair_variants; wrong results, right timing)
        return;
#endif
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (hh == 0 || np > 32) {
                uint32_t jo[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) jo[r] = Jof[8 * (4 * hh + r) + fr];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    f[4 * hh + r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rF, __builtin_elementwise_add_sat(jo[r], cbyte), 0, 0));
            }
        }
    };
    // ... and transposed by the stores into the pair-interleaved layout the splat reads four pairs at a time: channel c of pair
    // 4 g + t at Fst[g * 128 + 4 c' + t] with c' = 8 (c & 3) + (((c >> 2) + 4 ((c >> 1) & 1)) & 7), a permutation of the channels
    // under which these 4-byte stores (two ds_write2_b32 per loaded row part) AND the splat's 16-byte reads are conflict free
    const int pg = fr >> 2, ptq = fr & 3;
    float* const w01 = Fst + pg * 128 + 4 * fq + ptq;             // channels 4 fq, 4 fq + 1: + 0, + 32
    float* const w23 = Fst + pg * 128 + 4 * ((fq + 4) & 7) + ptq; // channels 4 fq + 2, 4 fq + 3: + 64, + 96
    auto f_publish = [&](int np, const f32x4 (&f)[8]) {
#ifdef PX_NOPUBLISH
        return;
#endif
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < 4 || np > 32) {
                w01[256 * r] = f[r].x;
                w01[256 * r + 32] = f[r].y;
                w23[256 * r + 64] = f[r].z;
                w23[256 * r + 96] = f[r].w;
            }
        }
    };
    // LDS byte addresses of this lane's operands of group 0: product 4 z' + (y', x') of the lane's block row, its channel
    const uint32_t a_rec0 = plds(Rec + 4 * (4 * half + (lane & 3)));
    const uint32_t a_fst0 = plds(Fst + 4 * (8 * (ch & 3) + (((ch >> 2) + 4 * ((ch >> 1) & 1)) & 7)));
    // One batch: 64 pairs at fixed staging addresses, `nblk` blocks of 8 (tools/gen_pair_splat.py)
    auto splat = [&](int nblk, const uint32_t (&c)[16], uint32_t a_rec, uint32_t a_fst) {
#ifdef PX_NOSPLAT
        return;
#endif
        uint32_t s0;
        __builtin_amdgcn_s_setprio(3);
        asm volatile(
#include "cconv_pair_splat.inc"
            : [s0] "=&s"(s0)
            : [pa] "v"(a_rec), [pf] "v"(a_fst), [nb] "s"(nblk), [c0] "s"(c[0]), [c1] "s"(c[1]), [c2] "s"(c[2]), [c3] "s"(c[3]),
              [c4] "s"(c[4]), [c5] "s"(c[5]), [c6] "s"(c[6]), [c7] "s"(c[7]), [c8] "s"(c[8]), [c9] "s"(c[9]), [c10] "s"(c[10]),
              [c11] "s"(c[11]), [c12] "s"(c[12]), [c13] "s"(c[13]), [c14] "s"(c[14]), [c15] "s"(c[15])
            : "scc", "m0", "memory", PAIR_FIXED_REGS);
        __builtin_amdgcn_s_setprio(0);
    };
    const int ptA = wave, ptB = wave + kPWaves;
    // this lane's part of a point's B row (its half-wave's planes, its channel's column) and of the parking area
    const uint32_t rowA = plds(Bt + ptA * kPRow + half * 512 + (((ch & 15) ^ (ptA & 15)) << 2));
    const uint32_t rowB_ = plds(Bt + ptB * kPRow + half * 512 + (((ch & 15) ^ (ptB & 15)) << 2));
    const uint32_t parkB = plds(Bt + ptB * kPRow + (ch & 15) + 16 * half);
    // Point A is done: merge its tiles in registers (lanes 0 .. 31: planes 0, 1; lanes 32 .. 63: planes 2, 3 of the lane's
    // channel).  Channels 0 .. 15 go to the point's B row; channels 16 .. 31 (the second chunk) wait raw in the row of point B,
    // which is free until that point is merged.  Then the tiles are cleared for point B.
    auto merge_first = [&]() {
#ifdef PX_NOMERGE
        return;
#endif
        asm volatile(
#include "cconv_pair_merge.inc"
            ::: "memory", PAIR_FIXED_REGS);
        if (ch < 16) {
            asm volatile(
#include "cconv_pair_store.inc"
                :: [b] "v"(rowA) : "memory", PAIR_FIXED_REGS);
        } else if (p.nchunks > 1) {
            asm volatile(
#include "cconv_pair_park.inc"
                :: [b] "v"(parkB) : "memory", PAIR_FIXED_REGS);
        }
        asm volatile(
#include "cconv_pair_zero.inc"
            ::: "memory", PAIR_FIXED_REGS);
    };

#ifdef PX_TRACE
    uint64_t pt[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t plast = __builtin_readcyclecounter();
    const uint64_t pstart = plast;
#endif
    bool a_done = false;
    if (padA == 0) {
        merge_first();
        a_done = true;
    }
    if (NB > 0) {
        // Stages (nothing hides a round trip at two waves per SIMD, so every load is issued a whole splat before its first
        // use): indices three batches ahead, positions two, geometry + index push + ALL feature loads of batch t + 1 before
        // the splat of batch t, published after it.
        int j1, j2;
        float nv1, nv2, px, py, pz;
        uint32_t cc[16];
        f32x4 ff[8];
        {
            int j0;
            float nv0, qx, qy, qz;
            ld_idx(0, j0, nv0);
            ld_idx(1, j1, nv1);
            ld_idx(2, j2, nv2);
            ld_pos(j0, qx, qy, qz);
            ld_pos(j1, px, py, pz);
            const PairRec first = geom(0, j0, nv0, qx, qy, qz);
            push_index(0, j0);
            push_rec(first);
            pack_classes(first.cls4, cc);
            pfence();
            f_issue(npairs(0), ff);
            f_publish(npairs(0), ff);
            pfence();
        }
        PT(0)
        for (int t = 0; t < NB; ++t) {
            // here: Rec / Fst / cc = batch t; (j1, nv1, px, py, pz) = batch t + 1; (j2, nv2) = the indices of batch t + 2
            const bool more = t + 1 < NB;
            const int np = npairs(t), np1 = more ? npairs(t + 1) : 0;
            PairRec nxt;
            // what this iteration requests lands in its OWN registers and moves to the loop-carried ones after the splat: a copy
            // placed before it would wait for every load in flight (the counter is in order)
            int jn;
            float nvn, qx, qy, qz;
            if (more) {
                nxt = geom(t + 1, j1, nv1, px, py, pz);
                push_index(t + 1, j1);
                pfence();
                PT(1)
                f_issue(np1, ff);
            }
            // (unconditional -- past the stream's end the index load is out of the buffer's range and returns entry 0 -- so that
            // the loaded registers are not merged with constants of another path: such a merge is a copy behind a full wait)
            ld_pos(j2, qx, qy, qz);
            ld_idx(t + 3, jn, nvn);
            PT(2)
            const int nblk = (np + 7) >> 3;
            // ONE splat site: the batch that holds the point boundary runs it twice -- A's blocks, A's merge, then B's blocks
            // with the staging addresses and the class bytes moved on by bb blocks (two groups of four pairs each)
            int b0 = 0, b1 = (t == tb && !a_done) ? bb : nblk;
            for (;;) {
                if (b1 > b0) {
                    if (b0 > 0) {  // (cc is dead after this batch: the next one packs its own)
#pragma unroll
                        for (int m = 0; m < 16; ++m) cc[m] = (uint32_t)__builtin_amdgcn_readlane(pk_cur, (4 * m + 8 * b0) & 63);
                    }
                    splat(b1 - b0, cc, a_rec0 + (uint32_t)(2 * kPRecG * 4) * (uint32_t)b0, a_fst0 + 1024u * (uint32_t)b0);
                }
                if (t != tb || a_done) break;
                PT(3)
                merge_first();
                a_done = true;
                b0 = bb;
                b1 = nblk;
            }
            PT(4)
            // (the compiler may not move the rotation of the in-flight loads -- register copies, each behind a wait for every
            // load issued before it -- in front of the splat: volatile asm statements keep their order)
            asm volatile("" : "+v"(jn), "+v"(nvn), "+v"(qx), "+v"(qy), "+v"(qz));
            if (more) {
                pfence();
                PT(5)
                f_publish(np1, ff);
                PT(6)
                push_rec(nxt);
                pack_classes(nxt.cls4, cc);
                j1 = j2;
                nv1 = nv2;
                j2 = jn;
                nv2 = nvn;
                px = qx;
                py = qy;
                pz = qz;
                pfence();
                PT(7)
            }
        }
    }
    PT(8)
    if (!a_done) merge_first();  // (row A ends exactly at the end of the stream's last batch and row B is empty)
    // point B: merge in place
    asm volatile(
#include "cconv_pair_merge.inc"
        ::: "memory", PAIR_FIXED_REGS);
    // (point A's second chunk comes back from the row of point B -- into the operand buffers, idle now -- before that row is written)
    if (ch >= 16 && p.nchunks > 1) {
        asm volatile(
#include "cconv_pair_unpark.inc"
            :: [b] "v"(parkB) : "memory", PAIR_FIXED_REGS);
    }

    PT(9)
    // ---------------- contraction with the packed filter (the B-fragment order of cconv_cls.hip), one 16-channel chunk at a time
    f32x4 acc[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    constexpr int kIt = 64 / kPWaves;          // blocks of a chunk per wave: t = wave + 8 it < 16 nq
    constexpr int kPre = NTT <= 2 ? 8 : 4;     // blocks whose filter fragments are requested together (NTT <= 2: a whole chunk)
    auto nq_of = [&](int chunk) { return (min(16, cin - 16 * chunk) + 3) >> 2; };
    // The fragments come through a buffer resource over the packed filter: the lane's part of the address (its row of the
    // fragment) is formed once, a block's and a column tile's part is a SCALAR offset -- one memory instruction and two scalar
    // ones per fragment instead of a 64-bit multiply-add chain per lane (the paired 24 -> 64 layer requests ~36 per tile).
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wp, 0, (int)((uint32_t)p.nchunks * 64u * (uint32_t)p.NT * 1024u), 0x00020000);
    const uint32_t w_lane = ((uint32_t)mg * (uint32_t)p.NT * 16u + (uint32_t)mi) * 16u;
    auto w_issue = [&](int chunk, int it0, f32x4 (&bw)[kPre][NTT]) {
        const int nq = nq_of(chunk);
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int it = it0 + q;
            if (kPWaves * it < 16 * nq) {
                const int t = wave + kPWaves * it;
                int tq, tr;
                blk_divmod(t, nq, tq, tr);
                const int blk = tq * 4 + tr;
                const uint32_t w_blk = (uint32_t)(chunk * 64 + blk) * (uint32_t)p.NT * 1024u;
                const uint32_t wm = p.wmask >> (4 * (4 * chunk + tr));  // (all-zero filter blocks: not fetched; cin <= 32: quads 0 .. 7)
#pragma unroll
                for (int n = 0; n < NTT; ++n)
                    if (n < p.NT && ((wm >> n) & 1))
                        bw[q][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, w_lane, w_blk + 256u * (uint32_t)n, 0));
            }
        }
    };
    const float out_prev = epilogue_prefetch(p, pt0, PTM, tid);
    f32x4 bw[kPre][NTT];
    w_issue(0, 0, bw);  // the first fragments' round trip runs under the tile's first barrier
#pragma unroll 1
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        if ((ch >> 4) == chunk) {
            // this lane's channel belongs to the chunk: its 2 planes x 16 cells go to the B rows (point A's first chunk is there
            // already; its second comes from the registers it was unparked into)
            if (chunk == 1) {
                asm volatile(
#include "cconv_pair_store_parked.inc"
                    :: [b] "v"(rowA) : "memory", PAIR_FIXED_REGS);
            }
            asm volatile(
#include "cconv_pair_store.inc"
                :: [b] "v"(rowB_) : "memory", PAIR_FIXED_REGS);
        }
        PT(16)
        __syncthreads();
        PT(17)
        const int nq = nq_of(chunk);
#pragma unroll
        for (int it0 = 0; it0 < kIt; it0 += kPre) {
#pragma unroll
            for (int q = 0; q < kPre; ++q) {
                const int it = it0 + q;
                if (kPWaves * it < 16 * nq) {
                    const int t = wave + kPWaves * it;
                    int tq, tr;
                blk_divmod(t, nq, tq, tr);
                const int blk = tq * 4 + tr;
                    const f32x4 av = *(const f32x4*)(Bt + (size_t)mi * kPRow + ((blk * 16 + mg * 4) ^ (mi << 2)));
                    const uint32_t wm = p.wmask >> (4 * (4 * chunk + tr));
#pragma unroll
                    for (int n = 0; n < NTT; ++n) {
                        if (n < p.NT && ((wm >> n) & 1)) {
                            const f32x4 bv = bw[q][n];
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[n], 0, 0, 0);
                        }
                    }
                }
            }
            if (it0 + kPre < kIt) {
                if (kPWaves * (it0 + kPre) < 16 * nq) w_issue(chunk, it0 + kPre, bw);
            } else if (chunk + 1 < p.nchunks) {
                w_issue(chunk + 1, 0, bw);
            }
        }
        PT(18)
        __syncthreads();
        PT(19)
    }

    PT(13)
    // ---------------- cross-wave reduction + epilogue ----------------
    float* red = Bt;  // [kPWaves][16][16 * NT]
    const int ncol = 16 * p.NT;
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        if (n < p.NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * 16 + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        }
    }
    PT(20)
    __syncthreads();
    PT(21)
    for (int e = tid; e < PTM * cout; e += kPThreads) {
        const int ptt = e / cout, o = e % cout;
        const int64_t ii = pt0 + ptt;
        if (ii >= p.n_out) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kPWaves; ++w) v += red[((size_t)w * 16 + ptt) * ncol + o];
        if (p.bias) v += p.bias[o];
        float* dst = p.out + ii * cout + o;
        if (p.flags & DMCF_FLAG_ACCUMULATE) v += e == tid ? out_prev : *dst;
        *dst = v;
    }
#ifdef PX_TRACE
    PT(14)
    if (lane == 0 && (tile & 15) == 0) {
        for (int k = 0; k < 24; ++k)
            if (k < 10 || k > 12) atomicAdd(&g_ptrace[k], pt[k]);
        atomicAdd(&g_ptrace[10], plast - pstart);
        atomicAdd(&g_ptrace[11], 1ull);
        atomicAdd(&g_ptrace[12], (unsigned long long)NB);
    }
#endif
}
#ifdef PX_TRACE
}
extern "C" int dmcf_ptrace(unsigned long long* out) {
    unsigned long long z[24] = {0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dmcf::g_ptrace), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dmcf::g_ptrace), z, sizeof(z));
    return 0;
}
namespace dmcf {
#endif

static constexpr size_t kPairLds = (size_t)(PTM * kPRow + kPWaves * kPWaveF) * sizeof(float);

// Same filters and flags as cconv_z3.hip (no antisymmetric form); 4 .. 32 input channels.
bool cconv_pair_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "pair": force, anything else: never
    if (e && e[0] != 'p') return false;
    if (dx != 4 || dy != 4 || dz != 4) return false;
    if (a->flags & DMCF_FLAG_SYMMETRIC) return false;
    if (a->coordinate_mapping != DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING || a->interpolation != DMCF_INTERP_LINEAR ||
        !(a->flags & DMCF_FLAG_ALIGN_CORNERS) || (a->flags & DMCF_FLAG_NORMALIZE))
        return false;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin & 3) || cin > 32 || cout > 16 * kPMaxNT) return false;
    if ((uintptr_t)a->inp_features & 15) return false;
    // 24-bit multiplies form the byte offsets of feature and position rows; the buffers must stay below 2 GB
    if (a->n_inp >= (1 << 24) || a->n_inp * (int64_t)cin * 4 >= ((int64_t)1 << 31)) return false;
    if (e) return true;
    // more than 16 channels (one walk instead of splat D's two) and rows long enough to pay for the per-point merge: the 3e8-pair
    // layers 24 -> 8 / 24 -> 4 take 7.7 / 4.8 ms here against 8.8 / 6.9 with splat E, the 33-pair layers 3.4 - 4.5 against 2.7 - 3.7
    return cin > 16 && a->row_length_hint == 2;
}

int cconv_pair_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream) {
    const int NT = (p.cout + 15) / 16;
    float* packed = (float*)workspace;
    const int nchunks = cconv_cls_pack(a, packed, stream);  // the B-fragment order of cconv_cls.hip, 16 channels per chunk
    p.Wp = packed;
    p.NT = NT;
    p.nchunks = nchunks;
    const int64_t ntiles = (p.n_out + PTM - 1) / PTM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    const void* fn;
    if (cconv_plain(a))
        fn = NT <= 1 ? (const void*)cconv_pair_kernel<1, true>
                     : (NT <= 2 ? (const void*)cconv_pair_kernel<2, true> : (const void*)cconv_pair_kernel<4, true>);
    else
        fn = NT <= 1 ? (const void*)cconv_pair_kernel<1, false>
                     : (NT <= 2 ? (const void*)cconv_pair_kernel<2, false> : (const void*)cconv_pair_kernel<4, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPairLds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(kPThreads), kargs, kPairLds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

}  // namespace dmcf
