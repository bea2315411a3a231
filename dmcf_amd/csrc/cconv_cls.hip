// CConv / ASCC for 4x4x4 filters: CLASS-SORTED splat on v_mfma_f32_16x16x4_f32, four neighbour pairs per instruction.
//
// A pair's trilinear footprint is 2 x 2 x 2 cells of the 4 x 4 x 4 filter.  Pairs with the same base plane bz and
// base row by -- one of 9 classes -- only touch the 16 cells (z' in 0..1, y' in 0..1, x in 0..3) of the tile
// (bz + z', by + y', x), and for such pairs the splat is a dense rank-4 update per instruction
//
//     D[m = (z', y', x)][n = channel] += sum_k A[m][k] F[k][n],     A[m][k] = hat(X_k - x) * w_k[z'][y'],
//
// with k = 4 pairs of the class and w = window * (1 -+ fz) * (1 -+ fy).  That is 8 clocks of matrix core per pair and
// 16 channels (half of the products are useful), against 16 for the pair-per-instruction form (cconv_blk.hip) and
// 32 for the unsorted 64-cell form (cconv_mfma.hip); the A operand costs three VALU operations per instruction.
//
// A wave owns two output points of the workgroup's 16-point tile and walks their neighbour lists as ONE stream of
// 64-pair batches (loads run ahead across the point boundary).  Per batch it
//   1. computes the geometry (lane = pair, as in the other kernels) and the pair's class,
//   2. orders the pairs by class with 9 ballots (classes padded to multiples of 4 with zero-weight slots),
//   3. pushes {X, w[4]}, the class of each group of 4 slots and the neighbour index to the pair's slot of the ordered
//      batch in LDS, and loads the feature rows of the ordered slots (16 bytes per lane, half a batch ahead),
//   4. splats: a hand-scheduled block (tools/gen_cls_splat.py -> cconv_cls_splat.inc) reads each group's operands at
//      fixed LDS offsets, one group ahead, and issues ONE matrix instruction whose accumulator registers are addressed
//      relative to M0 = 4 * class: 36 VGPRs (v92 .. v127, see below) hold the whole 64-cell x 16-channel B_i of the point.
// When a point's last batch is done its 9 tiles are merged into the point's row of the B tile [16 points][64 cells x 16
// channels] in LDS (four stores, five 4-float read-modify-writes), which the contraction with the packed filter reads as
// in cconv_blk.hip.
//
// LDS (80 KB per workgroup, two workgroups per CU): B tile 64 KB + 2 KB per wave for {X, w} and the group classes; the
// feature staging (48 slots x 16 channels = half a batch) and the index buffer live in the B row of the wave's SECOND
// point, which is free until that point's tiles are merged.
//
// Accumulation order = the ordered-batch order (stable inside a class), then the fixed merge order: deterministic.
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {

constexpr int kCWaves = 8;
constexpr int kCThreads = 64 * kCWaves;
constexpr int CTM = 2 * kCWaves;  // output points per workgroup = rows of the B tile
constexpr int kCRows = 16;        // M of the contraction MFMA
constexpr int CCH = 16;           // channels per pass
constexpr int kCRow = 1024;       // floats per B row: k' = (z * 4 + y) * 64 + channel * 4 + x
constexpr int kSlots = 96;        // slots of an ordered batch: 64 pairs + at most 3 padding slots per class
constexpr int kHalfSlots = 48;    // feature staging holds half of them
constexpr int kGrp = 20;          // floats per group of 4 slots: w[4][4], X[4]
constexpr int kXst = 512;         // floats per wave outside the B tile: 24 groups
constexpr int kCMaxNT = 4;
constexpr int kNoPair = 9;

// The nine class tiles (9 x 4 registers) live in v92 .. v127, outside the compiler's allocation (amdgpu_num_vgpr on the
// kernel; every asm statement that touches them lists them as clobbered): the splat addresses the accumulator operands of
// its matrix instruction RELATIVE to M0 = 4 * class (s_set_gpr_idx_on, mode src2 | dst -- it applies to v_mfma on gfx950,
// tools/ubench/mfma_gpr_idx.hip), which only works on consecutive registers, and a tuple of 36 has no register class.  The
// computed jump into a table of nine matrix instructions this replaces cost 5 scalar instructions and two taken branches per
// group of four pairs: 9 % of the 3e8-pair layers (measured with the jump removed).
#define CLS_TILE_REGS                                                                                                     \
    "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", \
        "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121",    \
        "v122", "v123", "v124", "v125", "v126", "v127"
// (on gfx90a and later the attribute counts HALF of the unified file -- the compiler doubles it: 46 = v0 .. v91; it also rounds
// up to the allocation granule of 8)
constexpr int kClsCompilerVgprs = 46;

__device__ __forceinline__ void cls_zero_tiles() {
    asm volatile(
        "v_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\tv_mov_b32 v95, 0\n\t"
        "v_mov_b32 v96, 0\n\tv_mov_b32 v97, 0\n\tv_mov_b32 v98, 0\n\tv_mov_b32 v99, 0\n\t"
        "v_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
        "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\t"
        "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
        "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"
        "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\t"
        "v_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\t"
        "v_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\tv_mov_b32 v126, 0\n\tv_mov_b32 v127, 0"
        ::: "memory", CLS_TILE_REGS);
}

struct ClsRec {  // per pair, in the registers of the owner lane
    float x;     // clamped filter coordinate in [0, 3]
    f32x4 w;     // window * wz(z') * wy(y'), index 2 z' + y'
};

__device__ __forceinline__ uint32_t cls_lds_addr(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q;
}

// cross-lane communication through LDS inside a wave: keeps the compiler from reordering around it
__device__ __forceinline__ void xfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SINGLE: the layer is one 16-channel chunk (every layer this kernel serves by default): the contraction's accumulators then
// live only from the chunk's barrier to the epilogue, not across the batch loop, where every register counts
// PLAIN: see cconv_plain() in cconv_common.h
template <int NTT, bool NARROW, bool SYM, bool SINGLE, bool PLAIN>
__global__ __launch_bounds__(kCThreads, 4) __attribute__((amdgpu_num_vgpr(kClsCompilerVgprs))) void cconv_cls_kernel(const CconvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = p.cin, cout = p.cout;
    float* Bt = smem;                                 // [CTM][kCRow], 4-float groups XOR-swizzled by the row
    float* Gs = smem + CTM * kCRow + wave * kXst;     // [24][kGrp]
    unsigned char* Cst = (unsigned char*)(Gs + 24 * kGrp);  // [24]: class of each group
    float* Fst = Bt + (wave + kCWaves) * kCRow;       // [48][16] in the row of this wave's second point
    int* Jst = (int*)(Fst + kHalfSlots * 16);         // [96]
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * CTM;
    const int window = PLAIN ? (int)DMCF_WINDOW_POLY6 : p.window;
    const float* const nval = PLAIN ? nullptr : p.nval;
    const float* const imp = PLAIN ? nullptr : p.inp_imp;
    constexpr bool symmetric = SYM;  // the antisymmetric form (a template parameter: plain layers skip its 12 adds per feature round)

    // splat roles (A / B operands of 16x16x4): tile row m = lane & 15 = (z', y', x), pair k = lane >> 4, channel lane & 15
    const int mk = lane >> 4, mn = lane & 15;
    const float xm = (float)(lane & 3);
    const int widx = (lane >> 2) & 3;
    // feature load roles: lane -> (slot, 4 channels); set per channel chunk
    // contraction roles
    const int mi = lane & 15, mg = lane >> 4;
    const uint32_t gs_lds = cls_lds_addr(Gs), fst_lds = cls_lds_addr(Fst);
    // feature rows through a buffer resource (as in cconv_z3.hip): a slot's byte offset is one 24-bit multiply, formed when
    // its index is published; padding slots and channel blocks past cin are out-of-range offsets, answered with zeros
    constexpr uint32_t kOob = 0xffffffffu;
    const uint32_t rowB = (uint32_t)cin * 4u;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)p.inp_feat, 0, (int)((uint32_t)p.n_inp * rowB), 0x00020000);

    f32x4 acc[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    float out_prev = 0.0f;
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int c0 = chunk * CCH;
        const int nch = min(CCH, cin - c0);
        // Layers of at most 8 input channels stage 8 channels per slot: the features of a WHOLE ordered batch fit the staging,
        // so a batch has one feature round and one splat instead of two.
        constexpr bool narrow = NARROW;  // launched for layers of at most 8 input channels
        const int fr = narrow ? lane >> 1 : lane >> 2, fc4 = narrow ? lane & 1 : lane & 3;
        const int spi = narrow ? 32 : 16;     // slots per load instruction
        const int fstride = narrow ? 8 : 16;  // staged floats per slot
        const bool fch_ok = c0 + 4 * fc4 < cin;
        const int fch = fch_ok ? c0 + 4 * fc4 : 0;
        // The batches of the wave's two points form ONE stream (point A's batches, then point B's): index, position and
        // feature loads run ahead across the point boundary, so a point's first batch does not start with four dependent
        // round trips to memory.
        int64_t rbs[2];
        int nts[2], nbs[2];
        float oxs[2], oys[2], ozs[2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            const int64_t i = pt0 + wave + kCWaves * pp;
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
            nbs[pp] = (nts[pp] + 63) >> 6;
        }
        const int nbA = nbs[0], NB = nbs[0] + nbs[1];
        cls_zero_tiles();

        // ONE buffer over both rows of the wave (they are 8 rows apart in the list): offsets past a row's end are replaced by
        // an out-of-range one and read as index 0 (a valid point, unused)
        const int nt0 = __builtin_amdgcn_readfirstlane(nts[0]), nt1 = __builtin_amdgcn_readfirstlane(nts[1]);
        const int64_t rb0 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rbs[0] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rbs[0]);
        const int64_t rb1 = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rbs[1] >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rbs[1]);
        const int64_t gapB = nt1 > 0 ? rb1 - rb0 : 0;  // entries from row A's start to row B's (rows of a list are stored in order)
        const bool near = gapB >= 0 && gapB + nt1 < ((int64_t)1 << 29) && nt0 < (1 << 29);
        const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.idx + rb0), 0, near ? (int)(max((int64_t)nt0, gapB + nt1) * 4) : 0, 0x00020000);
        const uint32_t offB = (uint32_t)gapB * 4u;
        // batch t of the stream -> is this lane's pair valid
        auto valid = [&](int t) -> bool {
            const bool pp = t >= nbA;
            return t < NB && 64 * (t - (pp ? nbA : 0)) + lane < (pp ? nt1 : nt0);
        };
        auto ld_idx = [&](int t, int& j, float& nv) {
            const bool pp = t >= nbA, ok = valid(t);
            const int o = 64 * (t - (pp ? nbA : 0)) + lane;
            j = 0;
            nv = 0.0f;
            if (near) {
                j = (int)__builtin_amdgcn_raw_buffer_load_b32(rI, ok ? (uint32_t)o * 4u + (pp ? offB : 0u) : kOob, 0, 0);
            } else if (ok) {
                j = p.idx[(pp ? rb1 : rb0) + o];
            }
            if (nval && ok) nv = nval[(pp ? rb1 : rb0) + o];
        };
        auto ld_pos = [&](int j, float& x, float& y, float& z) {  // a scalar base + one 24-bit multiply
            const float* q = (const float*)((const char*)p.inp_pos + (size_t)__umul24((uint32_t)j, 12u));
            x = q[0];
            y = q[1];
            z = q[2];
        };
        auto geom = [&](int t, int j, float nv, float x, float y, float z, int& cls) -> ClsRec {
            ClsRec c;
            const bool pp = t >= nbA;
            x -= pp ? oxs[1] : oxs[0];
            y -= pp ? oys[1] : oys[0];
            z -= pp ? ozs[1] : ozs[0];
            float a = window_value(window, nval ? nv : rel_dist2(x, y, z), p.inv_r2, p.window_fac);
            if (imp) a *= imp[j];
            filter_coords<false>(x, y, z, p);
            c.x = fminf(3.0f, fmaxf(0.0f, x));
            y = fminf(3.0f, fmaxf(0.0f, y));
            z = fminf(3.0f, fmaxf(0.0f, z));
            const float yf = fminf(floorf(y), 2.0f), zf = fminf(floorf(z), 2.0f);
            const float fy = y - yf, fz = z - zf;
            cls = valid(t) ? 3 * (int)zf + (int)yf : kNoPair;
            const float a0 = a * (1.0f - fz), a1 = a * fz;
            c.w = (f32x4){a0 * (1.0f - fy), a0 * fy, a1 * (1.0f - fy), a1 * fy};
            return c;
        };
        // Ordered batch: class c occupies slots [cb[c], cb[c + 1]), a multiple of 4 long; the lanes of a class keep
        // their order.  cb[] is wave uniform.
        struct Order {
            int pos;     // slot of this lane's pair (lanes without a pair: unused)
            int cb[10];
        };
        auto order = [&](int cls) -> Order {
            Order o;
            o.pos = 0;
            int base = 0;
#pragma unroll
            for (int c = 0; c < 9; ++c) {
                const uint64_t m = __ballot(cls == c);
                const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
                o.pos = cls == c ? rank : o.pos;
                o.cb[c] = base;
                base += (__builtin_popcountll(m) + 3) & ~3;
            }
            o.cb[9] = base;
            return o;
        };
        auto push_index = [&](int j, int cls, int pos) {
            Jst[lane] = -1;
            if (lane < kSlots - 64) Jst[64 + lane] = -1;
            xfence();
            if (cls != kNoPair) Jst[pos] = (int)__umul24((uint32_t)j, rowB);  // the byte offset of the slot's feature row
        };
        auto push_rec = [&](const ClsRec& c, int cls, int pos) {
            *(f32x4*)(Gs + 4 * lane) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};  // padding slots: weight 0, X 0
            if (lane < kGrp * 6 - 64) *(f32x4*)(Gs + 4 * (64 + lane)) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            xfence();
            if (cls != kNoPair) {
                float* g = Gs + (pos >> 2) * kGrp + (pos & 3);
                g[16] = c.x;
                *(f32x4*)(g + 3 * (pos & 3)) = c.w;
                if ((pos & 3) == 0) Cst[pos >> 2] = (unsigned char)(4 * cls);  // see splat
            }
        };
        // 16-byte feature loads of half h of the ordered batch (narrow chunks: h = 0, the whole batch): three groups of 16
        // (32) slots, lane = (slot, 4 channels).
        // Padding slots (offset -1), slots past the batch and channels past cin are out of the buffer's range: the loads are
        // unconditional and return zeros there.
        const uint32_t cbyte = fch_ok ? 4u * (uint32_t)fch : kOob;
        auto f_issue = [&](int h, f32x4 (&f)[3]) {
            uint32_t jo[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) jo[k] = (uint32_t)Jst[kHalfSlots * h + spi * k + fr];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                f[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rF, __builtin_elementwise_add_sat(jo[k], cbyte), 0, 0));
        };
        // (the antisymmetric form adds the output point's own features: a padding slot then holds f_i, times weight 0)
        auto f_publish = [&](int t, const f32x4 (&f)[3]) {
            if constexpr (symmetric) {
                f32x4 fi4 = {0.0f, 0.0f, 0.0f, 0.0f};
                if (fch_ok) fi4 = *(const f32x4*)(p.inp_feat + (pt0 + wave + (t >= nbA ? kCWaves : 0)) * cin + fch);
#pragma unroll
                for (int k = 0; k < 3; ++k) *(f32x4*)(Fst + (spi * k + fr) * fstride + 4 * fc4) = f[k] + fi4;
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) *(f32x4*)(Fst + (spi * k + fr) * fstride + 4 * fc4) = f[k];
            }
        };
        // Splat of half h: the (at most 12) groups at fixed staging addresses -- no address arithmetic -- each into the tile
        // of its class.  The class of a group is wave uniform: the owner of a group's first slot left it in Cst (as the
        // register distance of the class's tile from tile 0), three broadcast reads bring the 12 bytes into scalar registers and
        // M0-relative addressing picks the accumulator of the matrix instruction.  Hand scheduled (tools/gen_cls_splat.py).
        auto splat = [&](int h, int nslots) {
            const int lo = kHalfSlots * h;
            const int ng = (min(lo + kHalfSlots, nslots) - lo) >> 2;  // wave uniform
            const uint32_t pw = gs_lds + ((lo >> 2) * kGrp + 4 * mk + widx) * 4;
            const uint32_t px = gs_lds + ((lo >> 2) * kGrp + 16 + mk) * 4;
            const uint32_t pf = fst_lds + (16 * mk + mn) * 4;
            const uint32_t* cw = (const uint32_t*)(Cst + (lo >> 2));
            const uint32_t c0 = __builtin_amdgcn_readfirstlane(cw[0]), c1 = __builtin_amdgcn_readfirstlane(cw[1]),
                           c2 = __builtin_amdgcn_readfirstlane(cw[2]);
            float xa, wa, fa, xb, wb, fb, av;
            uint32_t sc;
            __builtin_amdgcn_s_setprio(3);  // the wave in its splat gets the matrix pipe first (16 -> 16: -2 %)
            asm volatile(
#include "cconv_cls_splat.inc"
                : [xa] "=&v"(xa), [wa] "=&v"(wa), [fa] "=&v"(fa), [xb] "=&v"(xb), [wb] "=&v"(wb), [fb] "=&v"(fb),
                  [a] "=&v"(av), [sc] "=&s"(sc)
                : [px] "v"(px), [pw] "v"(pw), [pf] "v"(pf), [xm] "v"(xm), [ng] "s"(ng), [c0] "s"(c0), [c1] "s"(c1), [c2] "s"(c2)
                : "scc", "m0", "memory", CLS_TILE_REGS);
            __builtin_amdgcn_s_setprio(0);
        };
        // the same for a whole batch staged with 8 channels per slot (columns 8..15 of the tiles repeat 0..7: never read)
        auto splat8 = [&](int nslots) {
            const int ng = nslots >> 2;  // wave uniform, <= 24
            const uint32_t pw = gs_lds + (4 * mk + widx) * 4;
            const uint32_t px = gs_lds + (16 + mk) * 4;
            const uint32_t pf = fst_lds + (8 * mk + (mn & 7)) * 4;
            const uint32_t* cw = (const uint32_t*)Cst;
            uint32_t c[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) c[q] = __builtin_amdgcn_readfirstlane(cw[q]);
            float xa, wa, fa, xb, wb, fb, av;
            uint32_t sc;
            __builtin_amdgcn_s_setprio(3);
            asm volatile(
#include "cconv_cls_splat8.inc"
                : [xa] "=&v"(xa), [wa] "=&v"(wa), [fa] "=&v"(fa), [xb] "=&v"(xb), [wb] "=&v"(wb), [fb] "=&v"(fb),
                  [a] "=&v"(av), [sc] "=&s"(sc)
                : [px] "v"(px), [pw] "v"(pw), [pf] "v"(pf), [xm] "v"(xm), [ng] "s"(ng), [c0] "s"(c[0]), [c1] "s"(c[1]),
                  [c2] "s"(c[2]), [c3] "s"(c[3]), [c4] "s"(c[4]), [c5] "s"(c[5])
                : "scc", "m0", "memory", CLS_TILE_REGS);
            __builtin_amdgcn_s_setprio(0);
        };
        // Merge the 9 tiles into the point's B row and clear them.  D layout of 16x16x4: lane (group G = lane >> 4 =
        // (z', y'), channel lane & 15), register r = x  ->  k' = ((bz + z') * 4 + by + y') * 64 + channel * 4 + x.
        // The four classes with even (bz, by) tile the 16 (z, y) rows exactly: they are plain stores.  The others are added
        // in three rounds of classes with disjoint rows.
        auto merge = [&](int pt) {
            float* Brow = Bt + pt * kCRow;
            const int gz = lane >> 5, gy = (lane >> 4) & 1;
            const int col = (mn ^ (pt & 15)) << 2;
            // tile c goes to rows (c / 3 + gz, c % 3 + gy): byte offset (c / 3) * 1024 + (c % 3) * 256 from this lane's base
            const uint32_t mb = cls_lds_addr(Brow + (gz * 4 + gy) * 64 + col);
            xfence();  // the staging lives in the row of the second point
            asm volatile(
                // the tiles were written by matrix instructions: cover their write -> read distance
                "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                // The four classes with even (bz, by) tile the 16 (z, y) rows exactly: plain stores.  The other five are added to
                // them through the registers of tile 0, free once it is stored (the LDS operations of a wave execute in order:
                // a read finds the stores -- and the sums -- before it; ds_add_f32 instead costs 11 000 clocks per point).
                "ds_write_b128 %[b], v[92:95] offset:0\n\t"       // class 0
                "ds_write_b128 %[b], v[100:103] offset:512\n\t"   // class 2
                "ds_write_b128 %[b], v[116:119] offset:2048\n\t"  // class 6
                "ds_write_b128 %[b], v[124:127] offset:2560\n\t"  // class 8
                "ds_read_b128 v[92:95], %[b] offset:256\n\t"  // class 1
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_add_f32 v92, v92, v96\n\tv_add_f32 v93, v93, v97\n\tv_add_f32 v94, v94, v98\n\tv_add_f32 v95, v95, v99\n\t"
                "ds_write_b128 %[b], v[92:95] offset:256\n\t"
                "ds_read_b128 v[92:95], %[b] offset:2304\n\t"  // class 7
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_add_f32 v92, v92, v120\n\tv_add_f32 v93, v93, v121\n\tv_add_f32 v94, v94, v122\n\tv_add_f32 v95, v95, v123\n\t"
                "ds_write_b128 %[b], v[92:95] offset:2304\n\t"
                "ds_read_b128 v[92:95], %[b] offset:1024\n\t"  // class 3
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_add_f32 v92, v92, v104\n\tv_add_f32 v93, v93, v105\n\tv_add_f32 v94, v94, v106\n\tv_add_f32 v95, v95, v107\n\t"
                "ds_write_b128 %[b], v[92:95] offset:1024\n\t"
                "ds_read_b128 v[92:95], %[b] offset:1536\n\t"  // class 5
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_add_f32 v92, v92, v112\n\tv_add_f32 v93, v93, v113\n\tv_add_f32 v94, v94, v114\n\tv_add_f32 v95, v95, v115\n\t"
                "ds_write_b128 %[b], v[92:95] offset:1536\n\t"
                "ds_read_b128 v[92:95], %[b] offset:1280\n\t"  // class 4
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_add_f32 v92, v92, v108\n\tv_add_f32 v93, v93, v109\n\tv_add_f32 v94, v94, v110\n\tv_add_f32 v95, v95, v111\n\t"
                "ds_write_b128 %[b], v[92:95] offset:1280\n\t"
                "s_waitcnt lgkmcnt(0)"
                :
                : [b] "v"(mb)
                : "memory", CLS_TILE_REGS);
            cls_zero_tiles();
            xfence();
        };

        if (nbA == 0) merge(wave);
        if (NB > 0) {
            // stages: indices two batches ahead, positions (issued once the indices have had half an iteration to arrive)
            // and geometry + order + feature loads one batch ahead
            int jA, jB, cl;
            float nvA, nvB, px, py, pz;
            ld_idx(0, jA, nvA);
            ld_idx(1, jB, nvB);
            ld_pos(jA, px, py, pz);
            const ClsRec first = geom(0, jA, nvA, px, py, pz, cl);
            Order oc = order(cl);
            f32x4 ff[3];
            push_index(jA, cl, oc.pos);
            push_rec(first, cl, oc.pos);
            xfence();
            f_issue(0, ff);
            jA = jB;
            nvA = nvB;
            ld_pos(jA, px, py, pz);
            // (the stream's LAST batch is peeled off both loops: there is no next batch to prepare -- for rows of ~30 pairs,
            // one batch per point, that was a quarter of the vector instructions)
            if constexpr (narrow) {
                for (int t = 0; t + 1 < NB; ++t) {
                    // here: (jA, nvA, px, py, pz) = batch t + 1.  One feature round per batch: the loads of batch t + 1 are
                    // issued before the splat of batch t and published after it.
                    const int nslots = oc.cb[9];
                    f_publish(t, ff);
                    ld_idx(t + 2, jB, nvB);
                    const ClsRec nxt = geom(t + 1, jA, nvA, px, py, pz, cl);
                    const Order on = order(cl);
                    push_index(jA, cl, on.pos);
                    jA = jB;
                    nvA = nvB;
                    ld_pos(jA, px, py, pz);
                    xfence();
                    f_issue(0, ff);
                    splat8(nslots);
                    push_rec(nxt, cl, on.pos);
                    oc = on;
                    if (t == nbA - 1) merge(wave);
                }
                f_publish(NB - 1, ff);
                xfence();
                splat8(oc.cb[9]);
                if (NB - 1 == nbA - 1) merge(wave);
            } else {
                for (int t = 0; t + 1 < NB; ++t) {
                    // here: (jA, nvA, px, py, pz) = batch t + 1.  Every wait on a load below finds that load the youngest one
                    // outstanding (or the younger ones long issued): the counter the hardware offers is in order.
                    const int nslots = oc.cb[9];
                    const bool two = nslots > kHalfSlots;
                    f_publish(t, ff);
                    ld_idx(t + 2, jB, nvB);
                    // ---- half 0 of this batch; the feature loads of half 1 fly meanwhile
                    if (two) f_issue(1, ff);
                    xfence();
                    splat(0, nslots);
                    // geometry + order of the next batch
                    const ClsRec nxt = geom(t + 1, jA, nvA, px, py, pz, cl);
                    const Order on = order(cl);
                    push_index(jA, cl, on.pos);
                    xfence();
                    // ---- half 1; the positions of batch t + 2 and the feature loads of the next batch's half 0 fly meanwhile
                    if (two) f_publish(t, ff);
                    jA = jB;
                    nvA = nvB;
                    ld_pos(jA, px, py, pz);
                    f_issue(0, ff);
                    if (two) {
                        xfence();
                        splat(1, nslots);
                    }
                    push_rec(nxt, cl, on.pos);
                    oc = on;
                    if (t == nbA - 1) merge(wave);
                }
                {
                    const int nslots = oc.cb[9];
                    const bool two = nslots > kHalfSlots;
                    f_publish(NB - 1, ff);
                    if (two) f_issue(1, ff);
                    xfence();
                    splat(0, nslots);
                    if (two) {
                        xfence();  // (half 0's staging reads are done)
                        f_publish(NB - 1, ff);
                        xfence();
                        splat(1, nslots);
                    }
                    if (NB - 1 == nbA - 1) merge(wave);
                }
            }
        }
        if (chunk == p.nchunks - 1) out_prev = epilogue_prefetch(p, pt0, CTM, tid);  // (its round trip: under the merge and the contraction)
        // ---------------- contraction of this channel chunk on the matrix cores ----------------
        // 16-wide k' blocks: blk = (z * 4 + y) * 4 + channel / 4 for t = wave + 8 it; blocks of channels past the chunk's end
        // are skipped.  The filter fragments of this wave's blocks are requested kPre blocks at a time, the first ones BEFORE
        // the second point's merge and the tile's barrier: their round trip runs under those.  The loop this replaces loaded a
        // block's fragments right before its matrix instructions -- one exposed L2 round trip per block, eight per tile
        // (cconv_z3.hip found the same: ~1500 clocks each with every wave of the workgroup in the same phase).
        const int nq = (nch + 3) >> 2;
        constexpr int kIt = 64 / kCWaves;                       // blocks of a chunk per wave
        // (fragments + accumulators must fit the compiler's registers; a multi-chunk layer's accumulators live across the walk)
        constexpr int kPre = SINGLE ? (NTT <= 1 ? 8 : (NTT <= 2 ? 4 : 2)) : (NTT <= 1 ? 4 : (NTT <= 2 ? 2 : 1));
        // (through a buffer resource over the packed filter: the lane's part of a fragment's address is formed once, the block's
        // and the column tile's part is a scalar offset -- cconv_pair.hip)
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
            (void*)p.Wp, 0, (int)((uint32_t)p.nchunks * 64u * (uint32_t)p.NT * 1024u), 0x00020000);
        const uint32_t w_lane = ((uint32_t)mg * (uint32_t)p.NT * 16u + (uint32_t)mi) * 16u;
        auto w_issue = [&](int it0, f32x4 (&bw)[kPre][NTT]) {
#pragma unroll
            for (int q = 0; q < kPre; ++q) {
                const int t = wave + kCWaves * (it0 + q);
                if (t < 16 * nq) {
                    int tq, tr;
                    blk_divmod(t, nq, tq, tr);
                    const int blk = tq * 4 + tr;
                    const uint32_t w_blk = (uint32_t)(chunk * 64 + blk) * (uint32_t)p.NT * 1024u;
                    // (all-zero filter blocks of a block-diagonal pair of layers: neither fetched nor multiplied)
                    const int wq = 4 * chunk + tr;  // channel quad; the mask holds quads 0 .. 7
                    const uint32_t wm = wq < 8 ? p.wmask >> (4 * wq) : 0xfu;
#pragma unroll
                    for (int n = 0; n < NTT; ++n)
                        if (n < p.NT && ((wm >> n) & 1))
                            bw[q][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, w_lane, w_blk + 256u * (uint32_t)n, 0));
                }
            }
        };
        f32x4 bw[kPre][NTT];
        // (written under conditions, inside the chunk loop: without a definition here the compiler carries "the previous
        // chunk's value" -- these registers -- across the whole walk of the list)
#pragma unroll
        for (int q = 0; q < kPre; ++q)
#pragma unroll
            for (int n = 0; n < NTT; ++n) asm volatile("" : "=v"(bw[q][n]));
        w_issue(0, bw);
        merge(wave + kCWaves);
        __syncthreads();
        if constexpr (SINGLE) {
#pragma unroll
            for (int n = 0; n < NTT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int it0 = 0; it0 < kIt; it0 += kPre) {
#pragma unroll
            for (int q = 0; q < kPre; ++q) {
                const int t = wave + kCWaves * (it0 + q);
                if (t < 16 * nq) {
                    int tq, tr;
                    blk_divmod(t, nq, tq, tr);
                    const int blk = tq * 4 + tr;
                    const f32x4 av = *(const f32x4*)(Bt + (size_t)mi * kCRow + ((blk * 16 + mg * 4) ^ (mi << 2)));
                    const int wq = 4 * chunk + tr;
                    const uint32_t wm = wq < 8 ? p.wmask >> (4 * wq) : 0xfu;
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
            if (it0 + kPre < kIt && wave + kCWaves * (it0 + kPre) < 16 * nq) w_issue(it0 + kPre, bw);
        }
        __syncthreads();
    }

    // ---------------- cross-wave reduction + epilogue ----------------
    float* red = Bt;  // [kCWaves][kCRows][16*NT]
    const int ncol = 16 * p.NT;
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        if (n < p.NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * kCRows + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        }
    }
    __syncthreads();
    for (int e = tid; e < CTM * cout; e += kCThreads) {
        const int ptt = e / cout, o = e % cout;
        const int64_t ii = pt0 + ptt;
        if (ii >= p.n_out) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kCWaves; ++w) v += red[((size_t)w * kCRows + ptt) * ncol + o];
        if (p.bias) v += p.bias[o];
        float* dst = p.out + ii * cout + o;
        if (p.flags & DMCF_FLAG_ACCUMULATE) v += e == tid ? out_prev : *dst;
        *dst = v;
    }
}

static constexpr size_t kClsLds = (size_t)(CTM * kCRow + kCWaves * kXst) * sizeof(float);

size_t cconv_cls_packed_floats(int cin, int cout) {
    const int nchunks = (cin + CCH - 1) / CCH, NT = (cout + 15) / 16;
    return (size_t)nchunks * 64 * 4 * NT * 16 * 4 + 16;  // + a block of zeros (see f_issue)
}

// Packs [4][4][4][cin][cout] (optionally mirrored: ASCC, utils/convolutions.py:410-412) into the B-fragment order of
// v_mfma_f32_16x16x4_f32 for the k' order of the B rows above:
//   Wp[chunk][blk][g][n][j][q] = W[z][y][x = q][16 chunk + 4 (blk & 3) + g][16 n + j],   z * 4 + y = blk >> 2
__global__ void pack_filter_cls(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int nchunks, int NT,
                                int symmetric, int sym_axis) {
    const int64_t total = (int64_t)nchunks * 64 * 4 * NT * 16 * 4 + 16;
    const int hd[3] = {(symmetric && sym_axis == 0) ? 2 : 4, (symmetric && sym_axis == 1) ? 2 : 4,
                       (symmetric && sym_axis == 2) ? 2 : 4};
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t s = e;
        const int q = (int)(s & 3); s >>= 2;
        const int j = (int)(s & 15); s >>= 4;
        const int n = (int)(s % NT); s /= NT;
        const int g = (int)(s & 3); s >>= 2;
        const int blk = (int)(s & 63); s >>= 6;
        const int chunk = (int)s;
        const int ci = chunk * CCH + 4 * (blk & 3) + g, o = 16 * n + j;
        float v = 0.0f;
        if (chunk < nchunks && ci < cin && o < cout) {
            int c3[3] = {blk >> 4, (blk >> 2) & 3, q};
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

// Same filters and flags as cconv_blk.hip.  Measured on MI355X against it at 3.07e8 pairs / 265 per output: 6.0 against
// 7.4 ms (16 -> 16), 11.6 against 14.6 ms (24 -> 8); 8 -> 32: 5.8 ms against 6.0 for the LDS splat; 4 -> 32: 5.6 against
// 4.7 (LDS splat); at 29 pairs per output (32 -> 32, one batch per point) 4.2 against 4.1 ms.  Picked for at least 8 input
// channels; the rule must not look at the list (capacity, padding): the same step gives bit-identical results whichever
// neighbour-list representation it runs on.
bool cconv_cls_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "lds" / "mfma" / "blk" / "cls" / "direct": force one implementation
    if (e && e[0] != 'c') return false;
    if (dx != 4 || dy != 4 || dz != 4) return false;
    if (a->coordinate_mapping != DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING || a->interpolation != DMCF_INTERP_LINEAR ||
        !(a->flags & DMCF_FLAG_ALIGN_CORNERS) || (a->flags & DMCF_FLAG_NORMALIZE))
        return false;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin & 3) || cout > 16 * kCMaxNT) return false;
    if ((uintptr_t)a->inp_features & 15) return false;
    // 24-bit multiplies form the byte offsets of feature and position rows; the feature buffer must stay below 2 GB
    if (a->n_inp >= (1 << 24) || a->n_inp * (int64_t)cin * 4 >= ((int64_t)1 << 31)) return false;
    if (e) return true;
    return cin >= 8;
}

int cconv_cls_pack(const dmcf_cconv_args* a, float* packed, hipStream_t stream) {
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    const int nchunks = (cin + CCH - 1) / CCH, NT = (cout + 15) / 16;
    const int64_t total = (int64_t)cconv_cls_packed_floats(cin, cout);
    const unsigned g = (unsigned)((total + 255) / 256);
    if (!(a->flags & DMCF_FLAG_FILTER_PACKED))  // (else the workspace still holds it: dmcf_hip.h)
        hipLaunchKernelGGL(pack_filter_cls, dim3(g < 2048u ? g : 2048u), dim3(256), 0, stream, a->filters, packed, cin, cout, nchunks,
                       NT, (a->flags & DMCF_FLAG_SYMMETRIC) ? 1 : 0, a->sym_axis);
    return nchunks;
}

int cconv_cls_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream) {
    const int NT = (p.cout + 15) / 16;
    float* packed = (float*)workspace;
    const int nchunks = cconv_cls_pack(a, packed, stream);
    p.Wp = packed;
    p.NT = NT;
    p.nchunks = nchunks;
    const int64_t ntiles = (p.n_out + CTM - 1) / CTM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    const bool sym = (p.flags & DMCF_FLAG_SYMMETRIC) != 0;
    const void* fn;
    const bool plain = !sym && cconv_plain(a);  // (ASCC's window is not poly6: no antisymmetric plain instantiations)
#define CLS_PICK(NARROW, SYM, SINGLE, PLAIN)                                                                       \
    (NT <= 1 ? (const void*)cconv_cls_kernel<1, NARROW, SYM, SINGLE, PLAIN>                                         \
             : (NT <= 2 ? (const void*)cconv_cls_kernel<2, NARROW, SYM, SINGLE, PLAIN>                              \
                        : (const void*)cconv_cls_kernel<4, NARROW, SYM, SINGLE, PLAIN>))
#define CLS_PICK2(NARROW, SINGLE) \
    (sym ? CLS_PICK(NARROW, true, SINGLE, false) : (plain ? CLS_PICK(NARROW, false, SINGLE, true) : CLS_PICK(NARROW, false, SINGLE, false)))
    if (p.cin <= 8)
        fn = CLS_PICK2(true, true);
    else if (nchunks == 1)
        fn = CLS_PICK2(false, true);
    else
        fn = CLS_PICK2(false, false);
#undef CLS_PICK2
#undef CLS_PICK
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kClsLds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(kCThreads), kargs, kClsLds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

}  // namespace dmcf
