// CConv for 4x4x4 filters and 17 .. 32 input channels in ONE pass over the neighbour list: PLANE-SORTED splat on
// v_mfma_f32_32x32x2_f32, two neighbour pairs x 32 channels per instruction.
//
// cconv_cls.hip holds the 64-cell x 16-channel B_i of an output point in 36 VGPRs (9 class tiles) and walks the list once
// per 16 channels; its time is mostly per-batch work that does not depend on the channel count (index / position loads,
// geometry, ordering, staging), so a 24-channel layer costs two full walks (11.2 ms for the 3e8-pair 24 -> 8 layer against
// 5.8 ms for 16 -> 16).  Here the pairs are ordered by their base PLANE bz only (3 classes); the pairs of a class touch the
// 32 cells (z' in 0..1, y in 0..3, x in 0..3) of the planes (bz, bz + 1) and the splat is
//
//     D[m = (z', y, x)][n = channel] += sum_k A[m][k] F[k][n],    A[m][k] = w_k[z'] * hat(Y_k - y) * hat(X_k - x),
//
// k = 2 pairs of the class, n = 32 channels: three 32 x 32 tiles = 48 VGPRs hold B_i for all 32 channels.  16 clocks of
// matrix core per pair and 16 channels -- twice the class-sorted form, which spends them on two walks instead.
//
// B_i of a point is 8 KB: a 16-point tile does not fit the LDS next to the staging.  So a wave owns ONE point (16 waves,
// 1024 threads, one workgroup per CU, the same 16 waves per CU as the other kernels), keeps the merged B_i in registers and
// hands it to the contraction in two halves of 16 channels through the 64 KB B tile the other kernels use.
//
// Per 61-pair batch (61 pairs + at most 3 padding slots = 64 slots = 32 instructions): geometry (lane = pair), order by
// plane with 3 ballots, the slot's record (the four hat(X - x) and the eight w[z'] hat(Y - y)), its feature row's offset and
// its group's class byte to the pair's slot, feature rows of half a batch at a time by 16-byte loads (lane = (slot, 4
// channels)) into a 4 KB staging area; the splat of a half is one hand-scheduled block (tools/gen_z3_splat.py): groups of two
// slots at static staging offsets, three ds_read_b32 and one multiply per matrix instruction, the tile picked by M0.
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {
// Diagnostic build (make -C dmcf_amd/csrc trace -> variants/TRACE.so, read by tools/ztrace.py): cycle stamps at the phase
// boundaries of the kernel, summed over every 16th tile.  Compiled out of the product library.
#ifdef ZX_TRACE
__device__ unsigned long long g_ztrace[16];
#define ZT(k) { const uint64_t now_ = __builtin_readcyclecounter(); zt[k] += now_ - zlast; zlast = now_; }
#else
#define ZT(k)
#endif

constexpr int kZWaves = 16;
constexpr int kZThreads = 64 * kZWaves;
constexpr int ZTM = kZWaves;      // output points per workgroup = rows of the B tile
constexpr int kZRow = 1024;       // floats per B row: k' = (z * 4 + y) * 64 + channel * 4 + x (16 channels)
constexpr int kZPairs = 61;       // pairs per batch
constexpr int kZSlots = 64;       // + padding to even class sizes
constexpr int kZRec = 12;         // floats per slot record: hat(X - x) (4), w[z'] * hat(Y - y) (8)
constexpr int kZWaveF = kZSlots * kZRec + kZSlots + 8;  // per wave outside the B tile: records + indices + 32 class bytes
constexpr int kZMaxNT = 4;
constexpr int kZNoPair = 3;


// The three plane-class tiles (3 x 16 registers) live in v80 .. v127, outside the compiler's allocation (amdgpu_num_vgpr on
// the kernel -- on gfx90a and later the attribute counts HALF of the unified file: 40 = v0 .. v79 -- and every asm statement
// that touches them lists them as clobbered): the splat addresses the accumulator operands of its matrix instruction relative
// to M0 = 16 * class (s_set_gpr_idx_on, mode src2 | dst; tools/ubench/mfma_gpr_idx.hip), so a group's class is data, not
// control flow -- no class runs, no single groups between them (tools/gen_z3_splat.py).
#define Z3_TILE_REGS                                                                                                      \
    "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96",   \
        "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", \
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125",   \
        "v126", "v127"
constexpr int kZ3CompilerVgprs = 40;

__device__ __forceinline__ void zfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float zhat(float d) { return __builtin_amdgcn_fmed3f(1.0f - fabsf(d), 0.0f, 1.0f); }

typedef uint32_t u32x4z __attribute__((ext_vector_type(4)));
constexpr uint32_t kZOob = 0xffffffffu;  // a byte offset no buffer holds: the load returns zeros

// 32-bit LDS byte address of a pointer into the dynamic shared array
__device__ __forceinline__ uint32_t zlds(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q;
}

// PLAIN: see cconv_plain() in cconv_common.h
template <int NTT, bool PLAIN>
__global__ __launch_bounds__(kZThreads, 1) __attribute__((amdgpu_num_vgpr(kZ3CompilerVgprs))) void cconv_z3_kernel(const CconvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = p.cin, cout = p.cout;
    const int window = PLAIN ? (int)DMCF_WINDOW_POLY6 : p.window;
    const float* const nval = PLAIN ? nullptr : p.nval;
    const float* const imp = PLAIN ? nullptr : p.inp_imp;
    float* Bt = smem;                                    // [ZTM][kZRow], 4-float groups XOR-swizzled by the row
    float* Fst = Bt + wave * kZRow;                      // [32 slots][32 channels]: this wave's B row, free until the merge
    float* Rec = smem + ZTM * kZRow + wave * kZWaveF;    // [64 slots][kZRec]
    int* Jst = (int*)(Rec + kZSlots * kZRec);            // [64 slots] neighbour index, -1: padding
    unsigned char* Cst = (unsigned char*)(Jst + kZSlots);  // [32 groups] 16 * plane class of the group
    const int tile = (int)(blockIdx.x % 8) * p.tiles_per_xcd + (int)(blockIdx.x / 8);
    if (tile >= p.ntiles) return;
    const int64_t pt0 = (int64_t)tile * ZTM;

    // splat roles (A / B operands of 32x32x2): row m = lane & 31 = (z', y, x), pair k = lane >> 5, channel lane & 31
    const int hk = lane >> 5, jn = lane & 31;
    const int zc = (lane >> 4) & 1;
    // feature load roles: lane -> (slot lane >> 3 of a group of 8, channels 4 (lane & 7) ..)
    const int fr = lane >> 3, fc4 = lane & 7;
    const bool fch_ok = 4 * fc4 < cin;
    // feature rows through a buffer resource: the address of a row is ONE 24-bit multiply (the slot's byte offset is formed
    // when its index is published), and a padding slot / a channel block past cin is an out-of-range offset that the
    // hardware answers with zeros -- no 64-bit address arithmetic, no selects, no block of zeros to point at
    const uint32_t rowB = (uint32_t)cin * 4u;
    const uint32_t cbyte = fch_ok ? 16u * (uint32_t)fc4 : kZOob;
    const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)p.inp_feat, 0, (int)((uint32_t)p.n_inp * rowB), 0x00020000);
    // contraction roles
    const int mi = lane & 15, mg = lane >> 4;

    // this wave's point
    const int64_t i = pt0 + wave;
    int64_t rb = 0;
    int nt = 0;
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (i < p.n_out) {
        rb = p.rs[i];
        int64_t re = p.cnt ? rb + p.cnt[i] : p.rs[i + 1];
        if (re > p.pair_cap) re = rb;
        nt = (int)min(re - rb, (int64_t)0x7fffff00);
        ox = p.out_pos[3 * i];
        oy = p.out_pos[3 * i + 1];
        oz = p.out_pos[3 * i + 2];
    }
    // the point is the wave's: keep its row start, length and position in scalar registers (addresses then are a scalar base
    // plus a 32-bit lane offset instead of 64-bit vector arithmetic)
    rb = ((int64_t)__builtin_amdgcn_readfirstlane((int)(rb >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)rb);
    nt = __builtin_amdgcn_readfirstlane(nt);
    ox = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ox)));
    oy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oy)));
    oz = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, oz)));
    const int NB = (nt + kZPairs - 1) / kZPairs;
    // the row as a buffer of nt entries: entries past its end (and the lanes past a batch's 61 pairs) read as index 0
    const __amdgpu_buffer_rsrc_t rI = __builtin_amdgcn_make_buffer_rsrc((void*)(p.idx + rb), 0, nt * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc((void*)(nval ? nval + rb : p.inp_pos), 0,
                                                                         nval ? nt * 4 : 0, 0x00020000);

    {  // t0 = v[80:95], t1 = v[96:111], t2 = v[112:127]
        asm volatile(
            "v_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\tv_mov_b32 v85, 0\n\t"
            "v_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\tv_mov_b32 v88, 0\n\tv_mov_b32 v89, 0\n\tv_mov_b32 v90, 0\n\tv_mov_b32 v91, 0\n\t"
            "v_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\tv_mov_b32 v95, 0\n\tv_mov_b32 v96, 0\n\tv_mov_b32 v97, 0\n\t"
            "v_mov_b32 v98, 0\n\tv_mov_b32 v99, 0\n\tv_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\t"
            "v_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\t"
            "v_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"
            "v_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\t"
            "v_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\tv_mov_b32 v126, 0\n\tv_mov_b32 v127, 0"
            ::: "memory", Z3_TILE_REGS);
    }

    auto valid = [&](int t) -> bool { return lane < kZPairs && kZPairs * t + lane < nt; };
    auto ld_idx = [&](int t, int& j, float& nv) {
        const uint32_t off = lane < kZPairs ? (uint32_t)(kZPairs * t + lane) * 4u : kZOob;
        j = (int)__builtin_amdgcn_raw_buffer_load_b32(rI, off, 0, 0);
        nv = nval ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rV, off, 0, 0)) : 0.0f;
    };
    // (lanes without a pair hold index 0: a valid row, unused.  A scalar base + 32-bit lane offset; the 96-bit buffer load
    // builtin of this compiler loses two of its three components here)
    auto ld_pos = [&](int j, float& x, float& y, float& z) {
        const float* q = (const float*)((const char*)p.inp_pos + (size_t)__umul24((uint32_t)j, 12u));
        x = q[0];
        y = q[1];
        z = q[2];
    };
    auto geom = [&](int t, int j, float nv, float x, float y, float z, int& cls) -> f32x4 {
        f32x4 c;
        x -= ox;
        y -= oy;
        z -= oz;
        float a = window_value(window, nval ? nv : rel_dist2(x, y, z), p.inv_r2, p.window_fac);
        if (imp) a *= imp[j];
        filter_coords<false>(x, y, z, p);
        c.x = fminf(3.0f, fmaxf(0.0f, x));
        c.y = fminf(3.0f, fmaxf(0.0f, y));
        z = fminf(3.0f, fmaxf(0.0f, z));
        const float zf = fminf(floorf(z), 2.0f), fz = z - zf;
        cls = valid(t) ? (int)zf : kZNoPair;
        c.z = a * (1.0f - fz);
        c.w = a * fz;
        return c;
    };
    // Ordered batch: plane class c occupies slots [cb[c], cb[c + 1]), an even number; the lanes of a class keep their order.
    struct Order {
        int pos;
        int cb[4];
    };
    auto order = [&](int cls) -> Order {
        Order o;
        o.pos = 0;
        int base = 0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const uint64_t m = __ballot(cls == c);
            const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, base));
            o.pos = cls == c ? rank : o.pos;
            o.cb[c] = base;
            base += (__builtin_popcountll(m) + 1) & ~1;
        }
        o.cb[3] = base;
        return o;
    };
    uint32_t* Jof = (uint32_t*)Jst;  // byte offset of the slot's feature row, kZOob: padding
    auto push_index = [&](int j, int cls, int pos) {
        Jof[lane] = kZOob;
        zfence();
        if (cls != kZNoPair) Jof[pos] = __umul24((uint32_t)j, rowB);
    };
    // record of a slot: the four hat(X - x) and the eight products w[z'] * hat(Y - y) -- formed here once per pair (lane =
    // pair), so that the splat's A operand is ONE multiply per matrix instruction
    auto push_rec = [&](const f32x4& c, int cls, int pos) {
        float* r = Rec + kZRec * lane;
        *(f32x4*)(r) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};  // padding slots: weight 0
        *(f32x4*)(r + 4) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        *(f32x4*)(r + 8) = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        zfence();
        if (cls != kZNoPair) {
            const f32x4 hx = {zhat(c.x), zhat(c.x - 1.0f), zhat(c.x - 2.0f), zhat(c.x - 3.0f)};
            const f32x4 hy = {zhat(c.y), zhat(c.y - 1.0f), zhat(c.y - 2.0f), zhat(c.y - 3.0f)};
            r = Rec + kZRec * pos;
            *(f32x4*)(r) = hx;
            *(f32x4*)(r + 4) = c.z * hy;
            *(f32x4*)(r + 8) = c.w * hy;
            if ((pos & 1) == 0) Cst[pos >> 1] = (unsigned char)(16 * cls);  // (a class has an even number of slots)
        }
    };
    // feature rows of half h of the ordered batch: four groups of 8 slots, lane = (slot, 4 channels)
    auto f_issue = [&](int h, f32x4 (&f)[4]) {
        uint32_t jo[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) jo[k] = Jof[32 * h + 8 * k + fr];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            f[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rF, __builtin_elementwise_add_sat(jo[k], cbyte), 0, 0));
    };
    auto f_publish = [&](const f32x4 (&f)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4*)(Fst + (8 * k + fr) * 32 + 4 * fc4) = f[k];
    };
    // groups [g0, g1) of half h, all of one plane class, into that class's tile.  LDS byte addresses of this lane's operands
    // of group 0: hat(X - x) of slot hk, the product w[z'] hat(Y - y) of its row, the feature of channel jn
    const uint32_t a_hx = zlds(Rec + kZRec * hk + (lane & 3));
    const uint32_t a_wy = zlds(Rec + kZRec * hk + 4 + 4 * zc + ((lane >> 2) & 3));
    const uint32_t a_f = zlds(Fst + 32 * hk + jn);
    // Half h of the ordered batch: groups 16 h .. min(16 h + 16, slots / 2) - 1 at fixed staging addresses, their classes as 16
    // bytes in four scalar registers (tools/gen_z3_splat.py).
    auto splat = [&](int h, const Order& o) {
        const int ng = min(16, (o.cb[3] >> 1) - 16 * h);  // wave uniform
        const uint32_t px = a_hx + (uint32_t)(8 * kZRec * 16) * (uint32_t)h;
        const uint32_t pw = a_wy + (uint32_t)(8 * kZRec * 16) * (uint32_t)h;
        const u32x4z cw = *(const u32x4z*)(Cst + 16 * h);
        const uint32_t c0 = __builtin_amdgcn_readfirstlane(cw.x), c1 = __builtin_amdgcn_readfirstlane(cw.y),
                       c2 = __builtin_amdgcn_readfirstlane(cw.z), c3 = __builtin_amdgcn_readfirstlane(cw.w);
        float x0, x1, x2, x3, w0, w1, w2, w3, f0, f1, f2, f3;
        uint32_t s0, s1;
        __builtin_amdgcn_s_setprio(3);  // the wave that reaches its splat first gets the matrix pipe: -3 .. 7 %
        asm volatile(
#include "cconv_z3_splat.inc"
            : [x0] "=&v"(x0), [x1] "=&v"(x1), [x2] "=&v"(x2), [x3] "=&v"(x3), [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2),
              [w3] "=&v"(w3), [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2), [f3] "=&v"(f3), [s0] "=&s"(s0), [s1] "=&s"(s1)
            : [px] "v"(px), [pw] "v"(pw), [pf] "v"(a_f), [ng] "s"(ng), [c0] "s"(c0), [c1] "s"(c1), [c2] "s"(c2), [c3] "s"(c3)
            : "scc", "m0", "memory", Z3_TILE_REGS);
        __builtin_amdgcn_s_setprio(0);
    };

#ifdef ZX_TRACE
    uint64_t zt[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0};
    uint64_t zlast = __builtin_readcyclecounter();
    const uint64_t zstart = zlast;
#endif
    if (NB > 0) {
        int jA, jB, cl;
        float nvA, nvB, px, py, pz;
        ld_idx(0, jA, nvA);
        ld_idx(1, jB, nvB);
        ld_pos(jA, px, py, pz);
        const f32x4 first = geom(0, jA, nvA, px, py, pz, cl);
        Order oc = order(cl);
        f32x4 ff[4];
        push_index(jA, cl, oc.pos);
        push_rec(first, cl, oc.pos);
        zfence();
        f_issue(0, ff);
        jA = jB;
        nvA = nvB;
        ld_pos(jA, px, py, pz);
        ZT(0)
        for (int t = 0; t + 1 < NB; ++t) {
            // here: (jA, nvA, px, py, pz) = batch t + 1, ff = the features of half 0 of batch t
            const bool two = oc.cb[3] > 32;
            f_publish(ff);
            ld_idx(t + 2, jB, nvB);
            if (two) f_issue(1, ff);
            zfence();
            ZT(1)
            splat(0, oc);
            ZT(2)
            // geometry + order of the next batch; its indices replace this batch's (all read by now)
            const f32x4 nxt = geom(t + 1, jA, nvA, px, py, pz, cl);
            const Order on = order(cl);
            zfence();
            push_index(jA, cl, on.pos);
            zfence();
            ZT(3)
            if (two) f_publish(ff);
            jA = jB;
            nvA = nvB;
            ld_pos(jA, px, py, pz);
            f_issue(0, ff);
            ZT(4)
            if (two) {
                zfence();
                splat(1, oc);
            }
            ZT(5)
            zfence();
            push_rec(nxt, cl, on.pos);
            oc = on;
            ZT(6)
        }
        // The row's LAST batch, peeled: there is no next batch to prepare.  Rows of ~30 pairs are this batch only, and the
        // geometry, order and records of a batch that does not exist were a quarter of their vector instructions.
        {
            const bool two = oc.cb[3] > 32;
            f_publish(ff);
            if (two) f_issue(1, ff);
            zfence();
            splat(0, oc);
            if (two) {
                f_publish(ff);
                zfence();
                splat(1, oc);
            }
        }
        ZT(7)
    }

    // B_i of this lane's channel jn, rows y = 2 yb + hk: D layout of 32x32x2 is lane (rows 8 b + 4 (lane >> 5) + r, column
    // lane & 31), register 4 b + r; with m = z' * 16 + y * 4 + x that is z' = b >> 1, y = 2 (b & 1) + hk, x = r.  Planes
    // shared by two classes are added here, in registers: no read-modify-write in LDS.
    // The merged tile stays in the fixed registers, in place: plane 0 = v80 .. v87, plane 1 = v88 .. v95 (+= t1's lower half),
    // plane 2 = v104 .. v111 (+= t2's lower half), plane 3 = v120 .. v127 -- the contraction below has the compiler's 80
    // registers to itself.  (The tiles were written by matrix instructions: the s_nops cover their write -> read distance.)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\tv_add_f32 v88, v88, v96\n\tv_add_f32 v89, v89, v97\n\tv_add_f32 v90, v90, v98\n\tv_add_f32 v91, v91, v99\n\tv_add_f32 v92, v92, v100\n\tv_add_f32 v93, v93, v101\n\tv_add_f32 v94, v94, v102\n\tv_add_f32 v95, v95, v103\n\tv_add_f32 v104, v104, v112\n\tv_add_f32 v105, v105, v113\n\tv_add_f32 v106, v106, v114\n\tv_add_f32 v107, v107, v115\n\tv_add_f32 v108, v108, v116\n\tv_add_f32 v109, v109, v117\n\tv_add_f32 v110, v110, v118\n\tv_add_f32 v111, v111, v119" ::: "memory", Z3_TILE_REGS);

    f32x4 acc[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n) acc[n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float* Brow = Bt + wave * kZRow;
    // The filter fragments of this wave's k' blocks -- blk = (z * 4 + y) * 4 + channel / 4 for t = wave + 16 it, blocks of
    // channels past the chunk's end are skipped -- are requested for a WHOLE chunk at once, chunk 0's (and, with at most 32
    // output channels, chunk 1's) before the B rows are written: their round trip runs under the tile's first barrier.  The
    // loop this replaces loaded a block's fragments right before its matrix instructions: one exposed L2 round trip per
    // block, 6 - 8 per tile, ~1500 clocks each with all 16 waves of the CU in the same phase (tools/ztrace.py).
    constexpr int kIt = 64 / kZWaves;          // blocks of a chunk per wave: t = wave + 16 it < 16 nq
    // (fragments + accumulators must fit the compiler's 80 registers: both chunks at once up to two column tiles, two
    // blocks at a time for four)
    constexpr int kPre = NTT <= 2 ? kIt : 2;   // blocks whose fragments are requested together
    constexpr bool kBoth = NTT <= 2;           // registers for the fragments of both chunks
    auto nq_of = [&](int chunk) { return (min(16, cin - 16 * chunk) + 3) >> 2; };
    // (through a buffer resource over the packed filter: the lane's part of a fragment's address is formed once, the block's and
    // the column tile's part is a scalar offset -- cconv_pair.hip)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.Wp, 0, (int)((uint32_t)p.nchunks * 64u * (uint32_t)p.NT * 1024u), 0x00020000);
    const uint32_t w_lane = ((uint32_t)mg * (uint32_t)p.NT * 16u + (uint32_t)mi) * 16u;
    auto w_issue = [&](int chunk, int it0, f32x4 (&bw)[kPre][NTT]) {
        const int nq = nq_of(chunk);
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
            const int it = it0 + q;
            if (kZWaves * it < 16 * nq) {
                const int t = wave + kZWaves * it;
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
    const float out_prev = epilogue_prefetch(p, pt0, ZTM, tid);
    f32x4 bw[kBoth ? 2 : 1][kPre][NTT];
    w_issue(0, 0, bw[0]);
    if (kBoth && p.nchunks > 1) w_issue(1, 0, bw[kBoth ? 1 : 0]);
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
        if (chunk >= p.nchunks) break;
        if ((jn >> 4) == chunk) {
            const int col = ((jn & 15) ^ (wave & 15)) << 2;
            // row (z * 4 + 2 yb + hk) of the B row: byte offset z * 1024 + yb * 512 from this lane's base
            const uint32_t rb = zlds(Brow + hk * 64 + col);
            asm volatile(
                "ds_write_b128 %0, v[80:83] offset:0\n\tds_write_b128 %0, v[84:87] offset:512\n\t"
                "ds_write_b128 %0, v[88:91] offset:1024\n\tds_write_b128 %0, v[92:95] offset:1536\n\t"
                "ds_write_b128 %0, v[104:107] offset:2048\n\tds_write_b128 %0, v[108:111] offset:2560\n\t"
                "ds_write_b128 %0, v[120:123] offset:3072\n\tds_write_b128 %0, v[124:127] offset:3584"
                :: "v"(rb) : "memory", Z3_TILE_REGS);
        }
        __syncthreads();
        if (chunk == 0) { ZT(13) }
        const int nq = nq_of(chunk);
        f32x4(&bc)[kPre][NTT] = bw[kBoth ? chunk : 0];
#pragma unroll
        for (int it0 = 0; it0 < kIt; it0 += kPre) {
#pragma unroll
            for (int q = 0; q < kPre; ++q) {
                const int it = it0 + q;
                if (kZWaves * it < 16 * nq) {
                    const int t = wave + kZWaves * it;
                    int tq, tr;
                blk_divmod(t, nq, tq, tr);
                const int blk = tq * 4 + tr;
                    const f32x4 av = *(const f32x4*)(Bt + (size_t)mi * kZRow + ((blk * 16 + mg * 4) ^ (mi << 2)));
                    const uint32_t wm = p.wmask >> (4 * (4 * chunk + tr));
#pragma unroll
                    for (int n = 0; n < NTT; ++n) {
                        if (n < p.NT && ((wm >> n) & 1)) {
                            const f32x4 bv = bc[q][n];
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv.z, acc[n], 0, 0, 0);
                            acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv.w, acc[n], 0, 0, 0);
                        }
                    }
                }
            }
            if (it0 + kPre < kIt && kZWaves * (it0 + kPre) < 16 * nq) w_issue(chunk, it0 + kPre, bw[kBoth ? chunk : 0]);
        }
        if (!kBoth && chunk + 1 < p.nchunks) w_issue(chunk + 1, 0, bw[0]);
        __syncthreads();
    }

    ZT(14)
    // ---------------- cross-wave reduction + epilogue ----------------
    float* red = Bt;  // [kZWaves][16][16*NT]
    const int ncol = 16 * p.NT;
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        if (n < p.NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((size_t)wave * 16 + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
        }
    }
    __syncthreads();
    ZT(15)
    for (int e = tid; e < ZTM * cout; e += kZThreads) {
        const int ptt = e / cout, o = e % cout;
        const int64_t ii = pt0 + ptt;
        if (ii >= p.n_out) continue;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kZWaves; ++w) v += red[((size_t)w * 16 + ptt) * ncol + o];
        if (p.bias) v += p.bias[o];
        float* dst = p.out + ii * cout + o;
        if (p.flags & DMCF_FLAG_ACCUMULATE) v += e == tid ? out_prev : *dst;
        *dst = v;
    }
#ifdef ZX_TRACE
    ZT(9)
    if (lane == 0 && (tile & 15) == 0) {
        for (int k = 0; k < 16; ++k) if (k < 10 || k > 12) atomicAdd(&g_ztrace[k], zt[k]);
        atomicAdd(&g_ztrace[10], zlast - zstart);
        atomicAdd(&g_ztrace[11], 1ull);
        atomicAdd(&g_ztrace[12], (unsigned long long)NB);
    }
#endif
}
#ifdef ZX_TRACE
}
extern "C" int dmcf_ztrace(unsigned long long* out) {
    unsigned long long z[16] = {0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dmcf::g_ztrace), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dmcf::g_ztrace), z, sizeof(z));
    return 0;
}
namespace dmcf {
#endif

static constexpr size_t kZ3Lds = (size_t)(ZTM * kZRow + kZWaves * kZWaveF) * sizeof(float);

// Same filters and flags as cconv_cls.hip, without the antisymmetric form; 17 .. 32 input channels.
bool cconv_z3_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "z3": force, anything else: never
    if (e && e[0] != 'z') return false;
    if (dx != 4 || dy != 4 || dz != 4) return false;
    if (a->flags & DMCF_FLAG_SYMMETRIC) return false;
    if (a->coordinate_mapping != DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING || a->interpolation != DMCF_INTERP_LINEAR ||
        !(a->flags & DMCF_FLAG_ALIGN_CORNERS) || (a->flags & DMCF_FLAG_NORMALIZE))
        return false;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin & 3) || cin > 32 || cout > 16 * kZMaxNT) return false;
    if ((uintptr_t)a->inp_features & 15) return false;
    // 24-bit multiplies form the byte offsets of feature and position rows; the buffers must stay below 2 GB
    if (a->n_inp >= (1 << 24) || a->n_inp * (int64_t)cin * 4 >= ((int64_t)1 << 31)) return false;
    if (e) return true;
    return cin > 16;
}

int cconv_z3_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream) {
    const int NT = (p.cout + 15) / 16;
    float* packed = (float*)workspace;
    const int nchunks = cconv_cls_pack(a, packed, stream);  // the B-fragment order of cconv_cls.hip, 16 channels per chunk
    p.Wp = packed;
    p.NT = NT;
    p.nchunks = nchunks;
    const int64_t ntiles = (p.n_out + ZTM - 1) / ZTM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    const unsigned grid = (unsigned)p.tiles_per_xcd * 8u;
    const void* fn;
    if (cconv_plain(a))
        fn = NT <= 1 ? (const void*)cconv_z3_kernel<1, true>
                     : (NT <= 2 ? (const void*)cconv_z3_kernel<2, true> : (const void*)cconv_z3_kernel<4, true>);
    else
        fn = NT <= 1 ? (const void*)cconv_z3_kernel<1, false>
                     : (NT <= 2 ? (const void*)cconv_z3_kernel<2, false> : (const void*)cconv_z3_kernel<4, false>);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kZ3Lds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(kZThreads), kargs, kZ3Lds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

}  // namespace dmcf
