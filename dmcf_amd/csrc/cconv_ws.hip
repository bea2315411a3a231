// CConv for 4x4x4 filters and up to 32 input channels with WAVE SPECIALISATION: the splat of cconv_pair.hip (splat F, one
// neighbour pair per v_mfma_f32_4x4x1_16B_f32) in PRODUCER waves, the contraction with the packed filter in CONSUMER waves of
// the same persistent workgroup -- splat H.
//
// Why: every other splat kernel here runs ONE set of waves through both phases of a 16-point tile -- row bounds -> indices ->
// positions -> features (four dependent round trips), splat, barrier, contraction, sum over waves, store -- with one workgroup
// per CU, so a CU does one thing at a time: of the ~30k clocks of a tile of the 32 -> 32 layer the matrix pipe needs 12.5k
// (DESIGN section 8 item -2); the start of a 1024-thread workgroup alone is ~4k (tools/ubench/contract.hip).  Here a workgroup
// is persistent (one per CU, tiles by stride inside its XCD's range) and its waves keep ONE role each:
//
//   waves 0 .. 3  producers (one per SIMD): each owns FOUR consecutive output points of the tile as one dense stream of 64-pair
//                 batches (cconv_pair.hip's two-row stream, generalised: a point's pairs start at a block of 8, the batch that
//                 holds a point boundary runs the one splat site once per segment).  A finished point is merged in registers and
//                 stored to its row of the B tile, all 32 channels at once.
//   waves 4 .. 7  consumers (one per SIMD): when the tile is full each PULLS its quarter of the k' blocks of all 16 rows into
//                 registers (v128 .. v255: 32 blocks x 4), releases the tile, and runs its 4 NT matrix instructions per block
//                 from there with filter fragments streamed from L2 (a ring of 8 blocks ahead, which runs on across tiles: a
//                 consumer's blocks are the same for every tile); the four partial sums meet in LDS and are stored (bias,
//                 DMCF_FLAG_ACCUMULATE) one tile later, behind the next barrier.
//
// Two workgroup barriers per tile: "full" (producers have stored the 16 rows) and "free" (consumers hold them in registers).
// Between them the producers are in the first round trips of the next tile, which need no LDS of the tile; the feature staging
// of a producer is the row of its LAST point (dead once that point's splat is done), so the B tile is 128 KB and the workgroup
// 154 KB of LDS.  Every wave has 256 registers (two waves per SIMD): the compiler v0 .. v115, the producers' operand buffers
// and class tiles / the consumers' A fragments above that, outside its allocation (tests/test_fixed_registers.py).
//
// Accumulation order = list order inside a class, the fixed merge order, then blocks in order per consumer and consumers in
// order: deterministic, and independent of the grid.
#include <stdlib.h>

#include "cconv_common.h"

namespace dmcf {

// Diagnostic build (make -C dmcf_amd/csrc ws_trace -> variants/WTRACE.so, read by tools/wtrace.py): cycle stamps at the phase
// boundaries of both roles, summed over the first producer / consumer of every 16th workgroup.  Compiled out of the product.
#ifdef WS_TRACE
__device__ unsigned long long g_wtrace[32];
#define WT(k) { const uint64_t now_ = __builtin_readcyclecounter(); wt[k] += now_ - wlast; wlast = now_; }
#else
#define WT(k)
#endif

constexpr int kWProd = 4, kWCons = 4;
constexpr int kWThreads = 64 * (kWProd + kWCons);
constexpr int WTM = 16;             // output points per tile = rows of the B tile
constexpr int kWPts = WTM / kWProd; // points per producer
constexpr int kWRow = 2048;         // floats per B row: [2 chunks of 16 channels][k' = (z * 4 + y) * 64 + channel * 4 + x]
constexpr int kWRecG = 36;          // floats per record group: 8 products x 4 pairs, padded (bank = 4 g + 4 q + t)
constexpr int kWRec = 16 * kWRecG;
constexpr int kWWaveF = kWRec + 64 + 384; // per producer: records + index buffer + the ring's row tables (64 x (2 + 4))
constexpr int kWMaxNT = 2;           // (the LDS: 128 KB of B tile + 18 KB of producer staging + 4 x 2 KB of sums per column tile)
constexpr int kWRed = 2 * kWCons * WTM * 16 * kWMaxNT;  // partial sums of the consumers, two tiles' worth (see the consumers' loop)
constexpr int kWCompilerVgprs = 56; // (the attribute counts HALF of the unified file: v0 .. v111)

#define WS_FIXED_REGS                                                                                                      \
    "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128",      \
        "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142",      \
        "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156",      \
        "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170",      \
        "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184",      \
        "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198",      \
        "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212",      \
        "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226",      \
        "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240",      \
        "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// workgroup barrier for LDS hand-overs: every LDS access of this wave has completed, nothing moves across it
#define WS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

__device__ __forceinline__ void wfence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t wlds(const void* q) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)q;
}

__device__ __forceinline__ int wsgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float wsgprf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

struct WsRec {     // per pair, in the registers of its owner lane
    f32x4 lo, hi;  // a w_z[z'] w_y[y'] w_x[x'], index 2 y' + x', for z' = 0 / 1
    int cls4;      // 4 * ((bz * 3 + by) * 3 + bx)
};

constexpr uint32_t kWOob = 0xffffffffu;  // a byte offset no buffer holds: the load returns zeros

typedef const __attribute__((address_space(4))) CconvParams* WsKP;

// filter_coords<false> for a 4 x 4 x 4 filter (ball -> cube, align_corners): needs one scalar of the parameter block
__device__ __forceinline__ void ws_filter_coords(float& x, float& y, float& z, float inv_extent) {
    const float s = 2.0f * inv_extent;
    x *= s; y *= s; z *= s;
    sphere_to_cyl(x, y, z);
    cyl_to_cube(x, y);
    x *= 0.5f; y *= 0.5f; z *= 0.5f;
    x = (x + 0.5f) * 3.0f;
    y = (y + 0.5f) * 3.0f;
    z = (z + 0.5f) * 3.0f;
}

// One k' block of the consumer: A fragments of block IT sit in v[128 + 4 IT .. + 3]
#define WS_PULL(IT, ADDR) \
    asm volatile("ds_read_b128 v[128+4*(" #IT "):128+4*(" #IT ")+3], %0" ::"v"(ADDR) : "memory", WS_FIXED_REGS)
// ... multiplied into column tile N, whose accumulator is v[112 + 4 N .. + 3] -- FIXED registers as well: between two asm
// statements the compiler may copy a value it owns, and a vector read of a matrix result that is younger than 11 wait states
// returns the old contents of its last register (rows 4 g + 3 of the tile) -- the compiler inserts those wait states only
// behind matrix instructions it emitted itself
#define WS_MFMA(IT, N, BV)                                                                                                  \
    asm volatile("v_mfma_f32_16x16x4_f32 v[112+4*" #N ":112+4*" #N "+3], v[128+4*(" #IT ")+0], %0, v[112+4*" #N ":112+4*" #N "+3]\n\t" \
                 "v_mfma_f32_16x16x4_f32 v[112+4*" #N ":112+4*" #N "+3], v[128+4*(" #IT ")+1], %1, v[112+4*" #N ":112+4*" #N "+3]\n\t" \
                 "v_mfma_f32_16x16x4_f32 v[112+4*" #N ":112+4*" #N "+3], v[128+4*(" #IT ")+2], %2, v[112+4*" #N ":112+4*" #N "+3]\n\t" \
                 "v_mfma_f32_16x16x4_f32 v[112+4*" #N ":112+4*" #N "+3], v[128+4*(" #IT ")+3], %3, v[112+4*" #N ":112+4*" #N "+3]"       \
                 :                                                                                                          \
                 : "v"((BV).x), "v"((BV).y), "v"((BV).z), "v"((BV).w)                                                      \
                 : WS_FIXED_REGS)
// two column tiles, interleaved: a matrix instruction whose accumulator the previous one wrote issues a few clocks later
#define WS_MFMA2(IT, BV0, BV1)                                                                                              \
    asm volatile("v_mfma_f32_16x16x4_f32 v[112:115], v[128+4*(" #IT ")+0], %0, v[112:115]\n\t"                               \
                 "v_mfma_f32_16x16x4_f32 v[116:119], v[128+4*(" #IT ")+0], %4, v[116:119]\n\t"                               \
                 "v_mfma_f32_16x16x4_f32 v[112:115], v[128+4*(" #IT ")+1], %1, v[112:115]\n\t"                               \
                 "v_mfma_f32_16x16x4_f32 v[116:119], v[128+4*(" #IT ")+1], %5, v[116:119]\n\t"                               \
                 "v_mfma_f32_16x16x4_f32 v[112:115], v[128+4*(" #IT ")+2], %2, v[112:115]\n\t"                               \
                 "v_mfma_f32_16x16x4_f32 v[116:119], v[128+4*(" #IT ")+2], %6, v[116:119]\n\t"                               \
                 "v_mfma_f32_16x16x4_f32 v[112:115], v[128+4*(" #IT ")+3], %3, v[112:115]\n\t"                               \
                 "v_mfma_f32_16x16x4_f32 v[116:119], v[128+4*(" #IT ")+3], %7, v[116:119]"                                    \
                 :                                                                                                          \
                 : "v"((BV0).x), "v"((BV0).y), "v"((BV0).z), "v"((BV0).w), "v"((BV1).x), "v"((BV1).y), "v"((BV1).z), "v"((BV1).w) \
                 : WS_FIXED_REGS)
#define WS_ACC_READ(N, DST)                                                                                                 \
    asm volatile("v_mov_b32 %0, v[112+4*" #N "+0]\n\tv_mov_b32 %1, v[112+4*" #N "+1]\n\tv_mov_b32 %2, v[112+4*" #N "+2]\n\t"     \
                 "v_mov_b32 %3, v[112+4*" #N "+3]"                                                                          \
                 : "=v"((DST).x), "=v"((DST).y), "=v"((DST).z), "=v"((DST).w)                                             \
                 :                                                                                                          \
                 : WS_FIXED_REGS)


// ---- the producers' per-pair stages (plain functions: a lambda's closure object -- a dozen captured values -- is stored to
// scratch and never read when its body is inlined this late)

// the ring of rows of a producer: lane r % 64 holds row r's first stream position and its end (start + pairs) and, for the
// first row of a tile, the tile's base in the index array and the entries the four rows span from there
struct WsRing {
    int start, end, rblo, rbhi, span;
};

// index (and explicit neighbour value) of the pair at stream position sp of the tile whose rows are r4 .. r4 + 3 and start at
// (the tile's first batch), o1, o2, o3; positions without a pair read entry 0 of the buffer.  kk = the ring entry of the row.
__device__ __forceinline__ void ws_ld_idx(int sp, int r4, int o1, int o2, int o3, const float* Tab, __amdgpu_buffer_rsrc_t rI, const float* nval,
                                          int64_t rb0, int& j, float& nv, int& kk) {
    kk = (r4 + (sp >= o1 ? 1 : 0) + (sp >= o2 ? 1 : 0) + (sp >= o3 ? 1 : 0)) & 63;
    const f32x2 ed = *(const f32x2*)(Tab + 2 * kk);
    // (__builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler: by value through __float_as_int)
    const int e = __float_as_int(ed.x), d = __float_as_int(ed.y);
    const bool ok = sp < e;
    const uint32_t o = (uint32_t)sp * 4u + (uint32_t)d;
    j = (int)__builtin_amdgcn_raw_buffer_load_b32(rI, ok ? o : kWOob, 0, 0);
    nv = 0.0f;
    if (nval && ok) nv = nval[rb0 + (o >> 2)];
}

__device__ __forceinline__ void ws_ld_pos(const float* inp_pos, int j, float& x, float& y, float& z) {  // a scalar base + one 24-bit multiply
    const float* q = (const float*)((const char*)inp_pos + (size_t)__umul24((uint32_t)j, 12u));
    x = q[0];
    y = q[1];
    z = q[2];
}

// the pair's 8 trilinear products and its class
__device__ __forceinline__ WsRec ws_geom(int sp, int kk, const float* Tab, int window, const float* nval, const float* imp,
                                         float inv_r2, float window_fac, float inv_extent, int j, float nv, float x, float y, float z, bool& ok) {
    WsRec c;
    const f32x4 o = *(const f32x4*)(Tab + 128 + 4 * kk);
    ok = sp < __float_as_int(o.w);
    x -= o.x;
    y -= o.y;
    z -= o.z;
    float a = window_value(window, nval ? nv : rel_dist2(x, y, z), inv_r2, window_fac);
    if (imp) a *= imp[j];
    a = ok ? a : 0.0f;  // lanes without a pair: weight zero in class 0, features out of range
    ws_filter_coords(x, y, z, inv_extent);
    x = fminf(3.0f, fmaxf(0.0f, x));
    y = fminf(3.0f, fmaxf(0.0f, y));
    z = fminf(3.0f, fmaxf(0.0f, z));
    const float xf = fminf(floorf(x), 2.0f), yf = fminf(floorf(y), 2.0f), zf = fminf(floorf(z), 2.0f);
    const float fx = x - xf, fy = y - yf, fz = z - zf;
    c.cls4 = ok ? 4 * (((int)zf * 3 + (int)yf) * 3 + (int)xf) : 0;
    const float a0 = a * (1.0f - fz), a1 = a * fz;
    const float y00 = (1.0f - fy) * (1.0f - fx), y01 = (1.0f - fy) * fx, y10 = fy * (1.0f - fx), y11 = fy * fx;
    c.lo = (f32x4){a0 * y00, a0 * y01, a0 * y10, a0 * y11};
    c.hi = (f32x4){a1 * y00, a1 * y01, a1 * y10, a1 * y11};
    return c;
}

// Rows 32 wi .. 32 wi + 31 of a producer (8 tiles) -> the ring: lanes / table entries 32 (wi & 1) .. + 31.  Synchronous (one
// round trip to the row bounds and the output positions), once per 8 tiles.
__device__ __forceinline__ void ws_refill(WsKP kp0, int wi, int lane, int wave, int t_begin, int t_end, int nslots, float* Tab, int& stream_end,
                                          WsRing& ring) {
    WsKP kp = kp0;
    asm volatile("" : "+s"(kp));
    const int lb = 32 * (wi & 1);
    const bool act = (unsigned)(lane - lb) < 32u;
    const int r = 32 * wi + (lane - lb);
    const int tile = t_begin + (r >> 2) * nslots;
    const int64_t i = (int64_t)tile * WTM + kWPts * wave + (r & 3);
    int64_t rb = 0;
    int nt = 0;
    float ox = 0.0f, oy = 0.0f, oz = 0.0f;
    if (act && tile < t_end && i < kp->n_out) {
        const int64_t* const rs = kp->rs;
        const int32_t* const cnt = kp->cnt;
        const float* const out_pos = kp->out_pos;
        const int64_t b = rs[i];
        int64_t e = cnt ? b + cnt[i] : rs[i + 1];
        if (e > kp->pair_cap) e = b;
        rb = b;
        nt = (int)min(e - b, (int64_t)0x01ffffc0);
        ox = out_pos[3 * i];
        oy = out_pos[3 * i + 1];
        oz = out_pos[3 * i + 2];
    }
    // stream positions: a row owns its pairs rounded up to blocks of 8 (at least one), a tile whole batches
    const int slots = (max(nt, 1) + 7) & ~7;
    const int q = lane & 3;
    const int s1 = __shfl_up(slots, 1), s2 = __shfl_up(slots, 2), s3 = __shfl_up(slots, 3);
    const int intra = (q >= 1 ? s1 : 0) + (q >= 2 ? s2 : 0) + (q >= 3 ? s3 : 0);
    const int ttot = (__shfl(intra + slots, lane | 3) + 63) & ~63;
    int tp = 0, wtot = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        const int tm = __builtin_amdgcn_readlane(ttot, lb + 4 * m);
        if (((lane - lb) >> 2) > m) tp += tm;
        wtot += tm;
    }
    const int start = stream_end + tp + intra;
    // ONE buffer per tile over its four rows (consecutive rows of the list): entry e of a row sits gap + e entries behind the
    // start of the tile's first row
    const int64_t rb0 = ((int64_t)__shfl((int)(rb >> 32), lane & ~3) << 32) | (uint32_t)__shfl((int)rb, lane & ~3);
    const int64_t gap = nt > 0 ? rb - rb0 : 0;
    // (eligibility bounds n_inp -- and with it every row -- by 2^24 entries: four consecutive rows always fit one buffer)
    if (__builtin_amdgcn_ballot_w64(act && (gap < 0 || gap + nt >= ((int64_t)1 << 29))) != 0) __builtin_trap();
    int sp = (int)gap + nt;
    sp = max(sp, __shfl_xor(sp, 1));
    sp = max(sp, __shfl_xor(sp, 2));
    if (act) {
        const int e = start + nt;
        const int d = (int)(((uint32_t)(int)gap - (uint32_t)start) * 4u);
        *(f32x2*)(Tab + 2 * lane) = (f32x2){__int_as_float(e), __int_as_float(d)};
        *(f32x4*)(Tab + 128 + 4 * lane) = (f32x4){ox, oy, oz, __int_as_float(e)};
        ring.start = start;
        ring.end = e;
        ring.rblo = (int)rb0;
        ring.rbhi = (int)(rb0 >> 32);
        ring.span = sp;
    }
    stream_end += wtot;
    wfence();
}

// PLAIN: see cconv_plain() in cconv_common.h
template <int NTT, bool PLAIN>
__global__ __launch_bounds__(kWThreads, 1) __attribute__((amdgpu_num_vgpr(kWCompilerVgprs))) void cconv_ws_kernel(const CconvParams p_arg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // The parameter block is read where it is used, through the kernarg pointer made opaque once per tile: a persistent loop
    // otherwise keeps all ~50 dwords of it (and everything derived from them that does not change from tile to tile) live in
    // scalar registers across both roles' loops -- hundreds of spills (DESIGN section 8 item -1 has the earlier attempts).
    const WsKP kp0 = (WsKP)__builtin_amdgcn_kernarg_segment_ptr();
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wsgpr(tid >> 6);
    const int cin = kp0->cin, cout = kp0->cout;
    float* Bt = smem;  // [WTM][kWRow], 4-float groups XOR-swizzled by the row inside each chunk
    // the tiles of this workgroup: its XCD's range (consecutive tiles share an L2), by stride of the XCD's workgroups
    const int xcd = (int)(blockIdx.x & 7), slot = (int)(blockIdx.x >> 3), nslots = (int)(gridDim.x >> 3);
    const int t_begin = xcd * kp0->tiles_per_xcd + slot;
    const int t_end = min((xcd + 1) * kp0->tiles_per_xcd, kp0->ntiles);
    if (t_begin >= t_end) return;

    if (wave < kWProd) {
        // =============================================== PRODUCER ===============================================
        float* Rec = smem + WTM * kWRow + wave * kWWaveF;  // [16 groups][kWRecG]: product q of pair 4 g + t at g * kWRecG + 4 q + t
        uint32_t* Jof = (uint32_t*)(Rec + kWRec);          // [64]: byte offset of the pair's feature row (kWOob: no pair)
        float* Tab = Rec + kWRec + 64;                     // ring of 64 rows: [64][2] {end, byte delta}, then [64][4] {x, y, z, end}
        float* Fst = Bt + (kWPts * wave + kWPts - 1) * kWRow;  // [16 groups][32 channels (permuted)][4 pairs]: the LAST point's row
        // splat roles: this lane's channel (B operand, accumulator column) and plane offset z'
        const int ch = lane & 31, half = lane >> 5;
        // feature load roles: lane -> (pair fr of a round of 8, channels 4 fq .. 4 fq + 3)
        const int fr = lane >> 3, fq = lane & 7;
        const uint32_t rowBy = (uint32_t)cin * 4u;
        const uint32_t cbyte = 4 * fq < cin ? 16u * (uint32_t)fq : kWOob;
        const int pg = fr >> 2, ptq = fr & 3;
        float* const w01 = Fst + pg * 128 + 4 * fq + ptq;              // channels 4 fq, 4 fq + 1: + 0, + 32
        float* const w23 = Fst + pg * 128 + 4 * ((fq + 4) & 7) + ptq;  // channels 4 fq + 2, 4 fq + 3: + 64, + 96
        const uint32_t a_rec0 = wlds(Rec + 4 * (4 * half + (lane & 3)));
        const uint32_t a_fst0 = wlds(Fst + 4 * (8 * (ch & 3) + (((ch >> 2) + 4 * ((ch >> 1) & 1)) & 7)));
        // this lane's part of row 0 of the B tile (its channel's chunk and column, its half-wave's planes), without the swizzle
        const uint32_t row_lane = wlds(Bt + (ch >> 4) * 1024 + half * 512);

        asm volatile(
            ".irp r,148,150,152,154,156,158,160,162,164,166,168,170,172,174,176,178,180,182,184,186,188,190,192,194,196,198,200,202,"
            "204,206,208,210,212,214,216,218,220,222,224,226,228,230,232,234,236,238,240,242,244,246,248,250,252,254\n\t"
            "v_mov_b64 v[\\r:\\r+1], 0\n\t.endr" ::: "memory", WS_FIXED_REGS);

        // ---- ONE continuous stream of 64-pair batches over ALL tiles of this producer.  Row r (tile r / 4 of the workgroup's
        // tiles, the producer's point r % 4 of it) owns the stream positions [start_r, start_r + nt_r); start_r is a multiple of
        // 8 (the splat runs in blocks of 8 pairs and a block feeds ONE point's tiles), a tile's first row starts a batch (the
        // feature staging of a batch is the B row of the tile's last point), and an empty row still owns one block, so every
        // tile has a batch.  The loads run ahead ACROSS tiles (indices three batches, positions two, features one): with one
        // batch loop per tile the four dependent round trips of a tile's start (row bounds -> indices -> positions -> features,
        // ~8k clocks) were exposed in every tile -- a producer is alone on its SIMD but for the consumer's matrix instructions.
        // What a stage needs of a row lives in a ring of 64 rows: lane r % 64 of the registers below and entry r % 64 of the
        // table in LDS; 32 rows (8 tiles) are refilled at a time, 16 or more rows ahead of the splat.
        const int T = (t_end - t_begin + nslots - 1) / nslots;  // tiles of this workgroup
        WsRing ring = {0, 0, 0, 0, 0};
        int stream_end = 0;
        ws_refill(kp0, 0, lane, wave, t_begin, t_end, nslots, Tab, stream_end, ring);
        ws_refill(kp0, 1, lane, wave, t_begin, t_end, nslots, Tab, stream_end, ring);

        WsKP kp = kp0;
        asm volatile("" : "+s"(kp));
        const int window = PLAIN ? (int)DMCF_WINDOW_POLY6 : kp->window;
        const float* const nval = PLAIN ? nullptr : kp->nval;
        const float* const imp = PLAIN ? nullptr : kp->inp_imp;
        const float inv_r2 = kp->inv_r2, window_fac = kp->window_fac, inv_extent = kp->inv_extent;
        const float* const inp_pos = kp->inp_pos;
        const int32_t* const idx = kp->idx;
        const __amdgpu_buffer_rsrc_t rF = __builtin_amdgcn_make_buffer_rsrc((void*)kp->inp_feat, 0, (int)((uint32_t)kp->n_inp * rowBy), 0x00020000);

        auto push_index = [=](int j, bool ok) { Jof[lane] = ok ? __umul24((uint32_t)j, rowBy) : kWOob; };
        auto push_rec = [=](const WsRec& c) {
            float* r = Rec + (lane >> 2) * kWRecG + (lane & 3);
            r[0] = c.lo.x; r[4] = c.lo.y; r[8] = c.lo.z; r[12] = c.lo.w;
            r[16] = c.hi.x; r[20] = c.hi.y; r[24] = c.hi.z; r[28] = c.hi.w;
        };
        int pk_cur = 0;  // the packed class bytes of the staged batch (lanes 0, 4, 8, ...: four pairs each)
        auto pack_classes = [=](int cls4, uint32_t (&c)[16]) -> int {
            int pk = cls4 | (__builtin_amdgcn_mov_dpp(cls4, 0xb1, 0xf, 0xf, true) << 8);   // quad_perm [1, 0, 3, 2]
            pk = pk | (__builtin_amdgcn_mov_dpp(pk, 0x4e, 0xf, 0xf, true) << 16);          // quad_perm [2, 3, 0, 1]
#pragma unroll
            for (int m = 0; m < 16; ++m) c[m] = (uint32_t)__builtin_amdgcn_readlane(pk, 4 * m);
            return pk;
        };
        auto f_issue = [=](int np, f32x4 (&f)[8]) {
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
        auto f_publish = [=](int np, const f32x4 (&f)[8]) {
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
        auto splat = [=](int nblk, const uint32_t (&c)[16], uint32_t a_rec, uint32_t a_fst) {
            uint32_t s0;
            __builtin_amdgcn_s_setprio(3);
            asm volatile(
#include "cconv_pair_splat.inc"
                : [s0] "=&s"(s0)
                : [pa] "v"(a_rec), [pf] "v"(a_fst), [nb] "s"(wsgpr(nblk)), [c0] "s"(wsgpr(c[0])), [c1] "s"(wsgpr(c[1])), [c2] "s"(wsgpr(c[2])),
                  [c3] "s"(wsgpr(c[3])), [c4] "s"(wsgpr(c[4])), [c5] "s"(wsgpr(c[5])), [c6] "s"(wsgpr(c[6])), [c7] "s"(wsgpr(c[7])),
                  [c8] "s"(wsgpr(c[8])), [c9] "s"(wsgpr(c[9])), [c10] "s"(wsgpr(c[10])), [c11] "s"(wsgpr(c[11])), [c12] "s"(wsgpr(c[12])),
                  [c13] "s"(wsgpr(c[13])), [c14] "s"(wsgpr(c[14])), [c15] "s"(wsgpr(c[15]))
                : "scc", "m0", "memory", WS_FIXED_REGS);
            __builtin_amdgcn_s_setprio(0);
        };

        // ---- the index stage's tile: rows 4 tq .. 4 tq + 3 start at (the tile's first batch), o1, o2, o3; its batches end at gendq
        int tq = 0, gq = 0, o1 = 0, o2 = 0, o3 = 0, gendq = 0, elastq = 0;
        int64_t rb0q = 0;
        __amdgpu_buffer_rsrc_t rI;
#define WS_TILE_CTX()                                                                                                        \
    {                                                                                                                        \
        const int l0_ = (4 * tq) & 63;                                                                                       \
        o1 = __builtin_amdgcn_readlane(ring.start, l0_ + 1);                                                                 \
        o2 = __builtin_amdgcn_readlane(ring.start, l0_ + 2);                                                                 \
        o3 = __builtin_amdgcn_readlane(ring.start, l0_ + 3);                                                                 \
        elastq = __builtin_amdgcn_readlane(ring.end, l0_ + 3);                                                               \
        gendq = __builtin_amdgcn_readlane(ring.start, (l0_ + 4) & 63) >> 6;                                                  \
        rb0q = ((int64_t)__builtin_amdgcn_readlane(ring.rbhi, l0_) << 32) | (uint32_t)__builtin_amdgcn_readlane(ring.rblo, l0_); \
        rI = __builtin_amdgcn_make_buffer_rsrc((void*)(idx + rb0q), 0, __builtin_amdgcn_readlane(ring.span, l0_) * 4, 0x00020000); \
    }
        // index (and explicit neighbour value) of the pair at this lane's position of batch gq; kk = the ring entry of its row;
        // np = the batch's pairs (positions past the tile's last pair hold none)
#define WS_IDX_STAGE(J, NV, KK, NP)                                                                                          \
    {                                                                                                                        \
        if (gq >= gendq) {                                                                                                   \
            ++tq;                                                                                                            \
            WS_TILE_CTX()                                                                                                    \
        }                                                                                                                    \
        ws_ld_idx(64 * gq + lane, 4 * tq, o1, o2, o3, Tab, rI, nval, rb0q, J, NV, KK);                                       \
        NP = max(0, min(64, elastq - 64 * gq));                                                                              \
        ++gq;                                                                                                                \
    }
        WS_TILE_CTX()

        int ti = 0;      // the splat stage's tile
        int rdone = 0;   // ... and its next row to complete
        // Row rdone is done: merge its tiles in registers (lanes 0 .. 31: planes 0, 1; lanes 32 .. 63: planes 2, 3 of the lane's
        // channel), store all 32 channels to its B row, clear the tiles for the next point.
        auto complete = [=, &rdone]() {
            const int row = kWPts * wave + (rdone & 3);
            const uint32_t b = row_lane + (uint32_t)row * (kWRow * 4u) + ((uint32_t)((ch & 15) ^ (row & 15)) << 4);
            asm volatile(
#include "cconv_pair_merge.inc"
                ::: "memory", WS_FIXED_REGS);
            asm volatile(
#include "cconv_ws_store.inc"
                :: [b] "v"(b) : "memory", WS_FIXED_REGS);
            asm volatile(
#include "cconv_pair_zero.inc"
                ::: "memory", WS_FIXED_REGS);
            ++rdone;
        };

        // Stages (nothing hides a round trip at one producer per SIMD but the consumer's matrix instructions, so every load is
        // issued a whole splat before its first use): indices three batches ahead, positions two, geometry + index push + ALL
        // feature loads of batch g + 1 before the splat of batch g, published after it.
#ifdef WS_TRACE
        uint64_t wt[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t wlast = __builtin_readcyclecounter();
        const uint64_t wstart = wlast;
        int nbatches = 0;
#endif
        int j1, j2, kk1, kk2, np0, np1, np2;
        float nv1, nv2, px, py, pz;
        uint32_t cc[16];
        f32x4 ff[8];
        {
            int j0, kk0;
            float nv0, qx, qy, qz;
            bool ok0;
            WS_IDX_STAGE(j0, nv0, kk0, np0)
            WS_IDX_STAGE(j1, nv1, kk1, np1)
            WS_IDX_STAGE(j2, nv2, kk2, np2)
            ws_ld_pos(inp_pos, j0, qx, qy, qz);
            ws_ld_pos(inp_pos, j1, px, py, pz);
            const WsRec first = ws_geom(lane, kk0, Tab, window, nval, imp, inv_r2, window_fac, inv_extent, j0, nv0, qx, qy, qz, ok0);
            push_index(j0, ok0);
            push_rec(first);
            pk_cur = pack_classes(first.cls4, cc);
            wfence();
            f_issue(np0, ff);
        }
        WT(0)
        WS_BARRIER();  // "free": the B rows (and the staging in them) are ours
        WT(5)
        f_publish(np0, ff);
        wfence();
        WT(6)
        // the splat stage's tile: its real end in blocks, the batch after its last
        // (an empty last row owns one block, so that a tile's last row completes in the tile's last batch)
        int nblk_tile = (max(__builtin_amdgcn_readlane(ring.end, 3), __builtin_amdgcn_readlane(ring.start, 3) + 1) + 7) >> 3;
        int gend = __builtin_amdgcn_readlane(ring.start, 4) >> 6;
#pragma unroll 1
        for (int g = 0;; ++g) {
            // here: Rec / Fst / cc = batch g; (j1, nv1, kk1, px, py, pz, np1) = batch g + 1; (j2, nv2, kk2, np2) = batch g + 2
            const bool tile_last = g + 1 == gend;
            const bool more = !(tile_last && ti + 1 == T);
            WsRec nxt;
            int jn, kkn, npn;
            float nvn, qx, qy, qz;
            if (more) {
                bool ok1;
                nxt = ws_geom(64 * (g + 1) + lane, kk1, Tab, window, nval, imp, inv_r2, window_fac, inv_extent, j1, nv1, px, py, pz, ok1);
                push_index(j1, ok1);
                wfence();
                f_issue(np1, ff);
            }
            ws_ld_pos(inp_pos, j2, qx, qy, qz);
            WS_IDX_STAGE(jn, nvn, kkn, npn)
            WT(0)
            // ONE splat site, run once per segment of the batch: a segment ends where a point's pairs end
            const int blo = 8 * g, bhi = min(8 * g + 8, nblk_tile);
            int b0 = blo;
            bool tile_done = false;
            for (;;) {
                const int gbk = (rdone & 3) == 3 ? nblk_tile : (__builtin_amdgcn_readlane(ring.start, (rdone + 1) & 63) >> 3);
                const int b1 = min(gbk, bhi);
                if (b1 > b0) {
                    const int sh = b0 - blo;
                    if (sh > 0) {  // (cc is dead after this batch: the next one packs its own)
#pragma unroll
                        for (int m = 0; m < 16; ++m) cc[m] = (uint32_t)__builtin_amdgcn_readlane(pk_cur, (4 * m + 8 * sh) & 63);
                    }
                    splat(b1 - b0, cc, a_rec0 + (uint32_t)(2 * kWRecG * 4) * (uint32_t)sh, a_fst0 + 1024u * (uint32_t)sh);
                    b0 = b1;
                    WT(1)
                }
                if (gbk > bhi) break;
                const bool last_row = (rdone & 3) == 3;
                complete();
                WT(2)
                if (last_row) {
                    tile_done = true;
                    break;
                }
            }
            // (the compiler may not move the rotation of the in-flight loads -- register copies, each behind a wait for every load
            // issued before it -- in front of the splat: volatile asm statements keep their order)
            asm volatile("" : "+v"(jn), "+v"(nvn), "+v"(qx), "+v"(qy), "+v"(qz), "+v"(kkn));
#ifdef WS_TRACE
            ++nbatches;
#endif
            WT(7)
            if (tile_done) {
                WS_BARRIER();  // "full": the 16 rows of the tile are in LDS
                WT(3)
                ++ti;
                if ((ti & 7) == 0) ws_refill(kp0, (ti >> 3) + 1, lane, wave, t_begin, t_end, nslots, Tab, stream_end, ring);
                nblk_tile = (max(__builtin_amdgcn_readlane(ring.end, (4 * ti + 3) & 63), __builtin_amdgcn_readlane(ring.start, (4 * ti + 3) & 63) + 1) + 7) >> 3;
                gend = __builtin_amdgcn_readlane(ring.start, (4 * ti + 4) & 63) >> 6;
                WT(4)
                WS_BARRIER();  // "free": the consumers hold the tile in registers
                WT(5)
            }
            if (!more) break;
            wfence();
            f_publish(np1, ff);
            push_rec(nxt);
            pk_cur = pack_classes(nxt.cls4, cc);
            j1 = j2;
            nv1 = nv2;
            kk1 = kk2;
            np1 = np2;
            j2 = jn;
            nv2 = nvn;
            kk2 = kkn;
            np2 = npn;
            px = qx;
            py = qy;
            pz = qz;
            wfence();
            WT(6)
        }
        WS_BARRIER();  // (the barrier behind which the consumers store the last tile's sums)
#ifdef WS_TRACE
        if (lane == 0 && wave == 0 && (blockIdx.x & 15) == 0) {
            for (int k = 0; k < 8; ++k) atomicAdd(&g_wtrace[k], wt[k]);
            atomicAdd(&g_wtrace[8], wlast - wstart);
            atomicAdd(&g_wtrace[9], (unsigned long long)T);
            atomicAdd(&g_wtrace[10], (unsigned long long)nbatches);
        }
#endif
    } else {
        // =============================================== CONSUMER ===============================================
        const int cw = wave - kWProd;             // 0 .. 3
        const int ctid = tid - 64 * kWProd;       // 0 .. 255
        const int mi = lane & 15, mg = lane >> 4;
        float* const red0 = smem + WTM * kWRow + kWProd * kWWaveF;  // two buffers of [kWCons][16][16 * NT]
        const int red_half = kWCons * WTM * 16 * kWMaxNT;
        int tpar = 0;  // (tile parity: which buffer this tile's sums go to)
        const int NT = kp0->NT, ncol = 16 * NT;
        // This consumer's k' blocks: b = cw + 4 it of the 16 (z, y) x cin / 4 channel quads, b = zy * nqt + quad (it < cin).  What a
        // block needs -- the offset of its A fragments in a B row, the offset of its filter fragments in the packed filter -- is
        // the same for every tile: lane it of two registers holds them (one v_readlane per use; computed per block in scalar
        // code, they were hoisted out of the tile loop and spilled).  The block count is rounded up to a multiple of the ring of
        // filter fragments (8 blocks in flight, which runs on across tiles), so that block it finds its fragments in slot it % 8;
        // the blocks past cin in that count have A fragments of zeros and filter offsets outside the buffer (zeros as well).
        constexpr int kRing = 8;
        const int cin_ring = (cin + kRing - 1) & ~(kRing - 1);
        const int nqt = cin >> 2;
        int a_tab, w_tab, w_next;  // lane it: block it's A offset (bytes, inside a row: chunk and k' block) / filter offset / block it + 8's
        {
            const int b = cw + 4 * (lane & 31);
            const int zy = b / nqt, qg = b - zy * nqt;
            const int chunk = qg >> 2, blk = zy * 4 + (qg & 3);
            const bool real = (lane & 31) < cin;
            a_tab = real ? chunk * 4096 + blk * 64 : -1;
            w_tab = real ? (chunk * 64 + blk) * NT * 1024 : 0x7fff0000;
            int nx = (lane & 31) + kRing;
            if (nx >= cin_ring) nx -= cin_ring;
            w_next = __shfl(w_tab, nx);
        }
        const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(
            (void*)kp0->Wp, 0, (int)((uint32_t)kp0->nchunks * 64u * (uint32_t)NT * 1024u), 0x00020000);
        uint32_t w_lane = ((uint32_t)mg * (uint32_t)NT * 16u + (uint32_t)mi) * 16u;
        f32x4 bw[kRing][NTT];
        // A fragments: lane (row mi, k' group mg) of block blk of chunk reads 4 floats at row * kWRow + chunk * 1024 +
        // ((blk * 16 + mg * 4) ^ (mi << 2)) = ... + ((blk * 16) ^ lane_x)
        uint32_t a_lane = wlds(Bt + mi * kWRow);
        const uint32_t lane_x = (uint32_t)((mg << 2) ^ (mi << 2)) << 2;
        // epilogue roles: this lane's outputs e = ctid + 256 r of the tile (point e / cout, channel e % cout) -- the same for every tile
        int e_pt[NTT], e_o[NTT];
        float e_bias[NTT];
        const float* const bias = kp0->bias;
        const bool accumulate = (kp0->flags & DMCF_FLAG_ACCUMULATE) != 0;
#pragma unroll
        for (int r = 0; r < NTT; ++r) {
            const int e = ctid + 256 * r;
            e_pt[r] = e / cout;
            e_o[r] = e - e_pt[r] * cout;
            e_bias[r] = (bias && e < WTM * cout) ? bias[e_o[r]] : 0.0f;
        }
        float prev[NTT];
        int64_t prev_pt0 = -1;
        auto reduce_store = [&](WsKP kp, const float* red) {
            const int64_t n_out = kp->n_out;
            float* const out = kp->out;
#pragma unroll
            for (int r = 0; r < NTT; ++r) {
                const int e = ctid + 256 * r;
                const int64_t ii = prev_pt0 + e_pt[r];
                if (e < WTM * cout && ii < n_out) {
                    float v = 0.0f;
#pragma unroll
                    for (int w = 0; w < kWCons; ++w) v += red[(w * 16 + e_pt[r]) * ncol + e_o[r]];
                    if (bias) v += e_bias[r];
                    if (accumulate) v += prev[r];
                    out[ii * cout + e_o[r]] = v;
                }
            }
        };
#define WS_RING(IT) bw[(IT) % kRing]
#define WS_WLOAD(IT, OFF)                                                                                                    \
    {                                                                                                                        \
        const uint32_t o_ = w_lane + (uint32_t)__builtin_amdgcn_readlane(OFF, IT);                                           \
        WS_RING(IT)[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, o_, 0, 0));                     \
        if (NTT > 1) WS_RING(IT)[1 % NTT] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rW, o_ + 256u, 0, 0)); \
    }
        WS_WLOAD(0, w_tab) WS_WLOAD(1, w_tab) WS_WLOAD(2, w_tab) WS_WLOAD(3, w_tab)
        WS_WLOAD(4, w_tab) WS_WLOAD(5, w_tab) WS_WLOAD(6, w_tab) WS_WLOAD(7, w_tab)
        WS_BARRIER();  // "free" of the first tile
#ifdef WS_TRACE
        uint64_t wt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t wlast = __builtin_readcyclecounter();
        const uint64_t wstart = wlast;
#endif
#pragma unroll 1
        for (int tile = t_begin; tile < t_end; tile += nslots) {
            const int64_t pt0 = (int64_t)tile * WTM;
            WsKP kp = kp0;
            asm volatile("" : "+s"(kp), "+v"(a_lane), "+v"(w_lane), "+v"(a_tab), "+v"(w_next));
            WS_BARRIER();  // "full"
            WT(0)
#ifdef WS_DBG_NOCONS
            WS_BARRIER();
            continue;
#endif
            // ---- pull this consumer's blocks of the 16 rows into v128 .. v255
#define WS_PULL1(IT)                                                                                         \
    {                                                                                                            \
        const int ao_ = __builtin_amdgcn_readlane(a_tab, IT);                                                    \
        if (ao_ >= 0) {                                                                                          \
            WS_PULL(IT, a_lane + (((uint32_t)ao_) ^ lane_x));                                                    \
        } else {                                                                                                 \
            asm volatile("v_mov_b64 v[128+4*(" #IT "):128+4*(" #IT ")+1], 0\n\tv_mov_b64 v[128+4*(" #IT ")+2:128+4*(" #IT ")+3], 0" ::: WS_FIXED_REGS); \
        }                                                                                                        \
    }
#define WS_PULL8(G)                                                                                              \
    if (8 * (G) < cin_ring) {                                                                                    \
        WS_PULL1(8 * (G) + 0) WS_PULL1(8 * (G) + 1) WS_PULL1(8 * (G) + 2) WS_PULL1(8 * (G) + 3)                  \
        WS_PULL1(8 * (G) + 4) WS_PULL1(8 * (G) + 5) WS_PULL1(8 * (G) + 6) WS_PULL1(8 * (G) + 7)                  \
    }
            WS_PULL8(0) WS_PULL8(1) WS_PULL8(2) WS_PULL8(3)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WT(1)
            WS_BARRIER();  // "free" (the pull has landed)
            WT(2)
            // ---- the previous tile's sums: every consumer wrote its part before the "full" barrier above.  They sit in the buffer
            // of the OTHER parity, which nobody writes before the next "full" barrier -- so this runs behind "free", off the
            // producers' critical path (in front of it, it held them up for as long as the pull itself: tools/wtrace.py)
            if (prev_pt0 >= 0) reduce_store(kp, red0 + (tpar ^ 1) * red_half);
            // ---- what the epilogue adds to, requested now (read behind the next barrier)
#pragma unroll
            for (int r = 0; r < NTT; ++r) {
                const int e = ctid + 256 * r;
                prev[r] = 0.0f;
                if (accumulate && e < WTM * cout && pt0 + e_pt[r] < kp->n_out) prev[r] = kp->out[(pt0 + e_pt[r]) * cout + e_o[r]];
            }
            prev_pt0 = pt0;
            // ---- contraction from registers
            // (a vector write -> matrix read as the accumulator: two wait states)
            asm volatile(".irp r,112,114,116,118,120,122,124,126\n\tv_mov_b64 v[\\r:\\r+1], 0\n\t.endr\n\ts_nop 1" ::: WS_FIXED_REGS);
            // Straight-line code with ONE way out per group of 8 blocks: the compiler then counts the filter loads in flight and
            // waits for exactly the block's own (s_waitcnt vmcnt(14)); with a branch around every block it waited for ALL of them
            // -- the eight blocks' worth just requested included: a round trip to L2 per block, 27k clocks per tile for the 8.2k
            // of matrix time of a 32 -> 32 layer (tools/wtrace.py).
#define WS_BLOCK(IT)                                                      \
    if (NTT > 1) {                                                        \
        WS_MFMA2(IT, WS_RING(IT)[0], WS_RING(IT)[1 % NTT]);               \
    } else {                                                              \
        WS_MFMA(IT, 0, WS_RING(IT)[0]);                                   \
    }                                                                     \
    WS_WLOAD(IT, w_next)
#define WS_BLOCK8(G)                                                                                             \
    WS_BLOCK(8 * (G) + 0) WS_BLOCK(8 * (G) + 1) WS_BLOCK(8 * (G) + 2) WS_BLOCK(8 * (G) + 3)                      \
    WS_BLOCK(8 * (G) + 4) WS_BLOCK(8 * (G) + 5) WS_BLOCK(8 * (G) + 6) WS_BLOCK(8 * (G) + 7)
            do {
                WS_BLOCK8(0)
                if (cin_ring <= 8) break;
                WS_BLOCK8(1)
                if (cin_ring <= 16) break;
                WS_BLOCK8(2)
                if (cin_ring <= 24) break;
                WS_BLOCK8(3)
            } while (false);
            // ---- partial sums (the results of the last matrix instructions are 11 wait states away from a vector read)
            asm volatile("s_nop 15" ::: WS_FIXED_REGS);
            f32x4 acc[NTT];
            WS_ACC_READ(0, acc[0]);
            if (NTT > 1) WS_ACC_READ(1, acc[1 % NTT]);
            if (NTT > 2) {
                WS_ACC_READ(2, acc[2 % NTT]);
                WS_ACC_READ(3, acc[3 % NTT]);
            }
#pragma unroll
            for (int n = 0; n < NTT; ++n) {
                if (n < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) red0[tpar * red_half + (cw * 16 + 4 * mg + r) * ncol + n * 16 + mi] = acc[n][r];
                }
            }
            tpar ^= 1;
            WT(3)
        }
        WS_BARRIER();  // every consumer's sums of the last tile are in LDS
#ifdef WS_TRACE
        if (lane == 0 && cw == 0 && (blockIdx.x & 15) == 0) {
            for (int k = 0; k < 4; ++k) atomicAdd(&g_wtrace[16 + k], wt[k]);
            atomicAdd(&g_wtrace[20], wlast - wstart);
        }
#endif
#ifndef WS_DBG_NOCONS
        reduce_store(kp0, red0 + (tpar ^ 1) * red_half);
#endif
    }
}

#ifdef WS_TRACE
}
extern "C" int dmcf_wtrace(unsigned long long* out) {
    unsigned long long z[32] = {0};
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dmcf::g_wtrace), sizeof(z));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(dmcf::g_wtrace), z, sizeof(z));
    return 0;
}
namespace dmcf {
#endif

static constexpr size_t kWsLds = (size_t)(WTM * kWRow + kWProd * kWWaveF + kWRed) * sizeof(float);

// Same filters and flags as cconv_pair.hip; 4 .. 32 input channels.
bool cconv_ws_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx) {
    const char* e = getenv("DMCF_CCONV_KERNEL");  // "ws": force, anything else: never
    if (e && e[0] != 'w') return false;
    if (dx != 4 || dy != 4 || dz != 4) return false;
    if (a->flags & DMCF_FLAG_SYMMETRIC) return false;
    if (a->coordinate_mapping != DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING || a->interpolation != DMCF_INTERP_LINEAR ||
        !(a->flags & DMCF_FLAG_ALIGN_CORNERS) || (a->flags & DMCF_FLAG_NORMALIZE))
        return false;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin & 3) || cin > 32 || cout > 16 * kWMaxNT) return false;
    if ((uintptr_t)a->inp_features & 15) return false;
    // 24-bit multiplies form the byte offsets of feature and position rows; the buffers must stay below 2 GB
    if (a->n_inp >= (1 << 24) || a->n_inp * (int64_t)cin * 4 >= ((int64_t)1 << 31)) return false;
    if (e) return true;
    // short rows (the layers at the network's base radius) whose contraction is at least as much work as their splat: the 32 -> 32
    // layer takes 2.92 ms here against 3.39 with splat E, 24 -> 16 2.30 against 2.62; 16 -> 32 and 8 -> 16 lose (2.31 / 1.91 against
    // 1.84 / 1.28: the producers are alone on their SIMDs and every phase of theirs runs at its latency) -- profiles/r05_microbench.md
    return a->row_length_hint == 1 && cin >= 24;
}

int cconv_ws_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream) {
    const int NT = (p.cout + 15) / 16;
    float* packed = (float*)workspace;
    const int nchunks = cconv_cls_pack(a, packed, stream);  // the B-fragment order of cconv_cls.hip, 16 channels per chunk
    p.Wp = packed;
    p.NT = NT;
    p.nchunks = nchunks;
    const int64_t ntiles = (p.n_out + WTM - 1) / WTM;
    if (ntiles > 0x7fffffff / 8) return DMCF_EUNSUPPORTED;
    p.ntiles = (int)ntiles;
    p.tiles_per_xcd = (int)((ntiles + 7) / 8);
    // persistent: one workgroup per CU (the LDS admits no second one), an eighth of them per XCD
    const int per_xcd = max(1, min(device_cu_count() / 8, p.tiles_per_xcd));
    const unsigned grid = (unsigned)per_xcd * 8u;
    const void* fn;
    if (cconv_plain(a))
        fn = NT <= 1 ? (const void*)cconv_ws_kernel<1, true> : (const void*)cconv_ws_kernel<2, true>;
    else
        fn = NT <= 1 ? (const void*)cconv_ws_kernel<1, false> : (const void*)cconv_ws_kernel<2, false>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWsLds);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    void* kargs[] = {(void*)&p};
    e = hipLaunchKernel(fn, dim3(grid), dim3(kWThreads), kargs, kWsLds, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

}  // namespace dmcf
