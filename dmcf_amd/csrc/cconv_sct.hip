// CConv onto a coarse LATTICE with few output channels, FILTER FIRST and INPUT STATIONARY -- splat S (round 6).
//
// The layers that gather the particles (s0, ~1.1M points, 24 / 16 channels) onto the coarsest grid_pos lattice (s2, ~150k points,
// 4 / 8 channels; models/hrnet.py:83-93, radius 0.4 = 4 lattice cells: ~1900 pairs per row, 2.9e8 pairs) spend their time in
// the splat: 8 trilinear corners x Cin multiply-adds per pair into the output point's B tile, then one small contraction.  With
// Cout <= 8 the other order is cheaper by Cin / Cout:
//
//     G_j[cell][o] = sum_c f_j[c] W[cell][c][o]                           once per INPUT point   (64 cells x Cout: 1 - 2 KB)
//     out_i[o]    += a_ij * sum_{8 corners k} w_k(ij) G_j[cell_k][o]      per pair: 8 x Cout multiply-adds, every lane a pair
//
// A gather form of this reads 8 x Cout scattered floats of G_j per pair from a 1 - 2 GB array: cache arithmetic kills it
// (splat S).  So the walk is TRANSPOSED: a wave owns an input point j, keeps G_j in LDS, and walks j's row of
// the TRANSPOSED list (the s2 -> s0 list the step builds anyway: row j = the lattice points within R of particle j), one pair
// per lane: geometry, 8 LDS reads of G_j, 8 x Cout / 2 packed multiply-adds, and an ADD INTO THE OUTPUT POINT.  Those adds are the
// problem of every scatter form -- 2.9e8 x Cout of them, and a float sum whose order depends on the schedule.  Here:
//
//   * input points are processed in BLOCKS of m x m x m lattice cells (dmcf_cconv_scatter_plan counting-sorts them by block on the
//     device).  All outputs a block can reach lie in a box of D^3 lattice cells, D = m + 2 reach + 1, whose accumulators live
//     in LDS ([Cout][D^3] 64-bit integers, channel major: conflict free for distinct slots).  A block of ~500 particles (m = 4) adds
//     ~130k pairs into ~2000 slots and flushes each touched slot ONCE with a global atomic: ~70 pairs per flushed value.
//   * sums are FIXED POINT: a contribution c is added as round(c * 2^s) into a 64-bit integer, 2^s = 2^46 / (a power-of-two bound
//     of |c|: cin max |f| * max |W| >= max_j |f_j|_1 max |W|, formed on the device inside the call).  Integer addition is associative, so LDS
//     atomics, global atomics and any schedule give THE SAME BITS: the step stays bit reproducible with no ordering, no staging
//     and no barrier in the pair loop.  Resolution: 2^-46 of the bound per term (terms are float32: 2^-24 of themselves; 2^-30
//     was tried first and showed in the dam break, where a few splashing particles set a bound a thousand times a typical term).
//
// Workgroup = 8 or 16 waves (sct_waves: 8 while two workgroups fit a CU's LDS -- block_cells 2 --, else 16 in one -- block_cells 4,
// the default: a quarter of the flushes).  A block's rows go by in chunks of 16 (32): G of the chunk = F[rows x Cin] . W[Cin x 64 Cout]
// on the matrix cores (the filter stays in registers as B fragments, each wave forms its 16-column tiles), double buffered in LDS,
// one barrier per chunk; then every wave walks two of the chunk's rows as one stream of 64-pair batches with the indices two
// batches and the output positions one batch ahead.  LDS at Cout = 4, block_cells 4: accumulators 70 KB + slot -> output index
// 8.8 KB + G chunks 64 KB.  Measured (profiles/r06_kernel_pmc.md, case S4): 4.1 vector + 0.7 scalar + 0.26 LDS instructions per pair,
// VALU active 78 % at 3.75 waves per SIMD (splat F on the same pairs: 4.0 + 2.7 + 0.9 and one matrix instruction per pair, 65 % busy
// at 1.8 waves).
//
// Restrictions (the dispatch in dmcf_amd/utils/convolutions.py checks them, the entry point returns DMCF_EUNSUPPORTED): 4x4x4 filter,
// Cout 4 or 8, Cin <= 32, linear interpolation, align_corners, volume-preserving map, poly6 or no window (formed from the positions),
// no per-point importance, no normalisation; output points on a lattice of spacing `voxel` (cell = rint((x - x_0) / voxel)).
#include <algorithm>

#include "cconv_common.h"

namespace dmcf {


// The plan (dmcf_cconv_scatter_plan): input points counting-sorted by the block of m^3 lattice cells they lie in.  Blocks live in
// a table over a REGION of at most kSRegion^3 blocks around the points' mean; a point outside it (a stray far from the fluid)
// becomes a block of its own (`overflow`).  Orders inside the plan (points inside a block, the list of non-empty blocks) come
// out of atomics and differ from run to run -- which is fine exactly because the sums are integers.
constexpr int kSRegion = 128;
constexpr int64_t kSTable = (int64_t)kSRegion * kSRegion * kSRegion;

struct SctHeader {
    float origin[3];       // out_positions[0]: lattice cell of x = rint((x - origin) / voxel)
    float inv_voxel;
    int32_t reg_min[3];    // block coordinates (units of m cells) of the region's first block
    int32_t dims[3];
    int32_t ncells;
    int32_t n_blocks;      // non-empty table cells
    int32_t n_overflow;
    int32_t m, reach;
    uint32_t bb_min[3], bb_max[3];  // order-preserving encoding of the points' bounding box
    double sum[3];
    int32_t pad[8];
};
static_assert(sizeof(SctHeader) <= 256, "header layout");

struct SctPlanLayout {
    size_t off_header, off_cell_start, off_cell_fill, off_key, off_sorted, off_blocks, off_overflow, off_scan, scan_bytes, total;
};

static SctPlanLayout sct_plan_layout(int64_t n) {
    SctPlanLayout L;
    size_t o = 0;
    L.off_header = o;      o += 256;
    L.off_cell_start = o;  o += align_up((size_t)(kSTable + 1) * 4, 256);
    L.off_cell_fill = o;   o += align_up((size_t)(kSTable + 1) * 4, 256);
    L.off_key = o;         o += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    L.off_sorted = o;      o += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    L.off_blocks = o;      o += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    L.off_overflow = o;    o += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    L.scan_bytes = scan_tmp_bytes(kSTable + 1);
    L.off_scan = o;        o += align_up(L.scan_bytes, 256);
    L.total = o;
    return L;
}

struct SctParams {
    const float* W;          // [4][4][4][cin][cout]
    const float* out_pos;    // [n_out][3]
    const float* inp_pos;    // [n_inp][3]
    const float* inp_feat;   // [n_inp][cin]
    const int32_t* t_idx;    // transposed list: row j = output indices of input point j
    const int64_t* t_rs;     // row begin (CSR row splits, or j * stride for padded rows)
    const int32_t* t_cnt;    // optional pairs per row (padded rows); NULL = CSR
    int64_t t_cap;           // entries t_idx holds
    const SctHeader* hdr;    // the plan
    const uint32_t* cell_start;
    const int32_t* sorted;
    const int32_t* blocks;
    const int32_t* overflow;
    const uint32_t* bound;   // device [2]: float bits of max_j |f_j|_1 and max |W| (sct_bound_kernel)
    unsigned long long* acc; // [n_out][cout] 64-bit sums (zeroed by the launch)
    int64_t n_out, n_inp;
    int cin, D;
    float inv_extent, inv_r2, window_fac;
    int window;
    int* err;                // device flag: a pair fell outside its block's slot box (must stay 0)
    int* counter;            // device counter the workgroups draw their blocks from (zeroed by the launch)
};

__device__ __forceinline__ uint32_t sct_f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sct_ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// 2^s with 2^s * bound <= 2^46 (bound = max_j |f_j|_1 * max |W| >= any pair's contribution; rows hold < 2^16 pairs)
__device__ __forceinline__ float sct_scale(const uint32_t* bound, float window_fac) {
    const float b = __uint_as_float(bound[0]) * __uint_as_float(bound[1]) * fmaxf(1.0f, fabsf(window_fac));
    int e = 0;
    if (b > 0.0f && isfinite(b)) (void)frexpf(b, &e);  // b = m 2^e, 0.5 <= m < 1
    return ldexpf(1.0f, 46 - e);
}

// round(y) as a 64-bit integer for |y| <= 2^46, exact above 1: y = hi 2^23 + lo with both parts exact in float32
__device__ __forceinline__ long long sct_fixed(float y) {
    const float hi = truncf(y * 0x1p-23f);
    const float lo = fmaf(-hi, 0x1p23f, y);
    return (long long)(int)hi * (1ll << 23) + (long long)__float2int_rn(lo);
}

// ---- plan kernels -------------------------------------------------------------------------------------------------------
__global__ void sct_plan_init(SctHeader* h, const float* __restrict__ out_pos, float voxel, int m, int reach) {
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            h->origin[a] = out_pos[a];
            h->bb_min[a] = 0xffffffffu;
            h->bb_max[a] = 0u;
            h->sum[a] = 0.0;
        }
        h->inv_voxel = 1.0f / voxel;
        h->n_blocks = 0;
        h->n_overflow = 0;
        h->m = m;
        h->reach = reach;
    }
}

__global__ __launch_bounds__(256) void sct_plan_stats(const float* __restrict__ pts, int64_t n, SctHeader* h) {
    __shared__ float red[4][9];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY}, sm[3] = {0.0f, 0.0f, 0.0f};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            if (isfinite(v)) {
                mn[a] = fminf(mn[a], v);
                mx[a] = fmaxf(mx[a], v);
                sm[a] += v;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, kWave));
            sm[a] += __shfl_xor(sm[a], d, kWave);
        }
        if (lane_id() == 0) {
            red[threadIdx.x >> 6][a] = mn[a];
            red[threadIdx.x >> 6][3 + a] = mx[a];
            red[threadIdx.x >> 6][6 + a] = sm[a];
        }
    }
    __syncthreads();
    // (one set of atomics per workgroup: from every wave of a 2048-workgroup grid they took 0.77 ms on nine addresses)
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        float lo = red[0][a], hi = red[0][3 + a];
        double su = red[0][6 + a];
        for (int w = 1; w < 4; ++w) {
            lo = fminf(lo, red[w][a]);
            hi = fmaxf(hi, red[w][3 + a]);
            su += red[w][6 + a];
        }
        atomicMin(&h->bb_min[a], sct_f2ord(lo));
        atomicMax(&h->bb_max[a], sct_f2ord(hi));
        atomicAdd(&h->sum[a], su);
    }
}

// Lanes of a wave that hold the same key in a ROW (consecutive points of an ordered scene share their block) add to the key's
// counter ONCE: `leader` = first lane of the caller's run, `len` = its length.  A million single adds on a few thousand addresses
// serialise in L2 (0.17 ms per pass on the 100^3 box).
__device__ __forceinline__ void sct_runs(int k, int& leader, int& len) {
    const int lane = lane_id();
    const int prev = __shfl_up(k, 1, kWave);
    const bool start = lane == 0 || k != prev;
    const unsigned long long mask = __ballot(start);
    const unsigned long long upto = mask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    leader = 63 - __clzll(upto);
    const unsigned long long rest = leader == 63 ? 0ull : (mask >> (leader + 1));
    len = rest ? __ffsll((long long)rest) : 64 - leader;
}

__device__ __forceinline__ int sct_block_coord(float x, float origin, float inv_voxel, int m) {
    return (int)floorf((x - origin) * inv_voxel / (float)m);
}

__global__ void sct_plan_region(SctHeader* h, int64_t n) {
    if (threadIdx.x != 0) return;
    int64_t cells = 1;
    for (int a = 0; a < 3; ++a) {
        const float lo = sct_ord2f(h->bb_min[a]), hi = sct_ord2f(h->bb_max[a]);
        int b0 = 0, b1 = 0;
        if (lo <= hi) {
            b0 = sct_block_coord(lo, h->origin[a], h->inv_voxel, h->m);
            b1 = sct_block_coord(hi, h->origin[a], h->inv_voxel, h->m);
        }
        if ((int64_t)b1 - b0 + 1 > kSRegion) {  // strays: the region is centred on the mean, what falls outside becomes overflow rows
            const int bc = sct_block_coord((float)(h->sum[a] / (double)(n > 0 ? n : 1)), h->origin[a], h->inv_voxel, h->m);
            b0 = bc - kSRegion / 2;
            b1 = b0 + kSRegion - 1;
        }
        h->reg_min[a] = b0;
        h->dims[a] = b1 - b0 + 1;
        cells *= h->dims[a];
    }
    h->ncells = (int32_t)cells;
}

__global__ __launch_bounds__(256) void sct_plan_key(const float* __restrict__ pts, int64_t n, const SctHeader* __restrict__ h,
                                                    int32_t* __restrict__ key, uint32_t* __restrict__ cell_count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int k = 0;
    bool in = i < n;
    int stride = 1;
    if (i < n) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            const int b = isfinite(v) ? sct_block_coord(v, h->origin[a], h->inv_voxel, h->m) - h->reg_min[a] : -1;
            in = in && b >= 0 && b < h->dims[a];
            k += b * stride;
            stride *= h->dims[a];
        }
    }
    k = in ? k : -1;
    if (i < n) key[i] = k;
    int leader, len;
    sct_runs(k, leader, len);
    if (k >= 0 && lane_id() == leader) atomicAdd(&cell_count[k], (uint32_t)len);
}

__global__ __launch_bounds__(256) void sct_plan_scatter(int64_t n, SctHeader* h, const int32_t* __restrict__ key,
                                                        const uint32_t* __restrict__ cell_start, uint32_t* __restrict__ cell_fill,
                                                        int32_t* __restrict__ sorted, int32_t* __restrict__ overflow) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k = i < n ? key[i] : -2;  // (-1: outside the region; -2: past the end)
    int leader, len;
    sct_runs(k, leader, len);
    uint32_t base = 0;
    if (lane_id() == leader) {
        if (k >= 0) base = cell_start[k] + atomicAdd(&cell_fill[k], (uint32_t)len);
        else if (k == -1) base = (uint32_t)atomicAdd(&h->n_overflow, len);
    }
    base = __shfl(base, leader, kWave) + (uint32_t)(lane_id() - leader);
    if (k >= 0) sorted[base] = (int32_t)i;
    else if (k == -1) overflow[base] = (int32_t)i;
}

__global__ __launch_bounds__(256) void sct_plan_blocks(SctHeader* h, const uint32_t* __restrict__ cell_start, int32_t* __restrict__ blocks) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= h->ncells) return;
    if (cell_start[c + 1] > cell_start[c]) blocks[atomicAdd(&h->n_blocks, 1)] = (int32_t)c;
}

// cin * max |f| (>= max_j |f_j|_1) and max |W| as float bits (non-negative floats order like their bits).  The row-wise 1-norm was
// the first form: one 96-byte row per lane, 0.125 ms on the 1.1M x 24 features; a flat, coalesced maximum is 4x faster and costs at
// most log2(cin) < 5 of the 46 bits.
__global__ __launch_bounds__(256) void sct_bound_kernel(const float* __restrict__ feat, int64_t n, int cin, const float* __restrict__ W, int64_t nw,
                                                        uint32_t* __restrict__ bound) {
    float f1 = 0.0f, wm = 0.0f;
    const int64_t total = n * cin, t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if ((total & 3) == 0 && (((uintptr_t)feat) & 15) == 0) {
        const f32x4* f4 = (const f32x4*)feat;
        for (int64_t i = t0; i < (total >> 2); i += stride) {
            const f32x4 v = f4[i];
            f1 = fmaxf(fmaxf(f1, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else {
        for (int64_t i = t0; i < total; i += stride) f1 = fmaxf(f1, fabsf(feat[i]));
    }
    f1 *= (float)cin;
    for (int64_t i = t0; i < nw; i += stride) wm = fmaxf(wm, fabsf(W[i]));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        f1 = fmaxf(f1, __shfl_xor(f1, d, kWave));
        wm = fmaxf(wm, __shfl_xor(wm, d, kWave));
    }
    if (lane_id() == 0) {
        if (f1 > 0.0f) atomicMax(&bound[0], __float_as_uint(f1));
        if (wm > 0.0f) atomicMax(&bound[1], __float_as_uint(wm));
    }
}

__device__ __forceinline__ void lds_add_i64(unsigned long long* p, long long v) {
    __hip_atomic_fetch_add(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Diagnostic builds (wrong results, right timing of what is left): -DSX_NOFLUSH no global atomics, -DSX_NOATOM no LDS adds,
// -DSX_NOGATHER G_j read at one fixed cell.  Compiled out of the product library.
// WAVES = 8: chunks of 16 rows, two workgroups per CU when the accumulator box is small (block_cells 2 at Cout = 4);
// WAVES = 16: chunks of 32 rows, one workgroup per CU with a box of up to 13^3 slots (block_cells 4: a quarter of the flushes).
template <int COUT, int WAVES>
__global__ __launch_bounds__(64 * WAVES, (COUT == 4 && WAVES == 8) ? 2 : 1) void cconv_sct_kernel(const SctParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int kSWaves = WAVES, kSThreads = 64 * WAVES;
    constexpr int GROW = 64 * COUT;          // floats per G row
    constexpr int NTW = GROW / 16 / kSWaves;  // 16-column tiles of the G chunk per wave
    constexpr int MT = WAVES / 8;             // 16-row tiles per chunk
    constexpr int CH = 16 * MT;               // rows per chunk: every wave walks two of them
    static_assert(NTW >= 1, "a wave needs at least one tile");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, q4 = lane >> 4;
    const int cin = p.cin, D = p.D, NS = D * D * D;
    const int NSp = (NS + 3) & ~3;
    // LDS: accumulators [COUT][NSp] u64 | slot -> output index [NSp] | G chunks [2][CH rows][GROW]
    unsigned long long* Acc = (unsigned long long*)smem_raw;
    int* Sidx = (int*)(Acc + (size_t)COUT * NSp);
    float* Gc = (float*)(Sidx + NSp);

    // the filter as B fragments of v_mfma_f32_16x16x4_f32: G[16 rows][GROW] = F[16][cin] . Wf[cin][GROW], Wf[k][cell * COUT + o] =
    // W[cell][k][o]; lane (r16, q4) holds Wf[16 j + 4 q4 + i][16 t + r16] for its wave's tiles t (as misc.hip: dense_rows)
    float wf[2][4][NTW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int tt = 0; tt < NTW; ++tt) {
                const int k = 16 * j + 4 * q4 + i, col = 16 * (wave * NTW + tt) + r16;
                wf[j][i][tt] = k < cin ? p.W[((int64_t)(col / COUT) * cin + k) * COUT + (col % COUT)] : 0.0f;
            }
    for (int s = tid; s < COUT * NSp; s += kSThreads) Acc[s] = 0ull;
    for (int s = tid; s < NSp; s += kSThreads) Sidx[s] = -1;
    const float S = sct_scale(p.bound, p.window_fac);
    const SctHeader* const H = p.hdr;
    const float ox0 = H->origin[0], oy0 = H->origin[1], oz0 = H->origin[2], inv_voxel = H->inv_voxel;
    const int n_blocks = H->n_blocks, n_work = n_blocks + H->n_overflow;
    CconvParams gp;  // (only what filter_coords<false> reads)
    gp.inv_extent = p.inv_extent;
    gp.sx = gp.sy = gp.sz = 4;
    __syncthreads();

    // the A operand of a chunk: 16 consecutive sorted rows, lane (r16, q4) holds F[row r16][16 j + 4 q4 .. + 3]
    auto load_a = [&](const int32_t* rows, int first, int s1, f32x4 (&a)[2]) {
        const int rp = first + r16;
        const int64_t j = rp < s1 ? rows[rp] : -1;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            a[jj] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            const int k = 16 * jj + 4 * q4;
            if (j >= 0 && k < cin) {
                const float* fp = p.inp_feat + j * cin + k;
                if ((cin & 3) == 0) {
                    a[jj] = *(const f32x4*)fp;
                } else {
                    a[jj].x = fp[0];
                    if (k + 1 < cin) a[jj].y = fp[1];
                    if (k + 2 < cin) a[jj].z = fp[2];
                    if (k + 3 < cin) a[jj].w = fp[3];
                }
            }
        }
    };

    __shared__ int next_block;
    for (;;) {
        // blocks are handed out by a counter: they differ in size (the fluid's surface, the boundary shell)
        if (tid == 0) next_block = atomicAdd(p.counter, 1);
        __syncthreads();
        const int b = next_block;
        if (b >= n_work) break;
        // a table cell's rows, or ONE overflow row as a block of its own (its box starts `reach` cells below the row's own cell)
        const int32_t* rows;
        int s0, s1, bx0, by0, bz0;
        if (b < n_blocks) {
            const int c = p.blocks[b];
            rows = p.sorted;
            s0 = (int)p.cell_start[c];
            s1 = (int)p.cell_start[c + 1];
            const int cxy = c / H->dims[0];
            bx0 = (c - cxy * H->dims[0] + H->reg_min[0]) * H->m - H->reach;
            by0 = (cxy % H->dims[1] + H->reg_min[1]) * H->m - H->reach;
            bz0 = (cxy / H->dims[1] + H->reg_min[2]) * H->m - H->reach;
        } else {
            rows = p.overflow;
            s0 = b - n_blocks;
            s1 = s0 + 1;
            const int64_t j = rows[s0];
            bx0 = (int)floorf((p.inp_pos[3 * j] - ox0) * inv_voxel) - H->reach;
            by0 = (int)floorf((p.inp_pos[3 * j + 1] - oy0) * inv_voxel) - H->reach;
            bz0 = (int)floorf((p.inp_pos[3 * j + 2] - oz0) * inv_voxel) - H->reach;
        }
        f32x4 av[MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) load_a(rows, s0 + 16 * mt, s1, av[mt]);
        int buf = 0;
        for (int c0 = s0; c0 < s1; c0 += CH, buf ^= 1) {
            float* Gb = Gc + buf * CH * GROW;
            // ---- G of the chunk's 16 rows on the matrix cores, this wave's NTW tiles
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4 acc[NTW];
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt) acc[tt] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (16 * j < cin) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int tt = 0; tt < NTW; ++tt)
                                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][j][i], wf[j][i][tt], acc[tt], 0, 0, 0);
                    }
                }
                // D layout: lane (r16, q4) holds rows 4 q4 + rr, column 16 t + r16
#pragma unroll
                for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) Gb[(16 * mt + 4 * q4 + rr) * GROW + 16 * (wave * NTW + tt) + r16] = acc[tt][rr];
            }
            // this wave's two rows of the chunk (wave uniform), and the next chunk's operand on its way
            int64_t rbs[2];
            int cnts[2];
            float pxs[2], pys[2], pzs[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rp = c0 + wave + kSWaves * h;
                rbs[h] = 0; cnts[h] = 0; pxs[h] = pys[h] = pzs[h] = 0.0f;
                if (rp < s1) {
                    const int64_t j = __builtin_amdgcn_readfirstlane(rows[rp]);
                    pxs[h] = p.inp_pos[3 * j]; pys[h] = p.inp_pos[3 * j + 1]; pzs[h] = p.inp_pos[3 * j + 2];
                    rbs[h] = p.t_rs[j];
                    int cnt = p.t_cnt ? p.t_cnt[j] : (int)(p.t_rs[j + 1] - rbs[h]);
                    if (rbs[h] + cnt > p.t_cap) cnt = 0;  // (a row past the buffer: the search skipped it and the caller repeats the step)
                    cnts[h] = cnt;
                }
            }
            if (c0 + CH < s1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) load_a(rows, c0 + CH + 16 * mt, s1, av[mt]);
            }
            const int nb0 = (cnts[0] + 63) >> 6, nb = nb0 + ((cnts[1] + 63) >> 6);
            // the two rows as ONE stream of 64-pair batches: indices two batches ahead, output positions one
            // (loads are UNCONDITIONAL -- past a row's end the clamped address returns some entry, discarded by `ok` -- so that the
            // compiler can count them: behind a branch it waits for the newest load at the top of every batch)
            auto idx_ok = [&](int t, int64_t& e) -> bool {
                const int h = t >= nb0;
                const int k = 64 * (t - (h ? nb0 : 0)) + lane;
                e = min(max(rbs[h] + k, (int64_t)0), p.t_cap - 1);
                return t < nb && k < cnts[h];
            };
            int64_t e0, e1;
            const bool ok0 = idx_ok(0, e0), ok1 = idx_ok(1, e1);
            int iA = p.t_idx[e0], iB = p.t_idx[e1];
            iA = ok0 ? iA : -1;
            iB = ok1 ? iB : -1;
            float qxA, qyA, qzA;
            {
                const int64_t g = iA < 0 ? 0 : iA;
                qxA = p.out_pos[3 * g]; qyA = p.out_pos[3 * g + 1]; qzA = p.out_pos[3 * g + 2];
            }
            asm volatile("" : "+v"(iA), "+v"(iB), "+v"(qxA), "+v"(qyA), "+v"(qzA));  // (nothing in flight when the loop starts)
            __syncthreads();  // G of this chunk is complete (and every wave is done with the chunk before the last: its buffer is free)
            for (int t = 0; t < nb; ++t) {
                int64_t eC;
                const bool okC = idx_ok(t + 2, eC);
                int iC = p.t_idx[eC];
                const int64_t gB = iB < 0 ? 0 : iB;
                float qxB = p.out_pos[3 * gB], qyB = p.out_pos[3 * gB + 1], qzB = p.out_pos[3 * gB + 2];
                const int h = t >= nb0;
                const float* Gw = Gb + (wave + kSWaves * h) * GROW;
                const bool valid = iA >= 0;
                float x = pxs[h] - qxA, y = pys[h] - qyA, z = pzs[h] - qzA;
                float a = p.window == DMCF_WINDOW_NONE ? 1.0f : window_value(DMCF_WINDOW_POLY6, rel_dist2(x, y, z), p.inv_r2, p.window_fac);
                a = valid ? a * S : 0.0f;
                filter_coords<false>(x, y, z, gp);
                int bx, by, bz;
                float wx0, wx1, wy0, wy1, wz0, wz1;
                axis_weights_linear(x, 4, bx, wx0, wx1);
                axis_weights_linear(y, 4, by, wy0, wy1);
                axis_weights_linear(z, 4, bz, wz0, wz1);
#ifdef SX_NOGATHER
                const float* gc = Gw + (lane & 1) * COUT;
#else
                const float* gc = Gw + ((bz * 4 + by) * 4 + bx) * COUT;
#endif
                // corner weights in Open3D's product order (x-weight * y-weight) * z-weight, times the window (and 2^s)
                // (packed arithmetic: v_pk_mul_f32 / v_pk_fma_f32, two channels per instruction)
                const f32x2 wxy0 = (f32x2){wx0, wx1} * (f32x2){wy0, wy0}, wxy1 = (f32x2){wx0, wx1} * (f32x2){wy1, wy1};
                f32x2 acc2[COUT / 2];
#pragma unroll
                for (int o = 0; o < COUT / 2; ++o) acc2[o] = (f32x2){0.0f, 0.0f};
#pragma unroll
                for (int zz = 0; zz < 2; ++zz) {
                    const float wz = (zz ? wz1 : wz0) * a;
#pragma unroll
                    for (int yy = 0; yy < 2; ++yy) {
                        const f32x2 wab = (yy ? wxy1 : wxy0) * (f32x2){wz, wz};
                        const f32x2 wa = (f32x2){wab.x, wab.x}, wb = (f32x2){wab.y, wab.y};
                        const float* gq = gc + (zz * 16 + yy * 4) * COUT;
#pragma unroll
                        for (int q = 0; q < COUT; q += 4) {
                            const f32x4 ga = *(const f32x4*)(gq + q), gb = *(const f32x4*)(gq + COUT + q);
                            acc2[q / 2] = __builtin_elementwise_fma(wa, (f32x2){ga.x, ga.y}, __builtin_elementwise_fma(wb, (f32x2){gb.x, gb.y}, acc2[q / 2]));
                            acc2[q / 2 + 1] = __builtin_elementwise_fma(wa, (f32x2){ga.z, ga.w}, __builtin_elementwise_fma(wb, (f32x2){gb.z, gb.w}, acc2[q / 2 + 1]));
                        }
                    }
                }
                float acc[COUT];
#pragma unroll
                for (int o = 0; o < COUT / 2; ++o) { acc[2 * o] = acc2[o].x; acc[2 * o + 1] = acc2[o].y; }
                // the output point's slot in the block's box
                const int cx = (int)rintf((qxA - ox0) * inv_voxel) - bx0, cy = (int)rintf((qyA - oy0) * inv_voxel) - by0,
                          cz = (int)rintf((qzA - oz0) * inv_voxel) - bz0;
                const bool inside = (unsigned)cx < (unsigned)D && (unsigned)cy < (unsigned)D && (unsigned)cz < (unsigned)D;
                if (valid && !inside) *p.err = 1;
                if (valid && inside) {
                    const int slot = (cz * D + cy) * D + cx;
                    Sidx[slot] = iA;
#ifndef SX_NOATOM
#pragma unroll
                    for (int o = 0; o < COUT; ++o) lds_add_i64(Acc + o * NSp + slot, sct_fixed(acc[o]));
#else
                    if (acc[0] + acc[COUT - 1] == 123.456f) Acc[slot] = 1;
#endif
                }
                // (the loads issued at the top of this iteration land in their own registers and move to the loop-carried ones HERE,
                // behind the batch's arithmetic: a copy the compiler places earlier waits for every load in flight -- the counter
                // is in order -- and the stream stops being pipelined; cconv_pair.hip has the same pin)
                asm volatile("" : "+v"(iC), "+v"(qxB), "+v"(qyB), "+v"(qzB) : : "memory");
                iA = iB; qxA = qxB; qyA = qyB; qzA = qzB;
                iB = okC ? iC : -1;
            }
        }
        __syncthreads();
        // ---- flush: every touched slot once, and clear for the next block
        for (int s = tid; s < NS; s += kSThreads) {
            const int i = Sidx[s];
            if (i >= 0) {
                Sidx[s] = -1;
#pragma unroll
                for (int o = 0; o < COUT; ++o) {
                    const unsigned long long v = Acc[o * NSp + s];
                    Acc[o * NSp + s] = 0ull;
#ifndef SX_NOFLUSH
                    if (v) __hip_atomic_fetch_add(p.acc + (int64_t)i * COUT + o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
                    if (v == 0x123456789ull) p.acc[0] = v;
#endif
                }
            }
        }
        __syncthreads();
    }
}

// out[i][o] (+)= acc[i][o] * 2^-s + bias[o]
__global__ void cconv_sct_finish(const long long* __restrict__ acc, const uint32_t* __restrict__ bound, const float* __restrict__ bias,
                                 float* __restrict__ out, int64_t n, int cout, int accumulate, float window_fac) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float v = (float)((double)acc[e] / (double)sct_scale(bound, window_fac));
    if (bias) v += bias[e % cout];
    out[e] = accumulate ? out[e] + v : v;
}

static size_t sct_lds_bytes(int cout, int D, int waves) {
    const int NS = D * D * D, NSp = (NS + 3) & ~3;
    return (size_t)cout * NSp * 8 + (size_t)NSp * 4 + (size_t)2 * 2 * waves * 64 * cout * 4 + 16;
}

// 8 waves while two workgroups fit a CU, else 16 waves in one (cout 4) -- 0: the box does not fit at all
static int sct_waves(int cout, int D) {
    if (2 * sct_lds_bytes(cout, D, 8) <= 160 * 1024) return 8;
    if (cout == 4 && sct_lds_bytes(cout, D, 16) <= 160 * 1024) return 16;
    if (sct_lds_bytes(cout, D, 8) <= 160 * 1024) return 8;
    return 0;
}

}  // namespace dmcf

using namespace dmcf;

static int sct_check(const dmcf_cconv_scatter_args* a) {
    if (!a || !a->filters || !a->out_positions || !a->inp_positions || !a->inp_features || !a->t_index || !a->t_row_begin || !a->plan || !a->out)
        return DMCF_EINVAL;
    if (a->n_out <= 0 || a->n_inp <= 0 || a->filter_dims[3] <= 0 || a->extent <= 0.0f || a->block_cells <= 0 || a->reach <= 0) return DMCF_EINVAL;
    if (a->filter_dims[0] != 4 || a->filter_dims[1] != 4 || a->filter_dims[2] != 4 || (a->filter_dims[4] != 4 && a->filter_dims[4] != 8) ||
        a->filter_dims[3] > 32)
        return DMCF_EUNSUPPORTED;
    if (a->window != DMCF_WINDOW_NONE && a->window != DMCF_WINDOW_POLY6) return DMCF_EUNSUPPORTED;
    if (a->flags & ~(DMCF_FLAG_ALIGN_CORNERS | DMCF_FLAG_ACCUMULATE)) return DMCF_EUNSUPPORTED;
    if (!(a->flags & DMCF_FLAG_ALIGN_CORNERS)) return DMCF_EUNSUPPORTED;
    const int D = a->block_cells + 2 * a->reach + 1;
    if (sct_waves(a->filter_dims[4], D) == 0) return DMCF_EUNSUPPORTED;
    return DMCF_OK;
}

extern "C" size_t dmcf_cconv_scatter_plan_bytes(int64_t n_inp) { return n_inp > 0 ? sct_plan_layout(n_inp).total : 0; }

extern "C" int dmcf_cconv_scatter_plan(const float* inp_positions, int64_t n_inp, const float* out_positions, int64_t n_out, float voxel,
                                       float extent, int32_t block_cells, void* plan, size_t plan_bytes, void* stream_) {
    if (!inp_positions || !out_positions || !plan || n_inp <= 0 || n_out <= 0 || !(voxel > 0.0f) || !(extent > 0.0f) || block_cells <= 0 ||
        n_inp >= ((int64_t)1 << 31))
        return DMCF_EINVAL;
    const SctPlanLayout L = sct_plan_layout(n_inp);
    if (plan_bytes < L.total) return DMCF_EWORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)plan;
    SctHeader* h = (SctHeader*)(ws + L.off_header);
    uint32_t* cell_start = (uint32_t*)(ws + L.off_cell_start);
    uint32_t* cell_fill = (uint32_t*)(ws + L.off_cell_fill);
    int32_t* key = (int32_t*)(ws + L.off_key);
    const int reach = (int)ceilf(0.5f * extent / voxel - 1e-4f);
    // (the table is cleared whole: 2 x 8 MB at a few TB/s -- the region's cell count is only known on the device)
    if (hipMemsetAsync(cell_fill, 0, (size_t)(kSTable + 1) * 4, stream) != hipSuccess) { check_launch(); return DMCF_ELAUNCH; }
    hipLaunchKernelGGL(sct_plan_init, dim3(1), dim3(64), 0, stream, h, out_positions, voxel, (int)block_cells, reach);
    const int g = (int)std::min<int64_t>((n_inp + 255) / 256, 512);
    hipLaunchKernelGGL(sct_plan_stats, dim3(g), dim3(256), 0, stream, inp_positions, n_inp, h);
    hipLaunchKernelGGL(sct_plan_region, dim3(1), dim3(64), 0, stream, h, n_inp);
    const unsigned gp = (unsigned)((n_inp + 255) / 256);
    // counts go to cell_fill, their exclusive scan to cell_start, then cell_fill is cleared again and serves as the fill cursor
    hipLaunchKernelGGL(sct_plan_key, dim3(gp), dim3(256), 0, stream, inp_positions, n_inp, (const SctHeader*)h, key, cell_fill);
    int rc = check_launch();
    if (rc != DMCF_OK) return rc;
    rc = scan_exclusive_u32(cell_fill, cell_start, kSTable + 1, ws + L.off_scan, L.scan_bytes, stream);
    if (rc != DMCF_OK) return rc;
    if (hipMemsetAsync(cell_fill, 0, (size_t)(kSTable + 1) * 4, stream) != hipSuccess) { check_launch(); return DMCF_ELAUNCH; }
    hipLaunchKernelGGL(sct_plan_scatter, dim3(gp), dim3(256), 0, stream, n_inp, h, (const int32_t*)key, (const uint32_t*)cell_start, cell_fill,
                       (int32_t*)(ws + L.off_sorted), (int32_t*)(ws + L.off_overflow));
    hipLaunchKernelGGL(sct_plan_blocks, dim3((unsigned)((kSTable + 255) / 256)), dim3(256), 0, stream, h, (const uint32_t*)cell_start,
                       (int32_t*)(ws + L.off_blocks));
    return check_launch();
}

extern "C" size_t dmcf_cconv_scatter_workspace_bytes(const dmcf_cconv_scatter_args* a) {
    if (!a || a->n_out <= 0 || a->filter_dims[4] <= 0) return 0;
    return align_up((size_t)a->n_out * a->filter_dims[4] * 8, 256) + 256;
}

extern "C" int dmcf_cconv_scatter_forward(const dmcf_cconv_scatter_args* a, void* workspace, size_t workspace_bytes, void* stream_) {
    const int rc = sct_check(a);
    if (rc != DMCF_OK) return rc;
    if (!workspace || workspace_bytes < dmcf_cconv_scatter_workspace_bytes(a)) return DMCF_EWORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    const size_t acc_bytes = align_up((size_t)a->n_out * cout * 8, 256);
    // tail of the workspace: [0] error flag, [1] block counter, [2..3] the bound's two maxima
    int* tail = (int*)((char*)workspace + acc_bytes);
    if (hipMemsetAsync(workspace, 0, acc_bytes + 256, stream) != hipSuccess) {
        check_launch();
        return DMCF_ELAUNCH;
    }
    const SctPlanLayout L = sct_plan_layout(a->n_inp);
    const char* pl = (const char*)a->plan;
    SctParams p;
    p.W = a->filters;
    p.out_pos = a->out_positions;
    p.inp_pos = a->inp_positions;
    p.inp_feat = a->inp_features;
    p.t_idx = a->t_index;
    p.t_rs = a->t_row_begin;
    p.t_cnt = a->t_row_count;
    p.t_cap = a->t_capacity;
    p.hdr = (const SctHeader*)(pl + L.off_header);
    p.cell_start = (const uint32_t*)(pl + L.off_cell_start);
    p.sorted = (const int32_t*)(pl + L.off_sorted);
    p.blocks = (const int32_t*)(pl + L.off_blocks);
    p.overflow = (const int32_t*)(pl + L.off_overflow);
    p.bound = (const uint32_t*)(tail + 2);
    p.acc = (unsigned long long*)workspace;
    p.n_out = a->n_out;
    p.n_inp = a->n_inp;
    p.cin = cin;
    p.D = a->block_cells + 2 * a->reach + 1;
    p.inv_extent = 1.0f / a->extent;
    const float radius = 0.5f * a->extent;
    p.inv_r2 = 1.0f / (radius * radius);
    p.window_fac = a->window_fac;
    p.window = a->window;
    p.err = a->error_flag ? a->error_flag : tail;
    p.counter = tail + 1;
    hipLaunchKernelGGL(sct_bound_kernel, dim3((unsigned)std::min<int64_t>((a->n_inp + 255) / 256, 1024)), dim3(256), 0, stream, a->inp_features,
                       a->n_inp, cin, a->filters, (int64_t)64 * cin * cout, (uint32_t*)(tail + 2));
    const int waves = sct_waves(cout, p.D);
    const size_t lds = sct_lds_bytes(cout, p.D, waves) - 16;
    const bool two = 2 * (lds + 16) <= 160 * 1024;
    const int grid = (int)std::min<int64_t>((int64_t)device_cu_count() * (two ? 2 : 1), a->n_inp);
    auto launch = [&](auto kernel) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * waves), lds, stream, p);
    };
    if (cout == 4 && waves == 8) launch(cconv_sct_kernel<4, 8>);
    else if (cout == 4) launch(cconv_sct_kernel<4, 16>);
    else launch(cconv_sct_kernel<8, 8>);
    int r = check_launch();
    if (r != DMCF_OK) return r;
    const int64_t n = a->n_out * cout;
    hipLaunchKernelGGL(cconv_sct_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, (const long long*)workspace,
                       (const uint32_t*)(tail + 2), a->bias, a->out, n, cout, (a->flags & DMCF_FLAG_ACCUMULATE) ? 1 : 0, a->window_fac);
    return check_launch();
}
