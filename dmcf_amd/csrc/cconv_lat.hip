// CConv between two point sets that lie on ALIGNED REGULAR LATTICES (the coarse scales of the multi-scale models: both come
// out of grid_pos with the same centre, utils/tools/losses.py:136-181, and the spacing of one is an integer multiple of
// the other's): x_in - x_out only takes the values d * voxel, d an integer vector, and everything ml3d.ops.continuous_conv
// (utils/convolutions.py:414-431) evaluates per neighbour pair -- window, ball -> cube map, trilinear weights -- depends on
// d alone.  So
//
//     out_i = sum_d  W_d^T f_{cell(i) + d},     W_d[c][o] = window(d) * sum_{8 corners t} w_t(d) W[cell_t(d)][c][o]
//
// one [Cin x Cout] matrix per stencil offset (a few hundred to ~2000 offsets inside the radius), no neighbour search, no
// per-pair geometry: a 3-D convolution.  The caller lays the input features out by lattice cell (a dense volume over the
// input lattice's bounding box, zeros in empty cells), so the operand rows of a tile are contiguous memory:
//   lat_build_filters   one thread per element of the per-offset matrices, packed in MFMA B-fragment order
//   lat_rows_*          the launch's output points in lattice order, compacted: (volume offset of the cell, output row) pairs
//   lat_conv_kernel     tile = 16 consecutive ROWS of that list (M of v_mfma_f32_16x16x4_f32; in a filled region these are
//                       16 consecutive cells along x, which is what the tile was until round 5), N = 16 output channels,
//                       K = 4 stencil offsets of one input channel.  Per 4 offsets and tile one 16-byte load per lane and
//                       4 input channels (the stencil slides along x: the rows are re-read from L1); the per-offset
//                       matrices are staged through LDS in chunks and shared by the 4 waves x 2 tiles of a workgroup.  A
//                       first version gathered rows through a cell -> point table in point order: 2.3 ms for the s1 -> s1
//                       layer of the 1M-particle scene, bound by 9.5 GB of scattered 32-byte reads.
// The offsets are nominal (d * voxel in fp32); the reference forms fl(x_in) - fl(x_out), which differs by ~1 ulp of |x|:
// up to 1e-5 of the output scale at |x| ~ 6.  Pairs at exactly the radius have window 0 either way.
#include "cconv_common.h"

namespace dmcf {

struct LatParams {
    const float* Wp;          // [ceil(S / 4)][4 KS][NT][64]
    const int32_t* stencil;   // [S][4]: dx, dy, dz of the input cell relative to out_cell * out_step
    int S, KS, NT, cin, cout;
    const float* vol;          // [idim z][y][x][cin]
    int imin[3], idim[3];      // (x, y, z)
    const int32_t* otab;       // [odim z][y][x]: output point index or -1
    int omin[3], odim[3];
    int inp_step, out_stride, phase[3];
    int amin[3], adim[3];      // box of the base vectors a: output cell a * out_stride + phase, input cell a * inp_step + d
    int tiles_x;               // ceil(adim x / 16)
    int64_t ntiles;
    const float* bias;
    float* out;
    int flags;
};

// The output rows of a launch (or of the parts of a batch), compacted.  Until round 5 a tile was 16 consecutive CELLS of the launch's
// box: a box stretched by stray points (particles that left the scene carry lattice points with them) is mostly tiles with one point
// or none, each walking the whole stencil -- the four lattice layers of the 1M box went 2.6 -> 9.9 ms per step over 20 steps while
// their outputs grew by 30 %.  Now: count the occupied cells per 1024-cell block of the box, scan, write (offset, row) pairs in cell
// order; the convolution walks tiles of 16 consecutive pairs, whatever cells they come from.
typedef int lat_i32x2 __attribute__((ext_vector_type(2)));
constexpr int kLatMaxParts = 8;
constexpr int kLatRowsPerGroup = 128;  // rows of one workgroup (4 waves x kLatTW tiles x 16): a part's rows start at a multiple
constexpr int kLatCellsPerBlock = 1024;
struct LatBatch {
    LatParams part[kLatMaxParts];
    int64_t cfirst[kLatMaxParts + 1];  // first cell block of each part (lat_rows_count / _write)
    int n;
    uint32_t* blk;          // [cfirst[n] + 1]: occupied cells per block, then their exclusive scan
    int64_t* start;         // [n + 1] (device): first row of each part in `rows`, a multiple of kLatRowsPerGroup
    int64_t* count;         // [n] (device): rows of each part
    lat_i32x2* rows;        // (byte offset of the row's cell in the volume at stencil offset 0, output row)
    int64_t rows_capacity;
};

// cell c of a part's box (16-cell runs along x, then y, then z): its output row (or -1) and the volume offset of its cell
__device__ __forceinline__ int lat_cell(const LatParams& p, int64_t c, int& rowb) {
    const int64_t tile = c >> 4;
    const int m = (int)(c & 15);
    rowb = 0;
    if (tile >= p.ntiles) return -1;
    const int xb = (int)(tile % p.tiles_x);
    const int ax = p.amin[0] + xb * 16 + m, ay = p.amin[1] + (int)(tile / p.tiles_x % p.adim[1]),
              az = p.amin[2] + (int)(tile / ((int64_t)p.tiles_x * p.adim[1]));
    const int ox = ax * p.out_stride + p.phase[0] - p.omin[0], oy = ay * p.out_stride + p.phase[1] - p.omin[1],
              oz = az * p.out_stride + p.phase[2] - p.omin[2];
    if (!(xb * 16 + m < p.adim[0] && (unsigned)ox < (unsigned)p.odim[0] && (unsigned)oy < (unsigned)p.odim[1] &&
          (unsigned)oz < (unsigned)p.odim[2]))
        return -1;
    const int ix = ax * p.inp_step - p.imin[0], iy = ay * p.inp_step - p.imin[1], iz = az * p.inp_step - p.imin[2];
    rowb = ((iz * p.idim[1] + iy) * p.idim[0] + ix) * p.cin * 4;
    return p.otab[((int64_t)oz * p.odim[1] + oy) * p.odim[0] + ox];
}

__device__ __forceinline__ int lat_part_of_block(const LatBatch& b, int64_t block) {
    int i = 0;
    while (i + 1 < b.n && block >= b.cfirst[i + 1]) ++i;
    return i;
}

__global__ __launch_bounds__(256) void lat_rows_count(const LatBatch b) {
    __shared__ uint32_t wsum[4];
    const int i = lat_part_of_block(b, blockIdx.x);
    const int64_t c0 = ((int64_t)blockIdx.x - b.cfirst[i]) * kLatCellsPerBlock;
    uint32_t mine = 0;
    for (int r = 0; r < kLatCellsPerBlock / 256; ++r) {
        int rowb;
        mine += (uint32_t)__popcll(__ballot(lat_cell(b.part[i], c0 + r * 256 + threadIdx.x, rowb) >= 0));
    }
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;  // (every lane of a wave holds the wave's count)
    __syncthreads();
    if (threadIdx.x == 0) {
        b.blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (blockIdx.x == 0) b.blk[b.cfirst[b.n]] = 0;  // the scan's last entry = the total
    }
}

// (after the exclusive scan of blk) where each part's rows begin, and how many it has
__global__ void lat_rows_starts(const LatBatch b) {
    if (threadIdx.x != 0) return;
    int64_t s = 0;
    for (int i = 0; i < b.n; ++i) {
        const int64_t n = (int64_t)b.blk[b.cfirst[i + 1]] - (int64_t)b.blk[b.cfirst[i]];
        b.start[i] = s;
        b.count[i] = n;
        s += (n + kLatRowsPerGroup - 1) / kLatRowsPerGroup * kLatRowsPerGroup;
    }
    b.start[b.n] = s;
}

__global__ __launch_bounds__(256) void lat_rows_write(const LatBatch b) {
    __shared__ uint32_t wsum[4];
    const int i = lat_part_of_block(b, blockIdx.x);
    const int64_t c0 = ((int64_t)blockIdx.x - b.cfirst[i]) * kLatCellsPerBlock;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t at = b.start[i] + ((int64_t)b.blk[blockIdx.x] - (int64_t)b.blk[b.cfirst[i]]);
    for (int r = 0; r < kLatCellsPerBlock / 256; ++r) {
        int rowb;
        const int oi = lat_cell(b.part[i], c0 + r * 256 + threadIdx.x, rowb);
        const uint64_t mk = __ballot(oi >= 0);
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(mk);
        __syncthreads();
        if (oi >= 0) {
            int64_t pos = at + __popcll(mk & ((1ull << lane) - 1ull));
            for (int v = 0; v < wave; ++v) pos += wsum[v];
            if (pos < b.rows_capacity) b.rows[pos] = (lat_i32x2){rowb, oi};
        }
        at += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void lat_build_filters(const float* __restrict__ W, float* __restrict__ Wp,
                                                         const int32_t* __restrict__ stencil, int S, int KS, int NT, CconvParams p,
                                                         float vx, float vy, float vz, float sx, float sy, float sz) {
    // [group of 4 offsets][channel c < 4 KS][NT][lane = (q, n)]: the B operand of the product for channel c of the offsets
    // 4 g .. 4 g + 3 (lane (q, n) holds W_{4g+q}[c][16 nt + n]); offsets past S are zero matrices
    const int64_t total = (int64_t)((S + 3) / 4) * 4 * KS * NT * 64;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = e;
        const int n = (int)(t & 15); t >>= 4;
        const int q = (int)(t & 3); t >>= 2;
        const int nt = (int)(t % NT); t /= NT;
        const int c = (int)(t % (4 * KS)); t /= 4 * KS;
        const int s = (int)t * 4 + q, o = nt * 16 + n;
        float v = 0.0f;
        if (s < S && c < p.cin && o < p.cout) {
            float x = (float)stencil[4 * s] * vx - sx, y = (float)stencil[4 * s + 1] * vy - sy, z = (float)stencil[4 * s + 2] * vz - sz;
            const float d2 = (x * x + y * y) + z * z;
            const float a = window_value(p.window, d2, p.inv_r2, p.window_fac);
            filter_coords<true>(x, y, z, p);
            int bx, by, bz;
            float wx[2], wy[2], wz[2];
            axis_weights(x, p.sx, p.interp, bx, wx[0], wx[1]);
            axis_weights(y, p.sy, p.interp, by, wy[0], wy[1]);
            axis_weights(z, p.sz, p.interp, bz, wz[0], wz[1]);
            for (int iz = 0; iz < 2; ++iz)
                for (int iy = 0; iy < 2; ++iy)
                    for (int ix = 0; ix < 2; ++ix) {
                        const float w = wz[iz] * wy[iy] * wx[ix];
                        if (w == 0.0f) continue;  // also the "+1" cells that do not exist on size-1 axes
                        const int cz = min(bz + iz, p.sz - 1), cy = min(by + iy, p.sy - 1), cx = min(bx + ix, p.sx - 1);
                        v += w * W[((((int64_t)cz * p.sy + cy) * p.sx + cx) * p.cin + c) * p.cout + o];
                    }
            v *= a;
        }
        Wp[e] = v;
    }
}

constexpr int kLatTW = 2;   // 16-cell tiles per wave
constexpr int kLatCH = 32;  // stencil offsets per LDS chunk of the per-offset matrices
constexpr int kLatG = kLatCH / 4;
static_assert(kLatRowsPerGroup == 4 * kLatTW * 16, "rows of a workgroup");

// Workgroup = 4 waves x kLatTW tiles.  Every wave walks the whole stencil for its own tiles (no cross-wave reduction); the
// per-offset matrices are staged through LDS in chunks of kLatCH offsets and shared by the waves.
// One v_mfma_f32_16x16x4_f32 multiplies the 16 cells of a tile with FOUR STENCIL OFFSETS of one input channel (k = offset):
// lane (m, q) loads all channels of cell m at offset 4 g + q with one 16-byte load per 4 channels, and register c of that
// load is the A operand of the product for channel c.  (The first version had k = 4 channels of one offset: one 4-byte
// load per lane, tile and offset -- four times the load instructions for the same bytes, and the coarse -> fine layer ran at
// 4x its matrix time, bound by the rate of the address unit.)
template <int NTT, int KST>
__device__ __forceinline__ void lat_conv_body(const LatParams& p, const lat_i32x2* __restrict__ rows, const int64_t first, const int64_t end) {
    __shared__ __attribute__((aligned(16))) float Ws[2][kLatCH * KST * NTT * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, q = lane >> 4;
    constexpr int C = 4 * KST;
    int oidx[kLatTW];       // output point of this lane's row (lanes of one row agree), -1: none, -2: tile without points
    int rowb[kLatTW];       // byte offset of the row's cell in the volume for offset (0, 0, 0)
    f32x4 acc[kLatTW][NTT];
    bool any = false;
#pragma unroll
    for (int t = 0; t < kLatTW; ++t) {
        const int64_t e = first + ((wave * kLatTW + t) << 4) + m;
        oidx[t] = -1;
        // (rows past the end read the cells of the launch's first cell and write nothing)
        rowb[t] = (((p.amin[2] * p.inp_step - p.imin[2]) * p.idim[1] + (p.amin[1] * p.inp_step - p.imin[1])) * p.idim[0] +
                   (p.amin[0] * p.inp_step - p.imin[0])) * p.cin * 4;
        if (e < end) {
            const lat_i32x2 rw = rows[e];
            rowb[t] = rw.x;
            oidx[t] = rw.y;
        }
        if (__ballot(oidx[t] >= 0) == 0) oidx[t] = -2;  // nothing to compute in this tile (the whole wave agrees)
        any |= oidx[t] != -2;
#pragma unroll
        for (int n = 0; n < NTT; ++n) acc[t][n] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    }
    if (!__syncthreads_or(any ? 1 : 0)) return;  // (the padding at the end of the launch)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const int NG = (p.S + 3) / 4;
    // The matrices of chunk k + 1 are fetched into registers before the products of chunk k and stored to the other LDS
    // buffer after them: one barrier per chunk, and the L2 latency of the fetch hides behind the matrix instructions.
    constexpr int kQuads = kLatCH * KST * NTT * 64 / 4 / 256;  // 16-byte pieces of a chunk per thread
    f32x4 wr[kQuads];
    i32x4 dv;
    auto fetch = [&](int s0) {
        const int g0 = s0 / 4, ng = min(kLatG, NG - g0);
#pragma unroll
        for (int i = 0; i < kQuads; ++i) {
            // chunk layout [group][c][n < NTT][64]; the packed array has NT (<= NTT) tiles per channel
            const int e = (threadIdx.x + 256 * i) * 4;
            const int l = e & 63, n = (e >> 6) % NTT, gc = (e >> 6) / NTT;
            wr[i] = (gc < ng * C && n < p.NT) ? *(const f32x4*)(p.Wp + (((int64_t)g0 * C + gc) * p.NT + n) * 64 + l)
                                              : (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
        }
        // stencil offset s0 + l in lane l < kLatCH (past S: the last offset, whose matrix is zero)
        dv = *(const i32x4*)(p.stencil + 4 * min(s0 + (lane & (kLatCH - 1)), p.S - 1));
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < kQuads; ++i) *(f32x4*)(Ws[buf] + (threadIdx.x + 256 * i) * 4) = wr[i];
    };
    fetch(0);
    stash(0);
    __syncthreads();
    for (int s0 = 0, k = 0; s0 < p.S; s0 += kLatCH, ++k) {
        const int ng = min(kLatG, NG - s0 / 4);
        const float* Wc = Ws[k & 1];
        // byte offset of the cell at this lane's stencil offset; the volume is padded so that every row + offset is inside
        // it (checked on the host): no bounds tests
        const int dof = ((dv.z * p.idim[1] + dv.y) * p.idim[0] + dv.x) * p.cin * 4;
        int dofg[kLatG];  // ... and of this lane's offset 4 g + q of each group
#pragma unroll
        for (int g = 0; g < kLatG; ++g) dofg[g] = __shfl(dof, 4 * g + q, 64);
        const bool more = s0 + kLatCH < p.S;
        if (more) fetch(s0 + kLatCH);
        if (any) {
            f32x4 f[3][kLatTW][KST];
            auto gather = [&](int g, f32x4 (&fv)[kLatTW][KST]) {
#pragma unroll
                for (int t = 0; t < kLatTW; ++t) {
                    const f32x4* src = (const f32x4*)((const char*)p.vol + (size_t)(uint32_t)(rowb[t] + dofg[g]));
#pragma unroll
                    for (int ks = 0; ks < KST; ++ks) fv[t][ks] = src[ks];
                }
            };
            auto products = [&](int g, const f32x4 (&fv)[kLatTW][KST]) {
#pragma unroll
                for (int c = 0; c < C; ++c)
#pragma unroll
                    for (int n = 0; n < NTT; ++n) {
                        const float w = Wc[((g * C + c) * NTT + n) * 64 + lane];
#pragma unroll
                        for (int t = 0; t < kLatTW; ++t)
                            acc[t][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[t][c >> 2][c & 3], w, acc[t][n], 0, 0, 0);
                    }
            };
            // software pipeline, two groups of loads ahead of the products (groups past ng repeat a valid offset, unused)
            gather(0, f[0]);
            gather(1, f[1]);
#pragma unroll
            for (int g = 0; g < kLatG; ++g) {
                if (g >= ng) break;
                if (g + 2 < kLatG) gather(g + 2, f[(g + 2) % 3]);
                products(g, f[g % 3]);
            }
        }
        if (more) stash((k + 1) & 1);
        __syncthreads();
    }
    // D layout: lane (rows 4 (lane >> 4) + r, column lane & 15); the row's output point sits in lane 4 q + r of oidx
#pragma unroll
    for (int t = 0; t < kLatTW; ++t) {
        if (oidx[t] == -2) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oi = __shfl(oidx[t], 4 * q + r, 64);
            if (oi < 0) continue;
#pragma unroll
            for (int n = 0; n < NTT; ++n) {
                const int o = 16 * n + m;
                if (o >= p.cout) continue;
                float v = acc[t][n][r];
                if (p.bias) v += p.bias[o];
                float* dst = p.out + (int64_t)oi * p.cout + o;
                if (p.flags & DMCF_FLAG_ACCUMULATE) v += *dst;
                *dst = v;
            }
        }
    }
}

// One grid for all parts of a batch (a single launch is a batch of one; the eight parity classes of a coarse -> fine layer share
// volume, filter shape and output, and each alone is too small to fill the chip): workgroup g serves rows [128 g, 128 g + 128) of
// the list, which belong to ONE part -- the parts' first rows are multiples of 128.
template <int NTT, int KST>
__global__ __launch_bounds__(256) void lat_conv_kernel(const LatBatch b) {
    const int64_t first = (int64_t)blockIdx.x * kLatRowsPerGroup;
    if (first >= b.start[b.n]) return;
    int i = 0;
    while (i + 1 < b.n && first >= b.start[i + 1]) ++i;
    lat_conv_body<NTT, KST>(b.part[i], b.rows, first, b.start[i] + b.count[i]);
}

static size_t lat_packed_floats(const dmcf_lattice_conv_args* a) {
    const int KS = (a->filter_dims[3] + 3) / 4, NT = (a->filter_dims[4] + 15) / 16;
    return (size_t)((a->n_offsets + 3) / 4) * 4 * KS * NT * 64;
}

static int lat_validate(const dmcf_lattice_conv_args* a) {
    if (!a) return DMCF_EINVAL;
    for (int k = 0; k < 5; ++k)
        if (a->filter_dims[k] <= 0) return DMCF_EINVAL;
    if (a->n_out < 0 || a->n_offsets < 0 || a->inp_step <= 0 || a->out_stride <= 0 || !(a->extent > 0.0f)) return DMCF_EINVAL;
    if (a->n_out > 0 && (!a->filters || !a->out_table || !a->out || !a->inp_volume || (a->n_offsets > 0 && !a->offsets)))
        return DMCF_EINVAL;
    for (int k = 0; k < 3; ++k)
        if (a->inp_dims[k] <= 0 || a->out_dims[k] <= 0 || a->base_dims[k] <= 0 || !(a->voxel[k] >= 0.0f)) return DMCF_EINVAL;
    if (a->flags & (DMCF_FLAG_SYMMETRIC | DMCF_FLAG_NORMALIZE)) return DMCF_EUNSUPPORTED;
    if (a->window == DMCF_WINDOW_EXPLICIT) return DMCF_EUNSUPPORTED;
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    if ((cin != 4 && cin != 8) || cout > 32) return DMCF_EUNSUPPORTED;
    return DMCF_OK;
}

}  // namespace dmcf

namespace dmcf {

// validates one launch, enqueues its filter build into `packed`, fills the kernel parameters
static int lat_prepare(const dmcf_lattice_conv_args* a, float* packed, hipStream_t stream, LatParams& p, int64_t& groups) {
    const int cin = a->filter_dims[3], cout = a->filter_dims[4];
    const int KS = (cin + 3) / 4, NT = (cout + 15) / 16;
    for (int k = 0; k < 3; ++k) {  // every cell a * inp_step + d the launch can touch lies inside the volume
        const int64_t ext = k == 0 ? (int64_t)((a->base_dims[0] + 15) / 16) * 16 : a->base_dims[k];
        const int64_t lo = (int64_t)a->base_min[k] * a->inp_step - a->reach[k];
        const int64_t hi = ((int64_t)a->base_min[k] + ext - 1) * a->inp_step + a->reach[k];
        if (a->reach[k] < 0 || lo < a->inp_min[k] || hi > (int64_t)a->inp_min[k] + a->inp_dims[k] - 1) return DMCF_EINVAL;
    }
    if ((int64_t)a->inp_dims[0] * a->inp_dims[1] * a->inp_dims[2] * cin > 0x1fffffff) return DMCF_EUNSUPPORTED;  // 32-bit byte offsets
    CconvParams cp = {};
    cp.sz = a->filter_dims[0]; cp.sy = a->filter_dims[1]; cp.sx = a->filter_dims[2];
    cp.K = cp.sx * cp.sy * cp.sz;
    cp.cin = cin; cp.cout = cout;
    cp.inv_extent = 1.0f / a->extent;
    const float radius = 0.5f * a->extent;
    cp.inv_r2 = 1.0f / (radius * radius);
    cp.window_fac = a->window_fac;
    cp.window = a->window;
    cp.mapping = a->coordinate_mapping;
    cp.interp = a->interpolation;
    cp.flags = a->flags;
    if (a->n_offsets > 0) {
        const int64_t total = (int64_t)lat_packed_floats(a);
        const unsigned g = (unsigned)((total + 255) / 256);
        hipLaunchKernelGGL(lat_build_filters, dim3(g < 4096u ? g : 4096u), dim3(256), 0, stream, a->filters, packed, a->offsets,
                           (int)a->n_offsets, KS, NT, cp, a->voxel[0], a->voxel[1], a->voxel[2], a->rel_shift[0], a->rel_shift[1],
                           a->rel_shift[2]);
    }
    p.Wp = packed;
    p.stencil = a->offsets;
    p.S = (int)a->n_offsets; p.KS = KS; p.NT = NT; p.cin = cin; p.cout = cout;
    p.vol = a->inp_volume;
    p.otab = a->out_table;
    for (int k = 0; k < 3; ++k) {
        p.imin[k] = a->inp_min[k]; p.idim[k] = a->inp_dims[k];
        p.omin[k] = a->out_min[k]; p.odim[k] = a->out_dims[k];
        p.phase[k] = a->out_phase[k];
        p.amin[k] = a->base_min[k];
        p.adim[k] = a->base_dims[k];
    }
    p.inp_step = a->inp_step;
    p.out_stride = a->out_stride;
    p.tiles_x = (a->base_dims[0] + 15) / 16;
    p.ntiles = (int64_t)p.tiles_x * a->base_dims[1] * a->base_dims[2];
    p.bias = a->bias; p.out = a->out; p.flags = a->flags;
    groups = (p.ntiles * 16 + kLatCellsPerBlock - 1) / kLatCellsPerBlock;  // cell blocks of the compaction
    return DMCF_OK;
}

// workspace of a batch: [packed matrices of every part][blk in][blk scanned][scan tmp][start, count][rows]
struct LatLayout {
    size_t off_blk, off_scan, off_tmp, tmp_bytes, off_start, off_rows, total;
    int64_t nblk, rows_capacity;
};

static LatLayout lat_layout(const dmcf_lattice_conv_args* parts, int n_parts) {
    LatLayout L;
    size_t off = 256;
    L.nblk = 0;
    for (int i = 0; i < n_parts; ++i) {
        off += align_up(lat_packed_floats(parts + i) * sizeof(float), 256);
        const int64_t cells = (int64_t)((parts[i].base_dims[0] + 15) / 16) * 16 * parts[i].base_dims[1] * parts[i].base_dims[2];
        L.nblk += (cells + kLatCellsPerBlock - 1) / kLatCellsPerBlock;
    }
    L.off_blk = off;    off += align_up((size_t)(L.nblk + 1) * 4, 256);
    L.off_scan = off;   off += align_up((size_t)(L.nblk + 1) * 4, 256);
    L.tmp_bytes = scan_tmp_bytes(L.nblk + 1);
    L.off_tmp = off;    off += align_up(L.tmp_bytes, 256);
    L.off_start = off;  off += 256;
    // every output point is a row of at most one part; each part's rows start at a multiple of 128
    L.rows_capacity = (parts[0].n_out + kLatRowsPerGroup - 1) / kLatRowsPerGroup * kLatRowsPerGroup + (int64_t)kLatRowsPerGroup * n_parts;
    L.off_rows = off;   off += align_up((size_t)L.rows_capacity * sizeof(lat_i32x2), 256);
    L.total = off;
    return L;
}

static int lat_launch(const dmcf_lattice_conv_args* parts, int n_parts, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    for (int i = 0; i < n_parts; ++i) {
        const int rc = lat_validate(parts + i);
        if (rc != DMCF_OK) return rc;
        if (parts[i].filter_dims[3] != parts[0].filter_dims[3] || parts[i].filter_dims[4] != parts[0].filter_dims[4] ||
            parts[i].n_out != parts[0].n_out || parts[i].out != parts[0].out)
            return DMCF_EINVAL;  // one kernel instantiation and one output for the whole grid
    }
    if (parts[0].n_out == 0) return DMCF_OK;
    if (!workspace || ((uintptr_t)workspace & 255)) return DMCF_EINVAL;
    const LatLayout L = lat_layout(parts, n_parts);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    if (L.nblk > 0x7ffffffe) return DMCF_EUNSUPPORTED;
    LatBatch b;
    b.n = n_parts;
    char* ws = (char*)workspace;
    char* wp = ws;
    int64_t first = 0;
    for (int i = 0; i < n_parts; ++i) {
        int64_t blocks;
        const int rc = lat_prepare(parts + i, (float*)wp, stream, b.part[i], blocks);
        if (rc != DMCF_OK) return rc;
        wp += align_up(lat_packed_floats(parts + i) * sizeof(float), 256);
        b.cfirst[i] = first;
        first += blocks;
    }
    for (int i = n_parts; i <= kLatMaxParts; ++i) b.cfirst[i] = first;
    if (first != L.nblk) return DMCF_EINVAL;
    uint32_t* blk_in = (uint32_t*)(ws + L.off_blk);
    b.blk = blk_in;
    b.start = (int64_t*)(ws + L.off_start);
    b.count = b.start + kLatMaxParts + 1;
    b.rows = (lat_i32x2*)(ws + L.off_rows);
    b.rows_capacity = L.rows_capacity;
    hipLaunchKernelGGL(lat_rows_count, dim3((unsigned)L.nblk), dim3(256), 0, stream, b);
    const int rc = scan_exclusive_u32(blk_in, (uint32_t*)(ws + L.off_scan), L.nblk + 1, ws + L.off_tmp, L.tmp_bytes, stream);
    if (rc != DMCF_OK) return rc;
    b.blk = (uint32_t*)(ws + L.off_scan);
    hipLaunchKernelGGL(lat_rows_starts, dim3(1), dim3(64), 0, stream, b);
    hipLaunchKernelGGL(lat_rows_write, dim3((unsigned)L.nblk), dim3(256), 0, stream, b);
    const dim3 grid((unsigned)(L.rows_capacity / kLatRowsPerGroup)), block(256);
    const int KS = b.part[0].KS, NT = b.part[0].NT;
    if (KS == 1 && NT == 1) hipLaunchKernelGGL((lat_conv_kernel<1, 1>), grid, block, 0, stream, b);
    else if (KS == 1) hipLaunchKernelGGL((lat_conv_kernel<2, 1>), grid, block, 0, stream, b);
    else if (NT == 1) hipLaunchKernelGGL((lat_conv_kernel<1, 2>), grid, block, 0, stream, b);
    else hipLaunchKernelGGL((lat_conv_kernel<2, 2>), grid, block, 0, stream, b);
    return check_launch();
}

}  // namespace dmcf

using namespace dmcf;

extern "C" {

size_t dmcf_lattice_conv_workspace_bytes(const dmcf_lattice_conv_args* a) {
    if (lat_validate(a) != DMCF_OK) return 256;
    return lat_layout(a, 1).total;
}

size_t dmcf_lattice_conv_batch_workspace_bytes(const dmcf_lattice_conv_args* parts, int32_t n_parts) {
    if (!parts || n_parts < 1 || n_parts > kLatMaxParts) return 256;
    for (int i = 0; i < n_parts; ++i)
        if (lat_validate(parts + i) != DMCF_OK) return 256;
    return lat_layout(parts, n_parts).total;
}

int dmcf_lattice_conv_forward(const dmcf_lattice_conv_args* a, void* workspace, size_t workspace_bytes, dmcf_stream_t stream_) {
    if (!a) return DMCF_EINVAL;
    return lat_launch(a, 1, workspace, workspace_bytes, (hipStream_t)stream_);
}

int dmcf_lattice_conv_forward_batch(const dmcf_lattice_conv_args* parts, int32_t n_parts, void* workspace, size_t workspace_bytes,
                                    dmcf_stream_t stream_) {
    if (!parts || n_parts < 1 || n_parts > kLatMaxParts) return DMCF_EINVAL;
    return lat_launch(parts, n_parts, workspace, workspace_bytes, (hipStream_t)stream_);
}

}  // extern "C"
