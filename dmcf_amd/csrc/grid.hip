// grid_pos (utils/tools/losses.py:136-181): the lattice points of every voxel corner touched by a particle, in the
// order tf.unique gives them (first appearance in the candidate list).  The reference builds the 16 N candidate cell
// indices, runs tf.unique on them and decodes; the torch form of that (sort + unique + scatter-min + argsort) costs
// ~5 ms per call at 1M particles.  Here:
//
//   bounds : centre (optional mean of the positions, deterministic two-stage double sum), extrema of floor(pos/vs -+ h)
//            -> header {minp, dims, cells} on the device; the host reads it once to size the dense cell table
//   count  : atomicMin(first[cell], k) over all candidates k = row * n_off + off (row < N: floor(s - h), row >= N:
//            floor(s + h); off in meshgrid 'ij' order); then per row the number of candidates that own their cell
//            (first[cell] == k) and an exclusive scan of those counts
//   write  : the owners, in candidate order, decoded to positions exactly as :172-179
//
// A row of the second half whose cell equals the first half's can never own a cell (same cells, larger k) and is
// skipped.  All integer work: the result is bit-identical to the sort-based formulation.
#include "common.h"

namespace dmcf {

struct GridHeader {  // 64 bytes, written by the device
    int32_t minp[3];
    int32_t dims[3];
    int64_t cells;   // dims product (0 when n == 0)
    int64_t total;   // number of lattice points (valid after count)
    float center[3];
    int32_t pad_;
};

struct GridParams {
    const float* pos;
    int64_t n;
    float vs[3];     // clamped voxel size
    float h[3];      // hysteresis per axis (0 on collapsed axes)
    int lo[3], len[3];  // offset range per axis: lo .. lo + len - 1
    int centralize;
    float voxel[3];  // unclamped voxel size (output scaling)
};

constexpr int kGridBlocks = 512;

__device__ __forceinline__ void grid_scaled(const GridParams& p, const GridHeader* h, int64_t i, float (&s)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float v = p.pos[3 * i + a];
        if (p.centralize) v = __fsub_rn(v, h->center[a]);
        s[a] = __fdiv_rn(v, p.vs[a]);
    }
}

// stage 1 of the mean: per-block double sums
__global__ __launch_bounds__(256) void grid_sum(const float* __restrict__ pos, int64_t n, double* __restrict__ part) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        s[0] += (double)pos[3 * i];
        s[1] += (double)pos[3 * i + 1];
        s[2] += (double)pos[3 * i + 2];
    }
    __shared__ double red[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s[a] += __shfl_xor(s[a], d, kWave);
        if (lane_id() == 0) red[a][threadIdx.x >> 6] = s[a];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        part[blockIdx.x * 3 + a] = (red[a][0] + red[a][1]) + (red[a][2] + red[a][3]);
    }
}

// one wavefront: lane l sums the blocks l, l + 64, ... in rising order, the 64 partial sums meet in a fixed butterfly --
// deterministic (a serial loop over the blocks was 54 us of dependent double adds behind dependent loads)
__global__ void grid_center(const double* __restrict__ part, int nblocks, int64_t n, const float* __restrict__ given,
                            GridHeader* h) {
    const int lane = lane_id();
    if (given) {
        if (lane < 3) h->center[lane] = given[lane];
        return;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double s = 0.0;
        for (int b = lane; b < nblocks; b += kWave) s += part[b * 3 + a];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, kWave);
        if (lane == 0) h->center[a] = n > 0 ? (float)(s / (double)n) : 0.0f;
    }
}

// extrema of the scaled positions, per block
__global__ __launch_bounds__(256) void grid_extrema(const GridParams p, const GridHeader* h, float* __restrict__ part) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * 256) {
        float s[3];
        grid_scaled(p, h, i, s);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            // fminf / fmaxf drop NaNs: turn any non-finite coordinate into infinite extrema (reported as an error)
            const bool ok = fabsf(s[a]) < INFINITY;
            mn[a] = ok ? fminf(mn[a], s[a]) : -INFINITY;
            mx[a] = ok ? fmaxf(mx[a], s[a]) : INFINITY;
        }
    }
    __shared__ float red[2][3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], d, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], d, kWave));
        }
        if (lane_id() == 0) {
            red[0][a][threadIdx.x >> 6] = mn[a];
            red[1][a][threadIdx.x >> 6] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        part[blockIdx.x * 6 + a] = fminf(fminf(red[0][a][0], red[0][a][1]), fminf(red[0][a][2], red[0][a][3]));
        part[blockIdx.x * 6 + 3 + a] = fmaxf(fmaxf(red[1][a][0], red[1][a][1]), fmaxf(red[1][a][2], red[1][a][3]));
    }
}

__global__ void grid_finish_bounds(const GridParams p, const float* __restrict__ part, int nblocks, GridHeader* h) {
    // one wavefront: the blocks' extrema by lanes, then a butterfly (min / max: any order gives the same bits)
    const int lane = lane_id();
    float mns[3], mxs[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float mn = INFINITY, mx = -INFINITY;
        for (int b = lane; b < nblocks; b += kWave) {
            mn = fminf(mn, part[b * 6 + a]);
            mx = fmaxf(mx, part[b * 6 + 3 + a]);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn = fminf(mn, __shfl_xor(mn, d, kWave));
            mx = fmaxf(mx, __shfl_xor(mx, d, kWave));
        }
        mns[a] = mn;
        mxs[a] = mx;
    }
    if (lane != 0) return;
    int64_t cells = p.n > 0 ? 1 : 0;
    for (int a = 0; a < 3; ++a) {
        const float mn = mns[a], mx = mxs[a];
        int lo = 0, d = 0;
        if (p.n > 0 && isfinite(mn) && isfinite(mx)) {
            // :167-170 -- floor is monotone: the extrema of the candidates follow from the extrema of the positions
            lo = (int)floorf(__fsub_rn(mn, p.h[a])) + p.lo[a];
            const int hi = (int)floorf(__fadd_rn(mx, p.h[a])) + p.lo[a] + p.len[a] - 1;
            d = hi - lo + 1;
        } else if (p.n > 0) {
            cells = -1;  // non-finite positions: the caller reports an error
        }
        h->minp[a] = lo;
        h->dims[a] = d;
        if (cells > 0) cells *= d;
    }
    h->cells = cells;
    h->total = 0;
}

// the two candidate base cells of particle i (relative to minp); returns whether the "+h" cell differs from the "-h" one
__device__ __forceinline__ bool grid_bases(const GridParams& p, const GridHeader* h, int64_t i, int (&c0)[3], int (&c1)[3]) {
    float s[3];
    grid_scaled(p, h, i, s);
    bool diff = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c0[a] = (int)floorf(__fsub_rn(s[a], p.h[a])) - h->minp[a];
        c1[a] = (int)floorf(__fadd_rn(s[a], p.h[a])) - h->minp[a];
        diff |= c0[a] != c1[a];
    }
    return diff;
}

// A scene that has dissolved into spray (the 100k dam break after ~40 steps, the 1M box once particles leave the shell) has a
// bounding box of billions of cells for a million occupied ones: the dense table `first[cell]` does not fit.  HASHED table
// (table_cells < 0 at the entry points: -table_cells slots, a power of two): slot = open addressing on the 64-bit cell index,
// first[slot] as before, the keys behind the `first` array.  Same three passes, same owners, same order: bit-identical output
// (round 6; until then a sort-based torch formulation with six host round trips took over -- 25 ms per step of the dam break).
constexpr unsigned long long kGridEmpty = ~0ull;
__device__ __forceinline__ uint32_t grid_hash(int64_t cell, uint32_t mask) {
    return (uint32_t)(((unsigned long long)cell * 0x9E3779B97F4A7C15ull) >> 32) & mask;
}
template <bool INSERT>
__device__ __forceinline__ int64_t grid_slot(unsigned long long* keys, uint32_t mask, int64_t cell) {
    uint32_t s = grid_hash(cell, mask);
    for (uint32_t probe = 0; probe <= mask; ++probe, s = (s + 1) & mask) {
        if (INSERT) {
            const unsigned long long old = atomicCAS(keys + s, kGridEmpty, (unsigned long long)cell);
            if (old == kGridEmpty || old == (unsigned long long)cell) return s;
        } else {
            const unsigned long long k = keys[s];
            if (k == (unsigned long long)cell) return s;
            if (k == kGridEmpty) return -1;
        }
    }
    return -1;  // (table full: the caller sized it for every candidate, so this does not happen)
}

// MODE 0: atomicMin pass; MODE 1: count owners per row; MODE 2: write owners
template <int MODE>
__global__ __launch_bounds__(256) void grid_pass(const GridParams p, const GridHeader* __restrict__ h,
                                                 uint32_t* __restrict__ first, int32_t* __restrict__ counts,
                                                 const int64_t* __restrict__ row_off, float* __restrict__ out,
                                                 int64_t out_capacity, int64_t table_cells) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = row < 2 * p.n;
    const bool second = row >= p.n;
    const int64_t i = live ? (second ? row - p.n : row) : 0;
    int c0[3] = {0, 0, 0}, c1[3] = {0, 0, 0};
    const bool diff = p.n > 0 && grid_bases(p, h, i, c0, c1);
    const int* c = second ? c1 : c0;
    const int64_t d0 = h->dims[0], d01 = d0 * h->dims[1];
    const bool active = live && (!second || diff);
    if (MODE == 0) {
        // A row whose base cell equals that of the row in the lane below proposes the same cells with larger ranks: it can win
        // none of them.  Particles mostly arrive in spatial order (neighbours in the array share a voxel), so this removes
        // about half of the atomics -- the same-address ones, which serialise in L2.
        const int64_t key = active ? c[0] + c[1] * d0 + c[2] * d01 : -1 - (int64_t)(threadIdx.x & 63);
        const int64_t below = __shfl_up(key, 1, 64);
        if (!active || ((threadIdx.x & 63) != 0 && below == key)) return;
    } else if (!active) {
        if (MODE == 1 && live) counts[row] = 0;
        return;
    }
    const int noff = p.len[0] * p.len[1] * p.len[2];
    const uint32_t k0 = (uint32_t)row * (uint32_t)noff;
    int cnt = 0;
    int64_t w = MODE == 2 ? row_off[row] : 0;
    int o = 0;
    for (int i0 = 0; i0 < p.len[0]; ++i0)
        for (int i1 = 0; i1 < p.len[1]; ++i1)
            for (int i2 = 0; i2 < p.len[2]; ++i2, ++o) {
                const int g0 = c[0] + p.lo[0] + i0, g1 = c[1] + p.lo[1] + i1, g2 = c[2] + p.lo[2] + i2;
                int64_t cell = g0 + g1 * d0 + g2 * d01;
                if (table_cells < 0) {
                    const uint32_t mask = (uint32_t)(-table_cells) - 1u;
                    unsigned long long* keys = (unsigned long long*)(first + (-table_cells));
                    cell = MODE == 0 ? grid_slot<true>(keys, mask, cell) : grid_slot<false>(keys, mask, cell);
                    if (cell < 0) continue;
                } else if (cell < 0 || cell >= table_cells) continue;  // table smaller than the header says: never out of bounds
                const uint32_t k = k0 + (uint32_t)o;
                if (MODE == 0) {
                    atomicMin(first + cell, k);
                } else if (first[cell] == k) {
                    if (MODE == 1) {
                        ++cnt;
                    } else if (w < out_capacity) {
                        // :172-179
                        const int g[3] = {g0 + h->minp[0], g1 + h->minp[1], g2 + h->minp[2]};
#pragma unroll
                        for (int a = 0; a < 3; ++a) {
                            const float t = __fmul_rn((float)g[a], p.voxel[a]);
                            out[3 * w + a] = p.centralize ? __fadd_rn(t, h->center[a]) : __fadd_rn(t, __fdiv_rn(p.voxel[a], 2.0f));
                        }
                        ++w;
                    }
                }
            }
    if (MODE == 1) counts[row] = cnt;
}

__global__ void grid_store_total(const int64_t* __restrict__ row_off, int64_t rows, GridHeader* h) {
    if (threadIdx.x == 0) h->total = row_off[rows];
}

struct GridLayout {
    size_t off_header, off_part, off_counts, off_rowoff, off_scan, total;
};

static GridLayout grid_layout(int64_t n) {
    GridLayout L;
    size_t o = 0;
    L.off_header = o; o += 256;
    L.off_part = o; o += align_up((size_t)kGridBlocks * 6 * sizeof(double), 256);
    L.off_counts = o; o += align_up((size_t)(2 * n + 1) * 4, 256);
    L.off_rowoff = o; o += align_up((size_t)(2 * n + 2) * 8, 256);
    L.off_scan = o; o += align_up(scan_tmp_bytes(2 * n + 1), 256);
    L.total = o;
    return L;
}

static int grid_params(GridParams& p, const float* pos, int64_t n, const float* voxel, int centralize, int pad,
                       float hyst) {
    if (n < 0 || (n > 0 && !pos) || !voxel || pad < 0 || pad > 8) return DMCF_EINVAL;
    p.pos = pos;
    p.n = n;
    p.centralize = centralize ? 1 : 0;
    int64_t noff = 1;
    for (int a = 0; a < 3; ++a) {
        const bool active = voxel[a] >= 1e-5f;
        p.voxel[a] = voxel[a];
        p.vs[a] = active ? voxel[a] : 1e-5f;
        p.h[a] = active ? hyst : 0.0f;
        p.lo[a] = active ? -pad : 0;           // :151-161
        p.len[a] = active ? 2 + 2 * pad : 1;
        noff *= p.len[a];
    }
    if (2 * n * noff >= (int64_t)0xffffffffLL) return DMCF_EUNSUPPORTED;  // candidate ranks are 32-bit
    return DMCF_OK;
}

}  // namespace dmcf

using namespace dmcf;

extern "C" {

size_t dmcf_grid_pos_workspace_bytes(int64_t n_points) {
    if (n_points < 0) return 0;
    return grid_layout(n_points).total;
}

int dmcf_grid_pos_bounds(const float* positions, int64_t n_points, const float* voxel_size, int centralize,
                         const float* center, int pad, float hyst, void* workspace, size_t workspace_bytes,
                         dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GridParams p;
    int rc = grid_params(p, positions, n_points, voxel_size, centralize, pad, hyst);
    if (rc != DMCF_OK) return rc;
    if (!workspace || ((uintptr_t)workspace & 255)) return DMCF_EINVAL;
    const GridLayout L = grid_layout(n_points);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    char* ws = (char*)workspace;
    GridHeader* h = (GridHeader*)(ws + L.off_header);
    const int64_t want = (n_points + 255) / 256;
    const int nb = (int)(want < 1 ? 1 : (want > kGridBlocks ? kGridBlocks : want));
    if (p.centralize) {
        double* part = (double*)(ws + L.off_part);
        if (!center) hipLaunchKernelGGL(grid_sum, dim3(nb), dim3(256), 0, stream, positions, n_points, part);
        hipLaunchKernelGGL(grid_center, dim3(1), dim3(64), 0, stream, part, nb, n_points, center, h);
    }
    float* fpart = (float*)(ws + L.off_part);
    hipLaunchKernelGGL(grid_extrema, dim3(nb), dim3(256), 0, stream, p, h, fpart);
    hipLaunchKernelGGL(grid_finish_bounds, dim3(1), dim3(64), 0, stream, p, fpart, nb, h);
    return check_launch();
}

int dmcf_grid_pos_count(const float* positions, int64_t n_points, const float* voxel_size, int centralize, int pad,
                        float hyst, void* workspace, size_t workspace_bytes, void* cell_table, int64_t table_cells,
                        dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GridParams p;
    int rc = grid_params(p, positions, n_points, voxel_size, centralize, pad, hyst);
    if (rc != DMCF_OK) return rc;
    if (!workspace || ((uintptr_t)workspace & 255) || (table_cells != 0 && !cell_table)) return DMCF_EINVAL;
    if (table_cells < 0 && ((-table_cells) & (-table_cells - 1))) return DMCF_EINVAL;  // hashed: a power of two of slots
    const GridLayout L = grid_layout(n_points);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    char* ws = (char*)workspace;
    GridHeader* h = (GridHeader*)(ws + L.off_header);
    uint32_t* first = (uint32_t*)cell_table;
    int32_t* counts = (int32_t*)(ws + L.off_counts);
    int64_t* row_off = (int64_t*)(ws + L.off_rowoff);
    const int64_t rows = 2 * n_points;
    // (hashed: 4 bytes of `first` + 8 bytes of key per slot, all ones = empty)
    const size_t table_bytes = table_cells >= 0 ? (size_t)table_cells * 4 : (size_t)(-table_cells) * 12;
    if (table_bytes > 0 && hipMemsetAsync(first, 0xff, table_bytes, stream) != hipSuccess) return DMCF_ELAUNCH;
    if (rows > 0) {
        const unsigned g = (unsigned)((rows + 255) / 256);
        hipLaunchKernelGGL((grid_pass<0>), dim3(g), dim3(256), 0, stream, p, h, first, counts, (const int64_t*)nullptr,
                           (float*)nullptr, (int64_t)0, table_cells);
        hipLaunchKernelGGL((grid_pass<1>), dim3(g), dim3(256), 0, stream, p, h, first, counts, (const int64_t*)nullptr,
                           (float*)nullptr, (int64_t)0, table_cells);
    }
    rc = scan_counts_to_row_splits(counts, row_off, rows, ws + L.off_scan, L.total - L.off_scan, stream);
    if (rc != DMCF_OK) return rc;
    hipLaunchKernelGGL(grid_store_total, dim3(1), dim3(64), 0, stream, row_off, rows, h);
    return check_launch();
}

int dmcf_grid_pos_write(const float* positions, int64_t n_points, const float* voxel_size, int centralize, int pad,
                        float hyst, void* workspace, size_t workspace_bytes, const void* cell_table, int64_t table_cells,
                        float* out, int64_t out_capacity, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GridParams p;
    int rc = grid_params(p, positions, n_points, voxel_size, centralize, pad, hyst);
    if (rc != DMCF_OK) return rc;
    if (!workspace || ((uintptr_t)workspace & 255) || out_capacity < 0 || (out_capacity > 0 && (!out || !cell_table)))
        return DMCF_EINVAL;
    const GridLayout L = grid_layout(n_points);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    char* ws = (char*)workspace;
    const GridHeader* h = (const GridHeader*)(ws + L.off_header);
    const int64_t rows = 2 * n_points;
    if (rows > 0 && out_capacity > 0) {
        const unsigned g = (unsigned)((rows + 255) / 256);
        hipLaunchKernelGGL((grid_pass<2>), dim3(g), dim3(256), 0, stream, p, h, (uint32_t*)cell_table,
                           (int32_t*)(ws + L.off_counts), (const int64_t*)(ws + L.off_rowoff), out, out_capacity, table_cells);
    }
    return check_launch();
}

}  // extern "C"
