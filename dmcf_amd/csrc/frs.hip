// Fixed-radius neighbour search for gfx950: cell-sorted uniform grid + one wavefront per query.
//
// Replaces ml3d.layers.FixedRadiusSearch (reference call sites utils/convolutions.py:207-210,354-358,
// utils/tools/losses.py:296-298).  The Open3D structure behind that layer (spatial hash with ~64
// points per bin, 8 corner bins per query) is NOT what is built here; only its result contract is
// kept (see include/dmcf_hip.h).  MI355X-first choices:
//   * cell edge ~= radius/3 (then radius/2; radius-sized or coarser cells when the grid would be too sparse or too big) and
//     points re-ordered by cell into one float4 {x,y,z,index} array, so a query reads <= 49 contiguous x-runs with
//     coalesced 16-B loads instead of chasing indices; each run is trimmed to the chord of the search sphere in its row
//     of cells (~45 % of the candidates are hits; 27 % for the untrimmed box of R/2 cells, ~6.5 % for the reference's 2R
//     hash bins);
//   * one 64-lane wavefront per query: lanes stride the flattened candidate list, hits are compacted
//     with a ballot + mbcnt prefix (no atomics, no LDS), rows come out in a deterministic order;
//   * everything that sizes the grid (bounding box, cell edge, dims) is computed on the device into a
//     header inside the workspace, so the build needs no host round trip.
#include "cconv_common.h"  // window_value

namespace dmcf {

#ifndef FRS_CELL_DIV
#define FRS_CELL_DIV 3  // preferred cells per radius (measured on the 1M box: 2 -> 3 cuts the 3e8-pair searches by 9 %, 4 needs two row batches)
#endif

constexpr double kFrsSpread = 3.0;  // the grid's box: mean +- this many standard deviations of the points, see frs_finish_header

struct FrsHeader {
    float origin[3];
    float inv_cell[3];
    int32_t dims[3];
    int32_t ncells;
    uint32_t bb_min[3];  // order-preserving uint encoding of floats
    uint32_t bb_max[3];
    float radius;
    int32_t n_points;
    double sum[3], sumsq[3];  // of the point coordinates: mean and spread for the grid's box (frs_finish_header)
    int32_t pad[2];
};
static_assert(sizeof(FrsHeader) == 128, "header layout");

struct FrsLayout {
    int64_t n, m, table;
    size_t off_header, off_cell_start, off_cell_fill, off_point_cell, off_tmp_idx, off_sorted, off_counts,
        off_scan, scan_bytes, off_flags, total;
};

static int64_t frs_table_size(int64_t n) {
    int64_t t = 4 * n;
    if (t < 4096) t = 4096;
    if (t > ((int64_t)1 << 26)) t = (int64_t)1 << 26;
    return t;
}

static FrsLayout frs_layout(int64_t n, int64_t m) {
    FrsLayout L;
    L.n = n;
    L.m = m;
    L.table = frs_table_size(n);
    size_t off = 0;
    L.off_header = off;        off += align_up(sizeof(FrsHeader), 256);
    L.off_cell_start = off;    off += align_up((size_t)(L.table + 1) * 4, 256);
    L.off_cell_fill = off;     off += align_up((size_t)(L.table + 1) * 4, 256);
    L.off_point_cell = off;    off += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    L.off_tmp_idx = off;       off += align_up((size_t)(n > 0 ? n : 1) * 4, 256);
    L.off_sorted = off;        off += align_up((size_t)(n > 0 ? n : 1) * 16, 256);
    L.off_counts = off;        off += align_up((size_t)(m > 0 ? m : 1) * 4, 256);
    const int64_t scan_n = (L.table + 1) > m ? (L.table + 1) : m;
    L.scan_bytes = scan_tmp_bytes(scan_n);
    L.off_scan = off;          off += L.scan_bytes;
    // one byte per query: "this query may see a point the reference cannot" (open3d visibility flags, see frs_fix); every entry
    // point writes the bytes it then reads inside ONE call, so searches on a shared structure only need stream order
    L.off_flags = off;         off += align_up((size_t)(m > 0 ? m : 1), 256);
    L.total = off;
    return L;
}

__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void frs_init_header(FrsHeader* h, float radius, int32_t n) {
    if (threadIdx.x == 0) {
        for (int a = 0; a < 3; ++a) {
            h->bb_min[a] = 0xffffffffu;
            h->bb_max[a] = 0u;
            h->sum[a] = 0.0;
            h->sumsq[a] = 0.0;
        }
        h->radius = radius;
        h->n_points = n;
    }
}

__global__ __launch_bounds__(256) void frs_bbox(const float* __restrict__ pts, int64_t n, FrsHeader* h) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float sm[3] = {0.0f, 0.0f, 0.0f}, sq[3] = {0.0f, 0.0f, 0.0f};  // (a thread sums a few dozen values: float is enough)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
            if (isfinite(v)) {
                sm[a] += v;
                sq[a] += v * v;
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
            sq[a] += __shfl_xor(sq[a], d, kWave);
        }
    }
    // one atomic per block and component: same-address atomics serialise in L2 (measured: 24k of them = 240 us)
    __shared__ float red[4][3][4];
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            red[0][a][w] = mn[a];
            red[1][a][w] = mx[a];
            red[2][a][w] = sm[a];
            red[3][a][w] = sq[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float lo = fminf(fminf(red[0][a][0], red[0][a][1]), fminf(red[0][a][2], red[0][a][3]));
        const float hi = fmaxf(fmaxf(red[1][a][0], red[1][a][1]), fmaxf(red[1][a][2], red[1][a][3]));
        atomicMin(&h->bb_min[a], f2ord(lo));
        atomicMax(&h->bb_max[a], f2ord(hi));
        atomicAdd(&h->sum[a], (double)red[2][a][0] + (double)red[2][a][1] + (double)red[2][a][2] + (double)red[2][a][3]);
        atomicAdd(&h->sumsq[a], (double)red[3][a][0] + (double)red[3][a][1] + (double)red[3][a][2] + (double)red[3][a][3]);
    }
}

// one thread: choose the cell edge (>= 1.001 R so that [q-R, q+R] spans at most 3 cells) and coarsen
// it until the dense grid fits the cell table.  Coarser cells only add candidates, never lose any.
__device__ void frs_finish_header_one(FrsHeader* h, int64_t table) {
    float lo[3], ext[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = ord2f(h->bb_min[a]);
        const float hi = ord2f(h->bb_max[a]);
        ext[a] = hi - lo[a];
        if (!(ext[a] >= 0.0f) || !isfinite(ext[a])) ext[a] = 0.0f;  // empty / non-finite input
        if (!isfinite(lo[a])) lo[a] = 0.0f;
        // The grid covers the BULK of the points: mean +- kSpread standard deviations (a uniformly filled box is +-1.73 of
        // them wide), clipped to the bounding box.  Points beyond are binned into the border cells -- which have no outer
        // face, so every query still sees every point in range -- instead of stretching the grid: a few particles that left
        // the scene (a splash, a droplet falling forever) otherwise coarsen the cells for everybody (dam-break rollout: the
        // search went from 1 to 12 ms per step while 100 of 100,000 particles fell out of the tank).
        if (h->n_points > 0) {
            const double mean = h->sum[a] / (double)h->n_points;
            const double var = h->sumsq[a] / (double)h->n_points - mean * mean;
            const double sd = var > 0.0 ? sqrt(var) : 0.0;
            const float blo = (float)(mean - kFrsSpread * sd), bhi = (float)(mean + kFrsSpread * sd);
            if (isfinite(blo) && isfinite(bhi) && bhi > blo) {
                const float nlo = fmaxf(lo[a], blo), nhi = fminf(hi, bhi);
                if (nhi >= nlo) {
                    lo[a] = nlo;
                    ext[a] = nhi - nlo;
                }
            }
        }
    }
    // Preferred cell edge R/3, then R/2 (the rows of cells are trimmed to the chord of the search sphere, so finer cells
    // mean fewer candidates: ~45 % of them are hits at R/3; the untrimmed box of R-sized cells gives 15 %), unless the
    // grid would be mostly empty (fewer than one point per two cells) or does not fit the table.  Then R, then coarser.
    // "1.001": [q-R, q+R] spans at most 2s+1 cells of edge 1.001 R / s.
    int32_t d[3];
    float cell = h->radius * 1.001f;
    for (int div = FRS_CELL_DIV; div >= 2; --div) {
        const float c = h->radius * 1.001f / (float)div;
        double prod = 1.0;
        for (int a = 0; a < 3; ++a) prod *= floor((double)ext[a] / (double)c) + 1.0;
        if (prod <= (double)table && (double)h->n_points >= 0.5 * prod) {
            cell = c;
            break;
        }
    }
    for (int it = 0; it < 64; ++it) {
        double prod = 1.0;
        for (int a = 0; a < 3; ++a) {
            const double cnt = floor((double)ext[a] / (double)cell) + 1.0;
            d[a] = cnt > 2.0e9 ? 2000000000 : (int32_t)cnt;
            prod *= cnt;
        }
        if (prod <= (double)table) break;
        const float grow = (float)cbrt(prod / (double)table) * 1.02f;
        cell *= grow > 1.05f ? grow : 1.05f;
    }
    for (int a = 0; a < 3; ++a) {
        h->origin[a] = lo[a];
        h->inv_cell[a] = 1.0f / cell;
        h->dims[a] = d[a];
    }
    h->ncells = d[0] * d[1] * d[2];
}

__global__ void frs_finish_header(FrsHeader* h, int64_t table) {
    if (threadIdx.x == 0) frs_finish_header_one(h, table);
}

// cell coordinate along one axis; the SAME function is used for points and for the ends of a query's
// range, so monotonicity of floor((x - o) * inv) guarantees every in-range point is visited.
__device__ __forceinline__ int cell_coord(float x, float origin, float inv, int dim) {
    float c = floorf((x - origin) * inv);
    c = fminf(fmaxf(c, -1.0f), (float)dim);  // also maps NaN to -1
    return (int)c;
}

__global__ __launch_bounds__(256) void frs_count_cells(const float* __restrict__ pts, int64_t n,
                                                       const FrsHeader* __restrict__ h,
                                                       int32_t* __restrict__ point_cell,
                                                       uint32_t* __restrict__ cell_count, uint32_t* __restrict__ slot_of) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c[a] = cell_coord(pts[3 * i + a], h->origin[a], h->inv_cell[a], h->dims[a]);
        c[a] = min(max(c[a], 0), h->dims[a] - 1);
    }
    const int32_t cell = (c[2] * h->dims[1] + c[1]) * h->dims[0] + c[0];
    point_cell[i] = cell;
    // (the counter's old value is the point's place among its cell's members: the scatter needs no second round of atomics)
    slot_of[i] = atomicAdd(&cell_count[cell], 1u);
}

__global__ __launch_bounds__(256) void frs_scatter(int64_t n, const int32_t* __restrict__ point_cell,
                                                   const uint32_t* __restrict__ cell_start,
                                                   const uint32_t* __restrict__ slot_of, int32_t* __restrict__ tmp_idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    tmp_idx[cell_start[point_cell[i]] + slot_of[i]] = (int32_t)i;
}

// the atomic scatter leaves each cell's members in arbitrary order; rank every point inside its cell
// by index so the sorted array (and therefore every CSR row) is deterministic.
__global__ __launch_bounds__(256) void frs_rank_and_place(const float* __restrict__ pts, int64_t n,
                                                          const int32_t* __restrict__ point_cell,
                                                          const uint32_t* __restrict__ cell_start,
                                                          const int32_t* __restrict__ tmp_idx,
                                                          float4* __restrict__ sorted) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t cell = point_cell[i];
    const uint32_t b = cell_start[cell], e = cell_start[cell + 1];
    uint32_t rank = 0;
    for (uint32_t s = b; s < e; ++s) rank += (tmp_idx[s] < (int32_t)i) ? 1u : 0u;
    sorted[b + rank] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __int_as_float((int32_t)i));
}

// The whole build as ONE launch of one workgroup, for point sets of a few thousand (the 2-D scenes, BASELINE.json configs 2 / 3,
// build six tables of 2 - 4k points per step): ten launches of one or a handful of workgroups each -- header, box, cell edge,
// memset, histogram, three of the scan, scatter, rank -- cost their ~45 us of launch-to-launch latency, not their work.  Same
// phases, same results (the box sums are added in another order: they only place the grid, which no result depends on).
constexpr int kSmallThreads = 1024;
constexpr int64_t kSmallBuildMax = 16384;  // points; the cell table then has at most 65536 entries
__global__ __launch_bounds__(kSmallThreads) void frs_build_small(const float* __restrict__ pts, int64_t n, float radius,
                                                                  FrsHeader* h, int64_t table, uint32_t* __restrict__ cell_start,
                                                                  uint32_t* __restrict__ cell_fill, int32_t* __restrict__ point_cell,
                                                                  int32_t* __restrict__ tmp_idx, float4* __restrict__ sorted) {
    __shared__ float red[4][3][kSmallThreads / 64];
    __shared__ uint32_t part[kSmallThreads];
    const int tid = threadIdx.x, w = tid >> 6;
    uint32_t* slot_of = (uint32_t*)sorted;  // (the places inside the cells wait in the sorted array's memory, written last)
    // box, sums
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    float sm[3] = {0.0f, 0.0f, 0.0f}, sq[3] = {0.0f, 0.0f, 0.0f};
    for (int64_t i = tid; i < n; i += kSmallThreads) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * i + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
            if (isfinite(v)) {
                sm[a] += v;
                sq[a] += v * v;
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
            sq[a] += __shfl_xor(sq[a], d, kWave);
        }
        if (lane_id() == 0) {
            red[0][a][w] = mn[a];
            red[1][a][w] = mx[a];
            red[2][a][w] = sm[a];
            red[3][a][w] = sq[a];
        }
    }
    for (int64_t e = tid; e <= table; e += kSmallThreads) cell_fill[e] = 0u;
    __syncthreads();
    if (tid == 0) {
        h->radius = radius;
        h->n_points = (int32_t)n;
        for (int a = 0; a < 3; ++a) {
            float lo = INFINITY, hi = -INFINITY;
            double s1 = 0.0, s2 = 0.0;
            for (int v = 0; v < kSmallThreads / 64; ++v) {
                lo = fminf(lo, red[0][a][v]);
                hi = fmaxf(hi, red[1][a][v]);
                s1 += (double)red[2][a][v];
                s2 += (double)red[3][a][v];
            }
            h->bb_min[a] = n > 0 ? f2ord(lo) : 0xffffffffu;
            h->bb_max[a] = n > 0 ? f2ord(hi) : 0u;
            h->sum[a] = s1;
            h->sumsq[a] = s2;
        }
        frs_finish_header_one(h, table);
    }
    __threadfence_block();
    __syncthreads();
    // histogram: each point keeps the counter's old value
    for (int64_t i = tid; i < n; i += kSmallThreads) {
        int c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            c[a] = cell_coord(pts[3 * i + a], h->origin[a], h->inv_cell[a], h->dims[a]);
            c[a] = min(max(c[a], 0), h->dims[a] - 1);
        }
        const int32_t cell = (c[2] * h->dims[1] + c[1]) * h->dims[0] + c[0];
        point_cell[i] = cell;
        slot_of[i] = atomicAdd(&cell_fill[cell], 1u);
    }
    __syncthreads();
    // exclusive scan of the table + 1 counters: a contiguous piece per thread, the pieces' sums scanned in LDS
    const int64_t per = (table + 1 + kSmallThreads - 1) / kSmallThreads;
    const int64_t e0 = min((int64_t)tid * per, table + 1), e1 = min(e0 + per, table + 1);
    uint32_t mine = 0;
    for (int64_t e = e0; e < e1; ++e) mine += cell_fill[e];
    part[tid] = mine;
    __syncthreads();
    for (int d = 1; d < kSmallThreads; d <<= 1) {
        const uint32_t add = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    uint32_t run = part[tid] - mine;
    for (int64_t e = e0; e < e1; ++e) {
        cell_start[e] = run;
        run += cell_fill[e];
    }
    __syncthreads();
    for (int64_t i = tid; i < n; i += kSmallThreads) tmp_idx[cell_start[point_cell[i]] + slot_of[i]] = (int32_t)i;
    __syncthreads();
    // rank every point inside its cell by index: a deterministic sorted array
    for (int64_t i = tid; i < n; i += kSmallThreads) {
        const int32_t cell = point_cell[i];
        const uint32_t b = cell_start[cell], e = cell_start[cell + 1];
        uint32_t rank = 0;
        for (uint32_t t = b; t < e; ++t) rank += (tmp_idx[t] < (int32_t)i) ? 1u : 0u;
        // (slot_of aliases `sorted`: every place was read by the scatter above, before the barrier)
        sorted[b + rank] = make_float4(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], __int_as_float((int32_t)i));
    }
}

#ifndef FRS_WIN
#define FRS_WIN 4
#endif
constexpr int kWin = FRS_WIN;  // candidate windows (64 each) in flight per iteration

__device__ __forceinline__ int wave_inclusive_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// ---- what Open3D's search can SEE (DMCF_FRS_OPEN3D_VOXEL_WALK, DMCF_FRS_OPEN3D_CORNER_VOXELS) -- opt-in emulations ------------
// open3d 0.15.2 (FixedRadiusSearchImpl.h, restated in oracle/dmcf_oracle.c) hashes the points into voxels of edge 2 R and, for a
// query, visits the hash bins of its own voxel and of the 8 voxels holding the corners q +- R (round 3
// read it as the 8 corner voxels alone -- DMCF_FRS_OPEN3D_CORNER_VOXELS keeps that reading).  In exact arithmetic those voxels
// cover the search sphere, and the DEFAULT search (neither flag) returns exactly that: the set of the distance test.  In float
// they need not: with q a rounding step from the middle of a voxel, floor(fl(q - R) / 2R) and floor(fl(q + R) / 2R) can be TWO
// apart -- the voxel between them, where the query itself and most of its neighbours live, is visited only as the query's own
// voxel (VOXEL_WALK: the row keeps what lies in that one voxel and loses the rest, typically a third of it) or not at all
// (CORNER_VOXELS: a nearly empty row); one query in ~10^6 per search of the 1M-particle rollout.  A pair at distance R within
// rounding can likewise sit in a voxel one step outside the corners.  With a flag set the scan reproduces that visibility: a
// hit counts only if the point's voxel hashes into one of the query's bins.  Tested where it can matter, at almost no cost
// elsewhere: every hit of a query whose corner voxels are two apart on some axis, otherwise only hits in the outermost shell
// of the sphere, where rounding could put the point's voxel outside [corner-, corner+].
constexpr int kO3dFlags = DMCF_FRS_OPEN3D_CORNER_VOXELS | DMCF_FRS_OPEN3D_VOXEL_WALK;
__device__ __forceinline__ uint64_t o3d_spatial_hash(int x, int y, int z) {
    const uint32_t hsh = ((uint32_t)x * 73856096u) ^ ((uint32_t)y * 193649663u) ^ ((uint32_t)z * 83492791u);
    return (uint64_t)(int64_t)(int32_t)hsh;  // (int arithmetic, converted to size_t: sign extended)
}

struct O3dView {
    float q[3];
    float radius, inv_voxel;
    float r2_inner;      // hits with d^2 above this need the exact visibility test (-1: all of them)
    bool own;            // the query's own voxel is visited too (DMCF_FRS_OPEN3D_VOXEL_WALK)
};

// voxels of the corners q - R, q + R per axis
__device__ __forceinline__ void o3d_corners(const O3dView& v, int (&vlo)[3], int (&vhi)[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        vlo[a] = (int)floorf(__fmul_rn(__fsub_rn(v.q[a], v.radius), v.inv_voxel));
        vhi[a] = (int)floorf(__fmul_rn(__fadd_rn(v.q[a], v.radius), v.inv_voxel));
    }
}

__device__ __forceinline__ O3dView o3d_view(const float (&q)[3], float radius, float r2, int flags) {
    O3dView v;
    v.own = (flags & DMCF_FRS_OPEN3D_VOXEL_WALK) != 0;
    const float voxel = __fmul_rn(2.0f, radius);
    v.inv_voxel = __fdiv_rn(1.0f, voxel);
    v.radius = radius;
    // The corner voxels of an axis can only be two apart when q sits within rounding of the MIDDLE of a voxel: a cheap test
    // (distance of q / voxel from k + 1/2 below 1e-3 -- rounding is 1e-7 of q / voxel, coordinates beyond 1000 voxels count as
    // near) picks the one query in ~300 that gets the exact one
    bool near = false;
    float qmax = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        v.q[a] = q[a];
        const float u = q[a] * v.inv_voxel;
        near |= fabsf(__builtin_amdgcn_fractf(u) - 0.5f) < 1e-3f;
        qmax = fmaxf(qmax, fabsf(u));
    }
    near |= !(qmax < 1000.0f);
    // d^2 <= R^2 (1 - delta) puts every coordinate of the point inside [fl(q - R), fl(q + R)] -- whose voxels are the corners'
    // or lie between them -- when delta exceeds twice the relative rounding of q +- R (6e-8 (2 q / voxel + 1)) plus that of the
    // squared distance and of R^2 (3e-7); a quarter of margin
    const float delta = 1.5e-7f * (2.0f * qmax + 1.0f) + 4e-7f;
    bool all = !(delta < 0.5f);
    if (__ballot(near) != 0ull) {  // (wave uniform: one query per wave)
        int vlo[3], vhi[3];
        o3d_corners(v, vlo, vhi);
        all |= vhi[0] - vlo[0] > 1 || vhi[1] - vlo[1] > 1 || vhi[2] - vlo[2] > 1;
    }
    v.r2_inner = all ? -1.0f : r2 * (1.0f - delta);
    return v;
}

__device__ __forceinline__ bool o3d_visible(float px, float py, float pz, const O3dView& v, int n_points) {
    int vlo[3], vhi[3];
    o3d_corners(v, vlo, vhi);
    const int x = (int)floorf(__fmul_rn(px, v.inv_voxel)), y = (int)floorf(__fmul_rn(py, v.inv_voxel)),
              z = (int)floorf(__fmul_rn(pz, v.inv_voxel));
    if ((x == vlo[0] || x == vhi[0]) && (y == vlo[1] || y == vhi[1]) && (z == vlo[2] || z == vhi[2])) return true;
    const int ox = (int)floorf(__fmul_rn(v.q[0], v.inv_voxel)), oy = (int)floorf(__fmul_rn(v.q[1], v.inv_voxel)),
              oz = (int)floorf(__fmul_rn(v.q[2], v.inv_voxel));
    if (v.own && x == ox && y == oy && z == oz) return true;
    // not one of the visited voxels: still found if its bin is (the FixedRadiusSearch layer's table: n / 64 bins, 1 .. 2^25)
    int64_t size = (int64_t)n_points / 64;
    size = size < 1 ? 1 : (size > 33554432 ? 33554432 : size);
    const uint64_t bin = o3d_spatial_hash(x, y, z) % (uint64_t)size;
    if (v.own && o3d_spatial_hash(ox, oy, oz) % (uint64_t)size == bin) return true;
    for (int c = 0; c < 8; ++c) {
        const uint64_t cb = o3d_spatial_hash((c & 1) ? vhi[0] : vlo[0], (c & 2) ? vhi[1] : vlo[1], (c & 4) ? vhi[2] : vlo[2]) % (uint64_t)size;
        if (cb == bin) return true;
    }
    return false;
}

// The candidate scan of one query by one wavefront.  MODE 0: count the hits; MODE 1: write them to the CSR row at
// out_base; MODE 2: add window(d^2 / R^2) of every hit to `wsum` (per lane; the caller reduces over the wave).
// Returns the number of hits.  Hits come out in a fixed order (cell rows, then position in the cell-sorted array).
// EXACT (open3d visibility flags only): every hit that could lie outside the reference's 8 bins takes the exact
// visibility test -- the form of the FIXUP kernel (frs_fix), which re-scans the few queries the hot kernels flag.  !EXACT is the
// hot form: it only NOTICES such a hit (d^2 above r2_inner: one compare per window) and reports it through `redo`.  The test
// itself (a 64-bit modulo) stays out of the hot kernels: with it inside they needed 71 instead of 52 registers -- seven
// instead of eight waves per SIMD for a latency-bound scan -- and the searches of a step took 15 - 20 % longer.
template <int MODE, bool EXACT>
__device__ __forceinline__ int32_t frs_scan(float qx, float qy, float qz, const FrsHeader* __restrict__ h,
                                            const uint32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
                                            float radius, int flags, int64_t out_base, int32_t* __restrict__ nbr_index,
                                            float* __restrict__ nbr_dist, int window, float inv_r2, float& wsum,
                                            uint32_t* marks, int32_t row_cap, bool& redo) {
    const int lane = lane_id();
    int mark_tag = 0;
    for (int w = 0; w < kWin; ++w) marks[w * kWave + lane] = 0;  // LDS is not cleared between workgroups: stale tags of an earlier wave must not match ours
    const float q[3] = {qx, qy, qz};
    const float r2 = __fmul_rn(radius, radius);
    const bool o3d = (flags & kO3dFlags) != 0;
    O3dView view;
    view.r2_inner = r2;
    view.own = false;
    if (o3d) view = o3d_view(q, radius, r2, flags);
    int lo[3], hi[3];
    bool empty = h->ncells <= 0 || h->n_points <= 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // slack >> float rounding of (q +- R) and of the distance test; keeps the candidate set a superset
        const float slack = 1e-4f * radius + 4.8e-7f * (fabsf(q[a]) + radius);
        // (both ends clamped INTO the grid: the grid covers the bulk of the points, the border cells hold everything binned
        // from beyond it, and a query out there must look into them)
        lo[a] = min(max(cell_coord(q[a] - radius - slack, h->origin[a], h->inv_cell[a], h->dims[a]), 0), h->dims[a] - 1);
        hi[a] = max(min(cell_coord(q[a] + radius + slack, h->origin[a], h->inv_cell[a], h->dims[a]), h->dims[a] - 1), 0);
        empty |= lo[a] > hi[a];
    }
    int32_t cnt = 0;
    if (!empty) {
        const int ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
        const bool ignore = (flags & DMCF_FRS_IGNORE_QUERY_POINT) != 0;
        // (y,z) rows are taken 64 at a time (49 at most when the cell edge is ~R/3)
        for (int row0 = 0; row0 < ny * nz; row0 += kWave) {
            const int r = row0 + lane;
            int32_t start = 0, len = 0;
            if (r < ny * nz) {
                const int cy = lo[1] + r % ny, cz = lo[2] + r / ny;
                const int32_t base = (cz * h->dims[1] + cy) * h->dims[0];
                // Trim the row to the chord of the search sphere: the box of cells around the query holds 15.6 R^3 of
                // candidates for a 4.2 R^3 sphere.  gap = distance (in cells) from the query to the row's slab of cells along
                // y / z -- the outermost cells also hold everything binned from beyond the grid, so they have no outer face --
                // minus a slack far above the rounding of the binning; the x range then only covers the chord.  Still a
                // superset of the neighbours: the distance test below decides.
                auto gap = [&](int a, int c) -> float {
                    const float u = (q[a] - h->origin[a]) * h->inv_cell[a];
                    float g = 0.0f;
                    if (c > 0) g = fmaxf(g, (float)c - u);
                    if (c < h->dims[a] - 1) g = fmaxf(g, u - (float)(c + 1));
                    g = fmaxf(g - (1e-3f + 1e-6f * fabsf(u)), 0.0f);
                    return g * __builtin_amdgcn_rcpf(h->inv_cell[a]);  // 1 ulp: far inside the slack
                };
                const float dy = gap(1, cy), dz = gap(2, cz);
                const float rs = radius * 1.0002f;
                const float h2 = rs * rs - dy * dy - dz * dz;
                if (h2 >= 0.0f) {
                    const float hx = __builtin_amdgcn_sqrtf(h2) * 1.0001f;
                    const float slack = 1e-4f * radius + 4.8e-7f * (fabsf(q[0]) + radius);
                    // (clamped INTO the grid like lo / hi: a chord that lies entirely beyond the grid's x range still has to
                    // visit the border cell, which holds the points binned from out there)
                    const int xlo = max(min(cell_coord(q[0] - hx - slack, h->origin[0], h->inv_cell[0], h->dims[0]), h->dims[0] - 1), lo[0]);
                    const int xhi = min(max(cell_coord(q[0] + hx + slack, h->origin[0], h->inv_cell[0], h->dims[0]), 0), hi[0]);
                    if (xlo <= xhi) {
                        start = (int32_t)cell_start[base + xlo];
                        len = (int32_t)cell_start[base + xhi + 1] - start;
                    }
                }
            }
            // inclusive scan of run lengths across lanes (DPP: no LDS round trips)
            const int32_t incl = wave_inclusive_add(len);
            const int32_t total = __builtin_amdgcn_readlane(incl, kWave - 1);
            const int32_t excl = incl - len;
            const int32_t rel = start - excl;  // candidate c of this run sits at sorted[rel + flat]
            // Which run does flat index f fall into?  The non-empty runs are numbered consecutively in lane order (ballot +
            // mbcnt) and leave their `rel` in a 64-entry per-wave LDS table under that number.  Every non-empty run that STARTS
            // inside the window [f0, f0 + 64) tags its start slot of a per-wave LDS array (the tag is the window counter, so
            // the array never needs clearing); a ballot over the slots gives the mask of starts, and the run of slot l is
            // (runs started before the window) + (starts at slots <= l) - 1: one LDS round trip, two mbcnt and a table read.
            // (Round 1: a 6-step ds_bpermute binary search, 3.8 ms for a 307M-pair list; rounds 2 - 3: the start slots carried
            // the run's LANE and a 6-step DPP max-scan spread it over the slots, 3.1 ms, a fifth of the loop's instructions
            // and 27 s_nops per four windows behind the DPP hazards.)
            const unsigned long long nonempty = __ballot(len > 0);
            const int crun = __builtin_amdgcn_mbcnt_hi((unsigned)(nonempty >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nonempty, 0));
            int32_t* relT = (int32_t*)(marks + kWin * kWave);
            if (len > 0) relT[crun] = rel;
            int started = 0;                   // non-empty runs that start before the current window
            auto locate = [&](int32_t f0, uint32_t* mk) -> int32_t {
                const uint32_t tag = (uint32_t)(++mark_tag);
                const int32_t sl = excl - f0;
                if (len > 0 && sl >= 0 && sl < kWave) mk[sl] = tag;
                // lanes talk to each other through LDS here: without a (wavefront-scope) fence the compiler may keep
                // using this lane's own last value of mk[lane] (seen: it sank the load into the store's branch)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const bool mine = mk[lane] == tag;
                const unsigned long long starts = __ballot(mine);
                const int upto = __builtin_amdgcn_mbcnt_hi((unsigned)(starts >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)starts, 0)) +
                                 (mine ? 1 : 0);
                const int run = max(started + upto - 1, 0);
                started += __popcll(starts);
                return relT[run] + f0 + lane;
            };
            auto test = [&](int32_t f0, const float4& p) {
                const int32_t f = f0 + lane;
                bool hit = false;
                float d2 = 0.0f;
                int32_t pidx = 0;
                if (f < total) {
                    d2 = dist2_unfused(p.x, p.y, p.z, qx, qy, qz);
                    hit = d2 <= r2;
                    if (ignore && p.x == qx && p.y == qy && p.z == qz) hit = false;
                    pidx = __float_as_int(p.w);
                }
                if (o3d) {  // (rare: the outermost shell of the sphere, or a query whose corner voxels are two apart)
                    const bool check = hit && d2 > view.r2_inner;
                    if (EXACT) {
                        if (__ballot(check) != 0ull) {
                            if (check) hit = o3d_visible(p.x, p.y, p.z, view, h->n_points);
                        }
                    } else {
                        redo |= __ballot(check) != 0ull;
                    }
                }
                const unsigned long long mask = __ballot(hit);
                if (MODE == 1 && hit) {
                    const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                 __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
                    const int32_t slot = cnt + before;
                    if (slot < row_cap) {  // padded rows: hits beyond the row's capacity are counted, not written
                        const int64_t o = out_base + slot;
                        nbr_index[o] = pidx;
                        if (nbr_dist) nbr_dist[o] = d2;
                    }
                }
                if (MODE == 2 && hit) wsum += window_value(window, d2, inv_r2, 1.0f);
                cnt += __popcll(mask);
            };
            // kWin windows per iteration: their lookups and candidate loads are in flight together
            for (int32_t f0 = 0; f0 < total; f0 += kWin * kWave) {
                int32_t src[kWin];
                float4 pv[kWin];
#pragma unroll
                for (int w = 0; w < kWin; ++w) src[w] = (w == 0 || f0 + w * kWave < total) ? locate(f0 + w * kWave, marks + w * kWave) : 0;
#pragma unroll
                for (int w = 0; w < kWin; ++w) {
                    pv[w] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (f0 + w * kWave + lane < total) pv[w] = sorted[src[w]];
                }
#pragma unroll
                for (int w = 0; w < kWin; ++w)
                    if (w == 0 || f0 + w * kWave < total) test(f0 + w * kWave, pv[w]);
            }
        }
    }
    return cnt;
}

// One wavefront per query.  WRITE=false: counts[q] = number of hits.  WRITE=true: fill the CSR row.  qflags (with an
// open3d visibility flag): EACH pass marks the queries whose row may hold a point the reference cannot see and leaves them to
// the frs_fix launch behind it -- the write pass does not rely on the count pass's marks (another search on the same
// structure may have run in between); it never writes beyond the row's exact length, which the count pass put in row_splits.
template <bool WRITE>
__global__ __launch_bounds__(256) void frs_query(const float* __restrict__ queries, int64_t m,
                                                 const FrsHeader* __restrict__ h,
                                                 const uint32_t* __restrict__ cell_start,
                                                 const float4* __restrict__ sorted, float radius, int flags,
                                                 int32_t* __restrict__ counts, const int64_t* __restrict__ row_splits,
                                                 int32_t* __restrict__ nbr_index, float* __restrict__ nbr_dist,
                                                 int64_t capacity, uint8_t* __restrict__ qflags) {
    __shared__ uint32_t marks[4][kWin * kWave + kWave];
    const int64_t qi = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= m) return;  // whole wave leaves
    // a row that does not fit the caller's buffers is skipped as a whole (the caller detects the overflow from
    // row_splits[m] > capacity and repeats the search with exact buffers)
    if (WRITE && row_splits[qi + 1] > capacity) return;
    const bool o3d = (flags & kO3dFlags) != 0;
    const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
    float unused = 0.0f;
    bool redo = false;
    // (a row the reference sees less of than the distance test holds fewer entries than this scan finds: capped here, rewritten
    // by frs_fix)
    const int32_t cap = WRITE && o3d ? (int32_t)(row_splits[qi + 1] - row_splits[qi]) : 0x7fffffff;
    const int32_t cnt = frs_scan<WRITE ? 1 : 0, false>(qx, qy, qz, h, cell_start, sorted, radius, flags, WRITE ? row_splits[qi] : 0,
                                                       nbr_index, nbr_dist, 0, 0.0f, unused, marks[threadIdx.x >> 6], cap, redo);
    if (lane_id() == 0) {
        if (!WRITE) counts[qi] = cnt;
        if (o3d) qflags[qi] = redo ? 1 : 0;
    }
}

// Single pass into padded rows (see dmcf_frs_search_padded): row qi starts at qi * stride.
__global__ __launch_bounds__(256) void frs_query_padded(const float* __restrict__ queries, int64_t m,
                                                        const FrsHeader* __restrict__ h, const uint32_t* __restrict__ cell_start,
                                                        const float4* __restrict__ sorted, float radius, int flags,
                                                        int64_t stride, int64_t* __restrict__ row_begin,
                                                        int32_t* __restrict__ row_count, int32_t* __restrict__ nbr_index,
                                                        float* __restrict__ nbr_dist, int32_t* __restrict__ max_count,
                                                        uint8_t* __restrict__ qflags) {
    __shared__ uint32_t marks[4][kWin * kWave + kWave];
    const int64_t qi = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= m) return;
    const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
    float unused = 0.0f;
    bool redo = false;
    const int32_t cnt = frs_scan<1, false>(qx, qy, qz, h, cell_start, sorted, radius, flags, qi * stride, nbr_index, nbr_dist, 0, 0.0f,
                                           unused, marks[threadIdx.x >> 6], (int32_t)min(stride, (int64_t)0x7fffffff), redo);
    if (lane_id() == 0) {
        row_begin[qi] = qi * stride;
        if (qi == m - 1) row_begin[m] = m * stride;
        row_count[qi] = (int32_t)min((int64_t)cnt, stride);
        if (flags & kO3dFlags) qflags[qi] = redo ? 1 : 0;
        // same-address atomics serialise (1.1M of them cost ~3 ms per search): only rows that beat the value currently
        // visible try; a stale read merely costs a redundant atomic.  (A flagged row may shrink in frs_fix: its first count
        // still bounds it.)
        if (cnt > __hip_atomic_load(max_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(max_count, cnt);
    }
}

// compute_density (utils/tools/losses.py:285-306) without the pair list: out[q] = sum over the points within R of
// window(|x - q|^2 / R^2).  Same candidate scan as the search; nothing but the sums leaves the kernel.
__global__ __launch_bounds__(256) void frs_window_sum(const float* __restrict__ queries, int64_t m,
                                                      const FrsHeader* __restrict__ h, const uint32_t* __restrict__ cell_start,
                                                      const float4* __restrict__ sorted, float radius, int flags, int window,
                                                      float* __restrict__ out, uint8_t* __restrict__ qflags) {
    __shared__ uint32_t marks[4][kWin * kWave + kWave];
    const int64_t qi = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (qi >= m) return;
    const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
    float wsum = 0.0f;
    bool redo = false;
    frs_scan<2, false>(qx, qy, qz, h, cell_start, sorted, radius, flags, 0, nullptr, nullptr, window, 1.0f / (radius * radius), wsum,
                       marks[threadIdx.x >> 6], 0x7fffffff, redo);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d, kWave);
    if (lane_id() == 0) {
        out[qi] = wsum;
        if (flags & kO3dFlags) qflags[qi] = redo ? 1 : 0;
    }
}

// The FIXUP of the open3d visibility flags: the hot kernels above return the set of the distance test and flag the queries
// whose row holds a hit the reference might not see (one whose voxel could lie outside the 8 corner voxels: the outermost
// shell of the sphere, or any hit of a query whose corner voxels are two apart).  A small persistent grid walks the flags and
// scans each flagged query again with the exact test: KIND 0 recounts (counts[q]), 1 writes the CSR row, 2 rewrites the
// padded row and its count, 3 recomputes the window sum.
template <int KIND>
__global__ __launch_bounds__(256) void frs_fix(const float* __restrict__ queries, int64_t m, const FrsHeader* __restrict__ h,
                                               const uint32_t* __restrict__ cell_start, const float4* __restrict__ sorted,
                                               float radius, int flags, const uint8_t* __restrict__ qflags,
                                               int32_t* __restrict__ counts, const int64_t* __restrict__ row_splits, int64_t stride,
                                               int32_t* __restrict__ nbr_index, float* __restrict__ nbr_dist, int64_t capacity,
                                               int window, float* __restrict__ out) {
    __shared__ uint32_t marks[4][kWin * kWave + kWave];
    const int lane = lane_id();
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t base = wave * kWave; base < m; base += nwaves * kWave) {
        const int64_t mine = base + lane;
        unsigned long long todo = __ballot(mine < m && qflags[mine] != 0);
        while (todo) {
            const int k = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int64_t qi = base + k;
            if (KIND == 1 && row_splits[qi + 1] > capacity) continue;
            const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
            float wsum = 0.0f;
            bool redo = false;
            const int64_t ob = KIND == 1 ? row_splits[qi] : (KIND == 2 ? qi * stride : 0);
            const int32_t cap = KIND == 2 ? (int32_t)min(stride, (int64_t)0x7fffffff) : 0x7fffffff;
            const int32_t cnt = frs_scan<KIND == 0 ? 0 : (KIND == 3 ? 2 : 1), true>(qx, qy, qz, h, cell_start, sorted, radius, flags, ob, nbr_index,
                                                                                   nbr_dist, window, 1.0f / (radius * radius), wsum,
                                                                                   marks[threadIdx.x >> 6], cap, redo);
            if (KIND == 3) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) wsum += __shfl_xor(wsum, d, kWave);
            }
            if (lane == 0) {
                if (KIND == 0) counts[qi] = cnt;
                if (KIND == 2) counts[qi] = (int32_t)min((int64_t)cnt, stride);
                if (KIND == 3) out[qi] = wsum;
            }
        }
    }
}

}  // namespace dmcf

using namespace dmcf;

static bool frs_flags_ok(int flags) {  // known bits, and at most one reading of the reference's walk
    return (flags & ~(DMCF_FRS_IGNORE_QUERY_POINT | kO3dFlags)) == 0 && (flags & kO3dFlags) != kO3dFlags;
}

static constexpr unsigned kFixGrid = 1024;  // 4096 waves walk the query flags (frs_fix)

extern "C" {

size_t dmcf_frs_workspace_bytes(int64_t n_points, int64_t n_queries) {
    if (n_points < 0 || n_queries < 0) return 0;
    return frs_layout(n_points, n_queries).total;
}

int dmcf_frs_build(const float* points, int64_t n, float radius, void* workspace, size_t workspace_bytes,
                   dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (n < 0 || n > 0x7fffffff || !workspace || !(radius > 0.0f) || (n > 0 && !points)) return DMCF_EINVAL;
    if (((uintptr_t)workspace & 255) != 0) return DMCF_EINVAL;
    // the layout does not depend on the number of queries up to off_counts; validate with m = 0
    const FrsLayout L = frs_layout(n, 0);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    char* ws = (char*)workspace;
    FrsHeader* h = (FrsHeader*)(ws + L.off_header);
    uint32_t* cell_start = (uint32_t*)(ws + L.off_cell_start);
    uint32_t* cell_fill = (uint32_t*)(ws + L.off_cell_fill);
    int32_t* point_cell = (int32_t*)(ws + L.off_point_cell);
    int32_t* tmp_idx = (int32_t*)(ws + L.off_tmp_idx);
    float4* sorted = (float4*)(ws + L.off_sorted);

    if (n <= kSmallBuildMax && !getenv("DMCF_FRS_NO_SMALL_BUILD")) {
        hipLaunchKernelGGL(frs_build_small, dim3(1), dim3(kSmallThreads), 0, stream, points, n, radius, h, L.table, cell_start, cell_fill,
                           point_cell, tmp_idx, sorted);
        return check_launch();
    }
    hipLaunchKernelGGL(frs_init_header, dim3(1), dim3(64), 0, stream, h, radius, (int32_t)n);
    if (n > 0) {
        const unsigned g = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(frs_bbox, dim3(g < 512u ? g : 512u), dim3(256), 0, stream, points, n, h);
    }
    hipLaunchKernelGGL(frs_finish_header, dim3(1), dim3(64), 0, stream, h, L.table);
    // cell_fill is the histogram: count (each point keeps the counter's old value) -> scan into cell_start -> scatter
    if (hipMemsetAsync(cell_fill, 0, (size_t)(L.table + 1) * 4, stream) != hipSuccess) return DMCF_ELAUNCH;
    if (n > 0) {
        const unsigned g = (unsigned)((n + 255) / 256);
        // (the places inside the cells wait in the sorted array's memory, which is written last)
        hipLaunchKernelGGL(frs_count_cells, dim3(g), dim3(256), 0, stream, points, n, h, point_cell, cell_fill, (uint32_t*)sorted);
    }
    // the scan scratch sits after the (query-count dependent) counts array; during the build nothing
    // else lives there, so use the tail of the workspace the caller actually gave us
    const size_t scan_need = scan_tmp_bytes(L.table + 1);
    if (workspace_bytes < L.off_counts + scan_need) return DMCF_EWORKSPACE;
    int rc = scan_exclusive_u32(cell_fill, cell_start, L.table + 1, ws + workspace_bytes - scan_need, scan_need, stream);
    if (rc != DMCF_OK) return rc;
    if (n > 0) {
        const unsigned g = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(frs_scatter, dim3(g), dim3(256), 0, stream, n, point_cell, cell_start, (const uint32_t*)sorted, tmp_idx);
        hipLaunchKernelGGL(frs_rank_and_place, dim3(g), dim3(256), 0, stream, points, n, point_cell, cell_start,
                           tmp_idx, sorted);
    }
    return check_launch();
}

int dmcf_frs_count(const float* queries, int64_t m, int64_t n, float radius, int flags, void* workspace,
                   size_t workspace_bytes, int64_t* row_splits, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!frs_flags_ok(flags)) return DMCF_EINVAL;
    if (m < 0 || n < 0 || !workspace || !(radius > 0.0f) || !row_splits || (m > 0 && !queries)) return DMCF_EINVAL;
    const FrsLayout L = frs_layout(n, m);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    char* ws = (char*)workspace;
    const FrsHeader* h = (const FrsHeader*)(ws + L.off_header);
    const uint32_t* cell_start = (const uint32_t*)(ws + L.off_cell_start);
    const float4* sorted = (const float4*)(ws + L.off_sorted);
    int32_t* counts = (int32_t*)(ws + L.off_counts);
    if (m > 0) {
        const unsigned g = (unsigned)((m + 3) / 4);
        uint8_t* qflags = (uint8_t*)(ws + L.off_flags);
        hipLaunchKernelGGL((frs_query<false>), dim3(g), dim3(256), 0, stream, queries, m, h, cell_start, sorted,
                           radius, flags, counts, (const int64_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, (int64_t)0, qflags);
        if (flags & kO3dFlags)
            hipLaunchKernelGGL((frs_fix<0>), dim3(kFixGrid), dim3(256), 0, stream, queries, m, h, cell_start, sorted, radius, flags,
                               (const uint8_t*)qflags, counts, (const int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr, (float*)nullptr,
                               (int64_t)0, 0, (float*)nullptr);
    }
    return scan_counts_to_row_splits(counts, row_splits, m, ws + L.off_scan, L.scan_bytes, stream);
}

int dmcf_frs_write(const float* queries, int64_t m, int64_t n, float radius, int flags, const void* workspace,
                   size_t workspace_bytes, const int64_t* row_splits, int32_t* neighbors_index,
                   float* neighbors_distance, int64_t pair_capacity, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!frs_flags_ok(flags)) return DMCF_EINVAL;
    if (m < 0 || n < 0 || !workspace || !(radius > 0.0f) || !row_splits || (m > 0 && !queries)) return DMCF_EINVAL;
    if (m == 0) return DMCF_OK;
    if (!neighbors_index || pair_capacity < 0) return DMCF_EINVAL;
    const FrsLayout L = frs_layout(n, m);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    char* ws = (char*)workspace;  // (the query flags live in it: see frs_layout)
    const FrsHeader* h = (const FrsHeader*)(ws + L.off_header);
    const uint32_t* cell_start = (const uint32_t*)(ws + L.off_cell_start);
    const float4* sorted = (const float4*)(ws + L.off_sorted);
    const unsigned g = (unsigned)((m + 3) / 4);
    uint8_t* qflags = (uint8_t*)(ws + L.off_flags);  // (written by the count pass of this search)
    hipLaunchKernelGGL((frs_query<true>), dim3(g), dim3(256), 0, stream, queries, m, h, cell_start, sorted, radius,
                       flags, (int32_t*)nullptr, row_splits, neighbors_index, neighbors_distance, pair_capacity, qflags);
    if (flags & kO3dFlags)
        hipLaunchKernelGGL((frs_fix<1>), dim3(kFixGrid), dim3(256), 0, stream, queries, m, h, cell_start, sorted, radius, flags,
                           (const uint8_t*)qflags, (int32_t*)nullptr, row_splits, (int64_t)0, neighbors_index, neighbors_distance,
                           pair_capacity, 0, (float*)nullptr);
    return check_launch();
}

int dmcf_frs_search_padded(const float* queries, int64_t m, int64_t n, float radius, int flags, const void* workspace,
                           size_t workspace_bytes, int64_t row_stride, int64_t* row_begin, int32_t* row_count,
                           int32_t* neighbors_index, float* neighbors_distance, int32_t* max_count, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!frs_flags_ok(flags)) return DMCF_EINVAL;
    if (m < 0 || n < 0 || !workspace || !(radius > 0.0f) || !row_begin || !max_count || (m > 0 && (!queries || !row_count)))
        return DMCF_EINVAL;
    if (row_stride < 0 || (m > 0 && row_stride > 0 && !neighbors_index)) return DMCF_EINVAL;
    const FrsLayout L = frs_layout(n, m);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    if (m == 0) return hipMemsetAsync(row_begin, 0, 8, stream) == hipSuccess ? DMCF_OK : DMCF_ELAUNCH;
    char* ws = (char*)workspace;  // (the query flags live in it: see frs_layout)
    const unsigned g = (unsigned)((m + 3) / 4);
    uint8_t* qflags = (uint8_t*)(ws + L.off_flags);
    const FrsHeader* h = (const FrsHeader*)(ws + L.off_header);
    const uint32_t* cell_start = (const uint32_t*)(ws + L.off_cell_start);
    const float4* sorted = (const float4*)(ws + L.off_sorted);
    hipLaunchKernelGGL(frs_query_padded, dim3(g), dim3(256), 0, stream, queries, m, h, cell_start, sorted, radius, flags, row_stride,
                       row_begin, row_count, neighbors_index, neighbors_distance, max_count, qflags);
    if (flags & kO3dFlags)
        hipLaunchKernelGGL((frs_fix<2>), dim3(kFixGrid), dim3(256), 0, stream, queries, m, h, cell_start, sorted, radius, flags,
                           (const uint8_t*)qflags, row_count, (const int64_t*)nullptr, row_stride, neighbors_index, neighbors_distance,
                           (int64_t)0, 0, (float*)nullptr);
    return check_launch();
}

int dmcf_frs_window_sum(const float* queries, int64_t m, int64_t n, float radius, int flags, int window,
                        const void* workspace, size_t workspace_bytes, float* out, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!frs_flags_ok(flags)) return DMCF_EINVAL;
    if (m < 0 || n < 0 || !workspace || !(radius > 0.0f) || (m > 0 && (!queries || !out))) return DMCF_EINVAL;
    if (window < DMCF_WINDOW_NONE || window > DMCF_WINDOW_CUBIC_GRAD) return DMCF_EINVAL;
    if (m == 0) return DMCF_OK;
    const FrsLayout L = frs_layout(n, m);
    if (workspace_bytes < L.total) return DMCF_EWORKSPACE;
    char* ws = (char*)workspace;  // (the query flags live in it: see frs_layout)
    const unsigned g = (unsigned)((m + 3) / 4);
    uint8_t* qflags = (uint8_t*)(ws + L.off_flags);
    const FrsHeader* h = (const FrsHeader*)(ws + L.off_header);
    const uint32_t* cell_start = (const uint32_t*)(ws + L.off_cell_start);
    const float4* sorted = (const float4*)(ws + L.off_sorted);
    hipLaunchKernelGGL(frs_window_sum, dim3(g), dim3(256), 0, stream, queries, m, h, cell_start, sorted, radius, flags, window, out, qflags);
    if (flags & kO3dFlags)
        hipLaunchKernelGGL((frs_fix<3>), dim3(kFixGrid), dim3(256), 0, stream, queries, m, h, cell_start, sorted, radius, flags,
                           (const uint8_t*)qflags, (int32_t*)nullptr, (const int64_t*)nullptr, (int64_t)0, (int32_t*)nullptr,
                           (float*)nullptr, (int64_t)0, window, out);
    return check_launch();
}

}  // extern "C"
