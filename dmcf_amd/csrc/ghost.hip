// Ghost selection of the block-sharded rollout (dmcf_amd/parallel.py): which of a rank's points lie within
// a halo width of which peer block -- for ALL the widths a step needs at once.
//
// The host form of this (GhostPlan, round 1 - 4) is ~40 torch calls per plan: masks of the points near a face, squared gaps to the
// peers' blocks, nonzero / bincount / gathers, once for the set's widest plan and again for every narrower plan derived from it:
// ~600 small launches per rank and step that the host paces while the GPU waits.  Here it is a count and a
// write kernel per point set:
//
//   level(i, b) = the number of widths w (given in DESCENDING order) with gap2(point i, box b) <= width2[w]
//   list(w)     = for b = 0 .. B - 1: the points with level(i, b) > w, in point order
//
// so list(0) is the widest plan's send list (grouped by peer, then by point -- the order the host form produced) and every list(w) a
// subset of it in the same order, which is what the derived plans were.  The receiving side runs the same two kernels over the copies
// it received, with ONE box (its own block): its list(w) are the positions of the narrower plans' ghosts inside the widest plan's.
// gap2 is evaluated by this code on both sides, on bit-identical coordinates: sender and receiver decide identically.
//
// Two phases because the sizes are data: count -> (the caller reads the totals, or knows them from the peers) -> write.
//
// OWNERSHIP is the same selection with another test (one "width", widths2[0] < 0): level(i, b) = 1 when lo <= x < hi on every axis
// -- BlockDecomposition.owner's half-open blocks -- so list(0) is the stable order by owner and totals[0] the rows per owner: the
// migration of a step without its stable sort (one block sort + ~10 merge passes over all particles).
#include "common.h"

namespace dmcf {

constexpr int kGhostThreads = 256;
constexpr int kGhostMaxBoxes = 64;
constexpr int kGhostMaxWidths = 8;

struct GhostParams {
    const float* pos;
    int64_t n;
    const float* boxes;  // [B][6]: lo x, y, z, hi x, y, z (+-inf on open sides)
    int B, W;
    float w2[kGhostMaxWidths];
    uint32_t* blk;       // [nblk][B][W]: per block counts, then (after ghost_scan) exclusive offsets inside the column
    int64_t* totals;     // [W][B]
    int64_t* starts;     // [W][B]: where box b's part of list(w) begins
    int64_t nblk;
};

__device__ __forceinline__ float ghost_gap2(float x, float y, float z, const float* b) {
    // (the arithmetic of BlockDecomposition.gap2: per axis max(lo - x, x - hi, 0), squared, (gx + gy) + gz -- un-fused)
    const float gx = fmaxf(fmaxf(__fsub_rn(b[0], x), __fsub_rn(x, b[3])), 0.0f);
    const float gy = fmaxf(fmaxf(__fsub_rn(b[1], y), __fsub_rn(y, b[4])), 0.0f);
    const float gz = fmaxf(fmaxf(__fsub_rn(b[2], z), __fsub_rn(z, b[5])), 0.0f);
    return __fadd_rn(__fadd_rn(__fmul_rn(gx, gx), __fmul_rn(gy, gy)), __fmul_rn(gz, gz));
}

__device__ __forceinline__ int ghost_level(float x, float y, float z, const float* b, const GhostParams& p) {
    if (p.w2[0] < 0.0f)  // ownership (a NaN or infinite coordinate has no owner: the caller compares the totals with n)
        return (x >= b[0] && x < b[3] && y >= b[1] && y < b[4] && z >= b[2] && z < b[5]) ? 1 : 0;
    const float g2 = ghost_gap2(x, y, z, b);
    int lv = 0;
#pragma unroll
    for (int w = 0; w < kGhostMaxWidths; ++w)
        if (w < p.W && g2 <= p.w2[w]) ++lv;  // (NaN gaps -- a NaN coordinate -- count as outside)
    return lv;
}

__global__ __launch_bounds__(kGhostThreads) void ghost_count(const GhostParams p) {
    __shared__ uint32_t cnt[kGhostMaxBoxes * kGhostMaxWidths];
    __shared__ float box[kGhostMaxBoxes * 6];
    for (int e = threadIdx.x; e < p.B * p.W; e += kGhostThreads) cnt[e] = 0;
    for (int e = threadIdx.x; e < p.B * 6; e += kGhostThreads) box[e] = p.boxes[e];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kGhostThreads + threadIdx.x;
    if (i < p.n) {
        const float x = p.pos[3 * i], y = p.pos[3 * i + 1], z = p.pos[3 * i + 2];
        for (int b = 0; b < p.B; ++b) {
            const int lv = ghost_level(x, y, z, box + 6 * b, p);
            for (int w = 0; w < lv; ++w) atomicAdd(&cnt[b * p.W + w], 1u);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < p.B * p.W; e += kGhostThreads) p.blk[(size_t)blockIdx.x * p.B * p.W + e] = cnt[e];
}

// one workgroup per column (b, w): exclusive prefix of the blocks' counts, the column's total
__global__ __launch_bounds__(kGhostThreads) void ghost_scan(const GhostParams p) {
    __shared__ uint32_t part[kGhostThreads];
    __shared__ uint64_t carry;
    const int col = blockIdx.x, b = col / p.W, w = col % p.W;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < p.nblk; base += kGhostThreads) {
        const int64_t k = base + threadIdx.x;
        const uint32_t v = k < p.nblk ? p.blk[(size_t)k * p.B * p.W + col] : 0u;
        part[threadIdx.x] = v;
        __syncthreads();
        // (256 entries: a Hillis-Steele scan in LDS)
        for (int d = 1; d < kGhostThreads; d <<= 1) {
            const uint32_t add = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        const uint64_t c0 = carry;
        if (k < p.nblk) p.blk[(size_t)k * p.B * p.W + col] = (uint32_t)(c0 + part[threadIdx.x] - v);
        __syncthreads();
        if (threadIdx.x == kGhostThreads - 1) carry = c0 + part[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.totals[(size_t)w * p.B + b] = (int64_t)carry;
}

// starts[w][b] = sum of totals[w][b' < b] (one thread per width)
__global__ void ghost_starts(const GhostParams p) {
    const int w = threadIdx.x;
    if (w >= p.W) return;
    int64_t s = 0;
    for (int b = 0; b < p.B; ++b) {
        p.starts[(size_t)w * p.B + b] = s;
        s += p.totals[(size_t)w * p.B + b];
    }
}

struct GhostWrite {
    int64_t* rows;                        // all lists in one buffer
    int64_t list_start[kGhostMaxWidths];  // where list(w) begins in it
    int64_t list_cap[kGhostMaxWidths];    // ... and how many entries it may hold (entries past it are dropped)
};

__global__ __launch_bounds__(kGhostThreads) void ghost_write(const GhostParams p, const GhostWrite o) {
    __shared__ float box[kGhostMaxBoxes * 6];
    __shared__ uint32_t wsum[kGhostMaxWidths][kGhostThreads / 64];
    for (int e = threadIdx.x; e < p.B * 6; e += kGhostThreads) box[e] = p.boxes[e];
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * kGhostThreads + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float x = 0.0f, y = 0.0f, z = 0.0f;
    if (i < p.n) {
        x = p.pos[3 * i];
        y = p.pos[3 * i + 1];
        z = p.pos[3 * i + 2];
    }
    for (int b = 0; b < p.B; ++b) {
        const int lv = i < p.n ? ghost_level(x, y, z, box + 6 * b, p) : 0;
        // stable rank inside the workgroup, per width: lanes below in the wave + the waves below
        uint32_t below[kGhostMaxWidths];
#pragma unroll
        for (int w = 0; w < kGhostMaxWidths; ++w) {
            if (w >= p.W) break;
            const uint64_t m = __ballot(lv > w);
            below[w] = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wsum[w][wave] = (uint32_t)__popcll(m);
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < kGhostMaxWidths; ++w) {
            if (w >= p.W) break;
            if (lv > w) {
                uint32_t r = below[w];
                for (int v = 0; v < wave; ++v) r += wsum[w][v];
                const int64_t at = p.starts[(size_t)w * p.B + b] + p.blk[(size_t)blockIdx.x * p.B * p.W + b * p.W + w] + r;
                if (at < o.list_cap[w]) o.rows[o.list_start[w] + at] = i;
            }
        }
        __syncthreads();
    }
}

static bool ghost_params(GhostParams& p, const float* pos, int64_t n, const float* boxes, int n_boxes, const float* widths2, int n_widths,
                         void* workspace, size_t workspace_bytes) {
    if (n < 0 || n_boxes < 1 || n_boxes > kGhostMaxBoxes || n_widths < 1 || n_widths > kGhostMaxWidths || !boxes || !widths2) return false;
    if (n > 0 && !pos) return false;
    for (int w = 0; w + 1 < n_widths; ++w)
        if (!(widths2[w] >= widths2[w + 1])) return false;  // descending: list(w + 1) is a subset of list(w)
    if (!(widths2[n_widths - 1] >= 0.0f) && n_widths != 1) return false;  // (negative: ownership, alone)
    p.pos = pos;
    p.n = n;
    p.boxes = boxes;
    p.B = n_boxes;
    p.W = n_widths;
    for (int w = 0; w < kGhostMaxWidths; ++w) p.w2[w] = w < n_widths ? widths2[w] : -1.0f;
    p.nblk = (n + kGhostThreads - 1) / kGhostThreads;
    const size_t need = align_up((size_t)p.nblk * n_boxes * n_widths * sizeof(uint32_t), 256) + 2 * (size_t)n_boxes * n_widths * sizeof(int64_t);
    if (!workspace || ((uintptr_t)workspace & 255) || workspace_bytes < need + 256) return false;
    p.blk = (uint32_t*)workspace;
    p.totals = (int64_t*)((char*)workspace + align_up((size_t)p.nblk * n_boxes * n_widths * sizeof(uint32_t), 256));
    p.starts = p.totals + (size_t)n_boxes * n_widths;
    return true;
}

}  // namespace dmcf

using namespace dmcf;

extern "C" {

size_t dmcf_ghost_workspace_bytes(int64_t n, int32_t n_boxes, int32_t n_widths) {
    if (n < 0 || n_boxes < 1 || n_widths < 1) return 256;
    const size_t nblk = (size_t)((n + kGhostThreads - 1) / kGhostThreads);
    return 512 + align_up(nblk * n_boxes * n_widths * sizeof(uint32_t), 256) + 2 * (size_t)n_boxes * n_widths * sizeof(int64_t);
}

int dmcf_ghost_count(const float* positions, int64_t n, const float* boxes, int32_t n_boxes, const float* widths2, int32_t n_widths,
                     int64_t* totals, void* workspace, size_t workspace_bytes, dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GhostParams p;
    if (!totals || !ghost_params(p, positions, n, boxes, n_boxes, widths2, n_widths, workspace, workspace_bytes)) return DMCF_EINVAL;
    if (p.nblk > 0x7fffffff) return DMCF_EUNSUPPORTED;
    if (p.nblk > 0) hipLaunchKernelGGL(ghost_count, dim3((unsigned)p.nblk), dim3(kGhostThreads), 0, stream, p);
    hipLaunchKernelGGL(ghost_scan, dim3((unsigned)(n_boxes * n_widths)), dim3(kGhostThreads), 0, stream, p);
    hipLaunchKernelGGL(ghost_starts, dim3(1), dim3(64), 0, stream, p);
    // the caller's copy of the totals, [W][B]
    hipError_t e = hipMemcpyAsync(totals, p.totals, (size_t)n_boxes * n_widths * sizeof(int64_t), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) {
        g_last_hip_error = (int)e;
        return DMCF_ELAUNCH;
    }
    return check_launch();
}

int dmcf_ghost_write(const float* positions, int64_t n, const float* boxes, int32_t n_boxes, const float* widths2, int32_t n_widths,
                     int64_t* rows, const int64_t* list_start, const int64_t* list_capacity, void* workspace, size_t workspace_bytes,
                     dmcf_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    GhostParams p;
    if (!list_start || !list_capacity || !ghost_params(p, positions, n, boxes, n_boxes, widths2, n_widths, workspace, workspace_bytes))
        return DMCF_EINVAL;
    GhostWrite o;
    o.rows = rows;
    int64_t any = 0;
    for (int w = 0; w < kGhostMaxWidths; ++w) {
        o.list_start[w] = w < n_widths ? list_start[w] : 0;
        o.list_cap[w] = w < n_widths ? list_capacity[w] : 0;
        if (o.list_start[w] < 0 || o.list_cap[w] < 0) return DMCF_EINVAL;
        any += o.list_cap[w];
    }
    if (any > 0 && !rows) return DMCF_EINVAL;
    if (p.nblk > 0 && any > 0) hipLaunchKernelGGL(ghost_write, dim3((unsigned)p.nblk), dim3(kGhostThreads), 0, stream, p, o);
    return check_launch();
}

}  // extern "C"
