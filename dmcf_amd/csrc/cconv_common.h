// Device helpers shared by the two CConv kernels (cconv.hip: LDS read-modify-write splat, any filter size;
// cconv_mfma.hip: matrix-core splat for filters of up to 64 cells).  Not part of the C ABI.
#pragma once
#include "common.h"

namespace dmcf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct CconvParams {
    const float* Wp;  // packed filter, see pack_filter
    int sx, sy, sz, K, cin, cout;
    const float* out_pos;
    const float* inp_pos;
    const float* inp_feat;
    const float* inp_imp;
    const int32_t* idx;
    const int64_t* rs;
    const int32_t* cnt;    // optional pairs per row (padded lists: row i = [rs[i], rs[i] + cnt[i])); NULL = CSR
    const float* nval;
    int64_t n_out;
    int64_t n_inp;     // rows of inp_pos / inp_feat (the bound of the kernels' buffer resources)
    int64_t pair_cap;  // entries the neighbour buffers hold; rows reaching past it are treated as empty (a search
                       // enqueued with estimated sizes skipped them, the caller repeats the step)
    float inv_extent, inv_r2, window_fac;
    int window, mapping, interp, flags;
    const float* bias;
    float* out;
    int PS;        // plane stride of B in floats (see cell_offset)
    int zgroup;    // LDS splat: corner bits of the phase-2 lanes are (x, z, y) instead of (x, y, z)
    int KCp;       // row stride of B in floats: sz*PS padded to 4 (mod 64)
    int nblocks;   // sz*PS/16 : 16-wide k blocks per chunk
    int NT;        // ceil(cout/16)
    int nchunks;   // ceil(cin/CC)
    uint32_t wmask;  // filter blocks worth multiplying: bit 4 * (channel quad) + (16-column tile); all ones without a hint
    int bfloats;   // floats reserved for B / the reduction buffer (whichever is larger)
    int ntiles, tiles_per_xcd;
    int KT;        // matrix-core splat only: number of 16-cell tiles (ceil(K/16))
    float* partial;  // matrix-core splat, small launches: [nchunks][n_out][cout] partial sums, one channel chunk per workgroup row
    int csplit;      // ... 1: blockIdx.y = the workgroup's chunk (cconv_mfma.hip, kSplitMaxOut)
};

// ---- per-pair math (float restatement of Open3D's CoordinateTransformation.h, see oracle/dmcf_oracle.c).
// The splat's phase 1 is VALU-issue bound (measured: ~47 % of the kernel), so divisions and square roots
// use the 1-ulp hardware approximations (v_rcp_f32 / v_sqrt_f32 / v_rsq_f32) instead of the IEEE
// sequences (~10 instructions each) and atan -- only ever called with |t| <= 1 -- is a degree-17 odd
// polynomial (max error 1.1e-7).  Filter coordinates move by ~1e-7 relative against libm; the parity bar on
// CConv outputs is 1e-5 and the neighbour SETS are decided elsewhere (frs.hip, exact arithmetic).
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }

__device__ __forceinline__ float atan_unit(float t) {  // |t| <= 1
    const float u = t * t;
    float p = 0.0024567211512476206f;
    p = fmaf(p, u, -0.01440134271979332f);
    p = fmaf(p, u, 0.039781197905540466f);
    p = fmaf(p, u, -0.07234854996204376f);
    p = fmaf(p, u, 0.1049894466996193f);
    p = fmaf(p, u, -0.14161229133605957f);
    p = fmaf(p, u, 0.19985906779766083f);
    p = fmaf(p, u, -0.33332598209381104f);
    p = fmaf(p, u, 0.9999998807907104f);
    return t * p;
}

__device__ __forceinline__ void sphere_to_cyl(float& x, float& y, float& z) {
    const float rho2 = x * x + y * y;
    const float sq_norm = rho2 + z * z;
    const float norm = fast_sqrt(sq_norm);
    const bool polar = 1.25f * z * z > rho2;
    // polar cap: s = sqrt(3 norm / (norm + |z|)), z' = sign(z) norm;  belt: s = norm / rho, z' = 3/2 z
    const float s_cap = fast_sqrt(3.0f * norm * fast_rcp(norm + fabsf(z)));
    const float s_belt = norm * fast_rsq(rho2);
    const float s = polar ? s_cap : s_belt;
    const float zz = polar ? copysignf(norm, z) : 1.5f * z;
    const bool tiny = sq_norm < 1e-12f;
    x = tiny ? 0.0f : x * s;
    y = tiny ? 0.0f : y * s;
    z = tiny ? 0.0f : zz;
}

__device__ __forceinline__ void cyl_to_cube(float& x, float& y) {
    const float sq_norm = x * x + y * y;
    const float norm = fast_sqrt(sq_norm);
    const float four_over_pi = 1.2732395447351628f;
    const bool xmajor = fabsf(y) <= fabsf(x);
    const float num = xmajor ? y : x, den = xmajor ? x : y;
    const float tmp = copysignf(norm, den);
    const float other = tmp * four_over_pi * atan_unit(num * fast_rcp(den));
    const bool tiny = sq_norm < 1e-12f;
    const float nx = xmajor ? tmp : other, ny = xmajor ? other : tmp;
    x = tiny ? 0.0f : nx;
    y = tiny ? 0.0f : ny;
}

template <bool GENERIC>
__device__ __forceinline__ void filter_coords(float& x, float& y, float& z, const CconvParams& p) {
    if (!GENERIC || p.mapping == DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING) {
        const float s = 2.0f * p.inv_extent;
        x *= s; y *= s; z *= s;
        sphere_to_cyl(x, y, z);
        cyl_to_cube(x, y);
        x *= 0.5f; y *= 0.5f; z *= 0.5f;
    } else if (p.mapping == DMCF_MAP_BALL_TO_CUBE_RADIAL) {
        const float s = 2.0f * p.inv_extent;
        x *= s; y *= s; z *= s;
        const float radius = fast_sqrt(x * x + y * y + z * z);
        const float abs_max = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        const float k = abs_max < 1e-8f ? 0.0f : 0.5f * radius * fast_rcp(abs_max);
        x *= k; y *= k; z *= k;
    } else {
        x *= p.inv_extent; y *= p.inv_extent; z *= p.inv_extent;
    }
    if (!GENERIC || (p.flags & DMCF_FLAG_ALIGN_CORNERS)) {
        x = (x + 0.5f) * (float)(p.sx - 1);
        y = (y + 0.5f) * (float)(p.sy - 1);
        z = (z + 0.5f) * (float)(p.sz - 1);
    } else {
        x = x * (float)p.sx + (float)(p.sx / 2);
        y = y * (float)p.sy + (float)(p.sy / 2);
        z = z * (float)p.sz + (float)(p.sz / 2);
        if (p.sx % 2 == 0) x -= 0.5f;
        if (p.sy % 2 == 0) y -= 0.5f;
        if (p.sz % 2 == 0) z -= 0.5f;
    }
}

// Squared length of a pair's relative position x_in - x_out, bit for bit what the search returns for the pair (frs.hip:
// dist2_unfused of the same two positions).  neighbors_value == NULL with a distance window: the kernels evaluate the window
// on this, and the lists need no distance array (half the bytes the search writes and one load stream less here).
__device__ __forceinline__ float rel_dist2(float x, float y, float z) {
    return __fadd_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)), __fmul_rn(z, z));
}

// window functions of utils/tools/losses.py:8-44 on q = d^2 / R^2
__device__ __forceinline__ float window_value(int window, float v, float inv_r2, float fac) {
    if (window == DMCF_WINDOW_NONE) return 1.0f;
    if (window == DMCF_WINDOW_EXPLICIT) return v;
    const float q = v * inv_r2;
    switch (window) {
        case DMCF_WINDOW_POLY6: {
            const float t = 1.0f - q;
            return fac * fminf(fmaxf(t * t * t, 0.0f), 1.0f);
        }
        case DMCF_WINDOW_CUBIC: {
            const float s = fast_sqrt(q);
            float r = 0.0f;
            if (q <= 1.0f) r = (s <= 0.5f) ? 6.0f * (s * s * s - q) + 1.0f : 2.0f * (1.0f - s) * (1.0f - s) * (1.0f - s);
            return fac * (4.0f / 3.0f) * r;
        }
        case DMCF_WINDOW_LINEAR: return fac * (1.0f - fast_sqrt(q));
        case DMCF_WINDOW_PEAK: return fac * (1.0f - 2.0f * fast_sqrt(q) + q);
        case DMCF_WINDOW_CUBIC_GRAD: {
            const float s = fast_sqrt(q);
            float r = 0.0f;
            if (q <= 1.0f) r = (s <= 0.5f) ? 18.0f * q - 12.0f * s : -6.0f * (1.0f - s) * (1.0f - s);
            return fac * (4.0f / 3.0f) * r;
        }
    }
    return 1.0f;
}

// Interpolation along one axis as (base cell b, weight of b, weight of b+1) with 0 <= b <= max(s-2, 0),
// so that the two cells are always inside the filter array (for s == 1 the second weight is 0 and
// the lane offsets of "b+1" collapse onto b).  Equivalent to Open3D's clamped / bordered / nearest
// lookups: weights that the library would put on a clamped duplicate or outside cell are 0 here.
__device__ __forceinline__ void axis_weights(float x, int s, int interp, int& b, float& w0, float& w1) {
    const int bmax = s >= 2 ? s - 2 : 0;
    if (interp == DMCF_INTERP_NEAREST) {
        int c = (int)roundf(x);
        c = min(max(c, 0), s - 1);
        b = min(c, bmax);
        w0 = (c == b) ? 1.0f : 0.0f;
        w1 = 1.0f - w0;
        return;
    }
    if (interp == DMCF_INTERP_LINEAR) {  // coordinate clamping
        x = fminf((float)(s - 1), fmaxf(0.0f, x));
        const float xf = fminf(floorf(x), (float)bmax);
        b = (int)xf;
        const float a = x - xf;  // in [0,1]; == 1 exactly when x == s-1 (then all weight on cell s-1)
        w0 = 1.0f - a;
        w1 = a;
        if (s == 1) { w0 = 1.0f; w1 = 0.0f; }
        return;
    }
    // LINEAR_BORDER: cells xf and xf+1 with weights (1-a, a); cells outside [0, s-1] contribute nothing
    const float xf = floorf(x);
    const float a = x - xf;
    const float c0 = xf, c1 = xf + 1.0f;
    const bool in0 = c0 >= 0.0f && c0 <= (float)(s - 1), in1 = c1 >= 0.0f && c1 <= (float)(s - 1);
    const float v0 = in0 ? 1.0f - a : 0.0f, v1 = in1 ? a : 0.0f;
    if (!in0 && !in1) { b = 0; w0 = w1 = 0.0f; return; }
    if (in0 && in1) { b = (int)c0; w0 = v0; w1 = v1; return; }  // then c0 <= s-2
    if (in0) {  // c0 == s-1, c1 outside
        if (s >= 2) { b = s - 2; w0 = 0.0f; w1 = v0; } else { b = 0; w0 = v0; w1 = 0.0f; }
        return;
    }
    // in1 only: c1 == 0, c0 == -1
    b = 0; w0 = v1; w1 = 0.0f;
}

// INTERP_LINEAR only, branch free (the non-GENERIC instantiation)
__device__ __forceinline__ void axis_weights_linear(float x, int s, int& b, float& w0, float& w1) {
    const float bmax = (float)(s >= 2 ? s - 2 : 0);
    x = fminf((float)(s - 1), fmaxf(0.0f, x));
    const float xf = fminf(floorf(x), bmax);
    b = (int)xf;
    const float a = x - xf;
    w0 = 1.0f - a;
    w1 = a;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// t / nq and t % nq for 0 <= t < 64 and nq in 1 .. 4 (k' blocks of the contraction: nq = channel quads of the chunk) without an
// integer division -- a generic one is ~25 instructions, and the contraction loops form it per block and wave: 3 % of a tile of
// the 33-pair layers.  (t * m) >> 7 with m = 128 / nq rounded up is exact below 64.
__device__ __forceinline__ void blk_divmod(int t, int nq, int& quot, int& rem) {
    const int m = nq == 1 ? 128 : (nq == 2 ? 64 : (nq == 3 ? 43 : 32));
    quot = (t * m) >> 7;
    rem = t - quot * nq;
}



// defined in cconv.hip, also used by the matrix-core path
__global__ void pack_filter(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int d2, int cin,
                            int cout, int CC, int PS, int nchunks, int nblocks, int NT, int symmetric, int sym_axis);

// cconv_mfma.hip
bool cconv_mfma_eligible(int K, int cin, int cout);
size_t cconv_mfma_packed_floats(int K, int cin, int cout);
size_t cconv_mfma_partial_floats(int K, int cin, int cout, int64_t n_out);  // (0: the launch does not split its chunks)
int cconv_mfma_launch(CconvParams p, const dmcf_cconv_args* a, int dz, int dy, int dx, void* workspace, hipStream_t stream);


// cconv_blk.hip
bool cconv_blk_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx);
size_t cconv_blk_packed_floats(int cin, int cout);
int cconv_blk_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream);


// cconv_cls.hip
bool cconv_cls_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx);
size_t cconv_cls_packed_floats(int cin, int cout);
int cconv_cls_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream);
int cconv_cls_pack(const dmcf_cconv_args* a, float* packed, hipStream_t stream);  // enqueues the packing, returns the chunk count
// DMCF_FLAG_ACCUMULATE: the value the epilogue adds to, requested EARLY (before the contraction): read in the epilogue itself it is
// a dependent round trip at the very end of every tile, with nothing left to hide it -- as long as the elementwise kernel it
// replaces.  A tile's outputs are contiguous (point-major): element e = point * cout + channel sits at out + pt0 * cout + e; this
// is the element of the thread's first epilogue iteration (e = tid), later iterations read theirs in place.
__device__ __forceinline__ float epilogue_prefetch(const CconvParams& p, int64_t pt0, int pts, int tid) {
    if (!(p.flags & DMCF_FLAG_ACCUMULATE)) return 0.0f;
    const int64_t lim = min((int64_t)pts, p.n_out - pt0) * p.cout;
    return tid < lim ? p.out[pt0 * p.cout + tid] : 0.0f;
}

// The "plain" layer: poly6 window on squared distances re-formed from the positions, no per-point importance -- every CConv of
// the networks here once the lists carry no distances.  Splats D and E have instantiations with these three choices compiled
// in: the window's branch ladder, the distance / importance loads and their predicates otherwise run once per batch.
inline bool cconv_plain(const dmcf_cconv_args* a) {
    return a->window == DMCF_WINDOW_POLY6 && !a->neighbors_value && !a->inp_importance;
}

// cconv_z3.hip
bool cconv_z3_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx);
int cconv_z3_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream);


// cconv_pair.hip
bool cconv_pair_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx);
int cconv_pair_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream);


// cconv_ws.hip
bool cconv_ws_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx);
int cconv_ws_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream);


// cconv_p16.hip
bool cconv_p16_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx);
int cconv_p16_launch(CconvParams p, const dmcf_cconv_args* a, void* workspace, hipStream_t stream);


// cconv_direct.hip
bool cconv_direct_eligible(const dmcf_cconv_args* a, int dz, int dy, int dx);
size_t cconv_direct_packed_floats(int dz, int dy, int dx, int cin);
int cconv_direct_launch(CconvParams p, const dmcf_cconv_args* a, int dz, int dy, int dx, void* workspace, hipStream_t stream);

}  // namespace dmcf
