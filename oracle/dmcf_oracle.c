/*
 * dmcf_oracle.c -- CPU ORACLE for the DMCF per-step hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing under dmcf_amd/ (the product) may import, link or call this file.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / the timed
 * CPU baseline.
 *
 * PARITY UNPINNED.  The arithmetic of this path does not live in /root/reference: it is in the
 * third-party dependency open3d==0.15.2 on tensorflow==2.5 (requirements.txt:1-2), which is
 * neither vendored nor installable here, and the reference ships no tests / golden vectors
 * (SURVEY.md section 8c).  This file restates the published Open3D 0.15.2 CPU algorithms
 *     open3d/core/nns/FixedRadiusSearchImpl.h, NeighborSearchCommon.h
 *     open3d/ml/impl/continuous_conv/ContinuousConv.h, CoordinateTransformation.h
 * from knowledge of that library, anchored on the reference's call sites:
 *     utils/convolutions.py:207-210,354-358   FixedRadiusSearch(metric, ignore_query_point, return_distances)
 *     utils/convolutions.py:414-431           continuous_conv(**_conv_values)
 *     utils/convolutions.py:433-458           ASCC second pass
 *     models/pbf_model.py:450-453             reduce_subarrays_sum
 * Every constant recalled from Open3D (hash primes, <= vs <, mapping formulas, clamping) sits
 * in this one file next to an "EXT" tag so that a later capture of true golden vectors
 * (tools/capture_golden.py, to be run on a TF2.5+Open3D0.15.2 machine) can confirm or correct it.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 * -ffp-contract=off keeps float arithmetic un-fused so the HIP path can be compared
 * bit-for-bit on the integer outputs (neighbour sets) that depend on float comparisons.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DMCF_REF_OK 0
#define DMCF_REF_EINVAL -1

/* ------------------------------------------------------------------------------------------
 * Fixed radius search (EXT: FixedRadiusSearchImpl.h / NeighborSearchCommon.h)
 * ---------------------------------------------------------------------------------------- */

/* EXT NeighborSearchCommon.h SpatialHash(): int arithmetic (wrapping), result converted to
 * size_t, i.e. sign-extended before the modulo. */
static inline uint64_t spatial_hash(int32_t x, int32_t y, int32_t z) {
    uint32_t h = ((uint32_t)x * 73856096u) ^ ((uint32_t)y * 193649663u) ^ ((uint32_t)z * 83492791u);
    return (uint64_t)(int64_t)(int32_t)h;
}

/* EXT ComputeVoxelIndex(): floor(pos * inv_voxel_size) cast to int. */
static inline void voxel_index(const float p[3], float inv_voxel_size, int32_t v[3]) {
    v[0] = (int32_t)floorf(p[0] * inv_voxel_size);
    v[1] = (int32_t)floorf(p[1] * inv_voxel_size);
    v[2] = (int32_t)floorf(p[2] * inv_voxel_size);
}

/* EXT FixedRadiusSearch layer: hash_table_size = clamp(factor * n, 1, 32*2^20), factor 1/64. */
int64_t dmcf_ref_hash_table_size(int64_t num_points, double hash_table_size_factor) {
    int64_t s = (int64_t)(hash_table_size_factor * (double)num_points);
    if (s < 1) s = 1;
    if (s > (int64_t)33554432) s = 33554432;
    return s;
}

/* EXT BuildSpatialHashTableCPU: count -> inclusive prefix sum -> scatter ids.  The library
 * scatters with atomics from a parallel loop (order inside a bin is unspecified); the oracle
 * uses ascending point id, which is one of the legal orders. */
int dmcf_ref_build_spatial_hash_table(const float* points, int64_t n, float radius,
                                      int64_t hash_table_size, uint32_t* cell_splits /*[size+1]*/,
                                      uint32_t* index /*[n]*/) {
    if (n < 0 || hash_table_size < 1 || !(radius > 0.0f)) return DMCF_REF_EINVAL;
    const float voxel_size = 2 * radius;
    const float inv_voxel_size = 1 / voxel_size;
    memset(cell_splits, 0, sizeof(uint32_t) * (size_t)(hash_table_size + 1));
    for (int64_t i = 0; i < n; ++i) {
        int32_t v[3];
        voxel_index(points + 3 * i, inv_voxel_size, v);
        uint64_t h = spatial_hash(v[0], v[1], v[2]) % (uint64_t)hash_table_size;
        cell_splits[h + 1] += 1;
    }
    for (int64_t h = 0; h < hash_table_size; ++h) cell_splits[h + 1] += cell_splits[h];
    uint32_t* fill = (uint32_t*)calloc((size_t)hash_table_size, sizeof(uint32_t));
    if (!fill) return DMCF_REF_EINVAL;
    for (int64_t i = 0; i < n; ++i) {
        int32_t v[3];
        voxel_index(points + 3 * i, inv_voxel_size, v);
        uint64_t h = spatial_hash(v[0], v[1], v[2]) % (uint64_t)hash_table_size;
        index[cell_splits[h] + fill[h]++] = (uint32_t)i;
    }
    free(fill);
    return DMCF_REF_OK;
}

/* Squared L2 distance, un-fused, ((dx*dx + dy*dy) + dz*dz).  EXT NeighborsDist<L2>. */
static inline float dist2(const float a[3], const float b[3]) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return (dx * dx + dy * dy) + dz * dz;
}

/* Which hash bins a query visits.  Two readings of EXT _FixedRadiusSearchCPU exist in this project (neither can be
 * checked against the library here; tools/capture_golden.py holds the inputs that tell them apart):
 *   DMCF_REF_BINS_OWN_AND_CORNERS (the DEFAULT): a std::set of bins that first receives the bin
 *       of the query's OWN voxel, floor(q / 2R), and then the bins of the 8 voxels holding the corners q +- R;
 *   DMCF_REF_BINS_CORNERS (round 3's reading): the 8 corner voxels only.
 * In exact arithmetic both are the same set of voxels (the own voxel is one of the corners') and cover the search sphere.
 * In float they differ from each other and from the sphere for a query a rounding step from the MIDDLE of a voxel on some
 * axis: floor(fl(q - R) / 2R) and floor(fl(q + R) / 2R) are then two apart.  CORNERS never visits the voxel between them
 * (the row comes out nearly empty); OWN_AND_CORNERS visits it, but still not the voxels that share the own voxel's
 * coordinate on that axis and a corner's coordinate on the others (the row loses whatever lies there). */
#define DMCF_REF_BINS_CORNERS 0
#define DMCF_REF_BINS_OWN_AND_CORNERS 1
/* NOT a reading of the library: all 27 voxels around the query's own one, which cover the sphere whatever the rounding --
 * the set of the distance test (== dmcf_ref_bruteforce_search) at the cost of a hash search.  What the product's default
 * search returns is checked against this at sizes the O(n m) loop cannot reach. */
#define DMCF_REF_BINS_ALL_27 2

/* EXT _FixedRadiusSearchCPU: for every query, the bins chosen by `bin_set` (above; deduplicated, visited in ascending
 * bin id), test dist <= threshold with
 * threshold = radius*radius (L2, inclusive), optional skip of points whose coordinates equal
 * the query's.  Two passes: nbr_index == NULL -> only counts are written to row_splits
 * (as an exclusive prefix sum, int64, length m+1); otherwise indices (+ squared distances).
 * Returns the total number of pairs, or a negative error. */
int64_t dmcf_ref_fixed_radius_search(const float* points, int64_t n, const float* queries, int64_t m,
                                     float radius, int ignore_query_point, int64_t hash_table_size,
                                     const uint32_t* cell_splits, const uint32_t* index,
                                     int64_t* row_splits, int32_t* nbr_index, float* nbr_dist, int bin_set) {
    if (n < 0 || m < 0 || hash_table_size < 1 || !(radius > 0.0f)) return DMCF_REF_EINVAL;
    if (bin_set < DMCF_REF_BINS_CORNERS || bin_set > DMCF_REF_BINS_ALL_27) return DMCF_REF_EINVAL;
    const float voxel_size = 2 * radius;
    const float inv_voxel_size = 1 / voxel_size;
    const float threshold = radius * radius;
    const int count_only = (nbr_index == NULL);
    if (count_only) row_splits[0] = 0;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t qi = 0; qi < m; ++qi) {
        const float* q = queries + 3 * qi;
        uint64_t bins[27];
        int nbins = 0;
        if (bin_set != DMCF_REF_BINS_CORNERS) {
            int32_t v[3];
            voxel_index(q, inv_voxel_size, v);
            bins[nbins++] = spatial_hash(v[0], v[1], v[2]) % (uint64_t)hash_table_size;
            if (bin_set == DMCF_REF_BINS_ALL_27)
                for (int dz = -1; dz <= 1; ++dz)
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            uint64_t h = spatial_hash(v[0] + dx, v[1] + dy, v[2] + dz) % (uint64_t)hash_table_size;
                            int k = 0;
                            while (k < nbins && bins[k] != h) ++k;
                            if (k == nbins) bins[nbins++] = h;
                        }
        }
        for (int dz = -1; dz <= 1 && bin_set != DMCF_REF_BINS_ALL_27; dz += 2)
            for (int dy = -1; dy <= 1; dy += 2)
                for (int dx = -1; dx <= 1; dx += 2) {
                    float p[3] = {q[0] + radius * (float)dx, q[1] + radius * (float)dy,
                                  q[2] + radius * (float)dz};
                    int32_t v[3];
                    voxel_index(p, inv_voxel_size, v);
                    uint64_t h = spatial_hash(v[0], v[1], v[2]) % (uint64_t)hash_table_size;
                    int k = 0;
                    while (k < nbins && bins[k] != h) ++k;
                    if (k == nbins) bins[nbins++] = h;
                }
        /* ascending bin id (std::set order) */
        for (int a = 1; a < nbins; ++a) {
            uint64_t t = bins[a];
            int b = a - 1;
            while (b >= 0 && bins[b] > t) { bins[b + 1] = bins[b]; --b; }
            bins[b + 1] = t;
        }
        int64_t cnt = 0;
        int64_t base = count_only ? 0 : row_splits[qi];
        for (int b = 0; b < nbins; ++b) {
            for (uint32_t s = cell_splits[bins[b]]; s < cell_splits[bins[b] + 1]; ++s) {
                uint32_t j = index[s];
                const float* p = points + 3 * (int64_t)j;
                if (ignore_query_point && p[0] == q[0] && p[1] == q[1] && p[2] == q[2]) continue;
                float d = dist2(p, q);
                if (d <= threshold) {
                    if (!count_only) {
                        nbr_index[base + cnt] = (int32_t)j;
                        if (nbr_dist) nbr_dist[base + cnt] = d;
                    }
                    ++cnt;
                }
            }
        }
        if (count_only) row_splits[qi + 1] = cnt;
    }
    if (count_only) {
        for (int64_t qi = 0; qi < m; ++qi) row_splits[qi + 1] += row_splits[qi];
    }
    return row_splits[m];
}

/* Brute force O(n*m) statement of the same contract, index-ascending rows.  Independent of
 * the hash constants; used to check both the restatement above and the HIP path. */
int64_t dmcf_ref_bruteforce_search(const float* points, int64_t n, const float* queries, int64_t m,
                                   float radius, int ignore_query_point, int64_t* row_splits,
                                   int32_t* nbr_index, float* nbr_dist) {
    const float threshold = radius * radius;
    const int count_only = (nbr_index == NULL);
    if (count_only) row_splits[0] = 0;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t qi = 0; qi < m; ++qi) {
        const float* q = queries + 3 * qi;
        int64_t cnt = 0;
        int64_t base = count_only ? 0 : row_splits[qi];
        for (int64_t j = 0; j < n; ++j) {
            const float* p = points + 3 * j;
            if (ignore_query_point && p[0] == q[0] && p[1] == q[1] && p[2] == q[2]) continue;
            float d = dist2(p, q);
            if (d <= threshold) {
                if (!count_only) {
                    nbr_index[base + cnt] = (int32_t)j;
                    if (nbr_dist) nbr_dist[base + cnt] = d;
                }
                ++cnt;
            }
        }
        if (count_only) row_splits[qi + 1] = cnt;
    }
    if (count_only)
        for (int64_t qi = 0; qi < m; ++qi) row_splits[qi + 1] += row_splits[qi];
    return row_splits[m];
}

/* EXT reduce_subarrays_sum (models/pbf_model.py:450): out[i] = sum(values[rs[i]:rs[i+1]]). */
int dmcf_ref_reduce_subarrays_sum(const float* values, const int64_t* row_splits, int64_t m,
                                  float* out) {
    for (int64_t i = 0; i < m; ++i) {
        float s = 0.0f;
        for (int64_t k = row_splits[i]; k < row_splits[i + 1]; ++k) s += values[k];
        out[i] = s;
    }
    return DMCF_REF_OK;
}

/* ------------------------------------------------------------------------------------------
 * Continuous convolution (EXT: ContinuousConv.h _CConvComputeFeaturesCPU,
 * CoordinateTransformation.h).  Written once as a macro body and instantiated for float
 * (the restatement) and double (error budgeting of the float result).
 * ---------------------------------------------------------------------------------------- */

enum { MAP_BALL_TO_CUBE_RADIAL = 0, MAP_BALL_TO_CUBE_VOLUME_PRESERVING = 1, MAP_IDENTITY = 2 };
enum { INTERP_LINEAR = 0, INTERP_LINEAR_BORDER = 1, INTERP_NEAREST = 2 };

#define DEFINE_CCONV(NAME, REAL, SQRT, FABS, ATAN, FLOOR, COPYSIGN, FMAX, FMIN, ROUND)                  \
    /* EXT MapSphereToCylinder */                                                                      \
    static inline void NAME##_sphere_to_cyl(REAL* x, REAL* y, REAL* z) {                                \
        REAL sq_norm = *x * *x + *y * *y + *z * *z;                                                     \
        REAL norm = SQRT(sq_norm);                                                                      \
        if (sq_norm < (REAL)1e-12) {                                                                    \
            *x = *y = *z = 0;                                                                           \
        } else if ((REAL)(5.0 / 4) * *z * *z > (*x * *x + *y * *y)) {                                   \
            REAL s = SQRT(3 * norm / (norm + FABS(*z)));                                                \
            *x *= s;                                                                                    \
            *y *= s;                                                                                    \
            *z = COPYSIGN(norm, *z);                                                                    \
        } else {                                                                                        \
            REAL s = norm / SQRT(*x * *x + *y * *y);                                                    \
            *x *= s;                                                                                    \
            *y *= s;                                                                                    \
            *z *= (REAL)(3.0 / 2);                                                                      \
        }                                                                                               \
    }                                                                                                   \
    /* EXT MapCylinderToCube */                                                                         \
    static inline void NAME##_cyl_to_cube(REAL* x, REAL* y, REAL* z) {                                  \
        (void)z;                                                                                        \
        REAL sq_norm = *x * *x + *y * *y;                                                               \
        REAL norm = SQRT(sq_norm);                                                                      \
        if (sq_norm < (REAL)1e-12) {                                                                    \
            *x = *y = 0;                                                                                \
        } else if (FABS(*y) <= FABS(*x)) {                                                              \
            REAL tmp = COPYSIGN(norm, *x);                                                              \
            *y = tmp * (REAL)(4 / M_PI) * ATAN(*y / *x);                                                \
            *x = tmp;                                                                                   \
        } else {                                                                                        \
            REAL tmp = COPYSIGN(norm, *y);                                                              \
            *x = tmp * (REAL)(4 / M_PI) * ATAN(*x / *y);                                                \
            *y = tmp;                                                                                   \
        }                                                                                               \
    }                                                                                                   \
    /* EXT ComputeFilterCoordinates: relative position -> coordinate in the filter array.            */ \
    static inline void NAME##_filter_coords(REAL* x, REAL* y, REAL* z, const int fs[3],                 \
                                            REAL inv_extent, int align_corners, int mapping) {          \
        if (mapping == MAP_BALL_TO_CUBE_RADIAL) {                                                       \
            *x *= 2 * inv_extent; *y *= 2 * inv_extent; *z *= 2 * inv_extent;                           \
            REAL radius = SQRT(*x * *x + *y * *y + *z * *z);                                            \
            REAL abs_max = FMAX(FABS(*x), FMAX(FABS(*y), FABS(*z)));                                    \
            if (abs_max < (REAL)1e-8) {                                                                 \
                *x = *y = *z = 0;                                                                       \
            } else {                                                                                    \
                *x *= (REAL)0.5 * radius / abs_max;                                                     \
                *y *= (REAL)0.5 * radius / abs_max;                                                     \
                *z *= (REAL)0.5 * radius / abs_max;                                                     \
            }                                                                                           \
        } else if (mapping == MAP_BALL_TO_CUBE_VOLUME_PRESERVING) {                                     \
            *x *= 2 * inv_extent; *y *= 2 * inv_extent; *z *= 2 * inv_extent;                           \
            NAME##_sphere_to_cyl(x, y, z);                                                              \
            NAME##_cyl_to_cube(x, y, z);                                                                \
            *x *= (REAL)0.5; *y *= (REAL)0.5; *z *= (REAL)0.5;                                          \
        } else {                                                                                        \
            *x *= inv_extent; *y *= inv_extent; *z *= inv_extent;                                       \
        }                                                                                               \
        if (align_corners) {                                                                            \
            *x += (REAL)0.5; *y += (REAL)0.5; *z += (REAL)0.5;                                          \
            *x *= (REAL)(fs[0] - 1); *y *= (REAL)(fs[1] - 1); *z *= (REAL)(fs[2] - 1);                  \
        } else {                                                                                        \
            *x *= (REAL)fs[0]; *y *= (REAL)fs[1]; *z *= (REAL)fs[2];                                    \
            *x += (REAL)(fs[0] / 2); *y += (REAL)(fs[1] / 2); *z += (REAL)(fs[2] / 2);                  \
            if (fs[0] % 2 == 0) *x -= (REAL)0.5;                                                        \
            if (fs[1] % 2 == 0) *y -= (REAL)0.5;                                                        \
            if (fs[2] % 2 == 0) *z -= (REAL)0.5;                                                        \
        }                                                                                               \
    }                                                                                                   \
    /* EXT InterpolationVec<...>::Interpolate: up to 8 (weight, spatial cell) pairs.                */ \
    static inline int NAME##_interpolate(REAL x, REAL y, REAL z, const int fs[3], int interpolation,    \
                                         REAL w[8], int cell[8]) {                                      \
        if (interpolation == INTERP_NEAREST) {                                                          \
            int xi = (int)ROUND(x), yi = (int)ROUND(y), zi = (int)ROUND(z);                             \
            xi = xi < 0 ? 0 : (xi > fs[0] - 1 ? fs[0] - 1 : xi);                                        \
            yi = yi < 0 ? 0 : (yi > fs[1] - 1 ? fs[1] - 1 : yi);                                        \
            zi = zi < 0 ? 0 : (zi > fs[2] - 1 ? fs[2] - 1 : zi);                                        \
            w[0] = 1;                                                                                   \
            cell[0] = (zi * fs[1] + yi) * fs[0] + xi;                                                   \
            return 1;                                                                                   \
        }                                                                                               \
        if (interpolation == INTERP_LINEAR) { /* coordinate clamping */                                 \
            x = FMIN((REAL)(fs[0] - 1), FMAX((REAL)0, x));                                              \
            y = FMIN((REAL)(fs[1] - 1), FMAX((REAL)0, y));                                              \
            z = FMIN((REAL)(fs[2] - 1), FMAX((REAL)0, z));                                              \
        }                                                                                               \
        REAL xf = FLOOR(x), yf = FLOOR(y), zf = FLOOR(z);                                               \
        int xi0 = (int)xf, yi0 = (int)yf, zi0 = (int)zf;                                                \
        int xi1 = xi0 + 1, yi1 = yi0 + 1, zi1 = zi0 + 1;                                                \
        REAL a = x - xf, b = y - yf, c = z - zf;                                                        \
        REAL ww[8] = {(1 - a) * (1 - b) * (1 - c), a * (1 - b) * (1 - c), (1 - a) * b * (1 - c),        \
                      a * b * (1 - c),             (1 - a) * (1 - b) * c, a * (1 - b) * c,              \
                      (1 - a) * b * c,             a * b * c};                                          \
        int xs[2] = {xi0, xi1}, ys[2] = {yi0, yi1}, zs[2] = {zi0, zi1};                                 \
        for (int t = 0; t < 8; ++t) {                                                                   \
            int xi = xs[t & 1], yi = ys[(t >> 1) & 1], zi = zs[(t >> 2) & 1];                           \
            if (interpolation == INTERP_LINEAR) {                                                       \
                xi = xi > fs[0] - 1 ? fs[0] - 1 : xi;                                                   \
                yi = yi > fs[1] - 1 ? fs[1] - 1 : yi;                                                   \
                zi = zi > fs[2] - 1 ? fs[2] - 1 : zi;                                                   \
                w[t] = ww[t];                                                                           \
            } else { /* LINEAR_BORDER: zero outside the array */                                        \
                int inside = xi >= 0 && xi < fs[0] && yi >= 0 && yi < fs[1] && zi >= 0 && zi < fs[2];   \
                w[t] = inside ? ww[t] : 0;                                                              \
                if (!inside) { xi = yi = zi = 0; }                                                      \
            }                                                                                           \
            cell[t] = (zi * fs[1] + yi) * fs[0] + xi;                                                   \
        }                                                                                               \
        return 8;                                                                                       \
    }                                                                                                   \
    /* out[i,:] = sum_j a_ij * sum_c f_j[c] * g(Lambda(x_j - x_i))[c,:]                              */ \
    /* filters [D(z),H(y),W(x),Cin,Cout] row-major; positions [*,3] xyz; CSR from the search.        */ \
    int NAME(const float* filters, const int32_t dims[5], const float* out_pos, int64_t m,              \
             float extent, const float* inp_pos, int64_t n, const float* inp_feat,                      \
             const float* inp_importance /*nullable [n]*/, const int32_t* nbr_index,                    \
             const int64_t* row_splits, const float* nbr_importance /*nullable [P]*/,                   \
             int align_corners, int mapping, int interpolation, int normalize, float* out) {            \
        (void)n;                                                                                        \
        const int fs[3] = {dims[2], dims[1], dims[0]}; /* filter_size_xyz */                            \
        const int K = dims[0] * dims[1] * dims[2], cin = dims[3], cout = dims[4];                       \
        if (K < 1 || cin < 1 || cout < 1 || !(extent > 0.0f)) return DMCF_REF_EINVAL;                   \
        const REAL inv_extent = (REAL)1 / (REAL)extent;                                                 \
        int err = 0;                                                                                    \
        _Pragma("omp parallel")                                                                         \
        {                                                                                               \
            REAL* B = (REAL*)malloc(sizeof(REAL) * (size_t)K * cin);                                    \
            REAL* acc = (REAL*)malloc(sizeof(REAL) * (size_t)cout);                                     \
            if (!B || !acc) err = 1;                                                                    \
            _Pragma("omp for schedule(dynamic, 32)")                                                    \
            for (int64_t i = 0; i < m; ++i) {                                                           \
                if (err) continue;                                                                      \
                memset(B, 0, sizeof(REAL) * (size_t)K * cin);                                           \
                REAL normalizer = 0;                                                                    \
                for (int64_t p = row_splits[i]; p < row_splits[i + 1]; ++p) {                           \
                    const int64_t j = nbr_index[p];                                                     \
                    REAL x = (REAL)(inp_pos[3 * j + 0] - out_pos[3 * i + 0]);                           \
                    REAL y = (REAL)(inp_pos[3 * j + 1] - out_pos[3 * i + 1]);                           \
                    REAL z = (REAL)(inp_pos[3 * j + 2] - out_pos[3 * i + 2]);                           \
                    const REAL n_imp = nbr_importance ? (REAL)nbr_importance[p] : (REAL)1;              \
                    normalizer += n_imp;                                                                \
                    NAME##_filter_coords(&x, &y, &z, fs, inv_extent, align_corners, mapping);           \
                    REAL w[8];                                                                          \
                    int cell[8];                                                                        \
                    int nw = NAME##_interpolate(x, y, z, fs, interpolation, w, cell);                   \
                    REAL importance = inp_importance ? (REAL)inp_importance[j] : (REAL)1;               \
                    if (nbr_importance) importance *= n_imp;                                            \
                    for (int t = 0; t < nw; ++t)                                                        \
                        for (int c = 0; c < cin; ++c)                                                   \
                            B[cell[t] * cin + c] += w[t] * ((REAL)inp_feat[j * cin + c] * importance);  \
                }                                                                                       \
                for (int o = 0; o < cout; ++o) acc[o] = 0;                                              \
                for (int kc = 0; kc < K * cin; ++kc) {                                                  \
                    const REAL b = B[kc];                                                               \
                    if (b != 0)                                                                         \
                        for (int o = 0; o < cout; ++o) acc[o] += (REAL)filters[(size_t)kc * cout + o] * b; \
                }                                                                                       \
                if (normalize && normalizer != 0)                                                       \
                    for (int o = 0; o < cout; ++o) acc[o] /= normalizer;                                \
                for (int o = 0; o < cout; ++o) out[i * cout + o] = (float)acc[o];                       \
            }                                                                                           \
            free(B);                                                                                    \
            free(acc);                                                                                  \
        }                                                                                               \
        return err ? DMCF_REF_EINVAL : DMCF_REF_OK;                                                     \
    }

DEFINE_CCONV(dmcf_ref_continuous_conv, float, sqrtf, fabsf, atanf, floorf, copysignf, fmaxf, fminf, roundf)
DEFINE_CCONV(dmcf_ref_continuous_conv_f64, double, sqrt, fabs, atan, floor, copysign, fmax, fmin, round)

/* Filter coordinates of single relative positions, exposed so tests can pin the mapping against
 * hand-computed analytic cases (on-axis neighbours, 2-D / 1-D degeneracy). */
int dmcf_ref_filter_coordinates(const float* rel /*[cnt,3]*/, int64_t cnt, float extent,
                                const int32_t ksize_zyx[3], int align_corners, int mapping,
                                float* coords /*[cnt,3] x,y,z in filter-array units*/) {
    const int fs[3] = {ksize_zyx[2], ksize_zyx[1], ksize_zyx[0]};
    const float inv_extent = 1.0f / extent;
    for (int64_t i = 0; i < cnt; ++i) {
        float x = rel[3 * i], y = rel[3 * i + 1], z = rel[3 * i + 2];
        dmcf_ref_continuous_conv_filter_coords(&x, &y, &z, fs, inv_extent, align_corners, mapping);
        coords[3 * i] = x;
        coords[3 * i + 1] = y;
        coords[3 * i + 2] = z;
    }
    return DMCF_REF_OK;
}
