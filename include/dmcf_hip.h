/*
 * dmcf_hip.h -- C ABI of libdmcf_hip.so: the MI355X (gfx950) implementation of DMCF's per-step
 * particle hot path.  This is the drop-in boundary: these entry points are what the reference's
 * Python layer binds in place of the Open3D (open3d==0.15.2) TensorFlow operators it calls.
 * Paths below are relative to the reference tree (tum-pbs/DMCF).
 *
 *   reference operator (call site)                               replaced by
 *   ------------------------------------------------------------ -------------------------------
 *   ml3d.layers.FixedRadiusSearch = build_spatial_hash_table +   dmcf_frs_build / dmcf_frs_count /
 *     fixed_radius_search  (utils/convolutions.py:207-210,        dmcf_frs_write
 *     354-358; utils/tools/losses.py:296-298,339-341)
 *   window functions (utils/tools/losses.py:8-44) applied at     DMCF_WINDOW_* fused into
 *     utils/convolutions.py:359-379                               dmcf_cconv_forward
 *   ml3d.ops.continuous_conv (utils/convolutions.py:414-431)     dmcf_cconv_forward
 *   ASCC: mirror :410-412 + second continuous_conv :433-458      dmcf_cconv_forward(DMCF_FLAG_SYMMETRIC)
 *   o3dml.ops.reduce_subarrays_sum (models/pbf_model.py:450-453) dmcf_reduce_subarrays_sum
 *   tf.keras.layers.Dense (models/hrnet.py:49,93-99;             dmcf_dense_forward
 *     models/pbf_model.py:134-152)
 *   tf.reduce_min / reduce_max of the positions                  dmcf_points_aabb
 *     (models/pbf_model.py:330-336)
 *   farthest_point_sample / gather_point (utils/tools/sampling.cu) dmcf_farthest_point_sample / dmcf_gather_point
 *   grid_pos: candidate cells + tf.unique + decode               dmcf_grid_pos_bounds / _count / _write
 *     (utils/tools/losses.py:136-181, called from :266-272)
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, sizes, a stream handle (hipStream_t passed as void*); no
 *     torch / HIP types in any signature.
 *   - every function returns DMCF_OK (0) or a negative DMCF_E* code; nothing throws; arguments are
 *     validated on the host; kernels are enqueued on `stream` and NOT synchronised.
 *   - ownership: the caller owns every buffer.  The library never allocates device memory; scratch
 *     comes from a caller-supplied workspace whose size the *_workspace_bytes functions report.
 *   - stateless and re-entrant: ordering only through `stream`.
 *   - layouts: positions [n,3] float32 row-major xyz (z = 0 in 2-D scenes); features [n,C] float32
 *     row-major; filters [D(z),H(y),W(x),Cin,Cout] float32 row-major; CSR neighbour lists with
 *     int32 indices and int64 row splits (the dtypes of Open3D 0.15.2's op); distances are
 *     squared L2.
 *   - two-phase search because the number of pairs is data dependent:
 *       build -> count (fills row_splits) -> caller reads row_splits[m], allocates -> write.
 */
#ifndef DMCF_HIP_H_
#define DMCF_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMCF_OK 0
#define DMCF_EINVAL (-1)      /* bad argument (null pointer, negative size, unsupported option) */
#define DMCF_EWORKSPACE (-2)  /* workspace too small */
#define DMCF_ELAUNCH (-3)     /* HIP reported an error when enqueuing (see dmcf_last_hip_error) */
#define DMCF_EUNSUPPORTED (-4) /* valid in the reference, not implemented on this path yet */

typedef void* dmcf_stream_t; /* hipStream_t */

/* library / ABI version (major*10000 + minor*100 + patch) and error text */
int dmcf_version(void);
const char* dmcf_error_string(int code);
int dmcf_last_hip_error(void); /* hipError_t of the last failed enqueue on this thread */

/* ------------------------------------------------------------------------------------------------
 * Fixed-radius search.  Replaces ml3d.layers.FixedRadiusSearch(metric='L2', ignore_query_point,
 * return_distances)(points, queries, radius)  (utils/convolutions.py:207-210, 354-358).
 * Result contract (what the reference's callers observe): for every query i the SET
 *   { j : ((dx*dx + dy*dy) + dz*dz) <= radius*radius }     float32, un-fused, inclusive,
 * minus points whose coordinates equal the query's when DMCF_FRS_IGNORE_QUERY_POINT is set.
 * The order inside a row is implementation defined in the reference (hash-bin order, atomics);
 * here it is deterministic: ascending grid cell (z, y, x), then ascending point index.
 * ---------------------------------------------------------------------------------------------- */
#define DMCF_FRS_IGNORE_QUERY_POINT 1
/* Opt-in emulations of what open3d 0.15.2's hash walk can SEE in float arithmetic.  The library hashes the points into voxels
 * of edge 2 R and visits, for a query, the bins of its own voxel and of the 8 voxels holding the corners q +- R (SURVEY.md
 * section 8 row a1).  In exact arithmetic those voxels cover the search sphere: the set above, which is what the search returns
 * WITHOUT either flag -- symmetric lists, on which the ASCC layer's momentum conservation rests (models/sym_net.py:42-53).  In
 * float, for about one query in 10^6 (q a rounding step from the middle of a voxel) the corner voxels of an axis are TWO
 * apart and the walk misses part of the sphere; a pair at distance R within rounding can sit one voxel outside as well.
 *   DMCF_FRS_OPEN3D_VOXEL_WALK     a hit counts only if the point's voxel hashes into the bin of the query's own voxel or of
 *                                  one of the 8 corner voxels (table of clamp(n_points / 64, 1, 2^25) bins, the layer's
 *                                  default) -- such a query keeps what lies in its own voxel;
 *   DMCF_FRS_OPEN3D_CORNER_VOXELS  the same with the 8 corner voxels alone (round 3's reading of the library: such a query's
 *                                  row comes out nearly empty).
 * Each is bit-exact against the restatement of that walk in oracle/ (dmcf_ref_fixed_radius_search, bin_set); which of the two
 * the library implements is one of the questions tools/capture_golden.py settles.  At most one of them may be set. */
#define DMCF_FRS_OPEN3D_CORNER_VOXELS 2
#define DMCF_FRS_OPEN3D_VOXEL_WALK 4

/* bytes of workspace for a search structure over n_points that will serve up to n_queries queries */
size_t dmcf_frs_workspace_bytes(int64_t n_points, int64_t n_queries);

/* build the cell-sorted uniform grid of `points` in `workspace` (device memory, 256-B aligned) */
int dmcf_frs_build(const float* points, int64_t n_points, float radius, void* workspace,
                   size_t workspace_bytes, dmcf_stream_t stream);

/* count neighbours of each query and write the int64 exclusive prefix sum to row_splits[0..m];
 * n_points / radius / workspace must be the ones given to dmcf_frs_build */
int dmcf_frs_count(const float* queries, int64_t n_queries, int64_t n_points, float radius, int flags,
                   void* workspace, size_t workspace_bytes, int64_t* row_splits, dmcf_stream_t stream);

/* write neighbors_index[P] (int32) and, if not NULL, neighbors_distance[P] (squared L2).  pair_capacity = number
 * of entries the two output buffers hold.  A caller that read row_splits[m] passes exactly that.  A caller that
 * wants NO host round trip allocates from an estimate (e.g. the previous time step's count plus slack), enqueues
 * count + write back to back and checks row_splits[m] <= pair_capacity later: rows that would not fit are skipped
 * as a whole, never written out of bounds. */
int dmcf_frs_write(const float* queries, int64_t n_queries, int64_t n_points, float radius, int flags,
                   const void* workspace, size_t workspace_bytes, const int64_t* row_splits,
                   int32_t* neighbors_index, float* neighbors_distance, int64_t pair_capacity,
                   dmcf_stream_t stream);

/* Single-pass search into PADDED rows, for callers that know an upper bound of the row lengths (a rollout takes the
 * largest row of the previous time step + slack): row i is written at neighbors_index[i * row_stride ...], its length
 * (clamped to row_stride) goes to row_count[i], row_begin[i] = i * row_stride for i = 0..n_queries, and the largest
 * unclamped length is max-ed into *max_count (a device int32 the caller zeroed) -- max_count > row_stride means rows
 * were truncated and the caller must repeat with a larger stride or with dmcf_frs_count / dmcf_frs_write.  No count
 * pass, no prefix scan: one candidate scan per query.  Same neighbours in the same order per row as dmcf_frs_write;
 * dmcf_cconv_forward consumes the result through args->neighbors_row_count. */
int dmcf_frs_search_padded(const float* queries, int64_t n_queries, int64_t n_points, float radius, int flags,
                           const void* workspace, size_t workspace_bytes, int64_t row_stride, int64_t* row_begin,
                           int32_t* row_count, int32_t* neighbors_index, float* neighbors_distance, int32_t* max_count,
                           dmcf_stream_t stream);

/* compute_density (utils/tools/losses.py:285-306; models/pbf_model.py:351-355, pipelines/simulator.py:227-243):
 *   out[q] = sum over the points p within `radius` of query q of window(|p - q|^2 / radius^2)
 * evaluated inside the candidate scan of the search -- the pair list is never materialised.  `window` is a
 * DMCF_WINDOW_* id (DMCF_WINDOW_NONE counts the neighbours, DMCF_WINDOW_EXPLICIT sums the squared distances);
 * flags as for dmcf_frs_count.  Uses the workspace of dmcf_frs_build(points). */
int dmcf_frs_window_sum(const float* queries, int64_t n_queries, int64_t n_points, float radius, int flags, int window,
                        const void* workspace, size_t workspace_bytes, float* out, dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Continuous convolution (CConv) and its antisymmetric variant (ASCC).
 * Replaces ml3d.ops.continuous_conv(filters, out_positions, extents[1,1], offset=0, inp_positions,
 * inp_features, inp_importance, neighbors_index, neighbors_row_splits, neighbors_importance,
 * align_corners, coordinate_mapping, interpolation, normalize)  (utils/convolutions.py:414-431):
 *     out[i,:] = 1/psi_i * sum_{p in row i} a_p * s_j * sum_c f_j[c] * g(Lambda(x_j - x_i))[c,:]
 * with j = neighbors_index[p], a_p the per-neighbour importance, s_j the per-point importance,
 * psi_i = sum_p a_p when DMCF_FLAG_NORMALIZE (else 1).
 * With DMCF_FLAG_SYMMETRIC the call computes the whole ASCC layer body
 * (utils/convolutions.py:410-412 and 433-458) in one pass:
 *     g = concat([-flip_zyx(filters), filters], axis=sym_axis);
 *     out[i,:] = sum_p a_p * sum_c (f_j[c] + f_i[c]) * g(Lambda(x_j - x_i))[c,:]
 * which requires the output points to be the input points 0..n_out-1 (the reference passes the same set
 * for both, models/sym_net.py:66; n_inp > n_out is the sharded case: owned points first, ghosts after).
 * ---------------------------------------------------------------------------------------------- */
enum dmcf_mapping {
    DMCF_MAP_BALL_TO_CUBE_RADIAL = 0,
    DMCF_MAP_BALL_TO_CUBE_VOLUME_PRESERVING = 1,
    DMCF_MAP_IDENTITY = 2
};
enum dmcf_interpolation { DMCF_INTERP_LINEAR = 0, DMCF_INTERP_LINEAR_BORDER = 1, DMCF_INTERP_NEAREST = 2 };
/* how the per-neighbour importance a_p is obtained from `neighbors_value` */
enum dmcf_window {
    DMCF_WINDOW_NONE = 0,       /* a_p = 1; neighbors_value ignored (may be NULL) */
    DMCF_WINDOW_EXPLICIT = 1,   /* a_p = neighbors_value[p] (user supplied importance) */
    /* a_p = w(q), q = neighbors_value[p] / radius^2, neighbors_value = squared distances
     * (utils/convolutions.py:359-362, 375-379; formulas utils/tools/losses.py:8-44).
     * neighbors_value == NULL: the squared distance is re-formed from the two positions, with the operations and the order
     * dmcf_frs_write / dmcf_frs_search_padded use for the distances they return -- bit-identical results, and the list needs
     * no distance array. */
    DMCF_WINDOW_POLY6 = 2,
    DMCF_WINDOW_CUBIC = 3,
    DMCF_WINDOW_LINEAR = 4,
    DMCF_WINDOW_PEAK = 5,
    DMCF_WINDOW_CUBIC_GRAD = 6
};
#define DMCF_FLAG_ALIGN_CORNERS 1
#define DMCF_FLAG_NORMALIZE 2
#define DMCF_FLAG_SYMMETRIC 4  /* ASCC: filters is the stored half kernel, see above */
#define DMCF_FLAG_ACCUMULATE 8 /* out += result instead of out = result (add_merge, models/hrnet.py:115-116) */
#define DMCF_FLAG_FILTER_PACKED 32 /* the caller's promise that `workspace` still holds what the previous dmcf_cconv_forward with
                                      the SAME filters (values included), filter_dims, flags & (SYMMETRIC | sym_axis) and kernel
                                      choice (dmcf_cconv_kernel_name returns the same string for both calls) left there: the
                                      filter is not packed again (one launch less per layer; inference weights do not change from
                                      step to step, and a step of a 2,000-particle scene is paced by its launches).  The library
                                      keeps no state: whoever sets the flag owns the workspace between the calls. */
#define DMCF_FLAG_SKIP_SELF 16 /* pairs with neighbors_index[p] == the output row -- and pairs whose two positions are EQUAL, which
                                  is the test the search applies (ignore_query_point drops every point at the query position) --
                                  carry weight zero: a list searched WITH the query points serves a layer that ignores them (radius_search_ignore_query_points,
                                  utils/convolutions.py:207-210) when its outputs are its first n_out inputs -- the ASCC
                                  head then shares the list of the trunk's same-scale layers instead of searching again.
                                  Only the kernel for <= 4 output channels and large filters implements it
                                  (dmcf_cconv_kernel_name: "cconv_direct_kernel..."); DMCF_EUNSUPPORTED otherwise */

typedef struct dmcf_cconv_args {
    const float* filters;      /* [D,H,W,Cin,Cout]; with SYMMETRIC: dims[sym_axis] is the stored half size */
    int32_t filter_dims[5];    /* D(z), H(y), W(x), Cin, Cout of `filters` as passed */
    int32_t sym_axis;          /* 0..2, index into (z,y,x); only with DMCF_FLAG_SYMMETRIC */
    const float* out_positions; /* [n_out,3] */
    int64_t n_out;
    const float* inp_positions; /* [n_inp,3] */
    int64_t n_inp;
    const float* inp_features;   /* [n_inp,Cin] */
    const float* inp_importance; /* [n_inp] or NULL (always NULL in DMCF: models/hrnet.py:91-92) */
    const int32_t* neighbors_index;       /* [P] */
    const int64_t* neighbors_row_splits;  /* [n_out+1] */
    const float* neighbors_value;         /* [P] squared distances (or NULL) or importances, see dmcf_window */
    float extent;              /* scalar filter extent (diameter); radius = extent/2 */
    float window_fac;          /* multiplier of the window function ("fac", losses.py:8), normally 1 */
    int32_t window;            /* enum dmcf_window */
    int32_t coordinate_mapping; /* enum dmcf_mapping */
    int32_t interpolation;      /* enum dmcf_interpolation */
    int32_t flags;              /* DMCF_FLAG_* */
    const float* bias;          /* [Cout] or NULL; added after normalisation (convolutions.py:466-467) */
    float* out;                 /* [n_out,Cout] */
    int64_t n_pairs;            /* entries in neighbors_index / neighbors_value (>= P = neighbors_row_splits[n_out]);
                                 * rows reaching past it are treated as empty (see dmcf_frs_write pair_capacity) */
    const int32_t* neighbors_row_count; /* optional [n_out]: PADDED lists as written by dmcf_frs_search_padded -- row i is
                                   neighbors_index[row_splits[i] .. row_splits[i] + row_count[i]); NULL = CSR rows
                                   [row_splits[i], row_splits[i+1]) */
    uint32_t filter_tile_mask; /* optional hint about filter blocks that are ALL ZERO (the block-diagonal filters of two layers
                                  launched as one, models/hrnet.py:85-92 twice on one list): bit 4 * (c / 4) + o / 16 is set
                                  when input channels 4 (c / 4) .. + 3 have a non-zero weight into output channels
                                  16 (o / 16) .. + 15 in some filter cell (c < 32, o < 64).  0 = no hint (every block is
                                  multiplied).  Kernels may skip the fetch and the products of unset blocks; results are
                                  identical for finite features (a skipped product is an exact zero). */
    int32_t row_length_hint;   /* optional, a property of the LAYER the caller knows from its configuration (the dispatch never
                                  looks at the neighbour list itself: the same step gives the same bits on CSR and on padded
                                  lists): 0 = unknown, 1 = rows of tens of neighbours (a layer at the network's base radius,
                                  models/hrnet.py:86 with inp_scale = out_scale = 0), 2 = rows of hundreds or more (any wider
                                  radius).  With 2, layers of 17 .. 32 input channels and 4 x 4 x 4 filters take the
                                  pair-per-instruction kernel ("cconv_pair_kernel..."), which needs long rows to pay for
                                  its per-point merge; with 1, layers of 24 .. 32 input channels and at most 32 output
                                  channels take the wave-specialised kernel ("cconv_ws_kernel...": that splat in producer
                                  waves, the contraction in consumer waves of a persistent workgroup), which pays when a
                                  row's contraction is as much work as its splat. */
} dmcf_cconv_args;

size_t dmcf_cconv_workspace_bytes(const dmcf_cconv_args* args);

int dmcf_cconv_forward(const dmcf_cconv_args* args, void* workspace, size_t workspace_bytes,
                       dmcf_stream_t stream);
/* Diagnostics: the name of the device kernel dmcf_cconv_forward dispatches these arguments to (the dispatch looks at the
 * layer -- filter shape, channel counts, flags -- never at the neighbour list), as rocprofv3 prints it without the
 * namespace, e.g. "cconv_z3_kernel<1, true>".  bench.py groups its per-launch HIP-event timings by it. */
int dmcf_cconv_kernel_name(const dmcf_cconv_args* args, char* name, size_t name_bytes);

/* ------------------------------------------------------------------------------------------------
 * ml3d.ops.continuous_conv (utils/convolutions.py:414-431) FROM particles ONTO a coarse grid_pos lattice with few output
 * channels -- the layers models/hrnet.py:83-93 builds for (input scale 0, output scale >= 1) with layer_channels[..][scale] of
 * 4 or 8 (configs/Liquid3d.yml:11: [[16], [8], [4]] / [[32], [16], [8]]).  Same operator, other order of evaluation ("filter
 * first", input stationary; dmcf_amd/csrc/cconv_sct.hip):  G_j = f_j . W once per INPUT point, then per pair the trilinear
 * interpolation of G_j added into the output point.  It walks the TRANSPOSED neighbour list: row j = the output points within
 * extent / 2 of input point j -- the list FixedRadiusSearch returns for (points = out_positions, queries = inp_positions), which
 * holds the same pairs as the forward list when the search's neighbour set is symmetric (the default set of this library).
 * Sums are formed in 64-bit fixed point (a term c enters as round(c * 2^s), 2^s * max_j |f_j|_1 * max |W| <= 2^46, both maxima
 * formed on the device inside the call), so the result does not depend on the order of the additions: bit reproducible like
 * every other kernel here, with LDS and global atomics doing the adds.
 *
 * dmcf_cconv_scatter_plan (once per pair of point sets and radius; every layer between them shares it) counting-sorts the input
 * points by the block of block_cells^3 lattice cells they lie in (cell of a point = floor((x - out_positions[0]) / voxel)); all
 * outputs a block reaches lie in a box of (block_cells + 2 reach + 1)^3 lattice cells whose 64-bit accumulators live in LDS and
 * are flushed once per block.  No host round trip; points far outside the bulk (beyond a 128^3-block region around the mean) are
 * handled one by one.  `plan` is caller-owned device memory of dmcf_cconv_scatter_plan_bytes(n_inp) bytes.
 *
 * Restrictions (DMCF_EUNSUPPORTED otherwise): 4x4x4 filters, cout 4 or 8, cin <= 32, DMCF_WINDOW_NONE / POLY6 evaluated from the
 * positions, volume-preserving map, linear interpolation; flags: DMCF_FLAG_ALIGN_CORNERS (required) | DMCF_FLAG_ACCUMULATE;
 * block_cells + 2 reach + 1 <= 13 at cout 4 (11 at cout 8): the box must fit the CU's LDS.
 * workspace: dmcf_cconv_scatter_workspace_bytes (the 64-bit sums; zeroed by the call).  error_flag (optional device int32, the
 * caller zeroes it): set to 1 when a pair fell outside its block's box, i.e. the plan was not made from these positions,
 * voxel and radius -- the pair is then dropped.
 * ---------------------------------------------------------------------------------------------- */
size_t dmcf_cconv_scatter_plan_bytes(int64_t n_inp);
int dmcf_cconv_scatter_plan(const float* inp_positions, int64_t n_inp, const float* out_positions, int64_t n_out, float voxel,
                            float extent, int32_t block_cells, void* plan, size_t plan_bytes, dmcf_stream_t stream);

typedef struct dmcf_cconv_scatter_args {
    const float* filters;          /* [4][4][4][cin][cout] */
    int32_t filter_dims[5];
    const float* out_positions;    /* [n_out][3], a lattice of spacing voxel */
    int64_t n_out;
    const float* inp_positions;    /* [n_inp][3] */
    int64_t n_inp;
    const float* inp_features;     /* [n_inp][cin] */
    const int32_t* t_index;        /* transposed list: output indices, row j = input point j */
    const int64_t* t_row_begin;    /* [n_inp + 1] CSR row splits, or [n_inp] row begins when t_row_count is given */
    const int32_t* t_row_count;    /* optional [n_inp] (padded rows); NULL = CSR */
    int64_t t_capacity;            /* entries t_index holds */
    const void* plan;              /* dmcf_cconv_scatter_plan's output for these positions */
    int32_t block_cells;           /* as passed to the plan */
    int32_t reach;                 /* ceil(extent / 2 / voxel) */
    float extent;
    float window_fac;
    int32_t window;
    int32_t flags;
    const float* bias;             /* [cout] or NULL */
    float* out;                    /* [n_out][cout] */
    int32_t* error_flag;
} dmcf_cconv_scatter_args;

size_t dmcf_cconv_scatter_workspace_bytes(const dmcf_cconv_scatter_args* args);
int dmcf_cconv_scatter_forward(const dmcf_cconv_scatter_args* args, void* workspace, size_t workspace_bytes,
                               dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * ml3d.ops.continuous_conv (utils/convolutions.py:414-431) between two point sets on ALIGNED REGULAR LATTICES -- the
 * coarse scales of the multi-scale models: models/hrnet.py:85-92 convolves between the outputs of grid_pos
 * (utils/tools/losses.py:136-181, via get_dilated_pos :266-272), which share one centre and whose voxel sizes are integer
 * multiples of each other.  Input point j sits at centre + inp_cell_j * voxel, output point i at
 * centre + out_cell_i * (output spacing); every per-pair quantity of the operator depends only on the integer offset
 * of the input cell from the output point, so no neighbour list is needed: the caller passes the offsets inside the
 * radius, the input features laid out by lattice cell (a dense volume over the bounding box of the input lattice) and
 * a cell -> output point table of the output lattice.  Same filters / extent / window / mapping / interpolation /
 * ALIGN_CORNERS / bias / ACCUMULATE semantics as dmcf_cconv_forward; SYMMETRIC, NORMALIZE, DMCF_WINDOW_EXPLICIT and
 * importances are not supported (DMCF_EUNSUPPORTED: use the neighbour-list form).  Results agree with the neighbour-list
 * form to the rounding of the positions (the reference subtracts rounded positions, this form uses d * voxel).
 * ---------------------------------------------------------------------------------------------- */
typedef struct dmcf_lattice_conv_args {
    const float* filters;        /* [D,H,W,Cin,Cout] */
    int32_t filter_dims[5];
    const float* inp_volume;     /* [inp_dims z][y][x][Cin]: the input features by lattice cell, zeros where no point is */
    int32_t inp_min[3];          /* (x,y,z) cell of entry 0 of inp_volume, in units of the input lattice */
    int32_t inp_dims[3];         /* (x,y,z) */
    const int32_t* out_table;    /* [out_dims z][y][x]: index of the output point in that cell of the OUTPUT lattice, or -1 */
    int32_t out_min[3];          /* (x,y,z) cell of entry 0 of out_table, in units of the output lattice */
    int32_t out_dims[3];
    int64_t n_out;               /* rows of `out` */
    /* The launch covers the output cells  o = a * out_stride + out_phase  for the integer vectors a of the box
     * [base_min, base_min + base_dims); the stencil of such a cell is the input cells  a * inp_step + d.
     *   same lattice:            inp_step = out_stride = 1, phase 0;
     *   outputs 2x coarser:      inp_step = 2, out_stride = 1, phase 0;
     *   outputs 2x finer:        inp_step = 1, out_stride = 2, one launch per phase in {0,1}^3, and the output sits
     *                            rel_shift = phase * (output spacing) away from the input cell a * inp_step. */
    int32_t inp_step;
    int32_t out_stride;
    int32_t out_phase[3];
    int32_t base_min[3], base_dims[3];
    float rel_shift[3];          /* x_in - x_out = d * voxel - rel_shift */
    float voxel[3];              /* (x,y,z) spacing of the input lattice */
    const int32_t* offsets;      /* [n_offsets,4]: (dx,dy,dz,0) with |d * voxel - rel_shift| <= extent / 2 */
    int64_t n_offsets;
    int32_t reach[3];            /* max |d| per axis over `offsets`.  The volume must hold EVERY cell the launch can touch --
                                  * a * inp_step + d for a in the base box (its x extent rounded up to a multiple of 16)
                                  * and |d| <= reach -- so the kernel reads without bounds checks (DMCF_EINVAL otherwise):
                                  * the caller pads the volume with zero cells */
    float extent;
    float window_fac;
    int32_t window;              /* enum dmcf_window (applied to |d * voxel|^2 / radius^2) */
    int32_t coordinate_mapping;
    int32_t interpolation;
    int32_t flags;               /* DMCF_FLAG_ALIGN_CORNERS | DMCF_FLAG_ACCUMULATE */
    const float* bias;           /* [Cout] or NULL */
    float* out;                  /* [n_out,Cout]; rows of points that are in out_table are written */
} dmcf_lattice_conv_args;

size_t dmcf_lattice_conv_workspace_bytes(const dmcf_lattice_conv_args* args);
int dmcf_lattice_conv_forward(const dmcf_lattice_conv_args* args, void* workspace, size_t workspace_bytes,
                              dmcf_stream_t stream);
/* Up to 8 launches of the same layer (same filters, volume, output; e.g. the eight parity classes of outputs on the finer
 * lattice) as ONE grid: each alone is too small to fill the chip. */
size_t dmcf_lattice_conv_batch_workspace_bytes(const dmcf_lattice_conv_args* parts, int32_t n_parts);
int dmcf_lattice_conv_forward_batch(const dmcf_lattice_conv_args* parts, int32_t n_parts, void* workspace,
                                    size_t workspace_bytes, dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * o3dml.ops.reduce_subarrays_sum(values, row_splits) (models/pbf_model.py:450-453):
 *   out[i] = sum(values[row_splits[i] : row_splits[i+1]]);  values == NULL means all ones.
 * ---------------------------------------------------------------------------------------------- */
int dmcf_reduce_subarrays_sum(const float* values, const int64_t* row_splits, int64_t n_rows, float* out,
                              dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * tf.keras.layers.Dense of the networks (models/pbf_model.py:134-152 fluid_dense / obs_dense, models/hrnet.py:49 the
 * same-scale branch :93-99, models/cconv.py:44):  out[n, m] = x[n, k] W[k, m] (+ bias[m]) (+ residual[n, m]) -- a million rows,
 * a few tens of columns: a wavefront keeps the filter as matrix-instruction fragments and walks 16-row tiles.  k a multiple of 4
 * up to 64, m up to 64, x 16-byte aligned; anything else: DMCF_EUNSUPPORTED (the caller keeps its library GEMM).  out may not
 * alias x; it may be the residual.
 * ---------------------------------------------------------------------------------------------- */
int dmcf_dense_forward(const float* x, int64_t n, int32_t k, const float* W, int32_t m, const float* bias,
                       const float* residual, float* out, dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The fluid's axis-aligned bounding box, tf.reduce_min / reduce_max(pos, axis=0) of the boundary crop every step begins with
 * (models/pbf_model.py:330-336): out[0..2] = min x, y, z, out[3..5] = max.  n == 0: +inf / -inf.  A NaN coordinate makes both
 * bounds of its axis NaN, as the reference's reductions do.  Workspace: dmcf_points_aabb_workspace_bytes() bytes.
 * ---------------------------------------------------------------------------------------------- */
size_t dmcf_points_aabb_workspace_bytes(void);
int dmcf_points_aabb(const float* points, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                     dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Ghost selection of the block-sharded rollout (no reference counterpart: tum-pbs/DMCF has no distributed code -- SURVEY.md
 * section 8e; the caller is dmcf_amd/parallel.py): which points lie within a halo width of which axis-aligned block, for
 * several widths at once.
 *   level(i, b) = number of w with gap2(points[i], boxes[b]) <= widths2[w]     (widths2: HOST array, DESCENDING, <= 8 entries)
 *   list(w)     = for b = 0 .. n_boxes - 1 (<= 64): the i with level(i, b) > w, in ascending i
 * gap2 = squared Euclidean distance to the box, per axis max(lo - x, x - hi, 0), un-fused float32; boxes [n_boxes][6] =
 * lo x, y, z, hi x, y, z on the DEVICE, +-inf on open sides.
 *   count : totals[w * n_boxes + b] = entries box b contributes to list(w) (DEVICE int64)
 *   write : list(w) -> rows[list_start[w] ..) (DEVICE int64 point indices), at most list_capacity[w] entries (HOST arrays); must
 *           follow a count with the same arguments and the same workspace.
 * OWNERSHIP: n_widths == 1 with widths2[0] < 0 replaces the test by "lo <= x < hi on every axis" (half-open blocks: a point has one
 * owner); list(0) is then the stable order of the points by owning box and totals[b] the points box b owns.  A NaN or infinite
 * coordinate is in no box: the totals then sum to less than n.
 * Workspace: dmcf_ghost_workspace_bytes(n, n_boxes, n_widths).
 * ---------------------------------------------------------------------------------------------- */
size_t dmcf_ghost_workspace_bytes(int64_t n, int32_t n_boxes, int32_t n_widths);
int dmcf_ghost_count(const float* points, int64_t n, const float* boxes, int32_t n_boxes, const float* widths2, int32_t n_widths,
                     int64_t* totals, void* workspace, size_t workspace_bytes, dmcf_stream_t stream);
int dmcf_ghost_write(const float* points, int64_t n, const float* boxes, int32_t n_boxes, const float* widths2, int32_t n_widths,
                     int64_t* rows, const int64_t* list_start, const int64_t* list_capacity, void* workspace, size_t workspace_bytes,
                     dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * grid_pos(pos, voxel_size, centralize, pad, hyst) (utils/tools/losses.py:136-181; the coarse point sets of the
 * multi-scale models, called through get_dilated_pos :249-284): the corners of every voxel a particle touches
 * (with +-hyst hysteresis), de-duplicated in tf.unique order (first appearance in the candidate list), decoded to
 * positions.  Three phases because two sizes are data dependent:
 *   bounds : lattice origin (mean of the positions when `centralize` and `center` == NULL, else *center), integer
 *            bounding box of the candidates -> 64-byte header at the start of `workspace`:
 *              int32 minp[3], int32 dims[3], int64 cells (= dims product, -1 if a position is not finite),
 *              int64 total, float center[3]
 *            the caller reads `cells` and provides a table of that many uint32
 *   count  : fills the table, counts the lattice points -> header.total; the caller reads it and allocates out
 *   write  : out[total,3] float32
 * `voxel_size` is a HOST pointer to 3 floats (axes below 1e-5 collapse, :139-141); `center` a DEVICE pointer to 3
 * floats or NULL.  The same (positions, voxel_size, centralize, pad, hyst, workspace) must be passed to all
 * three calls.  Candidate ranks are 32-bit: 2 * n_points * (2 + 2 pad)^3 must stay below 2^32
 * (DMCF_EUNSUPPORTED otherwise).
 * SPARSE scenes (a few particles far from the rest: `cells` in the billions for a million occupied ones): pass
 * table_cells = -S to count / write, S a power of two >= twice the number of candidates 2 * n_points * (2 + 2 pad)^3 at most
 * occupied, and a table of 12 * S bytes: the kernels then hash the cell index into S slots (open addressing) instead of
 * indexing a dense table.  Same points, same order.
 * ---------------------------------------------------------------------------------------------- */
size_t dmcf_grid_pos_workspace_bytes(int64_t n_points);
int dmcf_grid_pos_bounds(const float* positions, int64_t n_points, const float* voxel_size, int centralize,
                         const float* center, int pad, float hyst, void* workspace, size_t workspace_bytes,
                         dmcf_stream_t stream);
int dmcf_grid_pos_count(const float* positions, int64_t n_points, const float* voxel_size, int centralize, int pad,
                        float hyst, void* workspace, size_t workspace_bytes, void* cell_table, int64_t table_cells,
                        dmcf_stream_t stream);
int dmcf_grid_pos_write(const float* positions, int64_t n_points, const float* voxel_size, int centralize, int pad,
                        float hyst, void* workspace, size_t workspace_bytes, const void* cell_table, int64_t table_cells,
                        float* out, int64_t out_capacity, dmcf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * farthest_point_sample(npoint, inp[1,n,3]) and gather_point(inp, idx) (utils/tools/sampling.py / sampling.cu:125-201):
 * the sub-sampling of get_dilated_pos for multi-scale configs without voxel_size (utils/tools/losses.py:274-282) and
 * the index lists of HRNet's cross-scale Dense branch (models/hrnet.py:100-113).  One point set per call (the
 * reference's batch dimension is always 1 on this path).  sample_index[0] = 0; sample j maximises the float32
 * squared distance to samples 0..j-1; equal maxima resolve as in the reference's 512-thread kernel (smallest
 * index mod 512, then smallest index).  Sequential in the samples by definition: O(n_points * n_samples).
 * ---------------------------------------------------------------------------------------------- */
size_t dmcf_fps_workspace_bytes(int64_t n_points);
int dmcf_farthest_point_sample(const float* points, int64_t n_points, int64_t n_samples, void* workspace,
                               size_t workspace_bytes, int32_t* sample_index, dmcf_stream_t stream);
int dmcf_gather_point(const float* inp, const int32_t* index, int64_t n_index, int channels, float* out,
                      dmcf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DMCF_HIP_H_ */
