/*
 * mink_hip.h -- C ABI of libmink_hip.so: the MI355X (gfx950) native backend for the
 * sparse-3D-convolution classification hot path of POSTECH-CVLab/NeRF-Downstream
 * (co3d_3d/train.py -> Mink-ResNet14/34 on PeRFception-CO3D plenoxel grids).
 *
 * The reference has no C FFI of its own for this path: its seam is the Python module
 * API of the third-party MinkowskiEngine (ME), whose native backend
 * (`MinkowskiEngineBackend._C`, imported by the reference at
 * co3d_3d/src/models/mink/modules/sparse_conv.py:7-12) is what these entry points
 * replace.  Each declaration cites the reference call site(s) it serves; paths are
 * relative to /root/reference.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless named `*_host`;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is
 *     enqueued asynchronously on it, nothing here synchronises the device;
 *   - the caller owns every buffer (no hidden allocations, no global state) so the
 *     library is safe to call from a graph-captured or multi-stream host; every scratch
 *     buffer is passed WITH ITS SIZE (`workspace_bytes`, ABI version 2): the size a launch needs can depend on a
 *     plan (split factors, row splits), so the library checks it against the plan it is about to launch and returns
 *     MINK_EINVAL instead of writing past the end of a buffer sized for another plan;
 *   - return value 0 = success, otherwise a negative MINK_E* code and
 *     mink_last_error() (thread-local) describes the failure;
 *   - coordinates are int32 rows (batch, x, y, z); row counts are int64; feature
 *     matrices are row-major fp32 with an explicit leading dimension (`ld*`, in
 *     elements) so channel-padded layouts (e.g. 27 -> 28) need no copy;
 *   - a "neighbour table" nbr[n_out][K] (int32, -1 = no neighbour) is the
 *     output-stationary form of ME's kernel map: nbr[o][k] = input row located at
 *     coordinate(o) + offset(k).  K = kernel volume, offsets enumerated x fastest /
 *     z slowest (k = (dx+1) + 3(dy+1) + 9(dz+1) for a 3^3 kernel; witness
 *     sparse_conv.py:375-379).
 */
#ifndef MINK_HIP_H
#define MINK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MINK_OK 0
#define MINK_EINVAL (-1)   /* bad argument (shape / alignment / NULL) */
#define MINK_ELAUNCH (-2)  /* HIP launch or runtime error */

/* Device-side status word bits (written by the coordinate kernels into `status`). */
#define MINK_STATUS_RANGE 1u     /* a coordinate fell outside the packable range */
#define MINK_STATUS_UNSORTED 2u  /* batch column is not non-decreasing */
#define MINK_STATUS_NOT_ASCENDING 4u /* mink_coords_build_levels, informational: the input rows' keys are not strictly ascending
                                        (when CLEAR the rows were their own unique rows and level 0's hash map was left empty) */

const char *mink_last_error(void);
int mink_abi_version(void);

/* ------------------------------------------------------------------ coordinate maps
 * Coordinate hash map: open addressing, linear probing, 64-bit packed keys
 * (16 bits each for batch | x+2^15 | y+2^15 | z+2^15), int32 values.
 * Replaces ME's CoordinateMapCPU/GPU (insert_and_map / stride / kernel_map) behind
 * ME.TensorField(...).sparse() (models/mink/base_model.py:10-13, models/mink/resnet.py:164)
 * and behind every MinkowskiConvolution / MinkowskiSumPooling forward
 * (modules/common.py:116-125, resnet.py:62-64).
 */

/* Number of hash slots (power of two) required for `n` keys. */
int64_t mink_table_capacity(int64_t n);

/* Bytes of scratch required by mink_coords_unique for `n` rows. */
int64_t mink_unique_workspace_bytes(int64_t n);

/* Key generation.  mode 0: float field rows (b,x,y,z) -> floor() (ME A1 quantisation);
 * mode 1: int32 rows copied; both then floor each spatial coordinate to a multiple of
 * `out_ts` (>=1; 1 = identity) which is ME's stride map key functor
 * (CoordinateMap::stride, used by sparse_conv.py:403-405).
 * keys[n] receives the packed keys; *status |= MINK_STATUS_RANGE on overflow. */
int mink_coords_make_keys(const void *coords, int mode, int64_t n, int32_t out_ts, uint64_t *keys,
                          uint32_t *status, void *stream);

/* insert_and_map with FIRST-OCCURRENCE row numbering (what ME's CPU insert_and_map
 * yields for TensorField.sparse(); the same order is used for stride maps, see
 * DESIGN.md "row order").
 *   table_keys[cap], table_vals[cap] : the hash map (cap = mink_table_capacity(n));
 *                                      on return vals hold the unique row id
 *   out_coords[n][4]   : coordinates of the unique rows (first n_unique rows valid)
 *   unique_index[n]    : first input row of each unique row
 *   inverse[n]         : unique row of every input row (the stride map in2out)
 *   n_unique           : device int32 scalar
 */
int mink_coords_unique(const uint64_t *keys, int64_t n, uint64_t *table_keys, int32_t *table_vals,
                       int64_t cap, int32_t *out_coords, int32_t *unique_index, int32_t *inverse,
                       int32_t *n_unique, void *workspace, int64_t workspace_bytes, void *stream);

/* The whole coordinate pyramid of one batch in ONE call and with NO host synchronisation:
 * level 0 = insert_and_map of the input rows (mode as in mink_coords_make_keys, floored to
 * out_ts_host[0], normally 1); level l>0 = stride map of level l-1 floored to out_ts_host[l].
 * Row counts of the levels stay on the device (each level's kernels read the previous count
 * from `meta`), so every buffer is sized for the upper bound `n`:
 *   table_keys[l][cap], table_vals[l][cap], out_coords[l][n][4], index_b[l][n] (inverse map /
 *   in2out of level l), index_a[l][n] (first-occurrence rows; may be NULL for l > 0).
 *   meta[nlev+2] (device int32): unique rows per level, then the status word, then the batch
 *   count (batch index of the last row + 1).  The caller reads `meta` back once.
 * Input rows whose packed keys are strictly ascending (the order of a voxel grid's `links`) need no hash insert at
 * level 0: every row is compared with the one before it, and only when a pair is out of order
 * (MINK_STATUS_NOT_ASCENDING in the status word) does level 0 go through its hash map.  In the ascending case that
 * map is left EMPTY (all slots cleared): a caller that wants look-ups through it inserts the level's rows itself
 * (mink_coords_make_keys + mink_coords_unique).
 * The hash map of level l > 0 occupies only the first mink_table_capacity(rows of level l-1) slots of its
 * buffer -- the capacity for the rows it really receives, computed on the device; a later look-up in it
 * (mink_kernel_map) must be given that capacity, which the caller can compute once it has read `meta`.
 * Replaces the chain TensorField.sparse() -> stride() x5 of one forward pass
 * (resnet.py:164-173, sparse_conv.py:403-405). */
int64_t mink_levels_workspace_bytes(int64_t n);
int mink_coords_build_levels(const void *coords, int mode, int64_t n, int32_t nlev,
                             const int32_t *out_ts_host, uint64_t *const *table_keys,
                             int32_t *const *table_vals, int64_t cap, int32_t *const *out_coords,
                             int32_t *const *index_a, int32_t *const *index_b, int32_t *meta,
                             void *workspace, int64_t workspace_bytes, void *stream);

/* Kernel map as a neighbour table: for every output row o and offset k probe the INPUT
 * map at out_coords[o] + offsets[k].  offsets_host[K][3] are already scaled by
 * dilation * input tensor stride.  If nbr_t != NULL it must be pre-filled with -1
 * ([n_in][K]) and receives the transposed table nbr_t[i][k] = o (used by dgrad).
 * Replaces CoordinateMapManager::kernel_map (witness for the Python-visible format:
 * sparse_conv.py:90-96,124-143). */
int mink_kernel_map(const uint64_t *in_table_keys, const int32_t *in_table_vals, int64_t in_cap,
                    const int32_t *out_coords, int64_t n_out, const int32_t *offsets_host, int32_t K,
                    int32_t *nbr, int32_t *nbr_t, void *stream);

/* Every kernel map of one forward/backward pass in ONE call (one descriptor per map; same
 * semantics as mink_kernel_map, nbr_t is cleared to -1 here).  Cuts the per-map host cost. */
typedef struct MinkKernelMapDesc {
  const uint64_t *in_table_keys;
  const int32_t *in_table_vals;
  int64_t in_cap;
  const int32_t *out_coords;
  int64_t n_out;
  int64_t n_in;
  int32_t *nbr;
  int32_t *nbr_t; /* may be NULL */
  int32_t K;
  int32_t offsets[81]; /* [K][3], scaled by dilation * input tensor stride */
  /* Optional block index of the INPUT map (blk_table != NULL): neighbour look-ups go through a second, much smaller
   * hash map keyed by the 4x4x4-cell BLOCK of a coordinate instead of through the per-voxel map.  A block entry
   * holds the occupancy mask of its 64 cells and the start of its run in `blk_rowids`; the row of a cell is
   * blk_rowids[base + popcount(mask below the cell)].  The 27 neighbours of a voxel fall into at most 8 blocks and
   * neighbouring voxels share them, so a wave touches a handful of cache lines where the per-voxel map (one random
   * 64-byte line for the key and one for the value, per probe) touched thousands: 1.5 GB of HBM traffic per stem table
   * became the size of the table itself.  Results are identical (bit-exact tables).  Descriptors that share an input
   * map share the buffers; exactly the first of them sets blk_build = 1 and the index is built by that call. */
  const int32_t *in_coords; /* [n_in][4] rows of the input map */
  int32_t in_ts;            /* its tensor stride (cells are coordinates / in_ts) */
  int32_t blk_build;
  uint64_t *blk_table;      /* [blk_cap][2]: block key, inverted occupancy mask */
  int32_t *blk_base;        /* [blk_cap] */
  int32_t *blk_slot;        /* [n_in] scratch: block slot of every input row */
  int32_t *blk_rowids;      /* [max(n_in, 8)] */
  int32_t *blk_counter;     /* [1]: the build fills it with 0xFFFFFFFF (= every block found a slot) and writes 0 if blk_cap was too
                               small (rows of blocks that found no slot are then missing from the tables: the caller's error,
                               reported instead of a hang; see also mink_set_overflow_sink) */
  int64_t blk_cap;          /* power of two >= 2 * the number of occupied blocks (<= n_in; the blocks of the map at tensor stride ts
                               are the cells of the map at 4 ts, so a caller that holds that map knows the number) */
} MinkKernelMapDesc;
int mink_kernel_map_batch(int32_t n, const MinkKernelMapDesc *descs, void *stream);

/* Process-wide overflow sink of the block-index builds: a 32-bit word in PINNED host memory (device-mapped; NULL = none), zero
 * initialised by the caller.  A build whose blk_cap was too small writes 1 into it (besides clearing its own blk_counter) and
 * nobody ever refills it, so a caller that looks at the word before queuing its next batch learns of a violated capacity
 * promise without a device synchronisation or a copy -- one batch late at most, instead of silently missing neighbours. */
int mink_set_overflow_sink(int32_t *host_word);

/* ME-format rulebook from a neighbour table: per offset k the (in,out) pairs ordered
 * by output row, built with wave64 ballot + prefix sums.
 *   counts[K+1]      : exclusive scan of pairs per offset (device, int32)
 *   pairs_in/out[P]  : concatenated lists (P <= n_out*K), may be NULL to only count
 *   workspace        : >= mink_rulebook_workspace_bytes(n_out, K)
 */
int64_t mink_rulebook_workspace_bytes(int64_t n_out, int32_t K);
int mink_rulebook(const int32_t *nbr, int64_t n_out, int32_t K, int32_t *counts, int32_t *pairs_in,
                  int32_t *pairs_out, void *workspace, int64_t workspace_bytes, void *stream);

/* Parity-class permutation of the rows of a tensor-stride-`ts` map: rows grouped by the parity
 * of (coordinate / ts) per axis (8 classes), each class segment padded to a multiple of `pad`
 * rows with -1.  perm[mink_class_partition_rows(n,pad)].  Used by the input-gradient of
 * stride-2 convolutions (resnet_block.py:29-37 with stride 2): all rows of one class are reached
 * through the same 2^p kernel offsets, so class-pure tiles skip the other offsets. */
int64_t mink_class_partition_rows(int64_t n, int32_t pad);
int64_t mink_class_partition_workspace_bytes(int64_t n);
int mink_class_partition(const int32_t *coords, int64_t n, int32_t ts, int32_t pad, int32_t *perm,
                         void *workspace, int64_t workspace_bytes, void *stream);
/* The same for up to eight maps in four launches (fill, count, scan, place; blockIdx.y = map) instead of four per map:
 * what a prepared batch asks for (one permutation per strided convolution of the network). */
typedef struct MinkClassPartitionDesc {
  const int32_t *coords; /* [n][4] */
  int64_t n;
  int32_t ts, pad;
  int32_t *perm;         /* mink_class_partition_rows(n, pad) entries */
  void *workspace;       /* mink_class_partition_workspace_bytes(n), 256-byte aligned */
  int64_t workspace_bytes;
} MinkClassPartitionDesc;
int mink_class_partition_batch(int32_t n_maps, const MinkClassPartitionDesc *descs, void *stream);

/* Row ranges of each batch index in a coordinate list whose batch column is
 * non-decreasing (guaranteed by ME.utils.sparse_collate, data/utils.py:25-30).
 * batch_offsets[B+1].  Sets MINK_STATUS_UNSORTED otherwise.  (ME origin_map.) */
int mink_batch_offsets(const int32_t *coords, int64_t n, int32_t B, int32_t *batch_offsets,
                       uint32_t *status, void *stream);

/* ------------------------------------------------------------------ sparse convolution
 * Replaces ME ConvolutionForward/BackwardKernel behind ME.MinkowskiConvolution
 * (modules/common.py:116-125; the algorithm is re-stated by the reference at
 * sparse_conv.py:122-144): out[o] = sum_k in[nbr[o][k]] @ W[k]  (+ bias).
 */

/* Output-stationary gather-GEMM on the fp32 matrix cores.
 *   x[n_in][ldx], cin  : gathered operand (every table entry is -1 or < n_in)
 *   w                  : w_transposed == 0:  w[K][cin][cout]   (forward)
 *                        w_transposed == 1:  w[K][cout][cin]   (dgrad: pass the forward
 *                                            kernel and swap cin/cout)
 *   flip_k             : use w[K-1-k] for table column k (dgrad of a stride-1 conv
 *                        re-uses the forward table: nbr_t[i][k] == nbr[i][K-1-k])
 *   nbr[n_out][K]      : neighbour table;  y[n_out][ldy] receives cout columns.  (cin == 28, the flattened-K
 *                        stem path: taken only when n_in < 2^24 - 1 -- its rows are addressed with a 24-bit
 *                        multiply; larger inputs take the general path.)
 *   bias[cout] or NULL
 *   row_perm/n_virtual : optional row permutation (NULL/0 = identity): tile row v computes
 *                        output row row_perm[v], -1 entries are padding (mink_class_partition)
 *   ksplit             : >1 splits the K offsets over `ksplit` workgroups per tile and
 *                        reduces through `workspace` (ksplit*n_out*cout floats).  With a class
 *                        permutation of an fp32 mid layer (cin >= 64, K >= 8) the split is over
 *                        the cin/32 channel chunks instead (<= cin/32 slices; a tile of one
 *                        parity class has few live offsets, every slice sees all of them)
 */
/* Test / measurement knob (never needed by a caller): bit 9 = no flattened-K stem path, bit 10 = tiled instead of streaming
 * stem weight gradient, bits 12-15 / 16-27 = force the weight-gradient offset grouping / row-split count (sweeps; this is the
 * plan change the workspace_bytes arguments guard against), bit 28 = bf16 math keeps the fp32 stem weight gradient, bit 29 =
 * plain workgroup order in the streaming weight gradient, bit 30 / 31 = the dense gather-GEMM instead of the row-compacted
 * kernel for the mid layers / the strided data gradients (the tests compare the two).  Returns the previous low byte. */
int mink_conv_set_stagger(int units);

/* Which software pipeline the row-compacted mid-layer kernel (compact_gemm_kernel) runs: bit 0 = the stride-1 forward / data
 * gradient launches, bit 1 = the class-permuted strided data gradients on the THREE-STAGE form (three LDS stages of the gathered-row
 * tile, MFMA operands read one item ahead; three workgroups per CU), 0 = the two-stage form (four workgroups per CU).  Results are
 * bit-identical either way.  Returns the previous mode. */
int mink_conv_set_pipeline(int mode);

/* Measurement only: while `buf` (device memory, 5 x uint64 per workgroup, `capacity_workgroups` of them) is set, the mid-layer
 * forward kernel runs the instantiation that carries the timing switches and every workgroup records the shader clock at its start,
 * at the head of its item loop, at its epilogue and at its end (stores drained), plus its hardware id (scripts/kbench.py ctrace).
 * NULL: off. */
int mink_conv_trace(void *buf, int64_t capacity_workgroups);
/* Matrix-core arithmetic of mink_conv_gather_gemm (forward / input gradient):
 *   0 = exact fp32 MFMA (default), 1 = bf16 operands with fp32 accumulation (BASELINE config
 *   "bf16 mixed precision"), 3 = split-bf16 (hi/lo, three products; ~1e-5 relative).
 * HBM tensors stay fp32 in every mode.  Returns the previous mode. */
int mink_conv_set_math(int mode);
int mink_conv_get_math(void); /* the current mode, unchanged */
/* Split-K factor the library recommends for a layer (1 for large row counts).  row_classes != 0:
 * the launch will pass a class-partitioned row_perm (stride-2 dgrad), n_out = its n_virtual. */
int mink_conv_plan_ksplit(int64_t n_out, int32_t K, int32_t cout, int32_t row_classes);
/* The same with the reduction width known (what the block sequencer and the Python layer ask): for a class-permuted fp32 mid
 * layer (row_classes != 0) the channel-chunk split of the row-compacted kernel, for an fp32 mid layer its offset split (<= 14
 * slabs), otherwise mink_conv_plan_ksplit's answer.  The workspace is 4 * ksplit * n_out * cout bytes in every case. */
int mink_conv_plan(int64_t n_rows, int32_t K, int32_t cin, int32_t cout, int32_t row_classes);
/* flip_k: bit 0 = read the weights of offset K-1-k for offset k (data gradient of a stride-1 convolution through the
 * forward table); bit 1 = ACCUMULATE, y[row] += result instead of y[row] = result -- for an un-split launch whose
 * row_perm visits every output row at most once (rows it does not visit are left untouched): the data gradient of
 * the 1x1x1 strided shortcut convolution is added into the main branch's gradient at the 1/8 of the rows it reaches. */
int mink_conv_gather_gemm(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *w, int32_t w_transposed,
                          int32_t flip_k, const int32_t *nbr, int64_t n_out, int32_t K,
                          const int32_t *row_perm, int64_t n_virtual, float *y, int32_t ldy,
                          int32_t cout, const float *bias, int32_t ksplit, float *workspace,
                          int64_t workspace_bytes, void *stream);
/* Forward convolution that also emits the column statistics of its output for the batch norm that
 * follows (reference resnet_block.py:53-60: every 3x3x3 convolution feeds a norm): the un-split
 * kernel sums its tile in the epilogue, the split-K reduce sums while it adds the slabs -- y is
 * not read again.  stats_out: double [<= 512][2][cout]; *stats_rows (host) = partial rows written,
 * 0 when this launch shape cannot produce them (the caller then calls mink_bn_stats);
 * stats_ws >= mink_conv_stats_workspace_bytes(n_out, cout). */
int64_t mink_conv_stats_workspace_bytes(int64_t n_out, int32_t cout);
int mink_conv_gather_gemm_stats(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *w, const int32_t *nbr,
                                int64_t n_out, int32_t K, float *y, int32_t ldy, int32_t cout,
                                const float *bias, int32_t ksplit, float *workspace, int64_t workspace_bytes,
                                double *stats_out, int32_t *stats_rows, void *stats_ws, int64_t stats_ws_bytes, void *stream);

/* Weight gradient dW[k] = X[nbr[.][k]]^T @ dY, split over row blocks and reduced
 * deterministically (no atomics).  x has n_in rows (every nbr entry is -1 or in [0, n_in)).
 * workspace >= mink_conv_wgrad_workspace_bytes(). */
int64_t mink_conv_wgrad_workspace_bytes(int64_t n_out, int32_t K, int32_t cin, int32_t cout);
int mink_conv_wgrad(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *dy, int32_t ldy,
                    int32_t cout, const int32_t *nbr, int64_t n_out, int32_t K, float *dw, void *workspace,
                    int64_t workspace_bytes, void *stream);
/* Weight gradient of a convolution whose output y feeds  pool(relu(bn(y)))  (the stem of the
 * reference ResNets, resnet.py:58-64) and whose input needs no gradient: the gradient with respect
 * to y is recomputed from (y, pooled gradient, batch statistics, dgamma, dbeta -- mink_bn_relu_pool_bwd
 * with dx = NULL) inside the kernel's operand load and never written to memory.
 * Only for shapes the streaming kernel takes (K = 27, cin <= 32, many rows): ask _supported() first. */
int mink_conv_wgrad_bn_relu_pool_supported(int64_t n_in, int32_t ldx, int32_t cin, int64_t n_out, int32_t K,
                                           int32_t cout);
int mink_conv_wgrad_bn_relu_pool(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *y, int32_t cout,
                                 const float *dy_pool, int64_t n_pool, const int32_t *in2out, const float *mean,
                                 const float *invstd, const float *gamma, const float *beta,
                                 const float *dgamma, const float *dbeta, const int32_t *nbr, int64_t n_out,
                                 int32_t K, float *dw, void *workspace, int64_t workspace_bytes, void *stream);

/* Data gradient of a 1x1x1 convolution as a plain GEMM: y[n][N] = x[n][Kd] @ W^T with the FORWARD kernel W[N = cin][Kd = cout]
 * (the `downsample` convolution of a residual block, models/mink/resnet.py:120-128), one fp32 MFMA accumulator chain per
 * element, k ascending -- the same sum mink_conv_gather_gemm forms for a one-column table.  Kd a multiple of 4. */
int mink_dense_xwt(const float *x, const float *w, int64_t n, int32_t Kd, int32_t N, float *y, void *stream);
/* dst[idx[r]][0..C) += src[r][0..C) for every r < n_src with idx[r] >= 0; the idx values must be distinct (no atomics).
 * With idx = the forward table of a kernel-volume-1 strided convolution this adds the shortcut's data gradient into the
 * rows of the block-input gradient it reaches (about one in eight). */
int mink_rows_scatter_add(const float *src, const int32_t *idx, int64_t n_src, int32_t C, float *dst, void *stream);

/* ------------------------------------------------------------------ pooling / reductions
 * MinkowskiSumPooling(k=2,s=2) (resnet.py:62-64): out[o] = sum_k in[nbr[o][k]] with the
 * 2^3 children table; backward dIn[i] = dOut[in2out[i]].
 */
int mink_pool_sum_fwd(const float *x, int32_t ldx, int32_t C, const int32_t *nbr, int64_t n_out, int32_t K,
                      float *y, void *stream);
int mink_pool_sum_bwd(const float *dy, int32_t C, const int32_t *in2out, int64_t n_in, float *dx,
                      void *stream);

/* MinkowskiGlobalAvgPooling (resnet.py:15-22,175): y[b] = mean of rows
 * [batch_offsets[b], batch_offsets[b+1]); backward dx[i] = dy[b]/N_b. */
int mink_global_avg_fwd(const float *x, int32_t C, const int32_t *batch_offsets, int32_t B, float *y,
                        void *stream);
int mink_global_avg_bwd(const float *dy, int32_t C, const int32_t *batch_offsets, int32_t B, int64_t n,
                        float *dx, void *stream);

/* Classifier head in one launch each way: logits[b] = mean_{rows of batch b} x[row] @ W + bias, i.e.
 * MinkowskiGlobalAvgPooling followed by the kernel-volume-1 `final` convolution with bias (models/mink/resnet.py:15-22,
 * 93-99,175-177; W is ME's (Cin, Cout) kernel of a `use_mm` convolution, bias (1, Cout) or NULL).  pooled[B][C] is an
 * output the backward pass reads.  Backward: dx[n][C] (may be NULL), dw[C][ncls], dbias[ncls] (may be NULL). */
int mink_head_forward(const float *x, const int32_t *batch_offsets, int32_t B, int32_t C, const float *w, const float *bias,
                      int32_t ncls, float *pooled, float *logits, void *stream);
int mink_head_backward(const float *dlogits, const float *pooled, const float *w, const int32_t *batch_offsets, int32_t B,
                       int32_t C, int32_t ncls, float *dw, float *dbias, float *dx, void *stream);
/* torch.nn.functional.cross_entropy(logits, labels) with its defaults (mean over the batch; reference
 * modules/classification_training.py:33) as one launch each way.  labels: int64 [B]; prob[B][ncls] (softmax) is kept for
 * backward; *loss a device scalar; a label outside [0, ncls) makes the loss NaN.  grad_loss: device scalar. */
int mink_softmax_ce_forward(const float *logits, const int64_t *labels, int32_t B, int32_t ncls, float *prob, float *loss,
                            void *stream);
int mink_softmax_ce_backward(const float *prob, const int64_t *labels, const float *grad_loss, int32_t B, int32_t ncls,
                             float *dlogits, void *stream);

/* TensorField.sparse() feature averaging (ME UNWEIGHTED_AVERAGE, resnet.py:164):
 * y[u] = mean_{j in [seg[u],seg[u+1])} x[members[j]] (members sorted by input row). */
int mink_segment_mean(const float *x, int32_t ldx, int32_t C, const int32_t *members, const int32_t *seg,
                      int64_t n_out, float *y, void *stream);

/* ------------------------------------------------------------------ batch norm / relu / add
 * MinkowskiBatchNorm == torch.nn.BatchNorm1d on the feature matrix
 * (modules/common.py:22-24; witness resnet.py:101-105), MinkowskiReLU (resnet.py:61),
 * residual `out += residual` (modules/resnet_block.py:66).
 */
int64_t mink_bn_workspace_bytes(int64_t n, int32_t C);

/* Batch statistics: mean[C], invstd[C] = 1/sqrt(biased var + eps); if running_mean !=
 * NULL also updates running stats with `momentum` (unbiased variance, torch semantics). */
int mink_bn_stats(const float *x, int64_t n, int32_t C, float eps, float momentum, float *mean,
                  float *invstd, float *running_mean, float *running_var, void *workspace, int64_t workspace_bytes,
                  void *stream);

/* y = [relu]( (x-mean)*invstd*gamma + beta [+ residual] ) */
int mink_bn_apply(const float *x, int64_t n, int32_t C, const float *mean, const float *invstd,
                  const float *gamma, const float *beta, const float *residual, int32_t relu, float *y,
                  void *stream);
/* Training-mode forward in one call: mink_bn_stats followed by mink_bn_apply (mean / invstd are
 * outputs kept for backward).  A single-launch variant for small layers (one workgroup per four
 * channels over all rows) was measured 3-9x slower than the three launches (strided 16-byte
 * column reads) and is not provided. */
int mink_bn_fwd(const float *x, int64_t n, int32_t C, float eps, float momentum, const float *gamma,
                const float *beta, const float *residual, int32_t relu, float *y, float *mean, float *invstd,
                float *running_mean, float *running_var, void *workspace, int64_t workspace_bytes, void *stream);
/* Statistics from column partials [rows][2][C] (sum, sum of squares; double) that the producer of
 * x already computed -- mink_conv_gather_gemm_stats -- instead of a reduction pass over x. */
int mink_bn_stats_from_partials(const double *partial, int32_t rows, int64_t n, int32_t C, float eps,
                                float momentum, float *mean, float *invstd, float *running_mean,
                                float *running_var, void *stream);

/* Backward of the op above.  y (the forward output) is only read when relu != 0.
 * dgamma[C], dbeta[C]; dx[n][C]; dresidual (may be NULL) receives the masked grad. */
int mink_bn_bwd(const float *dy, const float *x, const float *y, int64_t n, int32_t C, const float *mean,
                const float *invstd, const float *gamma, int32_t relu, float *dx, float *dresidual,
                float *dgamma, float *dbeta, void *workspace, int64_t workspace_bytes, void *stream);

/* Split form for MinkowskiSyncBatchNorm (train.py:106-107): per-channel sums stay on the device
 * as doubles so the host can all-reduce them (RCCL) between the passes.
 *   mink_bn_reduce mode 0: sums = [sum x | sum x^2];  mode 1: [sum g | sum g*xhat] (g masked by y>0
 *   when y != NULL).  n_total: device double (rows over all ranks). */
int mink_bn_reduce(int32_t mode, const float *a, const float *b, const float *y, int64_t n, int32_t C,
                   const float *mean, const float *invstd, double *sums, void *workspace, int64_t workspace_bytes, void *stream);
int mink_bn_stats_from_sums(const double *sums, const double *n_total, int32_t C, float eps, float momentum,
                            float *mean, float *invstd, float *running_mean, float *running_var,
                            void *stream);
int mink_bn_bwd_from_sums(const float *dy, const float *x, const float *y, int64_t n, int32_t C,
                          const double *sums, const double *n_total, const float *mean,
                          const float *invstd, const float *gamma, int32_t relu, float *dx,
                          float *dresidual, float *scratch2c, void *stream);

/* Fused tail of the stem: y_pool = SumPool(relu(BN(x))) without materialising the normalised
 * [n,C] tensor (bn1 -> relu -> pool, resnet.py:58-64).  `nbr[n_out][K]` is the 2^3 children
 * table, `in2out[n]` the stride map.  Statistics come from mink_bn_stats; the ReLU mask of the
 * backward pass is recomputed from x. */
int mink_bn_relu_pool_fwd(const float *x, int32_t C, const float *mean, const float *invstd,
                          const float *gamma, const float *beta, const int32_t *nbr, int64_t n_out,
                          int32_t K, float *y, void *stream);
int mink_bn_relu_pool_bwd(const float *dy_pool, const float *x, int64_t n, int32_t C, const float *mean,
                          const float *invstd, const float *gamma, const float *beta,
                          const int32_t *in2out, float *dx, float *dgamma, float *dbeta,
                          void *workspace, int64_t workspace_bytes, void *stream);

/* Max pooling over a neighbour table whose windows may overlap -- the 3x3 stride-2 pooling of the dense 2-D
 * comparison network (torchvision ResNet, reference co3d_2d/src/model/models.py:18-23), also ME.MinkowskiMaxPooling.
 * arg[n_out][C] receives the input row of each maximum; backward gathers through the transposed table nbr_t[n_in][K]
 * (the windows containing a row), so there are no atomics. */
int mink_pool_max_fwd(const float *x, int32_t C, const int32_t *nbr, int64_t n_out, int32_t K, float *y, int32_t *arg,
                      void *stream);
int mink_pool_max_bwd(const float *dy, const int32_t *arg, int32_t C, const int32_t *nbr_t, int64_t n_in, int32_t K,
                      float *dx, void *stream);

/* Elementwise: mode 0: y = max(x,0); mode 1: dx = (y>0) ? dy : 0 (a=dy,b=y);
 * mode 2: y = a + b. */
int mink_eltwise(const float *a, const float *b, int64_t count, int32_t mode, float *y, void *stream);

/* Pointwise activations beyond ReLU -- ME.MinkowskiLeakyReLU / ELU / CELU / SELU / GELU / PReLU, which the reference's
 * layer factory lists at import time (co3d_3d/src/models/mink/modules/common.py:36-43) and MinkowskiFunctional mirrors
 * (:56-71).  kind: 1 leaky-ReLU (alpha = negative slope), 2 ELU(alpha), 3 CELU(alpha), 4 SELU, 5 GELU (erf form),
 * 6 PReLU (`slope`: C per-channel weights, or one shared weight when C == 1; channels are the fastest axis).
 * gy == NULL: out = f(x) over `count` elements; otherwise out = gy * f'(x). */
int mink_activation(const float *x, const float *gy, const float *slope, int32_t C, int64_t count, int32_t kind,
                    float alpha, float *out, void *stream);

/* ------------------------------------------------------------------ dataset front-end
 * PeRFception-CO3D `data.npz` batch on the device (reference co3d.py:160-166 de-quantisation,
 * :196-205 links -> coordinates and feature selection): `links[n]` flat indices x*ry*rz + y*rz + z,
 * `density[n]`, `sh_q[n][27]` uint8, per scene b (rows scene_offsets[b] .. scene_offsets[b+1])
 * `sh_scale[b][27]`, `sh_min[b][27]`.  Writes coords int32 [n][4] = (b, x, y, z) and the selected
 * feature columns: density at col_density, the 27 de-quantised SH coefficients at col_sh, ones at
 * col_ones, the three "xyzs" columns at col_xyzs (-1 = not selected; the columns must tile 0..C-1).
 * sh = float(sh_q) * scale + min with separate multiply and add, bit-identical to the numpy expression of
 * the reference.  xyzs (co3d.py:209-214, `configs/feature_coord.gin`): every point minus the mean of its own
 * three coordinates, divided by the largest such norm of its scene -- a per-scene reduction, done in a
 * first pass into scene_scratch[n_scenes] (required when col_xyzs >= 0); every operation rounded separately. */
int mink_decode_plenoxel(const int32_t *links, const float *density, const uint8_t *sh_q,
                         const int32_t *scene_offsets, int32_t n_scenes, const float *sh_scale,
                         const float *sh_min, int64_t n, int32_t reso_y, int32_t reso_z, int32_t col_density,
                         int32_t col_sh, int32_t col_ones, int32_t col_xyzs, float *scene_scratch, int32_t C,
                         int32_t *coords, float *feats, int32_t ldf, void *stream);

/* ------------------------------------------------------------------ scene augmentation
 * The reference augments every scene on the CPU in its DataLoader workers (co3d.py:216-219 applies
 * transforms.Compose of the gin-listed classes of transforms.py; configs/co3d_aug3.gin:2-23).  Here
 * the host only DRAWS the per-scene randomness and the whole batch is transformed on the device.
 * Per scene b a row of MINK_AUG_PARAMS floats (all matrices row-major, applied as row-vector * M):
 *     p = c * A + a                          stages before the flip (rotation, affine, ...)
 *     q_j = FLIP[j] ? max_j - p_j : p_j      max over the scene's surviving voxels (over all voxels
 *                                            when FLIP_ALL != 0: dropout listed after the flip)
 *     r = q * B + b + jitter * BJ            jitter_j = JITTER * (u_j - 0.5), u uniform [0,1)
 *     voxel kept iff u >= DROPOUT            (CoordinateDropout; 0 keeps everything)
 *     feats[:, raw column in [FEAT_START, FEAT_START + FEAT_DIM)] += (normal - 0.5) * FEAT_STD
 * (transforms.py: RandomRotation :351-358, RandomAffine :416-427, CoordinateDropout :255-265,
 * RandomHorizontalFlip :444-450, CoordinateUniformTranslation :288-294, CoordinateJitter :276-281,
 * RandomScale :367-373, RandomFeatureJitter :34-41.)  Differences from the reference, by design:
 * the dropout is an independent coin per voxel and keeps the voxel order (the reference draws an
 * exact-size random subset in random order); the per-voxel random numbers are Philox4x32-10 with
 * key = seed, counter = (voxel index in its scene, draw, streams[b], 0): draw 0 -> (dropout u, jitter
 * u_x, u_y, u_z), draw 1+j/4 -> four Box-Muller normals for noise columns 4(j/4)..4(j/4)+3.
 * Coordinates are computed in fp32 with every product and sum rounded separately, in the order
 * ((v0*M0j + v1*M1j) + v2*M2j) + t_j, so a CPU restatement reproduces them bit for bit. */
enum {
  MINK_AUG_A = 0,           /* 9 */
  MINK_AUG_a = 9,           /* 3 */
  MINK_AUG_FLIP = 12,       /* 3 flags */
  MINK_AUG_B = 15,          /* 9 */
  MINK_AUG_b = 24,          /* 3 */
  MINK_AUG_BJ = 27,         /* 9 */
  MINK_AUG_JITTER = 36,     /* 2 * jitter_std, 0 = none */
  MINK_AUG_DROPOUT = 37,    /* dropout ratio, 0 = none */
  MINK_AUG_FEAT_STD = 38,   /* 0 = none */
  MINK_AUG_FEAT_START = 39, /* raw-layout column, co3d.py:205-214: [xyzs 0:3 | density 3 | sh 4:31] */
  MINK_AUG_FEAT_DIM = 40,
  MINK_AUG_FLIP_ALL = 41,
  MINK_AUG_PARAMS = 44,
  MINK_AUG_MAX_CHANNELS = 32
};

int64_t mink_augment_workspace_bytes(int64_t n, int32_t n_scenes);

/* coords [n][4] (batch, x, y, z) sorted by batch, float32 or -- coords_are_int32 -- int32 as written by
 * mink_decode_plenoxel; feats [n][ldf] (C columns used);
 * scene_offsets [n_scenes+1], params [n_scenes][MINK_AUG_PARAMS], streams [n_scenes] on the device.
 * raw_cols[C] is a HOST array: the raw-layout column of every feature column (-1 = never jittered).
 * Writes the survivors, in order, to out_coords [>= n][4] / out_feats [>= n][ldo] and their number
 * to the device int *n_kept (n_kept == n whenever every DROPOUT is 0: no read-back needed then). */
int mink_augment_scenes(const void *coords, int32_t coords_are_int32, const float *feats, int64_t ldf, int32_t C, int64_t n,
                        const int32_t *scene_offsets, int32_t n_scenes, const float *params,
                        const uint32_t *streams, uint64_t seed, const int32_t *raw_cols, float *out_coords,
                        float *out_feats, int64_t ldo, int32_t *n_kept, void *workspace, int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ optimizer step over flat buffers
 * torch.optim.SGD's update with momentum and weight decay (the reference's optimizer: co3d_3d/src/modules/optim.py:12-14,
 * co3d_3d/configs/co3d_cls.gin) for parameters w, gradients g and momentum buffers m that each live in ONE flat fp32 buffer of
 * the same layout (n floats, a multiple of 4; padding elements are zero and stay zero):
 *   g' = g + weight_decay w;  m = momentum m + g';  w = w - lr m        (dampening 0, no Nesterov; a zero m = torch's first step)
 * zero_grad != 0 clears g behind the update (the next step's zero_grad memset).  One pass over 6 n floats. */
int mink_sgd_step(float *w, float *g, float *m, int64_t n, float lr, float momentum, float weight_decay, int32_t zero_grad, void *stream);

/* ------------------------------------------------------------------ whole residual blocks
 * One call = the complete launch sequence of one stage of the reference network, so the host pays one FFI call per
 * block instead of one per kernel (a Mink-ResNet14 step is ~300 launches; issued one by one from a Python autograd
 * graph the host, not the GPU, was the bottleneck).  Same kernels, same order, same results as the per-operator
 * entry points above -- these functions only sequence them:
 *
 *   mink_stem_forward/backward   conv1 -> bn1 -> relu -> SumPooling(2,2)            (models/mink/resnet.py:58-64,163-166)
 *   mink_block_forward/backward  BasicBlock: conv1-norm1-relu-conv2-norm2 (+downsample(x) or x) -relu
 *                                                                                   (modules/resnet_block.py:53-69)
 *
 * Three streams take part (any two may be the same stream): `compute` runs the chain, `branch` runs the shortcut
 * (1x1x1 strided convolution + norm) beside it, `wgrad` runs the weight gradients beside the data-gradient chain.
 * Cross-stream ordering inside a call uses events owned by the library; on return everything the NEXT call on
 * `compute` needs is ordered on `compute`, EXCEPT the weight gradients (dw of every convolution), which are complete
 * on `wgrad` -- the caller joins `wgrad` once, at the end of the backward pass.  Buffers must stay alive until then.
 * Batch norm runs in training mode (batch statistics; running statistics updated in place when given). */
typedef struct {
  const float *w;        /* [K][cin][cout] */
  float *dw;             /* backward: weight gradient [K][cin][cout] (written, not accumulated) */
  const int32_t *nbr;    /* [n_out][K] neighbour table in -> out */
  const int32_t *nbr_t;  /* backward, stride > 1: transposed table [n_in][K]; NULL for stride 1 (the flipped table) */
  const int32_t *perm;   /* backward, stride > 1: parity-class row order of the input map (mink_class_partition) */
  int64_t n_perm;        /* rows of `perm` (0 without one) */
  int32_t K, cin, cout, stride;
} MinkConvLayer;

typedef struct {
  const float *gamma, *beta;          /* [C] */
  float *running_mean, *running_var;  /* [C] or both NULL */
  float *dgamma, *dbeta;              /* backward: [C] (written) */
  float *mean, *invstd;               /* [C] batch statistics: written by forward, read by backward */
  float momentum, eps;
} MinkNormLayer;

typedef struct {
  void *compute, *branch, *wgrad;     /* hipStream_t */
  void *ws_compute, *ws_branch, *ws_wgrad; /* scratch per stream, 256-byte aligned */
  int64_t ws_bytes;                   /* size of EACH scratch buffer (>= mink_block_workspace_bytes) */
} MinkExec;

typedef struct {
  MinkConvLayer conv;   /* 3^3, stride 1, no bias; cin a multiple of 4 */
  MinkNormLayer norm;
  const int32_t *nbr_pool;  /* [n_pool][8] children table of the 2^3 sum pooling */
  const int32_t *in2out;    /* [n] row -> pooled row */
  int64_t n, n_pool;
  const float *x;  /* [n][cin] */
  float *y;        /* [n][cout] convolution output (kept for backward) */
  float *out;      /* [n_pool][cout] */
  const float *g_out;  /* backward: gradient of `out` */
  void *xb;        /* ABI 3, bf16 storage (needs mink_conv_set_math(1)): [n][32] bf16 copy of x, written by forward and read by
                      backward; `y` is then bf16 [n][cout].  NULL: fp32 storage */
  int32_t xb_ready; /* nonzero: `xb` already holds mink_rows_to_bf16(x) (made ahead of the step, e.g. beside the previous one):
                      forward does not write it */
} MinkStem;

typedef struct {
  MinkConvLayer conv1, conv2, down;   /* down.w == NULL: identity shortcut (then n_in == n_out, cin == cout) */
  MinkNormLayer norm1, norm2, normd;
  int64_t n_in, n_out;
  const float *x;   /* [n_in][cin] */
  float *y1, *h1, *y2, *yd, *sd, *out;  /* [n_out][planes] each: conv1 out, relu(norm1), conv2 out, shortcut conv out,
                                           normalised shortcut, block output (yd, sd unused without a down path) */
  /* backward */
  const float *g_out;  /* [n_out][planes] */
  float *g_x;          /* [n_in][cin], or NULL when the block input needs no gradient */
  float *g_tmp;        /* scratch for the intermediate gradients: mink_block_grad_scratch_floats() floats */
} MinkBasicBlock;

int64_t mink_block_workspace_bytes(int64_t n_in, int64_t n_out, int32_t cin, int32_t cout);
int64_t mink_block_grad_scratch_floats(int64_t n_in, int64_t n_out, int32_t cin, int32_t cout, int32_t has_down);
/* 1 when mink_stem_forward/backward accept this stem (else compose it from the per-operator calls) */
int mink_stem_supported(int64_t n, int32_t cin, int32_t cout, int32_t K);
int mink_stem_forward(const MinkStem *s, const MinkExec *ex);
int mink_stem_backward(const MinkStem *s, const MinkExec *ex);
int mink_block_forward(const MinkBasicBlock *b, const MinkExec *ex);
int mink_block_backward(const MinkBasicBlock *b, const MinkExec *ex);

/* ------------------------------------------------------------------ batch-norm finalize inside the apply pass
 * mink_bn_apply_from_partials = mink_bn_stats_from_partials + mink_bn_apply (the reference's MinkowskiBatchNorm forward,
 * co3d_3d/src/models/mink/modules/common.py:22-24) -- as ONE launch when `rows` is at most the fold limit (mink_bn_set_fold; at
 * most 128; DEFAULT 0 = never: see below), re-reading the partials in every workgroup of the pass costs at most ~8 MB in total,
 * and C is a multiple of 64: every workgroup of the apply pass sums the partial rows of its 64 channels itself, in the
 * order of the finalize kernel (results are bit-identical to the two-call sequence; mean / invstd / running statistics are
 * written by the first row chunk).  mink_bn_bwd folds its finalize into its apply pass under the same conditions.  Every
 * workgroup re-reads rows x 1 KB of partials and starts its pass behind that: measured SLOWER than the separate finalize launch
 * in every configuration (B=16: +5 % folding every layer, +-0 with the bytes rule; ResNet34 at B=4: +2.5 % with the bytes rule
 * -- DESIGN.md Appendix A), so it is off by default and kept as a tested, bit-identical option.  mink_bn_set_fold(max_rows) sets the limit (0: never; 1000 + r: r rows
 * without the total-bytes rule, for tests) and returns the previous one. */
int mink_bn_set_fold(int32_t max_rows);
/* mink_bn_bwd whose incoming gradient is still the `nslab` split-K slabs ([nslab][n][C] at dy_slabs) of the data-gradient
 * convolution that produced it (mink_conv_gather_gemm_slabs): the column-reduction pass sums them in slab order -- what
 * splitk_reduce would have written, bit for bit --, adds `addend` behind them (optional [n][C]: the gradient of a residual branch,
 * what mink_eltwise(.., 2, ..) would have added) and stores the sum to dy_sum on the way.  One or two launches and passes over
 * the gradient less per convolution backward.  C <= 1024. */
int mink_bn_bwd_slabs(const float *dy_slabs, int32_t nslab, const float *addend, float *dy_sum, const float *x, const float *y, int64_t n, int32_t C,
                      const float *mean, const float *invstd, const float *gamma, int32_t relu, float *dx, float *dresidual,
                      float *dgamma, float *dbeta, void *workspace, int64_t workspace_bytes, void *stream);
int mink_bn_apply_from_partials(const float *x, int64_t n, int32_t C, const double *partial, int32_t rows, float eps, float momentum,
                                const float *gamma, const float *beta, const float *residual, int32_t relu, float *y, float *mean,
                                float *invstd, float *running_mean, float *running_var, void *stream);

/* ------------------------------------------------------------------ few-row layers: batch norm in one launch
 * Below mink_bn_small_rows() rows (1024; 0 when switched off with mink_bn_set_small(0)) a batch norm as three dependent launches
 * (column partials, finalize, apply: the reference's MinkowskiBatchNorm = nn.BatchNorm1d, co3d_3d/src/models/mink/modules/
 * common.py:22-24, resnet_block.py:53-69) is launch latency, not bytes.  Here a workgroup owns 16 channels and ALL rows:
 *   mink_conv_gather_gemm_slabs  mink_conv_gather_gemm that LEAVES a split launch's partial slabs in `workspace`
 *                                ([*slabs_out][n_out][cout]; *slabs_out == 1: y is complete) -- no bias
 *   mink_bn_small_fwd            y = sum of `nslab` slabs (nslab == 0: y as given) -> batch statistics (mean, invstd, running
 *                                statistics as mink_bn_stats) -> out = [relu](bn(y) [+ residual])
 *   mink_bn_small_bwd            mink_bn_bwd in one launch; nslab > 0: the incoming gradient is the sum of `nslab` slabs at
 *                                `dy` ([nslab][n][C]) plus `addend` ([n][C], optional: a residual branch's gradient),
 *                                written to dy_sum
 * C must be a multiple of 16.  Results are deterministic; the summation order differs from the three-launch form (last-bit
 * differences), which is why the module-by-module path, the yardstick of the bitwise tests, never takes these. */
int32_t mink_bn_small_rows(void);
int mink_bn_set_small(int32_t on);
int mink_conv_gather_gemm_slabs(const float *x, int64_t n_in, int32_t ldx, int32_t cin, const float *w, int32_t w_transposed,
                                int32_t flip_k, const int32_t *nbr, int64_t n_out, int32_t K, const int32_t *row_perm,
                                int64_t n_virtual, float *y, int32_t ldy, int32_t cout, int32_t ksplit, float *workspace,
                                int64_t workspace_bytes, int32_t *slabs_out, void *stream);
int mink_bn_small_fwd(const float *slabs, int32_t nslab, int64_t n, int32_t C, float *y, float eps, float momentum,
                      const float *gamma, const float *beta, const float *residual, int32_t relu, float *out, float *mean,
                      float *invstd, float *running_mean, float *running_var, void *stream);
int mink_bn_small_bwd(const float *dy, int32_t nslab, const float *addend, float *dy_sum, const float *x, const float *y, int64_t n, int32_t C,
                      const float *mean, const float *invstd, const float *gamma, int32_t relu, float *dx, float *dresidual,
                      float *dgamma, float *dbeta, void *stream);

/* ------------------------------------------------------------------ the whole trunk as one call
 * Stem + every BasicBlock of a Mink-ResNet (the reference's ResNetBase.forward up to layer4:
 * co3d_3d/src/models/mink/resnet.py:107-161 `_make_layer`, :163-175 `forward`) in ONE host call per direction.
 * mink_net_forward / mink_net_backward sequence exactly what mink_stem_* and mink_block_* sequence (they call them),
 * block after block; what they add is the binding of every block to this batch -- rows, neighbour tables, activation
 * and gradient addresses -- from a handful of per-LEVEL records, so the host's per-step work no longer grows with the
 * depth of the network (Mink-ResNet34 at four scenes per GPU, BASELINE config #3's per-GPU shape: 3.6 ms of host per
 * 4.7 ms step with one call per block).
 *
 * Levels: level 0 is the stem's output (tensor stride 2), level l + 1 the map a stride-2 block takes level l to.
 * A block with conv1.stride == 2 goes from its level to the next and must carry a shortcut convolution (kernel
 * volume 1); a block with stride 1 stays on its level with an identity shortcut.
 *
 * The caller fills, once: every block's static fields (w, dw, K, cin, cout, stride of the three convolutions; gamma, beta,
 * running statistics, dgamma, dbeta, momentum, eps of the three norms) and the stem's; per step: the level records, the
 * stem's x / nbr / nbr_pool / in2out / n / n_pool / y / out / norm.mean / norm.invstd (the stem's activations stay
 * caller-owned: their layout depends on the storage type).  The library fills everything else in `blocks`.
 *
 * Activation arena (forward writes, backward reads; mink_net_sizes gives the size): per block, 256-byte aligned,
 * [y1 | h1 | y2 | out | (yd | sd)] [n_out][C] each, then six [C] statistics vectors.  The trunk's output is the last
 * block's `out` (net->out, net->out_rows).  Gradient arena (backward scratch): per block its mink_block_grad_scratch
 * floats and the gradient of its input.
 *
 * done_events (backward, optional; n_blocks entries from mink_event_create, NULL entries skipped): event i is recorded on
 * ex->wgrad once EVERY parameter gradient of block i (weights on `wgrad`, batch-norm scale / shift on `compute` /
 * `branch`) has been queued and ordered before it -- what a data-parallel caller makes a bucket's all-reduce wait for,
 * instead of for the tail of the streams (the whole backward pass is queued by the time this call returns).
 *
 * mink_set_block_done_hook (backward, data parallelism): called ON THE HOST, inside mink_net_backward, at the very point
 * where done_events[i] would be recorded -- block i's kernels are queued and ex->wgrad is ordered after them -- so that the
 * caller can issue that block's collective from ex->wgrad right there (what is queued on that stream so far IS what the
 * collective must wait for): no event, no stream of its own for the launches.  (A fifth busy hardware queue is what the
 * "eight hardware queue cliff" of DESIGN section 6 turned out to be.)  The hook must not call back into mink_net_*.  NULL = off.
 *
 * mink_set_stage_hook: instrumentation (race tests, timelines): called on the host before the stem (stage -1) and before
 * every block (stage i) of forward (backward = 0) and backward (= 1; there the stem comes last).  NULL = off. */
typedef struct {
  int64_t n;               /* rows of this level */
  const int32_t *nbr3;     /* [n][27] 3^3 stride-1 table of the level onto itself */
  const int32_t *down3;    /* [n][27] 3^3 stride-2 table from the finer level into this one (NULL at level 0) */
  const int32_t *down1;    /* [n][1] the shortcut's table (kernel volume 1, stride 2) */
  const int32_t *down3_t;  /* backward: [n_finer][27] transposed `down3` */
  const int32_t *perm;     /* backward: parity-class row order of the finer level (mink_class_partition), or NULL */
  int64_t n_perm;
} MinkLevelMaps;

typedef struct {
  MinkStem stem;
  MinkBasicBlock *blocks;
  int32_t n_blocks;
  int32_t with_stem;       /* 0: the caller runs mink_stem_forward / _backward itself (stem.out / stem.n_pool still describe level 0) */
  float *out;              /* written by mink_net_forward: [out_rows][C of the last block] */
  int64_t out_rows;
  const float *g_stem_out; /* written by mink_net_backward: gradient of the stem's output (in the gradient arena) */
} MinkNet;

typedef void (*MinkStageHook)(int32_t stage, int32_t backward);
int mink_set_stage_hook(MinkStageHook hook);
typedef void (*MinkBlockDoneHook)(int32_t block);
int mink_set_block_done_hook(MinkBlockDoneHook hook);
int mink_event_create(void **event_out);
int mink_event_destroy(void *event);
int mink_stream_wait_event(void *stream, void *event);
int mink_net_sizes(const MinkNet *net, const MinkLevelMaps *levels, int32_t n_levels, int64_t *act_floats, int64_t *grad_floats,
                   int64_t *ws_bytes);
int mink_net_forward(MinkNet *net, const MinkLevelMaps *levels, int32_t n_levels, float *arena, int64_t arena_floats,
                     const MinkExec *ex);
int mink_net_backward(MinkNet *net, const MinkLevelMaps *levels, int32_t n_levels, float *arena, int64_t arena_floats,
                      const float *g_out, float *grad_arena, int64_t grad_floats, const MinkExec *ex, void *const *done_events);

/* ------------------------------------------------------------------ bf16 storage of the full-resolution stage
 * BASELINE config "bf16 mixed precision", storage form: the network input and the stem convolution's output -- three
 * quarters of the activation bytes of a Mink-ResNet step -- are kept in HBM as bf16; the pooled level and everything
 * below stay fp32.  Arithmetic is that of mink_conv_set_math(1) (bf16 operands, fp32 accumulation); the batch-norm
 * statistics are those of the STORED (rounded) convolution output.
 *   mink_rows_to_bf16     x fp32 [n][c] (pitch ldx) -> xb bf16 [n][32], zero-padded (c <= 32)
 *   mink_stem_conv_bf16s  yb bf16 [n_out][64] = conv over nbr [n_out][27] of xb with w fp32 [27][cin][64];
 *                         stats_out: double [stats_rows][2][64] column (sum, sum of squares) partials for
 *                         mink_bn_stats_from_partials; stats_rows must equal mink_stem_conv_bf16s_stats_rows()
 *                         (one row per workgroup of the persistent launch)
 *   mink_bn_relu_pool_fwd_b16 / _bwd_b16: mink_bn_relu_pool_fwd / _bwd (parameter gradients only) reading that bf16 output
 *   mink_conv_wgrad_bn_relu_pool_b16: mink_conv_wgrad_bn_relu_pool with x = xb (pitch 32) and y bf16 */
int mink_rows_to_bf16(const float *x, int64_t n, int32_t c, int32_t ldx, void *xb, void *stream);
int mink_stem_conv_bf16s_supported(int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout);
int32_t mink_stem_conv_bf16s_stats_rows(void);
int mink_stem_conv_bf16s(const void *xb, int64_t n_in, const float *w, int32_t cin, const int32_t *nbr, int64_t n_out, int32_t K,
                         void *yb, int32_t cout, double *stats_out, int32_t stats_rows, void *stream);
int mink_bn_relu_pool_fwd_b16(const void *xb, int32_t C, const float *mean, const float *invstd, const float *gamma,
                              const float *beta, const int32_t *nbr, int64_t n_out, int32_t K, float *y, void *stream);
int mink_bn_relu_pool_bwd_b16(const float *dy_pool, const void *xb, int64_t n, int32_t C, const float *mean, const float *invstd,
                              const float *gamma, const float *beta, const int32_t *in2out, float *dgamma, float *dbeta,
                              void *workspace, int64_t workspace_bytes, void *stream);
int mink_conv_wgrad_bn_relu_pool_b16(const void *xb, int64_t n_in, int32_t cin, const void *yb, int32_t cout, const float *dy_pool,
                                     int64_t n_pool, const int32_t *in2out, const float *mean, const float *invstd,
                                     const float *gamma, const float *beta, const float *dgamma, const float *dbeta,
                                     const int32_t *nbr, int64_t n_out, int32_t K, float *dw, void *workspace,
                                     int64_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------ streams confined to a CU subset
 * A HIP stream whose kernels may only run on compute units [first_cu, first_cu + n_cus) of the device's
 * `total_cus` (hipExtStreamCreateWithCUMask; on a multi-XCD part consecutive mask bits go round the XCDs, so a
 * contiguous range is spread evenly over them).  The training step keeps several streams busy at once -- the
 * data-gradient chain, the weight gradients beside it, the next batch's coordinate maps (ME builds those on the
 * stream it computes on: models/mink/resnet.py:164 -> CoordinateMapManager) -- and an auxiliary stream that may
 * take every CU slows the kernels of the chain it runs beside; confined to a few CUs it still finishes in the
 * shadow of the chain.  The caller owns the stream. */
int mink_stream_create_cu_subset(int32_t first_cu, int32_t n_cus, int32_t total_cus, void **stream_out);
int mink_stream_destroy(void *stream);

/* ------------------------------------------------------------------ kernel timing (measurement only)
 * bench.py reports the roofline of the dominant convolution kernel from HIP events recorded on the stream each
 * kernel is launched on.  mode 0: off; 1: every convolution launch; 2: only launches matching (kind, K, cin, cout).
 * kind: 0 forward, 1 data gradient, 2 weight gradient, -1 (mode 2) any of them -- the passes of one layer.  mink_conv_timing_fetch synchronises the recorded events,
 * writes up to `max` entries and clears the list; returns the number written (or the number pending if out == NULL). */
typedef struct {
  int32_t kind, K, cin, cout;
  int64_t n_in, n_out;
  const int32_t *nbr;
  float ms;
} MinkTimingEntry;
int mink_conv_timing(int32_t mode, int32_t kind, int32_t K, int32_t cin, int32_t cout);
int64_t mink_conv_timing_fetch(MinkTimingEntry *out, int64_t max);

#ifdef __cplusplus
}
#endif
#endif /* MINK_HIP_H */
