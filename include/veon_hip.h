/*
 * veon_hip.h -- C ABI of libveon_hip.so: the MI355X (gfx950) drop-in for the
 * native half of VEON's Lift-Splat hot path.
 *
 * Conventions (mirroring the reference launchers, which are already plain C
 * functions over raw pointers -- mmdet3d/ops/bev_pool_v2/src/bev_pool.cpp:7-14):
 *   - every pointer is a DEVICE pointer owned by the caller; the library never
 *     allocates, frees or retains memory across calls (workspaces are passed in);
 *   - every launch goes to the `stream` argument (a hipStream_t passed as
 *     void*; NULL = the null stream), is asynchronous and is hipGraph-capturable;
 *   - every entry point returns VEON_OK (0) or a VEON_ERR_* code; nothing is
 *     printed.  The reference returns void and checks nothing;
 *   - index ARRAYS are int32 (the reference ABI); offsets derived from them are
 *     computed in 64 bits (the reference overflows int32 at batch >= 14 with
 *     C = 256: bev_pool_cuda.cu:41,46).
 * Paths below are relative to the reference tree.
 */
#ifndef VEON_HIP_H_
#define VEON_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: veon_vit_block_weights grew the trailing field q_log2 (round 3); every entry point
 * that existed under version 1 keeps its signature and meaning. */
#define VEON_ABI_VERSION 2

#define VEON_OK 0
#define VEON_ERR_BAD_ARG 1    /* null pointer, negative size, unsupported shape */
#define VEON_ERR_LAUNCH 2     /* hipGetLastError() != hipSuccess after launch   */
#define VEON_ERR_WORKSPACE 3  /* workspace too small                            */

/* output layouts of the fused forward */
#define VEON_LAYOUT_BZYXC 0 /* (B,Z,Y,X,C): QuickCumsumCuda's `out`, bev_pool.py:27 */
#define VEON_LAYOUT_BCZYX 1 /* (B,C,Z,Y,X): bev_pool_v2()'s return, bev_pool.py:91  */

/* storage type of the feature rows for the *_ex entry points */
#define VEON_FEAT_F32 0
#define VEON_FEAT_F16 1  /* IEEE half */
#define VEON_FEAT_BF16 2

int veon_abi_version(void);
/* 16-bit operand type this build of the library was compiled for: 0 = bf16
 * (libveon_hip.so), 1 = IEEE fp16 (libveon_hip_f16.so: the same sources with
 * -DVEON_HALF_FP16).  Every entry point named *_bf16 and every "bf16" buffer in
 * this header means "the half type of the build"; accumulation is fp32 in both. */
int veon_half_mode(void);
const char *veon_status_string(int status);

/*
 * Drop-in for `void bev_pool_v2(int c, int n_intervals, ...)`
 * (mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:125-131, declared
 * bev_pool.cpp:7-9).  Same argument order and meaning; `out` is
 * [n_voxels][c], must be zero-filled by the caller (bev_pool.py:27) and only
 * rows named by ranks_bev[interval_starts[i]] are written.  Any interval
 * order is accepted.  Per (interval, channel) the sum is a serial fmaf chain
 * in storage order, as the reference kernel's (:38-43).
 */
int veon_bev_pool_v2_fwd(int c, int n_intervals, const float *depth,
                         const float *feat, const int *ranks_depth,
                         const int *ranks_feat, const int *ranks_bev,
                         const int *interval_starts,
                         const int *interval_lengths, float *out, void *stream);

/*
 * Drop-in for `void bev_pool_v2_grad(int c, int n_intervals, ...)`
 * (bev_pool_cuda.cu:133-140, declared bev_pool.cpp:11-14).  Intervals are over
 * the ranks_feat-sorted point list (bev_pool.py:47-57).  depth_grad / feat_grad
 * must be zero-filled by the caller (bev_pool.py:67-68); out_grad is
 * [n_voxels][c].
 */
int veon_bev_pool_v2_bwd(int c, int n_intervals, const float *out_grad,
                         const float *depth, const float *feat,
                         const int *ranks_depth, const int *ranks_feat,
                         const int *ranks_bev, const int *interval_starts,
                         const int *interval_lengths, float *depth_grad,
                         float *feat_grad, void *stream);

/*
 * Pool plan for the fused forward.  The fused kernels walk the output in tiles
 * of veon_bev_pool_tile_voxels() consecutive voxel ranks; plan entry t is
 * {first interval, #intervals, first point, #points} of tile t.  `plan` must
 * hold veon_bev_pool_plan_ints(batch, voxels_per_batch) int32 values, 16-byte
 * aligned; the first 4*n_tiles are the entries, the rest is build scratch.
 * Build it once beside the five rank arrays when they are cached (the
 * accelerate=True path, view_transformer_raw.py:196-215), or per call
 * (two tiny kernels).  `counts` (optional DEVICE int[2] = {points, intervals})
 * overrides n_points / n_intervals, which then only bound the launch.
 * Requires the intervals ascending and unique in
 * ranks_bev[interval_starts[i]] -- what voxel_pooling_prepare_v2 produces
 * (view_transformer_raw.py:287-299).
 */
int veon_bev_pool_tile_voxels(void);
int64_t veon_bev_pool_plan_ints(int batch, int64_t voxels_per_batch);
int veon_bev_pool_plan(int n_intervals, int n_points, int batch,
                       int64_t voxels_per_batch, const int *ranks_bev,
                       const int *interval_starts, const int *counts, int *plan,
                       void *stream);

/*
 * Fused forward: zero-fill + pool (+ layout permute) in ONE pass; every
 * element of `out` is written exactly once, so `out` may be uninitialised.
 * Replaces the three full-volume passes of the reference
 * (new_zeros bev_pool.py:27, kernel store bev_pool_cuda.cu:46-47,
 * permute().contiguous() bev_pool.py:91).  Same summation as
 * veon_bev_pool_v2_fwd (serial fmaf chain per interval), so the results are
 * bit-identical to the three-pass structure.
 * Preconditions: as veon_bev_pool_plan; all ranks < batch * voxels_per_batch.
 */
int veon_bev_pool_v2_fwd_fused(int c, int n_intervals, int batch,
                               int64_t voxels_per_batch, const float *depth,
                               const float *feat, const int *ranks_depth,
                               const int *ranks_feat, const int *ranks_bev,
                               const int *interval_starts,
                               const int *interval_lengths, const int *plan,
                               float *out, int out_layout, void *stream);

/*
 * Fused forward + max-pool: LSSViewTransformerRaw.forward's
 * `bev_pool_v2` followed by the (dz,dy,dx) block max
 * (view_transformer_raw.py:545-553) in one kernel; the full-resolution volume
 * is never written.  out is (B, C, Z/dz, Y/dy, X/dx) fp32, every element
 * written.  Bit-equal to max-pooling veon_bev_pool_v2_fwd_fused's output.
 * row_first: veon_bev_pool_row_table(..., row_voxels = X, ...) -- first
 * interval of every (b,z,y) row of X voxels, B*Z*Y+1 entries (row_point, same
 * size, receives the first point of every row).  `counts` (optional DEVICE
 * int[2] = {points, intervals}, e.g. veon_lss_prepare's) overrides
 * n_points / n_intervals, which then only bound the launch (pass the buffer
 * capacity).  Same preconditions as veon_bev_pool_plan.
 */
int veon_bev_pool_row_table(int n_intervals, int n_points, int batch,
                            int64_t voxels_per_batch, int row_voxels,
                            const int *ranks_bev, const int *interval_starts,
                            const int *counts, int *row_first, int *row_point,
                            void *stream);
int veon_bev_pool_v2_fwd_maxpool(int c, int n_intervals, int batch, int Z, int Y,
                                 int X, int dz, int dy, int dx,
                                 const float *depth, const float *feat,
                                 const int *ranks_depth, const int *ranks_feat,
                                 const int *ranks_bev,
                                 const int *interval_starts,
                                 const int *interval_lengths,
                                 const int *row_first, float *out,
                                 void *stream);

/*
 * ---- "row" kernels: the wide-channel shapes (VEON: C = 256) ----------------
 * Same arithmetic as the fused entry points above (serial fmaf chain per voxel
 * in storage order, bev_pool_cuda.cu:38-43), different decomposition: the index
 * side is ONE dense table
 *     vstart[B*voxels_per_batch + 1]:  voxel v owns points
 *     [vstart[v], vstart[v+1]) of the rank-sorted arrays
 * (the exclusive scan of the voxel histogram -- veon_lss_prepare emits it as
 * `vstart`; veon_bev_pool_voxel_table builds it from the reference's arrays
 * ranks_bev / interval_starts, which must be ascending in voxel rank and tile
 * the point arrays, i.e. starts[i+1] == starts[i] + lengths[i] -- what
 * voxel_pooling_prepare_v2 produces, view_transformer_raw.py:287-299); the
 * gather side reads every feature row once per workgroup as one full-width wave
 * load (lanes = channels).  ranks_bev / interval_* are not read by the kernels.
 *
 * veon_bev_pool_v2_fwd_rows: zero-fill + pool + (B,C,Z,Y,X) layout in one pass
 *   (as veon_bev_pool_v2_fwd_fused_strided; plane_stride 0 = contiguous).
 *   c even.  `variant` selects the tile shape (0 = default).
 * veon_bev_pool_v2_fwd_rows_maxpool: pool + (2,2,2) block max
 *   (LSSViewTransformerRaw.forward, view_transformer_raw.py:545-553).
 *   out_padded_bf16 = 0: out is (B,C,Z/2,Y/2,X/2) fp32;  1: out is the interior
 *   of the Conv3d body's zero-padded channels-last bf16 grid (as
 *   veon_bev_pool_v2_fwd_maxpool_padded).  c % 4 == 0.
 * feat_elems: number of elements of `feat` (rows * c); must be < 2^31 (row offsets
 *   are 32-bit inside the kernels; every ranks_feat value must be a valid row).
 */
/*
 * 2x2x2 block max of a (B,C,Z,Y,X) fp32 volume -> (B,C,Z/2,Y/2,X/2) fp32: the
 * ds_feat step of LSSViewTransformerRaw.forward (view_transformer_raw.py:549-553:
 * rearrange + max over the three block axes) as one streaming pass, for volumes that
 * did not come out of the fused pool + max-pool kernel (the camera-sharded path: the
 * block max follows the cross-rank sum).  planes = B * C; Z, Y, X even.  NaN
 * propagates as in torch.amax.
 */
int veon_volume_maxpool2_f32(const float *in, float *out, int64_t planes, int Z, int Y,
                             int X, void *stream);

/* experiment knob of tools/poolbench.py / tools/xcd_order_ab.py; 0 = production.
 * bits 0-3: ablations of the row max-pool kernel (results invalid); bit 4: tile =
 * blockIdx instead of the XCD-grouped tile order; bits 8-11: (lg + 1) forces runs of
 * 2^lg tiles per XCD -- the order never changes a result. */
void veon_pool_debug_set(int flags);
/* tuning knobs of the row max-pool kernel (0 = built-in default): worker
 * workgroups, longest cold list, longest warm list */
void veon_pool_tune_set(int workers, int cold_max, int warm_max);
int64_t veon_bev_pool_voxel_table_ints(int batch, int64_t voxels_per_batch);
int veon_bev_pool_voxel_table(int n_intervals, int n_points, int batch,
                              int64_t voxels_per_batch, const int *ranks_bev,
                              const int *interval_starts, const int *counts,
                              int *vstart, void *stream);
int veon_bev_pool_v2_fwd_rows(int c, int batch, int64_t voxels_per_batch,
                              const float *depth, const void *feat, int feat_dtype,
                              const int *ranks_depth, const int *ranks_feat,
                              const int *vstart, float *out, int64_t plane_stride,
                              int64_t feat_elems, int variant, void *stream);
int veon_bev_pool_v2_fwd_rows_maxpool(int c, int batch, int Z, int Y, int X, int dz,
                                      int dy, int dx, const float *depth,
                                      const void *feat, int feat_dtype,
                                      const int *ranks_depth, const int *ranks_feat,
                                      const int *vstart, void *out, int out_padded_bf16,
                                      int64_t feat_elems, void *stream);
/*
 * The same with a caller-given order of the short-list ("cold") work: the pooled
 * volume is cut into chunks of veon_bev_pool_rows_maxpool_chunk() consecutive pooled
 * voxels per batch element (linear (zo, yo, xo) order, chunk id = b * chunks_per_batch
 * + index), cold workgroup i processes chunk chunk_order[i]; chunk_order must be a
 * permutation of all B * ceil(plane / chunk) chunk ids (NULL = built-in order).  Pure
 * scheduling: results never depend on it.  Workgroups are dealt to the eight XCDs
 * round-robin, so an order that gives every XCD the chunks of one azimuth sector around
 * the rig keeps the feature rows of that sector's cameras in the XCD's own 4 MiB L2
 * (veon_amd/ops/bev_pool_v2/bev_pool.py `cold_chunk_order`).
 */
int veon_bev_pool_rows_maxpool_chunk(void);
int veon_bev_pool_v2_fwd_rows_maxpool_ordered(
    int c, int batch, int Z, int Y, int X, int dz, int dy, int dx, const float *depth,
    const void *feat, int feat_dtype, const int *ranks_depth, const int *ranks_feat,
    const int *vstart, void *out, int out_padded_bf16, int64_t feat_elems,
    const int *chunk_order, void *stream);

/*
 * Half-precision feature rows.  QuickCumsumCuda.forward widens feat to fp32
 * before the kernel (bev_pool.py:21 `feat.contiguous().float()`); the *_ex
 * entry points read fp16 / bf16 rows (`feat_dtype` = VEON_FEAT_*) and widen in
 * registers -- identical arithmetic (the widening is exact, the fmaf chain and
 * the output stay fp32) at half the gather bytes and no fp32 copy of feat.
 * With VEON_FEAT_F32 they are the functions above.
 */
int veon_bev_pool_v2_fwd_fused_ex(int c, int n_intervals, int batch,
                                  int64_t voxels_per_batch, const float *depth,
                                  const void *feat, int feat_dtype,
                                  const int *ranks_depth, const int *ranks_feat,
                                  const int *ranks_bev,
                                  const int *interval_starts,
                                  const int *interval_lengths, const int *plan,
                                  float *out, int out_layout, void *stream);
/* (images, C, HW) -> (images, HW, C), 4- or 2-byte elements: the
 * `feat.permute(0, 1, 3, 4, 2)` of view_transform_core (view_transformer.py:273-275)
 * made contiguous, as bev_pool_v2 does on entry (bev_pool.py:21). */
int veon_feat_nchw_to_nhwc(const void *in, void *out, int elem_bytes, int images,
                           int C, int HW, void *stream);
/* As veon_bev_pool_v2_fwd_fused_ex with the (B,C,Z,Y,X) layout, but the channel
 * planes of `out` are `plane_stride` floats apart (>= voxels_per_batch): out is a
 * (B, C, plane_stride) buffer whose first voxels_per_batch floats of every plane
 * are written.  For consumers that take a strided view, and for
 * tools/stride_probe.py (placement sensitivity of the plane streams). */
int veon_bev_pool_v2_fwd_fused_strided(int c, int n_intervals, int batch,
                                       int64_t voxels_per_batch, const float *depth,
                                       const void *feat, int feat_dtype,
                                       const int *ranks_depth, const int *ranks_feat,
                                       const int *ranks_bev,
                                       const int *interval_starts,
                                       const int *interval_lengths, const int *plan,
                                       float *out, int64_t plane_stride,
                                       void *stream);
int veon_bev_pool_v2_fwd_maxpool_ex(int c, int n_intervals, int batch, int Z,
                                    int Y, int X, int dz, int dy, int dx,
                                    const float *depth, const void *feat,
                                    int feat_dtype, const int *ranks_depth,
                                    const int *ranks_feat, const int *ranks_bev,
                                    const int *interval_starts,
                                    const int *interval_lengths,
                                    const int *row_first, float *out,
                                    void *stream);

/*
 * The per-camera 3x3 algebra of get_lidar_coor
 * (view_transformer_raw.py:145,151): post_rots_inv = inv(post_rots),
 * combine = sensor2ego[:3,:3] @ inv(cam2imgs), trans = sensor2ego[:3,3], for BN
 * cameras (sensor2ego (BN,4,4), the others (BN,3,3)).  Stream-capturable
 * replacement for the reference's two torch.inverse calls: adjugate inverse in
 * double precision rounded to float (agrees with LAPACK / rocSOLVER to their
 * own rounding error).
 */
int veon_camera_matrices(int BN, const float *sensor2ego, const float *cam2imgs,
                         const float *post_rots, float *post_rots_inv,
                         float *combine, float *trans, void *stream);

/*
 * Frustum -> ego coordinates: the per-point half of get_lidar_coor
 * (mmdet3d/models/necks/view_transformer_raw.py:144-155).  xs[W], ys[H], ds[D]
 * are the frustum axes (create_frustum, :91-119); post_rots_inv = inv(post_rots)
 * and combine = sensor2ego[:3,:3] @ inv(cam2imgs) are (B,N,3,3), post_trans /
 * trans (B,N,3), bda (B,3,3).  coor is (B,N,D,H,W,3).  Bit-identical to the
 * reference's CPU result for identical matrices.
 */
int veon_lidar_coor(int B, int N, int D, int H, int W, const float *xs,
                    const float *ys, const float *ds,
                    const float *post_rots_inv, const float *post_trans,
                    const float *combine, const float *trans, const float *bda,
                    float *coor, void *stream);

/*
 * voxel_pooling_prepare_v2 (view_transformer_raw.py:244-302) as a counting
 * sort on the device, no host synchronisation inside.
 *   coor != NULL : voxelise the given (B,N,D,H,W,3) coordinates (the method's
 *                  own contract);
 *   coor == NULL : fuse the geometry of veon_lidar_coor in, never
 *                  materialising the coordinates.
 * grid_lower / grid_interval / grid_size are HOST float[3] (the reference's
 * float32 grid tensors, :74-89).  Outputs must hold B*N*D*H*W int32 each; the
 * first counts[0] (points kept) / counts[1] (intervals) entries are valid,
 * counts is DEVICE int[2].  Order inside an interval is ascending ranks_depth
 * (the reference's unstable argsort leaves it unspecified).  `plan` (optional,
 * veon_bev_pool_plan_ints(B, voxels_per_batch) int32, 16-B aligned; needs
 * voxels_per_batch % 64 == 0) receives the fused pool kernels' plan for free.
 * workspace: veon_lss_prepare_workspace_bytes(B*N*D*H*W, B*voxels_per_batch).
 */
int64_t veon_lss_prepare_workspace_bytes(int64_t num_points,
                                         int64_t num_voxels_total);
/*
 * The same prepare straight from the reference's per-camera tensors
 * (get_lidar_coor's arguments, view_transformer_raw.py:121-158: sensor2ego
 * (B,N,4,4), cam2imgs / post_rots (B,N,3,3), post_trans (B,N,3), bda (B,3,3)): the
 * camera algebra of veon_camera_matrices runs inside the first kernel (one launch
 * less, same arithmetic), five launches in all.  Extras for a steady-state,
 * hipGraph-captured caller:
 *   vstart (optional, B*voxels_per_batch + 1 int32): the dense voxel table of the
 *     row pool kernels (veon_bev_pool_voxel_table's output) -- it is the counting
 *     sort's own scan, so it costs nothing;
 *   hist_is_zero = 1: the caller zeroed the WHOLE workspace once (at allocation)
 *     and has only used it for completed calls of this function since: every
 *     call leaves the histogram zeroed again, so no memset node is needed.
 *     With 0 the histogram is cleared first (hipMemsetAsync).
 */
/*
 * AlignNetOcc3D.prepare_meta (align_net_occ3d.py:328-352) for one frame: out[b,n] =
 * inverse(ego2global[b,0]) @ ego2global[b,n] @ sensor2ego[b,n], (B,N,4,4) fp32 in and
 * out, the algebra in double precision (the reference casts to double, too).
 */
int veon_sensor2keyego(int B, int N, const float *sensor2ego, const float *ego2global,
                       float *out, void *stream);

int veon_lss_prepare_cameras(int B, int N, int D, int H, int W, const float *xs,
                             const float *ys, const float *ds,
                             const float *sensor2ego, const float *cam2imgs,
                             const float *post_rots, const float *post_trans,
                             const float *bda, const float *grid_lower,
                             const float *grid_interval, const float *grid_size,
                             int64_t voxels_per_batch, void *workspace,
                             int64_t workspace_bytes, int hist_is_zero,
                             int *ranks_bev, int *ranks_depth, int *ranks_feat,
                             int *interval_starts, int *interval_lengths, int *plan,
                             int *vstart, int *counts, void *stream);
/*
 * Sparse lift (opt-in): the same prepare, but a frustum point whose
 * depth weight depth_weights[(b,n,d,h,w)] (the (B,N,D,H,W) tensor the pool will
 * multiply by) is below depth_eps is dropped before the sort.  VEON's depth is a soft
 * two-hot distribution (view_transformer_raw.py:406-429: softmax of -4|d - c_k|
 * clamped at -16), so all but a handful of the D bins of a pixel carry ~1e-7 of the
 * mass: with depth_eps = 1e-6 the sort, the rank pass and the pool see ~20x fewer
 * points and every pooled sum moves by at most depth_eps * sum|feat| of the dropped
 * points (rtol ~1e-5, the reference's own fp32 reassociation noise).
 * depth_eps = 0 keeps every point (weights are >= 0).
 */
int veon_lss_prepare_cameras_sparse(
    int B, int N, int D, int H, int W, const float *xs, const float *ys, const float *ds,
    const float *sensor2ego, const float *cam2imgs, const float *post_rots,
    const float *post_trans, const float *bda, const float *grid_lower,
    const float *grid_interval, const float *grid_size, int64_t voxels_per_batch,
    void *workspace, int64_t workspace_bytes, int hist_is_zero, int *ranks_bev,
    int *ranks_depth, int *ranks_feat, int *interval_starts, int *interval_lengths,
    int *plan, int *vstart, int *counts, const float *depth_weights, float depth_eps,
    void *stream);
/*
 * Two-hot lift by construction (view_transformer_raw.py:406-429 +
 * 244-302 in one): the same prepare driven by veon_two_hot_window's per-pixel windows
 * instead of a (B,N,D,H,W) weight tensor.  Point (pixel, bin k) is kept iff k lies in
 * the pixel's kept window [q0, q0+nq), or the pixel's tail weight passed the threshold
 * and k lies outside its unclamped window [k0, k0+nk).  ranks_depth then indexes the
 * COMPACT weight table: pix*window_slots + (k in [k0,k0+nk) ? 1 + k - k0 : 0), so every
 * pool entry point of this header is called with depth = wts and computes the sums of
 * the dense lift over the kept points, in the dense lift's order (ascending point
 * index inside a voxel).  With eps = 0 in veon_two_hot_window that is the dense lift
 * to the bit; with eps > 0 every pooled sum moves by at most eps * sum|feat| over the
 * dropped points of its voxel.  ranks_feat / ranks_bev / intervals / vstart / plan /
 * counts as veon_lss_prepare_cameras.
 */
int veon_lss_prepare_cameras_twohot(
    int B, int N, int D, int H, int W, const float *xs, const float *ys, const float *ds,
    const float *sensor2ego, const float *cam2imgs, const float *post_rots,
    const float *post_trans, const float *bda, const float *grid_lower,
    const float *grid_interval, const float *grid_size, int64_t voxels_per_batch,
    void *workspace, int64_t workspace_bytes, int hist_is_zero, int *ranks_bev,
    int *ranks_depth, int *ranks_feat, int *interval_starts, int *interval_lengths,
    int *plan, int *vstart, int *counts, const int *win, int window_slots, void *stream);

int veon_lss_prepare(int B, int N, int D, int H, int W, const float *coor,
                     const float *xs, const float *ys, const float *ds,
                     const float *post_rots_inv, const float *post_trans,
                     const float *combine, const float *trans, const float *bda,
                     const float *grid_lower, const float *grid_interval,
                     const float *grid_size, int64_t voxels_per_batch,
                     void *workspace, int64_t workspace_bytes, int *ranks_bev,
                     int *ranks_depth, int *ranks_feat, int *interval_starts,
                     int *interval_lengths, int *plan, int *counts,
                     void *stream);

/*
 * Depth preparation (LSSViewTransformerRaw.downsample_depth /
 * get_two_hot_depth, view_transformer_raw.py:393-429).
 * veon_downsample_depth: depths (BN,H,W) -> out (BN,H/ds,W/ds), min over the
 *   non-zero pixels of each block (zeros count as 1e5).
 * veon_two_hot_depth: out (BN,D,H,W) = softmax over the D+1 bin centres
 *   c_k = k*step + (lo + step/2) of -gamma*|d - c_k| clamped at -16, last bin
 *   dropped.  ds == 0: depths is (BN,H,W); ds > 0: depths is (BN,H*ds,W*ds) and
 *   the block-min is fused in (the two calls of AlignNetOcc3D.prepare_depth,
 *   align_net_occ3d.py:320-326, in one pass).
 */
int veon_downsample_depth(int BN, int H, int W, int ds, const float *depths,
                          float *out, void *stream);
int veon_two_hot_depth(int BN, int H, int W, int ds, int D, float lo, float step,
                       float gamma, const float *depths, float *out,
                       void *stream);
/*
 * The same distribution in its compact, EXACT form (the
 * (BN,D,H,W) tensor is never written).  A pixel's D+1 logits are -gamma*|d - c_k|
 * where that is >= -16 and exactly -16 elsewhere (view_transformer_raw.py:419-421), so
 * its weights take distinct values only on the contiguous window of unclamped bins
 * [k0, k0+nk) and ONE value -- the tail exp(-16 - max)/sum -- on every other bin.
 *   veon_two_hot_window_slots(D, step, gamma) -> K = 1 + max window length
 *     (floor(32/(gamma*step)) + 2, capped at D); 0 on bad arguments.
 *   wts[pix*K + 0] = tail, wts[pix*K + 1 + j] = weight of bin k0 + j (0 beyond nk);
 *     bit-identical to veon_two_hot_depth's values of those bins.
 *   win[2*pix]     = k0 | nk << 16            unclamped window (bins < D)
 *   win[2*pix + 1] = q0 | nq << 16 | t << 31  [q0, q0+nq): the window bins whose
 *     weight is >= eps (contiguous: the weights are unimodal); t = tail >= eps.
 * depths / ds as veon_two_hot_depth (ds > 0 fuses the block-min).  pix runs over
 * (BN, H, W).  eps = 0 keeps every bin.  win must be 8-byte aligned.
 * Consumed by veon_lss_prepare_cameras_twohot; the pool kernels then take `wts` as
 * their depth table.
 */
int veon_two_hot_window_slots(int D, float step, float gamma);
int veon_two_hot_window(int BN, int H, int W, int ds, int D, float lo, float step,
                        float gamma, float eps, int K, const float *depths, int *win,
                        float *wts, void *stream);

/*
 * ViT encoder block kernels (bf16 operands on MFMA, fp32 accumulate, fp32
 * residual stream): the dense contractions of the DINOv2 blocks of
 * DepthAnythingV2 (mmdet3d/models/depth_anything/dinov2_layers/block.py:85-110,
 * attention.py:56-69, mlp.py:40-46, layer_scale.py:27) and of the CLIP
 * residual-attention blocks (semantic_net/clip_utils/visual.py:57-91,
 * attn_helper.py:303-314).  bf16 buffers are passed as void* (uint16 storage).
 *
 * veon_vit_cast_bf16 : fp32 -> bf16 (round to nearest even), n elements.
 * veon_vit_layernorm : x fp32 [T,d] -> bf16 [T,d]; nn.LayerNorm semantics
 *                      (biased variance, eps inside the sqrt), d <= 2048.
 * veon_vit_gemm      : C[M,N] = A[M,K] . W[N,K]^T (+ bias[N]); W is the
 *                      nn.Linear weight layout.  K % 64 == 0, N % 4 == 0.
 *                      epilogue 0: -> bf16 out;  1: GELU(erf) -> bf16 out;
 *                      2: QuickGELU x*sigmoid(1.702x) -> bf16 out;
 *                      3: resid[M,N] (fp32, in place) += gamma[N] * C
 *                         (gamma NULL = 1): LayerScale + residual add.
 *                      4: A.W^T * gamma[N] + bias[N] -> bf16 out (a 1x1x1
 *                         convolution + eval-mode BatchNorm);  5: the same + ReLU
 *                         (the ConvModules of PredHead3DOcc / PredHead3DSem,
 *                         align_net_occ3d.py:431-534);  6: sigmoid(4) - 0.5 (the
 *                         output activation of PredHead3DSem, :528-533).
 * veon_vit_attention : qkv bf16 [B,T,3,H,64] (q pre-scaled) -> out bf16
 *                      [B,T,H*64] = softmax(q k^T + bias) v, flash style.
 *                      bias (optional) fp32 [.,.,T,T] with batch / head strides
 *                      in elements (0 = broadcast).  head_dim must be 64.
 */
/* experiment knob of tools/gemm_bench.py: force the tile configuration of
 * veon_vit_gemm (-1 = automatic, 0 = small-tile kernel, 1..6 = ring-kernel tiles) */
void veon_gemm_ring_set(int config);
int veon_vit_cast_bf16(const float *in, void *out_bf16, int64_t n, void *stream);
/* Patch embedding (dinov2_layers/patch_embed.py: Conv2d(kernel = stride = patch)) as a
 * GEMM operand: img fp32 (B,C,H,W) -> bf16 rows [B*(skip + h*w)][kpad], row element
 * c*patch*patch + i*patch + j (the conv weight's order), zero beyond C*patch*patch,
 * ``skip`` zero rows in front of every image (class-token slot).  A remainder of H/W
 * modulo patch is not read (as the convolution). */
int veon_vit_patchify(const float *img, void *out_bf16, int B, int C, int H, int W,
                      int patch, int skip, int kpad, void *stream);
int veon_vit_layernorm(const float *x, const float *gamma, const float *beta,
                       void *out_bf16, int T, int d, float eps, void *stream);
int veon_vit_gemm(const void *a_bf16, const void *w_bf16, const float *bias,
                  const float *gamma, float *resid, void *out_bf16, int M, int N,
                  int K, int epilogue, void *stream);
int veon_vit_attention(const void *qkv_bf16, const float *bias,
                       int64_t bias_batch_stride, int64_t bias_head_stride,
                       void *out_bf16, int B, int T, int H, int head_dim,
                       void *stream);
/* The same with q ALSO multiplied by log2(e) (folded into the projection weights next
 * to head_dim^-0.5): the scores arrive in the exp2 domain and the kernel spends no
 * instruction on scaling them (the form veon_vit_block uses when
 * veon_vit_block_weights.q_log2 is set).  bias stays in natural-log units. */
int veon_vit_attention_log2(const void *qkv_bf16, const float *bias,
                            int64_t bias_batch_stride, int64_t bias_head_stride,
                            void *out_bf16, int B, int T, int H, int head_dim,
                            void *stream);

/*
 * The fused pool + max-pool writing straight into the Conv3d body's input: the
 * interior of the zero-padded channels-last bf16 grid [B][Z/dz+2][Y/dy+2][X/dx+2][C]
 * (see veon_conv3d_k3_bf16; allocate it zeroed once, the halo is never
 * written).  Values are the fp32 results of veon_bev_pool_v2_fwd_maxpool_ex
 * rounded to bf16 -- what veon_volume_pack_bf16 of that tensor would store.
 */
int veon_bev_pool_v2_fwd_maxpool_padded(int c, int n_intervals, int batch, int Z,
                                        int Y, int X, int dz, int dy, int dx,
                                        const float *depth, const void *feat,
                                        int feat_dtype, const int *ranks_depth,
                                        const int *ranks_feat,
                                        const int *ranks_bev,
                                        const int *interval_starts,
                                        const int *interval_lengths,
                                        const int *row_first,
                                        void *out_padded_bf16, void *stream);

/*
 * One whole pre-norm transformer block on the fp32 residual stream x [B*T, d]
 * (in place): x += g1*(proj(attn(LN1(x)))), x += g2*(fc2(act(fc1(LN2(x))))) --
 * the DINOv2 block (dinov2_layers/block.py:85-110; g = LayerScale) and the CLIP
 * ResidualAttentionBlock (g = NULL, act = QuickGELU).  Seven launches behind
 * one call, so an eager host pays one call per block instead of seven.
 * Weights: bf16 [out,in] row-major with q pre-scaled, fp32 vectors; gamma1/2
 * may be NULL.  act = 1 (GELU erf) or 2 (QuickGELU) -- the veon_vit_gemm
 * epilogue codes.  workspace: veon_vit_block_workspace_bytes(), 256-B aligned.
 */
typedef struct veon_vit_block_weights {
  const float *ln1_w, *ln1_b;
  const void *w_qkv;
  const float *b_qkv;
  const void *w_proj;
  const float *b_proj, *gamma1;
  const float *ln2_w, *ln2_b;
  const void *w_fc1;
  const float *b_fc1;
  const void *w_fc2;
  const float *b_fc2, *gamma2;
  float ln1_eps, ln2_eps;
  int mlp_dim, act;
  int q_log2;   /* 1: w_qkv / b_qkv's q rows carry head_dim^-0.5 * log2(e) (attention in
                   the exp2 domain, veon_vit_attention_log2); 0: head_dim^-0.5 only */
} veon_vit_block_weights;
/*
 * Split-K form of the residual GEMM (resid += gamma * (a @ w^T + bias)) for fc2-shaped
 * problems (long K, few output columns: M = 5406, N = 768 / 1024, K = 3072 / 4096): two
 * workgroups per output tile take half of K each, the first to finish parks its fp32
 * accumulators in `slab`, the second adds them to its own and runs the epilogue
 * (deterministic: a + b is commutative).  veon_vit_gemm_splitk_plan -> slab bytes needed
 * (0: the shape is not split).  sync_words (2 ints per tile): ZERO on entry, left zero.
 * veon_vit_block uses it for fc2 with its (by then free) qkv buffer as the slab; the
 * block workspace therefore has to be ZERO when first used (it ends in the sync words).
 */
int64_t veon_vit_gemm_splitk_plan(int M, int N, int K, int *tile_out);
int veon_vit_gemm_splitk(const void *a_bf16, const void *w_bf16, const float *bias,
                         const float *gamma, float *resid, int M, int N, int K,
                         void *slab, int64_t slab_bytes, int *sync_words,
                         int64_t sync_ints, void *stream);
int64_t veon_vit_block_workspace_bytes(int B, int T, int d, int mlp_dim);
int veon_vit_block(float *x, const veon_vit_block_weights *w,
                   const float *attn_bias, int64_t bias_batch_stride,
                   int64_t bias_head_stride, void *workspace,
                   int64_t workspace_bytes, int B, int T, int d, int H,
                   void *stream);
/*
 * LayerNorm of PADDED rows: x fp32 [T, ld], the token is the first d columns (the rest
 * is padding up to the multiples of 64 the GEMM kernels need); statistics over d,
 * out bf16 [T, ld] with zeros in the padding.  For transformer widths that are not a
 * multiple of 64 -- SAN's side-adapter ViT, width 240 / head_dim 40
 * (side_adaptor_in_veon.py:194-241, timm_wrapper.py:67-74): its blocks run on the
 * kernels above with the width padded to 256 and every head to 64 (zero weight rows /
 * columns, q scaled by 40^-0.5 as before).
 */
int veon_vit_layernorm_padded(const float *x, const float *gamma, const float *beta,
                              void *out_bf16, int T, int d, int ld, float eps,
                              void *stream);


/*
 * ---- 3x3x3 Conv3d body of the 3D alignment network ----
 * Replaces the Conv3d + BN3d (+ReLU, + identity) of ResBlock3D
 * (mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:363-399), which
 * the reference runs as mmcv ConvModules on cuDNN, by an implicit-GEMM MFMA
 * kernel.  Volumes are channels-last bf16 in a zero-padded grid
 * [B][Z+2][Y+2][X+2][C]; `*_padded` pointers address padded voxel (b=0,0,0,0)
 * and the caller must keep veon_conv3d_guard_rows(Y, X) rows of C readable,
 * finite bf16 values before that address and after the last padded row (the
 * kernel reads them for halo outputs, which it then writes as zeros; it never
 * writes guard rows).  w: [Cout][3][3][3][Cin] bf16 (torch's
 * weight.permute(0,2,3,4,1)).  y = conv(in)*scale[n] + shift[n] (eval-mode
 * BatchNorm folded by the caller; either may be NULL), + resid (optional,
 * padded layout, Cout channels), then `relu` = 0 nothing / 1 ReLU / 2 GELU (erf);
 * halo rows of `out` are written
 * as zeros so `out` is a valid padded input of the next conv.  Cin % 64 == 0,
 * Cout % 8 == 0.  out must not alias in.
 */
int64_t veon_conv3d_guard_rows(int Y, int X);
int veon_conv3d_k3_bf16(const void *in_padded, const void *w_bf16,
                        const float *scale, const float *shift,
                        const void *resid_padded, void *out_padded, int B, int Z,
                        int Y, int X, int Cin, int Cout, int relu, void *stream);
/*
 * The 2-D case: 3x3 stride-1 pad-1 convolution on images in the padded
 * channels-last bf16 grid [B][Y+2][X+2][C] (no z halo), same kernel with 9 taps,
 * same epilogue, same guard-row contract.  w: [Cout][3][3][Cin] bf16.  Used for
 * the 3x3 convolutions of DepthAnythingV2's DPT head
 * (mmdet3d/models/depth_anything/dpt.py:39-150, util/blocks.py) when the head
 * runs in bf16.
 */
/* experiment knob of tools/body_bench.py (ablations of the conv kernels' loads;
 * non-zero flags give WRONG results): 0 = normal. */
void veon_conv_debug_set(int flags);
/* Host-only: the (rows | cols << 16) tile the conv launcher picks for a problem
 * (kd = 3: veon_conv3d_k3_bf16 on B x Z x Y x X voxels; kd = 1: the 2-D convs on
 * B x Y x X output pixels, stride 1 or 2); -1 for an unsupported shape.  Lets the CPU
 * tests pin the selection rule (csrc/conv3d.hip: conv_pick_tile). */
int veon_conv_tile_choice(int kd, int B, int Z, int Y, int X, int Cin, int Cout, int stride);
int veon_conv2d_k3_bf16(const void *in_padded, const void *w_bf16,
                        const float *scale, const float *shift,
                        const void *resid_padded, void *out_padded, int B, int Y,
                        int X, int Cin, int Cout, int relu, void *stream);
/* The same with two optional extras of the epilogue (NULL = absent): a SECOND residual
 * image resid2 (added before the activation) and a second output out_relu that receives
 * relu(result) (not `out_padded` itself).  FeatureFusionBlock (util/blocks.py:86-148)
 * computes  x0 + RCU1(x1)  and every ResidualConvUnit starts with relu(input) (:49-83):
 * with these, conv2 of RCU1 writes x0 + conv + x1 and its ReLU in one pass, and the
 * layer*_rn convolutions write the ReLU of their output next to it -- no elementwise
 * add_ / clamp_min passes over the images. */
int veon_conv2d_k3_bf16_ex(const void *in_padded, const void *w_bf16, const float *scale,
                           const float *shift, const void *resid_padded,
                           const void *resid2_padded, void *out_padded,
                           void *out_relu_padded, int B, int Y, int X, int Cin, int Cout,
                           int relu, void *stream);
/* The same with stride 2 (Conv2d(k = 3, stride = 2, padding = 1): DPTHead.resize_layers[3],
 * depth_anything/dpt.py:69-72): in = padded image (B, Cin, Yin, Xin), out = padded image
 * (B, Cout, ceil(Yin/2), ceil(Xin/2)); only the needed output pixels are computed (the
 * activation rows of a tile are gathered by per-lane DMA addresses). */
int veon_conv2d_k3s2_bf16(const void *in_padded, const void *w_bf16, const float *scale,
                          const float *shift, const void *resid_padded, void *out_padded,
                          int B, int Yin, int Xin, int Cin, int Cout, int act,
                          void *stream);
/* (B,C,Y,X) fp32 or bf16 (nchw_is_bf16) <-> interior of the padded image grid */
int veon_image_pack_bf16(const void *nchw, int nchw_is_bf16, void *padded, int B,
                         int C, int Y, int X, void *stream);
int veon_image_unpack(const void *padded, void *nchw, int nchw_is_bf16, int B,
                      int C, int Y, int X, void *stream);
/* F.interpolate(mode='bilinear', align_corners=True) between two padded images
 * (interior written only; C % 8 == 0). */
int veon_image_resize_bilinear(const void *in_padded, void *out_padded, int B,
                               int C, int Yi, int Xi, int Yo, int Xo,
                               void *stream);
/* Tail of the occupancy path in one kernel (semantic_net/san_in_veon_temporal.py:
 * 196-211 + detectors/veon_temporal.py:219-227): sem (B,Q,zi,yi,xi) and bin
 * (B,2,zi,yi,xi) fp32 logits, each addressed through five ELEMENT strides
 * {b, c, z, y, x} (so the channels-last rows of the heads' GEMMs are read in place),
 * are upsampled trilinearly (align_corners=False) to (Zo,Yo,Xo):
 *   sem_out (B,Q,Zo,Yo,Xo), bin_out (B,2,Zo,Yo,Xo) fp32 contiguous,
 *   cls_out (B,Xo,Yo,Zo) int64 = argmax_c softmax(sem_out) where
 *     softmax(bin_out)[0] > 0.5 (and the best score > 0), else Q (= free). */
int veon_occ_classify(const float *sem, const int64_t *sem_strides, int Q,
                      const float *bin, const int64_t *bin_strides, int B, int zi,
                      int yi, int xi, int Zo, int Yo, int Xo, float *sem_out,
                      float *bin_out, int64_t *cls_out, void *stream);
/* ViT token rows -> padded image, with the pixel shuffle of a ConvTranspose2d(k = s,
 * stride = s) folded in (DPTHead.resize_layers[0:2], depth_anything/dpt.py:55-72:
 * the transposed convolution itself is a GEMM over the tokens whose output row holds
 * the s*s pixels [i][j][C] of one token).  rows: bf16 [B*tokens_per_image][row_elems]
 * (row_elems >= s*s*C); the first ``skip`` rows of every image (class token) are
 * passed over; out: padded image (B, C, s*h, s*w), interior written.  C % 8 == 0. */
int veon_tokens_to_image(const void *rows, int64_t row_elems, int tokens_per_image,
                         int skip, int h, int w, int s, int C, void *out_padded, int B,
                         void *stream);
/* out(y,x) = in(step*y, step*x) between padded images (B,C,Yi,Xi) ->
 * (B,C,ceil(Yi/step),ceil(Xi/step)): a stride-``step`` 3x3 pad-1 convolution
 * (DPTHead.resize_layers[3]) is veon_conv2d_k3_bf16 followed by this. */
int veon_image_subsample(const void *in_padded, void *out_padded, int B, int C, int Yi,
                         int Xi, int step, void *stream);
/* out (B,Y,X) fp32 = act(1x1 conv C -> 1 of a padded image + bias); C in {32, 64};
 * act 0 none / 1 ReLU / 2 sigmoid (the tail of DPTHead.output_conv2, dpt.py). */
int veon_image_dot(const void *in_padded, const float *w, float bias, float *out,
                   int B, int C, int Y, int X, int act, void *stream);
/* Physically contiguous device memory (hipExtMallocWithFlags +
 * hipDeviceMallocContiguous) for the lift's output volume, whose write pattern
 * (C planes 4*Z*Y*X bytes apart per workgroup) is sensitive to the page-table
 * fragment size; VEON_ERR_LAUNCH when the driver cannot provide it (the caller
 * then keeps an ordinary allocation).  No reference counterpart: the reference
 * lets torch.zeros allocate the volume (bev_pool.py:17-19). */
int veon_alloc_contiguous(void **ptr, int64_t bytes);
/* the same with any hipExtMallocWithFlags flag (probe tool) */
int veon_alloc_device_flags(void **ptr, int64_t bytes, unsigned flags);
int veon_free_device(void *ptr);
/* nn.LayerNorm over the last dim, fp32 in -> fp32 out, rows of d floats
 * (d % 128 == 0, d <= 1024): the token LayerNorms (ln_3 / ln_4 / pre_norm) of the
 * HSA network's blocks (highres_side_adaptor.py:108-135, 138-193). */
int veon_layernorm_f32(const float *x, const float *gamma, const float *beta,
                       float *out, int T, int d, float eps, void *stream);
/* LayerNorm(x + offset) of (B, L, d) fp32 tokens, where offset is the nearest-neighbour
 * resize (F.interpolate default mode: src = min(floor(dst * in / out), in - 1), the scale
 * in fp32) of a coarser map `add` [B][h*w][d] fp32 to the Y x X token map, added to the
 * LAST Y*X tokens of every sample: the tail of HighresSideAdaptorBlock.forward
 * (highres_side_adaptor.py:123-135 -- neck_add, interpolate, cat / add, ln_4) in one
 * pass instead of upsample + add + cat + LayerNorm. */
int veon_layernorm_f32_add_nearest(const float *x, const float *add, const float *gamma,
                                   const float *beta, float *out, int B, int L, int d,
                                   int Y, int X, int h, int w, float eps, void *stream);
/* the same LayerNorm of (B, Y*X, d) fp32 tokens, written as bf16 into the interior
 * of a zero-haloed channels-last image [B][Y+2][X+2][d] (halo untouched): ln_3
 * followed by the ConvBlock's permute / reshape to a feature map (:113-122, :38-40). */
int veon_layernorm_f32_to_padded(const float *x, const float *gamma, const float *beta,
                                 void *out_padded, int B, int Y, int X, int d,
                                 float eps, void *stream);
/* LayerNorm over the channels of every pixel of a padded channels-last bf16 image:
 * the nn.LayerNorm calls of ConvBlock.forward (highres_side_adaptor.py:31-52) with
 * their permute / reshape pairs.  out_tokens_f32 = 0: out is a padded bf16 image of
 * the same shape, halo written as zeros (input of the next 3x3 conv); 1: out is the
 * compact fp32 token tensor (B, Y*X, C), plus resid_tokens (same shape, fp32, may
 * be NULL): the `ConvBlock(ln_3(x)) + x` of the adaptor block (:122).
 * C % 8 == 0, C <= 1024. */
int veon_image_layernorm_bf16(const void *in_padded, const float *gamma,
                              const float *beta, void *out, int out_tokens_f32, int B,
                              int C, int Y, int X, float eps, const float *resid_tokens,
                              void *stream);
/* ---- temporal path, csrc/temporal.hip ---------------------
 * Sampling + attention core of TemporalDeformable.forward
 * (mmdet3d/models/semantic_net/side_adapter/align_net_occ3d.py:138-196), replacing
 * its repeat + F.grid_sample + two einsums + softmax.  All operands are padded
 * channels-last bf16 grids of one (B,Z,Y,X) shape: kv has 2*C channels laid out
 * per head as [key hd | value hd] (the reference's view of key_value_proj), q has
 * C, off has off_channels >= heads*samples*3 raw (pre-tanh) offsets ordered
 * (head, sample, axis); out (C channels) gets its interior rows written, the
 * halo is not touched.  samples must be 8; C/heads in {32, 64}. */
int veon_deform_attention_bf16(const void *kv_padded, const void *q_padded,
                               const void *off_padded, void *out_padded, int B,
                               int Z, int Y, int X, int C, int heads, int samples,
                               int off_channels, void *stream);
/* SANInVeonTemporal.align_after_lss (san_in_veon_temporal.py:325-365) on a padded
 * grid: out voxel (x,y,z) = trilinear sample of `in` at affine[b] (3x4 row-major,
 * voxel-index units) applied to (x,y,z,1); zero outside (F.grid_sample
 * padding_mode='zeros', align_corners=True). */
int veon_volume_warp_bf16(const void *in_padded, void *out_padded,
                          const float *affine, int B, int C, int Z, int Y, int X,
                          void *stream);
/* The coordinate chain of align_after_lss (san_in_veon_temporal.py:326-358) as one
 * 3x4 map in voxel-index units per sample, computed on the device in double:
 * affine[b] = S^-1 inv(prev2glob[b]) cur2glob[b] S, S = diag(step) + first voxel
 * centre.  cur2glob / prev2glob: device fp32 4x4 row-major, mat_stride floats
 * apart (>= 16); first_xyz / step_xyz: HOST doubles[3]; affine: device (B,3,4). */
int veon_warp_affine(const float *cur2glob, const float *prev2glob, int mat_stride,
                     const double *first_xyz, const double *step_xyz, float *affine,
                     int B, void *stream);
/* zero the halo rows of a padded grid (after a row-wise GEMM epilogue wrote its
 * shift there and a 3x3x3 conv is to consume it). */
int veon_volume_zero_halo_bf16(void *padded, int B, int C, int Z, int Y, int X,
                               void *stream);
/* (B,C,Z,Y,X) fp32 <-> interior of the padded channels-last bf16 grid (the halo
 * is not touched: allocate the grid zeroed once). */
int veon_volume_pack_bf16(const float *ncdhw, void *padded, int B, int C, int Z,
                          int Y, int X, void *stream);
int veon_volume_unpack_f32(const void *padded, float *ncdhw, int B, int C, int Z,
                           int Y, int X, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VEON_HIP_H_ */
