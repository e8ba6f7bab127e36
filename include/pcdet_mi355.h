/*
 * libpcdet_mi355.so -- C ABI of the MI355X (gfx950) sparse-voxel hot path for OpenPCDet.
 *
 * Plain pointers and sizes only (no torch types).  Unless a function says "host", every pointer is DEVICE
 * memory on the current HIP device, every call is asynchronous on `stream` (a hipStream_t passed as void*;
 * NULL = the legacy default stream), performs no allocation and no host synchronisation, and returns
 * PVX_OK or a negative PVX_ERR_* code (it never exits or throws; compare the reference's exit(-1),
 * pcdet/ops/iou3d_nms/src/iou3d_nms.cpp:14-38).  Scratch memory comes from the caller through
 * (workspace, workspace_bytes); the matching *_workspace_bytes() query says how much is needed.
 *
 * Citations are file:line under the reference tree (hcheng1005/OpenPCDet); "[spconv]" marks behaviour of
 * the third-party spconv / torch_scatter packages the reference delegates to (not vendored there; semantics
 * restated in SURVEY.md section 8a and pinned by oracle/ in this repo).
 */
#ifndef PCDET_MI355_H
#define PCDET_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVX_OK 0
#define PVX_ERR_INVALID (-1)    /* bad argument (null pointer, negative size, unsupported shape/dtype) */
#define PVX_ERR_WORKSPACE (-2)  /* workspace missing or too small */
#define PVX_ERR_HIP (-3)        /* a HIP call or launch failed; see pvx_last_hip_error() */
#define PVX_ERR_RANGE (-4)      /* a size exceeds what this build supports (documented per function) */

/* dtype codes for feature / weight buffers */
#define PVX_F32 0
#define PVX_F16 1
#define PVX_BF16 2

/* library identity: returns a static string "pcdet_mi355 <version> gfx950" (host) */
const char* pvx_version(void);

/* last hipError_t recorded by a call that returned PVX_ERR_HIP (host) */
int pvx_last_hip_error(void);

/* ------------------------------------------------------------------------------------------------------
 * Rotated-BEV IoU / NMS   (replaces the pybind module pcdet.ops.iou3d_nms.iou3d_nms_cuda,
 *                          pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:11-20)
 * Boxes are row-major [x, y, z, dx, dy, dz, heading] float32, 7 per box.
 * ---------------------------------------------------------------------------------------------------- */

/* out[n, m] = BEV overlap AREA of a[i] and b[j].
 * Replaces boxes_overlap_bev_gpu (iou3d_nms.cpp:73-92 -> boxes_overlap_kernel, iou3d_nms_kernel.cu:236-249). */
int pvx_boxes_overlap_bev(const float* a, int n, const float* b, int m, float* out, void* stream);

/* out[n, m] = BEV IoU.  Replaces boxes_iou_bev_gpu (iou3d_nms.cpp:117-135 -> kernel.cu:279-293). */
int pvx_boxes_iou_bev(const float* a, int n, const float* b, int m, float* out, void* stream);

/* out[n] = BEV overlap area of a[i] and b[i].  Replaces both paired_boxes_overlap_bev_gpu
 * (iou3d_nms.cpp:94-115 -> kernel.cu:251-264) and boxes_aligned_overlap_bev_gpu (iou3d_nms.cpp:50-71 ->
 * kernel.cu:266-277), which compute the same thing. */
int pvx_boxes_overlap_bev_paired(const float* a, const float* b, int n, float* out, void* stream);

/* Bytes of workspace pvx_nms / pvx_nms_batched need for `segments` problems of at most n_cap boxes (host). */
size_t pvx_nms_workspace_bytes(int n_cap, int segments);

/* Greedy NMS over boxes ALREADY sorted by descending score.
 *   normal = 0: rotated BEV IoU   (nms_gpu,        iou3d_nms.cpp:137-183 -> nms_kernel,        kernel.cu:295-339)
 *   normal = 1: axis-aligned IoU  (nms_normal_gpu, iou3d_nms.cpp:186-233 -> nms_normal_kernel, kernel.cu:356-400)
 * keep[0 .. *num_keep) receives the kept indices (ascending, int64 like the reference's `long`), both on the
 * DEVICE: unlike the reference there is no device->host copy of the mask and no host loop.
 * n <= 262144 (PVX_ERR_RANGE beyond; the mask workspace is n * ceil(n / 64) * 8 bytes, 8.6 GB at that size). */
int pvx_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep, int* num_keep,
            void* workspace, size_t workspace_bytes, void* stream);

/* `segments` independent NMS problems in one launch (e.g. 6 CenterHead heads x B frames,
 * pcdet/models/dense_heads/center_head.py:332-384 loops these in Python).  boxes is [segments, n_cap, 7];
 * counts[s] (device int32) says how many leading boxes of segment s are valid; keep is [segments, n_cap],
 * num_keep is [segments]. */
int pvx_nms_batched(const float* boxes, const int* counts, int segments, int n_cap, float thresh, int normal,
                    int64_t* keep, int* num_keep, void* workspace, size_t workspace_bytes, void* stream);

/* class_agnostic_nms (pcdet/models/model_utils/model_nms_utils.py:6-25) for `segments` problems in three launches:
 * score threshold (:8-11; pass -INFINITY for none), top-pre_max by descending score (:15, ties by ascending index;
 * pre_max <= 0 = n_cap), box gather (:16), NMS (:17-19, normal as for pvx_nms), keep[:post_max] mapped back to the
 * caller's box indices (:20,23-24).  boxes is [segments, n_cap, box_stride >= 7] (columns 0..6 are used), scores
 * [segments, n_cap], counts[s] (device int32, may be NULL = n_cap) the leading valid boxes of segment s.
 * selected is [segments, post_max] int64 (entries at and beyond num_selected[s] repeat the best-scored index),
 * num_selected [segments].  n_cap <= 16384 (one problem's sort keys live in one CU's LDS; PVX_ERR_RANGE beyond:
 * sort on the caller's side and use pvx_nms_batched).  No host synchronisation. */
size_t pvx_nms_select_workspace_bytes(int n_cap, int segments);
int pvx_nms_select_batched(const float* boxes, int box_stride, const float* scores, const int* counts, int segments,
                           int n_cap, float score_thresh, int pre_max, float nms_thresh, int normal, int post_max,
                           int64_t* selected, int* num_selected, void* workspace, size_t workspace_bytes, void* stream);

/* pvx_nms for callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc) and bring no workspace: the suppression mask lives in a static device array inside the library
 * (PVX_NMS_STATIC_MAX^2 / 8 bytes = 2 MiB per GPU the library runs on, held until the process ends).  Calls no
 * hipMalloc / hipFree or any allocating routine; n > PVX_NMS_STATIC_MAX returns PVX_ERR_RANGE.  Same results as
 * pvx_nms (same kernels).  It has NO counterpart in the reference.  All calls on one GPU share that GPU's array:
 * calls that overlap on one GPU (different streams) race; calls on one stream are ordered and need no clearing.
 * The array's address is resolved on the first call per device (hipGetSymbolAddress; make that call outside capture). */
#define PVX_NMS_STATIC_MAX 4096
int pvx_nms_static(const float* boxes, int n, float thresh, int normal, int64_t* keep, int* num_keep, void* stream);

/* HOST functions on HOST pointers -- the reference keeps CPU entry points in the same module because
 * DataLoader workers call them (pcdet/datasets/augmentor/database_sampler.py:489-493); they create no GPU
 * context.  Replace boxes_iou_bev_cpu / boxes_aligned_iou_bev_cpu (iou3d_cpu.cpp:232-273). */
int pvx_boxes_iou_bev_cpu(const float* a, int n, const float* b, int m, float* out);
int pvx_boxes_aligned_iou_bev_cpu(const float* a, const float* b, int n, float* out);

/* ------------------------------------------------------------------------------------------------------
 * Voxelization + mean VFE
 * ---------------------------------------------------------------------------------------------------- */

/* HOST: grid_xyz[j] = round((range[3+j] - range[j]) / vsize[j]) in binary32 -- the grid Point2VoxelCPU3d derives
 * from (vsize_xyz, coors_range_xyz) [spconv]; pcdet computes the same in DataProcessor.__init__
 * (pcdet/datasets/processor/data_processor.py:66-78 -> calculate_grid_size). */
int pvx_voxel_grid_size(const float* vsize_xyz, const float* range_xyz, int* grid_xyz);

/* HOST: workspace bytes for pvx_voxelize_hard. */
size_t pvx_voxelize_hard_workspace_bytes(int n_points, int max_voxels, int max_points_per_voxel);

/* Hard, first-come voxelization of ONE frame on the device, with MeanVFE fused in.
 * Replaces VoxelGeneratorWrapper.generate -> spconv.utils.Point2VoxelCPU3d.point_to_voxel [spconv]
 * (pcdet/datasets/processor/data_processor.py:45-61, call site :200-247) and MeanVFE.forward
 * (pcdet/models/backbones_3d/vfe/mean_vfe.py:14-33).
 *   points [n, c] f32, xyz in the first three columns (c >= 3); vsize_xyz[3], range_xyz[6] are HOST arrays.
 *   voxels [max_voxels, T, c] f32 or NULL; coords_zyx [max_voxels, 3] i32; num_points_per_voxel [max_voxels] i32 or NULL;
 *   mean [max_voxels, c] f32 or NULL (= voxels.sum(1) / max(num, 1)); point_to_voxel [n] i32 or NULL (-1 = dropped);
 *   num_voxels: device i32, = M.  Only rows < M of the outputs are written.
 * Voxel ids follow first appearance in input order, a voxel keeps its first T points in input order, cells first
 * seen after max_voxels ids were handed out are dropped: identical to the sequential CPU algorithm, bit-exact. */
int pvx_voxelize_hard(const float* points, int n, int c, const float* vsize_xyz, const float* range_xyz,
                      int max_points_per_voxel, int max_voxels, float* voxels, int* coords_zyx,
                      int* num_points_per_voxel, float* mean, int* point_to_voxel, int* num_voxels,
                      void* workspace, size_t workspace_bytes, void* stream);

/* pvx_voxelize_hard for `frames` clouds of n points each in one set of launches (points [frames, n, c]); every output gains
 * a leading [frames] dimension (voxels [frames, max_voxels, T, c], coords_zyx [frames, max_voxels, 3], num_points_per_voxel
 * [frames, max_voxels], point_to_voxel [frames, n], num_voxels [frames]).  Clouds shorter than n are padded with NaN rows
 * (never in range).  This is the test-time DOUBLE_FLIP path of transform_points_to_voxels (data_processor.py:225-242):
 * the cloud and its y-, x- and xy-flipped copies voxelized together.  Workspace: pvx_voxelize_batch_workspace_bytes. */
int pvx_voxelize_hard_batch(const float* points, int frames, int n, int c, const float* vsize_xyz, const float* range_xyz,
                            int max_points_per_voxel, int max_voxels, float* voxels, int* coords_zyx,
                            int* num_points_per_voxel, int* point_to_voxel, int* num_voxels, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Frame `batch_idx` of a batch, voxelized straight into the backbone's input rows with no host round trip: the same
 * first-come voxelization as pvx_voxelize_hard, but the M voxels are APPENDED at rows [base, base + M) of the
 * batch-level arrays, base = *row_base_dev (NULL = 0), fusing MeanVFE.forward, the batch-index column that
 * collate_batch prepends (pcdet/datasets/dataset.py:258-265) and the dtype / channel-padding cast in front of the
 * first sparse conv:
 *   coords_bzyx [row_capacity, 4] i32 = (batch_idx, z, y, x);
 *   mean [row_capacity, mean_stride] of mean_dtype (PVX_*): channels [0, c) = mean of the voxel's first T points,
 *   channels [c, mean_stride) = 0;  *num_voxels = M;  *row_end_dev = min(base + M, row_capacity) (may alias
 *   row_base_dev: a running row counter).  Rows beyond row_capacity are dropped. */
int pvx_voxelize_mean_append(const float* points, int n, int c, const float* vsize_xyz, const float* range_xyz,
                             int max_points_per_voxel, int max_voxels, int batch_idx, const int* row_base_dev,
                             int row_capacity, int* coords_bzyx, void* mean, int mean_stride, int mean_dtype,
                             int* num_voxels, int* row_end_dev, void* workspace, size_t workspace_bytes, void* stream);

/* HOST: workspace bytes for pvx_voxelize_mean_batch. */
size_t pvx_voxelize_batch_workspace_bytes(int frames, int n_points, int max_voxels, int max_points_per_voxel);

/* A whole batch in one launch sequence (8 launches for any number of frames, one frame per blockIdx.y):
 * points [frames, n, c] (pad short frames with NaN rows: they fall out of range and are dropped); frame f's M_f voxels
 * land at rows [sum_{g<f} M_g, +M_f) with batch index f, exactly what collate_batch builds on the host
 * (pcdet/datasets/dataset.py:258-265); num_voxels [frames]; *row_total_dev = min(sum M_f, row_capacity).
 * Same per-frame first-come semantics and outputs as pvx_voxelize_mean_append. */
int pvx_voxelize_mean_batch(const float* points, int frames, int n, int c, const float* vsize_xyz, const float* range_xyz,
                            int max_points_per_voxel, int max_voxels, int row_capacity, int* coords_bzyx, void* mean,
                            int mean_stride, int mean_dtype, int* num_voxels, int* row_total_dev, void* workspace,
                            size_t workspace_bytes, void* stream);

/* out[m, c] = voxels[m, T, c].sum(1) / max(num[m], 1).  MeanVFE.forward (mean_vfe.py:26-31) for callers that
 * already hold the padded voxel tensor. */
int pvx_mean_vfe(const float* voxels, const int* num_points_per_voxel, int m, int max_points_per_voxel, int c,
                 float* out, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Dynamic voxelization + segment reductions (the dynamic VFEs)
 * ---------------------------------------------------------------------------------------------------- */

/* HOST: bytes of the voxel index / of the workspace pvx_dynamic_voxelize needs for `batch` frames on grid_xyz
 * (ndim = 3: voxels; ndim = 2: pillars, grid_xyz[2] ignored).  index bytes = 0 if the grid is unsupported. */
size_t pvx_dynamic_voxelize_index_bytes(int batch, const int* grid_xyz, int ndim);
size_t pvx_dynamic_voxelize_workspace_bytes(int batch, const int* grid_xyz, int ndim);

/* Dynamic voxelization of a batch: no per-voxel point cap, no voxel cap, voxels sorted ascending by the x-major key
 * b*gx*gy*gz + x*gy*gz + y*gz + z.  Replaces the floor/mask/merge_coords/torch.unique/decode sequence of
 * DynamicMeanVFE.forward (pcdet/models/backbones_3d/vfe/dynamic_mean_vfe.py:53-72), DynamicVoxelVFE.forward
 * (dynamic_voxel_vfe.py:60-71,96-101), DynamicPillarVFE.forward (dynamic_pillar_vfe.py:93-103,132-138; ndim = 2) and
 * DynamicPillarVFESimple2D.forward (:193-204,231-236).
 *   points [n, row_floats] f32, columns (batch_idx, x, y, z, ...), row_floats >= 4; (n, n_dev) as for the rulebooks;
 *   index: pvx_dynamic_voxelize_index_bytes of device memory (left holding the voxel index of the batch);
 *   point_voxel [n] i32 = unq_inv, -1 for points outside the grid (the reference removes those rows; callers that need
 *     the compacted form select rows with point_voxel >= 0);
 *   point_cell [n, 3] i32 = (cx, cy, cz) of each kept point (the reference's points_coords), or NULL;
 *   voxel_coords_bzyx [voxel_capacity, 4] i32 (z = 0 for pillars), voxel_num_points [voxel_capacity] i32 = unq_cnt (or NULL);
 *   *num_voxels (device) = number of voxels M; voxels of rank >= voxel_capacity are dropped (their points get -1). */
int pvx_dynamic_voxelize(const float* points, int n, const int* n_dev, int row_floats, const float* vsize_xyz,
                         const float* range_xyz, const int* grid_xyz, int ndim, int batch, void* index,
                         int voxel_capacity, int* point_voxel, int* point_cell, int* voxel_coords_bzyx,
                         int* voxel_num_points, int* num_voxels, void* workspace, size_t workspace_bytes, void* stream);

/* Per-point input rows of the pillar / voxel feature nets: out[i] = [points[i, 1:] (use_absolute_xyz) or points[i, 4:],
 * xyz - mean_xyz[voxel], xyz - cell centre, |xyz| (with_distance)], zeros for points with point_voxel < 0.
 * Replaces dynamic_pillar_vfe.py:106-121 (z_uses_cell = 0: z - z_offset) and dynamic_voxel_vfe.py:74-90
 * (z_uses_cell = 1).  center_offset_xyz[j] = vsize[j] / 2 + range[j] (HOST arrays, as vsize_xyz).
 * out_floats must equal the row length this produces. */
int pvx_dynamic_point_features(const float* points, int n, int row_floats, const int* point_voxel, const int* point_cell,
                               const float* mean_xyz, const float* vsize_xyz, const float* center_offset_xyz,
                               int z_uses_cell, int use_absolute_xyz, int with_distance, float* out, int out_floats,
                               void* stream);

/* scatter_mean(src [n, c], index [n], dim=0) -> out [m, c] = segment sum / max(count, 1); rows with index outside
 * [0, m) are skipped.  Replaces torch_scatter.scatter_mean (external; call sites dynamic_mean_vfe.py:65,
 * dynamic_pillar_vfe.py:105, dynamic_voxel_vfe.py:73).  Sums accumulate in binary64 and are rounded once. */
size_t pvx_scatter_mean_workspace_bytes(int m, int c);
int pvx_scatter_mean(const float* src, const int* index, int n, const int* n_dev, int c, int m, float* out,
                     void* workspace, size_t workspace_bytes, void* stream);

/* scatter_max(src [n, c], index [n], dim=0) -> out [m, c] (0 for empty segments), argmax [m, c] i32 or NULL (smallest
 * source row attaining the maximum; n for empty segments).  Replaces torch_scatter.scatter_max (PFNLayerV2.forward,
 * dynamic_pillar_vfe.py:40). */
int pvx_scatter_max(const float* src, const int* index, int n, const int* n_dev, int c, int m, float* out, int* argmax,
                    void* stream);

/* index_add_ along dim 0: out [m, c] (same dtype as src: PVX_F32 / F16 / BF16) = sum of the src rows of each segment, 0
 * for empty segments.  Replaces features_unique.index_add_(0, _inv, features_cat) in VoxelResBackBone8xVoxelNeXt.bev_out
 * (pcdet/models/backbones_3d/spconv_backbone_voxelnext.py:155-156).  Segments with one row are copied exactly; shared
 * segments accumulate in binary64 and round once (the reference rounds to the feature dtype after every addition, in
 * arrival order). */
size_t pvx_scatter_sum_workspace_bytes(int m, int c);
int pvx_scatter_sum(const void* src, int dtype, const int* index, int n, const int* n_dev, int c, int m, void* out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Sparse convolution: site index, rulebook (indice pairs), implicit-GEMM forward, dense()
 *
 * Replaces what the reference gets from the third-party spconv package [spconv] behind
 * SparseConvTensor / SubMConv3d / SparseConv3d / SparseInverseConv3d
 * (pcdet/utils/spconv_utils.py:3-10; call sites pcdet/models/backbones_3d/spconv_backbone.py:8-27,30-67,
 *  79-120,196-237; pcdet/models/backbones_3d/spconv_backbone_voxelnext.py:81-138).
 *
 * Conventions: indices rows are (b, z, y, x) int32, 4 per row; shape_dhw = {D, H, W}; kernel offset
 * k = (kz*KH + ky)*KW + kx; weight is the reference's [Cout, kz, ky, kx, Cin] (libraries/3DSparseConvolution/
 * tool/centerpoint-export/funcs.py:46-63, pcdet/models/detectors/detector3d_template.py:357-368);
 * the rulebook is the neighbour table nbr[K][n_out]: input row feeding output row o through offset k, or -1.
 *
 * Row counts that depend on the data live on the DEVICE: wherever a function takes (n, n_dev), `n` is the row
 * capacity (grid size and row stride of nbr) and `n_dev` -- optional, NULL = use n -- points at the live count
 * an earlier call on the same stream produced.  Nothing here synchronises with the host.
 * 2-D convolutions (SubMConv2d / SparseConv2d) are the D = 1, kz = 1 case.
 * ---------------------------------------------------------------------------------------------------- */

/* HOST: bytes of one site index (occupancy bitmap + rank prefix, 16 B per 64 cells) over batch x D x H x W;
 * 0 if the grid is invalid or beyond 2^34 cells. */
size_t pvx_site_index_bytes(int batch, const int* shape_dhw);
/* HOST: workspace bytes for pvx_site_index_build / pvx_conv_out_index on that grid. */
size_t pvx_site_index_workspace_bytes(int batch, const int* shape_dhw);

/* Build the site index of a coordinate set: afterwards rank(cell) = position of the cell in ascending linear key
 * ((b*D + z)*H + y)*W + x among the active cells.  perm (optional, [n]) receives perm[rank] = input row, i.e. the
 * map from key order to the caller's row order (hard voxels arrive in first-come order).  *n_unique (device)
 * = number of distinct active cells.  Rows with out-of-grid coordinates are ignored.
 * Replaces spconv's hash-table build inside get_indice_pairs [spconv]. */
int pvx_site_index_build(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, void* index,
                         int* perm, int* n_unique, void* workspace, size_t workspace_bytes, void* stream);

/* A caller that keeps its index buffers from frame to frame (a serving plan) need not pay a fill of the whole grid per frame (340 MB
 * for level 0 of the nuScenes grid at batch 16): pvx_site_index_build_clean / pvx_conv_out_index_clean are the two builders WITHOUT
 * their fill -- precondition: every bitmap word of `index` is zero -- and pvx_site_index_unset restores that state after use by zeroing
 * the entries of the words that hold the given sites (coords: the rows the index was built from, or the coords_out of
 * pvx_conv_out_rulebook for an output index; n = their capacity, *n_dev the live count).  If *n_dev > n (rows were dropped at the
 * capacity, so some marked cells are not in coords) the whole index is cleared instead.  No reference counterpart (spconv
 * allocates and fills a fresh hash table per call). */
int pvx_site_index_build_clean(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, void* index,
                               int* perm, int* n_unique, void* workspace, size_t workspace_bytes, void* stream);
int pvx_site_index_unset(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, void* index, void* stream);

/* torch.unique(coords, dim=0, return_inverse=True) for coordinate rows: coords_out [out_capacity, 4] = the distinct
 * rows in ascending (b, z, y, x) order, inverse [n] = row of coords_out each input row maps to (-1 for rows outside the
 * grid or beyond out_capacity), *n_unique = their number; `index` is left holding the site index of coords_out.
 * Replaces spconv_backbone_voxelnext.py:154 (bev_out; pass z = 0 and shape {1, H, W} for its (b, y, x) rows).
 * Workspace: pvx_site_index_workspace_bytes. */
int pvx_unique_sites(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, void* index,
                     int out_capacity, int* coords_out, int* inverse, int* n_unique, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Rows into ascending key order: for every live row i with rank r in `index` (built by pvx_site_index_build over the
 * same coords, which must be unique), coords_out[r] = coords[i], rows_out[r] = rows[i] (row_bytes per row, a multiple
 * of 16) and rank_out[i] = r (optional).  Out of place.  After it the index addresses the sorted rows directly (perm =
 * NULL in pvx_subm_rulebook / pvx_conv_out_rulebook).  The reference has no counterpart: spconv leaves hard voxels in
 * the voxelizer's first-come order (data_processor.py:200-247) and pays for it in its gather kernels; this is the
 * serving plan's locality step (openpcdet_amd/engine.py). */
int pvx_sort_sites(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, const void* index,
                   const void* rows, int row_bytes, int* coords_out, void* rows_out, int* rank_out, void* stream);

/* Submanifold rulebook: nbr[k][o] = row of the site at coords[o] + (k - K/2) * dilation, or -1.  Output rows
 * are the input rows (SubMConv3d, spconv_backbone.py:12,39-46).  ksize odd per axis. */
int pvx_subm_rulebook(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, const void* index,
                      const int* perm, const int* ksize, const int* dilation, int* nbr, void* stream);

/* HOST: out = floor((in + 2p - d(k-1) - 1)/s) + 1 per axis (spconv.ops.get_conv_output_size [spconv], used by
 * libraries/3DSparseConvolution/tool/centerpoint-export/exptool.py:76-82). */
int pvx_conv_out_shape(const int* in_shape, const int* ksize, const int* stride, const int* padding,
                       const int* dilation, int* out_shape);

/* Strided sparse conv (SparseConv3d, spconv_backbone.py:14-15), step 1: mark every output cell reached by at least
 * one (input site, kernel offset) and build the OUTPUT grid's site index; *n_out (device) = number of output sites.
 * out_index has pvx_site_index_bytes(batch, out_shape) bytes and doubles as the site index of the output tensor. */
int pvx_conv_out_index(const int* coords_in, int n_in, const int* n_in_dev, int batch, const int* in_shape,
                       const int* ksize, const int* stride, const int* padding, const int* dilation, void* out_index,
                       int* n_out, void* workspace, size_t workspace_bytes, void* stream);

/* The same without the fill of out_index (see pvx_site_index_build_clean). */
int pvx_conv_out_index_clean(const int* coords_in, int n_in, const int* n_in_dev, int batch, const int* in_shape,
                             const int* ksize, const int* stride, const int* padding, const int* dilation, void* out_index,
                             int* n_out, void* workspace, size_t workspace_bytes, void* stream);

/* Step 2: coords_out [n_out, 4] in ascending key order (the row-order contract) and nbr [K][n_out] against the
 * input tensor's site index (in_perm as in pvx_site_index_build, NULL when input rows are already in key order). */
int pvx_conv_out_rulebook(const void* out_index, int n_out, const int* n_out_dev, int batch, const int* in_shape,
                          const int* ksize, const int* stride, const int* padding, const int* dilation,
                          const void* in_index, const int* in_perm, int* coords_out, int* nbr, void* stream);

/* Transposed use of a forward table (SparseInverseConv3d, spconv_backbone.py:17): nbr_inv[k][i] = o  <=>
 * nbr_fwd[k][o] = i.  nbr_fwd is [K][n_out_fwd], nbr_inv is [K][n_in_fwd]. */
int pvx_inverse_rulebook(const int* nbr_fwd, int kvol, int n_out_fwd, int n_in_fwd, int* nbr_inv, void* stream);

/* HOST: 1 when (cin, cout, dtype) runs on the MFMA implicit-GEMM kernel, 0 when on the generic fp32-math kernel. */
int pvx_sparse_conv_uses_mfma(int cin, int cout, int dtype);

/* HOST: bytes of the kernel-ready copy of a [cout, kvol, cin] weight. */
size_t pvx_sparse_conv_weight_bytes(int cout, int kvol, int cin, int dtype);
/* Re-lay a [cout, kvol, cin] weight (the module's parameter, dtype PVX_*) into the kernel's fragment order.
 * Done once per layer per weight update, not per forward. */
int pvx_sparse_conv_prepare_weight(const void* weight, int cout, int kvol, int cin, int dtype, void* prepared,
                                   void* stream);

/* out[o, :] = act( (bias + sum_k W[:, k, :] . feat_in[nbr[k][o], :]) * scale + shift + residual[o, :] )
 * for o < live rows.  feat_in [n_in, cin], out / residual [n_out, cout], all of `dtype`; bias / scale / shift are
 * f32 [cout] or NULL (scale and shift together: an eval-mode BatchNorm1d folded in, as the reference's own export
 * path folds it, centerpoint-export/funcs.py:46-63); relu != 0 applies max(., 0).  Accumulation is fp32.
 * Replaces SparseConvolution.forward's indice_conv / implicit_gemm [spconv]; one write per output row, no atomics. */
int pvx_sparse_conv_fwd(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                        const int* nbr, int kvol, int n_out, const int* n_out_dev, int cout, int dtype,
                        const float* scale, const float* shift, const void* residual, int relu, void* out, void* stream);

/* Tile book: a submanifold / strided rulebook with kvol = 27 re-packed for the slab conv kernels of one window size (no
 * reference counterpart; spconv keeps its pair tables private too).  Per 256-site tile: the row window of each dz group and
 * 27 x 256 u16 neighbour rows relative to it (layout: csrc/sparse_conv_slab_tb.h).  Halves the rulebook bytes every conv
 * layer re-reads and lets the kernel prefetch the next tile's block by LDS-DMA.
 *   pvx_tilebook_rcap   window rows the tile-book kernel of (cin, cout, kvol, dtype) wants, 0 = that shape has none
 *   pvx_tilebook_bytes  size of the book for a rulebook of n_out (capacity) columns
 *   pvx_tilebook_build  nbr [27, n_out] -> book; n_in = rows of the conv's input (entries >= n_in count as absent)
 * pvx_sparse_conv_fwd_tb = pvx_sparse_conv_fwd with the book of ITS rulebook (built with ITS rcap); tilebook NULL or a
 * shape without the form falls back to pvx_sparse_conv_fwd's kernels.  Results are bit-identical either way. */
int pvx_tilebook_rcap(int cin, int cout, int kvol, int dtype);
size_t pvx_tilebook_bytes(int n_out);
int pvx_tilebook_build(const int* nbr, int kvol, int n_out, const int* n_out_dev, int n_in, int rcap, void* book, void* stream);
/* Tiles dealt to workgroups by cost (no reference counterpart).  The persistent conv kernels walk their tiles round-robin; tiles differ
 * in cost (dz groups without pairs, groups with far codes) and a launch ends with the workgroup that drew the most.  After a book has
 * been built (any builder; each one invalidates the previous schedule), pvx_tilebook_schedule writes, into the book's own tail, which
 * tiles each workgroup of the kernel of (cin, cout, kvol, dtype) takes: same tiles, same sums, a shorter tail.  A shape whose kernel
 * takes no schedule is left alone (PVX_OK).  The book must have been sized by pvx_tilebook_bytes(n_out) with this n_out.
 * book == NULL: HOST query, 1 if that kernel walks a schedule for a book of n_out columns, else 0 (nothing is launched). */
int pvx_tilebook_schedule(void* book, int n_out, const int* n_out_dev, int cin, int cout, int kvol, int dtype, void* stream);
/* pvx_subm_rulebook for a 3 x 3 x 3 kernel and pvx_tilebook_build of its rulebook (n_in = n) in ONE pass: the 27 entries of a
 * tile's sites are in registers when the rulebook is written, so the book costs its stores only.  Same nbr, same book. */
int pvx_subm_rulebook_tb(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, const void* index,
                         const int* perm, const int* dilation, int* nbr, int rcap, void* book, void* stream);
int pvx_sparse_conv_fwd_tb(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                           const int* nbr, int kvol, int n_out, const int* n_out_dev, int cout, int dtype,
                           const float* scale, const float* shift, const void* residual, int relu, void* out,
                           const void* tilebook, int tilebook_rcap, void* stream);
/* When EVERY conv layer that reads a submanifold rulebook reads it through one book (an inference plan: the layers of a level
 * share a shape), the int32 rulebook is needed only where that book says "far" (a neighbour row outside the tile's window: a few
 * per cent of the entries), not at 108 B per site.
 *   pvx_subm_book            = pvx_subm_rulebook_tb that stores only those entries of nbr_far (same [27][n] addressing; every other
 *                              entry is left untouched and must not be read)
 *   pvx_sparse_conv_fwd_book = pvx_sparse_conv_fwd_tb restricted to the kernels that read the rulebook through the book: a call that
 *                              none of them takes returns PVX_ERR_RANGE (no fall-back to a form that would read all of nbr).
 * Results are bit-identical to pvx_subm_rulebook_tb + pvx_sparse_conv_fwd_tb.  No reference counterpart. */
/* pvx_conv_out_rulebook for a 3 x 3 x 3 kernel that also writes the tile book of its rulebook (window rcap; entries >= n_in count as
 * absent, as in pvx_tilebook_build) in the same pass; far_only != 0: nbr receives only the entries the book marks "far" (for
 * pvx_sparse_conv_fwd_book), else the whole rulebook as pvx_conv_out_rulebook writes it.  Same values either way. */
int pvx_conv_out_book(const void* out_index, int n_out, const int* n_out_dev, int batch, const int* in_shape, const int* ksize,
                      const int* stride, const int* padding, const int* dilation, const void* in_index, const int* in_perm,
                      int* coords_out, int* nbr, int n_in, int rcap, void* book, int far_only, void* stream);
/* HOST: 1 if pvx_sparse_conv_fwd_book takes a call of these sizes (a book-form kernel exists for the shape, the window matches, and the
 * feature tensors are below the 2 GiB its 32-bit buffer offsets address), else 0: a plan asks before it stores a rulebook far-only. */
int pvx_sparse_conv_book_form(int n_in, int cin, int cout, int kvol, int n_out, int dtype, int tilebook_rcap);
int pvx_subm_book(const int* coords, int n, const int* n_dev, int batch, const int* shape_dhw, const void* index,
                  const int* perm, const int* dilation, int* nbr_far, int rcap, void* book, void* stream);
int pvx_sparse_conv_fwd_book(const void* feat_in, int n_in, int cin, const void* prepared_weight, const float* bias,
                             const int* nbr_far, int kvol, int n_out, const int* n_out_dev, int cout, int dtype,
                             const float* scale, const float* shift, const void* residual, int relu, void* out,
                             const void* tilebook, int tilebook_rcap, void* stream);

/* Backward, weight gradient: grad_weight[co, k, ci] (f32, the module parameter's [cout, kvol, cin] layout) =
 * sum over live output rows o of grad_out[o, co] * feat_in[nbr[k][o], ci]; feat_in / grad_out are of `dtype`, fp32
 * accumulation, fixed summation order (partials per site chunk, then one ordered reduction: no float atomics).
 * Replaces the wgrad of spconv's SparseConvFunction / SubMConvFunction.backward [spconv] (training step, tools/train.py
 * through every conv of spconv_backbone.py).  The INPUT gradient has no entry point of its own: it is
 * pvx_sparse_conv_fwd over the inverse rulebook (pvx_inverse_rulebook; for submanifold convs the same rulebook with the
 * kernel offsets flipped) with the weight transposed to [cin, kvol, cout]. */
size_t pvx_sparse_conv_wgrad_workspace_bytes(int n_out, int kvol, int cin, int cout);
int pvx_sparse_conv_wgrad(const void* feat_in, int n_in, int cin, const void* grad_out, const int* nbr, int kvol, int n_out,
                          const int* n_out_dev, int cout, int dtype, float* grad_weight, void* workspace,
                          size_t workspace_bytes, void* stream);

/* SparseConvTensor.dense() (pcdet/models/backbones_2d/map_to_bev/height_compression.py:22): zero-fill
 * out [batch, c, D, H, W] and scatter the rows, writing the channel-major layout directly. */
int pvx_sparse_to_dense(const void* feat, const int* coords, int n, const int* n_dev, int c, int dtype, int batch,
                        const int* shape_dhw, void* out, void* stream);

/* The same tensor for rows that are in ascending (b, z, y, x) order -- row i is the site of rank i in `index`
 * (pvx_site_index_build of sorted rows, or the output index of pvx_conv_out_index: every level of spconv_backbone.py's
 * strided chain, so also the tensor height_compression.py:22 densifies).  One workgroup per group of y lines writes its run
 * of every channel plane in 8-byte stores, zeros included: no separate zero-fill, no scattered 2-byte stores.  16-bit dtypes,
 * even c, W % 4 == 0 and a line's share of the rows within 64 KiB of LDS, else PVX_ERR_RANGE (use pvx_sparse_to_dense).  When n
 * is 0 (with n_dev NULL) the index must still be a built (empty) one. */
int pvx_sparse_to_dense_sorted(const void* feat, const void* index, int n, const int* n_dev, int c, int dtype, int batch,
                               const int* shape_dhw, void* out, void* stream);

/* ---- CenterPoint head decode ------------------------------------------------------------
 * Per frame: the K largest values of heatmap[b] ([C, H, W], already sigmoid-ed) in descending order (equal values: lower
 * flat index c*H*W + y*W + x first), each decoded to x, y, z, dx, dy, dz, heading[, vx, vy] with the reference's float32
 * operation order, masked by post_center_limit_range (x, y, z within [lim[0:3], lim[3:6]]) and, when score_thresh is not
 * NULL, by score > *score_thresh; survivors are written in score order to boxes[b, 0:counts[b]] ([B, K, 7] or [B, K, 9]
 * with vel), scores[b, :], labels[b, :] (class index inside this head).  center [B,2,H,W], center_z [B,1,H,W],
 * dim [B,3,H,W] (already exp-ed), rot_cos / rot_sin: one [H,W] plane per frame, rot_batch_stride elements apart (2*H*W
 * when both are channel views of one [B,2,H,W] tensor), vel [B,2,H,W] or NULL.  K <= 1024.
 * Replaces centernet_utils.py:163-254 (_topk, _transpose_and_gather_feat, decode_bbox_from_heatmap) as called from
 * center_head.py:343-352; circle_nms and the iou branch are not taken by the reference's configs and are not built. */
int pvx_center_decode(const float* heatmap, const float* center, const float* center_z, const float* dim,
                      const float* rot_cos, const float* rot_sin, long long rot_batch_stride, const float* vel, int batch,
                      int num_class, int h, int w, int k, float feature_map_stride, const float* voxel_size_xy,
                      const float* range_min_xy, const float* post_center_limit_range, const float* score_thresh,
                      float* boxes, float* scores, int* labels, int* counts, void* stream);

/* The same selection + decode for a SPARSE head (VoxelNeXt): frame b owns rows [row_begin[b], row_begin[b+1]) of
 * obj [N, C] (scores, already sigmoid-ed; rows of a frame contiguous), flat index c * rows + r; center [N,2],
 * center_z [N,1], dim [N,3] (exp-ed), rot_cos / rot_sin [N,1], vel [N,2] or NULL, spatial [N, spatial_dims] int32 voxel
 * indices ending in (y, x); stride_rows [N] (one feature-map stride per row) or NULL for the scalar feature_map_stride.
 * x = ((x_idx + center_x) * stride) * voxel_x + range_min_x, likewise y.  Outputs as pvx_center_decode.
 * Replaces centernet_utils.py:256-367 (_topk_1d with nuscenes=True, gather_feat_idx, decode_bbox_from_voxels_nuscenes)
 * as called from voxelnext_head.py:generate_predicted_boxes. */
int pvx_voxel_decode(const float* obj, const int* row_begin, const int* spatial, int spatial_dims, const float* center,
                     const float* center_z, const float* dim, const float* rot_cos, const float* rot_sin, const float* vel,
                     const float* stride_rows, int batch, int num_class, int k, float feature_map_stride,
                     const float* voxel_size_xy, const float* range_min_xy, const float* post_center_limit_range,
                     const float* score_thresh, float* boxes, float* scores, int* labels, int* counts, void* stream);

/* ---- voxel query ---------------------------------------------------------------------------
 * For each of the m queries (new_xyz [m,3], new_coords [m,4] = b, z, y, x cell): the rows of up to nsample occupied voxels in
 * the cell neighbourhood +-max_range_zyx whose centre xyz[row] is within `radius` of the query, in ascending (dz, dy, dx)
 * order; a row is pre-filled with its first hit; queries without a hit get zeros and empty_mask[q] = 1 (optional).
 * point_indices [batch, z, y, x] int32 is the dense voxel -> row map (-1 = empty; common_utils.py:253-270).
 * Replaces voxel_query_gpu.cu:10-89 + voxel_query_utils.py:34-41 (VoxelQuery.forward). */
int pvx_voxel_query(const float* new_xyz, const int* new_coords, int m, const float* xyz, int n_voxels, const int* point_indices,
                    int batch, int z, int y, int x, const int* max_range_zyx, float radius, int nsample, int* idx,
                    unsigned char* empty_mask, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * VectorPool aggregation (PV-RCNN++ second stage).  Stacked layout as in the reference:
 * support_xyz [N, 3] / support_features [N, C] hold the frames one after another, xyz_batch_cnt [batch] (device int32) their
 * lengths; new_xyz [M, 3] and new_xyz_batch_cnt likewise.  neighbor_type 1 = ball, else cube; nsample <= 0 = no limit.
 * ---------------------------------------------------------------------------------------------------- */

/* For every new point the first min(nsample, 1000) support points of its frame within max_neighbour_distance (index order), then
 * for each of its num_total_grids grid centres [M, G, 3] the three nearest of those: dist2 [M, G, 3] (squared; +inf when the
 * list is empty), idx [M, G, 3] (row in support_xyz; -1 when empty; the nearest repeated when fewer than three).
 * *total_neighbors (device) = summed list lengths.  Replaces query_stacked_local_neighbor_idxs_kernel +
 * query_three_nn_by_stacked_local_idxs_kernel (vector_pool_gpu.cu:122-205, 19-86) and the guess-and-retry loop around them
 * (pointnet2_utils.py:331-353): the list stays in LDS, nothing is stacked in HBM. */
int pvx_vector_pool_three_nn(const float* support_xyz, int n_support, const int* xyz_batch_cnt, const float* new_xyz,
                             const float* new_xyz_grid_centers, const int* new_xyz_batch_cnt, int batch_size, int m,
                             int num_total_grids, float max_neighbour_distance, int nsample, int neighbor_type,
                             float* dist2, int* idx, int* total_neighbors, void* stream);

/* vector_pool_kernel_stack (vector_pool_gpu.cu:243-373): per new point and local grid cell (gx * gy * gz cells over the cube of
 * half-width max_neighbour_distance) the SUM (pooling_type 0) or the FIRST (1) of the support features / local offsets that
 * fall into it; new_features [M, G * c_each] (input channel i lands in i % c_each), new_local_xyz [M, G * 3],
 * point_cnt_of_grid [M, G], grouped_idxs [num_max_sum_points, 3] = (support row, new point, cell) of every pooled point for
 * the backward pass, *cum_sum (device) = how many there are (entries beyond the buffer are dropped: call again with
 * num_max_sum_points >= *cum_sum; unlike the reference the count does not depend on the buffer size).  Outputs are zeroed
 * by the call; division by the counts is the caller's (pointnet2_utils.py:419-423).  G <= 2048. */
int pvx_vector_pool(const float* support_xyz, const float* support_features, int n_support, const int* xyz_batch_cnt,
                    const float* new_xyz, const int* new_xyz_batch_cnt, int batch_size, int m, int num_c_in,
                    int num_c_each_grid, int num_grid_x, int num_grid_y, int num_grid_z, float max_neighbour_distance,
                    int use_xyz, int nsample, int neighbor_type, int pooling_type, float* new_features,
                    float* new_local_xyz, int* point_cnt_of_grid, int* grouped_idxs, int num_max_sum_points,
                    int* cum_sum, void* stream);

/* vector_pool_grad_kernel_stack (vector_pool_gpu.cu:433-458): grad_support_features [N, C] (zeroed by the call) +=
 * grad_new_features[new point, cell * c_each + channel % c_each] / max(count, 1) for every grouped entry. */
int pvx_vector_pool_grad(const float* grad_new_features, const int* point_cnt_of_grid, const int* grouped_idxs,
                         long long num_entries, int m, int n_support, int num_c_in, int num_c_each_grid,
                         int num_total_grids, float* grad_support_features, void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Training-mode BatchNorm1d over a sparse tensor's [n, c] feature matrix fused with (+ residual) -> ReLU: the
 * norm_fn -> ReLU tail of post_act_block (spconv_backbone.py:21-25) and bn -> (+= identity) -> relu of
 * SparseBasicBlock.forward (:52-64) in the training step (BASELINE.json configs[3]).  dtype = PVX_F32 / F16 / BF16 for x, y,
 * residual, dy, dx, dres; gamma, beta, statistics and their gradients are f32.  c must be a multiple of 16 bytes of elements
 * and <= 128 (pvx_bn_act_supported).  Semantics of torch.nn.functional.batch_norm(training=True): biased batch variance for
 * the normalisation, running statistics (NULL = not tracked) updated with `momentum` and the unbiased variance.
 *   fwd: y = relu?(gamma (x - mean) invstd + beta (+ residual)); save_mean / save_invstd [c] are kept for the backward pass
 *   bwd: g = dy masked by y > 0 (when relu); dbeta = sum g, dgamma = sum g xhat,
 *        dx = gamma invstd (g - mean(g) - xhat mean(g xhat)), dres = g (NULL = no residual)
 * ---------------------------------------------------------------------------------------------------- */
int pvx_bn_act_supported(int c, int dtype);
size_t pvx_bn_act_workspace_bytes(int c);
int pvx_bn_act_fwd(const void* x, int n, int c, int dtype, const float* gamma, const float* beta, float eps, float momentum,
                   float* running_mean, float* running_var, const void* residual, int relu, void* y, float* save_mean,
                   float* save_invstd, void* workspace, size_t workspace_bytes, void* stream);
int pvx_bn_act_bwd(const void* dy, const void* x, const void* y, int n, int c, int dtype, const float* gamma,
                   const float* save_mean, const float* save_invstd, int relu, void* dx, void* dres, float* dgamma,
                   float* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* The same two ops for SyncBatchNorm (tools/train.py:130-131): statistics over the rows of ALL ranks.  phase 1 leaves the local
 * column sums in sums[2 c + 1] (f64, device; forward: sum x, sum x^2; backward: sum g, sum g xhat, and dbeta / dgamma = the LOCAL
 * sums, which DDP averages like every other parameter gradient; the last entry = the local row count) and returns; the caller
 * all-reduces `sums` over its process group (RCCL) and calls again with phase 2, which finishes over n_total rows (n_total <= 0: the
 * all-reduced count in sums[2 c], no host read) and applies.  phase 0 is pvx_bn_act_fwd / _bwd. */
int pvx_bn_act_fwd_sync(const void* x, int n, int c, int dtype, const float* gamma, const float* beta, float eps,
                        float momentum, float* running_mean, float* running_var, const void* residual, int relu, void* y,
                        float* save_mean, float* save_invstd, double* sums, long long n_total, int phase, void* workspace,
                        size_t workspace_bytes, void* stream);
int pvx_bn_act_bwd_sync(const void* dy, const void* x, const void* y, int n, int c, int dtype, const float* gamma,
                        const float* save_mean, const float* save_invstd, int relu, void* dx, void* dres, float* dgamma,
                        float* dbeta, double* sums, long long n_total, int phase, void* workspace, size_t workspace_bytes,
                        void* stream);

/* ----------------------------------------------------------------------------------------------------
 * Point-level preparation ahead of the voxelizer: the world augmentations and the range mask /
 * shuffle the reference runs with numpy in DataLoader workers.  binary32 arithmetic in the reference's order (numpy 1.x casting:
 * the drawn float64 scalars act as float32), no contraction.
 * ---------------------------------------------------------------------------------------------------- */
enum {
    PVX_AUG_FLIP_X = 0,     /* random_flip_along_x, augmentor_utils.py:8-26: y -> -y, heading -> -heading, vy -> -vy */
    PVX_AUG_FLIP_Y = 1,     /* random_flip_along_y, :29-47: x -> -x, heading -> -(heading + pi), vx -> -vx */
    PVX_AUG_ROTATE = 2,     /* global_rotation, :50-71: a = cos, b = sin, c = angle (about z, x towards y) */
    PVX_AUG_SCALE = 3,      /* global_scaling, :74-91: a = scale */
    PVX_AUG_TRANSLATE = 4,  /* random_world_translation, data_augmentor.py:140-161: (a, b, c) added to xyz */
    PVX_AUG_LIMIT_HEADING = 5 /* boxes only, data_augmentor.py:309-311: heading -= floor(heading / a + b) * a (a = period) */
};
typedef struct { int kind; float a, b, c; } pvx_point_op;
#define PVX_AUG_MAX_OPS 16

/* Apply n_ops (HOST array, <= PVX_AUG_MAX_OPS) operations, in order, in place: to columns 0..2 of points [n, c] (c >= 3), or to
 * boxes [n, c] = (x, y, z, dx, dy, dz, heading[, vx, vy, ...]) with c == 7 or c >= 9 (columns >= 7 only scale). */
int pvx_points_transform(float* points, int n, int c, const pvx_point_op* ops, int n_ops, void* stream);
int pvx_boxes_transform(float* boxes, int n, int c, const pvx_point_op* ops, int n_ops, void* stream);

/* mask_points_and_boxes_outside_range + shuffle_points (data_processor.py:81-112) for the points of one frame: out receives the
 * rows points[perm[i]] (perm: device int64 permutation of 0..n-1; NULL = the keyed bijection of [0, n) that shuffle_seed selects
 * -- a 6-round Feistel network with cycle walking, computed in the kernel, for a shuffle without a permutation array or a sort
 * (the reference draws np.random.permutation, :107-110) -- or, with shuffle_seed == 0, the identity) that pass x0 <= x <= x1 && y0 <= y <= y1
 * (range_xy: HOST {x0, y0, x1, y1}, NULL = keep all; common_utils.py:78-81), in order of i; *count (device) = rows written.  A
 * uniform permutation followed by this stable compaction orders the kept rows uniformly at random, which is what shuffling after
 * the mask does.  out must not alias points; out has room for n rows. */
size_t pvx_points_prepare_workspace_bytes(int n);
int pvx_points_prepare(const float* points, int n, int c, const int64_t* perm, uint64_t shuffle_seed, const float* range_xy,
                       float* out, int* count, void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCDET_MI355_H */
