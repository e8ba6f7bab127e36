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
 * n <= 65536 (PVX_ERR_RANGE beyond). */
int pvx_nms(const float* boxes, int n, float thresh, int normal, int64_t* keep, int* num_keep,
            void* workspace, size_t workspace_bytes, void* stream);

/* `segments` independent NMS problems in one launch (e.g. 6 CenterHead heads x B frames,
 * pcdet/models/dense_heads/center_head.py:332-384 loops these in Python).  boxes is [segments, n_cap, 7];
 * counts[s] (device int32) says how many leading boxes of segment s are valid; keep is [segments, n_cap],
 * num_keep is [segments]. */
int pvx_nms_batched(const float* boxes, const int* counts, int segments, int n_cap, float thresh, int normal,
                    int64_t* keep, int* num_keep, void* workspace, size_t workspace_bytes, void* stream);

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

/* out[m, c] = voxels[m, T, c].sum(1) / max(num[m], 1).  MeanVFE.forward (mean_vfe.py:26-31) for callers that
 * already hold the padded voxel tensor. */
int pvx_mean_vfe(const float* voxels, const int* num_points_per_voxel, int m, int max_points_per_voxel, int c,
                 float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PCDET_MI355_H */
