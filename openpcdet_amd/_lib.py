"""ctypes binding of libpcdet_mi355.so (the C ABI declared in include/pcdet_mi355.h).

The library is the product: if it cannot be loaded every op raises -- there is no CPU or PyTorch fallback.
PyTorch only supplies device memory (tensor.data_ptr()) and the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PVX_LIB") or os.path.join(_HERE, "lib", "libpcdet_mi355.so")   # PVX_LIB: a diagnostic build of the same library

PVX_OK = 0
_ERR = {-1: "PVX_ERR_INVALID", -2: "PVX_ERR_WORKSPACE", -3: "PVX_ERR_HIP", -4: "PVX_ERR_RANGE"}
PVX_F32, PVX_F16, PVX_BF16 = 0, 1, 2
DTYPE_CODE = {torch.float32: PVX_F32, torch.float16: PVX_F16, torch.bfloat16: PVX_BF16}

_lib = None

c_void_p, c_int, c_float, c_size_t, c_int64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_int64

# name -> (restype, argtypes); must list every symbol include/pcdet_mi355.h declares
SIGNATURES = {
    "pvx_version": (ctypes.c_char_p, []),
    "pvx_last_hip_error": (c_int, []),
    "pvx_boxes_overlap_bev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "pvx_boxes_iou_bev": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "pvx_boxes_overlap_bev_paired": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pvx_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pvx_nms": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_nms_batched": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    "pvx_nms_select_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pvx_nms_select_batched": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_float, c_int, c_int,
                                       c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_vector_pool_three_nn": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int,
                                         c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pvx_vector_pool": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                c_int, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                c_void_p]),
    "pvx_vector_pool_grad": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p]),
    "pvx_bn_act_supported": (c_int, [c_int, c_int]),
    "pvx_bn_act_workspace_bytes": (c_size_t, [c_int]),
    "pvx_bn_act_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_bn_act_fwd_sync": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p, c_size_t, c_void_p]),
    "pvx_bn_act_bwd_sync": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p, c_size_t, c_void_p]),
    "pvx_bn_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_points_transform": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "pvx_boxes_transform": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "pvx_points_prepare_workspace_bytes": (c_size_t, [c_int]),
    "pvx_points_prepare": (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                   c_void_p]),
    "pvx_nms_static": (c_int, [c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "pvx_boxes_iou_bev_cpu": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "pvx_boxes_aligned_iou_bev_cpu": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "pvx_voxel_grid_size": (c_int, [c_void_p, c_void_p, c_void_p]),
    "pvx_voxelize_hard_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pvx_voxelize_hard_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_voxelize_hard": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_voxelize_mean_append": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                         c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_voxelize_batch_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pvx_voxelize_mean_batch": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                        c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_mean_vfe": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pvx_dynamic_voxelize_index_bytes": (c_size_t, [c_int, c_void_p, c_int]),
    "pvx_dynamic_voxelize_workspace_bytes": (c_size_t, [c_int, c_void_p, c_int]),
    "pvx_dynamic_voxelize": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                     c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_dynamic_point_features": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_int, c_int, c_void_p, c_int, c_void_p]),
    "pvx_scatter_mean_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pvx_scatter_mean": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_scatter_max": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "pvx_scatter_sum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "pvx_scatter_sum": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_unique_sites": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_size_t, c_void_p]),
    "pvx_center_decode": (c_int, [c_void_p] * 6 + [ctypes.c_longlong, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_float,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pvx_voxel_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int] + [c_void_p] * 7 + [c_int, c_int, c_int, ctypes.c_float] +
                         [c_void_p] * 9),
    "pvx_voxel_query": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                        ctypes.c_float, c_int, c_void_p, c_void_p, c_void_p]),
    "pvx_sort_sites": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                               c_void_p, c_void_p]),
    "pvx_site_index_bytes": (c_size_t, [c_int, c_void_p]),
    "pvx_site_index_workspace_bytes": (c_size_t, [c_int, c_void_p]),
    "pvx_site_index_build": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "pvx_site_index_build_clean": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_size_t, c_void_p]),
    "pvx_site_index_unset": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "pvx_conv_out_index_clean": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_subm_book": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                              c_void_p, c_void_p]),
    "pvx_subm_rulebook": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    "pvx_subm_rulebook_tb": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p]),
    "pvx_conv_out_shape": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pvx_conv_out_index": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "pvx_conv_out_rulebook": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pvx_conv_out_book": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "pvx_inverse_rulebook": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pvx_sparse_conv_uses_mfma": (c_int, [c_int, c_int, c_int]),
    "pvx_sparse_conv_weight_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pvx_sparse_conv_prepare_weight": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "pvx_sparse_conv_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                    c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "pvx_sparse_conv_fwd_tb": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                       c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "pvx_sparse_conv_fwd_book": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                         c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "pvx_sparse_conv_book_form": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "pvx_tilebook_rcap": (c_int, [c_int, c_int, c_int, c_int]),
    "pvx_tilebook_bytes": (c_size_t, [c_int]),
    "pvx_tilebook_build": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pvx_tilebook_schedule": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pvx_sparse_conv_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "pvx_sparse_conv_wgrad": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "pvx_sparse_to_dense": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p]),
    "pvx_sparse_to_dense_sorted": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                           c_void_p]),
}


def int3(v):
    """host int[3] from an int or a 3-sequence (kernel_size / stride / padding / dilation / shape arguments)."""
    if isinstance(v, int):
        v = (v, v, v)
    v = [int(x) for x in v]
    assert len(v) == 3
    return (ctypes.c_int * 3)(*v)


def lib():
    """Load (once) and return the shared library; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libpcdet_mi355.so not found at %s -- build it with `python -m openpcdet_amd.build` "
                "(there is no fallback path)" % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc, what):
    if rc != PVX_OK:
        extra = ""
        if rc == -3:
            extra = " (hipError %d)" % lib().pvx_last_hip_error()
        raise RuntimeError("%s failed: %s%s" % (what, _ERR.get(rc, str(rc)), extra))


def ptr(t):
    """Device (or host) pointer of a contiguous tensor, None for None/empty."""
    if t is None:
        return None
    assert t.is_contiguous(), "tensor must be contiguous"
    return t.data_ptr() if t.numel() > 0 else None


def stream(device=None):
    """Raw hipStream_t of PyTorch's current stream (as an integer for ctypes c_void_p)."""
    return torch.cuda.current_stream(device).cuda_stream


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
