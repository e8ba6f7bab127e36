// TEST INFRASTRUCTURE ONLY (oracle/): C-ABI shim around the REFERENCE's own CPU rotated-IoU
// (pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp:232-273), compiled from where it lies under /root/reference by
// oracle/Makefile into oracle/_ref/libiou3d_ref.so.  Nothing here is shipped or measured as product code;
// it exists so that the C restatement in oracle/iou3d_oracle.c can be pinned bit-for-bit against the real
// reference arithmetic and golden vectors can be generated (tests/golden/make_iou3d_golden.py).
#include <torch/extension.h>
#include <cstdint>

// declared in the reference's iou3d_cpu.h (pcdet/ops/iou3d_nms/src/iou3d_cpu.h:9-10)
int boxes_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);
int boxes_aligned_iou_bev_cpu(at::Tensor boxes_a_tensor, at::Tensor boxes_b_tensor, at::Tensor ans_iou_tensor);

extern "C" int ref_boxes_iou_bev_cpu(const float* a, int n, const float* b, int m, float* out) {
    auto opt = at::TensorOptions().dtype(at::kFloat);
    at::Tensor ta = at::from_blob(const_cast<float*>(a), {n, 7}, opt);
    at::Tensor tb = at::from_blob(const_cast<float*>(b), {m, 7}, opt);
    at::Tensor to = at::from_blob(out, {n, m}, opt);
    return boxes_iou_bev_cpu(ta, tb, to);
}

extern "C" int ref_boxes_aligned_iou_bev_cpu(const float* a, const float* b, int n, float* out) {
    auto opt = at::TensorOptions().dtype(at::kFloat);
    at::Tensor ta = at::from_blob(const_cast<float*>(a), {n, 7}, opt);
    at::Tensor tb = at::from_blob(const_cast<float*>(b), {n, 7}, opt);
    at::Tensor to = at::from_blob(out, {n, 1}, opt);
    return boxes_aligned_iou_bev_cpu(ta, tb, to);
}
