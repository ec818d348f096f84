// TEST INFRASTRUCTURE ONLY (oracle/): never imported by the product path.
//
// Thin extern "C" shim around the *unmodified* reference translation unit
// /root/reference/iou3d_nms/src/iou3d_cpu.cpp, compiled where it lies (see
// oracle/Makefile, target _ref/libiou3d_ref.so).  The reference TU is pulled in
// with #include so that its `inline` geometry helpers (box_overlap, iou_bev;
// iou3d_cpu.cpp:128-229) are reachable; nothing of it is copied into this repo.
//
// Headers it needs (<cuda.h>, <cuda_runtime_api.h>, torch) are the real ones
// shipped in this image (triton's bundled CUDA include dir, torch/include).
#include "iou3d_cpu.cpp"  // resolved through -I/root/reference/iou3d_nms/src

extern "C" {

// reference boxes_iou_bev_cpu (iou3d_cpu.cpp:232-252) driven through at::Tensor
int ref_boxes_iou_bev_cpu(const float* a, int n, const float* b, int m, float* out) {
    auto opt = torch::TensorOptions().dtype(torch::kFloat32);
    at::Tensor ta = torch::from_blob(const_cast<float*>(a), {n, 7}, opt);
    at::Tensor tb = torch::from_blob(const_cast<float*>(b), {m, 7}, opt);
    at::Tensor to = torch::from_blob(out, {n, m}, opt);
    return boxes_iou_bev_cpu(ta, tb, to);
}

// reference box_overlap (iou3d_cpu.cpp:128-220), same double loop as above
int ref_boxes_overlap_bev_cpu(const float* a, int n, const float* b, int m, float* out) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < m; j++)
            out[(long)i * m + j] = box_overlap(a + i * 7, b + j * 7);
    return 1;
}

}  // extern "C"
