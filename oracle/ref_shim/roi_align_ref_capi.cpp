// Plain-C entry point around the REFERENCE's own ROIAlign_forward_cpu (common/lib/roi_pooling/cpu/ROIAlign_cpu.cpp:221-258,
// compiled from /root/reference by oracle/build_ref.sh) so that tests can call it through ctypes.
// TEST INFRASTRUCTURE ONLY: used to pin oracle/roi_align_oracle.py; never loaded by the product path.
#include <torch/extension.h>

at::Tensor ROIAlign_forward_cpu(const at::Tensor& input, const at::Tensor& rois, const float spatial_scale, const int pooled_height,
                                const int pooled_width, const int sampling_ratio);

extern "C" int ref_roi_align_forward_f32(const float* input, int B, int C, int H, int W, const float* rois, int K, float spatial_scale,
                                         int pooled_h, int pooled_w, int sampling_ratio, float* out) {
  try {
    auto opt = at::TensorOptions().dtype(at::kFloat);
    at::Tensor in = at::from_blob(const_cast<float*>(input), {B, C, H, W}, opt);
    at::Tensor r = at::from_blob(const_cast<float*>(rois), {K, 5}, opt);
    at::Tensor o = ROIAlign_forward_cpu(in, r, spatial_scale, pooled_h, pooled_w, sampling_ratio).contiguous();
    std::memcpy(out, o.data_ptr<float>(), sizeof(float) * (size_t)o.numel());
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "ref_roi_align_forward_f32: %s\n", e.what());
    return -1;
  }
}
