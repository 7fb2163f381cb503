// ROIAlign forward / backward for gfx950 (NCHW fp32) -- the reference's only native op on the
// training path (common/lib/roi_pooling: vision.cpp:6-11, ROIAlign.h:11-45,
// cuda/ROIAlign_cuda.cu:15-122 forward, :125-254 backward, cpu/ROIAlign_cpu.cpp:17-219).
//
// Same arithmetic as the reference (no coordinate rounding, malformed RoIs forced to >= 1x1,
// samples outside [-1, size] contribute 0, bilinear weights with border clamping, average over the
// sampling grid), different mapping: the reference gives every output element (n,c,ph,pw) its own
// thread and recomputes the sample geometry per channel.  This is an HBM gather (4 loads + 1 store
// per output element), so here a workgroup owns (RoI, channel slice): each lane owns one output bin,
// computes that bin's sample positions / weights ONCE, then walks the channel planes -- per channel
// the wave issues 4 gathers from one small window of one H*W plane (L2/L1 friendly) and one fully
// coalesced store of the ph*pw plane.  Backward scatters with fp32 atomics into a zero-initialised
// grad_input exactly like the reference (:246-249).
#include "vlb_common.h"

#define ROI_MAX_HOIST 4  // sampling grids up to 2x2 keep their geometry in registers

struct RoiGeom {
  int p1, p2, p3, p4;  // plane offsets of the 4 neighbours (-1: sample outside -> contributes 0)
  float w1, w2, w3, w4;
};

__device__ __forceinline__ RoiGeom roi_sample(float y, float x, int height, int width) {
  RoiGeom g;
  if (y < -1.0f || y > (float)height || x < -1.0f || x > (float)width) {
    g.p1 = g.p2 = g.p3 = g.p4 = -1;
    g.w1 = g.w2 = g.w3 = g.w4 = 0.f;
    return g;
  }
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= height - 1) { y_high = y_low = height - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= width - 1) { x_high = x_low = width - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
  g.p1 = y_low * width + x_low; g.p2 = y_low * width + x_high;
  g.p3 = y_high * width + x_low; g.p4 = y_high * width + x_high;
  g.w1 = hy * hx; g.w2 = hy * lx; g.w3 = ly * hx; g.w4 = ly * lx;
  return g;
}

struct RoiBox {
  int batch;
  float start_w, start_h, bin_w, bin_h;
  int grid_h, grid_w;
};

__device__ __forceinline__ RoiBox roi_box(const float* __restrict__ roi, float scale, int ph_n, int pw_n, int sampling_ratio) {
  RoiBox r;
  r.batch = (int)roi[0];
  r.start_w = roi[1] * scale;
  r.start_h = roi[2] * scale;
  const float end_w = roi[3] * scale, end_h = roi[4] * scale;
  const float rw = fmaxf(end_w - r.start_w, 1.f), rh = fmaxf(end_h - r.start_h, 1.f);
  r.bin_h = rh / (float)ph_n;
  r.bin_w = rw / (float)pw_n;
  r.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph_n);
  r.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw_n);
  return r;
}

template <bool BWD>
__global__ __launch_bounds__(256) void roi_align_kernel(const float* __restrict__ src, const float* __restrict__ rois, float* __restrict__ dst,
                                                        int channels, int height, int width, int ph_n, int pw_n, float scale,
                                                        int sampling_ratio, int c_per_block) {
  // FWD: src = input [B,C,H,W], dst = output [K,C,ph,pw];  BWD: src = grad_output, dst = grad_input
  const int n = blockIdx.x;
  const int c_begin = blockIdx.y * c_per_block, c_end = min(channels, c_begin + c_per_block);
  const int bins = ph_n * pw_n;
  const RoiBox rb = roi_box(rois + (long)n * 5, scale, ph_n, pw_n, sampling_ratio);
  const float inv_count = 1.0f / (float)(rb.grid_h * rb.grid_w);
  const long plane = (long)height * width;
  const int nsamp = rb.grid_h * rb.grid_w;
  for (int bin = threadIdx.x; bin < bins; bin += 256) {
    const int ph = bin / pw_n, pw = bin % pw_n;
    if (nsamp <= ROI_MAX_HOIST) {
      RoiGeom g[ROI_MAX_HOIST];
#pragma unroll
      for (int s = 0; s < ROI_MAX_HOIST; ++s) {
        if (s < nsamp) {
          const int iy = s / rb.grid_w, ix = s % rb.grid_w;
          const float y = rb.start_h + ph * rb.bin_h + (iy + .5f) * rb.bin_h / (float)rb.grid_h;
          const float x = rb.start_w + pw * rb.bin_w + (ix + .5f) * rb.bin_w / (float)rb.grid_w;
          g[s] = roi_sample(y, x, height, width);
        } else {
          g[s].p1 = -1; g[s].p2 = g[s].p3 = g[s].p4 = -1; g[s].w1 = g[s].w2 = g[s].w3 = g[s].w4 = 0.f;
        }
      }
      for (int c = c_begin; c < c_end; ++c) {
        if (!BWD) {
          const float* pl = src + ((long)rb.batch * channels + c) * plane;
          float acc = 0.f;
#pragma unroll
          for (int s = 0; s < ROI_MAX_HOIST; ++s)
            if (s < nsamp && g[s].p1 >= 0)
              acc += g[s].w1 * pl[g[s].p1] + g[s].w2 * pl[g[s].p2] + g[s].w3 * pl[g[s].p3] + g[s].w4 * pl[g[s].p4];
          dst[((long)n * channels + c) * bins + bin] = acc * inv_count;
        } else {
          float* pl = dst + ((long)rb.batch * channels + c) * plane;
          const float go = src[((long)n * channels + c) * bins + bin];
#pragma unroll
          for (int s = 0; s < ROI_MAX_HOIST; ++s)
            if (s < nsamp && g[s].p1 >= 0) {
              atomicAdd(pl + g[s].p1, go * g[s].w1 * inv_count);
              atomicAdd(pl + g[s].p2, go * g[s].w2 * inv_count);
              atomicAdd(pl + g[s].p3, go * g[s].w3 * inv_count);
              atomicAdd(pl + g[s].p4, go * g[s].w4 * inv_count);
            }
        }
      }
    } else {  // large / adaptive sampling grids: geometry recomputed per channel
      for (int c = c_begin; c < c_end; ++c) {
        const long pbase = ((long)rb.batch * channels + c) * plane;
        const long obase = ((long)n * channels + c) * bins + bin;
        float acc = 0.f;
        const float go = BWD ? src[obase] : 0.f;
        for (int iy = 0; iy < rb.grid_h; ++iy) {
          const float y = rb.start_h + ph * rb.bin_h + (iy + .5f) * rb.bin_h / (float)rb.grid_h;
          for (int ix = 0; ix < rb.grid_w; ++ix) {
            const float x = rb.start_w + pw * rb.bin_w + (ix + .5f) * rb.bin_w / (float)rb.grid_w;
            const RoiGeom g = roi_sample(y, x, height, width);
            if (g.p1 < 0) continue;
            if (!BWD) {
              const float* pl = src + pbase;
              acc += g.w1 * pl[g.p1] + g.w2 * pl[g.p2] + g.w3 * pl[g.p3] + g.w4 * pl[g.p4];
            } else {
              float* pl = dst + pbase;
              atomicAdd(pl + g.p1, go * g.w1 * inv_count);
              atomicAdd(pl + g.p2, go * g.w2 * inv_count);
              atomicAdd(pl + g.p3, go * g.w3 * inv_count);
              atomicAdd(pl + g.p4, go * g.w4 * inv_count);
            }
          }
        }
        if (!BWD) dst[obase] = acc * inv_count;
      }
    }
  }
}

static int roi_cpb(int num_rois, int channels) {
  // enough (RoI, channel-slice) workgroups to cover the 256 CUs a few times over
  int slices = 1;
  while ((long)num_rois * slices < 2048 && slices < channels) slices *= 2;
  return vlb_cdiv(channels, slices);
}

extern "C" int vlb_roi_align_fwd(const float* input, const float* rois, float* output, int num_rois, int channels, int height,
                                 int width, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                 hipStream_t stream) {
  if (num_rois <= 0 || channels <= 0) return VLB_OK;  // empty input returns early (ROIAlign_cuda.cu:278-281)
  VLB_CHECK_ARG(input && rois && output, "vlb_roi_align_fwd: null argument");
  VLB_CHECK_ARG(height > 0 && width > 0 && pooled_h > 0 && pooled_w > 0, "vlb_roi_align_fwd: bad geometry");
  const int cpb = roi_cpb(num_rois, channels);
  hipLaunchKernelGGL(roi_align_kernel<false>, dim3(num_rois, vlb_cdiv(channels, cpb)), dim3(256), 0, stream, input, rois, output,
                     channels, height, width, pooled_h, pooled_w, spatial_scale, sampling_ratio, cpb);
  VLB_CHECK_LAUNCH("vlb_roi_align_fwd");
  return VLB_OK;
}

extern "C" int vlb_roi_align_bwd(const float* grad_output, const float* rois, float* grad_input, int num_rois, int batch,
                                 int channels, int height, int width, int pooled_h, int pooled_w, float spatial_scale,
                                 int sampling_ratio, hipStream_t stream) {
  VLB_CHECK_ARG(grad_input || batch * channels == 0, "vlb_roi_align_bwd: null grad_input");
  if ((long)batch * channels * height * width > 0)
    (void)hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)batch * channels * height * width, stream);
  if (num_rois <= 0 || channels <= 0) return VLB_OK;
  VLB_CHECK_ARG(grad_output && rois, "vlb_roi_align_bwd: null argument");
  const int cpb = roi_cpb(num_rois, channels);
  hipLaunchKernelGGL(roi_align_kernel<true>, dim3(num_rois, vlb_cdiv(channels, cpb)), dim3(256), 0, stream, grad_output, rois,
                     grad_input, channels, height, width, pooled_h, pooled_w, spatial_scale, sampling_ratio, cpb);
  VLB_CHECK_LAUNCH("vlb_roi_align_bwd");
  return VLB_OK;
}
