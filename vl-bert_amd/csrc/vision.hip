// End-to-end vision path for gfx950: the memory-bound kernels around the convolution GEMMs of the ResNet trunk and
// RoI head (common/backbone/resnet/resnet.py:75-199, common/fast_rcnn.py:55-100,144-156).
//
// Layout: activations are NHWC bf16 -- a feature map is a row-major [N*H*W, C] matrix, so a 1x1 convolution IS the
// MFMA NT GEMM of gemm.hip on the tensor as it lies (the reference's NCHW needs cuDNN's implicit transposes), a 3x3
// convolution is that GEMM over a gathered [rows, 9C] operand, frozen BatchNorm is folded into the bf16 working weights
// (scale) and the GEMM bias (shift), and ReLU / residual-add / ReLU-backward ride in the GEMM epilogues (act 2 / 7 / 8).
// What is left for this file is data movement, all of it 16-B per lane and coalesced along C:
//   vlb_conv_weight_prepare     fp32 master [O,KH,KW,I] x BN(gamma,beta,mean,var) -> bf16 forward operand [O,K],
//                               bf16 dgrad operand [I, mirrored taps, O], per-channel scale / shift
//   vlb_conv_wgrad_finalize     dW_master (+)= scale[o] * dW_folded
//   vlb_im2col_nhwc_bf16        [N,H,W,C] -> [N*OH*OW, KH*KW*C] (zero padding, stride, dilation)
//   vlb_im2col_image_f32        the 7x7/2 stem: fp32 NCHW image -> [N*OH*OW, 192] bf16 (147 taps x channels + pad)
//   vlb_maxpool3x3s2_nhwc       stem max-pool (forward only: stages 1-2 are frozen, resnet.py:223-233)
//   vlb_subsample2_nhwc / vlb_upsample2_zero_nhwc   the stride of the caffe-style stride-in-1x1 blocks (resnet.py:79)
//   vlb_roi_align_nhwc_fwd/bwd  ROIAlign (roi_align.py:11-44) on NHWC bf16 features; backward = fp32 atomics, dense per wave
//   vlb_relu_mask_cast          fp32 gradient -> bf16 where the saved activation > 0
//   vlb_avgpool_rows_fwd/bwd    AvgPool2d(14) + Flattener (common/fast_rcnn.py:80-84) into / out of the [B,R,4+2048] box rows
#include "vlb_common.h"

// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_weight_prepare_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const float* __restrict__ mean,
                                                                  const float* __restrict__ var, float eps, bf16_t* __restrict__ wf,
                                                                  bf16_t* __restrict__ wb, float* __restrict__ scale,
                                                                  float* __restrict__ shift, int O, int I, int T, int kf) {
  const long total = (long)O * T * I;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int i = (int)(idx % I);
    const int t = (int)((idx / I) % T);
    const int o = (int)(idx / ((long)I * T));
    const float s = gamma ? gamma[o] / sqrtf(var[o] + eps) : 1.0f;
    const bf16_t v = f2bf(w[idx] * s);
    wf[(long)o * kf + (long)t * I + i] = v;
    if (wb) wb[(long)i * T * O + (long)(T - 1 - t) * O + o] = v;
    if (i == 0 && t == 0) {
      if (scale) scale[o] = s;
      if (shift) shift[o] = gamma ? beta[o] - mean[o] * s : 0.f;
    }
  }
}

extern "C" int vlb_conv_weight_prepare(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                                       float eps, void* wf, void* wb, float* scale, float* shift, int O, int I, int taps, int kf,
                                       hipStream_t stream) {
  VLB_CHECK_ARG(w && wf, "vlb_conv_weight_prepare: null weight");
  VLB_CHECK_ARG(O > 0 && I > 0 && taps > 0 && kf >= taps * I, "vlb_conv_weight_prepare: bad shape O=%d I=%d taps=%d kf=%d", O, I, taps, kf);
  VLB_CHECK_ARG(!gamma || (beta && mean && var), "vlb_conv_weight_prepare: incomplete BatchNorm statistics");
  const long total = (long)O * taps * I;
  int blocks = vlb_cdiv(total, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv_weight_prepare_kernel, dim3(blocks), dim3(256), 0, stream, w, gamma, beta, mean, var, eps, (bf16_t*)wf,
                     (bf16_t*)wb, scale, shift, O, I, taps, kf);
  VLB_CHECK_LAUNCH("vlb_conv_weight_prepare");
  return VLB_OK;
}

// All convolutions of the network in ONE launch (93 trainable ones are re-folded after every optimizer step): desc[c] =
// {w, gamma, beta, mean, var, wf, wb, scale, shift, O, I, T, kf} as int64, block_start[c] = first 1024-element block of conv c.
__global__ __launch_bounds__(256) void conv_weight_prepare_batched_kernel(const int64_t* __restrict__ desc, const int32_t* __restrict__ block_start,
                                                                          int n, float eps) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {   // last conv whose first block is <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (block_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t* d = desc + (long)lo * 13;
  const float* w = (const float*)d[0];
  const float* gamma = (const float*)d[1];
  const float* beta = (const float*)d[2];
  const float* mean = (const float*)d[3];
  const float* var = (const float*)d[4];
  bf16_t* wf = (bf16_t*)d[5];
  bf16_t* wb = (bf16_t*)d[6];
  float* scale = (float*)d[7];
  float* shift = (float*)d[8];
  const int O = (int)d[9], I = (int)d[10], T = (int)d[11], kf = (int)d[12];
  const int lb = (int)blockIdx.x - block_start[lo];          // block index inside this convolution
  if ((O & 31) == 0 && (I & 31) == 0) {
    // 32(o) x 32(i) tile of one tap through LDS: w is read and wf written along i, wb (the transposed dgrad operand) is written
    // along o -- the straightforward mapping scattered 2-byte stores over wb (1.7 GB of write traffic for 85 MB of payload).
    __shared__ bf16_t tile[32][33];
    const int nib = I >> 5, nob = O >> 5;
    const int ib = lb % nib, ob = (lb / nib) % nob, t = lb / (nib * nob);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = ob * 32 + ty + 8 * r, i = ib * 32 + tx;
      const float sc = gamma ? gamma[o] / sqrtf(var[o] + eps) : 1.0f;
      const bf16_t v = f2bf(w[((long)o * T + t) * I + i] * sc);
      wf[(long)o * kf + (long)t * I + i] = v;
      tile[ty + 8 * r][tx] = v;
      if (ib == 0 && t == 0 && tx == 0) {
        if (scale) scale[o] = sc;
        if (shift) shift[o] = gamma ? beta[o] - mean[o] * sc : 0.f;
      }
    }
    if (wb) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib * 32 + ty + 8 * r, o = ob * 32 + tx;
        wb[(long)i * T * O + (long)(T - 1 - t) * O + o] = tile[tx][ty + 8 * r];
      }
    }
    return;
  }
  const long total = (long)O * T * I;
  const long base = (long)lb * 1024;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long idx = base + e * 256 + threadIdx.x;
    if (idx >= total) break;
    const int i = (int)(idx % I);
    const int t = (int)((idx / I) % T);
    const int o = (int)(idx / ((long)I * T));
    const float s = gamma ? gamma[o] / sqrtf(var[o] + eps) : 1.0f;
    const bf16_t v = f2bf(w[idx] * s);
    wf[(long)o * kf + (long)t * I + i] = v;
    if (wb) wb[(long)i * T * O + (long)(T - 1 - t) * O + o] = v;
    if (i == 0 && t == 0) {
      if (scale) scale[o] = s;
      if (shift) shift[o] = gamma ? beta[o] - mean[o] * s : 0.f;
    }
  }
}

extern "C" int vlb_conv_weight_prepare_batched(const int64_t* desc, const int32_t* block_start, int n, int total_blocks, float eps,
                                               hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0) return VLB_OK;
  VLB_CHECK_ARG(desc && block_start, "vlb_conv_weight_prepare_batched: null table");
  hipLaunchKernelGGL(conv_weight_prepare_batched_kernel, dim3(total_blocks), dim3(256), 0, stream, desc, block_start, n, eps);
  VLB_CHECK_LAUNCH("vlb_conv_weight_prepare_batched");
  return VLB_OK;
}

__global__ __launch_bounds__(256) void conv_wgrad_finalize_kernel(const float* __restrict__ dwf, const float* __restrict__ scale,
                                                                  float* __restrict__ g, int O, int kreal, int kf, int accumulate) {
  const long total = (long)O * kreal;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int o = (int)(idx / kreal), k = (int)(idx % kreal);
    const float v = dwf[(long)o * kf + k] * (scale ? scale[o] : 1.0f);
    g[idx] = accumulate ? g[idx] + v : v;
  }
}

extern "C" int vlb_conv_wgrad_finalize(const float* dwf, const float* scale, float* g, int O, int kreal, int kf, int accumulate,
                                       hipStream_t stream) {
  VLB_CHECK_ARG(dwf && g && O > 0 && kreal > 0 && kf >= kreal, "vlb_conv_wgrad_finalize: bad argument");
  int blocks = vlb_cdiv((long)O * kreal, 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(conv_wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, stream, dwf, scale, g, O, kreal, kf, accumulate);
  VLB_CHECK_LAUNCH("vlb_conv_wgrad_finalize");
  return VLB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// im2col: one 16-B chunk (8 channels of one tap of one output pixel) per thread
__global__ __launch_bounds__(256) void im2col_nhwc_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int N, int H, int W,
                                                          int C, int KH, int KW, int stride, int pad, int dil, int OH, int OW,
                                                          long ldcol) {
  const int c8n = C >> 3;
  const long per_row = (long)KH * KW * c8n;
  const long total = (long)N * OH * OW * per_row;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / per_row;
    const int q = (int)(idx % per_row);
    const int t = q / c8n, c8 = q % c8n;
    const int ky = t / KW, kx = t % KW;
    const int ox = (int)(row % OW);
    const int oy = (int)((row / OW) % OH);
    const int n = (int)(row / ((long)OW * OH));
    const int iy = oy * stride - pad + ky * dil, ix = ox * stride - pad + kx * dil;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *(const uint4*)(x + (((long)n * H + iy) * W + ix) * C + c8 * 8);
    *(uint4*)(col + row * ldcol + (long)t * C + c8 * 8) = v;
  }
}

extern "C" int vlb_im2col_nhwc_bf16(const void* x, void* col, long ldcol, int N, int H, int W, int C, int KH, int KW, int stride,
                                    int pad, int dil, hipStream_t stream) {
  if (N <= 0) return VLB_OK;
  VLB_CHECK_ARG(x && col, "vlb_im2col_nhwc_bf16: null argument");
  VLB_CHECK_ARG(C > 0 && (C % 8) == 0 && (ldcol % 8) == 0 && ldcol >= (long)KH * KW * C, "vlb_im2col_nhwc_bf16: C=%d ldcol=%ld", C, ldcol);
  VLB_CHECK_ARG(stride >= 1 && dil >= 1 && pad >= 0 && KH >= 1 && KW >= 1, "vlb_im2col_nhwc_bf16: bad geometry");
  const int OH = (H + 2 * pad - dil * (KH - 1) - 1) / stride + 1, OW = (W + 2 * pad - dil * (KW - 1) - 1) / stride + 1;
  VLB_CHECK_ARG(OH > 0 && OW > 0, "vlb_im2col_nhwc_bf16: empty output");
  const long total = (long)N * OH * OW * KH * KW * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(im2col_nhwc_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)col, N, H, W, C, KH, KW,
                     stride, pad, dil, OH, OW, ldcol);
  VLB_CHECK_LAUNCH("vlb_im2col_nhwc_bf16");
  return VLB_OK;
}

// stem: fp32 NCHW image, 7x7 stride 2 pad 3 (resnet.py:137-138); column k = (ky*7 + kx)*3 + c, zero from 147 up to ldcol
// STEM = true: the ResNet stem's geometry (3 channels, 7 x 7, stride 2, pad 3) as compile-time constants -- the per-element index
// arithmetic (k -> (ky, kx, c)) is then multiplies and shifts instead of five runtime integer divisions, which bound the kernel
template <bool STEM>
__global__ __launch_bounds__(256) void im2col_image_kernel(const float* __restrict__ img, bf16_t* __restrict__ col, int N, int Cin_, int H,
                                                           int W, int KH_, int KW_, int stride_, int pad_, int OH, int OW, int ldcol) {
  const int Cin = STEM ? 3 : Cin_, KH = STEM ? 7 : KH_, KW = STEM ? 7 : KW_, stride = STEM ? 2 : stride_, pad = STEM ? 3 : pad_;
  const int k8n = ldcol >> 3;
  const int kreal = KH * KW * Cin;
  const long total = (long)N * OH * OW * k8n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long row = idx / k8n;
    const int k0 = (int)(idx % k8n) * 8;
    const int ox = (int)(row % OW);
    const int oy = (int)((row / OW) % OH);
    const int n = (int)(row / ((long)OW * OH));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + e;
      v[e] = 0.f;
      if (k < kreal) {
        const int c = k % Cin, t = k / Cin, ky = t / KW, kx = t % KW;
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[e] = img[(((long)n * Cin + c) * H + iy) * W + ix];
      }
    }
    *(uint4*)(col + row * ldcol + k0) = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
  }
}

extern "C" int vlb_im2col_image_f32(const float* img, void* col, int ldcol, int N, int Cin, int H, int W, int KH, int KW, int stride,
                                    int pad, hipStream_t stream) {
  if (N <= 0) return VLB_OK;
  VLB_CHECK_ARG(img && col, "vlb_im2col_image_f32: null argument");
  VLB_CHECK_ARG((ldcol % 8) == 0 && ldcol >= KH * KW * Cin, "vlb_im2col_image_f32: ldcol=%d too small / unaligned", ldcol);
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  VLB_CHECK_ARG(OH > 0 && OW > 0, "vlb_im2col_image_f32: empty output");
  const long total = (long)N * OH * OW * (ldcol / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (Cin == 3 && KH == 7 && KW == 7 && stride == 2 && pad == 3)
    hipLaunchKernelGGL(im2col_image_kernel<true>, dim3((int)blocks), dim3(256), 0, stream, img, (bf16_t*)col, N, Cin, H, W, KH, KW, stride,
                       pad, OH, OW, ldcol);
  else
    hipLaunchKernelGGL(im2col_image_kernel<false>, dim3((int)blocks), dim3(256), 0, stream, img, (bf16_t*)col, N, Cin, H, W, KH, KW, stride,
                       pad, OH, OW, ldcol);
  VLB_CHECK_LAUNCH("vlb_im2col_image_f32");
  return VLB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void unpack8(const uint4& u, float (&f)[8]) {
  f[0] = bflo(u.x); f[1] = bfhi(u.x); f[2] = bflo(u.y); f[3] = bfhi(u.y);
  f[4] = bflo(u.z); f[5] = bfhi(u.z); f[6] = bflo(u.w); f[7] = bfhi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2bf(f[0], f[1]), pack2bf(f[2], f[3]), pack2bf(f[4], f[5]), pack2bf(f[6], f[7]));
}

// MaxPool2d(kernel 3, stride 2, padding 1) (resnet.py:141): padding never wins the max
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C,
                                                           int OH, int OW) {
  const int c8n = C >> 3;
  const long total = (long)N * OH * OW * c8n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n);
    const long row = idx / c8n;
    const int ox = (int)(row % OW);
    const int oy = (int)((row / OW) % OH);
    const int n = (int)(row / ((long)OW * OH));
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * 2 - 1 + ky;
      if (iy < 0 || iy >= H) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * 2 - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        float f[8];
        unpack8(*(const uint4*)(x + (((long)n * H + iy) * W + ix) * C + c8 * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], f[e]);
      }
    }
    *(uint4*)(y + row * C + c8 * 8) = pack8(m);
  }
}

// Raw-pixel masking of the masked regions (pretrain/data/datasets/conceptual_captions.py:201-206, coco_captions.py:240-244; e2e only,
// NETWORK.MASK_RAW_PIXELS): for every box with mvrc_op == 1,  image[:, int(y1):(int(y2)+1), int(x1):(int(x2)+1)] = 0  on the
// transformed (mean-subtracted) fp32 NCHW image, boxes in that image's pixel coordinates.  The reference does it per sample on the data
// worker; here one launch over the collated batch on the device (the image is already in HBM).  One workgroup per (box slot, sample),
// python slice semantics: negative starts count from the end, stops clamp to the extent.
__global__ __launch_bounds__(256) void mask_image_boxes_kernel(float* __restrict__ img, int C, int H, int W, const float* __restrict__ boxes,
                                                                long ldb, int R, const int64_t* __restrict__ ops) {
  const int r = blockIdx.x, b = blockIdx.y;
  if (ops[(long)b * R + r] != 1) return;
  const float* bx = boxes + ((long)b * R + r) * ldb;
  auto lo = [](float v, int n) { int i = (int)v; if (i < 0) i += n; return min(max(i, 0), n); };      // int(): truncation toward zero
  auto hi = [](float v, int n) { int i = (int)v + 1; if (i < 0) i += n; return min(max(i, 0), n); };
  const int x0 = lo(bx[0], W), y0 = lo(bx[1], H), x1 = hi(bx[2], W), y1 = hi(bx[3], H);
  const int w = x1 - x0, h = y1 - y0;
  if (w <= 0 || h <= 0) return;
  float* base = img + (long)b * C * H * W;
  const long n = (long)C * h * w;
  for (long i = threadIdx.x; i < n; i += 256) {
    const int x = (int)(i % w), y = (int)((i / w) % h), c = (int)(i / ((long)w * h));
    base[((long)c * H + y0 + y) * W + x0 + x] = 0.f;
  }
}

extern "C" int vlb_mask_image_boxes_f32(float* img, int N, int C, int H, int W, const float* boxes, long ldb, int R, const int64_t* mvrc_ops,
                                        hipStream_t stream) {
  if (N <= 0 || R <= 0) return VLB_OK;
  VLB_CHECK_ARG(img && boxes && mvrc_ops && C > 0 && H > 0 && W > 0 && ldb >= 4, "vlb_mask_image_boxes_f32: bad argument");
  hipLaunchKernelGGL(mask_image_boxes_kernel, dim3(R, N), dim3(256), 0, stream, img, C, H, W, boxes, ldb, R, mvrc_ops);
  VLB_CHECK_LAUNCH("vlb_mask_image_boxes_f32");
  return VLB_OK;
}

extern "C" int vlb_maxpool3x3s2_nhwc(const void* x, void* y, int N, int H, int W, int C, hipStream_t stream) {
  if (N <= 0) return VLB_OK;
  VLB_CHECK_ARG(x && y && C > 0 && (C % 8) == 0 && H > 0 && W > 0, "vlb_maxpool3x3s2_nhwc: bad argument");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * OH * OW * (C / 8);
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW);
  VLB_CHECK_LAUNCH("vlb_maxpool3x3s2_nhwc");
  return VLB_OK;
}

// y[n,oy,ox,:] = x[n,2oy,2ox,:]  (a 1x1 convolution with stride 2 reads exactly these pixels, resnet.py:79,158-160)
__global__ __launch_bounds__(256) void subsample2_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int N, int H, int W, int C, int OH,
                                                         int OW) {
  const int c8n = C >> 3;
  const long total = (long)N * OH * OW * c8n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n);
    const long row = idx / c8n;
    const int ox = (int)(row % OW);
    const int oy = (int)((row / OW) % OH);
    const int n = (int)(row / ((long)OW * OH));
    *(uint4*)(y + row * C + c8 * 8) = *(const uint4*)(x + (((long)n * H + 2 * oy) * W + 2 * ox) * C + c8 * 8);
  }
}

// dx[n,iy,ix,:] = dy[n,iy/2,ix/2,:] on even (iy,ix), 0 elsewhere (every element of dx is written)
__global__ __launch_bounds__(256) void upsample2_zero_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int N, int H, int W, int C,
                                                             int OH, int OW) {
  const int c8n = C >> 3;
  const long total = (long)N * H * W * c8n;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n);
    const long row = idx / c8n;
    const int ix = (int)(row % W);
    const int iy = (int)((row / W) % H);
    const int n = (int)(row / ((long)W * H));
    uint4 v = make_uint4(0, 0, 0, 0);
    if (((iy | ix) & 1) == 0) v = *(const uint4*)(dy + (((long)n * OH + (iy >> 1)) * OW + (ix >> 1)) * C + c8 * 8);
    *(uint4*)(dx + row * C + c8 * 8) = v;
  }
}

extern "C" int vlb_subsample2_nhwc(const void* x, void* y, int N, int H, int W, int C, hipStream_t stream) {
  if (N <= 0) return VLB_OK;
  VLB_CHECK_ARG(x && y && C > 0 && (C % 8) == 0 && H > 0 && W > 0, "vlb_subsample2_nhwc: bad argument");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  long blocks = ((long)N * OH * OW * (C / 8) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(subsample2_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW);
  VLB_CHECK_LAUNCH("vlb_subsample2_nhwc");
  return VLB_OK;
}

extern "C" int vlb_upsample2_zero_nhwc(const void* dy, void* dx, int N, int H, int W, int C, hipStream_t stream) {
  if (N <= 0) return VLB_OK;
  VLB_CHECK_ARG(dy && dx && C > 0 && (C % 8) == 0 && H > 0 && W > 0, "vlb_upsample2_zero_nhwc: bad argument");
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  long blocks = ((long)N * H * W * (C / 8) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(upsample2_zero_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, OH, OW);
  VLB_CHECK_LAUNCH("vlb_upsample2_zero_nhwc");
  return VLB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// ROIAlign on NHWC bf16 features.  Same sampling arithmetic as roi_align.hip (cuda/ROIAlign_cuda.cu:15-122,125-254):
// one WAVE per output bin, lanes across channels (16-B chunks), so the four neighbour reads of a sample are four
// contiguous C*2-byte rows and the backward atomics of a wave hit consecutive addresses.  RoIs come straight from the
// padded box rows of the batch: box k belongs to image k / boxes_per_image, coordinates boxes[k*ldbox + 0..3]
// (common/fast_rcnn.py:145-149 builds the same (batch_idx, x1, y1, x2, y2) rows); padded boxes (x1 <= -1.5,
// pretrain/data/collate_batch.py:39) produce zeros forward and receive no gradient.
// ------------------------------------------------------------------------------------------------------------------
struct NhwcSample {
  long p1, p2, p3, p4;   // pixel indices inside the image plane, -1: outside
  float w1, w2, w3, w4;
};

__device__ __forceinline__ NhwcSample nhwc_sample(float y, float x, int height, int width) {
  NhwcSample g;
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
  g.p1 = (long)y_low * width + x_low; g.p2 = (long)y_low * width + x_high;
  g.p3 = (long)y_high * width + x_low; g.p4 = (long)y_high * width + x_high;
  g.w1 = hy * hx; g.w2 = hy * lx; g.w3 = ly * hx; g.w4 = ly * lx;
  return g;
}

template <bool BWD>
__global__ __launch_bounds__(256) void roi_align_nhwc_kernel(const bf16_t* __restrict__ feat, float* __restrict__ dfeat,
                                                             const float* __restrict__ boxes, long ldbox, int boxes_per_image,
                                                             bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, int K, int C, int H,
                                                             int W, int ph_n, int pw_n, float scale, int sampling_ratio) {
  const int lane = threadIdx.x & 63;
  const int bins = ph_n * pw_n;
  const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= (long)K * bins) return;
  const int k = (int)(gw / bins), bin = (int)(gw % bins);
  const int ph = bin / pw_n, pw = bin % pw_n;
  const float* bx = boxes + (long)k * ldbox;
  const int batch = k / boxes_per_image;
  const int c8n = C >> 3;
  if (!(bx[0] > -1.5f)) {   // padded box
    if (!BWD)
      for (int c8 = lane; c8 < c8n; c8 += 64) *(uint4*)(out + gw * C + c8 * 8) = make_uint4(0, 0, 0, 0);
    return;
  }
  const float start_w = bx[0] * scale, start_h = bx[1] * scale, end_w = bx[2] * scale, end_h = bx[3] * scale;
  const float rw = fmaxf(end_w - start_w, 1.f), rh = fmaxf(end_h - start_h, 1.f);
  const float bin_h = rh / (float)ph_n, bin_w = rw / (float)pw_n;
  const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph_n);
  const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw_n);
  const float inv_count = 1.0f / (float)(grid_h * grid_w);
  const long plane = (long)H * W;
  for (int c8 = lane; c8 < c8n; c8 += 64) {
    float acc[8];
    if (BWD) {
      unpack8(*(const uint4*)(dout + gw * C + c8 * 8), acc);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= inv_count;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    }
    for (int iy = 0; iy < grid_h; ++iy) {
      const float y = start_h + ph * bin_h + (iy + .5f) * bin_h / (float)grid_h;
      for (int ix = 0; ix < grid_w; ++ix) {
        const float x = start_w + pw * bin_w + (ix + .5f) * bin_w / (float)grid_w;
        const NhwcSample g = nhwc_sample(y, x, H, W);
        if (g.p1 < 0) continue;
        const long base = (long)batch * plane;
        if (!BWD) {
          float a[8], b[8], c[8], d[8];
          unpack8(*(const uint4*)(feat + (base + g.p1) * C + c8 * 8), a);
          unpack8(*(const uint4*)(feat + (base + g.p2) * C + c8 * 8), b);
          unpack8(*(const uint4*)(feat + (base + g.p3) * C + c8 * 8), c);
          unpack8(*(const uint4*)(feat + (base + g.p4) * C + c8 * 8), d);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[e] += g.w1 * a[e] + g.w2 * b[e] + g.w3 * c[e] + g.w4 * d[e];
        } else {
          float* q1 = dfeat + (base + g.p1) * C + c8 * 8;
          float* q2 = dfeat + (base + g.p2) * C + c8 * 8;
          float* q3 = dfeat + (base + g.p3) * C + c8 * 8;
          float* q4 = dfeat + (base + g.p4) * C + c8 * 8;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            atomicAdd(q1 + e, acc[e] * g.w1);
            atomicAdd(q2 + e, acc[e] * g.w2);
            atomicAdd(q3 + e, acc[e] * g.w3);
            atomicAdd(q4 + e, acc[e] * g.w4);
          }
        }
      }
    }
    if (!BWD) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] *= inv_count;
      *(uint4*)(out + gw * C + c8 * 8) = pack8(acc);
    }
  }
}


// Backward with the RoI's feature window staged in LDS.  A 14x14 bin grid over an RoI that is a few feature pixels wide
// sends most of its 196 x 4 bilinear contributions per channel to the SAME handful of pixels: straight global atomics
// serialise on those addresses (5.8 ms for 288 RoIs x 1024 channels).  Here a workgroup owns (RoI, 32-channel slice),
// accumulates the slice of the RoI's window [wh x ww pixels][32 ch] with LDS atomics (lanes = channels: conflict-free
// within a pixel) and flushes every window element with ONE global atomic, 128 B per pixel row.  Windows that do not
// fit the LDS budget (whole-image boxes) fall back to direct atomics -- their contributions are spread out anyway.
#define ROI_BWD_CC 32
#define ROI_BWD_LDS_FLOATS 12288   // 48 KiB
__global__ __launch_bounds__(256) void roi_align_nhwc_bwd_lds_kernel(const bf16_t* __restrict__ dout, const float* __restrict__ boxes,
                                                                     long ldbox, int boxes_per_image, float* __restrict__ dfeat, int C,
                                                                     int H, int W, int ph_n, int pw_n, float scale, int sampling_ratio) {
  __shared__ float win[ROI_BWD_LDS_FLOATS];
  const int k = blockIdx.x;
  const int c0 = blockIdx.y * ROI_BWD_CC;
  const float* bx = boxes + (long)k * ldbox;
  if (!(bx[0] > -1.5f)) return;   // padded box: no gradient
  const int batch = k / boxes_per_image;
  const int bins = ph_n * pw_n;
  const float start_w = bx[0] * scale, start_h = bx[1] * scale, end_w = bx[2] * scale, end_h = bx[3] * scale;
  const float rw = fmaxf(end_w - start_w, 1.f), rh = fmaxf(end_h - start_h, 1.f);
  const float bin_h = rh / (float)ph_n, bin_w = rw / (float)pw_n;
  const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph_n);
  const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw_n);
  const float inv_count = 1.0f / (float)(grid_h * grid_w);
  // pixel window that any in-range sample of this RoI can touch
  const int ymin = max(0, (int)floorf(start_h)), ymax = min(H - 1, (int)floorf(start_h + rh) + 1);
  const int xmin = max(0, (int)floorf(start_w)), xmax = min(W - 1, (int)floorf(start_w + rw) + 1);
  if (ymin > ymax || xmin > xmax) return;   // RoI entirely outside the map: every sample is skipped
  const int wh = ymax - ymin + 1, ww = xmax - xmin + 1;
  const bool staged = (long)wh * ww * ROI_BWD_CC <= ROI_BWD_LDS_FLOATS;   // block-uniform
  if (staged) {
    for (int i = threadIdx.x; i < wh * ww * ROI_BWD_CC; i += 256) win[i] = 0.f;
    __syncthreads();
  }
  const int c = threadIdx.x & (ROI_BWD_CC - 1);
  const long plane0 = (long)batch * H * W;
  for (int bin = threadIdx.x / ROI_BWD_CC; bin < bins; bin += 256 / ROI_BWD_CC) {
    const int ph = bin / pw_n, pw = bin % pw_n;
    const float go = bf2f(dout[((long)k * bins + bin) * C + c0 + c]) * inv_count;
    for (int iy = 0; iy < grid_h; ++iy) {
      const float y = start_h + ph * bin_h + (iy + .5f) * bin_h / (float)grid_h;
      for (int ix = 0; ix < grid_w; ++ix) {
        const float x = start_w + pw * bin_w + (ix + .5f) * bin_w / (float)grid_w;
        const NhwcSample g = nhwc_sample(y, x, H, W);
        if (g.p1 < 0) continue;
        if (staged) {
          const int y1 = (int)(g.p1 / W) - ymin, x1 = (int)(g.p1 % W) - xmin, y4 = (int)(g.p4 / W) - ymin, x4 = (int)(g.p4 % W) - xmin;
          atomicAdd(&win[(y1 * ww + x1) * ROI_BWD_CC + c], go * g.w1);
          atomicAdd(&win[(y1 * ww + x4) * ROI_BWD_CC + c], go * g.w2);
          atomicAdd(&win[(y4 * ww + x1) * ROI_BWD_CC + c], go * g.w3);
          atomicAdd(&win[(y4 * ww + x4) * ROI_BWD_CC + c], go * g.w4);
        } else {
          atomicAdd(dfeat + (plane0 + g.p1) * C + c0 + c, go * g.w1);
          atomicAdd(dfeat + (plane0 + g.p2) * C + c0 + c, go * g.w2);
          atomicAdd(dfeat + (plane0 + g.p3) * C + c0 + c, go * g.w3);
          atomicAdd(dfeat + (plane0 + g.p4) * C + c0 + c, go * g.w4);
        }
      }
    }
  }
  if (!staged) return;
  __syncthreads();
  for (int i = threadIdx.x; i < wh * ww * ROI_BWD_CC; i += 256) {
    const float v = win[i];
    if (v == 0.f) continue;
    const int pix = i / ROI_BWD_CC, cc = i % ROI_BWD_CC;
    const int py = ymin + pix / ww, px = xmin + pix % ww;
    atomicAdd(dfeat + (plane0 + (long)py * W + px) * C + c0 + cc, v);
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Backward as a GATHER over the feature map (the path the engine takes).  The bilinear weight of sample (iy, ix) of bin (ph, pw) on
// pixel (py, px) is a product of a y-factor and an x-factor, the out-of-range test is an OR of a y- and an x-test, and the sum over
// the sampling grid factorises, so
//     dfeat[n, py, px, c] = sum_k  sum_ph Wy_k[ph, py]  sum_pw Wx_k[pw, px]  dout[k, ph, pw, c]
// with two small dense matrices per RoI (Wy: [bins_h, H], Wx: [bins_w, W], 1 / grid folded in).  Pass 1 (one workgroup per RoI)
// builds them from the SAME sampling arithmetic as the forward, plus per (RoI, row) / (RoI, column) the contiguous range of bins
// with a non-zero weight (sample coordinates are monotonic in the bin index).  Pass 2: one workgroup per feature pixel, lanes over
// channels (8 B each); lane j first fetches the ranges of RoI j, a ballot gives the RoIs that touch the pixel, and the wave walks
// only those, 4 independent 8-B loads per step.  Every output element has exactly one owner: no atomics, no memset, and the
// consumer's ReLU mask + bf16 cast (vlb_relu_mask_cast) is folded into the store.  288 RoIs x 14 x 14 x 1024 onto 8 x 38 x 63:
// 1.21 ms (LDS-window atomics + memset + mask pass) -> see DESIGN.md.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void roi_axis_sample(float v, int size, int& lo, int& hi, float& wl, float& wh) {
  // one axis of nhwc_sample(): lo / hi pixel and their weights; out of range -> weights 0
  if (v < -1.0f || v > (float)size) { lo = hi = 0; wl = wh = 0.f; return; }
  if (v <= 0.f) v = 0.f;
  lo = (int)v;
  if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
  wh = v - (float)lo;
  wl = 1.f - wh;
}

__global__ __launch_bounds__(128) void roi_axis_tables_kernel(const float* __restrict__ boxes, long ldbox, int H, int W, int ph_n, int pw_n,
                                                              float scale, int sampling_ratio, float* __restrict__ Wy,
                                                              float* __restrict__ Wx, unsigned short* __restrict__ yr,
                                                              unsigned short* __restrict__ xr) {
  const int k = blockIdx.x, tid = threadIdx.x;
  float* wy = Wy + (long)k * ph_n * H;
  float* wx = Wx + (long)k * pw_n * W;
  for (int i = tid; i < ph_n * H; i += 128) wy[i] = 0.f;
  for (int i = tid; i < pw_n * W; i += 128) wx[i] = 0.f;
  __syncthreads();
  const float* bx = boxes + (long)k * ldbox;
  if (bx[0] > -1.5f) {
    const float start_w = bx[0] * scale, start_h = bx[1] * scale, end_w = bx[2] * scale, end_h = bx[3] * scale;
    const float rw = fmaxf(end_w - start_w, 1.f), rh = fmaxf(end_h - start_h, 1.f);
    const float bin_h = rh / (float)ph_n, bin_w = rw / (float)pw_n;
    const int grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)ph_n);
    const int grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)pw_n);
    for (int t = tid; t < ph_n + pw_n; t += 128) {       // one thread per bin row / bin column: no races on its table row
      const bool isy = t < ph_n;
      const int p = isy ? t : t - ph_n, grid = isy ? grid_h : grid_w, size = isy ? H : W;
      const float start = isy ? start_h : start_w, bin = isy ? bin_h : bin_w, inv = 1.0f / (float)grid;
      float* row = isy ? wy + (long)p * H : wx + (long)p * W;
      for (int i = 0; i < grid; ++i) {
        const float v = start + p * bin + (i + .5f) * bin / (float)grid;
        int lo, hi; float wl, wh;
        roi_axis_sample(v, size, lo, hi, wl, wh);
        row[lo] += wl * inv;
        row[hi] += wh * inv;
      }
    }
  }
  __syncthreads();
  for (int t = tid; t < H + W; t += 128) {
    const bool isy = t < H;
    const int q = isy ? t : t - H, nb = isy ? ph_n : pw_n, size = isy ? H : W;
    const float* tab = isy ? wy : wx;
    int lo = 1, hi = 0;
    bool any = false;
    for (int p = 0; p < nb; ++p)
      if (tab[(long)p * size + q] != 0.f) { if (!any) lo = p; hi = p; any = true; }
    (isy ? yr + (long)k * H : xr + (long)k * W)[q] = (unsigned short)(lo | (hi << 8));
  }
}

__device__ __forceinline__ void fma4_bf16(float (&acc)[4], float w, const uint2& v) {
  acc[0] += w * __uint_as_float(v.x << 16);
  acc[1] += w * __uint_as_float(v.x & 0xFFFF0000u);
  acc[2] += w * __uint_as_float(v.y << 16);
  acc[3] += w * __uint_as_float(v.y & 0xFFFF0000u);
}

__global__ __launch_bounds__(256) void roi_align_nhwc_bwd_gather_kernel(const bf16_t* __restrict__ dout, const float* __restrict__ Wy,
                                                                        const float* __restrict__ Wx,
                                                                        const unsigned short* __restrict__ yr,
                                                                        const unsigned short* __restrict__ xr,
                                                                        const bf16_t* __restrict__ act, bf16_t* __restrict__ out_bf,
                                                                        float* __restrict__ out_f32, int R, int C, int H, int W, int ph_n,
                                                                        int pw_n) {
  const int pix = blockIdx.x;                        // (n, py, px)
  const int n = pix / (H * W), py = (pix / W) % H, px = pix % W;
  const int lane = threadIdx.x & 63;
  const int c4n = C >> 2;
  const long bins = (long)ph_n * pw_n;
  for (int c4b = (threadIdx.x & ~63); c4b < c4n; c4b += 256) {      // wave-uniform trip count: every lane takes part in the
    const bool live = c4b + lane < c4n;                              // range loads / ballot below, also when C / 4 is not a
    const int c4 = live ? c4b + lane : c4n - 1;                      // multiple of 64 (idle lanes shadow the last chunk)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < R; k0 += 64) {
      unsigned yv = 1u, xv = 1u;                     // (lo = 1, hi = 0): empty
      if (k0 + lane < R) {
        yv = yr[((long)n * R + k0 + lane) * H + py];
        xv = xr[((long)n * R + k0 + lane) * W + px];
      }
      unsigned long long m = __ballot((yv & 255u) <= (yv >> 8) && (xv & 255u) <= (xv >> 8));
      while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        const unsigned ys = __builtin_amdgcn_readlane(yv, j), xs = __builtin_amdgcn_readlane(xv, j);
        const int ylo = ys & 255u, yhi = ys >> 8, xlo = xs & 255u, xhi = xs >> 8;
        const long k = (long)n * R + k0 + j;
        const float* wyk = Wy + k * ph_n * H + py;
        const float* wxk = Wx + k * pw_n * W + px;
        const bf16_t* dk = dout + k * bins * C + (long)c4 * 4;
        for (int ph = ylo; ph <= yhi; ph += 2) {
          const int ph1 = min(ph + 1, yhi);
          const float wy0 = wyk[(long)ph * H], wy1 = (ph + 1 <= yhi) ? wyk[(long)ph1 * H] : 0.f;
          for (int pw = xlo; pw <= xhi; pw += 2) {
            const int pw1 = min(pw + 1, xhi);
            const float wx0 = wxk[(long)pw * W], wx1 = (pw + 1 <= xhi) ? wxk[(long)pw1 * W] : 0.f;
            const uint2 v00 = *(const uint2*)(dk + ((long)ph * pw_n + pw) * C);
            const uint2 v01 = *(const uint2*)(dk + ((long)ph * pw_n + pw1) * C);
            const uint2 v10 = *(const uint2*)(dk + ((long)ph1 * pw_n + pw) * C);
            const uint2 v11 = *(const uint2*)(dk + ((long)ph1 * pw_n + pw1) * C);
            fma4_bf16(acc, wy0 * wx0, v00);
            fma4_bf16(acc, wy0 * wx1, v01);
            fma4_bf16(acc, wy1 * wx0, v10);
            fma4_bf16(acc, wy1 * wx1, v11);
          }
        }
      }
    }
    if (!live) continue;
    const long o = (long)pix * C + (long)c4 * 4;
    if (out_f32) *(float4*)(out_f32 + o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    if (out_bf) {
      if (act) {                                      // gradient through the ReLU that produced `act` (vlb_relu_mask_cast)
        const uint2 a = *(const uint2*)(act + o);
        if (!(__uint_as_float(a.x << 16) > 0.f)) acc[0] = 0.f;
        if (!(__uint_as_float(a.x & 0xFFFF0000u) > 0.f)) acc[1] = 0.f;
        if (!(__uint_as_float(a.y << 16) > 0.f)) acc[2] = 0.f;
        if (!(__uint_as_float(a.y & 0xFFFF0000u) > 0.f)) acc[3] = 0.f;
      }
      uint2 r;
      r.x = (unsigned)f2bf(acc[0]) | ((unsigned)f2bf(acc[1]) << 16);
      r.y = (unsigned)f2bf(acc[2]) | ((unsigned)f2bf(acc[3]) << 16);
      *(uint2*)(out_bf + o) = r;
    }
  }
}

// bytes of table workspace vlb_roi_align_nhwc_bwd_gather needs for K RoIs
extern "C" long vlb_roi_align_gather_workspace_bytes(int K, int H, int W, int pooled_h, int pooled_w) {
  const long f = (long)K * ((long)pooled_h * H + (long)pooled_w * W) * 4;
  const long r = (long)K * (H + W) * 2;
  return (f + 15) / 16 * 16 + (r + 15) / 16 * 16;
}

// dx_bf16 [N,H,W,C] (and / or dx_f32) = ROIAlign backward of dout [K,ph,pw,C]; act (optional, bf16 [N,H,W,C]): dx_bf16 is zeroed
// where act <= 0.  K = N * boxes_per_image box rows.  Every element of the outputs is written (no accumulation, no zero-fill needed).
extern "C" int vlb_roi_align_nhwc_bwd_gather(const void* dout, const float* boxes, long ldbox, int boxes_per_image, const void* act,
                                             void* dx_bf16, float* dx_f32, void* workspace, long workspace_bytes, int N, int C, int H,
                                             int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                             hipStream_t stream) {
  if ((long)N * C * H * W <= 0) return VLB_OK;
  const int K = N * boxes_per_image;
  VLB_CHECK_ARG(dout && boxes && (dx_bf16 || dx_f32) && workspace, "vlb_roi_align_nhwc_bwd_gather: null argument");
  VLB_CHECK_ARG(C > 0 && (C % 4) == 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && pooled_h < 256 && pooled_w < 256 &&
                boxes_per_image > 0 && ldbox >= 4, "vlb_roi_align_nhwc_bwd_gather: bad geometry");
  VLB_CHECK_ARG(workspace_bytes >= vlb_roi_align_gather_workspace_bytes(K, H, W, pooled_h, pooled_w),
                "vlb_roi_align_nhwc_bwd_gather: workspace too small");
  VLB_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "vlb_roi_align_nhwc_bwd_gather: workspace must be 16-byte aligned");
  float* Wy = (float*)workspace;
  float* Wx = Wy + (long)K * pooled_h * H;
  const long f = ((long)K * ((long)pooled_h * H + (long)pooled_w * W) * 4 + 15) / 16 * 16;
  unsigned short* yr = (unsigned short*)((char*)workspace + f);
  unsigned short* xr = yr + (long)K * H;
  hipLaunchKernelGGL(roi_axis_tables_kernel, dim3(K), dim3(128), 0, stream, boxes, ldbox, H, W, pooled_h, pooled_w, spatial_scale,
                     sampling_ratio, Wy, Wx, yr, xr);
  VLB_CHECK_LAUNCH("vlb_roi_align_nhwc_bwd_gather(tables)");
  hipLaunchKernelGGL(roi_align_nhwc_bwd_gather_kernel, dim3(N * H * W), dim3(256), 0, stream, (const bf16_t*)dout, Wy, Wx, yr, xr,
                     (const bf16_t*)act, (bf16_t*)dx_bf16, dx_f32, boxes_per_image, C, H, W, pooled_h, pooled_w);
  VLB_CHECK_LAUNCH("vlb_roi_align_nhwc_bwd_gather");
  return VLB_OK;
}

extern "C" int vlb_roi_align_nhwc_fwd(const void* feat, const float* boxes, long ldbox, int boxes_per_image, void* out, int K, int C,
                                      int H, int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                      hipStream_t stream) {
  if (K <= 0) return VLB_OK;
  VLB_CHECK_ARG(feat && boxes && out, "vlb_roi_align_nhwc_fwd: null argument");
  VLB_CHECK_ARG(C > 0 && (C % 8) == 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && boxes_per_image > 0 && ldbox >= 4,
                "vlb_roi_align_nhwc_fwd: bad geometry");
  const long waves = (long)K * pooled_h * pooled_w;
  hipLaunchKernelGGL(roi_align_nhwc_kernel<false>, dim3(vlb_cdiv(waves, 4)), dim3(256), 0, stream, (const bf16_t*)feat, (float*)nullptr,
                     boxes, ldbox, boxes_per_image, (bf16_t*)out, (const bf16_t*)nullptr, K, C, H, W, pooled_h, pooled_w, spatial_scale,
                     sampling_ratio);
  VLB_CHECK_LAUNCH("vlb_roi_align_nhwc_fwd");
  return VLB_OK;
}

// dfeat (fp32 [N,H,W,C]) is zeroed here, like the reference's at::zeros grad_input (ROIAlign_cuda.cu:316)
extern "C" int vlb_roi_align_nhwc_bwd(const void* dout, const float* boxes, long ldbox, int boxes_per_image, float* dfeat, int K, int N,
                                      int C, int H, int W, int pooled_h, int pooled_w, float spatial_scale, int sampling_ratio,
                                      hipStream_t stream) {
  VLB_CHECK_ARG(dfeat || (long)N * C == 0, "vlb_roi_align_nhwc_bwd: null dfeat");
  if ((long)N * C * H * W > 0) (void)hipMemsetAsync(dfeat, 0, sizeof(float) * (size_t)N * C * H * W, stream);
  if (K <= 0) return VLB_OK;
  VLB_CHECK_ARG(dout && boxes, "vlb_roi_align_nhwc_bwd: null argument");
  VLB_CHECK_ARG(C > 0 && (C % 8) == 0 && H > 0 && W > 0 && pooled_h > 0 && pooled_w > 0 && boxes_per_image > 0 && ldbox >= 4,
                "vlb_roi_align_nhwc_bwd: bad geometry");
  if ((C % ROI_BWD_CC) == 0) {
    hipLaunchKernelGGL(roi_align_nhwc_bwd_lds_kernel, dim3(K, C / ROI_BWD_CC), dim3(256), 0, stream, (const bf16_t*)dout, boxes, ldbox,
                       boxes_per_image, dfeat, C, H, W, pooled_h, pooled_w, spatial_scale, sampling_ratio);
  } else {
    const long waves = (long)K * pooled_h * pooled_w;
    hipLaunchKernelGGL(roi_align_nhwc_kernel<true>, dim3(vlb_cdiv(waves, 4)), dim3(256), 0, stream, (const bf16_t*)nullptr, dfeat, boxes,
                       ldbox, boxes_per_image, (bf16_t*)nullptr, (const bf16_t*)dout, K, C, H, W, pooled_h, pooled_w, spatial_scale,
                       sampling_ratio);
  }
  VLB_CHECK_LAUNCH("vlb_roi_align_nhwc_bwd");
  return VLB_OK;
}

// dz = g where y > 0 else 0  (fp32 gradient of a post-ReLU activation -> bf16 gradient of its pre-activation)
__global__ __launch_bounds__(256) void relu_mask_cast_kernel(const float* __restrict__ g, const bf16_t* __restrict__ y, bf16_t* __restrict__ dz,
                                                             long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const float4 a = *(const float4*)(g + i * 8), b = *(const float4*)(g + i * 8 + 4);
    float f[8];
    unpack8(*(const uint4*)(y + i * 8), f);
    const float v[8] = {f[0] > 0.f ? a.x : 0.f, f[1] > 0.f ? a.y : 0.f, f[2] > 0.f ? a.z : 0.f, f[3] > 0.f ? a.w : 0.f,
                        f[4] > 0.f ? b.x : 0.f, f[5] > 0.f ? b.y : 0.f, f[6] > 0.f ? b.z : 0.f, f[7] > 0.f ? b.w : 0.f};
    *(uint4*)(dz + i * 8) = pack8(v);
  }
}

extern "C" int vlb_relu_mask_cast(const float* g, const void* y, void* dz, long n, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(g && y && dz && (n % 8) == 0, "vlb_relu_mask_cast: null argument or n=%ld not a multiple of 8", n);
  long blocks = (n / 8 + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(relu_mask_cast_kernel, dim3((int)blocks), dim3(256), 0, stream, g, (const bf16_t*)y, (bf16_t*)dz, n / 8);
  VLB_CHECK_LAUNCH("vlb_relu_mask_cast");
  return VLB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// AvgPool2d(14) + Flattener over the RoI head output y [K, P, C] (P = 14*14 pixels) written as fp32 into the feature
// slots of the padded box rows (row k, columns col0 .. col0+C of a [K, ld] fp32 matrix: boxes[..., 4:]) -- the input the
// precomputed-feature path reads, so everything downstream (obj_prep, obj_downsample) is shared.
// ------------------------------------------------------------------------------------------------------------------
// segm (optional, fp32 [K, P]): the per-pixel object mask VCR multiplies the RoI-head output with before pooling
// (common/fast_rcnn.py:152-156): out = mean_p( y[k,p,:] * segm[k,p] ).
__global__ __launch_bounds__(256) void avgpool_rows_fwd_kernel(const bf16_t* __restrict__ y, float* __restrict__ out, long ld, int col0, int P,
                                                               int C, int pad_col, const float* __restrict__ segm) {
  const int k = blockIdx.x;
  const int c8n = C >> 3;
  // padded box (x1 <= -1.5 in column pad_col of the same row): zero features, like the reference's zero-padded obj_reps_raw
  const float inv = (pad_col >= 0 && !(out[(long)k * ld + pad_col] > -1.5f)) ? 0.f : 1.0f / (float)P;
  for (int c8 = threadIdx.x; c8 < c8n; c8 += 256) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16_t* src = y + (long)k * P * C + c8 * 8;
    for (int p = 0; p < P; ++p) {
      float f[8];
      unpack8(*(const uint4*)(src + (long)p * C), f);
      const float m = segm ? segm[(long)k * P + p] : 1.0f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += f[e] * m;
    }
    float* o = out + (long)k * ld + col0 + c8 * 8;
    *(float4*)o = make_float4(s[0] * inv, s[1] * inv, s[2] * inv, s[3] * inv);
    *(float4*)(o + 4) = make_float4(s[4] * inv, s[5] * inv, s[6] * inv, s[7] * inv);
  }
}

extern "C" int vlb_avgpool_rows_fwd(const void* y, float* out, long ld, int col0, int pad_col, int K, int P, int C, const float* segm,
                                    hipStream_t stream) {
  if (K <= 0) return VLB_OK;
  VLB_CHECK_ARG(y && out && P > 0 && C > 0 && (C % 8) == 0, "vlb_avgpool_rows_fwd: bad argument");
  VLB_CHECK_ARG((ld % 4) == 0 && (col0 % 4) == 0 && ld >= col0 + C, "vlb_avgpool_rows_fwd: ld=%ld col0=%d must be multiples of 4", ld, col0);
  VLB_CHECK_ARG(pad_col < col0, "vlb_avgpool_rows_fwd: pad_col=%d must lie in front of the feature columns", pad_col);
  hipLaunchKernelGGL(avgpool_rows_fwd_kernel, dim3(K), dim3(256), 0, stream, (const bf16_t*)y, out, ld, col0, P, C, pad_col, segm);
  VLB_CHECK_LAUNCH("vlb_avgpool_rows_fwd");
  return VLB_OK;
}

// backward of [ReLU ->] AvgPool -> (input dropout of obj_downsample, common/fast_rcnn.py:106):
//   dz[k,p,c] = (y[k,p,c] > 0) * keep(k, drop_col0 + c) * drop_scale * dfeat[k,c] / P ;  boxes with x1 <= -1.5 (padding) get 0.
// keep() is the counter hash obj_prep_fwd used for element (row k, column drop_col0 + c) of its [K, drop_row_elems] output.
__global__ __launch_bounds__(256) void avgpool_rows_bwd_kernel(const bf16_t* __restrict__ dfeat, long lddf, const bf16_t* __restrict__ y,
                                                               const float* __restrict__ boxes, long ldbox, bf16_t* __restrict__ dz, int K,
                                                               int P, int C, uint32_t drop_thr, float drop_scale,
                                                               const uint32_t* __restrict__ seedp, uint32_t tag, uint32_t drop_row_elems,
                                                               uint32_t drop_col0, const float* __restrict__ segm) {
  const int c8n = C >> 3;
  const long total = (long)K * P * c8n;
  const uint32_t seed = (drop_thr && seedp) ? *seedp : 0u;
  const float inv = 1.0f / (float)P;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c8 = (int)(idx % c8n);
    const long row = idx / c8n;          // k*P + p
    const int k = (int)(row / P);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!boxes || boxes[(long)k * ldbox] > -1.5f) {
      float d[8], f[8];
      unpack8(*(const uint4*)(dfeat + (long)k * lddf + c8 * 8), d);
      unpack8(*(const uint4*)(y + row * C + c8 * 8), f);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g = d[e] * inv * (segm ? segm[row] : 1.0f);
        if (drop_thr) g = vlb_keep(seed, tag, (uint32_t)k * drop_row_elems + drop_col0 + (uint32_t)(c8 * 8 + e), drop_thr) ? g * drop_scale : 0.f;
        v[e] = f[e] > 0.f ? g : 0.f;
      }
    }
    *(uint4*)(dz + row * C + c8 * 8) = pack8(v);
  }
}

extern "C" int vlb_avgpool_rows_bwd(const void* dfeat, long lddf, const void* y, const float* boxes, long ldbox, void* dz, int K, int P,
                                    int C, float drop_p, const uint32_t* seed, uint32_t tag, uint32_t drop_row_elems, uint32_t drop_col0,
                                    const float* segm, hipStream_t stream) {
  if (K <= 0) return VLB_OK;
  VLB_CHECK_ARG(dfeat && y && dz && P > 0 && C > 0 && (C % 8) == 0 && (lddf % 8) == 0, "vlb_avgpool_rows_bwd: bad argument");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_avgpool_rows_bwd: dropout needs a device seed pointer");
  const uint32_t thr = vlb_drop_thr(drop_p);
  long blocks = ((long)K * P * (C / 8) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(avgpool_rows_bwd_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)dfeat, lddf, (const bf16_t*)y, boxes,
                     ldbox, (bf16_t*)dz, K, P, C, thr, vlb_drop_scale(thr), seed, tag, drop_row_elems, drop_col0, segm);
  VLB_CHECK_LAUNCH("vlb_avgpool_rows_bwd");
  return VLB_OK;
}
