// BertLayerNorm forward / backward for gfx950 (HBM-bound; one 64-lane wave per row).
//   y = (x - mean) / sqrt(var + eps) * gamma + beta        (biased variance, eps inside sqrt)
// Follows external/pytorch_pretrained_bert/modeling.py:222-235 (the pure-torch fallback that
// defines the reference semantics when apex FusedLayerNorm is absent).  The reference runs it
// as 8 elementwise passes over the row; here a row is read once (8-B bf16x4 loads), reduced
// with wave shuffles, and written once.  The residual add / bias / dropout that precede every
// encoder LayerNorm are fused into the producing GEMM's epilogue (gemm.hip), so `x` already is
// the pre-LN sum, and it is what backward re-reads (no separate x_hat tensor is stored).
#include "vlb_common.h"

#define LN_MAX_IT 8  // H <= 2048

// NIT = ceil(H / 256): per-lane register footprint follows the actual row width (H=768 -> 3)
template <int NIT>
struct Row4 {
  float v[NIT][4];
};

template <int NIT>
__device__ __forceinline__ void load_row_bf16(const bf16_t* x, int H, int lane, Row4<NIT>& r) {
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const uint2 w = *(const uint2*)(x + c);
      r.v[i][0] = bflo(w.x); r.v[i][1] = bfhi(w.x); r.v[i][2] = bflo(w.y); r.v[i][3] = bfhi(w.y);
    } else {
      r.v[i][0] = r.v[i][1] = r.v[i][2] = r.v[i][3] = 0.f;
    }
  }
}

// x: [rows] rows of H bf16 with row stride ldx (elements).  y: row stride ldy.
template <int NIT>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y, long ldy,
                                                            float* __restrict__ stats, int rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  Row4<NIT> r;
  load_row_bf16(x + (long)row * ldx, H, lane, r);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) s += (r.v[i][0] + r.v[i][1]) + (r.v[i][2] + r.v[i][3]);
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = r.v[i][k] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0 && stats) {
    stats[2 * (long)row] = mean;
    stats[2 * (long)row + 1] = rstd;
  }
  bf16_t* yr = y + (long)row * ldy;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const float4 g = *(const float4*)(gamma + c);
      const float4 b = *(const float4*)(beta + c);
      uint2 w;
      w.x = pack2bf((r.v[i][0] - mean) * rstd * g.x + b.x, (r.v[i][1] - mean) * rstd * g.y + b.y);
      w.y = pack2bf((r.v[i][2] - mean) * rstd * g.z + b.z, (r.v[i][3] - mean) * rstd * g.w + b.w);
      *(uint2*)(yr + c) = w;
    }
  }
}

// Backward.  dy: bf16 (or fp32 when dy_f32) rows; x: the saved pre-LN rows; stats: (mean, rstd).
//   dx = rstd * (g - mean_H(g) - xhat * mean_H(g * xhat)),  g = dy * gamma
//   dgamma += sum_rows dy * xhat ; dbeta += sum_rows dy          (fp32 atomics, one flush per block)
// Outputs (any may be null): dx bf16, dx_drop bf16 = dx * keep(idx) * scale  (the gradient that
// flows into the dense layer *before* its dropout; idx = row*H + col matches the GEMM epilogue),
// dx_acc fp32 (atomicAdd; used when several rows alias one input row, e.g. the broadcast
// text_visual_embeddings of pretrain/modules/resnet_vlbert_for_pretraining.py:132-135).
template <int NIT>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ dy_, long lddy, int dy_f32, const bf16_t* __restrict__ x,
                                                            long ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            bf16_t* __restrict__ dx, long lddx, bf16_t* __restrict__ dx_drop, long lddd,
                                                            uint32_t drop_thr, float drop_scale, const uint32_t* __restrict__ seedp,
                                                            uint32_t tag, float* __restrict__ dx_acc, long ldacc, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int rows, int H) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][H] block accumulators for dgamma / dbeta
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t seed = (drop_thr && seedp) ? *seedp : 0u;
  const bool want_gb = (dgamma != nullptr) || (dbeta != nullptr);
  if (want_gb) {
    for (int i = threadIdx.x; i < 2 * H; i += 256) red[i] = 0.f;
    __syncthreads();
  }
  float gsum[NIT][4], bsum[NIT][4];
#pragma unroll
  for (int i = 0; i < NIT; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) gsum[i][k] = bsum[i][k] = 0.f;
  float gam[NIT][4];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lane + 64 * i) * 4;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < H) g = *(const float4*)(gamma + c);
    gam[i][0] = g.x; gam[i][1] = g.y; gam[i][2] = g.z; gam[i][3] = g.w;
  }

  // Two rows per wave iteration: both rows' loads are issued before either row's reductions (memory-level
  // parallelism for an HBM-bound kernel that otherwise waits a full round trip per row).
  const int step = gridDim.x * 4;
  for (int row0 = blockIdx.x * 4 + wave; row0 < rows; row0 += 2 * step) {
    const int row1 = row0 + step;
    const bool has1 = row1 < rows;
    Row4<NIT> xr[2], dyr[2];
    float mean[2], rstd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = t ? (has1 ? row1 : row0) : row0;
      load_row_bf16(x + (long)row * ldx, H, lane, xr[t]);
      if (dy_f32) {
        const float* d = (const float*)dy_ + (long)row * lddy;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int c = (lane + 64 * i) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (c < H) v = *(const float4*)(d + c);
          dyr[t].v[i][0] = v.x; dyr[t].v[i][1] = v.y; dyr[t].v[i][2] = v.z; dyr[t].v[i][3] = v.w;
        }
      } else {
        load_row_bf16((const bf16_t*)dy_ + (long)row * lddy, H, lane, dyr[t]);
      }
      mean[t] = stats[2 * (long)row];
      rstd[t] = stats[2 * (long)row + 1];
    }
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float live = (t == 0 || has1) ? 1.f : 0.f;   // the duplicate of row0 must not count twice
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < H) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float xh = (xr[t].v[i][k] - mean[t]) * rstd[t];
            const float dyv = dyr[t].v[i][k] * live;
            gsum[i][k] += dyv * xh;
            bsum[i][k] += dyv;
            const float gv = dyv * gam[i][k];
            s1[t] += gv;
            s2[t] += gv * xh;
            xr[t].v[i][k] = xh;   // reuse storage: xhat
            dyr[t].v[i][k] = gv;  // reuse storage: g
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {   // four independent butterflies interleaved
      s1[0] += __shfl_xor(s1[0], o, 64); s2[0] += __shfl_xor(s2[0], o, 64);
      s1[1] += __shfl_xor(s1[1], o, 64); s2[1] += __shfl_xor(s2[1], o, 64);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t == 1 && !has1) break;
      const int row = t ? row1 : row0;
      const float m1 = s1[t] / (float)H, m2 = s2[t] / (float)H;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < H) {
          float o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = rstd[t] * (dyr[t].v[i][k] - m1 - xr[t].v[i][k] * m2);
          if (dx) *(uint2*)(dx + (long)row * lddx + c) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));
          if (dx_drop) {
            float d[4];
            if (drop_thr) {
              const uint32_t idx = (uint32_t)row * (uint32_t)H + (uint32_t)c;  // H%4==0 -> idx even
              const uint32_t h0 = vlb_rng_pair(seed, tag, idx >> 1), h1 = vlb_rng_pair(seed, tag, (idx >> 1) + 1);
              d[0] = ((h0 & 0xffffu) >= drop_thr) ? o[0] * drop_scale : 0.f;
              d[1] = ((h0 >> 16) >= drop_thr) ? o[1] * drop_scale : 0.f;
              d[2] = ((h1 & 0xffffu) >= drop_thr) ? o[2] * drop_scale : 0.f;
              d[3] = ((h1 >> 16) >= drop_thr) ? o[3] * drop_scale : 0.f;
            } else {
              d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3];
            }
            *(uint2*)(dx_drop + (long)row * lddd + c) = make_uint2(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]));
          }
          if (dx_acc) {
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(dx_acc + (long)row * ldacc + c + k, o[k]);
          }
        }
      }
    }
  }
  if (!want_gb) return;
  // per-wave partials -> LDS atomics -> one global atomic per column per block
  float* rg = red;
  float* rb = red + H;
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(rg + c + k, gsum[i][k]);
        atomicAdd(rb + c + k, bsum[i][k]);
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    if (dgamma) atomicAdd(dgamma + c, rg[c]);
    if (dbeta) atomicAdd(dbeta + c, rb[c]);
  }
}

extern "C" int vlb_layernorm_fwd(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                 float* stats, int rows, int H, float eps, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= 256 * LN_MAX_IT, "vlb_layernorm_fwd: unsupported H=%d", H);
  VLB_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0, "vlb_layernorm_fwd: row strides must be multiples of 4");
#define LN_FWD(NIT)                                                                                                         \
  hipLaunchKernelGGL(layernorm_fwd_kernel<NIT>, dim3(vlb_cdiv(rows, 4)), dim3(256), 0, stream, (const bf16_t*)x, ldx, gamma, \
                     beta, (bf16_t*)y, ldy, stats, rows, H, eps)
  const int nit = vlb_cdiv(H, 256);
  if (nit <= 1) LN_FWD(1); else if (nit == 2) LN_FWD(2); else if (nit == 3) LN_FWD(3); else if (nit == 4) LN_FWD(4); else LN_FWD(8);
#undef LN_FWD
  VLB_CHECK_LAUNCH("vlb_layernorm_fwd");
  return VLB_OK;
}

extern "C" int vlb_layernorm_bwd(const void* dy, long lddy, int dy_f32, const void* x, long ldx, const float* stats,
                                 const float* gamma, void* dx, long lddx, void* dx_drop, long lddd, float drop_p,
                                 const uint32_t* seed, uint32_t tag, float* dx_acc, long ldacc, float* dgamma, float* dbeta,
                                 int rows, int H, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= 256 * LN_MAX_IT, "vlb_layernorm_bwd: unsupported H=%d", H);
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_layernorm_bwd: dropout needs a device seed pointer");
  VLB_CHECK_ARG((long)rows * H < (1L << 32) || !(drop_p > 0.f), "vlb_layernorm_bwd: dropout index overflow");
  int blocks = vlb_cdiv(rows, 8);      // a wave handles two rows per iteration
  if (blocks < 1) blocks = 1;
  if (blocks > 768) blocks = 768;      // 3 workgroups per CU; fewer blocks = fewer dgamma/dbeta atomics
  const uint32_t thr = vlb_drop_thr(drop_p);
#define LN_BWD(NIT)                                                                                                            \
  hipLaunchKernelGGL(layernorm_bwd_kernel<NIT>, dim3(blocks), dim3(256), 2 * H * sizeof(float), stream, dy, lddy, dy_f32,       \
                     (const bf16_t*)x, ldx, stats, gamma, (bf16_t*)dx, lddx, (bf16_t*)dx_drop, lddd, thr, vlb_drop_scale(thr),  \
                     seed, tag, dx_acc, ldacc, dgamma, dbeta, rows, H)
  const int nit = vlb_cdiv(H, 256);
  if (nit <= 1) LN_BWD(1); else if (nit == 2) LN_BWD(2); else if (nit == 3) LN_BWD(3); else if (nit == 4) LN_BWD(4); else LN_BWD(8);
#undef LN_BWD
  VLB_CHECK_LAUNCH("vlb_layernorm_bwd");
  return VLB_OK;
}
