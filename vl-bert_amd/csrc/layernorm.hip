// BertLayerNorm forward / backward for gfx950 (HBM-bound; one 64-lane wave per row).
//   y = (x - mean) / sqrt(var + eps) * gamma + beta        (biased variance, eps inside sqrt)
// Follows external/pytorch_pretrained_bert/modeling.py:222-235 (the pure-torch fallback that
// defines the reference semantics when apex FusedLayerNorm is absent).  The reference runs it
// as 8 elementwise passes over the row; here a row is read once (8-B bf16x4 loads), reduced
// with wave shuffles, and written once.  The residual add / bias / dropout that precede every
// encoder LayerNorm are fused into the producing GEMM's epilogue (gemm.hip), so `x` already is
// the pre-LN sum, and it is what backward re-reads (no separate x_hat tensor is stored).
#include <stdlib.h>

#include "vlb_common.h"

#define LN_MAX_IT 8  // H <= 2048
// The dgamma/dbeta flush of 768 workgroups x 2H fp32 atomics (~1.2 M) costs as much as the whole HBM-bound row pass
// (atomic throughput, ~40 G/s).  With a workspace every workgroup stores its partial vector with plain 16-B stores
// and a small second kernel column-sums the LN_MAX_BLOCKS x 2H slab (4.7 MB at H=768) into the gradients.
#define LN_MAX_BLOCKS 1024  // 4 workgroups per CU (the 4-column backward runs 4 waves per SIMD)

// Overflow guard of the fp16 residual stream (DESIGN.md "precision"): the pre-LayerNorm sums are stored as IEEE fp16, |Z| > 65504
// becomes inf in the GEMM epilogue's conversion.  Random-init tests cannot see that; pretrained BERT has outlier channels.  Every
// LayerNorm forward notices (a row with a non-finite element has a non-finite mean / rstd) and raises a sticky per-device flag:
// bit 0 = an fp16 row, bit 1 = a bf16 row.  vlb_nonfinite_status() reads (and optionally clears) it; engine.loss_values() raises.
__device__ unsigned int g_vlb_ln_nonfinite = 0u;

// NIT = ceil(H / 256): per-lane register footprint follows the actual row width (H=768 -> 3)
template <int NIT>
struct Row4 {
  float v[NIT][4];
};

// f16: the row holds IEEE fp16 instead of bf16 (the encoder's pre-LayerNorm sums, written by vlb_gemm_nt_bf16_ex(out_f16))
// Loads are UNCONDITIONAL (a column beyond H is clamped to the row's last chunk and its values replaced by zeros) and all of a row's
// loads are issued before the first decode: with `if (c < H) load` hipcc branches around every load and waits vmcnt(0) behind it
// (three dependent HBM round trips per 1.5-KB row; the runtime fp16 / bf16 choice added a branch per element pair on top: it is now
// ONE wave-uniform branch around the decode of the whole row).
template <int NIT>
struct Raw4 {
  uint2 w[NIT];
};

template <int NIT>
__device__ __forceinline__ void raw_load_row(const bf16_t* x, int H, int lane, Raw4<NIT>& q) {
#pragma unroll
  for (int i = 0; i < NIT; ++i) q.w[i] = *(const uint2*)(x + min((lane + 64 * i) * 4, H - 4));
}

template <int NIT>
__device__ __forceinline__ void decode_row(const Raw4<NIT>& q, int H, int lane, Row4<NIT>& r, bool f16) {
  if (f16) {      // (wave-uniform; both arms are pure VALU, the loads are already in flight)
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const bool in = (lane + 64 * i) * 4 < H;
      r.v[i][0] = in ? hlo(q.w[i].x) : 0.f; r.v[i][1] = in ? hhi(q.w[i].x) : 0.f; r.v[i][2] = in ? hlo(q.w[i].y) : 0.f; r.v[i][3] = in ? hhi(q.w[i].y) : 0.f;
    }
  } else {
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const bool in = (lane + 64 * i) * 4 < H;
      r.v[i][0] = in ? bflo(q.w[i].x) : 0.f; r.v[i][1] = in ? bfhi(q.w[i].x) : 0.f; r.v[i][2] = in ? bflo(q.w[i].y) : 0.f; r.v[i][3] = in ? bfhi(q.w[i].y) : 0.f;
    }
  }
}

template <int NIT>
__device__ __forceinline__ void load_row_bf16(const bf16_t* x, int H, int lane, Row4<NIT>& r, bool f16 = false) {
  Raw4<NIT> q;
  raw_load_row(x, H, lane, q);
  decode_row(q, H, lane, r, f16);
}

// the same for an fp32 row (LayerNorm backward with an fp32 upstream gradient)
template <int NIT>
__device__ __forceinline__ void load_row_f32(const float* x, int H, int lane, Row4<NIT>& r) {
  float4 w[NIT];
#pragma unroll
  for (int i = 0; i < NIT; ++i) w[i] = *(const float4*)(x + min((lane + 64 * i) * 4, H - 4));
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const bool in = (lane + 64 * i) * 4 < H;
    r.v[i][0] = in ? w[i].x : 0.f; r.v[i][1] = in ? w[i].y : 0.f; r.v[i][2] = in ? w[i].z : 0.f; r.v[i][3] = in ? w[i].w : 0.f;
  }
}

// x: [rows] rows of H bf16 with row stride ldx (elements).  y: row stride ldy.
// RPW rows per wave, ALL their loads in flight before the first reduction, gamma / beta loaded once per wave: a wave that handles a single
// 1.5-KB row keeps too few bytes in flight for the HBM latency (the kernel ran at 4.4 TB/s where a plain streaming copy reaches 6.6,
// tools/hbm_stream_probe.hip) and reads 6 KB of gamma / beta from L2 for every 3 KB it moves.
template <int NIT, int RPW>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y, long ldy,
                                                            float* __restrict__ stats, int rows, int H, float eps, int x_f16) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
  if (row0 >= rows) return;
  Raw4<NIT> raw[RPW];
#pragma unroll
  for (int j = 0; j < RPW; ++j) raw_load_row(x + (long)min(row0 + j, rows - 1) * ldx, H, lane, raw[j]);      // (a row past the end re-reads the last one; nothing of it is stored)
  Row4<NIT> gm, bt;
  load_row_f32(gamma, H, lane, gm);          // (issued with the rows: nothing below waits for a second round trip)
  load_row_f32(beta, H, lane, bt);
#pragma unroll
  for (int j = 0; j < RPW; ++j) {
    const int row = row0 + j;
    Row4<NIT> r;
    decode_row(raw[j], H, lane, r, x_f16 != 0);
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
    if (row < rows) {      // (wave-uniform)
      if (lane == 0 && !(fabsf(mean) <= 3.0e38f && rstd <= 3.0e38f)) atomicOr(&g_vlb_ln_nonfinite, x_f16 ? 1u : 2u);     // (NaN compares false)
      if (lane == 0 && stats) {
        stats[2 * (long)row] = mean;
        stats[2 * (long)row + 1] = rstd;
      }
      bf16_t* yr = y + (long)row * ldy;
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const int c = (lane + 64 * i) * 4;
        if (c < H) {
          uint2 w;
          w.x = pack2bf((r.v[i][0] - mean) * rstd * gm.v[i][0] + bt.v[i][0], (r.v[i][1] - mean) * rstd * gm.v[i][1] + bt.v[i][1]);
          w.y = pack2bf((r.v[i][2] - mean) * rstd * gm.v[i][2] + bt.v[i][2], (r.v[i][3] - mean) * rstd * gm.v[i][3] + bt.v[i][3]);
          *(uint2*)(yr + c) = w;
        }
      }
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
// Lane layout: 16-B accesses, lane owns columns (lane + 64 i) * 8 .. +7 (H = 768: one full pass + one half-wave
// pass); two rows per wave iteration with all four row loads issued before the first reduction.
template <int NP>
struct Row8 {
  float v[NP][8];
};

template <int NP>
__device__ __forceinline__ void load_row8_bf16(const bf16_t* x, int H, int lane, Row8<NP>& r, bool f16 = false) {
  uint4 w[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) w[i] = *(const uint4*)(x + min((lane + 64 * i) * 8, H - 8));      // unconditional (see load_row_bf16)
  if (f16) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool in = (lane + 64 * i) * 8 < H;
      const uint32_t u[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { r.v[i][2 * k] = in ? hlo(u[k]) : 0.f; r.v[i][2 * k + 1] = in ? hhi(u[k]) : 0.f; }
    }
  } else {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const bool in = (lane + 64 * i) * 8 < H;
      const uint32_t u[4] = {w[i].x, w[i].y, w[i].z, w[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { r.v[i][2 * k] = in ? bflo(u[k]) : 0.f; r.v[i][2 * k + 1] = in ? bfhi(u[k]) : 0.f; }
    }
  }
}

template <int NP>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ dy_, long lddy, int dy_f32, const bf16_t* __restrict__ x,
                                                            long ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                            bf16_t* __restrict__ dx, long lddx, bf16_t* __restrict__ dx_drop, long lddd,
                                                            uint32_t drop_thr, float drop_scale, const uint32_t* __restrict__ seedp,
                                                            uint32_t tag, float* __restrict__ dx_acc, long ldacc, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ ws, int rows, int H, int x_f16) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][2][LW] per-wave partial dgamma / dbeta (lane-major)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t seed = (drop_thr && seedp) ? *seedp : 0u;
  const bool want_gb = (dgamma != nullptr) || (dbeta != nullptr);
  float gsum[NP][8], bsum[NP][8], gam[NP][8];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int c = (lane + 64 * i) * 8;
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
    if (c < H) { g0 = *(const float4*)(gamma + c); g1 = *(const float4*)(gamma + c + 4); }
    gam[i][0] = g0.x; gam[i][1] = g0.y; gam[i][2] = g0.z; gam[i][3] = g0.w;
    gam[i][4] = g1.x; gam[i][5] = g1.y; gam[i][6] = g1.z; gam[i][7] = g1.w;
#pragma unroll
    for (int k = 0; k < 8; ++k) gsum[i][k] = bsum[i][k] = 0.f;
  }

  const int nw = blockDim.x >> 6, step = gridDim.x * nw;
  for (int row0 = blockIdx.x * nw + wave; row0 < rows; row0 += 2 * step) {
    const int row1 = row0 + step;
    const bool has1 = row1 < rows;
    Row8<NP> xr[2], dyr[2];
    float mean[2], rstd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = t ? (has1 ? row1 : row0) : row0;
      load_row8_bf16(x + (long)row * ldx, H, lane, xr[t], x_f16 != 0);
      if (dy_f32) {
        const float* d = (const float*)dy_ + (long)row * lddy;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
          const int c = (lane + 64 * i) * 8;
          float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
          if (c < H) { v0 = *(const float4*)(d + c); v1 = *(const float4*)(d + c + 4); }
          dyr[t].v[i][0] = v0.x; dyr[t].v[i][1] = v0.y; dyr[t].v[i][2] = v0.z; dyr[t].v[i][3] = v0.w;
          dyr[t].v[i][4] = v1.x; dyr[t].v[i][5] = v1.y; dyr[t].v[i][6] = v1.z; dyr[t].v[i][7] = v1.w;
        }
      } else {
        load_row8_bf16((const bf16_t*)dy_ + (long)row * lddy, H, lane, dyr[t]);
      }
      const float2 ms = *(const float2*)(stats + 2 * (long)row);
      mean[t] = ms.x;
      rstd[t] = ms.y;
    }
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float live = (t == 0 || has1) ? 1.f : 0.f;   // the duplicate of row0 must not count twice
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const bool in = (lane + 64 * i) * 8 < H;         // out-of-range lanes hold zeros for dy: only xhat needs masking
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float xh = in ? (xr[t].v[i][k] - mean[t]) * rstd[t] : 0.f;
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
      for (int i = 0; i < NP; ++i) {
        const int c = (lane + 64 * i) * 8;
        if (c < H) {
          float o[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = rstd[t] * (dyr[t].v[i][k] - m1 - xr[t].v[i][k] * m2);
          if (dx) *(uint4*)(dx + (long)row * lddx + c) = make_uint4(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7]));
          if (dx_drop) {
            float d[8];
            if (drop_thr) {
              const uint32_t idx = (uint32_t)row * (uint32_t)H + (uint32_t)c;  // H%8==0 -> idx even
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const uint32_t h = vlb_rng_pair(seed, tag, (idx >> 1) + q);
                d[2 * q] = ((h & 0xffffu) >= drop_thr) ? o[2 * q] * drop_scale : 0.f;
                d[2 * q + 1] = ((h >> 16) >= drop_thr) ? o[2 * q + 1] * drop_scale : 0.f;
              }
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) d[k] = o[k];
            }
            *(uint4*)(dx_drop + (long)row * lddd + c) = make_uint4(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]), pack2bf(d[4], d[5]), pack2bf(d[6], d[7]));
          }
          if (dx_acc) {
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(dx_acc + (long)row * ldacc + c + k, o[k]);
          }
        }
      }
    }
  }
  if (!want_gb) return;
  // Per-wave partials -> LDS in LANE-MAJOR order (element (i,k) of lane l at (i*8+k)*64 + l: conflict-free plain stores,
  // no LDS atomics -- the natural column order has lanes 8 floats apart = 16-way bank conflicts), summed over the 4 waves
  // by the whole workgroup in the same order.  q -> column: ln_col_of().
  constexpr int LW = NP * 512;
  float* mine = red + wave * 2 * LW;
#pragma unroll
  for (int i = 0; i < NP; ++i)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      mine[(i * 8 + k) * 64 + lane] = gsum[i][k];
      mine[LW + (i * 8 + k) * 64 + lane] = bsum[i][k];
    }
  __syncthreads();
  for (int q = threadIdx.x; q < 2 * LW; q += 256) {
    const float v = (red[q] + red[2 * LW + q]) + (red[4 * LW + q] + red[6 * LW + q]);
    if (ws) ws[(long)blockIdx.x * 2 * LW + q] = v;   // partial vector of this workgroup; ln_param_finalize_kernel column-sums them
    else red[q] = v;                                  // (only this thread reads red[q])
  }
  if (ws) return;
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * H; c += 256) {    // no workspace: coalesced atomics in natural column order
    const int which = c >= H, col = c - which * H, chunk = col >> 3;
    const float v = red[which * LW + (((chunk >> 6) * 8 + (col & 7)) << 6) + (chunk & 63)];
    float* dst = which ? dbeta : dgamma;
    if (dst) atomicAdd(dst + col, v);
  }
}

// The same backward with 4-column (8-B) lane chunks -- the forward's mapping: a lane owns columns (lane + 64 i) * 4 .. +3.
// H = 768 is then three FULL passes (the 8-column form above runs one full and one half-wave pass and keeps 16 values per
// array per lane: 192 VGPRs, 2 waves per SIMD -- too few bytes in flight for a kernel that only streams: 2.8 TB/s measured);
// here the footprint is 12 values per array, two rows in flight, <= 128 VGPRs -> 4 waves per SIMD.
// Partial dgamma / dbeta vectors are lane-major with LW = NIT * 256: element (i, k) of lane l at (i*4 + k)*64 + l.
template <int NIT, int RIF>      // RIF rows in flight per wave iteration (2 while the registers allow it)
__global__ __launch_bounds__(256, (NIT * RIF <= 3 ? 4 : 1)) void layernorm_bwd4_kernel(const void* __restrict__ dy_, long lddy, int dy_f32, const bf16_t* __restrict__ x,
                                                             long ldx, const float* __restrict__ stats, const float* __restrict__ gamma,
                                                             bf16_t* __restrict__ dx, long lddx, bf16_t* __restrict__ dx_drop, long lddd,
                                                             uint32_t drop_thr, float drop_scale, const uint32_t* __restrict__ seedp,
                                                             uint32_t tag, float* __restrict__ dx_acc, long ldacc, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, float* __restrict__ ws, int rows, int H, int x_f16) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4 waves][2][LW] + gamma [4 waves][NIT][64] float4
  constexpr int LW = NIT * 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t seed = (drop_thr && seedp) ? *seedp : 0u;
  const bool want_gb = (dgamma != nullptr) || (dbeta != nullptr);
  float gsum[NIT][4], bsum[NIT][4];
  // gamma of the lane's columns lives in LDS (behind the reduction area, one float4 per (pass, lane), written and read by the SAME
  // lane: no barrier), not in 4 NIT registers for the whole row loop: with every load of an iteration in flight at once the kernel
  // would not fit the 128 registers of 4 waves per SIMD otherwise.  The opaque copy of the address keeps the reads inside the loop.
  float4* const gam_l = (float4*)(red + 8 * LW) + wave * NIT * 64 + lane;
  {
    Row4<NIT> g0;
    load_row_f32(gamma, H, lane, g0);
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      gam_l[i * 64] = make_float4(g0.v[i][0], g0.v[i][1], g0.v[i][2], g0.v[i][3]);
#pragma unroll
      for (int k = 0; k < 4; ++k) gsum[i][k] = bsum[i][k] = 0.f;
    }
  }
  const int nw = blockDim.x >> 6, step = gridDim.x * nw;
  for (int row0 = blockIdx.x * nw + wave; row0 < rows; row0 += RIF * step) {
    Row4<NIT> xr[RIF], dyr[RIF];
    float mean[RIF], rstd[RIF];
    bool has[RIF];
    // every load of the iteration (x rows, dy rows, row statistics) is issued before the first decode waits for one
    Raw4<NIT> qx[RIF], qd[RIF];
    float2 ms[RIF];
#pragma unroll
    for (int t = 0; t < RIF; ++t) {
      has[t] = row0 + t * step < rows;
      const int row = has[t] ? row0 + t * step : row0;
      raw_load_row(x + (long)row * ldx, H, lane, qx[t]);
      if (!dy_f32) raw_load_row((const bf16_t*)dy_ + (long)row * lddy, H, lane, qd[t]);
      ms[t] = *(const float2*)(stats + 2 * (long)row);
    }
#pragma unroll
    for (int t = 0; t < RIF; ++t) {
      const int row = has[t] ? row0 + t * step : row0;
      if (dy_f32) load_row_f32((const float*)dy_ + (long)row * lddy, H, lane, dyr[t]);
      else decode_row(qd[t], H, lane, dyr[t], false);
      decode_row(qx[t], H, lane, xr[t], x_f16 != 0);
      mean[t] = ms[t].x;
      rstd[t] = ms[t].y;
    }
    float s1[RIF], s2[RIF];
    float gam[NIT][4];
    {
      const float4* gp = gam_l;
      asm volatile("" : "+v"(gp));
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const float4 g4 = gp[i * 64];
        gam[i][0] = g4.x; gam[i][1] = g4.y; gam[i][2] = g4.z; gam[i][3] = g4.w;
      }
    }
#pragma unroll
    for (int t = 0; t < RIF; ++t) {
      s1[t] = s2[t] = 0.f;
      const float live = has[t] ? 1.f : 0.f;            // the duplicate of row0 must not count
#pragma unroll
      for (int i = 0; i < NIT; ++i) {
        const bool in = (lane + 64 * i) * 4 < H;         // out-of-range lanes hold zeros for dy: only xhat needs masking
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xh = in ? (xr[t].v[i][k] - mean[t]) * rstd[t] : 0.f;
          const float dyv = dyr[t].v[i][k] * live;
          gsum[i][k] += dyv * xh;
          bsum[i][k] += dyv;
          const float gv = dyv * gam[i][k];
          s1[t] += gv;
          s2[t] += gv * xh;
          xr[t].v[i][k] = xh;    // reuse storage: xhat
          dyr[t].v[i][k] = gv;   // reuse storage: g
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {     // independent butterflies interleaved
#pragma unroll
      for (int t = 0; t < RIF; ++t) { s1[t] += __shfl_xor(s1[t], o, 64); s2[t] += __shfl_xor(s2[t], o, 64); }
    }
#pragma unroll
    for (int t = 0; t < RIF; ++t) {
      if (!has[t]) continue;
      const int row = row0 + t * step;
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
            float d[4] = {o[0], o[1], o[2], o[3]};
            if (drop_thr) {
              const uint32_t idx = (uint32_t)row * (uint32_t)H + (uint32_t)c;      // H % 4 == 0 -> idx even
              const uint32_t h0 = vlb_rng_pair(seed, tag, idx >> 1), h1 = vlb_rng_pair(seed, tag, (idx >> 1) + 1);
              d[0] = ((h0 & 0xffffu) >= drop_thr) ? o[0] * drop_scale : 0.f;
              d[1] = ((h0 >> 16) >= drop_thr) ? o[1] * drop_scale : 0.f;
              d[2] = ((h1 & 0xffffu) >= drop_thr) ? o[2] * drop_scale : 0.f;
              d[3] = ((h1 >> 16) >= drop_thr) ? o[3] * drop_scale : 0.f;
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
  float* mine = red + wave * 2 * LW;
#pragma unroll
  for (int i = 0; i < NIT; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mine[(i * 4 + k) * 64 + lane] = gsum[i][k];
      mine[LW + (i * 4 + k) * 64 + lane] = bsum[i][k];
    }
  __syncthreads();
  for (int q = threadIdx.x; q < 2 * LW; q += 256) {
    const float v = (red[q] + red[2 * LW + q]) + (red[4 * LW + q] + red[6 * LW + q]);
    if (ws) ws[(long)blockIdx.x * 2 * LW + q] = v;
    else red[q] = v;
  }
  if (ws) return;
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * H; c += 256) {    // no workspace: coalesced atomics in natural column order
    const int which = c >= H, col = c - which * H, chunk = col >> 2;
    const float v = red[which * LW + (((chunk >> 6) * 4 + (col & 3)) << 6) + (chunk & 63)];
    float* dst = which ? dbeta : dgamma;
    if (dst) atomicAdd(dst + col, v);
  }
}

// (Round 4 built and REMOVED a software-pipelined form of the kernel above: next row's loads issued in front of the current row's stores
// through untracked inline-asm loads and one explicit `s_waitcnt vmcnt(stores per iteration)`.  5-7 % faster -- and unsound: vector-memory
// loads retire in order among themselves, stores among themselves, but NOT with respect to each other, so "at most 6 operations
// outstanding" did not prove the 7 older loads had landed once the 6 younger stores completed first.  It passed every test on warm data and
// produced NaN gradients after other tests had cooled the caches.  That is also why hipcc waits with vmcnt(0) wherever loads and stores are
// pending together -- it is not being conservative.  DESIGN.md §3, "HBM-bound kernels".)

// dgamma/dbeta += column sums of the `nslab` lane-major partial vectors [2][LW]: grid (2 LW / 64, 8)
__global__ __launch_bounds__(256) void ln_param_finalize_kernel(const float* __restrict__ ws, int nslab, int LW, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int H, int cpl_log2) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + tx;
  float s = 0.f;
  for (int r = blockIdx.y * 4 + ty; r < nslab; r += 4 * gridDim.y) s += ws[(long)r * 2 * LW + q];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0) {
    s = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
    const int r = q % LW, cpl = 1 << cpl_log2;      // columns per lane chunk: 8 (layernorm_bwd_kernel) or 4 (layernorm_bwd4_kernel)
    const int col = ((r & 63) + 64 * (r >> (6 + cpl_log2))) * cpl + ((r >> 6) & (cpl - 1));
    if (col < H) {
      if (q < LW) { if (dgamma) atomicAdd(dgamma + col, s); }
      else if (dbeta) atomicAdd(dbeta + col, s);
    }
  }
}

// rows per wave of the forward: VLB_LN_FWD_ROWS = 1 | 2 | 4 (0 / unset: 2 where at least two full rounds of single-row waves exist);
// run-time override for A/B measurements and tests: vlb_gemm_set_option("ln_fwd_rows", v) / ("ln_bwd4", v)
static int g_ln_fwd_rows = -1;
static int g_ln_bwd4 = -1;      // VLB_LN_BWD4: 1 (default) the 4-column kernel (2: two rows in flight at H = 768 / 1024); 0 the 8-column kernel
void vlb_ln_set_fwd_rows(int v) { g_ln_fwd_rows = v; }
void vlb_ln_set_bwd4(int v) { g_ln_bwd4 = v; }

extern "C" int vlb_layernorm_fwd(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                 float* stats, int rows, int H, float eps, int x_f16, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= 256 * LN_MAX_IT, "vlb_layernorm_fwd: unsupported H=%d", H);
  VLB_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0, "vlb_layernorm_fwd: row strides must be multiples of 4");
  if (g_ln_fwd_rows < 0) {
    const char* e = getenv("VLB_LN_FWD_ROWS");
    g_ln_fwd_rows = e ? atoi(e) : 0;
  }
  const int rpw_opt = g_ln_fwd_rows;
  const int rpw = (rpw_opt == 1 || rpw_opt == 2 || rpw_opt == 4) ? rpw_opt : (rows >= 2 * 256 * 32 ? 2 : 1);
#define LN_FWD_R(NIT, RPW)                                                                                                              \
  hipLaunchKernelGGL((layernorm_fwd_kernel<NIT, RPW>), dim3(vlb_cdiv(rows, 4 * RPW)), dim3(256), 0, stream, (const bf16_t*)x, ldx, gamma, \
                     beta, (bf16_t*)y, ldy, stats, rows, H, eps, x_f16)
#define LN_FWD(NIT)                                                                       \
  do {                                                                                    \
    if (rpw == 4 && NIT <= 4) LN_FWD_R(NIT, 4); else if (rpw >= 2 && NIT <= 4) LN_FWD_R(NIT, 2); else LN_FWD_R(NIT, 1); \
  } while (0)
  const int nit = vlb_cdiv(H, 256);
  if (nit <= 1) LN_FWD(1); else if (nit == 2) LN_FWD(2); else if (nit == 3) LN_FWD(3); else if (nit == 4) LN_FWD(4); else LN_FWD_R(8, 1);
#undef LN_FWD
#undef LN_FWD_R
  VLB_CHECK_LAUNCH("vlb_layernorm_fwd");
  return VLB_OK;
}

// Sticky non-finite flag of the LayerNorm forwards on the CURRENT device (see g_vlb_ln_nonfinite): returns its value (0 = clean;
// bit 0: an fp16 pre-LayerNorm row overflowed, bit 1: a bf16 row held inf / NaN), negative on a HIP error.  Synchronises the device.
extern "C" int vlb_nonfinite_status(int reset) {
  unsigned int v = 0u;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_vlb_ln_nonfinite), sizeof(v), 0, hipMemcpyDeviceToHost);
  if (e == hipSuccess && reset && v) {
    const unsigned int z = 0u;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_vlb_ln_nonfinite), &z, sizeof(z), 0, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    vlb_set_error("vlb_nonfinite_status: %s", hipGetErrorString(e));
    return VLB_ERR_HIP;
  }
  return (int)(v & 3u);
}


static int ln_bwd_blocks(int rows) {
  int blocks = vlb_cdiv(rows, 8);      // 4 waves per workgroup, one or two rows per wave iteration
  if (blocks < 1) blocks = 1;
  if (blocks > LN_MAX_BLOCKS) blocks = LN_MAX_BLOCKS;
  return blocks;
}

static void ln_lw(int H, int& LW, int& cpl_log2) {
  if (g_ln_bwd4) {
    const int nit = vlb_cdiv(H, 256);
    LW = (nit <= 4 ? nit : 8) * 256;
    cpl_log2 = 2;
  } else {
    const int np = vlb_cdiv(H, 512);
    LW = (np <= 3 ? np : 4) * 512;
    cpl_log2 = 3;
  }
}

static int ln_bwd_impl(const void* dy, long lddy, int dy_f32, const void* x, long ldx, const float* stats,
                       const float* gamma, void* dx, long lddx, void* dx_drop, long lddd, float drop_p,
                       const uint32_t* seed, uint32_t tag, float* dx_acc, long ldacc, float* dgamma, float* dbeta,
                       float* workspace, int rows, int H, int x_f16, bool finalize, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(H > 0 && (H % 8) == 0 && H <= 2048, "vlb_layernorm_bwd: unsupported H=%d (multiple of 8, <= 2048)", H);
  VLB_CHECK_ARG((lddy % 8) == 0 && (ldx % 8) == 0 && (lddx % 8) == 0 && (lddd % 8) == 0,
                "vlb_layernorm_bwd: row strides must be multiples of 8");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_layernorm_bwd: dropout needs a device seed pointer");
  VLB_CHECK_ARG((long)rows * H < (1L << 32) || !(drop_p > 0.f), "vlb_layernorm_bwd: dropout index overflow");
  if (g_ln_bwd4 < 0) {
    const char* v = getenv("VLB_LN_BWD4");
    g_ln_bwd4 = v ? atoi(v) : 1;
  }
  const int blocks = ln_bwd_blocks(rows);
  float* ws = (workspace && (dgamma || dbeta) && blocks > 32) ? workspace : nullptr;
  const uint32_t thr = vlb_drop_thr(drop_p);
  int LW, cpl_log2;
  if (g_ln_bwd4) {
    const int nit = vlb_cdiv(H, 256);
#define LN_BWD4(NIT, RIF)                                                                                                      \
  do {                                                                                                                          \
    constexpr size_t smem = (8 * NIT * 256 + 4 * NIT * 256) * sizeof(float);      /* reduction area + the lanes' gamma */          \
    if (smem > 65536)                                                                                                           \
      (void)hipFuncSetAttribute((const void*)layernorm_bwd4_kernel<NIT, RIF>, hipFuncAttributeMaxDynamicSharedMemorySize, smem); \
    hipLaunchKernelGGL((layernorm_bwd4_kernel<NIT, RIF>), dim3(blocks), dim3(256), smem, stream, dy, lddy, dy_f32,               \
                       (const bf16_t*)x, ldx, stats, gamma, (bf16_t*)dx, lddx, (bf16_t*)dx_drop, lddd, thr, vlb_drop_scale(thr), \
                       seed, tag, dx_acc, ldacc, dgamma, dbeta, ws, rows, H, x_f16);                                             \
  } while (0)
    // rows in flight per wave: 2 while the kernel stays near 128 VGPRs (4 waves per SIMD); g_ln_bwd4 == 2 forces 2 for H = 768 / 1024
    if (nit <= 1) LN_BWD4(1, 2); else if (nit == 2) LN_BWD4(2, 2);
    else if (nit == 3) { if (g_ln_bwd4 == 2) LN_BWD4(3, 2); else LN_BWD4(3, 1); }
    else if (nit == 4) { if (g_ln_bwd4 == 2) LN_BWD4(4, 2); else LN_BWD4(4, 1); }
    else LN_BWD4(8, 1);
#undef LN_BWD4
  } else {
#define LN_BWD(NP)                                                                                                             \
  hipLaunchKernelGGL(layernorm_bwd_kernel<NP>, dim3(blocks), dim3(256), 8 * NP * 512 * sizeof(float), stream, dy, lddy, dy_f32,  \
                     (const bf16_t*)x, ldx, stats, gamma, (bf16_t*)dx, lddx, (bf16_t*)dx_drop, lddd, thr, vlb_drop_scale(thr),  \
                     seed, tag, dx_acc, ldacc, dgamma, dbeta, ws, rows, H, x_f16)
    const int np = vlb_cdiv(H, 512);
    if (np <= 1) LN_BWD(1); else if (np == 2) LN_BWD(2); else if (np == 3) LN_BWD(3); else LN_BWD(4);
#undef LN_BWD
  }
  ln_lw(H, LW, cpl_log2);
  VLB_CHECK_LAUNCH("vlb_layernorm_bwd");
  if (ws && finalize) {
    hipLaunchKernelGGL(ln_param_finalize_kernel, dim3(2 * LW / 64, 8), dim3(256), 0, stream, ws, blocks, LW, dgamma, dbeta, H, cpl_log2);
    VLB_CHECK_LAUNCH("vlb_layernorm_bwd(finalize)");
  }
  return VLB_OK;
}

extern "C" int vlb_layernorm_bwd(const void* dy, long lddy, int dy_f32, const void* x, long ldx, const float* stats,
                                 const float* gamma, void* dx, long lddx, void* dx_drop, long lddd, float drop_p,
                                 const uint32_t* seed, uint32_t tag, float* dx_acc, long ldacc, float* dgamma, float* dbeta,
                                 float* workspace, int rows, int H, int x_f16, hipStream_t stream) {
  return ln_bwd_impl(dy, lddy, dy_f32, x, ldx, stats, gamma, dx, lddx, dx_drop, lddd, drop_p, seed, tag, dx_acc, ldacc, dgamma, dbeta,
                     workspace, rows, H, x_f16, true, stream);
}

// The same with the parameter-gradient finalize DEFERRED: the per-workgroup partial dgamma / dbeta vectors stay in `workspace`
// (which therefore must not be reused before they are consumed) and vlb_ln_param_finalize_batch adds the column sums of up to 32
// such workspaces into their gradients in ONE launch -- a training step runs 26 LayerNorm backwards whose parameter gradients are
// only needed by the optimizer (or a bucket's all-reduce).  vlb_layernorm_bwd_slabs(rows) = number of partial vectors the call
// leaves (0: the call was small enough to add its sums directly; nothing to finalize).
extern "C" int vlb_layernorm_bwd_deferred(const void* dy, long lddy, int dy_f32, const void* x, long ldx, const float* stats,
                                          const float* gamma, void* dx, long lddx, void* dx_drop, long lddd, float drop_p,
                                          const uint32_t* seed, uint32_t tag, float* dx_acc, long ldacc, float* dgamma, float* dbeta,
                                          float* workspace, int rows, int H, int x_f16, hipStream_t stream) {
  VLB_CHECK_ARG(workspace && (dgamma || dbeta), "vlb_layernorm_bwd_deferred: needs a workspace and a parameter gradient");
  return ln_bwd_impl(dy, lddy, dy_f32, x, ldx, stats, gamma, dx, lddx, dx_drop, lddd, drop_p, seed, tag, dx_acc, ldacc, dgamma, dbeta,
                     workspace, rows, H, x_f16, false, stream);
}

extern "C" int vlb_layernorm_bwd_slabs(int rows) {
  if (rows <= 0) return 0;
  const int blocks = ln_bwd_blocks(rows);
  return blocks > 32 ? blocks : 0;
}

struct LnFinalizeBatch {
  const float* ws[32];
  float* dgamma[32];
  float* dbeta[32];
  int nslab[32];
};

__global__ __launch_bounds__(256) void ln_param_finalize_batch_kernel(const LnFinalizeBatch b, int LW, int H, int cpl_log2) {
  __shared__ float part[4][64];
  const int e = blockIdx.z;
  const float* ws = b.ws[e];
  const int nslab = b.nslab[e];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int q = blockIdx.x * 64 + tx;
  float s = 0.f;
  for (int r = blockIdx.y * 4 + ty; r < nslab; r += 4 * gridDim.y) s += ws[(long)r * 2 * LW + q];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0) {
    s = (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]);
    const int r = q % LW, cpl = 1 << cpl_log2;
    const int col = ((r & 63) + 64 * (r >> (6 + cpl_log2))) * cpl + ((r >> 6) & (cpl - 1));
    if (col < H) {
      if (q < LW) { if (b.dgamma[e]) atomicAdd(b.dgamma[e] + col, s); }
      else if (b.dbeta[e]) atomicAdd(b.dbeta[e] + col, s);
    }
  }
}

extern "C" int vlb_ln_param_finalize_batch(int n, const float* const* ws, const int* nslab, float* const* dgamma, float* const* dbeta,
                                           int H, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(n <= 32 && ws && nslab && dgamma && dbeta, "vlb_ln_param_finalize_batch: 1..32 entries");
  VLB_CHECK_ARG(H > 0 && (H % 8) == 0 && H <= 2048, "vlb_ln_param_finalize_batch: unsupported H=%d", H);
  if (g_ln_bwd4 < 0) {
    const char* v = getenv("VLB_LN_BWD4");
    g_ln_bwd4 = v ? atoi(v) : 1;
  }
  LnFinalizeBatch b;
  for (int i = 0; i < n; ++i) {
    VLB_CHECK_ARG(ws[i] && nslab[i] > 0, "vlb_ln_param_finalize_batch: entry %d has no partial vectors", i);
    b.ws[i] = ws[i]; b.nslab[i] = nslab[i]; b.dgamma[i] = dgamma[i]; b.dbeta[i] = dbeta[i];
  }
  int LW, cpl_log2;
  ln_lw(H, LW, cpl_log2);
  hipLaunchKernelGGL(ln_param_finalize_batch_kernel, dim3(2 * LW / 64, 8, n), dim3(256), 0, stream, b, LW, H, cpl_log2);
  VLB_CHECK_LAUNCH("vlb_ln_param_finalize_batch");
  return VLB_OK;
}

extern "C" long vlb_layernorm_bwd_workspace_floats(int H) {
  if (H <= 0) return 0;
  const int np = vlb_cdiv(H, 512), nit = vlb_cdiv(H, 256);
  const long lw8 = (np <= 3 ? np : 4) * 512, lw4 = (nit <= 4 ? nit : 8) * 256;      // either backward kernel may run
  return (long)LN_MAX_BLOCKS * 2 * (lw8 > lw4 ? lw8 : lw4);
}
