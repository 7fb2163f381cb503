// Optimizer-side kernels over the FLAT parameter buffers (gfx950; pure HBM streaming):
//   * vlb_sumsq_f32      - sum of squares of the flat fp32 gradient (global-norm clip,
//                          torch.nn.utils.clip_grad_norm_ as called at common/trainer.py:139-145)
//   * vlb_adamw_step     - fused multi-tensor AdamW with the reference's semantics
//                          (common/nlp/bert/optimization.py:155-185: bias-corrected step size,
//                          eps outside the sqrt, decoupled decay applied AFTER the Adam update
//                          with the un-corrected lr), the clip coefficient folded in, and the
//                          bf16 working copy of the weights emitted in the same pass.
//   * vlb_cast_f32_bf16  - fp32 -> bf16 (initial / externally modified weights)
// The reference walks ~400 parameters in Python with ~8 elementwise kernels each; here the whole
// model is one launch over one contiguous buffer.  Hyper-parameters that change every step
// (lr, step count) live in a small DEVICE struct so a captured hipGraph can be replayed.
#include "vlb_common.h"

struct VlbAdamState {   // device-resident, 8 floats
  float lr;             // current lr (set by the host, or on the device by vlb_lr_schedule_step)
  float beta1, beta2, eps, weight_decay;
  float step;           // number of steps already taken (incremented by the kernel's block 0)
  float max_norm;       // <=0: no clipping
  float sumsq;          // sum of squares of the gradient (written by vlb_sumsq_f32)
};

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *(const float4*)(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long k = i; k < n; ++k) s += g[k] * g[k];
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sh[0] + sh[1] + sh[2] + sh[3]);
}

// Deterministic variant: per-block partial sums to a workspace, one block adds them in a fixed order.  The clip coefficient
// derived from this sum multiplies every gradient, so with data parallelism a run-to-run / rank-to-rank difference in the
// last bit (atomicAdd order above) makes the replicas' parameters drift apart; this one gives every rank the same bits.
// 4 consecutive gradient elements as fp32: fp32 buffer, or the bf16 wire image of the data-parallel exchange (parallel.py)
__device__ __forceinline__ float4 load_grad4(const float* g, long i) { return *(const float4*)(g + i); }
__device__ __forceinline__ float4 load_grad4(const bf16_t* g, long i) {
  const uint2 w = *(const uint2*)(g + i);
  return make_float4(bflo(w.x), bfhi(w.x), bflo(w.y), bfhi(w.y));
}
__device__ __forceinline__ float load_grad1(const float* g, long i) { return g[i]; }
__device__ __forceinline__ float load_grad1(const bf16_t* g, long i) { return bf2f(g[i]); }

// (four 16-B units per lane and iteration in flight; plain loads: the gradient was written a moment ago and is read again by AdamW)
__device__ __forceinline__ float4 load_grad4_stream(const float* g, long i) { return load_grad4(g, i); }
__device__ __forceinline__ float4 load_grad4_stream(const bf16_t* g, long i) { return load_grad4(g, i); }

template <typename GT>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const GT* __restrict__ g, long n, float* __restrict__ partials) {
  __shared__ float sh[4];
  float s = 0.f;
  const long stride = (long)gridDim.x * 4096;
  for (long i0 = (long)blockIdx.x * 4096 + threadIdx.x * 4; i0 < n; i0 += stride) {
    if (i0 + 3 * 1024 + 3 < n) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = load_grad4_stream(g, i0 + u * 1024);
#pragma unroll
      for (int u = 0; u < 4; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
    } else {
      for (int u = 0; u < 4; ++u) {
        const long i = i0 + u * 1024;
        if (i + 3 < n) {
          const float4 v = load_grad4(g, i);
          s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
          for (long k = i; k < n && k < i + 4; ++k) { const float x = load_grad1(g, k); s += x * x; }
        }
      }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
  __shared__ float sh[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out += (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// One element of the update, with every rounding spelled out: the flat kernel and the ranges kernel of the sharded optimizer must give
// the SAME BITS for the same element (a rank count must not change the trajectory), and under -ffp-contract=fast two differently shaped
// loops around the same expression were contracted into different FMA patterns.
//   g' = coef g ; m = b1 m + (1 - b1) g' ; v = b2 v + (1 - b2) g'^2 ; p -= step_size m / (sqrt(v) + eps) ; p -= lr wd p
__device__ __forceinline__ void adamw_scalars(const VlbAdamState* st, float grad_scale, float& coef, float& step_size) {
  const float step = st->step + 1.0f;
  coef = grad_scale;
  if (st->max_norm > 0.f) {      // clip_grad_norm_: min(max_norm / (total + 1e-6), 1)
    const float total = __fmul_rn(__fsqrt_rn(st->sumsq), grad_scale);
    coef = __fmul_rn(coef, fminf(__fdiv_rn(st->max_norm, __fadd_rn(total, 1e-6f)), 1.0f));
  }
  step_size = __fdiv_rn(__fmul_rn(st->lr, __fsqrt_rn(1.0f - powf(st->beta2, step))), 1.0f - powf(st->beta1, step));      // bias correction
}

__device__ __forceinline__ void adamw_update(float& p, float g, float& m, float& v, float coef, float b1, float b2, float eps,
                                             float step_size, float lr_wd) {
  const float gg = __fmul_rn(g, coef);
  m = __fmaf_rn(m, b1, __fmul_rn(1.0f - b1, gg));
  v = __fmaf_rn(v, b2, __fmul_rn(__fmul_rn(1.0f - b2, gg), gg));
  const float denom = __fadd_rn(__fsqrt_rn(v), eps);
  p = __fmaf_rn(-step_size, __fdiv_rn(m, denom), p);
  if (lr_wd > 0.f) p = __fmaf_rn(-lr_wd, p, p);
}

// One-shot grid: every lane owns ADAMW_UNITS 16-B units per stream (units 4 KiB apart, so a block touches ADAMW_UNITS consecutive 4-KiB
// lines of every stream) and all its loads are in flight before the first use.  Measured on MI355X (tools/hbm_stream_probe.hip, the same
// stream mix on 115 M elements): 6.43 TB/s in this form against 5.88 for the 2048-block grid-stride loop this kernel used to be (writes
// in particular: 6.6 vs 4.5 TB/s for a pure fill) -- profiles/r04_hbm_stream_probe.txt.
constexpr int ADAMW_UNITS = 2;

template <typename GT>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const GT* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p16, long n, VlbAdamState* __restrict__ st,
                                                    float grad_scale) {
  const float lr = st->lr, b1 = st->beta1, b2 = st->beta2, eps = st->eps, wd = st->weight_decay;
  float coef, step_size;
  adamw_scalars(st, grad_scale, coef, step_size);
  const long i0 = ((long)blockIdx.x * (256 * ADAMW_UNITS) + threadIdx.x) * 4;
  float pv[ADAMW_UNITS][4], gv[ADAMW_UNITS][4], mv[ADAMW_UNITS][4], vv[ADAMW_UNITS][4];
#pragma unroll
  for (int u = 0; u < ADAMW_UNITS; ++u) {
    const long i = i0 + (long)u * 1024;
    if (i + 3 < n) {      // p, m, v are touched once per step: streaming accesses (they would only evict the weights the next forward reads).
      // The master weights are loaded PLAIN: broadcasts, checkpoint loads and the replicated-fp32 exchange also write them, and a
      // non-temporal load has once returned stale data behind a writer that was not one of this library's kernels
      const float4 a = *(const float4*)(p + i), b = load_grad4(g, i), c = vlb_load_nt((const float4*)(m + i)),
                   d = vlb_load_nt((const float4*)(v + i));
      pv[u][0] = a.x; pv[u][1] = a.y; pv[u][2] = a.z; pv[u][3] = a.w;
      gv[u][0] = b.x; gv[u][1] = b.y; gv[u][2] = b.z; gv[u][3] = b.w;
      mv[u][0] = c.x; mv[u][1] = c.y; mv[u][2] = c.z; mv[u][3] = c.w;
      vv[u][0] = d.x; vv[u][1] = d.y; vv[u][2] = d.z; vv[u][3] = d.w;
    } else {
      for (int k = 0; k < 4; ++k) {
        const bool ok = i + k < n;
        pv[u][k] = ok ? p[i + k] : 0.f; gv[u][k] = ok ? load_grad1(g, i + k) : 0.f; mv[u][k] = ok ? m[i + k] : 0.f; vv[u][k] = ok ? v[i + k] : 0.f;
      }
    }
  }
#pragma unroll
  for (int u = 0; u < ADAMW_UNITS; ++u) {
    const long i = i0 + (long)u * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) adamw_update(pv[u][k], gv[u][k], mv[u][k], vv[u][k], coef, b1, b2, eps, step_size, lr * wd);
    if (i + 3 < n) {
      vlb_store_nt((float4*)(p + i), make_float4(pv[u][0], pv[u][1], pv[u][2], pv[u][3]));
      vlb_store_nt((float4*)(m + i), make_float4(mv[u][0], mv[u][1], mv[u][2], mv[u][3]));
      vlb_store_nt((float4*)(v + i), make_float4(vv[u][0], vv[u][1], vv[u][2], vv[u][3]));
      if (p16) *(uint2*)(p16 + i) = make_uint2(pack2bf(pv[u][0], pv[u][1]), pack2bf(pv[u][2], pv[u][3]));
    } else {
      for (int k = 0; k < 4 && i + k < n; ++k) {
        p[i + k] = pv[u][k]; m[i + k] = mv[u][k]; v[i + k] = vv[u][k];
        if (p16) p16[i + k] = f2bf(pv[u][k]);
      }
    }
  }
}

// SGD with momentum, the optimiser of the VCR fine-tuning runs (vcr/function/train.py:124-128: torch.optim.SGD(lr, momentum, weight_decay),
// dampening 0, no Nesterov):  d = coef * g + wd * p ;  buf = momentum * buf + d ;  p -= lr * buf   -- one pass over (p, g, buf),
// the bf16 working copy refreshed in the same pass.  coef = grad_scale x the clip_grad_norm_ factor from the device-resident
// squared norm (common/trainer.py:139-145), so no host round trip sits between the gradient norm and the update.
__global__ __launch_bounds__(256) void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                           bf16_t* __restrict__ p16, long n, float lr, float momentum, float wd,
                                                           const float* __restrict__ sumsq, float max_norm, float grad_scale) {
  float coef = grad_scale;
  if (sumsq && max_norm > 0.f) coef *= fminf(max_norm / (sqrtf(*sumsq) * grad_scale + 1e-6f), 1.0f);
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 a = *(const float4*)(p + i), b = *(const float4*)(g + i), c = *(const float4*)(buf + i);
      float pv[4] = {a.x, a.y, a.z, a.w}, mv[4] = {c.x, c.y, c.z, c.w};
      const float gv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        mv[k] = momentum * mv[k] + fmaf(wd, pv[k], coef * gv[k]);
        pv[k] -= lr * mv[k];
      }
      *(float4*)(p + i) = make_float4(pv[0], pv[1], pv[2], pv[3]);
      *(float4*)(buf + i) = make_float4(mv[0], mv[1], mv[2], mv[3]);
      if (p16) *(uint2*)(p16 + i) = make_uint2(pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3]));
    } else {
      for (long k = i; k < n; ++k) {
        const float mv = momentum * buf[k] + fmaf(wd, p[k], coef * g[k]);
        const float pv = p[k] - lr * mv;
        buf[k] = mv; p[k] = pv;
        if (p16) p16[k] = f2bf(pv);
      }
    }
  }
}

// runs after adamw_kernel in the same stream: step += 1, sumsq = 0 (ready for the next step)
__global__ void adam_advance_kernel(VlbAdamState* st) {
  st->step += 1.0f;
  st->sumsq = 0.f;
}

// Learning-rate schedule evaluated ON THE DEVICE from the step counter, so that a captured hipGraph replays
// with the right lr each step.  The reference steps its scheduler BEFORE optimizer.step() (common/trainer.py:131-135),
// and LambdaLR starts at last_epoch = 0, so optimizer step k (1-based) runs with base_lr * lambda(k) where
// k = steps already taken + 1.  kind: 0 ConstantLRSchedule | 1 WarmupConstantSchedule | 2 WarmupLinearSchedule
// (common/nlp/bert/optimization.py:27-62; "triangle" in pretrain/function/train.py:316-320).
__global__ void lr_schedule_kernel(VlbAdamState* st, int kind, float base_lr, float warmup_steps, float t_total) {
  const float k = st->step + 1.0f;
  float f = 1.0f;
  if (kind != 0) {
    if (k < warmup_steps) f = k / fmaxf(1.0f, warmup_steps);
    else if (kind == 2) f = fmaxf(0.0f, (t_total - k) / fmaxf(1.0f, t_total - warmup_steps));
  }
  st->lr = base_lr * f;
}

// (one-shot grid, two 16-B units per lane in flight: see adamw_kernel)
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
  const long i0 = ((long)blockIdx.x * 512 + threadIdx.x) * 4;
  float4 a[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long i = i0 + u * 1024;
    a[u] = i + 3 < n ? *(const float4*)(in + i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const long i = i0 + u * 1024;
    if (i + 3 < n) {
      *(uint2*)(out + i) = make_uint2(pack2bf(a[u].x, a[u].y), pack2bf(a[u].z, a[u].w));
    } else {
      for (long k = i; k < n; ++k) out[k] = f2bf(in[k]);
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) out[i] = bf2f(in[i]);
}

// seed <- hash(seed) : advances the device-resident dropout seed once per step (graph-replayable)
__global__ void rng_advance_kernel(uint32_t* seed) { *seed = vlb_hash32(*seed + 0x9E3779B9u) | 1u; }

static unsigned adamw_grid(long n) { return (unsigned)((n + 1024L * ADAMW_UNITS - 1) / (1024L * ADAMW_UNITS)); }

static int grid_for(long n4) {
  long b = (n4 + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int vlb_sumsq_f32(const float* g, long n, float* out, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(g && out, "vlb_sumsq_f32: null argument");
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, g, n, out);
  VLB_CHECK_LAUNCH("vlb_sumsq_f32");
  return VLB_OK;
}

extern "C" int vlb_sumsq_f32_det(const float* g, long n, float* partials, int partials_len, float* out, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(g && out && partials && partials_len >= 1, "vlb_sumsq_f32_det: null argument");
  int blocks = grid_for((n + 3) / 4);
  if (blocks > partials_len) blocks = partials_len;
  hipLaunchKernelGGL(sumsq_partial_kernel<float>, dim3(blocks), dim3(256), 0, stream, g, n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partials, blocks, out);
  VLB_CHECK_LAUNCH("vlb_sumsq_f32_det");
  return VLB_OK;
}

// the same on a bf16 gradient image (the wire format of the data-parallel exchange: the reduced gradient is consumed as it
// arrived, no conversion pass back to fp32)
extern "C" int vlb_sumsq_bf16_det(const void* g, long n, float* partials, int partials_len, float* out, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(g && out && partials && partials_len >= 1, "vlb_sumsq_bf16_det: null argument");
  VLB_CHECK_ARG(((uintptr_t)g % 8) == 0, "vlb_sumsq_bf16_det: gradient must be 8-byte aligned");
  int blocks = grid_for((n + 3) / 4);
  if (blocks > partials_len) blocks = partials_len;
  hipLaunchKernelGGL(sumsq_partial_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, (const bf16_t*)g, n, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partials, blocks, out);
  VLB_CHECK_LAUNCH("vlb_sumsq_bf16_det");
  return VLB_OK;
}

// state: device pointer to 8 floats {lr, beta1, beta2, eps, weight_decay, step, max_norm, sumsq}
extern "C" int vlb_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, long n, float* state, float grad_scale,
                              hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(p && g && m && v && state, "vlb_adamw_step: null argument");
  hipLaunchKernelGGL(adamw_kernel<float>, dim3(adamw_grid(n)), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p_bf16, n,
                     (VlbAdamState*)state, grad_scale);
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, stream, (VlbAdamState*)state);
  VLB_CHECK_LAUNCH("vlb_adamw_step");
  return VLB_OK;
}

extern "C" int vlb_sgd_momentum_step(float* p, const float* g, float* momentum_buf, void* p_bf16, long n, float lr, float momentum,
                                     float weight_decay, const float* sumsq, float max_norm, float grad_scale, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(p && g && momentum_buf, "vlb_sgd_momentum_step: null argument");
  VLB_CHECK_ARG(momentum >= 0.f && lr >= 0.f && weight_decay >= 0.f, "vlb_sgd_momentum_step: negative hyper-parameter");
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, p, g, momentum_buf, (bf16_t*)p_bf16, n, lr,
                     momentum, weight_decay, sumsq, max_norm, grad_scale);
  VLB_CHECK_LAUNCH("vlb_sgd_momentum_step");
  return VLB_OK;
}

// vlb_adamw_step with the gradient given as a bf16 image (see vlb_sumsq_bf16_det)
extern "C" int vlb_adamw_step_gbf16(float* p, const void* g_bf16, float* m, float* v, void* p_bf16, long n, float* state,
                                    float grad_scale, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(p && g_bf16 && m && v && state, "vlb_adamw_step_gbf16: null argument");
  VLB_CHECK_ARG(((uintptr_t)g_bf16 % 8) == 0, "vlb_adamw_step_gbf16: gradient must be 8-byte aligned");
  hipLaunchKernelGGL(adamw_kernel<bf16_t>, dim3(adamw_grid(n)), dim3(256), 0, stream, p, (const bf16_t*)g_bf16, m, v,
                     (bf16_t*)p_bf16, n, (VlbAdamState*)state, grad_scale);
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, stream, (VlbAdamState*)state);
  VLB_CHECK_LAUNCH("vlb_adamw_step_gbf16");
  return VLB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Sharded optimizer (data parallel, parallel.GradBuckets mode "sharded"): a rank owns one slice of every gradient bucket; the
// reduce-scatter leaves the reduced slices in a COMPACT gradient image, the clip norm is the all-reduced sum of the ranks'
// partial sums, AdamW touches the owned slices only (1/world of the 3.4 GB the replicated update streams) and writes the bf16
// working copy into a compact image that the all-gather distributes.  One launch each over a table of ranges.
// ranges (device int64): n x {p_start, g_start, length} (element offsets: into p / m / v, into the compact images g / p16c);
// block_start (device int32, n + 1): running count of `chunk`-element blocks.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int range_of_block(const int* __restrict__ block_start, int n, int b) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (b >= block_start[mid]) lo = mid; else hi = mid;
  }
  return lo;
}

template <typename GT>
__global__ __launch_bounds__(256) void sumsq_ranges_partial_kernel(const GT* __restrict__ g, const long* __restrict__ ranges,
                                                                   const int* __restrict__ block_start, int n, int chunk,
                                                                   float* __restrict__ partials) {
  __shared__ float sh[4];
  const int r = range_of_block(block_start, n, blockIdx.x);
  const long g0 = ranges[3 * r + 1], len = ranges[3 * r + 2];
  const long base = (long)(blockIdx.x - block_start[r]) * chunk;
  const long end = min(base + (long)chunk, len);
  float s = 0.f;
  for (long i = base + threadIdx.x * 4; i < end; i += 1024) {
    if (i + 3 < end) {
      const float4 v = load_grad4(g + g0, i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (long k = i; k < end; ++k) { const float x = load_grad1(g + g0, k); s += x * x; }
    }
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename GT>
__global__ __launch_bounds__(256) void adamw_ranges_kernel(float* __restrict__ p, const GT* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, bf16_t* __restrict__ p16c,
                                                           const long* __restrict__ ranges, const int* __restrict__ block_start, int n,
                                                           int chunk, const VlbAdamState* __restrict__ st, float grad_scale) {
  const float lr = st->lr, b1 = st->beta1, b2 = st->beta2, eps = st->eps, wd = st->weight_decay;
  float coef, step_size;
  adamw_scalars(st, grad_scale, coef, step_size);
  const int r = range_of_block(block_start, n, blockIdx.x);
  const long p0 = ranges[3 * r], g0 = ranges[3 * r + 1], len = ranges[3 * r + 2];
  const long base = (long)(blockIdx.x - block_start[r]) * chunk;
  const long end = min(base + (long)chunk, len);
  for (long i = base + threadIdx.x * 4; i < end; i += 1024) {
    const int cnt = (int)min(4L, end - i);      // (range starts / lengths are multiples of 4: cnt == 4 except for a ragged last range)
    float pv[4], gv[4], mv[4], vv[4];
    if (cnt == 4) {
      const float4 a = *(const float4*)(p + p0 + i), b = load_grad4(g + g0, i), c = vlb_load_nt((const float4*)(m + p0 + i)),
                   d = vlb_load_nt((const float4*)(v + p0 + i));      // (streaming accesses for the moments, plain for the master: as in adamw_kernel)
      pv[0] = a.x; pv[1] = a.y; pv[2] = a.z; pv[3] = a.w;
      gv[0] = b.x; gv[1] = b.y; gv[2] = b.z; gv[3] = b.w;
      mv[0] = c.x; mv[1] = c.y; mv[2] = c.z; mv[3] = c.w;
      vv[0] = d.x; vv[1] = d.y; vv[2] = d.z; vv[3] = d.w;
    } else {
      for (int k = 0; k < 4; ++k) {
        const bool ok = k < cnt;
        pv[k] = ok ? p[p0 + i + k] : 0.f; gv[k] = ok ? load_grad1(g + g0, i + k) : 0.f;
        mv[k] = ok ? m[p0 + i + k] : 0.f; vv[k] = ok ? v[p0 + i + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) adamw_update(pv[k], gv[k], mv[k], vv[k], coef, b1, b2, eps, step_size, lr * wd);
    if (cnt == 4) {
      vlb_store_nt((float4*)(p + p0 + i), make_float4(pv[0], pv[1], pv[2], pv[3]));
      vlb_store_nt((float4*)(m + p0 + i), make_float4(mv[0], mv[1], mv[2], mv[3]));
      vlb_store_nt((float4*)(v + p0 + i), make_float4(vv[0], vv[1], vv[2], vv[3]));
      if (p16c) *(uint2*)(p16c + g0 + i) = make_uint2(pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3]));
    } else {
      for (int k = 0; k < cnt; ++k) {
        p[p0 + i + k] = pv[k]; m[p0 + i + k] = mv[k]; v[p0 + i + k] = vv[k];
        if (p16c) p16c[g0 + i + k] = f2bf(pv[k]);
      }
    }
  }
}

// out += sum over the ranges of g^2 (fixed summation order; g = the compact reduced-gradient image, fp32 or bf16)
extern "C" int vlb_sumsq_ranges_det(const void* g, int g_is_bf16, const int64_t* ranges, const int32_t* block_start, int n, int total_blocks,
                                    int chunk, float* partials, int partials_len, float* out, hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0) return VLB_OK;
  VLB_CHECK_ARG(g && ranges && block_start && partials && out, "vlb_sumsq_ranges_det: null argument");
  VLB_CHECK_ARG(chunk >= 1024 && (chunk % 1024) == 0, "vlb_sumsq_ranges_det: chunk must be a positive multiple of 1024");
  VLB_CHECK_ARG(total_blocks <= partials_len, "vlb_sumsq_ranges_det: %d blocks need %d partial sums (workspace holds %d)", total_blocks,
                total_blocks, partials_len);
  VLB_CHECK_ARG(((uintptr_t)g % 16) == 0, "vlb_sumsq_ranges_det: gradient image must be 16-byte aligned");
  if (g_is_bf16)
    hipLaunchKernelGGL(sumsq_ranges_partial_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, stream, (const bf16_t*)g, (const long*)ranges,
                       (const int*)block_start, n, chunk, partials);
  else
    hipLaunchKernelGGL(sumsq_ranges_partial_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, (const float*)g, (const long*)ranges,
                       (const int*)block_start, n, chunk, partials);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, partials, total_blocks, out);
  VLB_CHECK_LAUNCH("vlb_sumsq_ranges_det");
  return VLB_OK;
}

// AdamW over the ranges (see vlb_adamw_step for the arithmetic); p_bf16_compact (nullable) receives the bf16 copy of the updated
// parameters at the ranges' g_start offsets; then step += 1 and sumsq = 0.
extern "C" int vlb_adamw_step_ranges(float* p, const void* g, int g_is_bf16, float* m, float* v, void* p_bf16_compact,
                                     const int64_t* ranges, const int32_t* block_start, int n, int total_blocks, int chunk, float* state,
                                     float grad_scale, hipStream_t stream) {
  VLB_CHECK_ARG(state, "vlb_adamw_step_ranges: null state");
  if (n > 0 && total_blocks > 0) {
    VLB_CHECK_ARG(p && g && m && v && ranges && block_start, "vlb_adamw_step_ranges: null argument");
    VLB_CHECK_ARG(chunk >= 1024 && (chunk % 1024) == 0, "vlb_adamw_step_ranges: chunk must be a positive multiple of 1024");
    VLB_CHECK_ARG(((uintptr_t)g % 16) == 0 && ((uintptr_t)p % 16) == 0, "vlb_adamw_step_ranges: buffers must be 16-byte aligned");
    if (g_is_bf16)
      hipLaunchKernelGGL(adamw_ranges_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, stream, p, (const bf16_t*)g, m, v,
                         (bf16_t*)p_bf16_compact, (const long*)ranges, (const int*)block_start, n, chunk, (const VlbAdamState*)state, grad_scale);
    else
      hipLaunchKernelGGL(adamw_ranges_kernel<float>, dim3(total_blocks), dim3(256), 0, stream, p, (const float*)g, m, v,
                         (bf16_t*)p_bf16_compact, (const long*)ranges, (const int*)block_start, n, chunk, (const VlbAdamState*)state, grad_scale);
  }
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, stream, (VlbAdamState*)state);
  VLB_CHECK_LAUNCH("vlb_adamw_step_ranges");
  return VLB_OK;
}

extern "C" int vlb_lr_schedule_step(float* state, int kind, float base_lr, float warmup_steps, float t_total, hipStream_t stream) {
  VLB_CHECK_ARG(state, "vlb_lr_schedule_step: null state");
  VLB_CHECK_ARG(kind >= 0 && kind <= 2, "vlb_lr_schedule_step: kind must be 0 (constant), 1 (warmup-constant) or 2 (warmup-linear)");
  VLB_CHECK_ARG(base_lr >= 0.f && warmup_steps >= 0.f, "vlb_lr_schedule_step: negative base_lr / warmup_steps");
  hipLaunchKernelGGL(lr_schedule_kernel, dim3(1), dim3(1), 0, stream, (VlbAdamState*)state, kind, base_lr, warmup_steps, t_total);
  VLB_CHECK_LAUNCH("vlb_lr_schedule_step");
  return VLB_OK;
}

extern "C" int vlb_cast_f32_bf16(const float* in, void* out, long n, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, stream, in, (bf16_t*)out, n);
  VLB_CHECK_LAUNCH("vlb_cast_f32_bf16");
  return VLB_OK;
}

extern "C" int vlb_cast_bf16_f32(const void* in, float* out, long n, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, (const bf16_t*)in, out, n);
  VLB_CHECK_LAUNCH("vlb_cast_bf16_f32");
  return VLB_OK;
}

// x *= alpha (the 1 / world average of an all-reduced gradient that leaves through autograd's .grad tensors: the DistributedDataParallel
// mirror of parallel.py; the engine's own data-parallel step folds the factor into the AdamW kernel instead)
__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ x, long n, float alpha) {
  const long stride = (long)gridDim.x * 256 * 4;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n && ((uintptr_t)(x + i) % 16) == 0) {
      float4 v = *(float4*)(x + i);
      v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
      *(float4*)(x + i) = v;
    } else {
      for (long k = i; k < n && k < i + 4; ++k) x[k] *= alpha;
    }
  }
}

extern "C" int vlb_scale_f32(float* x, long n, float alpha, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(x, "vlb_scale_f32: null argument");
  hipLaunchKernelGGL(scale_f32_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, stream, x, n, alpha);
  VLB_CHECK_LAUNCH("vlb_scale_f32");
  return VLB_OK;
}

extern "C" int vlb_rng_advance(uint32_t* seed, hipStream_t stream) {
  VLB_CHECK_ARG(seed, "vlb_rng_advance: null seed");
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, stream, seed);
  VLB_CHECK_LAUNCH("vlb_rng_advance");
  return VLB_OK;
}

// Zero several ranges of one fp32 buffer in a single launch (the gradients that are ACCUMULATED by atomics -- biases,
// LayerNorm parameters, small embedding tables -- while the GEMM weight gradients are overwritten by their producer).
// ranges (device int64): n x {start, length}; block_start (device int32, n+1): running count of 1024-float blocks.
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long* __restrict__ ranges,
                                                          const int* __restrict__ block_start, int n) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= block_start[mid]) lo = mid; else hi = mid;
  }
  const long start = ranges[2 * lo], len = ranges[2 * lo + 1];
  const long i = (long)(blockIdx.x - block_start[lo]) * 1024 + threadIdx.x * 4;
  float* q = base + start + i;
  if (i + 3 < len && ((uintptr_t)q % 16) == 0) {
    *(float4*)q = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int k = 0; k < 4 && i + k < len; ++k) q[k] = 0.f;
  }
}

extern "C" int vlb_zero_ranges_f32(float* base, const int64_t* ranges, const int32_t* block_start, int n, int total_blocks,
                                   hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0) return VLB_OK;
  VLB_CHECK_ARG(base && ranges && block_start, "vlb_zero_ranges_f32: null argument");
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(total_blocks), dim3(256), 0, stream, base, (const long*)ranges, (const int*)block_start, n);
  VLB_CHECK_LAUNCH("vlb_zero_ranges_f32");
  return VLB_OK;
}

// Copy several ranges between two fp32 buffers in a single launch: dst[dst_start + i] = src[src_start + i].
// Sharded data-parallel optimizer (parallel.GradBuckets.gather_params): the tensors the compute path reads as fp32 on EVERY rank -- Linear
// biases, LayerNorm gamma / beta, the mask embedding: ~0.15 % of the flat buffer -- are packed from the owner's master slices into one
// compact image, summed over the ranks (exactly one rank contributes a non-zero value per element) and unpacked into every rank's master.
// ranges (device int64): n x {src_start, dst_start, length}; block_start (device int32, n+1): running count of 1024-float blocks.
__global__ __launch_bounds__(256) void copy_ranges_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const long* __restrict__ ranges, const int* __restrict__ block_start, int n) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= block_start[mid]) lo = mid; else hi = mid;
  }
  const long s0 = ranges[3 * lo], d0 = ranges[3 * lo + 1], len = ranges[3 * lo + 2];
  const long i = (long)(blockIdx.x - block_start[lo]) * 1024 + threadIdx.x * 4;
  const float* q = src + s0 + i;
  float* w = dst + d0 + i;
  if (i + 3 < len && ((uintptr_t)q % 16) == 0 && ((uintptr_t)w % 16) == 0) {
    *(float4*)w = *(const float4*)q;
  } else {
    for (int k = 0; k < 4 && i + k < len; ++k) w[k] = q[k];
  }
}

extern "C" int vlb_copy_ranges_f32(const float* src, float* dst, const int64_t* ranges, const int32_t* block_start, int n,
                                   int total_blocks, hipStream_t stream) {
  if (n <= 0 || total_blocks <= 0) return VLB_OK;
  VLB_CHECK_ARG(src && dst && ranges && block_start, "vlb_copy_ranges_f32: null argument");
  hipLaunchKernelGGL(copy_ranges_kernel, dim3(total_blocks), dim3(256), 0, stream, src, dst, (const long*)ranges, (const int*)block_start, n);
  VLB_CHECK_LAUNCH("vlb_copy_ranges_f32");
  return VLB_OK;
}
