// Loss kernels of the VL-BERT pre-training heads (gfx950; HBM-bound row reductions).
//   * MLM: F.cross_entropy(logits[B*T, V], labels, ignore_index=-1), mean over labelled rows
//          (pretrain/modules/resnet_vlbert_for_pretraining.py:176-178)
//   * MVRC: soft_cross_entropy (common/utils/misc.py:124-151): rows are valid iff
//          |sum(target) - 1| < 0.1; loss = mean_valid( -sum_c log_softmax(x)_c * t_c )
// Forward and backward are fused: one pass computes the row's log-sum-exp (online max/sum), the
// second writes d(loss)/d(logits) IN PLACE over the bf16 logits (scaled by 1/n_valid * gscale).
// A copy of the logits is only kept when the caller asks for one (API parity / metrics).
// n_valid is produced on the device by the count kernels -> no host synchronisation
// (the reference does `.item()` at misc.py:140).
#include "vlb_common.h"

__device__ __forceinline__ void online_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  if (mn == -INFINITY) return;  // both empty
  s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
  m = mn;
}

// block-wide (256 threads) reduction of an online-softmax (max,sum) pair; result broadcast
__device__ __forceinline__ void block_online_reduce(float& m, float& s, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    online_merge(m, s, m2, s2);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) {
    sh[wave * 2] = m;
    sh[wave * 2 + 1] = s;
  }
  __syncthreads();
  m = sh[0];
  s = sh[1];
#pragma unroll
  for (int w = 1; w < 4; ++w) online_merge(m, s, sh[w * 2], sh[w * 2 + 1]);
}

__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// counts[0] = #rows with label >= 0
__global__ void count_labels_kernel(const int64_t* __restrict__ labels, int n, float* __restrict__ counts) {
  int c = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) c += (labels[i] >= 0);
  float f = wave_sum((float)c);
  if ((threadIdx.x & 63) == 0 && f != 0.f) atomicAdd(counts, f);
}

// One block per row.  logits: bf16 [rows, ld] (columns >= V are padding and are zeroed).
// out: loss_sum += row_loss / n_valid ;  logits <- dlogits * (gscale / n_valid).
// Two-group form (compacted MLM rows of the multitask wrapper): rows [0, *n_valid) belong to group 0 (mean over n_valid ->
// loss_out), the rows behind them to group 1 (mean over *n_valid2 -> loss_out2); n_valid2 == nullptr: one group.
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(bf16_t* __restrict__ logits, long ld, int V, const int64_t* __restrict__ labels,
                                                         const float* __restrict__ n_valid, float gscale, float* __restrict__ loss_out,
                                                         bf16_t* __restrict__ logits_copy, long ldcopy,
                                                         const float* __restrict__ n_valid2 = nullptr, float* __restrict__ loss_out2 = nullptr) {
  __shared__ float sh[16];
  const int row = blockIdx.x;
  if (n_valid2 && (float)row >= *n_valid) {
    n_valid = n_valid2;
    loss_out = loss_out2;
  }
  bf16_t* x = logits + (long)row * ld;
  const long label = labels[row];
  const int ldv = (int)ld;
  if (logits_copy) {
    bf16_t* cp = logits_copy + (long)row * ldcopy;
    for (int c = threadIdx.x * 8; c < V; c += 2048) {
      if (c + 8 <= V) *(uint4*)(cp + c) = *(const uint4*)(x + c);
      else for (int k = c; k < V; ++k) cp[k] = x[k];
    }
  }
  if (label < 0 || label >= V) {  // ignore_index: no loss, zero gradient row
    for (int c = threadIdx.x * 8; c < ldv; c += 2048) *(uint4*)(x + c) = make_uint4(0, 0, 0, 0);
    return;
  }
  float m = -INFINITY, s = 0.f;
  for (int c = threadIdx.x * 8; c < V; c += 2048) {
    const uint4 w = *(const uint4*)(x + c);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float a = bflo(ww[k]), b = bfhi(ww[k]);
      if (c + 2 * k < V) online_merge(m, s, a, 1.f);
      if (c + 2 * k + 1 < V) online_merge(m, s, b, 1.f);
    }
  }
  block_online_reduce(m, s, sh);
  const float lse = m + __logf(s);
  const float nv = fmaxf(*n_valid, 1.f);
  if (threadIdx.x == 0) atomicAdd(loss_out, (lse - bf2f(x[label])) / nv);
  const float sc = gscale / nv;
  __syncthreads();  // x[label] read above before anyone overwrites it
  for (int c = threadIdx.x * 8; c < ldv; c += 2048) {
    const uint4 w = *(const uint4*)(x + c);
    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c0 = c + 2 * k, c1 = c0 + 1;
      float g0 = (c0 < V) ? __expf(bflo(ww[k]) - lse) : 0.f;
      float g1 = (c1 < V) ? __expf(bfhi(ww[k]) - lse) : 0.f;
      if (c0 == label) g0 -= 1.f;
      if (c1 == label) g1 -= 1.f;
      o[k] = pack2bf(g0 * sc, g1 * sc);
    }
    *(uint4*)(x + c) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// MLM head compaction.  ~85 % of the text positions carry no label (ignore_index -1): their logits rows are never read by
// the loss and their d(logits) rows are exactly zero, so transform -> LayerNorm -> decoder (modeling.py:439-472) and the whole
// backward of the head only need the LABELLED rows.  This kernel builds the (stable, ascending) list of labelled positions:
//   sel_pos[k]  = position i in [0, n) of the k-th labelled row (-1 beyond the count)
//   sel_src[k]  = src_rows[i]: its row in the packed encoder output (-1 -> zero row)
//   labels_c[k] = its label (-1 beyond the count)
//   counts[0] = labelled rows among positions [0, n_split) (image-caption samples), counts[1] = among [n_split, n) (text-only
//   auxiliary samples of the multitask wrapper); *overflow |= (count > cap): the rows that did not fit are NOT trained on, so the
//   host treats the flag as an error (engine.loss_values) -- capacity is a contract with the data pipeline (masking probability).
// One 1024-thread block: every thread owns a contiguous run of positions, block scan of the run counts.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mlm_compact_kernel(const int64_t* __restrict__ labels, const int32_t* __restrict__ src_rows, int n,
                                                           int n_split, int V, int cap, int32_t* __restrict__ sel_pos,
                                                           int32_t* __restrict__ sel_src, int64_t* __restrict__ labels_c,
                                                           float* __restrict__ count0, float* __restrict__ count1,
                                                           int32_t* __restrict__ overflow) {
  __shared__ int wsum[16];
  __shared__ int total_s, n0_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int per = (n + 1023) / 1024, lo = tid * per, hi = min(n, lo + per);
  int c = 0, c0 = 0;
  for (int i = lo; i < hi; ++i) {
    const bool lab = labels[i] >= 0 && labels[i] < V;
    c += lab;
    c0 += lab && i < n_split;
  }
  // inclusive scan of c across the block
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  const float f0 = wave_sum((float)c0);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < 16; ++w) { const int t = wsum[w]; wsum[w] = run; run += t; }
    total_s = run;
    n0_s = 0;
  }
  __syncthreads();
  if (lane == 0 && f0 != 0.f) atomicAdd(&n0_s, (int)f0);
  __syncthreads();
  int k = wsum[wave] + incl - c;      // exclusive prefix of this thread's run
  for (int i = lo; i < hi; ++i) {
    const long lb = labels[i];
    if (lb >= 0 && lb < V) {
      if (k < cap) {
        sel_pos[k] = i;
        sel_src[k] = src_rows[i];
        labels_c[k] = lb;
      }
      ++k;
    }
  }
  const int total = total_s;
  for (int q = total + tid; q < cap; q += 1024) {   // padding behind the labelled rows
    sel_pos[q] = -1;
    sel_src[q] = -1;
    labels_c[q] = -1;
  }
  if (tid == 0) {
    const int kept = min(total, cap), n0 = min(n0_s, kept);
    *count0 = (float)n0;
    *count1 = (float)(kept - n0);
    if (total > cap) *overflow = 1;
  }
}

extern "C" int vlb_mlm_compact(const int64_t* labels, const int32_t* src_rows, int n, int n_split, int V, int cap, int32_t* sel_pos,
                               int32_t* sel_src, int64_t* labels_c, float* count0, float* count1, int32_t* overflow, hipStream_t stream) {
  VLB_CHECK_ARG(labels && src_rows && sel_pos && sel_src && labels_c && count0 && count1 && overflow, "vlb_mlm_compact: null argument");
  VLB_CHECK_ARG(n > 0 && cap > 0 && n_split >= 0 && n_split <= n, "vlb_mlm_compact: bad sizes");
  hipLaunchKernelGGL(mlm_compact_kernel, dim3(1), dim3(1024), 0, stream, labels, src_rows, n, n_split, V, cap, sel_pos, sel_src, labels_c,
                     count0, count1, overflow);
  VLB_CHECK_LAUNCH("vlb_mlm_compact");
  return VLB_OK;
}

// CE forward + backward on COMPACTED rows: counts are given (vlb_mlm_compact wrote them), two groups with their own means.
extern "C" int vlb_ce_fwd_bwd_compact(void* logits, long ld, int rows, int V, const int64_t* labels_c, const float* count0,
                                      const float* count1, float gscale, float* loss_out0, float* loss_out1, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(logits && labels_c && count0 && count1 && loss_out0 && loss_out1, "vlb_ce_fwd_bwd_compact: null argument");
  VLB_CHECK_ARG(ld >= V && (ld % 8) == 0, "vlb_ce_fwd_bwd_compact: ld=%ld must be >= V=%d and a multiple of 8", ld, V);
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(rows), dim3(256), 0, stream, (bf16_t*)logits, ld, V, labels_c, count0, gscale, loss_out0,
                     (bf16_t*)nullptr, 0L, count1, loss_out1);
  VLB_CHECK_LAUNCH("vlb_ce_fwd_bwd_compact");
  return VLB_OK;
}

// row validity for the soft-label loss: valid[r] = |sum_c t[r,c] - 1| < 0.1 ; counts[0] += #valid
__global__ __launch_bounds__(256) void soft_valid_kernel(const float* __restrict__ target, long ldt, int C, int rows, float* __restrict__ tsum,
                                                         float* __restrict__ counts) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* t = target + (long)row * ldt;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += t[c];
  s = wave_sum(s);
  if (lane == 0) {
    tsum[row] = s;
    if (fabsf(s - 1.f) < 0.1f) atomicAdd(counts, 1.f);
  }
}

// One block per row.  logits bf16 [rows, ld] -> in place d(logits); target fp32 [rows, ldt].
//   loss_row = lse * sum(t) - sum(t * x);  dlogit_c = (softmax_c * sum(t) - t_c) / n_valid
__global__ __launch_bounds__(256) void soft_ce_fwd_bwd_kernel(bf16_t* __restrict__ logits, long ld, int C, const float* __restrict__ target,
                                                              long ldt, const float* __restrict__ tsum, const float* __restrict__ n_valid,
                                                              float gscale, float* __restrict__ loss_out, bf16_t* __restrict__ logits_copy,
                                                              long ldcopy) {
  __shared__ float sh[16];
  const int row = blockIdx.x;
  bf16_t* x = logits + (long)row * ld;
  const float* t = target + (long)row * ldt;
  const int ldv = (int)ld;
  if (logits_copy) {
    bf16_t* cp = logits_copy + (long)row * ldcopy;
    for (int c = threadIdx.x; c < C; c += 256) cp[c] = x[c];
  }
  const float ts = tsum[row];
  if (!(fabsf(ts - 1.f) < 0.1f)) {
    for (int c = threadIdx.x; c < ldv; c += 256) x[c] = 0;
    return;
  }
  float m = -INFINITY, s = 0.f, dot = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float v = bf2f(x[c]);
    online_merge(m, s, v, 1.f);
    dot += t[c] * v;
  }
  block_online_reduce(m, s, sh);
  dot = block_sum(dot, sh + 8);
  const float lse = m + __logf(s);
  const float nv = fmaxf(*n_valid, 1.f);
  if (threadIdx.x == 0) atomicAdd(loss_out, (lse * ts - dot) / nv);
  const float sc = gscale / nv;
  for (int c = threadIdx.x; c < ldv; c += 256) {
    float g = 0.f;
    if (c < C) g = (__expf(bf2f(x[c]) - lse) * ts - t[c]) * sc;
    x[c] = f2bf(g);
  }
}

extern "C" int vlb_ce_fwd_bwd(void* logits, long ld, int rows, int V, const int64_t* labels, float* counts, float gscale,
                              float* loss_out, void* logits_copy, long ldcopy, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(logits && labels && counts && loss_out, "vlb_ce_fwd_bwd: null argument");
  VLB_CHECK_ARG(ld >= V && (ld % 8) == 0, "vlb_ce_fwd_bwd: ld=%ld must be >= V=%d and a multiple of 8", ld, V);
  (void)hipMemsetAsync(counts, 0, sizeof(float), stream);
  hipLaunchKernelGGL(count_labels_kernel, dim3(vlb_cdiv(rows, 256) > 64 ? 64 : vlb_cdiv(rows, 256)), dim3(256), 0, stream, labels, rows,
                     counts);
  hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(rows), dim3(256), 0, stream, (bf16_t*)logits, ld, V, labels, counts, gscale, loss_out,
                     (bf16_t*)logits_copy, ldcopy);
  VLB_CHECK_LAUNCH("vlb_ce_fwd_bwd");
  return VLB_OK;
}

extern "C" int vlb_soft_ce_fwd_bwd(void* logits, long ld, int rows, int C, const float* target, long ldt, float* tsum,
                                   float* counts, float gscale, float* loss_out, void* logits_copy, long ldcopy,
                                   hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(logits && target && tsum && counts && loss_out, "vlb_soft_ce_fwd_bwd: null argument");
  VLB_CHECK_ARG(ld >= C && ldt >= C, "vlb_soft_ce_fwd_bwd: bad leading dimensions");
  (void)hipMemsetAsync(counts, 0, sizeof(float), stream);
  hipLaunchKernelGGL(soft_valid_kernel, dim3(vlb_cdiv(rows, 4)), dim3(256), 0, stream, target, ldt, C, rows, tsum, counts);
  hipLaunchKernelGGL(soft_ce_fwd_bwd_kernel, dim3(rows), dim3(256), 0, stream, (bf16_t*)logits, ld, C, target, ldt, tsum, counts,
                     gscale, loss_out, (bf16_t*)logits_copy, ldcopy);
  VLB_CHECK_LAUNCH("vlb_soft_ce_fwd_bwd");
  return VLB_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// VQA answer loss (vqa/modules/resnet_vlbert_for_vqa.py:226): F.binary_cross_entropy_with_logits(logits[B,A], label) * A,
// i.e. (1/B) sum_b sum_a ( max(x,0) - x y + log(1 + exp(-|x|)) ); d/dx = (sigmoid(x) - y) / B.  Forward and backward fused like
// the CE kernels: logits (bf16 [rows, ld], columns >= A are padding and are zeroed) are overwritten by gscale * d(loss)/dx, the
// untouched logits are copied out when asked for (label_logits of the reference's outputs dict).  One block per row.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_logits_fwd_bwd_kernel(bf16_t* __restrict__ logits, long ld, int A, const float* __restrict__ label,
                                                                 long ldl, int rows, float gscale, float pos_weight,
                                                                 float* __restrict__ loss_out, bf16_t* __restrict__ logits_copy, long ldcopy) {
  __shared__ float sh[4];
  const int row = blockIdx.x;
  bf16_t* x = logits + (long)row * ld;
  const float* y = label + (long)row * ldl;
  const float inv_rows = 1.0f / (float)rows;
  float acc = 0.f;
  for (int a = threadIdx.x; a < (int)ld; a += 256) {
    if (a < A) {
      const float v = bf2f(x[a]), t = y[a];
      if (logits_copy) logits_copy[(long)row * ldcopy + a] = x[a];
      const float e = __expf(-fabsf(v));
      const float w = t > 0.5f ? pos_weight : 1.0f;       // element weight (the `weight=` tensor of vcr/modules/resnet_vlbert_for_vcr.py:336-339)
      acc += w * (fmaxf(v, 0.f) - v * t + log1pf(e));
      const float sig = v >= 0.f ? 1.0f / (1.0f + e) : e / (1.0f + e);
      x[a] = f2bf(w * (sig - t) * inv_rows * gscale);
    } else {
      x[a] = 0;
      if (logits_copy && a < (int)ldcopy) logits_copy[(long)row * ldcopy + a] = 0;
    }
  }
  const float s = block_sum(acc, sh);
  if (threadIdx.x == 0) atomicAdd(loss_out, s * inv_rows);
}

extern "C" int vlb_bce_logits_fwd_bwd(void* logits, long ld, int rows, int A, const float* label, long ldl, float gscale, float pos_weight,
                                      float* loss_out, void* logits_copy, long ldcopy, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(logits && label && loss_out && A > 0 && ld >= A && ldl >= A, "vlb_bce_logits_fwd_bwd: bad argument");
  VLB_CHECK_ARG(!logits_copy || ldcopy >= A, "vlb_bce_logits_fwd_bwd: ldcopy too small");
  hipLaunchKernelGGL(bce_logits_fwd_bwd_kernel, dim3(rows), dim3(256), 0, stream, (bf16_t*)logits, ld, A, label, ldl, rows, gscale,
                     pos_weight, loss_out, (bf16_t*)logits_copy, ldcopy);
  VLB_CHECK_LAUNCH("vlb_bce_logits_fwd_bwd");
  return VLB_OK;
}

// y = x * keep(idx) / (1 - p) with the counter RNG (element index = position in the [n] array); the same call on dy is the backward.
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long n, uint32_t drop_thr,
                                                           float drop_scale, const uint32_t* __restrict__ seedp, uint32_t tag) {
  const uint32_t seed = (drop_thr && seedp) ? *seedp : 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = bf2f(x[i]);
    y[i] = f2bf((!drop_thr || vlb_keep(seed, tag, (uint32_t)i, drop_thr)) ? v * (drop_thr ? drop_scale : 1.0f) : 0.f);
  }
}

extern "C" int vlb_dropout_bf16(const void* x, void* y, long n, float drop_p, const uint32_t* seed, uint32_t tag, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(x && y && n < (1L << 32), "vlb_dropout_bf16: bad argument");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_dropout_bf16: dropout needs a device seed pointer");
  const uint32_t thr = vlb_drop_thr(drop_p);
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3((int)blocks), dim3(256), 0, stream, (const bf16_t*)x, (bf16_t*)y, n, thr, vlb_drop_scale(thr),
                     seed, tag);
  VLB_CHECK_LAUNCH("vlb_dropout_bf16");
  return VLB_OK;
}
