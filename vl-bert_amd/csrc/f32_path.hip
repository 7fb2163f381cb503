// fp32 compute path of the VL-BERT ENCODER for gfx950 (CDNA4): every tensor of an encoder layer -- GEMM operands and results,
// LayerNorm, softmax, the residual stream, weights (the fp32 masters themselves) and gradients -- in fp32.
//
// What it is for: the reference's fp32 configurations (`TRAIN.FP16: false`: cfgs/pretrain/base_prec_4x16G_fp32.yaml:112,
// cfgs/vqa/large_4x16G_fp32.yaml:108) and north_star's 1e-3 fp32 tolerance.  Replaces, per encoder layer,
// external/pytorch_pretrained_bert/modeling.py:268-397 (BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput) in
// fp32; the host side is vl-bert_amd/encoder_f32.py.
//
// MI355X-first design of the fp32 GEMM: gfx950's native fp32 MFMA peaks at 157 TFLOP/s, 1/16 of the bf16 matrix rate.  An fp32
// product does not need it: split each operand element x = h + m + r with h = bf16(x), m = bf16(x - h) (|r| <= 2^-17 |x|; bf16
// keeps fp32's exponent, so there is no range problem, unlike an fp16 split) and take
//        a * b  ~=  ah*bh + ah*bm + am*bh            (3 bf16 MFMAs, fp32 accumulation, relative error ~2^-16 per product
//                                                      before the usual sqrt(K) averaging -- 1e-5 class, 100x inside the 1e-3 bar)
// The split happens ONCE per operand tile, on the way from global memory to LDS (the tile is then read by every wave of the
// workgroup), so the matrix pipe runs 3 bf16 instructions per fp32 product: an effective peak of 833 TFLOP/s instead of 157.
// The kernel is bound by the fp32 operand stream (4 B per element through L2 -> registers -> LDS), not by the MFMAs.
//
// Kernels: one batched "NT" GEMM C = epilogue(alpha * A . B^T) (linear layers forward / dgrad / wgrad on transposed operands,
// Q.K^T, P.V, and the attention backward products), a batched transpose (zero-padded, optional column sums = bias gradients),
// LayerNorm forward / backward, masked softmax (+ dropout) forward / backward.  Straightforward tiling (128 x 128 x 32, 4 waves,
// register-staged double buffering): this path exists for precision; the bf16 / fp16 builds are the fast ones.
#include <math.h>

#include <type_traits>

#include "vlb_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 tbf16x8;      // true bfloat16 whatever the library's 16-bit type is

__device__ __forceinline__ uint32_t pack2_true_bf16(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  const b2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float true_bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float true_bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

struct F32Gemm {
  const float* A; const float* B; float* C;
  long lda, ldb, ldc;
  int M, N, K;                       // K % 32 == 0, N % 4 == 0
  int nb2;                           // batch index z -> (i1 = z / nb2, i2 = z % nb2)
  long sA1, sA2, sB1, sB2, sC1, sC2; // element strides of the two batch levels
  const float* bias; long sBias1;    // bias[n] (+ i1 * sBias1): per-sample additive attention mask rows, or a Linear bias
  float alpha;
  int epi;                           // 0 none | 1 gelu (gelu' -> pre) | 2 relu | 3 x aux | 4 tanh | 5 x (aux > 0)
  const float* aux; long ldaux; float* pre; long ldpre;
  const float* res; long ldres;      // + res (after the dropout)
  uint32_t drop_thr; float drop_scale; const uint32_t* seed; uint32_t tag;
  int atomic;                        // C += (atomicAdd): split-K weight gradients / accumulation
  int k_per_split;
};

constexpr int BN = 128, BK = 32;      // BM = 32 * FM rows (FM = 4: 128-row tiles; FM = 2: 64-row tiles for launches that would leave CUs idle)

// fp32 tile rows -> (h, m) bf16 planes in LDS.  Row image: 128 B = 8 chunks of 16 B: logical chunks 0-3 = h of k [8c, 8c+8),
// 4-7 = m of the same k; physical slot = logical ^ ((row >> 1) & 7) (conflict-free ds_read_b128 of 16 rows, as in gemm.hip).
__device__ __forceinline__ void split_store(char* tile, int row, int kc, const float4& x0, const float4& x1) {
  const uint32_t h0 = pack2_true_bf16(x0.x, x0.y), h1 = pack2_true_bf16(x0.z, x0.w), h2 = pack2_true_bf16(x1.x, x1.y), h3 = pack2_true_bf16(x1.z, x1.w);
  const uint32_t m0 = pack2_true_bf16(x0.x - true_bf_lo(h0), x0.y - true_bf_hi(h0)), m1 = pack2_true_bf16(x0.z - true_bf_lo(h1), x0.w - true_bf_hi(h1));
  const uint32_t m2 = pack2_true_bf16(x1.x - true_bf_lo(h2), x1.y - true_bf_hi(h2)), m3 = pack2_true_bf16(x1.z - true_bf_lo(h3), x1.w - true_bf_hi(h3));
  const int sw = (row >> 1) & 7;
  *(uint4*)(tile + row * 128 + ((kc ^ sw) << 4)) = make_uint4(h0, h1, h2, h3);
  *(uint4*)(tile + row * 128 + (((kc + 4) ^ sw) << 4)) = make_uint4(m0, m1, m2, m3);
}

template <int FM>
__global__ __launch_bounds__(256, 2) void gemm_f32_split_kernel(const F32Gemm p) {
  constexpr int BM = 32 * FM, NA = FM / 2;                                       // NA staging items of A per thread
  __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * 128];      // 2 stages x (A 16 / 8 KiB + B 16 KiB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                                       // 2 x 2 waves of (16 FM) x 64
  const int ntm = (p.M + BM - 1) / BM;
  const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;                        // consecutive blocks share the B panel
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.z, i1 = z / p.nb2, i2 = z - i1 * p.nb2;
  const float* A = p.A + i1 * p.sA1 + i2 * p.sA2;
  const float* B = p.B + i1 * p.sB1 + i2 * p.sB2;
  float* C = p.C + i1 * p.sC1 + i2 * p.sC2;
  const int k_begin = blockIdx.y * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int ntk = (k_end - k_begin) / BK;
  if (ntk <= 0) return;

  // staging: item P = it * 256 + tid (it = 0, 1): row = P >> 2, k-chunk kc = P & 3 (8 consecutive k = 32 B of the row)
  const float* a_src[2];
  const float* b_src[2];
  int s_row[2], s_kc[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int P = it * 256 + tid;
    s_row[it] = P >> 2;
    s_kc[it] = P & 3;
    a_src[it] = A + (long)min(m0 + (s_row[it] & (BM - 1)), p.M - 1) * p.lda + k_begin + s_kc[it] * 8;
    b_src[it] = B + (long)min(n0 + s_row[it], p.N - 1) * p.ldb + k_begin + s_kc[it] * 8;
  }
  // register staging, TWO K tiles deep: tile kt+2 is requested while tile kt is multiplied and tile kt+1 (already in registers) waits
  // to be split into the other LDS buffer -- with one tile in flight the kernel was bound by memory-level parallelism (64 B per thread
  // outstanding against ~2 us of L2 latency)
  float4 ra[2][2][2], rb[2][2][2];
  auto fetch = [&](auto set_c, int kt) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it < NA) {
        ra[SET][it][0] = *(const float4*)(a_src[it] + kt * BK);
        ra[SET][it][1] = *(const float4*)(a_src[it] + kt * BK + 4);
      }
      rb[SET][it][0] = *(const float4*)(b_src[it] + kt * BK);
      rb[SET][it][1] = *(const float4*)(b_src[it] + kt * BK + 4);
    }
  };
  auto store = [&](auto set_c, int buf) {
    constexpr int SET = decltype(set_c)::value;
    char* sa = smem + buf * (BM + BN) * 128;
    char* sb = sa + BM * 128;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      if (it < NA) split_store(sa, s_row[it], s_kc[it], ra[SET][it][0], ra[SET][it][1]);
      split_store(sb, s_row[it], s_kc[it], rb[SET][it][0], rb[SET][it][1]);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x4 acc[FM][4];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15;
  const int c0 = (((lane >> 4) ^ (frow >> 1)) << 4);          // h plane of this lane's k group; the m plane is c0 ^ 64
  const int a_off = (wm * 16 * FM + frow) * 128 + c0;
  const int b_off = (wn * 64 + frow) * 128 + c0;

  auto compute = [&](int buf) {
    const char* sa = smem + buf * (BM + BN) * 128;
    const char* sb = sa + BM * 128;
    tbf16x8 ah[FM], am[FM], bh[4], bm[4];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      ah[i] = *(const tbf16x8*)(sa + a_off + i * 16 * 128);
      am[i] = *(const tbf16x8*)(sa + ((a_off + i * 16 * 128) ^ 64));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = *(const tbf16x8*)(sb + b_off + j * 16 * 128);
      bm[j] = *(const tbf16x8*)(sb + ((b_off + j * 16 * 128) ^ 64));
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {        // small terms first, then the leading one
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[j], ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], am[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
      }
  };
  // tile kt lives in LDS buffer kt & 1 and came through register set kt & 1
  fetch(S0{}, 0);
  if (ntk > 1) fetch(S1{}, 1);
  store(S0{}, 0);
  __syncthreads();
  for (int kt = 0; kt < ntk; kt += 2) {
    // even tile kt: buffer 0; registers: set 1 holds tile kt+1, set 0 is free for tile kt+2
    if (kt + 2 < ntk) fetch(S0{}, kt + 2);
    compute(0);
    if (kt + 1 < ntk) {
      store(S1{}, 1);
      __syncthreads();
      // odd tile kt+1: buffer 1; set 0 holds tile kt+2, set 1 is free for tile kt+3
      if (kt + 3 < ntk) fetch(S1{}, kt + 3);
      compute(1);
      if (kt + 2 < ntk) {
        store(S0{}, 0);
        __syncthreads();
      }
    }
  }

  // ---- epilogue: lane holds C[m][n .. n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4) ----
  const uint32_t seed = p.drop_thr ? *p.seed : 0u;
  const float* bias = p.bias ? p.bias + i1 * p.sBias1 : nullptr;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * 16 * FM + i * 16 + (lane & 15);
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (n >= p.N) continue;                                   // (N % 4 == 0: a group of 4 is inside or outside as a whole)
      float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
      if (p.atomic) {
        float* c = C + (long)m * p.ldc + n;
        if (gridDim.y == 1) {                 // a single K slice owns the element: plain read-modify-write
          float4 o = *(float4*)c;
          o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
          *(float4*)c = o;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(c + e, v[e]);
        }
        continue;
      }
      if (bias) {
        const float4 b = *(const float4*)(bias + n);
        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
      }
      if (p.epi == 1) {
        float d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gelu_both(v[e], v[e], d[e]);
        if (p.pre) *(float4*)(p.pre + (long)m * p.ldpre + n) = make_float4(d[0], d[1], d[2], d[3]);
      } else if (p.epi == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      } else if (p.epi == 3 || p.epi == 5) {
        const float4 a = *(const float4*)(p.aux + (long)m * p.ldaux + n);
        if (p.epi == 3) { v[0] *= a.x; v[1] *= a.y; v[2] *= a.z; v[3] *= a.w; }
        else { v[0] = a.x > 0.f ? v[0] : 0.f; v[1] = a.y > 0.f ? v[1] : 0.f; v[2] = a.z > 0.f ? v[2] : 0.f; v[3] = a.w > 0.f ? v[3] : 0.f; }
      } else if (p.epi == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tanhf(v[e]);
      }
      if (p.drop_thr) {
        const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = vlb_keep(seed, p.tag, idx + e, p.drop_thr) ? v[e] * p.drop_scale : 0.f;
      }
      if (p.res) {
        const float4 r = *(const float4*)(p.res + (long)m * p.ldres + n);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      *(float4*)(C + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// dst[c][r] = src[r][c] for r < R, 0 for R <= r < Rp; optional colsum[c] += sum_r src[r][c] (bias gradients).  32 x 32 tiles.
struct F32Transpose {
  const float* src; float* dst; float* colsum;
  long lds_, ldd;
  int R, C, Rp;
  int nb2; long sS1, sS2, sD1, sD2;
};

__global__ __launch_bounds__(256) void transpose_f32_kernel(const F32Transpose p) {
  __shared__ float tile[32][33];
  const int z = blockIdx.z, i1 = z / p.nb2, i2 = z - i1 * p.nb2;
  const float* src = p.src + i1 * p.sS1 + i2 * p.sS2;
  float* dst = p.dst + i1 * p.sD1 + i2 * p.sD2;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < p.R && c < p.C) ? src[(long)r * p.lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < p.C && r < p.Rp) dst[(long)c * p.ldd + r] = tile[tx][ty + 8 * k];
  }
  if (p.colsum && ty == 0 && c0 + tx < p.C) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) s += tile[r][tx];
    atomicAdd(p.colsum + c0 + tx, s);
  }
}

// ---- LayerNorm (BertLayerNorm, modeling.py:222-235) on fp32 rows; one wave per row, lane owns columns (lane + 64 i) * 4 .. + 3 ----
constexpr int LN_IT = 8;      // H <= 2048

__global__ __launch_bounds__(256) void ln_f32_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float* __restrict__ y, long ldy,
                                                         float* __restrict__ stats, int rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4 v[LN_IT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    v[i] = c < H ? *(const float4*)(x + (long)row * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += a * a + b * b + cc * cc + d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
  if (lane == 0 && stats) { stats[2 * (long)row] = mean; stats[2 * (long)row + 1] = rstd; }
#pragma unroll
  for (int i = 0; i < LN_IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const float4 g = *(const float4*)(gamma + c), b = *(const float4*)(beta + c);
      *(float4*)(y + (long)row * ldy + c) = make_float4((v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y,
                                                        (v[i].z - mean) * rstd * g.z + b.z, (v[i].w - mean) * rstd * g.w + b.w);
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dx_drop = dx * keep(row * H + col) * scale (the gradient entering
// the dense layer in front of the dropout); dgamma += dy * xhat, dbeta += dy (per-wave register partials, one atomic flush per wave).
__global__ __launch_bounds__(256) void ln_f32_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ x, long ldx,
                                                         const float* __restrict__ stats, const float* __restrict__ gamma,
                                                         float* __restrict__ dx, long lddx, float* __restrict__ dx_drop, long lddd,
                                                         uint32_t drop_thr, float drop_scale, const uint32_t* __restrict__ seedp, uint32_t tag,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int H) {
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const uint32_t seed = drop_thr ? *seedp : 0u;
  float4 pg[LN_IT], pb[LN_IT], gm[LN_IT];
#pragma unroll
  for (int i = 0; i < LN_IT; ++i) {
    pg[i] = pb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = (lane + 64 * i) * 4;
    gm[i] = c < H ? *(const float4*)(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row = wave_id; row < rows; row += nwaves) {
    const float mean = stats[2 * (long)row], rstd = stats[2 * (long)row + 1];
    float4 g[LN_IT], xh[LN_IT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        const float4 d = *(const float4*)(dy + (long)row * lddy + c), xv = *(const float4*)(x + (long)row * ldx + c);
        xh[i] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        g[i] = make_float4(d.x * gm[i].x, d.y * gm[i].y, d.z * gm[i].z, d.w * gm[i].w);
        s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
        s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
        pg[i].x += d.x * xh[i].x; pg[i].y += d.y * xh[i].y; pg[i].z += d.z * xh[i].z; pg[i].w += d.w * xh[i].w;
        pb[i].x += d.x; pb[i].y += d.y; pb[i].z += d.z; pb[i].w += d.w;
      } else {
        g[i] = xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    const float m1 = wave_sum(s1) / (float)H, m2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int i = 0; i < LN_IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        float o[4] = {rstd * (g[i].x - m1 - xh[i].x * m2), rstd * (g[i].y - m1 - xh[i].y * m2), rstd * (g[i].z - m1 - xh[i].z * m2),
                      rstd * (g[i].w - m1 - xh[i].w * m2)};
        if (dx) *(float4*)(dx + (long)row * lddx + c) = make_float4(o[0], o[1], o[2], o[3]);
        if (dx_drop) {
          if (drop_thr) {
            const uint32_t idx = (uint32_t)row * (uint32_t)H + (uint32_t)c;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = vlb_keep(seed, tag, idx + e, drop_thr) ? o[e] * drop_scale : 0.f;
          }
          *(float4*)(dx_drop + (long)row * lddd + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
  // the four waves' partial sums -> LDS -> one atomic per column per WORKGROUP (a flush per wave made the atomics, not the row pass,
  // the cost of this kernel: 82 us for 3664 x 1024 rows)
  __shared__ float red[4][2][256 * LN_IT / 2];      // [wave][gamma | beta][column], H <= 1024 per pass
  const int wv = threadIdx.x >> 6;
  for (int base = 0; base < H; base += 1024) {
#pragma unroll
    for (int i = 0; i < LN_IT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c >= base && c < base + 1024 && c < H) {
        *(float4*)&red[wv][0][c - base] = pg[i];
        *(float4*)&red[wv][1][c - base] = pb[i];
      }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 1024 && base + c < H; c += 256) {
      const float sg = (red[0][0][c] + red[1][0][c]) + (red[2][0][c] + red[3][0][c]);
      const float sb = (red[0][1][c] + red[1][1][c]) + (red[2][1][c] + red[3][1][c]);
      if (dgamma) atomicAdd(dgamma + base + c, sg);
      if (dbeta) atomicAdd(dbeta + base + c, sb);
    }
    __syncthreads();
  }
}

// ---- masked softmax over the keys of one (sample, head, query) row; scores already carry 1/sqrt(d) and the additive mask ----
// s: [rows][Sp] fp32 (columns >= S are padding).  p <- softmax(s[:, :S]) (0 in the padding), pd <- p * keep * scale (dropout on
// the probabilities, modeling.py:306-310); pd may alias p when there is no dropout.  One wave per row, Sp <= 256.
__global__ __launch_bounds__(256) void softmax_f32_fwd_kernel(const float* __restrict__ s, const float* __restrict__ mask01, int rows_per_sample,
                                                              float* __restrict__ p, float* __restrict__ pd, int rows, int S, int Sp,
                                                              uint32_t drop_thr, float drop_scale, const uint32_t* __restrict__ seedp, uint32_t tag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint32_t seed = drop_thr ? *seedp : 0u;
  float v[4];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < S) ? s[(long)row * Sp + c] : -3.0e38f;
    // additive attention mask of the reference: (1 - mask) * -10000 (common/visual_linguistic_bert.py:119-127)
    if (mask01 && c < S) v[i] += (1.0f - mask01[(long)(row / rows_per_sample) * S + c]) * -10000.0f;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    v[i] = (c < S) ? __expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  const float inv = 1.0f / wave_sum(sum);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < Sp) {
      const float pr = v[i] * inv;
      p[(long)row * Sp + c] = pr;
      if (pd != p || drop_thr)
        pd[(long)row * Sp + c] = (!drop_thr || vlb_keep(seed, tag, (uint32_t)row * (uint32_t)Sp + (uint32_t)c, drop_thr)) ? pr * drop_scale : 0.f;
    }
  }
}

// ds <- p * (dp - sum_k dp p), dp = dpd * keep * scale, in place over dpd (padding columns -> 0)
__global__ __launch_bounds__(256) void softmax_f32_bwd_kernel(const float* __restrict__ p, float* __restrict__ dpd, int rows, int S, int Sp,
                                                              uint32_t drop_thr, float drop_scale, const uint32_t* __restrict__ seedp,
                                                              uint32_t tag) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint32_t seed = drop_thr ? *seedp : 0u;
  float pr[4], dp[4];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    pr[i] = dp[i] = 0.f;
    if (c < S) {
      pr[i] = p[(long)row * Sp + c];
      float d = dpd[(long)row * Sp + c];
      if (drop_thr) d = vlb_keep(seed, tag, (uint32_t)row * (uint32_t)Sp + (uint32_t)c, drop_thr) ? d * drop_scale : 0.f;
      dp[i] = d;
      dot += d * pr[i];
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    if (c < Sp) dpd[(long)row * Sp + c] = pr[i] * (dp[i] - dot);
  }
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace

// C[M,N] (+)= epilogue(alpha * A[M,K] . B[N,K]^T), all fp32, batched over nb1 x nb2 with two stride levels (elements).
// epilogue order: * alpha, + bias[n] (row i1 * sBias1 of the bias tensor), activation (epi: 0 none | 1 GELU, GELU' -> pre | 2 ReLU |
// 3 x aux | 4 tanh | 5 keep where aux > 0), dropout(drop_p; element index m * N + n), + res.  atomic != 0: C += alpha * A.B^T by
// atomicAdd over splitk K slices (weight gradients; no bias / activation / dropout / residual then).  K % 32 == 0, N % 4 == 0,
// leading dimensions % 4 == 0, 16-byte aligned pointers.
extern "C" int vlb_gemm_nt_f32(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int M, int N, int K, int nb1, int nb2,
                               long sA1, long sA2, long sB1, long sB2, long sC1, long sC2, const float* bias, long sBias1, float alpha,
                               int epi, const float* aux, long ldaux, float* pre, long ldpre, const float* res, long ldres, float drop_p,
                               const uint32_t* seed, uint32_t tag, int atomic, int splitk, hipStream_t stream) {
  if (M <= 0 || N <= 0 || nb1 <= 0 || nb2 <= 0) return VLB_OK;
  VLB_CHECK_ARG(A && B && C, "vlb_gemm_nt_f32: null operand");
  VLB_CHECK_ARG(K > 0 && (K % 32) == 0, "vlb_gemm_nt_f32: K=%d must be a positive multiple of 32 (zero-pad the reduction dimension)", K);
  VLB_CHECK_ARG((N % 4) == 0 && (lda % 4) == 0 && (ldb % 4) == 0 && (ldc % 4) == 0, "vlb_gemm_nt_f32: N and the leading dimensions must be multiples of 4");
  VLB_CHECK_ARG(al16(A) && al16(B) && al16(C) && (!bias || al16(bias)) && (!res || al16(res)) && (!aux || al16(aux)) && (!pre || al16(pre)),
                "vlb_gemm_nt_f32: pointers must be 16-byte aligned");
  VLB_CHECK_ARG((sA1 % 4) == 0 && (sA2 % 4) == 0 && (sB1 % 4) == 0 && (sB2 % 4) == 0 && (sC1 % 4) == 0 && (sC2 % 4) == 0 && (sBias1 % 4) == 0,
                "vlb_gemm_nt_f32: batch strides must be multiples of 4 elements");
  VLB_CHECK_ARG(epi >= 0 && epi <= 5, "vlb_gemm_nt_f32: bad epilogue %d", epi);
  VLB_CHECK_ARG((epi != 3 && epi != 5) || aux, "vlb_gemm_nt_f32: epilogue 3 / 5 needs aux");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_gemm_nt_f32: dropout needs a device seed pointer");
  VLB_CHECK_ARG(!atomic || (!bias && epi == 0 && !(drop_p > 0.f) && !res), "vlb_gemm_nt_f32: the accumulating form takes no epilogue");
  VLB_CHECK_ARG((long)M * (long)N < (1L << 32) || !(drop_p > 0.f), "vlb_gemm_nt_f32: dropout index overflow");
  VLB_CHECK_ARG((long)nb1 * nb2 <= 65535, "vlb_gemm_nt_f32: too many batches");
  F32Gemm p = {};
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K;
  p.nb2 = nb2; p.sA1 = sA1; p.sA2 = sA2; p.sB1 = sB1; p.sB2 = sB2; p.sC1 = sC1; p.sC2 = sC2;
  p.bias = bias; p.sBias1 = sBias1; p.alpha = alpha; p.epi = epi; p.aux = aux; p.ldaux = ldaux; p.pre = pre; p.ldpre = ldpre;
  p.res = res; p.ldres = ldres; p.drop_thr = vlb_drop_thr(drop_p); p.drop_scale = vlb_drop_scale(p.drop_thr); p.seed = seed; p.tag = tag;
  p.atomic = atomic ? 1 : 0;
  const int ktiles = K / 32;
  int splits = (atomic && splitk > 1) ? (splitk > ktiles ? ktiles : splitk) : 1;
  const int per = vlb_cdiv(ktiles, splits);
  splits = vlb_cdiv(ktiles, per);
  p.k_per_split = per * 32;
  // tile height: 128 rows, or 64 when the launch would not give the 512 resident workgroups (2 per CU) a full round -- the N = 1024
  // GEMMs of a 16-sample micro-batch (M = 3664): 232 tiles of 128 rows vs 464 of 64
  const long wgs128 = (long)vlb_cdiv(M, 128) * vlb_cdiv(N, 128) * splits * nb1 * nb2;
  if (wgs128 < 460 && M > 64) {
    hipLaunchKernelGGL(gemm_f32_split_kernel<2>, dim3(vlb_cdiv(M, 64) * vlb_cdiv(N, 128), splits, nb1 * nb2), dim3(256), 0, stream, p);
  } else {
    hipLaunchKernelGGL(gemm_f32_split_kernel<4>, dim3(vlb_cdiv(M, 128) * vlb_cdiv(N, 128), splits, nb1 * nb2), dim3(256), 0, stream, p);
  }
  VLB_CHECK_LAUNCH("vlb_gemm_nt_f32");
  return VLB_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// "TN" form of the split GEMM:  C[Mo, No] (+)= alpha * A[R, Mo]^T . B[R, No]  -- the reduction runs over the R ROWS of two row-major fp32
// operands, i.e. over tensors exactly as the forward / backward passes left them: weight gradients dW = dY^T X (autograd's
// grad_output.t().mm(input)) and the attention products dV = P^T dO, dK = dS^T Q, without the transposed fp32 copies the NT kernel
// needed (1440 transpose launches and 23 ms of a VQA-large step, profiles/r04_vqa_fp32_kernel_stats.txt).
//   * a stage = 32 reduction rows; per operand two bf16 images [32 rows][128 columns] (h plane, m plane; 256-B rows), written from
//     registers after the split (16 B = 8 columns per plane and lane), the 32-B column blocks XOR-swizzled by
//     f(row) = (row & 3) | ((row >> 3) & 1) << 2 exactly as in gemm_tn_bf16_kernel (gemm.hip) -- here on the LDS write address;
//   * MFMA fragments (8 consecutive reduction rows of one column per lane) come out through the LDS transpose read
//     ds_read_b64_tr_b16, two per fragment and plane (compiler-visible builtin: no LDS-DMA in this kernel, so hipcc's own waits are exact);
//   * 128 x 128 output tile, 2 x 2 waves of 64 x 64, three MFMAs per fragment pair (bm.ah + bh.am + bh.ah), register staging two stages
//     deep, K slices over blockIdx.y with fp32 atomics (as the NT kernel);
//   * optional column sums of A (bias gradients), taken from the fp32 registers on their way to LDS by the tile_n == 0 workgroups.
// Rows >= R and columns >= Mo / No are loaded clamped and replaced by zeros.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short tn_s16x4;
typedef __attribute__((address_space(3))) tn_s16x4 tn_lds_s16x4;

__device__ __forceinline__ tbf16x8 tn_f32_frag(const char* img, int off) {      // rows +0..3 and +4..7 of the lane's column
  const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds_s16x4*)(img + off));
  const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds_s16x4*)(img + off + 4 * 256));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(tbf16x8, v);
}

// 8 consecutive columns of one reduction row -> the two planes (16 B each) at the row's swizzled chunk
__device__ __forceinline__ void split_store_tn(char* img_h, int row, int chunk, const float4& x0, const float4& x1) {
  const uint32_t h0 = pack2_true_bf16(x0.x, x0.y), h1 = pack2_true_bf16(x0.z, x0.w), h2 = pack2_true_bf16(x1.x, x1.y), h3 = pack2_true_bf16(x1.z, x1.w);
  const uint32_t m0 = pack2_true_bf16(x0.x - true_bf_lo(h0), x0.y - true_bf_hi(h0)), m1 = pack2_true_bf16(x0.z - true_bf_lo(h1), x0.w - true_bf_hi(h1));
  const uint32_t m2 = pack2_true_bf16(x1.x - true_bf_lo(h2), x1.y - true_bf_hi(h2)), m3 = pack2_true_bf16(x1.z - true_bf_lo(h3), x1.w - true_bf_hi(h3));
  const int f = (row & 3) | (((row >> 3) & 1) << 2);
  const int pc = ((((chunk >> 1) ^ f) << 1) | (chunk & 1));
  *(uint4*)(img_h + row * 256 + pc * 16) = make_uint4(h0, h1, h2, h3);
  *(uint4*)(img_h + 32 * 256 + row * 256 + pc * 16) = make_uint4(m0, m1, m2, m3);
}

template <bool COLSUM>
__global__ __launch_bounds__(256, 2) void gemm_f32_split_tn_kernel(const F32Gemm p, float* __restrict__ colsum) {
  constexpr int BR = 32, IMG = BR * 256, STAGE = 4 * IMG;      // A_h | A_m | B_h | B_m : 32 KiB per stage
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;                     // 2 x 2 waves of 64 x 64
  const int ntm = (p.M + 127) / 128;
  const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;
  const int m0 = tm * 128, n0 = tn * 128;
  const int z = blockIdx.z, i1 = z / p.nb2, i2 = z - i1 * p.nb2;
  const float* A = p.A + i1 * p.sA1 + i2 * p.sA2;
  const float* B = p.B + i1 * p.sB1 + i2 * p.sB2;
  float* C = p.C + i1 * p.sC1 + i2 * p.sC2;
  const int r_begin = blockIdx.y * p.k_per_split;
  const int r_end = min(p.K, r_begin + p.k_per_split);         // p.K = number of reduction rows R
  const int ntk = (r_end - r_begin + BR - 1) / BR;
  if (ntk <= 0) return;

  // staging: item P = it * 256 + tid (it = 0, 1): row = P >> 4 (0..31), 8-column chunk c = P & 15
  int s_row[2], s_c[2];
  bool a_ok[2], b_ok[2];
  const float* a_src[2];
  const float* b_src[2];
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int P = it * 256 + tid;
    s_row[it] = P >> 4;
    s_c[it] = P & 15;
    // a chunk that STARTS inside the operand is loaded whole: its columns beyond Mo / No are real memory (lda >= round8(Mo)) and only
    // reach output rows / columns the epilogue masks; chunks that start outside are clamped in bounds and replaced by zeros
    a_ok[it] = m0 + s_c[it] * 8 < p.M;
    b_ok[it] = n0 + s_c[it] * 8 < p.N;
    a_src[it] = A + min(m0 + s_c[it] * 8, ((p.M + 7) & ~7) - 8);
    b_src[it] = B + min(n0 + s_c[it] * 8, ((p.N + 7) & ~7) - 8);
  }
  float4 ra[2][2][2], rb[2][2][2];
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](auto set_c, int kt) {
    constexpr int SET = decltype(set_c)::value;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = r_begin + kt * BR + s_row[it];
      const long rr = min(r, r_end - 1);                        // unconditional, clamped loads (a load under a condition serialises)
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 a0 = *(const float4*)(a_src[it] + rr * p.lda), a1 = *(const float4*)(a_src[it] + rr * p.lda + 4);
      const float4 b0 = *(const float4*)(b_src[it] + rr * p.ldb), b1 = *(const float4*)(b_src[it] + rr * p.ldb + 4);
      const bool in = r < r_end;
      ra[SET][it][0] = (in && a_ok[it]) ? a0 : z4; ra[SET][it][1] = (in && a_ok[it]) ? a1 : z4;
      rb[SET][it][0] = (in && b_ok[it]) ? b0 : z4; rb[SET][it][1] = (in && b_ok[it]) ? b1 : z4;
    }
  };
  auto store = [&](auto set_c, int buf) {
    constexpr int SET = decltype(set_c)::value;
    char* sa = smem + buf * STAGE;
    char* sb = sa + 2 * IMG;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      split_store_tn(sa, s_row[it], s_c[it], ra[SET][it][0], ra[SET][it][1]);
      split_store_tn(sb, s_row[it], s_c[it], rb[SET][it][0], rb[SET][it][1]);
      if (COLSUM) {      // (both items of a thread cover the same 8 columns: c = tid & 15)
        cs[0] += ra[SET][it][0].x; cs[1] += ra[SET][it][0].y; cs[2] += ra[SET][it][0].z; cs[3] += ra[SET][it][0].w;
        cs[4] += ra[SET][it][1].x; cs[5] += ra[SET][it][1].y; cs[6] += ra[SET][it][1].z; cs[7] += ra[SET][it][1].w;
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment addressing (gemm.hip, "TN GEMM for weight gradients"): L = lane & 15 supplies (row L >> 2, 4 columns (L & 3) * 4) of a 4 x 16
  // block and receives the block's column L; lane group g = lane >> 4 takes reduction rows 8 g .. 8 g + 7
  const int L = lane & 15, g = lane >> 4;
  const int fl = (L >> 2) | ((g & 1) << 2);
  const int lane_off = (8 * g + (L >> 2)) * 256 + (L & 3) * 8;
  int a_fo[4], b_fo[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_fo[i] = lane_off + (((wm * 4 + i) ^ fl) << 5);
    b_fo[i] = lane_off + (((wn * 4 + i) ^ fl) << 5);
  }
  auto compute = [&](int buf) {
    const char* sa = smem + buf * STAGE;
    const char* sb = sa + 2 * IMG;
    tbf16x8 ah[4], am[4], bh[4], bm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = tn_f32_frag(sa, a_fo[i]);
      am[i] = tn_f32_frag(sa + IMG, a_fo[i]);
      bh[i] = tn_f32_frag(sb, b_fo[i]);
      bm[i] = tn_f32_frag(sb + IMG, b_fo[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {        // small terms first, then the leading one
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[j], ah[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], am[i], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
      }
  };
  // stage kt lives in LDS buffer kt & 1 and came through register set kt & 1 (see gemm_f32_split_kernel)
  fetch(S0{}, 0);
  if (ntk > 1) fetch(S1{}, 1);
  store(S0{}, 0);
  __syncthreads();
  for (int kt = 0; kt < ntk; kt += 2) {
    if (kt + 2 < ntk) fetch(S0{}, kt + 2);
    compute(0);
    if (kt + 1 < ntk) {
      store(S1{}, 1);
      __syncthreads();
      if (kt + 3 < ntk) fetch(S1{}, kt + 3);
      compute(1);
      if (kt + 2 < ntk) {
        store(S0{}, 0);
        __syncthreads();
      }
    }
  }
  if (COLSUM) {      // lanes l, l ^ 16, l ^ 32, l ^ 48 of a wave hold the same 8 columns (c = tid & 15)
    if (tn == 0 && z == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = cs[e];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        const int m = m0 + (lane & 15) * 8 + e;
        if (lane < 16 && m < p.M) atomicAdd(colsum + m, v);
      }
    }
  }
  // ---- epilogue: lane holds C[m][n .. n+3], m = .. + (lane & 15), n = .. + 4 * (lane >> 4) ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + L;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * g;
      if (n >= p.N) continue;                                   // (No % 4 == 0)
      const float v[4] = {acc[i][j][0] * p.alpha, acc[i][j][1] * p.alpha, acc[i][j][2] * p.alpha, acc[i][j][3] * p.alpha};
      float* c = C + (long)m * p.ldc + n;
      if (!p.atomic) {
        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
      } else if (gridDim.y == 1) {            // a single K slice owns the element: plain read-modify-write
        float4 o = *(float4*)c;
        o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
        *(float4*)c = o;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(c + e, v[e]);
      }
    }
  }
}

// C[Mo, No] (+)= alpha * A[R, Mo]^T . B[R, No] in fp32 (split products), batched over nb1 x nb2; atomic: accumulate (K slices allowed);
// colsum (nullable, unbatched): += column sums of A.
extern "C" int vlb_gemm_tn_f32(const float* A, long lda, const float* B, long ldb, float* C, long ldc, int R, int Mo, int No, int nb1, int nb2,
                               long sA1, long sA2, long sB1, long sB2, long sC1, long sC2, float alpha, int atomic, int splitk, float* colsum,
                               hipStream_t stream) {
  if (R <= 0 || Mo <= 0 || No <= 0 || nb1 <= 0 || nb2 <= 0) return VLB_OK;
  VLB_CHECK_ARG(A && B && C, "vlb_gemm_tn_f32: null operand");
  VLB_CHECK_ARG((No % 4) == 0 && (lda % 4) == 0 && (ldb % 4) == 0 && (ldc % 4) == 0 && lda >= ((Mo + 7) & ~7) && ldb >= ((No + 7) & ~7),
                "vlb_gemm_tn_f32: No must be a multiple of 4, the leading dimensions multiples of 4 and at least Mo / No rounded up to 8 "
                "(operand rows are read in whole 8-column chunks)");
  VLB_CHECK_ARG(al16(A) && al16(B) && al16(C), "vlb_gemm_tn_f32: pointers must be 16-byte aligned");
  VLB_CHECK_ARG((sA1 % 4) == 0 && (sA2 % 4) == 0 && (sB1 % 4) == 0 && (sB2 % 4) == 0 && (sC1 % 4) == 0 && (sC2 % 4) == 0,
                "vlb_gemm_tn_f32: batch strides must be multiples of 4 elements");
  VLB_CHECK_ARG(!colsum || nb1 * nb2 == 1, "vlb_gemm_tn_f32: column sums are for the unbatched form");
  VLB_CHECK_ARG((long)nb1 * nb2 <= 65535, "vlb_gemm_tn_f32: too many batches");
  F32Gemm p = {};
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = Mo; p.N = No; p.K = R;
  p.nb2 = nb2; p.sA1 = sA1; p.sA2 = sA2; p.sB1 = sB1; p.sB2 = sB2; p.sC1 = sC1; p.sC2 = sC2;
  p.alpha = alpha; p.atomic = atomic ? 1 : 0;
  const int stages = vlb_cdiv(R, 32);
  int splits = (atomic && splitk > 1) ? (splitk > stages ? stages : splitk) : 1;
  const int per = vlb_cdiv(stages, splits);
  splits = vlb_cdiv(stages, per);
  p.k_per_split = per * 32;
  const dim3 grid(vlb_cdiv(Mo, 128) * vlb_cdiv(No, 128), splits, nb1 * nb2);
  if (colsum) hipLaunchKernelGGL(gemm_f32_split_tn_kernel<true>, grid, dim3(256), 0, stream, p, colsum);
  else hipLaunchKernelGGL(gemm_f32_split_tn_kernel<false>, grid, dim3(256), 0, stream, p, (float*)nullptr);
  VLB_CHECK_LAUNCH("vlb_gemm_tn_f32");
  return VLB_OK;
}

// dst[c][r] = src[r][c] (r < R; zero for R <= r < Rp), batched with two stride levels; colsum (nullable, unbatched use) += column sums.
extern "C" int vlb_transpose_f32(const float* src, long lds, float* dst, long ldd, int R, int C, int Rp, int nb1, int nb2, long sS1, long sS2,
                                 long sD1, long sD2, float* colsum, hipStream_t stream) {
  if (R <= 0 || C <= 0 || nb1 <= 0 || nb2 <= 0) return VLB_OK;
  VLB_CHECK_ARG(src && dst && Rp >= R && ldd >= Rp, "vlb_transpose_f32: bad argument");
  VLB_CHECK_ARG(!colsum || nb1 * nb2 == 1, "vlb_transpose_f32: column sums are for the unbatched form");
  VLB_CHECK_ARG((long)nb1 * nb2 <= 65535, "vlb_transpose_f32: too many batches");
  F32Transpose p = {src, dst, colsum, lds, ldd, R, C, Rp, nb2, sS1, sS2, sD1, sD2};
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(vlb_cdiv(C, 32), vlb_cdiv(Rp, 32), nb1 * nb2), dim3(256), 0, stream, p);
  VLB_CHECK_LAUNCH("vlb_transpose_f32");
  return VLB_OK;
}

extern "C" int vlb_layernorm_f32_fwd(const float* x, long ldx, const float* gamma, const float* beta, float* y, long ldy, float* stats,
                                     int rows, int H, float eps, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(x && gamma && beta && y, "vlb_layernorm_f32_fwd: null argument");
  VLB_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= 256 * LN_IT && (ldx % 4) == 0 && (ldy % 4) == 0, "vlb_layernorm_f32_fwd: unsupported H=%d / strides", H);
  hipLaunchKernelGGL(ln_f32_fwd_kernel, dim3(vlb_cdiv(rows, 4)), dim3(256), 0, stream, x, ldx, gamma, beta, y, ldy, stats, rows, H, eps);
  VLB_CHECK_LAUNCH("vlb_layernorm_f32_fwd");
  return VLB_OK;
}

extern "C" int vlb_layernorm_f32_bwd(const float* dy, long lddy, const float* x, long ldx, const float* stats, const float* gamma, float* dx,
                                     long lddx, float* dx_drop, long lddd, float drop_p, const uint32_t* seed, uint32_t tag, float* dgamma,
                                     float* dbeta, int rows, int H, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(dy && x && stats && gamma, "vlb_layernorm_f32_bwd: null argument");
  VLB_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= 256 * LN_IT && (lddy % 4) == 0 && (ldx % 4) == 0 && (lddx % 4) == 0 && (lddd % 4) == 0,
                "vlb_layernorm_f32_bwd: unsupported H=%d / strides", H);
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_layernorm_f32_bwd: dropout needs a device seed pointer");
  VLB_CHECK_ARG((long)rows * H < (1L << 32), "vlb_layernorm_f32_bwd: dropout index overflow");
  int blocks = vlb_cdiv(rows, 8);           // >= 2 rows per wave, at most two workgroups per CU
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  const uint32_t thr = vlb_drop_thr(drop_p);
  hipLaunchKernelGGL(ln_f32_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dy, lddy, x, ldx, stats, gamma, dx, lddx, dx_drop, lddd, thr,
                     vlb_drop_scale(thr), seed, tag, dgamma, dbeta, rows, H);
  VLB_CHECK_LAUNCH("vlb_layernorm_f32_bwd");
  return VLB_OK;
}

extern "C" int vlb_softmax_f32_fwd(const float* s, const float* mask01, int rows_per_sample, float* p, float* pd, int rows, int S, int Sp,
                                   float drop_p, const uint32_t* seed, uint32_t tag, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(s && p && pd && S > 0 && Sp >= S && Sp <= 256, "vlb_softmax_f32_fwd: bad argument (rows of at most 256 keys)");
  VLB_CHECK_ARG(!mask01 || rows_per_sample > 0, "vlb_softmax_f32_fwd: the key mask needs rows_per_sample (heads x queries)");
  VLB_CHECK_ARG(!(drop_p > 0.f) || (seed && pd != p), "vlb_softmax_f32_fwd: dropout needs a seed and its own output");
  VLB_CHECK_ARG((long)rows * Sp < (1L << 32), "vlb_softmax_f32_fwd: dropout index overflow");
  const uint32_t thr = vlb_drop_thr(drop_p);
  hipLaunchKernelGGL(softmax_f32_fwd_kernel, dim3(vlb_cdiv(rows, 4)), dim3(256), 0, stream, s, mask01, rows_per_sample, p, pd, rows, S, Sp, thr,
                     vlb_drop_scale(thr), seed, tag);
  VLB_CHECK_LAUNCH("vlb_softmax_f32_fwd");
  return VLB_OK;
}

extern "C" int vlb_softmax_f32_bwd(const float* p, float* dpd, int rows, int S, int Sp, float drop_p, const uint32_t* seed, uint32_t tag,
                                   hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(p && dpd && S > 0 && Sp >= S && Sp <= 256, "vlb_softmax_f32_bwd: bad argument");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_softmax_f32_bwd: dropout needs a seed");
  const uint32_t thr = vlb_drop_thr(drop_p);
  hipLaunchKernelGGL(softmax_f32_bwd_kernel, dim3(vlb_cdiv(rows, 4)), dim3(256), 0, stream, p, dpd, rows, S, Sp, thr, vlb_drop_scale(thr), seed, tag);
  VLB_CHECK_LAUNCH("vlb_softmax_f32_bwd");
  return VLB_OK;
}
