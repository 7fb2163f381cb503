// Fused multi-head self-attention forward / backward for short sequences (S <= 128, d = 64)
// on gfx950 -- BertSelfAttention (external/pytorch_pretrained_bert/modeling.py:290-319):
//     scores = Q K^T / sqrt(d) + (1-mask)*-10000 ; P = softmax(scores) ; P = dropout(P) ; ctx = P V
// The reference materialises [B,h,S,S] scores/probs in HBM (15.7 MB per layer at B=32) and runs
// softmax/dropout as separate elementwise passes; here one 4-wave workgroup owns one (batch, head):
// K/V (and, in backward, Q/dO and their transposes) live in LDS, the SxS tile lives only in
// registers, and the backward recomputes P from the saved row log-sum-exp.
//
// MFMA (v_mfma_f32_16x16x32_bf16) operand plan -- no register transposes anywhere:
//  * the score tile is computed TRANSPOSED, S^T = K Q^T, with the 16 K-rows of a tile taken in the
//    order  key = 32u + 8(i>>2) + 4*half + (i&3).  In the C layout (lane: col = lane&15, rows
//    4*(lane>>4)+r) a lane then holds, for ONE query, keys 32u + 8g + [0..8) -- exactly the
//    (col, k = 8g+j) B-operand layout of the next MFMA whose reduction runs over keys
//    (ctx^T = V^T P^T forward, dQ^T = K^T dS^T backward).
//  * reductions over QUERIES (dV^T = dO^T P, dK^T = Q^T dS) use the other orientation, S = Q K^T
//    with permuted Q rows, recomputed by the wave that owns those keys (MFMA is cheap, LDS
//    round-trips of a 128x128 tile are not).
//  * "transposed" operands (V^T, K^T, Q^T, dO^T: rows = head dim, 8 contiguous keys/queries per
//    lane) come from LDS images written once per workgroup with an XOR-16 swizzle; row-major
//    images use the GEMM's (row>>1)&7 swizzle.  All fragment reads are ds_read_b128.
//  * every output fragment has 4 consecutive head-dim elements per lane -> 8-B bf16 stores.
// Dropout uses the counter RNG of vlb_common.h keyed on (b, h, q, key): forward and both backward
// orientations regenerate identical masks, nothing is stored.
#include "vlb_common.h"

#define ATT_SP 128      // padded sequence length handled by one workgroup
#define ATT_D 64
#define TILE_BYTES (ATT_SP * ATT_D * 2)  // 16 KiB, both for [128][64] and [64][128] images

// ---- LDS address helpers -------------------------------------------------------------------
// row-major image [128 rows][64]: 128-B rows, 8 x 16-B slots, slot' = slot ^ ((row>>1)&7)
__device__ __forceinline__ int rm_addr(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// transposed image [64 d][128 seq]: 256-B rows, 16 slots, slot' = slot ^ (d&15)
__device__ __forceinline__ int tr_addr(int d, int chunk) { return d * 256 + ((chunk ^ (d & 15)) << 4); }
__device__ __forceinline__ int tr_elem_addr(int d, int s) { return d * 256 + ((((s >> 3) ^ (d & 15))) << 4) + (s & 7) * 2; }
// row of the permuted "A operand" order: tile (blk32, half), fragment row i = lane&15
__device__ __forceinline__ int perm_row(int blk, int half, int i) { return 32 * blk + 8 * (i >> 2) + 4 * half + (i & 3); }

__device__ __forceinline__ bf16x8 lds_frag(const char* base, int addr) { return *(const bf16x8*)(base + addr); }

// Stage one [S][64] head slice (rows b*S.., row stride ld elements) into LDS.
//   rm  : row-major image (or null)     tr : transposed image (or null)
// NT threads, 1024/NT chunks of 16 B each; rows >= S are zero-filled.
template <int NT>
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, long ld, int S, char* rm, char* tr, int tid) {
#pragma unroll
  for (int it = 0; it < 1024 / NT; ++it) {
    const int P = it * NT + tid, row = P >> 3, ch = P & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < S) v = *(const uint4*)(g + (long)row * ld + ch * 8);
    if (rm) *(uint4*)(rm + rm_addr(row, ch)) = v;
    if (tr) {
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        *(bf16_t*)(tr + tr_elem_addr(ch * 8 + 2 * k, row)) = (bf16_t)(w[k] & 0xffffu);
        *(bf16_t*)(tr + tr_elem_addr(ch * 8 + 2 * k + 1, row)) = (bf16_t)(w[k] >> 16);
      }
    }
  }
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  uint32_t w[4] = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, *(uint4*)w);
}

struct AttnParams {
  const bf16_t* qkv;   // [B*S, 3H]
  const float* mask;   // [B,S] 1 = attend, 0 = masked (adds -10000 like the reference)
  bf16_t* ctx;         // [B*S, H]          (fwd out / bwd in)
  float* lse;          // [B, nh, S]        (fwd out / bwd in)
  const bf16_t* dctx;  // [B*S, H]          (bwd in)
  bf16_t* dqkv;        // [B*S, 3H]         (bwd out)
  int B, S, H, nh;
  float scale;
  uint32_t drop_thr; float drop_scale; const uint32_t* seed; uint32_t tag;
};

// =============================================================================================
// forward
// =============================================================================================
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;                                  // row-major [128][64]
  char* sVt = smem + TILE_BYTES;                    // transposed [64][128]
  float* sMB = (float*)(smem + 2 * TILE_BYTES);     // [128] additive mask (-inf beyond S)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const bf16_t* qbase = p.qkv + (long)b * S * ld + h * ATT_D;
  stage_tile<256>(qbase + p.H, ld, S, sK, nullptr, tid);
  stage_tile<256>(qbase + 2 * p.H, ld, S, nullptr, sVt, tid);
  if (tid < ATT_SP) sMB[tid] = (tid < S) ? (1.0f - p.mask[b * S + tid]) * -10000.0f : -INFINITY;

  const int q0 = wave * 32;
  if (q0 >= S) {  // whole wave beyond the sequence: nothing to compute (still must hit the barrier)
    __syncthreads();
    return;
  }
  // Q fragments straight from global in B-operand layout: lane (c, g) <- Q[q0+16qb+c][32ds+8g ..+8)
  bf16x8 qf[2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int q = min(q0 + qb * 16 + c, S - 1);
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) qf[qb][ds] = *(const bf16x8*)(qbase + (long)q * ld + ds * 32 + g * 8);
  }
  __syncthreads();

  const int U = (S + 31) >> 5;  // key blocks of 32 actually present
  f32x4 sc[4][2][2];            // [u][half][qb] : S^T tiles
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      sc[u][hf][0] = sc[u][hf][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (u < U) {
#pragma unroll
        for (int ds = 0; ds < 2; ++ds) {
          const bf16x8 kf = lds_frag(sK, rm_addr(perm_row(u, hf, c), ds * 4 + g));
          sc[u][hf][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ds], sc[u][hf][0], 0, 0, 0);
          sc[u][hf][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ds], sc[u][hf][1], 0, 0, 0);
        }
      }
    }

  // softmax over keys for the lane's query (one per qb): lane-local 32 values, then across g
  const uint32_t seed = (p.drop_thr && p.seed) ? *p.seed : 0u;
  bf16x8 pf[2][4];  // [qb][u]  P (after dropout) as MFMA operand, k = 8g + (4*half + r)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 32 * u + 8 * g + 4 * hf + r;
          const float v = (u < U) ? sc[u][hf][qb][r] * p.scale + sMB[key] : -INFINITY;
          sc[u][hf][qb][r] = v;
          mx = fmaxf(mx, v);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(sc[u][hf][qb][r] - mx);
          sc[u][hf][qb][r] = e;
          sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    const int q = q0 + qb * 16 + c;
    if (g == 0 && q < S) p.lse[((long)b * p.nh + h) * S + q] = mx + __logf(sum);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[4 * hf + r] = sc[u][hf][qb][r] * inv;
      if (p.drop_thr) {
        // idx = ((b*nh+h)*S + q)*S + key ; keys 32u+8g .. +8 are consecutive
        const uint32_t base = (((uint32_t)(b * p.nh + h) * (uint32_t)S + (uint32_t)min(q, S - 1)) * (uint32_t)S) + 32u * u + 8u * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = vlb_keep(seed, p.tag, base + j, p.drop_thr) ? v[j] * p.drop_scale : 0.f;
      }
      pf[qb][u] = pack8(v);
    }
  }

  // ctx^T[d][q] = sum_key V^T[d][key] P^T[key][q]
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < U) {
        const bf16x8 vf = lds_frag(sVt, tr_addr(dt * 16 + c, 4 * u + g));
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0][u], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1][u], o1, 0, 0, 0);
      }
    }
    // C layout: col = query c, rows = d = dt*16 + 4g + r
    const int qa = q0 + c, qb_ = q0 + 16 + c;
    if (qa < S) {
      uint2 w = {pack2bf(o0[0], o0[1]), pack2bf(o0[2], o0[3])};
      *(uint2*)(p.ctx + ((long)b * S + qa) * p.H + h * ATT_D + dt * 16 + 4 * g) = w;
    }
    if (qb_ < S) {
      uint2 w = {pack2bf(o1[0], o1[1]), pack2bf(o1[2], o1[3])};
      *(uint2*)(p.ctx + ((long)b * S + qb_) * p.H + h * ATT_D + dt * 16 + 4 * g) = w;
    }
  }
}

// =============================================================================================
// backward
// =============================================================================================
// 512 threads: waves 0-3 run orientation N (dK, dV for keys 32w..), waves 4-7 run orientation T (dQ for
// queries 32w..) CONCURRENTLY -- the two halves are independent, share the LDS images, and give the CU two
// waves per SIMD to hide LDS latency.
__global__ __launch_bounds__(512) void attn_bwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;                    // row-major
  char* sK = smem + 1 * TILE_BYTES;
  char* sV = smem + 2 * TILE_BYTES;
  char* sdO = smem + 3 * TILE_BYTES;
  char* sQt = smem + 4 * TILE_BYTES;  // transposed
  char* sKt = smem + 5 * TILE_BYTES;
  char* sdOt = smem + 6 * TILE_BYTES;
  float* sMB = (float*)(smem + 7 * TILE_BYTES);  // [128]
  float* sLSE = sMB + ATT_SP;                    // [128]
  float* sD = sLSE + ATT_SP;                     // [128]  rowsum(dO * O)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const bf16_t* qbase = p.qkv + (long)b * S * ld + h * ATT_D;
  const bf16_t* dobase = p.dctx + (long)b * S * p.H + h * ATT_D;
  const bf16_t* obase = p.ctx + (long)b * S * p.H + h * ATT_D;
  stage_tile<512>(qbase, ld, S, sQ, sQt, tid);
  stage_tile<512>(qbase + p.H, ld, S, sK, sKt, tid);
  stage_tile<512>(qbase + 2 * p.H, ld, S, sV, nullptr, tid);
  stage_tile<512>(dobase, p.H, S, sdO, sdOt, tid);
  // D[q] = sum_d dO[q][d] * O[q][d]   (8 lanes per row, 8 elements each)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int P = it * 512 + tid, row = P >> 3, ch = P & 7;
    float d = 0.f;
    if (row < S) {
      const uint4 a = *(const uint4*)(dobase + (long)row * p.H + ch * 8);
      const uint4 o = *(const uint4*)(obase + (long)row * p.H + ch * 8);
      d = bflo(a.x) * bflo(o.x) + bfhi(a.x) * bfhi(o.x) + bflo(a.y) * bflo(o.y) + bfhi(a.y) * bfhi(o.y) +
          bflo(a.z) * bflo(o.z) + bfhi(a.z) * bfhi(o.z) + bflo(a.w) * bflo(o.w) + bfhi(a.w) * bfhi(o.w);
    }
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (ch == 0) sD[row] = d;
  }
  if (tid < ATT_SP) {
    sMB[tid] = (tid < S) ? (1.0f - p.mask[b * S + tid]) * -10000.0f : -INFINITY;
    sLSE[tid] = (tid < S) ? p.lse[((long)b * p.nh + h) * S + tid] : 0.f;
  }
  __syncthreads();

  const uint32_t seed = (p.drop_thr && p.seed) ? *p.seed : 0u;
  const uint32_t bh = (uint32_t)(b * p.nh + h);
  const int U = (S + 31) >> 5;
  const int w32 = (wave & 3) * 32;
  const bool role_n = wave < 4;

  // ------------------------------------------------------------------------------------------
  // Orientation N (wave owns keys w32 .. w32+31): dV, dK (reductions over queries)
  // ------------------------------------------------------------------------------------------
  if (role_n && w32 < S) {
    bf16x8 kf[2][2], vf[2][2];  // B operands: lane (c,g) <- K/V[key = w32+16kb+c][32ds+8g..]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        kf[kb][ds] = lds_frag(sK, rm_addr(w32 + 16 * kb + c, ds * 4 + g));
        vf[kb][ds] = lds_frag(sV, rm_addr(w32 + 16 * kb + c, ds * 4 + g));
      }
    f32x4 dv[2][4], dk[2][4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dv[kb][dt] = dk[kb][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (v < U) {
        bf16x8 pd[2], dsf[2];  // per kb: dropped P and dS as operands (col = key c, k = query 8g+j)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
          float pv[8], dsv[8];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
              const int a = rm_addr(perm_row(v, hf, c), ds * 4 + g);
              s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(sQ, a), kf[kb][ds], s, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(sdO, a), vf[kb][ds], dp, 0, 0, 0);
            }
            const int key = w32 + 16 * kb + c;
            const float mb = sMB[key];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int q = 32 * v + 8 * g + 4 * hf + r;
              const float pr = __expf(s[r] * p.scale + mb - sLSE[q]);
              float keepf = 1.f;
              if (p.drop_thr)
                keepf = vlb_keep(seed, p.tag, (bh * (uint32_t)S + (uint32_t)min(q, S - 1)) * (uint32_t)S + (uint32_t)key, p.drop_thr)
                            ? p.drop_scale : 0.f;
              pv[4 * hf + r] = pr * keepf;
              dsv[4 * hf + r] = pr * (dp[r] * keepf - sD[q]);
            }
          }
          pd[kb] = pack8(pv);
          dsf[kb] = pack8(dsv);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 dot = lds_frag(sdOt, tr_addr(dt * 16 + c, 4 * v + g));
          const bf16x8 qt = lds_frag(sQt, tr_addr(dt * 16 + c, 4 * v + g));
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            dv[kb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pd[kb], dv[kb][dt], 0, 0, 0);
            dk[kb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, dsf[kb], dk[kb][dt], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int key = w32 + 16 * kb + c;
      if (key < S) {
        bf16_t* orow = p.dqkv + ((long)b * S + key) * ld + h * ATT_D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 wk = {pack2bf(dk[kb][dt][0] * p.scale, dk[kb][dt][1] * p.scale),
                      pack2bf(dk[kb][dt][2] * p.scale, dk[kb][dt][3] * p.scale)};
          uint2 wv = {pack2bf(dv[kb][dt][0], dv[kb][dt][1]), pack2bf(dv[kb][dt][2], dv[kb][dt][3])};
          *(uint2*)(orow + p.H + dt * 16 + 4 * g) = wk;
          *(uint2*)(orow + 2 * p.H + dt * 16 + 4 * g) = wv;
        }
      }
    }
  }

  // ------------------------------------------------------------------------------------------
  // Orientation T (wave owns queries w32 .. w32+31): dQ (reduction over keys)
  // ------------------------------------------------------------------------------------------
  if (!role_n && w32 < S) {
    bf16x8 qf[2][2], dof[2][2];  // B operands: lane (c,g) <- Q/dO[q = w32+16qb+c][32ds+8g..]
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int ds = 0; ds < 2; ++ds) {
        qf[qb][ds] = lds_frag(sQ, rm_addr(w32 + 16 * qb + c, ds * 4 + g));
        dof[qb][ds] = lds_frag(sdO, rm_addr(w32 + 16 * qb + c, ds * 4 + g));
      }
    f32x4 dq[2][4];
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[qb][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (u < U) {
        bf16x8 dsf[2];
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          float dsv[8];
          const int q = w32 + 16 * qb + c;
          const float lse = sLSE[q], dd = sD[q];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ds = 0; ds < 2; ++ds) {
              const int a = rm_addr(perm_row(u, hf, c), ds * 4 + g);
              s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(sK, a), qf[qb][ds], s, 0, 0, 0);
              dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(sV, a), dof[qb][ds], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = 32 * u + 8 * g + 4 * hf + r;
              const float pr = __expf(s[r] * p.scale + sMB[key] - lse);
              float keepf = 1.f;
              if (p.drop_thr)
                keepf = vlb_keep(seed, p.tag, (bh * (uint32_t)S + (uint32_t)min(q, S - 1)) * (uint32_t)S + (uint32_t)key, p.drop_thr)
                            ? p.drop_scale : 0.f;
              dsv[4 * hf + r] = pr * (dp[r] * keepf - dd);
            }
          }
          dsf[qb] = pack8(dsv);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 kt = lds_frag(sKt, tr_addr(dt * 16 + c, 4 * u + g));
          dq[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf[0], dq[0][dt], 0, 0, 0);
          dq[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, dsf[1], dq[1][dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int q = w32 + 16 * qb + c;
      if (q < S) {
        bf16_t* orow = p.dqkv + ((long)b * S + q) * ld + h * ATT_D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          uint2 wq = {pack2bf(dq[qb][dt][0] * p.scale, dq[qb][dt][1] * p.scale),
                      pack2bf(dq[qb][dt][2] * p.scale, dq[qb][dt][3] * p.scale)};
          *(uint2*)(orow + dt * 16 + 4 * g) = wq;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------- C ABI
static int check_attn(const char* name, int B, int S, int H, int nh) {
  VLB_CHECK_ARG(B > 0 && S > 0 && S <= ATT_SP, "%s: S=%d unsupported (1..%d)", name, S, ATT_SP);
  VLB_CHECK_ARG(nh > 0 && H == nh * ATT_D, "%s: head dim must be 64 (H=%d, heads=%d)", name, H, nh);
  VLB_CHECK_ARG((long)B * nh * S * S < (1L << 32), "%s: dropout index overflow", name);
  return VLB_OK;
}

extern "C" int vlb_attention_fwd(const void* qkv, const float* mask, void* ctx, float* lse, int B, int S, int H, int nh,
                                 float drop_p, const uint32_t* seed, uint32_t tag, hipStream_t stream) {
  int rc = check_attn("vlb_attention_fwd", B, S, H, nh);
  if (rc) return rc;
  VLB_CHECK_ARG(qkv && mask && ctx && lse, "vlb_attention_fwd: null argument");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_attention_fwd: dropout needs a device seed pointer");
  AttnParams p;
  p.qkv = (const bf16_t*)qkv; p.mask = mask; p.ctx = (bf16_t*)ctx; p.lse = lse; p.dctx = nullptr; p.dqkv = nullptr;
  p.B = B; p.S = S; p.H = H; p.nh = nh; p.scale = 0.125f;
  p.drop_thr = vlb_drop_thr(drop_p); p.drop_scale = vlb_drop_scale(p.drop_thr); p.seed = seed; p.tag = tag;
  const int smem = 2 * TILE_BYTES + ATT_SP * 4;
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * nh), dim3(256), smem, stream, p);
  VLB_CHECK_LAUNCH("vlb_attention_fwd");
  return VLB_OK;
}

extern "C" int vlb_attention_bwd(const void* qkv, const float* mask, const void* ctx, const float* lse, const void* dctx,
                                 void* dqkv, int B, int S, int H, int nh, float drop_p, const uint32_t* seed, uint32_t tag,
                                 hipStream_t stream) {
  int rc = check_attn("vlb_attention_bwd", B, S, H, nh);
  if (rc) return rc;
  VLB_CHECK_ARG(qkv && mask && ctx && lse && dctx && dqkv, "vlb_attention_bwd: null argument");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_attention_bwd: dropout needs a device seed pointer");
  AttnParams p;
  p.qkv = (const bf16_t*)qkv; p.mask = mask; p.ctx = (bf16_t*)ctx; p.lse = (float*)lse; p.dctx = (const bf16_t*)dctx;
  p.dqkv = (bf16_t*)dqkv;
  p.B = B; p.S = S; p.H = H; p.nh = nh; p.scale = 0.125f;
  p.drop_thr = vlb_drop_thr(drop_p); p.drop_scale = vlb_drop_scale(p.drop_thr); p.seed = seed; p.tag = tag;
  const int smem = 7 * TILE_BYTES + 3 * ATT_SP * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(B * nh), dim3(512), smem, stream, p);
  VLB_CHECK_LAUNCH("vlb_attention_bwd");
  return VLB_OK;
}
