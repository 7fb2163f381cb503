// Fused multi-head self-attention forward / backward for short sequences (S <= 128, or S <= 256 with two 16-row slices per
// wave: the large / VCR configurations with 128 text + 100 regions; d = 64)
// on gfx950 -- BertSelfAttention (external/pytorch_pretrained_bert/modeling.py:290-319):
//     scores = Q K^T / sqrt(d) + (1-mask)*-10000 ; P = softmax(scores) ; P = dropout(P) ; ctx = P V
// The reference materialises [B,h,S,S] scores/probs in HBM (15.7 MB per layer at B=32) and runs
// softmax/dropout as separate elementwise passes; here one 8-wave workgroup owns one (batch, head):
// the [S][64] head slices of Q/K/V (and dO in backward) live in LDS as ROW-MAJOR images, the SxS tile
// lives only in registers, and the backward recomputes P from the saved row log-sum-exp.
//
// MFMA (v_mfma_f32_16x16x32_bf16) operand plan -- no register transposes, no transposed copies:
//  * score tiles are computed TRANSPOSED, S^T = K Q^T, with the 16 K-rows of a tile taken in the order
//    key = 32u + 8(i>>2) + 4*half + (i&3).  In the C layout (lane: col = lane&15, rows 4*(lane>>4)+r) a lane
//    then holds, for ONE query, keys 32u + 8g + [0..8) -- exactly the (col, k = 8g+j) B-operand layout of the
//    next MFMA whose reduction runs over keys (ctx^T = V^T P^T forward, dQ^T = K^T dS^T backward).
//  * reductions over QUERIES (dV^T = dO^T P, dK^T = Q^T dS) use the other orientation, S = Q K^T with permuted
//    Q rows, recomputed by the wave that owns those keys (MFMA is cheap, LDS round-trips of the tile are not).
//  * LDS images store sequence row s at row rho(s) = (s & ~31) | bit2(s)<<4 | bits34(s)<<2 | (s&3), i.e. in the
//    permuted order above: the 16 rows of an A-operand fragment are CONSECUTIVE (bank-conflict-free
//    ds_read_b128 with the GEMM's (row>>1)&7 chunk swizzle), and the four keys 8g+4h+[0..4) of a transposed
//    fragment are four consecutive rows too.
//  * "transposed" operands (V^T, K^T, Q^T, dO^T: lane = head-dim row, 8 consecutive keys/queries) come straight
//    from the row-major images through the LDS transpose read ds_read_b64_tr_b16 (two per fragment).
//  * every wave owns a 16-row slice (8 waves x 16 = 128) and plays both roles in backward (dK/dV of its keys, then
//    dQ of its queries) so the register budget stays under 128 VGPRs and two workgroups fit a CU (LDS 66 KB).
//  * every output fragment has 4 consecutive head-dim elements per lane -> 8-B bf16 stores.
// Dropout uses the counter RNG of vlb_common.h keyed on (b, h, q, key): forward and both backward
// orientations regenerate identical masks, nothing is stored.
#include "vlb_common.h"

#define ATT_SP_MAX 256  // longest padded sequence one workgroup handles (template NU = SP / 32 key blocks: 4 or 8)
#define ATT_D 64
#define ATT_THREADS 512

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// ---- LDS image addressing ------------------------------------------------------------------
__device__ __forceinline__ int rho(int s) { return (s & ~31) | (((s >> 2) & 1) << 4) | (((s >> 3) & 3) << 2) | (s & 3); }
// byte address of 16-B chunk `chunk` (0..7) of LDS row `row`
__device__ __forceinline__ int rm_addr(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ bf16x8 lds_frag(const char* base, int addr) { return *(const bf16x8*)(base + addr); }

// A-operand fragment in permuted order: tile (blk32, half) = LDS rows 32*blk + 16*half + [0..16)
__device__ __forceinline__ bf16x8 frag_perm(const char* img, int blk, int half, int c, int chunk) {
  return lds_frag(img, rm_addr(32 * blk + 16 * half + c, chunk));
}
// B-operand fragment for 16 consecutive sequence positions s0..s0+15 (lane c -> position s0+c)
__device__ __forceinline__ bf16x8 frag_nat(const char* img, int s0, int c, int chunk) {
  return lds_frag(img, rm_addr(rho(s0 + c), chunk));
}
// Transposed fragment: lane (i = L, g) <- X[seq 32*blk + 8g + j][d0 + L], j = 0..7   (two ds_read_b64_tr_b16)
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int blk, int d0, int lane) {
  const int L = lane & 15, g = lane >> 4;
  const int b = d0 * 2 + (L & 3) * 8;                    // byte offset of this lane's 4 columns inside the row
  const int r_lo = 32 * blk + 4 * g + (L >> 2);          // rho of keys 8g+0..3 ; +16 for keys 8g+4..7
  const int r_hi = r_lo + 16;
  const int a_lo = r_lo * 128 + ((((b >> 4) ^ ((r_lo >> 1) & 7)) << 4) | (b & 15));
  const int a_hi = r_hi * 128 + ((((b >> 4) ^ ((r_hi >> 1) & 7)) << 4) | (b & 15));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + a_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + a_hi));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// Staging of [S][64] head slices (row stride ld elements) into row-major LDS images at permuted rows, rows >= S zero -- in TWO steps,
// so that a kernel issues the global loads of ALL its tiles (and whatever else its prologue reads) before the first wait.  The loads
// are unconditional: a row index beyond the sequence is clamped to the last row and the value replaced by zeros afterwards.  (With a
// load under `if (s < S)` hipcc branches around every single load and waits vmcnt(0) behind it: the prologue of the backward kernel
// was ten dependent HBM round trips, 65 % of its wave cycles parked -- profiles/r03_gemm_pmc.txt, attn_bwd2_kernel.)
template <int SP>
struct TileRegs {
  uint4 v[SP * 8 / ATT_THREADS];
};

template <int SP>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ g, long ld, int S, int tid, TileRegs<SP>& r) {
#pragma unroll
  for (int it = 0; it < SP * 8 / ATT_THREADS; ++it) {
    const int P = it * ATT_THREADS + tid, s = min(P >> 3, S - 1), ch = P & 7;
    r.v[it] = *(const uint4*)(g + (long)s * ld + ch * 8);
  }
}

template <int SP>
__device__ __forceinline__ void tile_store(const TileRegs<SP>& r, int S, char* img, int tid) {
#pragma unroll
  for (int it = 0; it < SP * 8 / ATT_THREADS; ++it) {
    const int P = it * ATT_THREADS + tid, s = P >> 3, ch = P & 7;
    const bool in = s < S;
    const uint4 v = make_uint4(in ? r.v[it].x : 0u, in ? r.v[it].y : 0u, in ? r.v[it].z : 0u, in ? r.v[it].w : 0u);
    *(uint4*)(img + rm_addr(rho(s), ch)) = v;
  }
}

// D[q] = sum_d dO[q][d] * O[q][d] from the staged dO chunks and the matching O chunks (8 lanes per row, 8 elements each)
template <int SP>
__device__ __forceinline__ void rowdot_store(const TileRegs<SP>& a_, const TileRegs<SP>& o_, int S, float* sD, int tid) {
#pragma unroll
  for (int it = 0; it < SP * 8 / ATT_THREADS; ++it) {
    const int P = it * ATT_THREADS + tid, row = P >> 3, ch = P & 7;
    const uint4 a = a_.v[it], o = o_.v[it];
    float d = bflo(a.x) * bflo(o.x) + bfhi(a.x) * bfhi(o.x) + bflo(a.y) * bflo(o.y) + bfhi(a.y) * bfhi(o.y) +
              bflo(a.z) * bflo(o.z) + bfhi(a.z) * bfhi(o.z) + bflo(a.w) * bflo(o.w) + bfhi(a.w) * bfhi(o.w);
    d = row < S ? d : 0.f;
    d += __shfl_xor(d, 1, 64);
    d += __shfl_xor(d, 2, 64);
    d += __shfl_xor(d, 4, 64);
    if (ch == 0) sD[row] = d;
  }
}

__device__ __forceinline__ bf16x8 pack8(const float* v) {
  uint32_t w[4] = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  return __builtin_bit_cast(bf16x8, *(uint4*)w);
}

// keep-mask x scale for 8 consecutive element indices base..base+7 (any parity): the RNG yields one 32-bit hash per
// PAIR of elements (idx>>1), 16 bits each -- 5 hashes cover the (at most) 5 pairs the run of 8 touches.
__device__ __forceinline__ void drop8(float* v, uint32_t key, uint32_t base, uint32_t thr, float scale) {
  const uint32_t odd = base & 1u, p0 = base >> 1;
  uint32_t h[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) h[k] = vlb_pair_hash(p0 + k, key);
  // the 16-bit fields of the run, in element order, are the 160-bit string h[0..4] read from bit 16 * odd: one funnel shift per
  // pair (v_alignbit_b32) instead of two selects per element
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t w = __builtin_amdgcn_alignbit(h[k + 1], h[k], odd << 4);
    v[2 * k] = ((w & 0xffffu) >= thr) ? v[2 * k] * scale : 0.f;
    v[2 * k + 1] = ((w >> 16) >= thr) ? v[2 * k + 1] * scale : 0.f;
  }
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
// forward: wave w owns queries 16w .. 16w+15
// =============================================================================================
template <int NU, int QS>   // NU key blocks of 32 (padded length SP = 32 NU), QS 16-query slices per wave (8 waves x QS x 16 = SP)
__global__ __launch_bounds__(ATT_THREADS, NU <= 4 ? 4 : 2) void attn_fwd_kernel(const AttnParams p) {
  constexpr int ATT_SP = 32 * NU, TILE_BYTES = ATT_SP * ATT_D * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sK = smem;
  char* sV = smem + TILE_BYTES;
  float* sMB = (float*)(smem + 2 * TILE_BYTES);     // [SP] additive mask (-inf beyond S)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const bf16_t* qbase = p.qkv + (long)b * S * ld + h * ATT_D;
  // every global read of the prologue is issued before the first wait: K and V tiles, the first Q fragment, the mask row
  TileRegs<ATT_SP> rk, rv;
  tile_load<ATT_SP>(qbase + p.H, ld, S, tid, rk);
  tile_load<ATT_SP>(qbase + 2 * p.H, ld, S, tid, rv);
  // Q fragment straight from global in B-operand layout: lane (c, g) <- Q[q0+c][32ds+8g ..+8)
  bf16x8 qf[2];
  auto load_q = [&](int q0) {
    const int q = min(q0 + c, S - 1);
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) qf[ds] = *(const bf16x8*)(qbase + (long)q * ld + ds * 32 + g * 8);
  };
  load_q(wave * 16);
  const float mrow = p.mask[b * S + min(tid, S - 1)];
  const uint32_t seed = (p.drop_thr && p.seed) ? *p.seed : 0u;
  tile_store<ATT_SP>(rk, S, sK, tid);
  tile_store<ATT_SP>(rv, S, sV, tid);
  if (tid < ATT_SP) sMB[tid] = (tid < S) ? (1.0f - mrow) * -10000.0f : -INFINITY;
  __syncthreads();
  const uint32_t key = vlb_rng_key(seed, p.tag);
  const int U = (S + 31) >> 5;  // key blocks of 32 actually present

#pragma unroll 1
  for (int qs = 0; qs < QS; ++qs) {
  const int q0 = (wave + 8 * qs) * 16;
  if (q0 >= S) break;
  if (qs > 0) load_q(q0);

  f32x4 sc[NU][2];              // [u][half] : S^T tiles (keys x this wave's 16 queries)
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      sc[u][hf] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (u < U) {
#pragma unroll
        for (int ds = 0; ds < 2; ++ds)
          sc[u][hf] = VLB_MFMA_16x16x32(frag_perm(sK, u, hf, c, ds * 4 + g), qf[ds], sc[u][hf], 0, 0, 0);
      }
    }

  // softmax over keys for the lane's query: lane-local 32 values, then across g
  float mx = -INFINITY;
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = (u < U) ? sc[u][hf][r] * p.scale + sMB[32 * u + 8 * g + 4 * hf + r] : -INFINITY;
        sc[u][hf][r] = v;
        mx = fmaxf(mx, v);
      }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(sc[u][hf][r] - mx);
        sc[u][hf][r] = e;
        sum += e;
      }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
  const int q = q0 + c;
  if (g == 0 && q < S) p.lse[((long)b * p.nh + h) * S + q] = mx + __logf(sum);

  const uint32_t qrow = ((uint32_t)(b * p.nh + h) * (uint32_t)S + (uint32_t)min(q, S - 1)) * (uint32_t)S;
  bf16x8 pf[NU];  // [u]  P (after dropout) as MFMA operand, k = 8g + (4*half + r)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    float v[8];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * hf + r] = sc[u][hf][r] * inv;
    if (p.drop_thr) drop8(v, key, qrow + 32u * u + 8u * g, p.drop_thr, p.drop_scale);   // keys 32u+8g..+8 consecutive
    pf[u] = pack8(v);
  }

  // ctx^T[d][q] = sum_key V^T[d][key] P^T[key][q]     (all lanes take part in the wave-wide MFMAs)
  bf16_t* orow = p.ctx + ((long)b * S + min(q, S - 1)) * p.H + h * ATT_D + 4 * g;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (u < U) o = VLB_MFMA_16x16x32(frag_tr(sV, u, dt * 16, lane), pf[u], o, 0, 0, 0);
    if (q < S) *(uint2*)(orow + dt * 16) = make_uint2(pack2bf(o[0], o[1]), pack2bf(o[2], o[3]));   // rows d = dt*16 + 4g + r
  }
  }   // qs
}

// =============================================================================================
// backward: wave w owns sequence positions 16w .. 16w+15, first as KEYS (dK, dV; reductions over all
// queries), then as QUERIES (dQ; reduction over all keys)
// =============================================================================================
template <int NU, int QS>
__global__ __launch_bounds__(ATT_THREADS, NU <= 4 ? 4 : 2) void attn_bwd_kernel(const AttnParams p) {
  constexpr int ATT_SP = 32 * NU, TILE_BYTES = ATT_SP * ATT_D * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;
  char* sK = smem + 1 * TILE_BYTES;
  char* sV = smem + 2 * TILE_BYTES;
  char* sdO = smem + 3 * TILE_BYTES;
  float* sMB = (float*)(smem + 4 * TILE_BYTES);  // [128]
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
  // every global read of the prologue is issued before the first wait (see tile_load): the four tiles, O for the row dots, mask, lse
  TileRegs<ATT_SP> rq, rk, rv, rdo, ro;
  tile_load<ATT_SP>(qbase, ld, S, tid, rq);
  tile_load<ATT_SP>(qbase + p.H, ld, S, tid, rk);
  tile_load<ATT_SP>(qbase + 2 * p.H, ld, S, tid, rv);
  tile_load<ATT_SP>(dobase, p.H, S, tid, rdo);
  tile_load<ATT_SP>(obase, p.H, S, tid, ro);
  const float mrow = p.mask[b * S + min(tid, S - 1)];
  const float lrow = p.lse[((long)b * p.nh + h) * S + min(tid, S - 1)];
  const uint32_t seed = (p.drop_thr && p.seed) ? *p.seed : 0u;
  tile_store<ATT_SP>(rq, S, sQ, tid);
  tile_store<ATT_SP>(rk, S, sK, tid);
  tile_store<ATT_SP>(rv, S, sV, tid);
  tile_store<ATT_SP>(rdo, S, sdO, tid);
  rowdot_store<ATT_SP>(rdo, ro, S, sD, tid);      // D[q] = sum_d dO[q][d] * O[q][d]
  if (tid < ATT_SP) {
    sMB[tid] = (tid < S) ? (1.0f - mrow) * -10000.0f : -INFINITY;
    sLSE[tid] = (tid < S) ? lrow : 0.f;
  }
  __syncthreads();

  const uint32_t key = vlb_rng_key(seed, p.tag);
  const uint32_t bh = (uint32_t)(b * p.nh + h);
  const int U = (S + 31) >> 5;
#pragma unroll 1
  for (int qs = 0; qs < QS; ++qs) {
  const int w16 = (wave + 8 * qs) * 16;
  if (w16 >= S) break;
  const int pos = w16 + c;                          // this lane's key (role N) / query (role T)
  const bool pos_ok = pos < S;

  // ------------------------------------------------------------------------------------------
  // role N: keys pos;  S[q][key] tiles = Q(perm rows) K^T ;  dV^T = dO^T Pd ,  dK^T = Q^T dS
  // ------------------------------------------------------------------------------------------
  {
    bf16x8 kf[2], vf[2];  // B operands: lane (c,g) <- K/V[key = w16+c][32ds+8g..]
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      kf[ds] = frag_nat(sK, w16, c, ds * 4 + g);
      vf[ds] = frag_nat(sV, w16, c, ds * 4 + g);
    }
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float mb = sMB[pos];
    const uint32_t kcol = (uint32_t)min(pos, S - 1);
    // (the 8-block instantiation is NOT unrolled across blocks: interleaving eight iterations blew the register budget)
#pragma unroll (NU <= 4 ? NU : 1)
    for (int v = 0; v < NU; ++v) {
      if (v < U) {
        float pv[8], dsv[8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ds = 0; ds < 2; ++ds) {
            s = VLB_MFMA_16x16x32(frag_perm(sQ, v, hf, c, ds * 4 + g), kf[ds], s, 0, 0, 0);
            dp = VLB_MFMA_16x16x32(frag_perm(sdO, v, hf, c, ds * 4 + g), vf[ds], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int q = 32 * v + 8 * g + 4 * hf + r;
            const float pr = __expf(s[r] * p.scale + mb - sLSE[q]);
            float keepf = 1.f;
            if (p.drop_thr) {
              const uint32_t idx = (bh * (uint32_t)S + (uint32_t)min(q, S - 1)) * (uint32_t)S + kcol;
              const uint32_t hsh = vlb_pair_hash(idx >> 1, key);
              keepf = (((idx & 1u) ? (hsh >> 16) : (hsh & 0xffffu)) >= p.drop_thr) ? p.drop_scale : 0.f;
            }
            pv[4 * hf + r] = pr * keepf;
            dsv[4 * hf + r] = pr * (dp[r] * keepf - sD[q]);
          }
        }
        const bf16x8 pd = pack8(pv), dsf = pack8(dsv);   // operands: col = key c, k = query 32v + 8g + j
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = VLB_MFMA_16x16x32(frag_tr(sdO, v, dt * 16, lane), pd, dv[dt], 0, 0, 0);
          dk[dt] = VLB_MFMA_16x16x32(frag_tr(sQ, v, dt * 16, lane), dsf, dk[dt], 0, 0, 0);
        }
      }
    }
    if (pos_ok) {
      bf16_t* orow = p.dqkv + ((long)b * S + pos) * ld + h * ATT_D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(uint2*)(orow + p.H + dt * 16) = make_uint2(pack2bf(dk[dt][0] * p.scale, dk[dt][1] * p.scale),
                                                     pack2bf(dk[dt][2] * p.scale, dk[dt][3] * p.scale));
        *(uint2*)(orow + 2 * p.H + dt * 16) = make_uint2(pack2bf(dv[dt][0], dv[dt][1]), pack2bf(dv[dt][2], dv[dt][3]));
      }
    }
  }

  // ------------------------------------------------------------------------------------------
  // role T: queries pos;  S^T[key][q] tiles = K(perm rows) Q^T ;  dQ^T = K^T dS^T
  // ------------------------------------------------------------------------------------------
  {
    bf16x8 qf[2], dof[2];  // B operands: lane (c,g) <- Q/dO[q = w16+c][32ds+8g..]
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      qf[ds] = frag_nat(sQ, w16, c, ds * 4 + g);
      dof[ds] = frag_nat(sdO, w16, c, ds * 4 + g);
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float lse = sLSE[pos], dd = sD[pos];
    const uint32_t qrow = (bh * (uint32_t)S + (uint32_t)min(pos, S - 1)) * (uint32_t)S;
#pragma unroll (NU <= 4 ? NU : 1)
    for (int u = 0; u < NU; ++u) {
      if (u < U) {
        float pr[8], dpv[8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ds = 0; ds < 2; ++ds) {
            s = VLB_MFMA_16x16x32(frag_perm(sK, u, hf, c, ds * 4 + g), qf[ds], s, 0, 0, 0);
            dp = VLB_MFMA_16x16x32(frag_perm(sV, u, hf, c, ds * 4 + g), dof[ds], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pr[4 * hf + r] = __expf(s[r] * p.scale + sMB[32 * u + 8 * g + 4 * hf + r] - lse);
            dpv[4 * hf + r] = dp[r];
          }
        }
        if (p.drop_thr) drop8(dpv, key, qrow + 32u * u + 8u * g, p.drop_thr, p.drop_scale);   // dP = dPd * keep * scale
        float dsv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) dsv[j] = pr[j] * (dpv[j] - dd);
        const bf16x8 dsf = pack8(dsv);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dq[dt] = VLB_MFMA_16x16x32(frag_tr(sK, u, dt * 16, lane), dsf, dq[dt], 0, 0, 0);
      }
    }
    if (pos_ok) {
      bf16_t* orow = p.dqkv + ((long)b * S + pos) * ld + h * ATT_D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *(uint2*)(orow + dt * 16) = make_uint2(pack2bf(dq[dt][0] * p.scale, dq[dt][1] * p.scale),
                                               pack2bf(dq[dt][2] * p.scale, dq[dt][3] * p.scale));
    }
  }
  }   // qs
}

// =============================================================================================
// backward, S <= 128, one evaluation of the S x S tile (round 2).
//
// The kernel above computes P and dS TWICE -- once per orientation -- because the reduction over queries (dV, dK) needs them with
// queries along the MFMA k index and the reduction over keys (dQ) with keys along it.  Measured, it is VALU-bound (80 % of the
// SIMD cycles issuing: exp, the dropout hash -- one per ELEMENT in the key-major orientation --, dS), not MFMA- or LDS-bound.
// Here every wave evaluates its 16 queries x all keys once (the query-major role: exp, one hash per element PAIR, dS, dQ), keeps
// the two bf16 tiles in registers (32 VGPRs), and after a barrier hands them over through the LDS space K and V no longer need:
// row-major [query][key] images, 16-B stores, XOR-swizzled 16-B chunks.  The key-major role then only runs the two products that
// reduce over queries, reading its B operands (8 consecutive queries of one key) with the LDS transpose read ds_read_b64_tr_b16
// -- no second exp / hash / dP.  The images share the 32 KiB of sK | sV, Pd first (dV), then dS (dK): LDS stays at 66 KiB, two
// workgroups per CU.
// =============================================================================================
__device__ __forceinline__ int pt_addr(int row, int chunk16) { return row * 256 + ((chunk16 ^ (row & 15)) << 4); }

// B operand of a product that reduces over QUERIES: lane (L = lane & 15, g) <- T[query 32 v + 8 g + j][key key0 + L], j = 0..7
__device__ __forceinline__ bf16x8 frag_pt(const char* img, int v, int key0, int lane) {
  const int L = lane & 15, g = lane >> 4;
  const int r_lo = 32 * v + 8 * g + (L >> 2), r_hi = r_lo + 4;
  const int cb = key0 * 2 + (L & 3) * 8;                       // byte offset of this lane's 4 keys inside the 256-B row
  const int a_lo = pt_addr(r_lo, cb >> 4) | (cb & 15), a_hi = pt_addr(r_hi, cb >> 4) | (cb & 15);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + a_lo));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + a_hi));
  const s16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, w);
}

__global__ __launch_bounds__(ATT_THREADS, 4) void attn_bwd2_kernel(const AttnParams p) {
  constexpr int NU = 4, ATT_SP = 128, TILE_BYTES = ATT_SP * ATT_D * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sQ = smem;
  char* sK = smem + 1 * TILE_BYTES;
  char* sV = smem + 2 * TILE_BYTES;
  char* sdO = smem + 3 * TILE_BYTES;
  char* sT = sK;                                   // [128 queries][128 keys] bf16 = sK | sV, after the query-major role
  float* sMB = (float*)(smem + 4 * TILE_BYTES);
  float* sLSE = sMB + ATT_SP;
  float* sD = sLSE + ATT_SP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / p.nh, h = blockIdx.x % p.nh;
  const int S = p.S;
  const long ld = 3L * p.H;
  const bf16_t* qbase = p.qkv + (long)b * S * ld + h * ATT_D;
  const bf16_t* dobase = p.dctx + (long)b * S * p.H + h * ATT_D;
  const bf16_t* obase = p.ctx + (long)b * S * p.H + h * ATT_D;
  // every global read of the prologue is issued before the first wait (see tile_load): the four tiles, O for the row dots, mask, lse
  TileRegs<ATT_SP> rq, rk, rv, rdo, ro;
  tile_load<ATT_SP>(qbase, ld, S, tid, rq);
  tile_load<ATT_SP>(qbase + p.H, ld, S, tid, rk);
  tile_load<ATT_SP>(qbase + 2 * p.H, ld, S, tid, rv);
  tile_load<ATT_SP>(dobase, p.H, S, tid, rdo);
  tile_load<ATT_SP>(obase, p.H, S, tid, ro);
  const float mrow = p.mask[b * S + min(tid, S - 1)];
  const float lrow = p.lse[((long)b * p.nh + h) * S + min(tid, S - 1)];
  const uint32_t seed = (p.drop_thr && p.seed) ? *p.seed : 0u;
  tile_store<ATT_SP>(rq, S, sQ, tid);
  tile_store<ATT_SP>(rk, S, sK, tid);
  tile_store<ATT_SP>(rv, S, sV, tid);
  tile_store<ATT_SP>(rdo, S, sdO, tid);
  rowdot_store<ATT_SP>(rdo, ro, S, sD, tid);      // D[q] = sum_d dO[q][d] * O[q][d]
  if (tid < ATT_SP) {
    sMB[tid] = (tid < S) ? (1.0f - mrow) * -10000.0f : -INFINITY;
    sLSE[tid] = (tid < S) ? lrow : 0.f;
  }
  __syncthreads();

  const uint32_t key = vlb_rng_key(seed, p.tag);
  const uint32_t bh = (uint32_t)(b * p.nh + h);
  const int U = (S + 31) >> 5;
  const int w16 = wave * 16;
  const int pos = w16 + c;                          // this lane's query (first role) / key (second role)
  const bool pos_ok = pos < S;
  const bool active = w16 < 32 * U;                 // this wave's 16 rows lie inside the processed 32-blocks

  // ------------------------------------------------------------------------------------------
  // query-major role: S^T[key][q] tiles = K(perm rows) Q^T ;  P, Pd, dS ;  dQ^T = K^T dS^T
  // ------------------------------------------------------------------------------------------
  bf16x8 pd[NU], dsf[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) pd[u] = dsf[u] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  if (w16 < S) {
    bf16x8 qf[2], dof[2];
#pragma unroll
    for (int ds = 0; ds < 2; ++ds) {
      qf[ds] = frag_nat(sQ, w16, c, ds * 4 + g);
      dof[ds] = frag_nat(sdO, w16, c, ds * 4 + g);
    }
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float lse = sLSE[pos], dd = sD[pos];
    const float rowf = pos_ok ? 1.f : 0.f;          // queries beyond S contribute nothing to the reductions over queries
    const uint32_t qrow = (bh * (uint32_t)S + (uint32_t)min(pos, S - 1)) * (uint32_t)S;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      if (u < U) {
        float pr[8], dpv[8], kv[8];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          f32x4 sc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ds = 0; ds < 2; ++ds) {
            sc = VLB_MFMA_16x16x32(frag_perm(sK, u, hf, c, ds * 4 + g), qf[ds], sc, 0, 0, 0);
            dp = VLB_MFMA_16x16x32(frag_perm(sV, u, hf, c, ds * 4 + g), dof[ds], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pr[4 * hf + r] = __expf(sc[r] * p.scale + sMB[32 * u + 8 * g + 4 * hf + r] - lse) * rowf;
            dpv[4 * hf + r] = dp[r];
            kv[4 * hf + r] = 1.f;
          }
        }
        if (p.drop_thr) drop8(kv, key, qrow + 32u * u + 8u * g, p.drop_thr, p.drop_scale);      // keep * 1/(1-p) per element
        float pdv[8], dsv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pdv[j] = pr[j] * kv[j];
          dsv[j] = pr[j] * (dpv[j] * kv[j] - dd);
        }
        pd[u] = pack8(pdv);
        dsf[u] = pack8(dsv);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dq[dt] = VLB_MFMA_16x16x32(frag_tr(sK, u, dt * 16, lane), dsf[u], dq[dt], 0, 0, 0);
      }
    }
    if (pos_ok) {
      bf16_t* orow = p.dqkv + ((long)b * S + pos) * ld + h * ATT_D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *(uint2*)(orow + dt * 16) = make_uint2(pack2bf(dq[dt][0] * p.scale, dq[dt][1] * p.scale),
                                               pack2bf(dq[dt][2] * p.scale, dq[dt][3] * p.scale));
    }
  }
  __syncthreads();                                  // every wave is done with sK / sV

  // ------------------------------------------------------------------------------------------
  // key-major role: dV^T = dO^T Pd ,  dK^T = Q^T dS  -- operands from the [query][key] images
  // ------------------------------------------------------------------------------------------
  f32x4 dv[4], dk[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (active) {                                     // lane (c, g): query row w16 + c, keys 32 u + 8 g + [0, 8)  (zeros beyond S)
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (u < U) *(bf16x8*)(sT + pt_addr(w16 + c, 4 * u + g)) = pd[u];
  }
  __syncthreads();
  if (w16 < S) {
#pragma unroll
    for (int v = 0; v < NU; ++v) {
      if (v < U) {
        const bf16x8 pt = frag_pt(sT, v, w16, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dv[dt] = VLB_MFMA_16x16x32(frag_tr(sdO, v, dt * 16, lane), pt, dv[dt], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                  // Pd consumed: the image is reused for dS
  if (active) {
#pragma unroll
    for (int u = 0; u < NU; ++u)
      if (u < U) *(bf16x8*)(sT + pt_addr(w16 + c, 4 * u + g)) = dsf[u];
  }
  __syncthreads();
  if (w16 < S) {
#pragma unroll
    for (int v = 0; v < NU; ++v) {
      if (v < U) {
        const bf16x8 st = frag_pt(sT, v, w16, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dk[dt] = VLB_MFMA_16x16x32(frag_tr(sQ, v, dt * 16, lane), st, dk[dt], 0, 0, 0);
      }
    }
    if (pos_ok) {
      bf16_t* orow = p.dqkv + ((long)b * S + pos) * ld + h * ATT_D + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *(uint2*)(orow + p.H + dt * 16) = make_uint2(pack2bf(dk[dt][0] * p.scale, dk[dt][1] * p.scale),
                                                     pack2bf(dk[dt][2] * p.scale, dk[dt][3] * p.scale));
        *(uint2*)(orow + 2 * p.H + dt * 16) = make_uint2(pack2bf(dv[dt][0], dv[dt][1]), pack2bf(dv[dt][2], dv[dt][3]));
      }
    }
  }
}

// ---------------------------------------------------------------------------------- C ABI
static int check_attn(const char* name, int B, int S, int H, int nh) {
  VLB_CHECK_ARG(B > 0 && S > 0 && S <= ATT_SP_MAX, "%s: S=%d unsupported (1..%d)", name, S, ATT_SP_MAX);
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
  if (S <= 128) {
    const int smem = 2 * (128 * ATT_D * 2) + 128 * 4;
    hipLaunchKernelGGL((attn_fwd_kernel<4, 1>), dim3(B * nh), dim3(ATT_THREADS), smem, stream, p);
  } else {
    const int smem = 2 * (256 * ATT_D * 2) + 256 * 4;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    hipLaunchKernelGGL((attn_fwd_kernel<8, 2>), dim3(B * nh), dim3(ATT_THREADS), smem, stream, p);
  }
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
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (128 * ATT_D * 2) + 3 * 128 * 4);
    (void)hipFuncSetAttribute((const void*)attn_bwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (128 * ATT_D * 2) + 3 * 128 * 4);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<8, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (256 * ATT_D * 2) + 3 * 256 * 4);
    attr_set = true;
  }
  static int bwd2 = -1;      // VLB_ATTN_BWD2: 1 (default) the single-evaluation backward for S <= 128; 0 the two-orientation kernel
  if (bwd2 < 0) {
    const char* v = getenv("VLB_ATTN_BWD2");
    bwd2 = v ? atoi(v) : 1;
  }
  if (S <= 128 && bwd2)
    hipLaunchKernelGGL(attn_bwd2_kernel, dim3(B * nh), dim3(ATT_THREADS), 4 * (128 * ATT_D * 2) + 3 * 128 * 4, stream, p);
  else if (S <= 128)
    hipLaunchKernelGGL((attn_bwd_kernel<4, 1>), dim3(B * nh), dim3(ATT_THREADS), 4 * (128 * ATT_D * 2) + 3 * 128 * 4, stream, p);
  else
    hipLaunchKernelGGL((attn_bwd_kernel<8, 2>), dim3(B * nh), dim3(ATT_THREADS), 4 * (256 * ATT_D * 2) + 3 * 256 * 4, stream, p);
  VLB_CHECK_LAUNCH("vlb_attention_bwd");
  return VLB_OK;
}
