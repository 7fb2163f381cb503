// bf16 "NT" GEMM for gfx950 (CDNA4):   C[M,N] (+)= A[M,K] · B[N,K]^T   (fp32 accumulate)
//
// Every matmul on the VL-BERT hot path is brought to this form by the host:
//   forward   Y  = X · W^T            (torch Linear weights are [out,in] = [N,K])
//   dgrad     dX = dY · (W^T)^T       (a bf16 W^T copy is kept next to W)
//   wgrad     dW = dY^T · (X^T)^T     (reduction over the M rows; split-K + fp32 atomics)
// (replaces the cuBLAS calls behind nn.Linear / torch.matmul in
//  external/pytorch_pretrained_bert/modeling.py:291-300,312,330,362,375,469.)
//
// Kernel structure (see DESIGN.md "GEMM"):
//   * 256 threads = 4 waves (2x2), block tile BMxBN (128x128 or 128x64), BK = 64;
//   * both operand tiles are staged HBM -> LDS with `global_load_lds_dwordx4` (LDS-DMA,
//     16 B/lane, no VGPR round trip), double buffered, one barrier per K tile;
//   * LDS image is [row][64 bf16] = 128-B rows; the 16-B slot of logical k-chunk kc of row r is
//     kc ^ ((r>>1)&7)  (XOR swizzle applied on the *source* address because the LDS-DMA
//     destination is lane-linear) -> ds_read_b128 fragment reads are bank-conflict free;
//   * v_mfma_f32_16x16x32_bf16 with the operands swapped (weights as the "A" operand) so
//     each lane ends up with 4 consecutive output columns of one row -> 8-B bf16 / 16-B
//     fp32 epilogue accesses;
//   * fused epilogue: +bias, erf-GELU (optionally also storing the pre-activation), ReLU,
//     x gelu'(aux) or x aux (GELU backward), dropout (counter RNG), +residual, bf16 or fp32 store,
//     fp32 atomic accumulation for split-K weight gradients.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "vlb_common.h"

#include "gemm_params.h"


// Epilogue specialised at compile time on (activation, dropout, residual, output mode, interior tile) so each
// variant is a straight-line body: interior tiles (the overwhelming majority) carry no bounds checks, row base
// pointers are formed once per fragment row and the 16-column fragment steps fold into immediate offsets; the
// bias vector depends only on the column fragment and is loaded once per column.
// OUT: 0 bf16 store | 1 fp32 store | 2 fp32 atomicAdd | 3 fp32 read-modify-write accumulate (tile owned by one block)
//
// bf16 results of interior tiles are STAGED through LDS (the stage buffer the K loop just finished with): the MFMA
// layout gives a lane 4 columns of one row, i.e. a wave store would touch 16 rows x 32 B -- half cache lines, and
// the L2 write-transaction rate (not bytes) bounded the short-K GEMMs.  From LDS every lane writes 16 B and a wave
// covers whole tile rows.  Tile image: [128 rows][BN*2 B], 16-B chunk index XOR (row & (BN/8-1)): conflict-free for
// the ds_write_b64 fragment writes, whose 16-lane groups hold 16 different rows at one column.
struct EpiStage {
  char* lds;         // 32 KiB staging image
  int m0, n0;        // tile origin
  int row_l, col_l;  // this lane's position inside the tile for fragment (0,0): row = row_l + 16 i, col = col_l + 16 j
  int tid, nthreads, bn;
};

__device__ __forceinline__ void epi_stage_put(const EpiStage& st, int i, int j, const float (&v)[4]) {
  const int row = st.row_l + i * 16, colb = (st.col_l + j * 16) * 2;
  const int pitch = st.bn * 2, mask = (st.bn >> 3) - 1;      // image rows are exactly BN bf16 wide
  *(uint2*)(st.lds + row * pitch + ((((colb >> 4) ^ (row & mask)) << 4) | (colb & 15))) =
      make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

__device__ __forceinline__ void epi_copy_out(const EpiStage& st, bf16_t* C, long ldc) {
  __syncthreads();
  const int ch_per_row = st.bn >> 3;                  // 16-B chunks per tile row (16 for BN=128, 8 for BN=64)
  for (int q = st.tid; q < 128 * ch_per_row; q += st.nthreads) {
    const int row = q / ch_per_row, ch = q % ch_per_row;
    const uint4 v = *(const uint4*)(st.lds + row * (st.bn * 2) + ((ch ^ (row & (ch_per_row - 1))) << 4));
    *(uint4*)(C + (long)(st.m0 + row) * ldc + st.n0 + ch * 8) = v;
  }
  __syncthreads();
}

// GEN (the run-time dispatched kernel only): also honours the LayerNorm-residual form (p.res_stats) and fp16 output (p.c_f16)
template <int ACT, bool DROP, bool RES, int OUT, bool INTERIOR, int FM, int FN, bool GEN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[FM][FN], int mb, int nb, const EpiStage& st) {
  const bool f16out = GEN && p.c_f16 != 0;
  const bool lnres = GEN && RES && p.res_stats != nullptr;
  const bool STAGED = INTERIOR && OUT == 0 && !f16out;       // block-uniform: every thread takes the same path
  const uint32_t seed = DROP ? *p.seed : 0u;
  float bias[FN][4];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int n = nb + j * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[j][r] = 0.f;
    if (p.bias) {
      if (INTERIOR || n + 3 < p.N) {
        const float4 b = *(const float4*)(p.bias + n);
        bias[j][0] = b.x; bias[j][1] = b.y; bias[j][2] = b.z; bias[j][3] = b.w;
      } else {
        for (int r = 0; r < 4; ++r) if (n + r < p.N) bias[j][r] = p.bias[n + r];
      }
    }
  }
  constexpr bool GELU = (ACT == 1 || ACT == 4);
  const bool pre_staged = STAGED && GELU && p.pre;
  if (pre_staged) {   // first pass: the tile saved for the GELU backward (pre-activation, or gelu'(x) for act 4)
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = acc[i][j][r] + bias[j][r];
          if (ACT == 4) {   // acc <- gelu(x) (final), v <- gelu'(x)
            float g, d;
            gelu_both(v[r], g, d);
            acc[i][j][r] = g;
            v[r] = d;
          } else {
            acc[i][j][r] = v[r];
          }
        }
        epi_stage_put(st, i, j, v);
        if (ACT == 4) asm volatile("" ::: "memory");   // keep the scheduler from interleaving all 32 gelu_both chains (it spilled)
      }
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) bias[j][r] = 0.f;
    epi_copy_out(st, p.pre, p.ldpre);
  }
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = mb + i * 16;
    if (!INTERIOR && m >= p.M) continue;
    const long offc = (long)m * p.ldc + nb + (long)blockIdx.y * p.c_split_stride;
    const bf16_t* aux_row = (ACT == 3 || ACT == 5 || ACT == 8) ? p.aux + (long)m * p.ldaux + nb : nullptr;
    bf16_t* pre_row = (GELU && p.pre && !STAGED) ? p.pre + (long)m * p.ldpre + nb : nullptr;
    const bf16_t* res_row = RES ? p.res + (long)m * p.ldres + nb : nullptr;
    const uint32_t idx_row = DROP ? (uint32_t)m * (uint32_t)p.N + (uint32_t)nb : 0u;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = nb + j * 16;
      if (!INTERIOR && n >= p.N) continue;
      const bool full = INTERIOR || (n + 3 < p.N);
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] + bias[j][r];
      if (GELU) {
        if (ACT == 4) {
          if (!pre_staged) {             // (staged: gelu already applied in the first pass)
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float g;
              gelu_both(v[r], g, d[r]);
              v[r] = g;
            }
            if (pre_row) {
              bf16_t* q = pre_row + j * 16;
              if (full) *(uint2*)q = make_uint2(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]));
              else for (int r = 0; r < 4 && n + r < p.N; ++r) q[r] = f2bf(d[r]);
            }
          }
        } else {
          if (pre_row) {
            bf16_t* q = pre_row + j * 16;
            if (full) *(uint2*)q = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
            else for (int r = 0; r < 4 && n + r < p.N; ++r) q[r] = f2bf(v[r]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
        }
      } else if (ACT == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (ACT == 6) {     // BertPooler (modeling.py:430-436): tanh(x) = 1 - 2 / (exp(2x) + 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * v[r]) + 1.0f);
      } else if (ACT == 3 || ACT == 5) {
        float u[4] = {0.f, 0.f, 0.f, 0.f};
        const bf16_t* q = aux_row + j * 16;
        if (full) {
          const uint2 w = *(const uint2*)q;
          u[0] = bflo(w.x); u[1] = bfhi(w.x); u[2] = bflo(w.y); u[3] = bfhi(w.y);
        } else {
          for (int r = 0; r < 4 && n + r < p.N; ++r) u[r] = bf2f(q[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= (ACT == 3) ? dgelu_f(u[r]) : u[r];
      }
      if (DROP) {
        const uint32_t idx = idx_row + j * 16;
        if ((idx & 1u) == 0) {  // n%4==0: element pairs (idx,idx+1), (idx+2,idx+3) share one hash each
          const uint32_t h0 = vlb_rng_pair(seed, p.tag, idx >> 1), h1 = vlb_rng_pair(seed, p.tag, (idx >> 1) + 1);
          v[0] = ((h0 & 0xffffu) >= p.drop_thr) ? v[0] * p.drop_scale : 0.f;
          v[1] = ((h0 >> 16) >= p.drop_thr) ? v[1] * p.drop_scale : 0.f;
          v[2] = ((h1 & 0xffffu) >= p.drop_thr) ? v[2] * p.drop_scale : 0.f;
          v[3] = ((h1 >> 16) >= p.drop_thr) ? v[3] * p.drop_scale : 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = vlb_keep(seed, p.tag, idx + r, p.drop_thr) ? v[r] * p.drop_scale : 0.f;
        }
      }
      if (RES) {
        const bf16_t* q = res_row + j * 16;
        if (lnres) {     // residual = LayerNorm output re-materialised from the fp16 pre-LN row, its (mean, rstd) and gamma / beta
          const float2 ms = *(const float2*)(p.res_stats + 2 * (long)m);
          if (full) {    // whole 4-column group: one 8-B row load, two 16-B parameter loads (the scalar form below is 13 loads)
            const uint2 w = *(const uint2*)q;
            const float4 g4 = *(const float4*)(p.res_gamma + n), b4 = *(const float4*)(p.res_beta + n);
            v[0] += fmaf((hlo(w.x) - ms.x) * ms.y, g4.x, b4.x);
            v[1] += fmaf((hhi(w.x) - ms.x) * ms.y, g4.y, b4.y);
            v[2] += fmaf((hlo(w.y) - ms.x) * ms.y, g4.z, b4.z);
            v[3] += fmaf((hhi(w.y) - ms.x) * ms.y, g4.w, b4.w);
          } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) v[r] += fmaf((h2f(q[r]) - ms.x) * ms.y, p.res_gamma[n + r], p.res_beta[n + r]);
          }
        } else if (full) {
          const uint2 w = *(const uint2*)q;
          v[0] += bflo(w.x); v[1] += bfhi(w.x); v[2] += bflo(w.y); v[3] += bfhi(w.y);
        } else {
          for (int r = 0; r < 4 && n + r < p.N; ++r) v[r] += bf2f(q[r]);
        }
      }
      if (ACT == 7) {            // Bottleneck tail (common/backbone/resnet/resnet.py:112-116): relu(conv + shift + residual)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
      } else if (ACT == 8) {     // ReLU backward on a gradient: (acc [+ res]) where the saved activation aux > 0
        const bf16_t* q = aux_row + j * 16;
        if (full) {
          const uint2 w = *(const uint2*)q;
          v[0] = bflo(w.x) > 0.f ? v[0] : 0.f; v[1] = bfhi(w.x) > 0.f ? v[1] : 0.f;
          v[2] = bflo(w.y) > 0.f ? v[2] : 0.f; v[3] = bfhi(w.y) > 0.f ? v[3] : 0.f;
        } else {
          for (int r = 0; r < 4 && n + r < p.N; ++r) v[r] = bf2f(q[r]) > 0.f ? v[r] : 0.f;
        }
      }
      if (OUT == 0) {
        if (STAGED) {
          epi_stage_put(st, i, j, v);
        } else {
          bf16_t* c = (bf16_t*)p.C + offc + j * 16;
          if (full) *(uint2*)c = make_uint2(pack2o(v[0], v[1], f16out), pack2o(v[2], v[3], f16out));
          else for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] = f16out ? f2h(v[r]) : f2bf(v[r]);
        }
      } else if (OUT == 1) {
        float* c = (float*)p.C + offc + j * 16;
        if (full) *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
        else for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] = v[r];
      } else if (OUT == 2) {
        float* c = (float*)p.C + offc + j * 16;
        for (int r = 0; r < 4 && (INTERIOR || n + r < p.N); ++r) atomicAdd(c + r, v[r]);
      } else {
        float* c = (float*)p.C + offc + j * 16;
        if (full) {
          const float4 o = *(const float4*)c;
          *(float4*)c = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
        } else {
          for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] += v[r];
        }
      }
    }
  }
  if (STAGED) epi_copy_out(st, (bf16_t*)p.C, p.ldc);
}

template <bool INTERIOR, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_dispatch(const GemmParams& p, f32x4 (&acc)[FM][FN], int mb, int nb, const EpiStage& st) {
  if (p.out_f32 == 3) gemm_epilogue<0, false, false, 3, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.out_f32 == 2) gemm_epilogue<0, false, false, 2, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.out_f32 == 1) gemm_epilogue<0, false, false, 1, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 1) gemm_epilogue<1, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 2) gemm_epilogue<2, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 3) gemm_epilogue<3, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 4) gemm_epilogue<4, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 5) gemm_epilogue<5, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 6) gemm_epilogue<6, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 7) gemm_epilogue<7, false, true, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else if (p.act == 8) {
    if (p.res) gemm_epilogue<8, false, true, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
    else gemm_epilogue<8, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  } else if (p.res) {
    if (p.drop_thr) gemm_epilogue<0, true, true, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
    else gemm_epilogue<0, false, true, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  } else if (p.drop_thr) gemm_epilogue<0, true, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
  else gemm_epilogue<0, false, false, 0, INTERIOR, FM, FN, true>(p, acc, mb, nb, st);
}

// EPI >= 0 compiles ONE fused epilogue into the kernel (bf16 output): 0 bias | 1 bias+GELU, gelu'(x) -> pre |
// 2 x aux | 3 bias+dropout+residual | 4 bias+residual | 5 bias+ReLU | 6 relu(bias+residual) | 7 (acc+residual) where
// aux>0 | 8 acc where aux>0 (6-8: the frozen-BN Bottleneck forward tail and its ReLU backward, vision.py).  With every variant inlined behind the runtime
// dispatch (EPI = -1, kept for the fp32 / rarely used forms) the 128-register 8-wave kernel spilled in its epilogues.
template <int EPI, bool INTERIOR, int FM, int FN>
__device__ __forceinline__ void gemm_epilogue_select(const GemmParams& p, f32x4 (&acc)[FM][FN], int mb, int nb, const EpiStage& st) {
  if constexpr (EPI < 0) {
    gemm_epilogue_dispatch<INTERIOR, FM, FN>(p, acc, mb, nb, st);
  } else {
    constexpr int ACT = EPI == 1 ? 4 : EPI == 2 ? 5 : EPI == 5 ? 2 : EPI == 6 ? 7 : (EPI == 7 || EPI == 8) ? 8 : 0;
    gemm_epilogue<ACT, EPI == 3, EPI == 3 || EPI == 4 || EPI == 6 || EPI == 7, 0, INTERIOR, FM, FN>(p, acc, mb, nb, st);
  }
}

template <int BM, int BN, int WGM, int WGN, int EPI, bool CONV = false>   // WGM x WGN waves per workgroup
// 2 workgroups per CU (LDS-limited) must also fit the register file: 16 waves/CU = 4 per SIMD for the 8-wave shape
// (<= 128 VGPRs), 2 per SIMD for the 4-wave shape; the second launch-bound argument is waves per SIMD.
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN >= 8) ? 4 : 2) void gemm_nt_bf16_kernel(const GemmParams p) {
  constexpr int BK = 64;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int NA = BM * 8 / NT, NB = BN * 8 / NT;   // 16-B chunks per thread per stage
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nt = p.ntm * p.ntn;
  const int k_begin = blockIdx.y * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int ntk = (k_end - k_begin) / BK;
  if ((int)blockIdx.x >= nt || ntk <= 0) return;
  // PERSISTENT workgroups: block b walks work items w = b, b+grid, b+2*grid, ... (grid = resident workgroups).
  // w -> tile: XCD-aware (block b runs on XCD b%8 and grid%8==0, so w%8 is this block's XCD: each XCD owns a
  // contiguous run of tiles, bijective for any tile count), then visited in groups of `tile_group` tile-rows,
  // column-major inside the group, so the ~64 workgroups resident on one XCD share <= tile_group A panels and a
  // few B panels (working set fits the 4 MB L2) instead of sweeping all of B for every A panel.
  auto tile_of = [&](int w, int& m0, int& n0) {
    const int xcd = w & 7, q = nt >> 3, r = nt & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    const int gm = p.tile_group, per_group = gm * p.ntn, gid = t / per_group, first = gid * gm;
    const int gsz = min(p.ntm - first, gm), rem = t - gid * per_group;
    m0 = (first + rem % gsz) * BM;
    n0 = (rem / gsz) * BN;
  };
  // per-thread staging addresses (16-B chunks; chunk P -> LDS byte 16*P):
  // P = it*NT + tid ; row r = P>>3 ; physical slot s = P&7 ; logical k-chunk kc = s ^ ((r>>1)&7)
  // CONV: a_src = the centre pixel's channel chunk, a_msk = which of the 9 taps fall inside the image for that pixel
  auto setup = [&](const bf16_t* (&a_src)[NA], uint32_t (&a_msk)[NA], const bf16_t* (&b_src)[NB], int m0, int n0) {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int P = it * NT + tid, r = P >> 3, kc = (P & 7) ^ ((r >> 1) & 7);
      const int m = min(m0 + r, p.M - 1);
      a_src[it] = p.A + (long)m * p.lda + k_begin + kc * 8;
      if constexpr (CONV) {
        const int ox = m % p.conv_W, oy = (m / p.conv_W) % p.conv_H;
        uint32_t msk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = oy + (t / 3 - 1) * p.conv_dil, ix = ox + (t % 3 - 1) * p.conv_dil;
          msk |= (uint32_t)(iy >= 0 && iy < p.conv_H && ix >= 0 && ix < p.conv_W) << t;
        }
        a_msk[it] = msk;
      }
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int P = it * NT + tid, r = P >> 3, kc = (P & 7) ^ ((r >> 1) & 7);
      b_src[it] = p.B + (long)min(n0 + r, p.N - 1) * p.ldb + k_begin + kc * 8;
    }
  };
  const int conv_cpt = CONV ? p.conv_C / BK : 1;     // K tiles per tap
  auto stage = [&](const bf16_t* (&a_src)[NA], uint32_t (&a_msk)[NA], const bf16_t* (&b_src)[NB], int buf, int kt) {
    char* sa = smem + buf * STAGE;
    char* sb = sa + A_BYTES;
    const int koff = kt * BK;
    if constexpr (CONV) {
      const int t = kt / conv_cpt, cb = kt - t * conv_cpt;                       // wave-uniform
      const long aoff = ((long)(t / 3 - 1) * p.conv_W + (t % 3 - 1)) * p.conv_dil * p.conv_C + cb * BK;
#pragma unroll
      for (int it = 0; it < NA; ++it) {
        const bf16_t* src = ((a_msk[it] >> t) & 1u) ? a_src[it] + aoff : p.zero;
        __builtin_amdgcn_global_load_lds(GLDS_PTR(src), LDS_PTR(sa + (it * NT + wave * 64) * 16), 16, 0, 0);
      }
    } else {
#pragma unroll
      for (int it = 0; it < NA; ++it)
        __builtin_amdgcn_global_load_lds(GLDS_PTR(a_src[it] + koff), LDS_PTR(sa + (it * NT + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < NB; ++it)
      __builtin_amdgcn_global_load_lds(GLDS_PTR(b_src[it] + koff), LDS_PTR(sb + (it * NT + wave * 64) * 16), 16, 0, 0);
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row = (wave tile row) + f*16 + (lane&15); (row>>1)&7 == (lane&15)>>1
  const int frow = lane & 15;
  const int c0 = (((lane >> 4) ^ (frow >> 1)) << 4);  // k-step 0; k-step 1 is c0 ^ 64
  const int a_off = (wm * WM + frow) * 128 + c0;
  const int b_off = (wn * WN + frow) * 128 + c0;

  const bf16_t* a_src[NA];
  const bf16_t* b_src[NB];
  uint32_t a_msk[NA];
  int w = blockIdx.x, m0, n0, buf = 0;
  if (!CONV && p.stagger > 0 && ((blockIdx.x >> 3) & 63) >= 32) {      // second-resident workgroup of its CU: phase offset
    for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
  }
  tile_of(w, m0, n0);
  setup(a_src, a_msk, b_src, m0, n0);
  stage(a_src, a_msk, b_src, 0, 0);
  __syncthreads();
  for (;;) {
    const int w_next = w + gridDim.x;
    const bool has_next = w_next < nt;
    int m0n = 0, n0n = 0;
    const bf16_t* a_nx[NA];
    const bf16_t* b_nx[NB];
    uint32_t m_nx[NA];
    for (int kt = 0; kt < ntk; ++kt) {
      if (kt + 1 < ntk) {
        stage(a_src, a_msk, b_src, buf ^ 1, kt + 1);
      } else if (has_next) {   // last K tile: the first stage of the NEXT output tile streams in under the epilogue
        tile_of(w_next, m0n, n0n);
        setup(a_nx, m_nx, b_nx, m0n, n0n);
        stage(a_nx, m_nx, b_nx, buf ^ 1, 0);
      }
      const char* sa = smem + buf * STAGE;
      const char* sb = sa + A_BYTES;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 af[FM], bfr[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(sa + ((a_off + i * 16 * 128) ^ (ks * 64)));
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[j] = *(const bf16x8*)(sb + ((b_off + j * 16 * 128) ^ (ks * 64)));
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = VLB_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
      if (kt + 1 < ntk) {
        __syncthreads();  // next stage landed (hipcc drains the LDS-DMA queue here) + WAR on the buffer just read
        buf ^= 1;
      }
    }
    // ---- epilogue: lane holds C[m][n..n+3], m = .. + (lane&15), n = .. + 4*(lane>>4) ----
    const int mb = m0 + wm * WM + (lane & 15), nb = n0 + wn * WN + 4 * (lane >> 4);
    // staging image = the stage buffer the last K tile was read from (the other one is receiving the next tile)
    const EpiStage st = {smem + buf * STAGE, m0, n0, wm * WM + (lane & 15), wn * WN + 4 * (lane >> 4), tid, NT, BN};
    if (m0 + BM <= p.M && n0 + BN <= p.N) {
      __syncthreads();   // every wave finished reading the K-loop operands of this buffer
      gemm_epilogue_select<EPI, true, FM, FN>(p, acc, mb, nb, st);
    } else {
      gemm_epilogue_select<EPI, false, FM, FN>(p, acc, mb, nb, st);
    }
    if (!has_next) break;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NA; ++it) { a_src[it] = a_nx[it]; if constexpr (CONV) a_msk[it] = m_nx[it]; }
#pragma unroll
    for (int it = 0; it < NB; ++it) b_src[it] = b_nx[it];
    w = w_next; m0 = m0n; n0 = n0n;
    __syncthreads();   // the prefetched first stage of the next tile has landed; everyone left the old buffers
    buf ^= 1;
  }
}

// ------------------------------------------------------------------------------------
// The same tiles (128x128 / 128x64, BK = 64, every fused epilogue) behind a RING of NS operand stages with counted waits.
//
// The kernel above has two stages and retires the whole LDS-DMA queue at every K tile (hipcc drains it in front of the
// compiler-visible fragment reads), i.e. a prefetch distance of one K tile: fine when many workgroups share a CU and cover each
// other's latency, but the per-GPU batches of a strong-scaling run (M = 3232 rows at 32 samples per GPU) give a launch only
// 150-600 tiles -- one or two per CU -- and those GEMMs ran at the L2 -> LDS latency, 0.7 us per K tile against 0.1 us of MFMA
// work.  Here stage g + NS - 1 is issued while stage g is consumed, the wait is `s_waitcnt vmcnt(stages still allowed in
// flight)`, the fragment reads are inline asm (invisible to the compiler's own waitcnt insertion) released by an explicit
// lgkmcnt wait, and the ring runs across output tiles (persistent workgroups; the next tile's first stages land under the
// epilogue, which stages its bf16 rows through the slot consumed last).  One barrier per K tile, as before.
// ------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void vlb_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    vlb_static_for<I + 1, N>(f);
  }
}

template <int OFF>
__device__ __forceinline__ void ring_lds_read(bf16x8& dst, uint32_t vaddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF));
}

template <int N>
__device__ __forceinline__ void ring_wait_vm() {      // gfx9 s_waitcnt: vmcnt = imm[15:14]:imm[3:0]; expcnt 7 / lgkmcnt 15 = no wait
  __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

template <int BM, int BN, int WGM, int WGN, int EPI, int NS>
__global__ __launch_bounds__(64 * WGM * WGN, 2) void gemm_nt_ring_kernel(const GemmParams p) {
  constexpr int BK = 64;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int NA = BM * 8 / NT, NB = BN * 8 / NT;   // 16-B chunks per thread per stage
  constexpr int NL = NA + NB;
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  static_assert(BM == 128, "the epilogue staging image is 128 rows");
  static_assert((NS - 1) * NL < 64, "vmcnt is a 6-bit counter");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nt = p.ntm * p.ntn;
  const int ntk = p.K / BK;
  if ((int)blockIdx.x >= nt || ntk <= 0) return;
  auto tile_of = [&](int w, int& m0, int& n0) {      // XCD-aware grouped order, as in gemm_nt_bf16_kernel
    const int xcd = w & 7, q = nt >> 3, r = nt & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    const int gm = p.tile_group, per_group = gm * p.ntn, gid = t / per_group, first = gid * gm;
    const int gsz = min(p.ntm - first, gm), rem = t - gid * per_group;
    m0 = (first + rem % gsz) * BM;
    n0 = (rem / gsz) * BN;
  };
  // ---- producer: (tile, K tile) of the next stage to issue; per-thread source pointers advance in place -------------
  const bf16_t* a_src[NA];
  const bf16_t* b_src[NB];
  int w_p = blockIdx.x, kt_p = 0, issued = 0, slot_p = 0;
  auto setup = [&](int w) {
    int m0, n0;
    tile_of(w, m0, n0);
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int P = it * NT + tid, r = P >> 3, kc = (P & 7) ^ ((r >> 1) & 7);
      a_src[it] = p.A + (long)min(m0 + r, p.M - 1) * p.lda + kc * 8;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int P = it * NT + tid, r = P >> 3, kc = (P & 7) ^ ((r >> 1) & 7);
      b_src[it] = p.B + (long)min(n0 + r, p.N - 1) * p.ldb + kc * 8;
    }
  };
  auto produce = [&]() {
    if (w_p >= nt) return;
    char* sa = smem + slot_p * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      __builtin_amdgcn_global_load_lds(GLDS_PTR(a_src[it]), LDS_PTR(sa + (it * NT + wave * 64) * 16), 16, 0, 0);
      a_src[it] += BK;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      __builtin_amdgcn_global_load_lds(GLDS_PTR(b_src[it]), LDS_PTR(sb + (it * NT + wave * 64) * 16), 16, 0, 0);
      b_src[it] += BK;
    }
    ++issued;
    slot_p = (slot_p + 1 == NS) ? 0 : slot_p + 1;
    if (++kt_p == ntk) {
      kt_p = 0;
      w_p += gridDim.x;
      if (w_p < nt) setup(w_p);
    }
  };

  f32x4 acc[FM][FN];
  const int frow = lane & 15;
  const int c0 = (((lane >> 4) ^ (frow >> 1)) << 4);
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const uint32_t a_rel = (uint32_t)((wm * WM + frow) * 128 + c0);             // k-step 1 fragment: chunk index ^ 4 (byte 64)
  const uint32_t b_rel = (uint32_t)(A_BYTES + (wn * WN + frow) * 128 + c0);

  setup(w_p);
#pragma unroll
  for (int s_ = 0; s_ < NS - 1; ++s_) produce();
  int g = 0, slot_c = 0;      // stages consumed so far, ring slot of stage g
  for (int w = blockIdx.x; w < nt; w += gridDim.x) {
    int m0, n0;
    tile_of(w, m0, n0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ntk; ++kt) {
      // stage g has landed once only the loads of the stages issued after it can still be outstanding (in-order retirement;
      // right after an epilogue the youngest operations are its stores, which only makes the wait conservative)
      const int ahead = issued - g - 1;
      if (NS >= 4 && ahead >= 3) ring_wait_vm<3 * NL>();
      else if (NS >= 3 && ahead == 2) ring_wait_vm<2 * NL>();
      else if (ahead == 1) ring_wait_vm<NL>();
      else ring_wait_vm<0>();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();     // every wave's part of stage g is in LDS; everyone is done reading stage g-1
      asm volatile("" ::: "memory");
      produce();                        // stage g + NS - 1 -> the slot stage g - 1 was read from
      const uint32_t so = lds0 + (uint32_t)(slot_c * STAGE);
      const uint32_t va0 = so + a_rel, vb0 = so + b_rel, va1 = so + (a_rel ^ 64u), vb1 = so + (b_rel ^ 64u);
      bf16x8 af[2][FM], bfr[2][FN];
      vlb_static_for<0, FN>([&](auto j_c) { constexpr int j = decltype(j_c)::value; ring_lds_read<j * 2048>(bfr[0][j], vb0); });
      vlb_static_for<0, FM>([&](auto i_c) { constexpr int i = decltype(i_c)::value; ring_lds_read<i * 2048>(af[0][i], va0); });
      vlb_static_for<0, FN>([&](auto j_c) { constexpr int j = decltype(j_c)::value; ring_lds_read<j * 2048>(bfr[1][j], vb1); });
      vlb_static_for<0, FM>([&](auto i_c) { constexpr int i = decltype(i_c)::value; ring_lds_read<i * 2048>(af[1][i], va1); });
      // k-step 0 starts under the LDS latency of the k-step 1 fragments (reads return in order)
      __builtin_amdgcn_s_waitcnt(0xC07F | ((FM + FN) << 8));      // lgkmcnt(FM + FN); vmcnt / expcnt: no wait
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = VLB_MFMA_16x16x32(bfr[0][j], af[0][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(0xC07F);                         // lgkmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = VLB_MFMA_16x16x32(bfr[1][j], af[1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ++g;
      slot_c = (slot_c + 1 == NS) ? 0 : slot_c + 1;
    }
    // ---- epilogue: lane holds C[m][n..n+3]; staging image = the slot consumed last (no stage in flight targets it) ----
    const int last = (slot_c == 0) ? NS - 1 : slot_c - 1;
    const int mb = m0 + wm * WM + (lane & 15), nb = n0 + wn * WN + 4 * (lane >> 4);
    const EpiStage st = {smem + last * STAGE, m0, n0, wm * WM + (lane & 15), wn * WN + 4 * (lane >> 4), tid, NT, BN};
    if (m0 + BM <= p.M && n0 + BN <= p.N) {
      __syncthreads();   // every wave finished reading the K-loop operands of that slot
      gemm_epilogue_select<EPI, true, FM, FN>(p, acc, mb, nb, st);
    } else {
      gemm_epilogue_select<EPI, false, FM, FN>(p, acc, mb, nb, st);
    }
    // one full drain per output tile, where the compiler can see it: with the epilogue's loads / stores still pending at the
    // loop back-edge hipcc protects their registers with an `s_waitcnt vmcnt(0)` INSIDE the K loop, which would empty the ring at
    // every K tile.  (The next tile's first NS - 1 stages were issued before the epilogue and have landed by now.)
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
}


// ------------------------------------------------------------------------------------
// NT GEMM with 256x256 tiles for the plain (bias-only, bf16-out) GEMMs with many output tiles: the tied decoder
// (N = vocabulary) and the QKV projection.
//
// The 128x128 kernel above streams 1/64 B of operands per FLOP through L2 -> LDS and that stream, not the MFMA pipe,
// is its ceiling.  A 256x256 tile halves the bytes per FLOP.  It costs 128 accumulator registers per wave
// (8 waves of 128x64, 2 per SIMD, ONE workgroup per CU), so the operand stream has to hide its latency without a second
// workgroup: a 3-stage ring of BK = 32 slabs (3 x 32 KB), LDS-DMA loads for stage g+2 issued in iteration g and
// retired by a COUNTED s_waitcnt vmcnt -- the queue is never drained in steady state -- plus a bare s_barrier.
// Fragment reads are inline asm (a compiler-visible LDS read that may alias a pending LDS-DMA gets an s_waitcnt
// vmcnt(0) from hipcc, which would drain the ring); their results are released to the MFMAs by an explicit lgkmcnt
// wait.  Counted waits go through __builtin_amdgcn_s_waitcnt so that the compiler's own bookkeeping sees them.
// Operand image: [rows][32 bf16] = 64-B rows, 16-B chunk c of row r at chunk c ^ G[(r>>2)&3], G = {0,3,2,1} (the
// 16-lane service groups of ds_read_b128 then cover 16 distinct bank slots); as everywhere the swizzle is applied to
// the LDS-DMA source address.  The ring runs across output tiles (persistent workgroups): the next tile's first stages
// stream in under the epilogue, which stages bf16 rows through the ring slot consumed last (64-row slabs) and drains
// the queue once per tile (vmcnt also counts stores).
// ------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ void lds_read_b128_asm(bf16x8& dst, uint32_t vaddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF));
}

__global__ __launch_bounds__(512, 2) void gemm_nt_256_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 256, NT = 512, NS = 3;
  constexpr int WM = 128, WN = 64, FM = 8, FN = 4;
  constexpr int NA = BM * 4 / NT, NB = BN * 4 / NT;            // 16-B chunks per thread per stage (4 per 64-B row)
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = p.ntm * p.ntn;
  const int ntk = p.K / 32;
  if ((int)blockIdx.x >= nt) return;

  auto tile_of = [&](int w, int& m0, int& n0) {   // same XCD-aware grouped order as the 2-stage kernel
    const int xcd = w & 7, q = nt >> 3, r = nt & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    const int gm = p.tile_group, per_group = gm * p.ntn, gid = t / per_group, first = gid * gm;
    const int gsz = min(p.ntm - first, gm), rem = t - gid * per_group;
    m0 = (first + rem % gsz) * BM;
    n0 = (rem / gsz) * BN;
  };
  // ---- producer: the (tile, k) of the next stage to issue and its per-thread source pointers (advanced in place)
  const bf16_t* a_src[NA];
  const bf16_t* b_src[NB];
  int w_p = blockIdx.x, kt_p = 0, issued = 0;
  auto setup = [&](int w) {
    int m0, n0;
    tile_of(w, m0, n0);
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int P = it * NT + tid, r = P >> 2, kc = (P & 3) ^ ((4 - ((r >> 2) & 3)) & 3);
      a_src[it] = p.A + (long)min(m0 + r, p.M - 1) * p.lda + kc * 8;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int P = it * NT + tid, r = P >> 2, kc = (P & 3) ^ ((4 - ((r >> 2) & 3)) & 3);
      b_src[it] = p.B + (long)min(n0 + r, p.N - 1) * p.ldb + kc * 8;
    }
  };
  auto produce = [&]() {
    if (w_p >= nt) return;
    char* sa = smem + (issued % NS) * STAGE;
    char* sb = sa + A_BYTES;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      __builtin_amdgcn_global_load_lds(GLDS_PTR(a_src[it]), LDS_PTR(sa + (it * NT + wave * 64) * 16), 16, 0, 0);
      a_src[it] += 32;
    }
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      __builtin_amdgcn_global_load_lds(GLDS_PTR(b_src[it]), LDS_PTR(sb + (it * NT + wave * 64) * 16), 16, 0, 0);
      b_src[it] += 32;
    }
    ++issued;
    if (++kt_p == ntk) {
      kt_p = 0;
      w_p += gridDim.x;
      if (w_p < nt) setup(w_p);
    }
  };

  f32x4 acc[FM][FN];
  const int frow = lane & 15;
  const int pc = ((lane >> 4) ^ ((4 - (frow >> 2)) & 3)) << 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const uint32_t a_off = lds0 + (wm * WM + frow) * 64 + pc;
  const uint32_t b_off = lds0 + A_BYTES + (wn * WN + frow) * 64 + pc;

  setup(w_p);
  produce();
  produce();
  int g = 0;   // stages consumed so far (ring slot = g % NS)
  for (int w = blockIdx.x; w < nt; w += gridDim.x) {
    int m0, n0;
    tile_of(w, m0, n0);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < ntk; ++kt, ++g) {
      // stage g has landed once at most the NA+NB loads of stage g+1 are outstanding (gfx9 s_waitcnt immediate:
      // vmcnt = imm[3:0], expcnt = 7 and lgkmcnt = 15 mean "no wait")
      if (issued - g >= 2) __builtin_amdgcn_s_waitcnt(0x0F70 | (NA + NB));
      else __builtin_amdgcn_s_waitcnt(0x0F70);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();     // every wave's part of stage g is in LDS; everyone is done reading stage g-1
      asm volatile("" ::: "memory");
      produce();                        // stage g+2 -> slot (g-1) % NS
      const uint32_t so = (uint32_t)((g % NS) * STAGE);
      const uint32_t va = so + a_off, vb = so + b_off;
      bf16x8 af[FM], bfr[FN];
      lds_read_b128_asm<0 * 1024>(bfr[0], vb); lds_read_b128_asm<1 * 1024>(bfr[1], vb);
      lds_read_b128_asm<2 * 1024>(bfr[2], vb); lds_read_b128_asm<3 * 1024>(bfr[3], vb);
      lds_read_b128_asm<0 * 1024>(af[0], va); lds_read_b128_asm<1 * 1024>(af[1], va);
      lds_read_b128_asm<2 * 1024>(af[2], va); lds_read_b128_asm<3 * 1024>(af[3], va);
      lds_read_b128_asm<4 * 1024>(af[4], va); lds_read_b128_asm<5 * 1024>(af[5], va);
      lds_read_b128_asm<6 * 1024>(af[6], va); lds_read_b128_asm<7 * 1024>(af[7], va);
      // the first half of the wave tile (A fragments 0-3) starts under the LDS latency of fragments 4-7
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = VLB_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));
#pragma unroll
      for (int i = 4; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = VLB_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- epilogue (+bias, bf16): lane holds C[m][n..n+3], m = .. + (lane&15) + 16 i, n = .. + 4 (lane>>4) + 16 j
    const int row_l = wm * WM + (lane & 15), col_l = wn * WN + 4 * (lane >> 4);
    float bias[FN][4];
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + col_l + j * 16 + r;
        bias[j][r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
      }
    bf16_t* C = (bf16_t*)p.C;
    if (m0 + BM <= p.M && n0 + BN <= p.N) {
      // interior tile: 64-row slabs through the ring slot consumed last ([64][512 B], 16-B chunk XOR (row & 31)), then
      // whole 512-B rows out with 16 B per lane
      char* st = smem + ((g - 1) % NS) * STAGE;
      for (int h = 0; h < 4; ++h) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // (h = 0: every wave's MFMAs have consumed their fragments of that slot)
        asm volatile("" ::: "memory");
        if (wm == (h >> 1)) {
#pragma unroll
          for (int i4 = 0; i4 < 4; ++i4) {
            const int i = (h & 1) * 4 + i4;
            const int row = (row_l + i * 16) & 63;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              const int colb = (col_l + j * 16) * 2;
              *(uint2*)(st + row * 512 + ((((colb >> 4) ^ (row & 31)) << 4) | (colb & 15))) =
                  make_uint2(pack2bf(acc[i][j][0] + bias[j][0], acc[i][j][1] + bias[j][1]),
                             pack2bf(acc[i][j][2] + bias[j][2], acc[i][j][3] + bias[j][3]));
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int it = 0; it < 64 * 32 / NT; ++it) {
          const int q = it * NT + tid, row = q >> 5, ch = q & 31;
          const uint4 v = *(const uint4*)(st + row * 512 + ((ch ^ (row & 31)) << 4));
          *(uint4*)(C + (long)(m0 + h * 64 + row) * p.ldc + n0 + ch * 8) = v;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int m = m0 + row_l + i * 16;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = n0 + col_l + j * 16;
          bf16_t* c = C + (long)m * p.ldc + n;
          if (n + 3 < p.N) {
            *(uint2*)c = make_uint2(pack2bf(acc[i][j][0] + bias[j][0], acc[i][j][1] + bias[j][1]),
                                    pack2bf(acc[i][j][2] + bias[j][2], acc[i][j][3] + bias[j][3]));
          } else {
            for (int r = 0; r < 4 && n + r < p.N; ++r) c[r] = f2bf(acc[i][j][r] + bias[j][r]);
          }
        }
      }
    }
    // one full drain per output tile, visible to the compiler (stores / spill traffic cannot force a vmcnt(0) into the
    // next K loop); the stages prefetched for the next tile have had the whole epilogue to land
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();          // nobody re-fills the staging slot while someone still copies out of it
    asm volatile("" ::: "memory");
  }
}

// ------------------------------------------------------------------------------------
// "TN" GEMM for weight gradients:  C[Mo,No] (fp32) (+)= A[R,Mo]^T · B[R,No]   (reduction over the R ROWS)
//   dW[n_out, k_in] = sum_rows dY[row, n_out] * X[row, k_in]        (autograd's grad_output.t().mm(input))
// Both operands are consumed exactly as the forward pass left them (row-major activations / gradients): no
// transposed copies in HBM.  A stage holds two [64 rows][128 cols] bf16 images (256-B rows) filled by LDS-DMA;
// MFMA fragments need 8 consecutive ROWS of one column per lane, which is what the gfx950 LDS transpose read
// ds_read_b64_tr_b16 delivers: within a 16-lane group, lane L supplies the 8-byte address of (row L>>2, 4 columns
// (L&3)*4..) of a 4x16 block and receives the block's column L (4 rows).  Two such reads (rows +0..3, +4..7) form
// one bf16x8 operand.  Bank conflicts: rows are 256 B = one full bank row, so the 32-B column block index is
// XOR-swizzled with f(row) = (row&3) | ((row>>3)&1)<<2  -> the 8 row segments one instruction pass touches
// (4 rows x 2 lane groups) fall on 8 distinct 32-B bank groups; as with the NT kernel the swizzle is applied to
// the per-lane SOURCE address of the lane-linear LDS-DMA.
// Rows >= R are sourced from a 16-B zero block (LDS-DMA cannot predicate), columns are clamped in-bounds (their
// products land in output rows/cols that the epilogue masks).
// Optional fused column sums of A (bias gradient): accumulated from the A fragments by the tile_n == 0 workgroups.
// ------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) uint4 vlb_zero16[2];

typedef __attribute__((ext_vector_type(4))) short s16x4;

// The transpose reads are INLINE ASM: for the compiler-visible builtin hipcc orders every read behind ALL pending
// LDS-DMA (s_waitcnt vmcnt(0) at the top of the K loop, right after the next stage's loads were issued), which
// serialised "fetch stage k+1" and "compute stage k" -- the TN kernel ran 25 % below the NT kernel for that reason.
// The asm reads are ordered by the workgroup barrier that published the stage; their results are released to the MFMAs
// by explicit lgkmcnt waits (LDS returns in order; a pending scalar load can only make a counted wait conservative).
template <int OFF>
__device__ __forceinline__ void lds_tr_read(s16x4& dst, uint32_t vaddr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF));
}

template <int KS, int IMG_OFF, int N>   // the N fragments of k-step KS of one operand image: rows +0..3 (lo) and +4..7 (hi)
__device__ __forceinline__ void tn_issue(s16x4 (&lo)[N], s16x4 (&hi)[N], const uint32_t (&va)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    lds_tr_read<IMG_OFF + KS * 32 * 256>(lo[i], va[i]);
    lds_tr_read<IMG_OFF + KS * 32 * 256 + 4 * 256>(hi[i], va[i]);
  }
}

template <int CNT, int NA, int NB>   // s_waitcnt lgkmcnt(CNT); the fragments it releases are tied in as operands
__device__ __forceinline__ void tn_wait(s16x4 (&alo)[NA], s16x4 (&ahi)[NA], s16x4 (&blo)[NB], s16x4 (&bhi)[NB]) {
  static_assert(NA == 4 && (NB == 2 || NB == 4), "written out for 4 A and 2|4 B fragments");
  if constexpr (NB == 2) {
    asm volatile("s_waitcnt lgkmcnt(%12)"
                 : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3]),
                   "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1])
                 : "n"(CNT));
  } else {
    asm volatile("s_waitcnt lgkmcnt(%16)"
                 : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3]),
                   "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]), "+v"(bhi[3])
                 : "n"(CNT));
  }
}

__device__ __forceinline__ bf16x8 tn_frag(s16x4 lo, s16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

template <bool COLSUM, int FN>
__device__ __forceinline__ void tn_mfma_step(s16x4 (&alo)[4], s16x4 (&ahi)[4], s16x4 (&blo)[FN], s16x4 (&bhi)[FN], f32x4 (&acc)[4][FN],
                                             float (&csum)[4]) {
  bf16x8 af[4], bfr[FN];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = tn_frag(alo[i], ahi[i]);
#pragma unroll
  for (int j = 0; j < FN; ++j) bfr[j] = tn_frag(blo[j], bhi[j]);
  if (COLSUM) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint4 w = __builtin_bit_cast(uint4, af[i]);
      csum[i] += (bflo(w.x) + bfhi(w.x)) + (bflo(w.y) + bfhi(w.y)) + (bflo(w.z) + bfhi(w.z)) + (bflo(w.w) + bfhi(w.w));
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
      acc[i][j] = VLB_MFMA_16x16x32(bfr[j], af[i], acc[i][j], 0, 0, 0);
}

// one stage = 64 reduction rows = 2 k-steps; va / vb: per-fragment LDS byte addresses inside the stage's A image
// (the B image follows at +16 KiB, folded into the instruction offsets)
template <bool COLSUM, int FN>
__device__ __forceinline__ void tn_compute_stage(const uint32_t (&va)[4], const uint32_t (&vb)[FN], f32x4 (&acc)[4][FN], float (&csum)[4]) {
  constexpr int IMG = 64 * 128 * 2;
  s16x4 alo[2][4], ahi[2][4], blo[2][FN], bhi[2][FN];
  tn_issue<0, 0, 4>(alo[0], ahi[0], va);
  tn_issue<0, IMG, FN>(blo[0], bhi[0], vb);
  tn_issue<1, 0, 4>(alo[1], ahi[1], va);
  tn_issue<1, IMG, FN>(blo[1], bhi[1], vb);
  // k-step 0 is complete once at most k-step 1's 2*(4+FN) reads are outstanding (lgkmcnt is a 4-bit counter: <= 15)
  tn_wait<(2 * (4 + FN) <= 15 ? 2 * (4 + FN) : 0), 4, FN>(alo[0], ahi[0], blo[0], bhi[0]);
  tn_mfma_step<COLSUM, FN>(alo[0], ahi[0], blo[0], bhi[0], acc, csum);
  __builtin_amdgcn_sched_barrier(0);   // keep k-step 0's MFMAs in front of the second wait (they cover k-step 1's LDS latency)
  tn_wait<0, 4, FN>(alo[1], ahi[1], blo[1], bhi[1]);
  tn_mfma_step<COLSUM, FN>(alo[1], ahi[1], blo[1], bhi[1], acc, csum);
  __builtin_amdgcn_sched_barrier(0);   // ... and all MFMAs in front of the stage barrier, whose vmcnt(0) they overlap
}

// OUT: 1 fp32 store (split-K slab, or overwrite) | 3 fp32 accumulate -- one instantiation each, like the NT kernels
// CONV: B is an NHWC activation [R = n*H*W rows, conv_C] and the output columns are the 9*conv_C im2col columns (conv_C % 128
// == 0, so a 128-wide column tile lies inside ONE tap): the weight gradient of a 3x3 convolution without the im2col image.
template <int WGN, int OUT, bool CONV = false>   // 2 x WGN waves: WGN = 4 -> 8 waves of 64x32 (two 8-wave workgroups per CU hide LDS/barrier latency)
__global__ __launch_bounds__(128 * WGN, WGN) void gemm_tn_bf16_kernel(const GemmParams p, float* __restrict__ colsum) {
  constexpr int BM = 128, BN = 128, BR = 64;          // output tile, reduction rows per stage
  constexpr int NT = 128 * WGN, NIT = 1024 / NT;      // threads, 16-B chunks per thread per operand image
  constexpr int FM = 4, FN = 8 / WGN;
  constexpr int IMG = BR * 128 * 2, STAGE = 2 * IMG;  // 16 KiB per operand image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nt = p.ntm * p.ntn;
  // tile order: each XCD owns a contiguous run of tiles (block b runs on XCD b%8 when the grid width is a multiple of 8),
  // walked in groups of `tile_group` tile-rows, column-major inside a group, so that the ~tiles/8 workgroups sharing an
  // L2 cover a near-square patch of the output: with a row-major run a wide output (dW of output.dense: 6 x 24 tiles)
  // had every XCD stream all of X -- the profile showed 3.3x the algorithmic HBM bytes, at 4.8 TB/s.
  int tile_m, tile_n;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nt >> 3, r = nt & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int gm = p.tile_group, per_group = gm * p.ntn, gid = t / per_group, first = gid * gm;
    const int gsz = min(p.ntm - first, gm), rem = t - gid * per_group;
    tile_m = first + rem % gsz;
    tile_n = rem / gsz;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int r_begin = blockIdx.y * p.k_per_split;
  const int r_end = min(p.K, r_begin + p.k_per_split);       // p.K = number of reduction rows R
  const int ntk = (r_end - r_begin + BR - 1) / BR;
  const int nfull = (r_end - r_begin) / BR;                   // stages whose 64 rows all exist

  // staging: chunk P = it*256 + tid -> row = P>>4 (0..63), physical 16-B chunk c = P&15;
  // physical 32-B block c>>1 holds logical block (c>>1) ^ f(row).  Source pointers advance by 64 rows per stage.
  const bf16_t* a_ptr[NIT];
  const bf16_t* b_ptr[NIT];
  int s_row[NIT];
  const int conv_tap = CONV ? n0 / p.conv_C : 0;
  const int conv_dy = (conv_tap / 3 - 1) * p.conv_dil, conv_dx = (conv_tap % 3 - 1) * p.conv_dil;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int P = it * NT + tid, row = P >> 4, c = P & 15;
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    const int lc = ((((c >> 1) ^ f) << 1) | (c & 1)) * 8;     // logical column offset inside the 128-wide tile
    s_row[it] = row;
    a_ptr[it] = p.A + (long)(r_begin + row) * p.lda + min(m0 + lc, (int)p.lda - 8);
    if constexpr (CONV)   // the shifted pixel's channel chunk; validity is decided per stage from the row's (y, x)
      b_ptr[it] = p.B + ((long)(r_begin + row) + (long)conv_dy * p.conv_W + conv_dx) * p.conv_C + (n0 - conv_tap * p.conv_C) + lc;
    else
      b_ptr[it] = p.B + (long)(r_begin + row) * p.ldb + min(n0 + lc, (int)p.ldb - 8);
  }
  const long a_step = 64 * p.lda, b_step = 64 * p.ldb;
  const bf16_t* zero = (const bf16_t*)vlb_zero16;

  auto stage_full = [&](int buf) {   // all 64 rows in range: unconditional LDS-DMA, pointers bumped afterwards
    char* sa = smem + buf * STAGE;
    char* sb = sa + IMG;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      __builtin_amdgcn_global_load_lds(GLDS_PTR(a_ptr[it]), LDS_PTR(sa + (it * NT + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLDS_PTR(b_ptr[it]), LDS_PTR(sb + (it * NT + wave * 64) * 16), 16, 0, 0);
      a_ptr[it] += a_step;
      b_ptr[it] += b_step;
    }
  };
  auto stage_tail = [&](int buf, int kt) {   // last, partial stage: rows >= r_end come from the zero block
    char* sa = smem + buf * STAGE;
    char* sb = sa + IMG;
    const int rbase = r_begin + kt * BR;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const bool ok = rbase + s_row[it] < r_end;
      const bf16_t* ga = ok ? a_ptr[it] : zero;
      const bf16_t* gb = ok ? b_ptr[it] : zero;
      __builtin_amdgcn_global_load_lds(GLDS_PTR(ga), LDS_PTR(sa + (it * NT + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLDS_PTR(gb), LDS_PTR(sb + (it * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  const float inv_w = CONV ? 1.0f / (float)p.conv_W : 0.f, inv_h = CONV ? 1.0f / (float)p.conv_H : 0.f;
  auto stage_conv = [&](int buf, int kt) {   // B rows gathered at (y + dy, x + dx) with zero fill outside the image
    char* sa = smem + buf * STAGE;
    char* sb = sa + IMG;
    const int rbase = r_begin + kt * BR;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int m = rbase + s_row[it];
      const bool ok = m < r_end;
      // m -> (y, x): float reciprocal + one correction step (m < 2^24)
      int q = (int)((float)m * inv_w);
      int x = m - q * p.conv_W;
      if (x < 0) { x += p.conv_W; --q; } else if (x >= p.conv_W) { x -= p.conv_W; ++q; }
      int q2 = (int)((float)q * inv_h);
      int y = q - q2 * p.conv_H;
      if (y < 0) y += p.conv_H; else if (y >= p.conv_H) y -= p.conv_H;
      const bool in = ok && (unsigned)(y + conv_dy) < (unsigned)p.conv_H && (unsigned)(x + conv_dx) < (unsigned)p.conv_W;
      const bf16_t* ga = ok ? a_ptr[it] : zero;
      const bf16_t* gb = in ? b_ptr[it] : zero;
      __builtin_amdgcn_global_load_lds(GLDS_PTR(ga), LDS_PTR(sa + (it * NT + wave * 64) * 16), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLDS_PTR(gb), LDS_PTR(sb + (it * NT + wave * 64) * 16), 16, 0, 0);
      a_ptr[it] += a_step;
      b_ptr[it] += b_step;
    }
  };
  auto stage = [&](int buf, int kt) {
    if constexpr (CONV) {
      stage_conv(buf, kt);
    } else {
      if (kt < nfull) stage_full(buf);
      else stage_tail(buf, kt);
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float csum[FM] = {0.f, 0.f, 0.f, 0.f};
  const bool do_colsum = (colsum != nullptr) && (tile_n == 0) && (wn == 0);   // wn==1 waves hold the same A fragments

  // fragment addressing (see header comment): L = lane&15, g = lane>>4
  const int L = lane & 15, g = lane >> 4;
  const int fl = (L >> 2) | ((g & 1) << 2);
  const int lane_off = (8 * g + (L >> 2)) * 256 + (L & 3) * 8;    // + 32*256*ks, + 4*256 for the high half
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  uint32_t a_fo[FM], b_fo[FN];   // fragment byte offsets inside an operand image: lane part + swizzled 32-B column block
#pragma unroll
  for (int i = 0; i < FM; ++i) a_fo[i] = lds0 + lane_off + (((wm * 4 + i) ^ fl) << 5);
#pragma unroll
  for (int j = 0; j < FN; ++j) b_fo[j] = lds0 + lane_off + (((wn * FN + j) ^ fl) << 5);

  // The K loop is instantiated twice OUTSIDE the column-sum branch: selecting the variant inside the loop makes
  // the 64 accumulator registers live across a branch and hipcc then shuttles them VGPR<->AGPR every iteration
  // (89 v_accvgpr_write per stage, as slow as the MFMAs themselves).
  auto k_loop = [&](auto with_colsum) {
    constexpr bool CS = decltype(with_colsum)::value;
    stage(0, 0);
    __syncthreads();
    for (int kt = 0; kt < ntk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < ntk) stage(cur ^ 1, kt + 1);
      uint32_t va[FM], vb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) va[i] = a_fo[i] + cur * STAGE;
#pragma unroll
      for (int j = 0; j < FN; ++j) vb[j] = b_fo[j] + cur * STAGE;
      tn_compute_stage<CS, FN>(va, vb, acc, csum);
      __syncthreads();
    }
  };
  if (ntk > 0) {
    if (do_colsum) k_loop(std::true_type{});
    else k_loop(std::false_type{});
  }
  if (do_colsum) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float v = csum[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int m = m0 + wm * 64 + i * 16 + L;
      if (g == 0 && m < p.M) atomicAdd(colsum + m, v);
    }
  }
  const int mb = m0 + wm * 64 + (lane & 15), nb = n0 + wn * (16 * FN) + 4 * (lane >> 4);
  const EpiStage st = {smem, m0, n0, wm * 64 + (lane & 15), wn * (16 * FN) + 4 * (lane >> 4), tid, NT, 128};   // fp32 outputs: unused
  if (m0 + BM <= p.M && n0 + BN <= p.N) gemm_epilogue<0, false, false, OUT, true, FM, FN>(p, acc, mb, nb, st);
  else gemm_epilogue<0, false, false, OUT, false, FM, FN>(p, acc, mb, nb, st);
}

// ------------------------------------------------------------------------------------
// bf16 transpose:  out[c][r] = in[r][c]  (out leading dim ldo >= R; pad columns untouched),
// optional fused column sum  colsum[c] += sum_r in[r][c]  (bias gradients).
// 64x64 tiles through an XOR-swizzled LDS image; BOTH global sides move 16 B per lane with 8 lanes
// covering one 128-B row segment.  Element (r,c) lives at row r, 16-B chunk (c>>3)^(r>>3): the gather of
// 8 rows x one column that builds a transposed 16-B chunk is then bank-conflict free.
// Fast path needs 16-B aligned rows on both sides (ldi, ldo multiples of 8); otherwise scalar accesses.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void transpose_tile(const bf16_t* __restrict__ in, long ldi, bf16_t* __restrict__ out, long ldo, int R, int C,
                                               float* __restrict__ colsum, int vec, int r0, int c0) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 64];
  __shared__ float csum[4][64];
  const int tid = threadIdx.x, sub = tid & 7, grp = tid >> 3;  // 8 lanes per 128-B row segment, 32 row groups
  float cs[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) cs[k] = 0.f;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int rl = grp + 32 * it, r = r0 + rl, c = c0 + sub * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < R) {
      if (vec && c + 8 <= C) {
        v = *(const uint4*)(in + (long)r * ldi + c);
      } else {
        bf16_t e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (c + k < C) ? in[(long)r * ldi + c + k] : (bf16_t)0;
        v = make_uint4(e[0] | (uint32_t)e[1] << 16, e[2] | (uint32_t)e[3] << 16, e[4] | (uint32_t)e[5] << 16, e[6] | (uint32_t)e[7] << 16);
      }
    }
    *(uint4*)(tile + rl * 64 + ((sub ^ (rl >> 3)) << 3)) = v;
    if (colsum) {
      cs[0] += bflo(v.x); cs[1] += bfhi(v.x); cs[2] += bflo(v.y); cs[3] += bfhi(v.y);
      cs[4] += bflo(v.z); cs[5] += bfhi(v.z); cs[6] += bflo(v.w); cs[7] += bfhi(v.w);
    }
  }
  if (colsum) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // lanes with equal `sub` hold the same 8 columns: reduce over xor 8,16,32
      cs[k] += __shfl_xor(cs[k], 8, 64);
      cs[k] += __shfl_xor(cs[k], 16, 64);
      cs[k] += __shfl_xor(cs[k], 32, 64);
    }
    if ((tid & 63) < 8) {
#pragma unroll
      for (int k = 0; k < 8; ++k) csum[tid >> 6][sub * 8 + k] = cs[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int cl = grp + 32 * it, c = c0 + cl, rb = sub * 8;  // output row c, rows rb..rb+7 of the tile
    bf16_t e[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = tile[(rb + k) * 64 + (((cl >> 3) ^ sub) << 3) + (cl & 7)];
    if (c < C) {
      const int r = r0 + rb;
      bf16_t* o = out + (long)c * ldo + r;
      if (vec && r + 8 <= R) {
        *(uint4*)o = make_uint4(e[0] | (uint32_t)e[1] << 16, e[2] | (uint32_t)e[3] << 16, e[4] | (uint32_t)e[5] << 16,
                                e[6] | (uint32_t)e[7] << 16);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (r + k < R) o[k] = e[k];
      }
    }
  }
  if (colsum && tid < 64 && c0 + tid < C) atomicAdd(colsum + c0 + tid, csum[0][tid] + csum[1][tid] + csum[2][tid] + csum[3][tid]);
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, long ldi, bf16_t* __restrict__ out,
                                                             long ldo, int R, int C, float* __restrict__ colsum, int vec) {
  transpose_tile(in, ldi, out, ldo, R, C, colsum, vec, blockIdx.y * 64, blockIdx.x * 64);
}

// Many small transposes in ONE launch (the per-step refresh of the ~50 transposed weight copies the dgrad GEMMs read:
// 5 us of launch + ramp each when issued one by one).  desc[m] = {in, ldi, out, ldo, R, C} (device, int64), tile_start[m] =
// first 64x64 tile of matrix m in the flat grid (n+1 entries); a block finds its matrix by binary search.
__global__ __launch_bounds__(256) void transpose_batched_kernel(const long* __restrict__ desc, const int* __restrict__ tile_start, int n) {
  int lo = 0, hi = n;   // tile_start[lo] <= blockIdx.x < tile_start[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int)blockIdx.x >= tile_start[mid]) lo = mid; else hi = mid;
  }
  const long* d = desc + 6 * lo;
  const bf16_t* in = (const bf16_t*)d[0];
  bf16_t* out = (bf16_t*)d[2];
  const long ldi = d[1], ldo = d[3];
  const int R = (int)d[4], C = (int)d[5];
  const int t = blockIdx.x - tile_start[lo], tx = (C + 63) >> 6;
  const int vec = ((ldi % 8) == 0 && (ldo % 8) == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0) ? 1 : 0;
  transpose_tile(in, ldi, out, ldo, R, C, nullptr, vec, (t / tx) * 64, (t % tx) * 64);
}

// ------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------
static int g_nt_stagger = -1;
void vlb_nt_set_stagger(int v) { g_nt_stagger = v; }

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

template <int BM, int BN, int WGM, int WGN, int EPI>
static int launch_gemm_cfg(GemmParams& p, int splits, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN) * 64 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<BM, BN, WGM, WGN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  static const int group = env_int("VLB_GEMM_TILE_GROUP", 4);
  p.ntm = vlb_cdiv(p.M, BM);
  p.ntn = vlb_cdiv(p.N, BN);
  p.tile_group = group < 1 ? 1 : group;
  // persistent grid: the workgroups that are resident at once (2 per CU with 48-64 KB LDS each on 256 CUs),
  // a multiple of 8 so that work item w and block b stay on the same XCD
  static const int resident = env_int("VLB_GEMM_RESIDENT", 512);
  int gx = p.ntm * p.ntn;
  int cap = resident / splits;
  if (cap < 8) cap = 8;
  cap &= ~7;
  if (gx > cap) gx = cap;
  dim3 grid(gx, splits);
  hipLaunchKernelGGL((gemm_nt_bf16_kernel<BM, BN, WGM, WGN, EPI>), grid, dim3(64 * WGM * WGN), smem, stream, p);
  VLB_CHECK_LAUNCH("vlb_gemm_nt_bf16");
  return VLB_OK;
}

// which single-epilogue instantiation serves this call (-1: the generic kernel with the runtime dispatch)
static int epi_class(const GemmParams& p) {
  static const int specialise = env_int("VLB_GEMM_EPI_SPECIALISE", 1);
  if (!specialise || p.out_f32 != 0 || p.res_stats || p.c_f16) return -1;
  if (p.act == 0) return p.res ? (p.drop_thr ? 3 : 4) : (p.drop_thr ? -1 : 0);
  if (p.act == 4) return 1;
  if (p.act == 5) return 2;
  if (p.act == 2) return 5;
  if (p.act == 7) return 6;
  if (p.act == 8) return p.res ? 7 : 8;
  return -1;
}

template <int BM, int BN, int WGM, int WGN>
static int launch_gemm_epi(GemmParams& p, int splits, hipStream_t stream) {
  switch (epi_class(p)) {
    case 0: return launch_gemm_cfg<BM, BN, WGM, WGN, 0>(p, splits, stream);
    case 1: return launch_gemm_cfg<BM, BN, WGM, WGN, 1>(p, splits, stream);
    case 2: return launch_gemm_cfg<BM, BN, WGM, WGN, 2>(p, splits, stream);
    case 3: return launch_gemm_cfg<BM, BN, WGM, WGN, 3>(p, splits, stream);
    case 4: return launch_gemm_cfg<BM, BN, WGM, WGN, 4>(p, splits, stream);
    case 5: return launch_gemm_cfg<BM, BN, WGM, WGN, 5>(p, splits, stream);
    case 6: return launch_gemm_cfg<BM, BN, WGM, WGN, 6>(p, splits, stream);
    case 7: return launch_gemm_cfg<BM, BN, WGM, WGN, 7>(p, splits, stream);
    case 8: return launch_gemm_cfg<BM, BN, WGM, WGN, 8>(p, splits, stream);
    default: return launch_gemm_cfg<BM, BN, WGM, WGN, -1>(p, splits, stream);
  }
}

// ring kernels: 128x128 (8 waves, 4 stages = 128 KiB: one workgroup per CU) and 128x64 (4 waves, 3 stages = 72 KiB: two per CU)
static int g_nt_ring = -1;        // VLB_GEMM_NT_RING: 0 off | 1 auto (default) | 2 force 128x128 | 3 force 128x64
void vlb_nt_set_ring(int v) { g_nt_ring = v; }

template <int BM, int BN, int WGM, int WGN, int EPI, int NS>
static int launch_ring_cfg(GemmParams& p, hipStream_t stream) {
  constexpr int smem = NS * (BM + BN) * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_ring_kernel<BM, BN, WGM, WGN, EPI, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  static const int group = env_int("VLB_GEMM_TILE_GROUP", 4);
  p.ntm = vlb_cdiv(p.M, BM);
  p.ntn = vlb_cdiv(p.N, BN);
  p.tile_group = group < 1 ? 1 : group;
  const int cap = 256 * (163840 / smem);       // resident workgroups (a multiple of 8: work item w and block b share an XCD)
  int gx = p.ntm * p.ntn;
  if (gx > cap) gx = cap;
  hipLaunchKernelGGL((gemm_nt_ring_kernel<BM, BN, WGM, WGN, EPI, NS>), dim3(gx), dim3(64 * WGM * WGN), smem, stream, p);
  VLB_CHECK_LAUNCH("vlb_gemm_nt_bf16(ring)");
  return VLB_OK;
}

template <int BM, int BN, int WGM, int WGN, int NS>
static int launch_ring_epi(GemmParams& p, hipStream_t stream) {
  switch (epi_class(p)) {
    case 0: return launch_ring_cfg<BM, BN, WGM, WGN, 0, NS>(p, stream);
    case 1: return launch_ring_cfg<BM, BN, WGM, WGN, 1, NS>(p, stream);
    case 2: return launch_ring_cfg<BM, BN, WGM, WGN, 2, NS>(p, stream);
    case 3: return launch_ring_cfg<BM, BN, WGM, WGN, 3, NS>(p, stream);
    case 4: return launch_ring_cfg<BM, BN, WGM, WGN, 4, NS>(p, stream);
    case 5: return launch_ring_cfg<BM, BN, WGM, WGN, 5, NS>(p, stream);
    case -1: return launch_ring_cfg<BM, BN, WGM, WGN, -1, NS>(p, stream);
    default: return 1;      // the Bottleneck-tail epilogues stay on the two-stage kernel
  }
}

// > 0: not taken
static int gemm_ring_try(GemmParams& p, int splits, bool want_narrow, hipStream_t stream) {
  if (g_nt_ring < 0) g_nt_ring = env_int("VLB_GEMM_NT_RING", 1);
  if (!g_nt_ring || splits != 1 || p.c_split_stride != 0 || p.k_per_split < p.K) return 1;
  // auto: measured on MI355X (tools/p8_check.py ring) the deeper prefetch only pays where a launch has about one tile per CU and a
  // long K loop -- the N = 768 GEMMs of a 32..64-sample per-GPU batch (M = 3232 / 6464: 70 -> 56 us for ffn2 fwd, 52 -> 42 us for
  // the QKV dgrad at M = 6464); with several workgroups per CU the two-stage kernel's co-resident workgroups already cover the
  // latency and its smaller LDS footprint wins (M >= 12928: 5-25 % slower with the ring)
  if (g_nt_ring == 1 && !(p.N <= 1024 && p.M <= 8192 && p.M >= 1024 && p.K >= 512)) return 1;
  const bool narrow = g_nt_ring == 3 || g_nt_ring == 1;
  (void)want_narrow;
  // (round 5, measured and removed: 128x128 tiles on FOUR waves -- 64x64 per wave, 0.5 fragment reads per MFMA instead of 0.75, 64
  // FLOP per operand byte instead of 43, three stages = 96 KiB, one block per CU; the N = 768 launches of a 32-sample batch are 156
  // such tiles.  us per launch at M = 3232 / 6464, K = 3072: 39.5 / 75.6 against 34.2 / 55.3 for the 128x64 form: one wave per SIMD
  // has nobody to hide its fragment-read latency behind.  gpurun_out/r5g/sk_bench.txt)
  if (narrow) return launch_ring_epi<128, 64, 2, 2, 3>(p, stream);
  return launch_ring_epi<128, 128, 2, 4, 4>(p, stream);
}


template <int BM, int BN>
static int launch_gemm(GemmParams& p, int splits, hipStream_t stream) {
  if (BN == 128) return launch_gemm_epi<BM, 128, 2, 4>(p, splits, stream);   // 8 waves of 64x32: 16 waves/CU hide LDS/barrier latency
  return launch_gemm_epi<BM, BN, 2, 2>(p, splits, stream);
}

static int gemm_nt_impl(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                        const float* bias, int act, const void* aux, long ldaux, void* pre, long ldpre,
                        const void* res, long ldres, const float* res_stats, const float* res_gamma, const float* res_beta, int out_f16,
                        float drop_p, const uint32_t* seed, uint32_t tag, int out_mode, int splitk, hipStream_t stream) {
  if (M <= 0 || N <= 0) return VLB_OK;
  VLB_CHECK_ARG(!res_stats || (res && res_gamma && res_beta && act == 0 && out_mode == 0),
                "vlb_gemm_nt_bf16_ex: the LayerNorm residual needs res (fp16 pre-LN rows), gamma, beta, act 0 and a 16-bit output");
  VLB_CHECK_ARG(!out_f16 || out_mode == 0, "vlb_gemm_nt_bf16_ex: out_f16 applies to the 16-bit output mode");
  VLB_CHECK_ARG(K > 0 && (K % 64) == 0, "vlb_gemm_nt_bf16: K=%d must be a positive multiple of 64", K);
  VLB_CHECK_ARG(A && B && C, "vlb_gemm_nt_bf16: null operand");
  VLB_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0, "vlb_gemm_nt_bf16: lda/ldb must be multiples of 8 elements");
  VLB_CHECK_ARG((ldc % 4) == 0, "vlb_gemm_nt_bf16: ldc must be a multiple of 4");
  VLB_CHECK_ARG(out_mode >= 0 && out_mode <= 3, "vlb_gemm_nt_bf16: bad out_mode %d", out_mode);
  VLB_CHECK_ARG(act >= 0 && act <= 8, "vlb_gemm_nt_bf16: bad act %d", act);
  VLB_CHECK_ARG((act != 3 && act != 5 && act != 8) || aux, "vlb_gemm_nt_bf16: act=3/5/8 needs aux");
  VLB_CHECK_ARG(act != 7 || res, "vlb_gemm_nt_bf16: act=7 (relu after residual) needs res");
  VLB_CHECK_ARG(act == 0 || act >= 7 || (!(drop_p > 0.f) && !res), "vlb_gemm_nt_bf16: an activation cannot be combined with dropout/residual");
  VLB_CHECK_ARG(act < 7 || !(drop_p > 0.f), "vlb_gemm_nt_bf16: act=7/8 cannot be combined with dropout");
  VLB_CHECK_ARG(out_mode == 0 || (act == 0 && !(drop_p > 0.f) && !res), "vlb_gemm_nt_bf16: fp32 outputs take bias only");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_gemm_nt_bf16: dropout needs a device seed pointer");
  VLB_CHECK_ARG((long)M * (long)N < (1L << 32) || !(drop_p > 0.f), "vlb_gemm_nt_bf16: dropout index overflow");
  GemmParams p = {};
  p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb;
  p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.act = act; p.aux = (const bf16_t*)aux; p.ldaux = ldaux; p.pre = (bf16_t*)pre; p.ldpre = ldpre;
  p.res = (const bf16_t*)res; p.ldres = ldres;
  p.res_stats = res_stats; p.res_gamma = res_gamma; p.res_beta = res_beta; p.c_f16 = out_f16 ? 1 : 0;
  p.drop_thr = vlb_drop_thr(drop_p); p.drop_scale = vlb_drop_scale(p.drop_thr); p.seed = seed; p.tag = tag;
  p.C = C; p.ldc = ldc; p.out_f32 = out_mode; p.c_split_stride = 0;
  if (g_nt_stagger < 0) g_nt_stagger = env_int("VLB_GEMM_NT_STAGGER", 0);
  p.stagger = g_nt_stagger;
  int splits = 1;
  const int ktiles = K / 64;
  if (out_mode == 2) {
    splits = splitk > 0 ? splitk : 1;
    if (splitk <= 0) {
      // auto split-K: 2 workgroups are resident per CU (64 KB LDS each) -> 512 slots on 256 CUs.  Pick the
      // smallest split count whose grid fills whole rounds of slots best (quantisation, not block count,
      // is what hurt: 540 blocks = 2 rounds at 53 % vs 504 blocks = 1 round at 98 %).
      const long tiles = (long)vlb_cdiv(M, 128) * vlb_cdiv(N, 128);
      const long slots = 512;
      double best = -1.0;
      for (int sp = 1; sp <= 32 && sp <= ktiles; ++sp) {
        const long blocks = tiles * sp;
        const long rounds = (blocks + slots - 1) / slots;
        const double eff = (double)blocks / (double)(rounds * slots);
        if (eff > best + 0.04) { best = eff; splits = sp; }
      }
    }
    if (splits > ktiles) splits = ktiles;
    if (splits < 1) splits = 1;
  }
  const int per = vlb_cdiv(ktiles, splits);
  splits = vlb_cdiv(ktiles, per);
  p.k_per_split = per * 64;
  // large-tile 8-phase core (gemm_p8.hip): bf16 outputs with the fused epilogues of the training step, enough tiles to give
  // every CU a 256-row tile
  if (splits == 1 && out_mode == 0) {
    const int took = vlb_gemm_p8_try(p, stream);
    if (took != 0) return took < 0 ? took : VLB_OK;
  }
  // 256x256 tiles (half the operand bytes per FLOP) for plain bf16 GEMMs whose B operand does not fit the L2s and that
  // have enough tiles for one workgroup per CU: the tied decoder X . E^T (B = 47 MB word embeddings; 1.00 -> 0.80 ms).
  // With an L2-resident weight matrix and K = 768 (QKV, FFN) the 128x128 kernel is as fast (measured) and keeps two
  // workgroups per CU.  VLB_GEMM_256: 0 off | 1 this rule (default) | n >= 2: every plain GEMM with at least n tiles.
  static const int use256 = env_int("VLB_GEMM_256", 1);
  const long tiles256 = (long)vlb_cdiv(M, 256) * vlb_cdiv(N, 256);
  const bool big_b = (long)N * K * 2 >= (24L << 20) && tiles256 >= 512;
  if (use256 && out_mode == 0 && splits == 1 && act == 0 && !res && !p.drop_thr &&
      (use256 == 1 ? big_b : tiles256 >= use256)) {
    constexpr int smem = 3 * (256 + 256) * 64;
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)gemm_nt_256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      attr_set = true;
    }
    static const int group5 = env_int("VLB_GEMM_256_GROUP", 2);
    p.ntm = vlb_cdiv(M, 256);
    p.ntn = vlb_cdiv(N, 256);
    p.tile_group = group5 < 1 ? 1 : group5;
    int gx = p.ntm * p.ntn;
    if (gx > 256) gx = 256;             // one workgroup per CU
    hipLaunchKernelGGL(gemm_nt_256_kernel, dim3(gx), dim3(512), smem, stream, p);
    VLB_CHECK_LAUNCH("vlb_gemm_nt_bf16(256)");
    return VLB_OK;
  }
  // narrow-N tile when the 128x128 grid would leave most CUs idle
  const long tiles128 = (long)vlb_cdiv(M, 128) * vlb_cdiv(N, 128) * splits;
  const bool narrow = tiles128 < 384 || N <= 64;
  {
    const int took = gemm_ring_try(p, splits, narrow, stream);
    if (took <= 0) return took;
  }
  if (narrow) return launch_gemm<128, 64>(p, splits, stream);
  return launch_gemm<128, 128>(p, splits, stream);
}

extern "C" int vlb_gemm_nt_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                const float* bias, int act, const void* aux, long ldaux, void* pre, long ldpre,
                                const void* res, long ldres, float drop_p, const uint32_t* seed, uint32_t tag,
                                int out_mode, int splitk, hipStream_t stream) {
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, pre, ldpre, res, ldres, nullptr, nullptr, nullptr, 0,
                      drop_p, seed, tag, out_mode, splitk, stream);
}

// The same with (a) the "LayerNorm residual": `res` = the fp16 pre-LayerNorm rows of the sublayer that produced the residual,
// res_stats = its [M][2] (mean, rstd), res_gamma / res_beta = that LayerNorm's parameters; the term added is the LayerNorm output
// re-materialised in fp32 -- (b) out_f16: the 16-bit result is written as IEEE fp16 (the pre-LayerNorm sum this GEMM produces).
// Together they keep the encoder's residual stream out of bf16 (BertSelfOutput / BertOutput, modeling.py:329-333,374-378).
extern "C" int vlb_gemm_nt_bf16_ex(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                   const float* bias, int act, const void* aux, long ldaux, void* pre, long ldpre,
                                   const void* res, long ldres, const float* res_stats, const float* res_gamma, const float* res_beta,
                                   int out_f16, float drop_p, const uint32_t* seed, uint32_t tag, int out_mode, int splitk,
                                   hipStream_t stream) {
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, act, aux, ldaux, pre, ldpre, res, ldres, res_stats, res_gamma, res_beta,
                      out_f16, drop_p, seed, tag, out_mode, splitk, stream);
}

// ------------------------------------------------------------------------------------
// Implicit 3x3 convolution (stride 1, padding = dilation) on an NHWC bf16 activation: the NT GEMM above with the im2col
// gather done by the LDS-DMA address generator -- no [rows, 9C] image in HBM, the activation is read through L2 nine
// times instead.  y[M, O] = epilogue( sum_taps x[pixel + tap] . w[O, tap, :]^T ), w = [O, 9*C] tap-major.
// Epilogues: bias (act 0), bias + ReLU (act 2: the Bottleneck's conv2), x (aux > 0) (act 8: its data gradient with the
// mirrored-tap operand of vlb_conv_weight_prepare).
// ------------------------------------------------------------------------------------
template <int BN, int WGN, int EPI>
static int launch_conv_cfg(GemmParams& p, hipStream_t stream) {
  constexpr int smem = 2 * (128 + BN) * 64 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_bf16_kernel<128, BN, 2, WGN, EPI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  static const int group = env_int("VLB_GEMM_TILE_GROUP", 4);
  p.ntm = vlb_cdiv(p.M, 128);
  p.ntn = vlb_cdiv(p.N, BN);
  p.tile_group = group < 1 ? 1 : group;
  int gx = p.ntm * p.ntn;
  if (gx > 512) gx = 512;
  hipLaunchKernelGGL((gemm_nt_bf16_kernel<128, BN, 2, WGN, EPI, true>), dim3(gx, 1), dim3(64 * 2 * WGN), smem, stream, p);
  VLB_CHECK_LAUNCH("vlb_conv3x3_nhwc_bf16");
  return VLB_OK;
}

extern "C" int vlb_conv3x3_nhwc_bf16(const void* x, int N, int H, int W, int C, int dil, const void* w, long ldw, void* y, long ldy,
                                     int O, const float* bias, int act, const void* aux, long ldaux, const void* zero16,
                                     hipStream_t stream) {
  if (N <= 0 || O <= 0) return VLB_OK;
  VLB_CHECK_ARG(x && w && y && zero16, "vlb_conv3x3_nhwc_bf16: null argument");
  VLB_CHECK_ARG(C > 0 && (C % 64) == 0, "vlb_conv3x3_nhwc_bf16: C=%d must be a multiple of 64", C);
  VLB_CHECK_ARG(H > 0 && W > 0 && dil >= 1, "vlb_conv3x3_nhwc_bf16: bad geometry");
  VLB_CHECK_ARG((ldw % 8) == 0 && ldw >= 9L * C && (ldy % 4) == 0, "vlb_conv3x3_nhwc_bf16: bad leading dimensions");
  VLB_CHECK_ARG(act == 0 || act == 2 || (act == 8 && aux), "vlb_conv3x3_nhwc_bf16: act must be 0, 2 or 8 (with aux)");
  VLB_CHECK_ARG((long)N * H * W < (1L << 31), "vlb_conv3x3_nhwc_bf16: too many rows");
  GemmParams p = {};
  p.A = (const bf16_t*)x; p.lda = C; p.B = (const bf16_t*)w; p.ldb = ldw;
  p.M = N * H * W; p.N = O; p.K = 9 * C; p.k_per_split = 9 * C;
  p.bias = bias; p.act = act; p.aux = (const bf16_t*)aux; p.ldaux = ldaux; p.pre = nullptr; p.ldpre = 0; p.res = nullptr; p.ldres = 0;
  p.drop_thr = 0; p.drop_scale = 1.f; p.seed = nullptr; p.tag = 0;
  p.C = y; p.ldc = ldy; p.out_f32 = 0; p.c_split_stride = 0;
  p.conv_C = C; p.conv_H = H; p.conv_W = W; p.conv_dil = dil; p.zero = (const bf16_t*)zero16;
  const long tiles128 = (long)vlb_cdiv(p.M, 128) * vlb_cdiv(O, 128);
  const bool narrow = tiles128 < 384 || O <= 64;
  if (act == 0) return narrow ? launch_conv_cfg<64, 2, 0>(p, stream) : launch_conv_cfg<128, 4, 0>(p, stream);
  if (act == 2) return narrow ? launch_conv_cfg<64, 2, 5>(p, stream) : launch_conv_cfg<128, 4, 5>(p, stream);
  return narrow ? launch_conv_cfg<64, 2, 8>(p, stream) : launch_conv_cfg<128, 4, 8>(p, stream);
}

// ------------------------------------------------------------------------------------
// Weight-gradient GEMM:  C[M,N] (fp32) += A[M,K] B[N,K]^T  with the reduction dimension K = (padded) row count
// of the activation, i.e. few output tiles and a very long K.  Split-K WITHOUT atomics: every K slice writes its
// fp32 partial tile to a workspace slab with plain 16-B stores, a streaming reduce kernel adds the slabs into C.
// (fp32 atomics cost ~75 us per 8M adds on this chip -- more than the MFMA work of the slices they merged.)
// The split count comes from a small cost model: rounds of 512 resident workgroups x K tiles per slice + slab
// traffic; splits == 1 accumulates straight into C (each tile owned by one workgroup, no atomics either).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int splits,
                                                            float* __restrict__ C, long ldc, int M, int N, int ldw,
                                                            bf16_t* __restrict__ Cb, long ldcb, int accumulate,
                                                            const float* __restrict__ rowscale = nullptr) {
  const int n4 = ldw >> 2;
  const long total = (long)M * n4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    const float* src = slabs + (long)m * ldw + n;
    float4 a = *(const float4*)src;
    for (int sp = 1; sp < splits; ++sp) {
      const float4 b = *(const float4*)(src + sp * slab_stride);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (rowscale) {   // folded-BatchNorm weight gradient: dW_master = scale[o] * dW_folded (vision path)
      const float sc = rowscale[m];
      a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
    }
    if (Cb) {   // bf16 result, overwritten (split-K dgrad)
      bf16_t* c = Cb + (long)m * ldcb + n;
      if (n + 3 < N) {
        *(uint2*)c = make_uint2(pack2bf(a.x, a.y), pack2bf(a.z, a.w));
      } else {
        const float av[4] = {a.x, a.y, a.z, a.w};
        for (int r = 0; r < 4 && n + r < N; ++r) c[r] = f2bf(av[r]);
      }
      continue;
    }
    float* c = C + (long)m * ldc + n;
    if (n + 3 < N) {
      if (accumulate) {
        const float4 o = *(const float4*)c;
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      *(float4*)c = a;
    } else {
      const float av[4] = {a.x, a.y, a.z, a.w};
      for (int r = 0; r < 4 && n + r < N; ++r) c[r] = accumulate ? c[r] + av[r] : av[r];
    }
  }
}

static int wgrad_pick_splits(int M, int N, int K, long workspace_floats) {
  const int ktiles = K / 64;
  const long tiles = (long)vlb_cdiv(M, 128) * vlb_cdiv(N, 128);
  const long ldw = (N + 3) / 4 * 4;
  const double t_k = 1.15, t_fix = 4.0, bw = 4.0e6;  // us per K tile (2 workgroups/CU), us per tile, bytes/us
  int best = 1;
  double best_t = 1e30;
  for (int sp = 1; sp <= 32 && sp <= ktiles; ++sp) {
    if (sp > 1 && (long)sp * M * ldw > workspace_floats) break;
    const long rounds = (tiles * sp + 511) / 512;
    const double t = rounds * (vlb_cdiv(ktiles, sp) * t_k + t_fix) + (sp > 1 ? 2.0 * sp * M * ldw * 4.0 / bw : 0.0);
    if (t < best_t * 0.97) { best_t = t; best = sp; }
  }
  return best;
}

extern "C" long vlb_wgrad_workspace_floats(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int sp = wgrad_pick_splits(M, N, K, 1L << 40);
  if (K >= 256 && (K % 128) == 0) {     // the large-tile core may want more slices
    const int sp8 = vlb_tn8_pick_splits(M, N, K);
    if (sp8 > sp) sp = sp8;
  }
  return sp > 1 ? (long)sp * M * ((N + 3) / 4 * 4) : 0;
}

extern "C" int vlb_wgrad_nt_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                                 float* workspace, long workspace_floats, hipStream_t stream) {
  if (M <= 0 || N <= 0) return VLB_OK;
  VLB_CHECK_ARG(K > 0 && (K % 64) == 0, "vlb_wgrad_nt_bf16: K=%d must be a positive multiple of 64", K);
  VLB_CHECK_ARG(A && B && C, "vlb_wgrad_nt_bf16: null operand");
  VLB_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0, "vlb_wgrad_nt_bf16: bad leading dimensions");
  const int splits = wgrad_pick_splits(M, N, K, workspace ? workspace_floats : 0);
  const int ktiles = K / 64;
  const int per = vlb_cdiv(ktiles, splits);
  const int nsp = vlb_cdiv(ktiles, per);
  const long ldw = (N + 3) / 4 * 4;
  GemmParams p = {};
  p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb;
  p.M = M; p.N = N; p.K = K; p.k_per_split = per * 64;
  p.bias = nullptr; p.act = 0; p.aux = nullptr; p.ldaux = 0; p.pre = nullptr; p.ldpre = 0; p.res = nullptr; p.ldres = 0;
  p.drop_thr = 0; p.drop_scale = 1.f; p.seed = nullptr; p.tag = 0;
  if (nsp == 1) {
    p.C = C; p.ldc = ldc; p.out_f32 = 3; p.c_split_stride = 0;
  } else {
    p.C = workspace; p.ldc = ldw; p.out_f32 = 1; p.c_split_stride = (long)M * ldw;
  }
  int rc = launch_gemm<128, 128>(p, nsp, stream);
  if (rc) return rc;
  if (nsp > 1) {
    long blocks = ((long)M * (ldw / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, workspace, (long)M * ldw, nsp, C, ldc, M, N,
                       (int)ldw, (bf16_t*)nullptr, 0L, 1);
    VLB_CHECK_LAUNCH("vlb_wgrad_nt_bf16(reduce)");
  }
  return VLB_OK;
}

// bf16 C[M,N] = A[M,K] B[N,K]^T for FEW output tiles and a very long K (the tied-decoder dgrad d_h = dlogits . E with
// K = vocabulary, at small per-GPU batch: 96 tiles for 256 CUs): same slab split-K as the weight gradients, the
// reduce kernel converts to bf16.  One K pass (splits == 1 or no workspace) falls through to the plain GEMM.
extern "C" int vlb_gemm_nt_bf16_splitk(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                                       float* workspace, long workspace_floats, hipStream_t stream) {
  if (M <= 0 || N <= 0) return VLB_OK;
  VLB_CHECK_ARG(K > 0 && (K % 64) == 0, "vlb_gemm_nt_bf16_splitk: K=%d must be a positive multiple of 64", K);
  VLB_CHECK_ARG(A && B && C, "vlb_gemm_nt_bf16_splitk: null operand");
  VLB_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0, "vlb_gemm_nt_bf16_splitk: bad leading dimensions");
  const int splits = wgrad_pick_splits(M, N, K, workspace ? workspace_floats : 0);
  const int ktiles = K / 64;
  const int per = vlb_cdiv(ktiles, splits);
  const int nsp = vlb_cdiv(ktiles, per);
  if (nsp <= 1)
    return vlb_gemm_nt_bf16(A, lda, B, ldb, C, ldc, M, N, K, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0.f, nullptr, 0, 0, 0, stream);
  const long ldw = (N + 3) / 4 * 4;
  GemmParams p = {};
  p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb;
  p.M = M; p.N = N; p.K = K; p.k_per_split = per * 64;
  p.bias = nullptr; p.act = 0; p.aux = nullptr; p.ldaux = 0; p.pre = nullptr; p.ldpre = 0; p.res = nullptr; p.ldres = 0;
  p.drop_thr = 0; p.drop_scale = 1.f; p.seed = nullptr; p.tag = 0;
  p.C = workspace; p.ldc = ldw; p.out_f32 = 1; p.c_split_stride = (long)M * ldw;
  int rc = launch_gemm<128, 128>(p, nsp, stream);
  if (rc) return rc;
  long blocks = ((long)M * (ldw / 4) + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, workspace, (long)M * ldw, nsp, (float*)nullptr, 0L, M, N,
                     (int)ldw, (bf16_t*)C, ldc, 0);
  VLB_CHECK_LAUNCH("vlb_gemm_nt_bf16_splitk(reduce)");
  return VLB_OK;
}

struct SplitkReduceGroup {
  const float* slabs[4];
  long slab_stride[4];
  float* C[4];
  long ldc[4];
  int splits[4], M[4], N[4], ldw[4];
};

// splitk_reduce_kernel (fp32 result) for up to four outputs in one launch: blockIdx.y selects the member
__global__ __launch_bounds__(256) void splitk_reduce_group_kernel(const SplitkReduceGroup g, int accumulate) {
  const int e = blockIdx.y;
  const float* slabs = g.slabs[e];
  const long slab_stride = g.slab_stride[e], ldc = g.ldc[e];
  float* C = g.C[e];
  const int splits = g.splits[e], M = g.M[e], N = g.N[e], ldw = g.ldw[e];
  const int n4 = ldw >> 2;
  const long total = (long)M * n4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i % n4) * 4;
    const float* src = slabs + (long)m * ldw + n;
    float4 a = *(const float4*)src;
    for (int sp = 1; sp < splits; ++sp) {
      const float4 b = *(const float4*)(src + sp * slab_stride);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float* c = C + (long)m * ldc + n;
    if (n + 3 < N) {
      if (accumulate) {
        const float4 o = *(const float4*)c;
        a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
      }
      vlb_store_nt((float4*)c, a);          // weight gradients are next read by the optimizer: keep them out of the caches
    } else {
      const float av[4] = {a.x, a.y, a.z, a.w};
      for (int r = 0; r < 4 && n + r < N; ++r) c[r] = accumulate ? c[r] + av[r] : av[r];
    }
  }
}

// dW[Mo,No] (fp32) (+)= A[R,Mo]^T B[R,No]; optional colsum[Mo] += column sums of A (bias gradient).
// accumulate == 0 overwrites dW (first micro-batch of an optimizer step: no zero fill and no read-modify-write).
static int wgrad_tn_impl(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int R, int Mo, int No, float* colsum,
                         const float* rowscale, float* workspace, long workspace_floats, int accumulate, hipStream_t stream) {
  if (Mo <= 0 || No <= 0 || R <= 0) return VLB_OK;
  VLB_CHECK_ARG(A && B && C, "vlb_wgrad_tn_bf16: null operand");
  VLB_CHECK_ARG((lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0 && lda >= 8 && ldb >= 8, "vlb_wgrad_tn_bf16: bad leading dimensions");
  VLB_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "vlb_wgrad_tn_bf16: operands must be 16-byte aligned");
  {   // large-tile core (gemm_tn8.hip) when the reduction is a whole number of 128-row units and there is enough work for 256x256 tiles
    GemmParams q = {};
    q.A = (const bf16_t*)A; q.lda = lda; q.B = (const bf16_t*)B; q.ldb = ldb;
    q.M = Mo; q.N = No; q.K = R;
    const int used = vlb_gemm_tn8_try(q, C, ldc, colsum, workspace, workspace_floats, accumulate, rowscale != nullptr, stream);
    if (used < 0) return used;
    if (used > 0) {
      if (used > 1 || rowscale) {
        const long ldw8 = (No + 3) / 4 * 4;
        long blocks = ((long)Mo * (ldw8 / 4) + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, workspace, (long)Mo * ldw8, used, C, ldc, Mo, No,
                           (int)ldw8, (bf16_t*)nullptr, 0L, accumulate, rowscale);
        VLB_CHECK_LAUNCH("vlb_wgrad_tn_bf16(tn8 reduce)");
      }
      return VLB_OK;
    }
  }
  const int Rp = vlb_cdiv(R, 64) * 64;
  const int splits = wgrad_pick_splits(Mo, No, Rp, workspace ? workspace_floats : 0);
  const int ktiles = Rp / 64;
  const int per = vlb_cdiv(ktiles, splits);
  const int nsp = vlb_cdiv(ktiles, per);
  const long ldw = (No + 3) / 4 * 4;
  const bool slab = nsp > 1 || rowscale != nullptr;      // a row scale is applied by the reduce kernel: always go through a slab
  VLB_CHECK_ARG(!rowscale || (workspace && workspace_floats >= (long)nsp * Mo * ldw), "vlb_wgrad_tn: rowscale needs a workspace of %ld floats",
                (long)nsp * Mo * ldw);
  GemmParams p = {};
  p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb;
  p.M = Mo; p.N = No; p.K = R; p.k_per_split = per * 64;
  p.bias = nullptr; p.act = 0; p.aux = nullptr; p.ldaux = 0; p.pre = nullptr; p.ldpre = 0; p.res = nullptr; p.ldres = 0;
  p.drop_thr = 0; p.drop_scale = 1.f; p.seed = nullptr; p.tag = 0;
  if (!slab) {
    p.C = C; p.ldc = ldc; p.out_f32 = accumulate ? 3 : 1; p.c_split_stride = 0;
  } else {
    p.C = workspace; p.ldc = ldw; p.out_f32 = 1; p.c_split_stride = (long)Mo * ldw;
  }
  p.ntm = vlb_cdiv(Mo, 128); p.ntn = vlb_cdiv(No, 128);
  {   // near-square per-XCD patches for wide outputs; tall outputs (decoder: 239 x 6 tiles) already share their A panel row-wise
    static const int tn_group = env_int("VLB_GEMM_TN_GROUP", -1);
    int gm = 1;
    if (tn_group > 0) gm = tn_group;
    else if (2 * p.ntn >= p.ntm) {
      const double per_xcd = (double)p.ntm * p.ntn / 8.0;
      gm = (int)(sqrt(per_xcd) + 0.5);
    }
    if (gm > p.ntm) gm = p.ntm;
    if (gm < 1) gm = 1;
    p.tile_group = gm;
  }
  constexpr int smem = 2 * 2 * 64 * 128 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<2, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<4, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  static const int waves8 = env_int("VLB_GEMM_TN_WAVES8", 1);
  const dim3 grid(p.ntm * p.ntn, nsp);
  if (waves8) {
    if (p.out_f32 == 3) hipLaunchKernelGGL((gemm_tn_bf16_kernel<4, 3>), grid, dim3(512), smem, stream, p, colsum);
    else hipLaunchKernelGGL((gemm_tn_bf16_kernel<4, 1>), grid, dim3(512), smem, stream, p, colsum);
  } else {
    if (p.out_f32 == 3) hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 3>), grid, dim3(256), smem, stream, p, colsum);
    else hipLaunchKernelGGL((gemm_tn_bf16_kernel<2, 1>), grid, dim3(256), smem, stream, p, colsum);
  }
  VLB_CHECK_LAUNCH("vlb_wgrad_tn_bf16");
  if (slab) {
    long blocks = ((long)Mo * (ldw / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, workspace, (long)Mo * ldw, nsp, C, ldc, Mo, No,
                       (int)ldw, (bf16_t*)nullptr, 0L, accumulate, rowscale);
    VLB_CHECK_LAUNCH("vlb_wgrad_tn_bf16(reduce)");
  }
  return VLB_OK;
}

extern "C" int vlb_wgrad_tn_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int R, int Mo, int No,
                                 float* colsum, float* workspace, long workspace_floats, int accumulate, hipStream_t stream) {
  return wgrad_tn_impl(A, lda, B, ldb, C, ldc, R, Mo, No, colsum, nullptr, workspace, workspace_floats, accumulate, stream);
}

// Up to 4 weight gradients over the SAME rows in one launch (the four Linear layers of a transformer block: output.dense,
// intermediate.dense, attention.output.dense, fused QKV): one item list of tiles x K slices on the large-tile core, so the slab
// split-K needs 2 slices instead of 7-28 per gradient (4x less fp32 slab traffic) and 1 launch replaces 4.  Falls back to n
// single calls when the group is outside what the grouped kernel covers.  Arrays are HOST arrays of n entries.
extern "C" int vlb_wgrad_tn_group_bf16(int n, const void* const* A, const long* lda, const void* const* B, const long* ldb,
                                       float* const* C, const long* ldc, int R, const int* Mo, const int* No, float* const* colsum,
                                       float* workspace, long workspace_floats, int accumulate, hipStream_t stream) {
  VLB_CHECK_ARG(n >= 1 && n <= 4 && A && lda && B && ldb && C && ldc && Mo && No, "vlb_wgrad_tn_group_bf16: bad arguments (1 <= n <= 4)");
  for (int i = 0; i < n; ++i) VLB_CHECK_ARG(A[i] && B[i] && C[i], "vlb_wgrad_tn_group_bf16: null operand %d", i);
  int slices[4];
  long ws_off[4];
  const int took = vlb_gemm_tn8_group(n, A, lda, B, ldb, C, ldc, R, Mo, No, colsum, workspace, workspace_floats, accumulate, slices, ws_off,
                                      stream);
  if (took < 0) return took;
  if (took == 0) {
    for (int i = 0; i < n; ++i) {
      const int rc = wgrad_tn_impl(A[i], lda[i], B[i], ldb[i], C[i], ldc[i], R, Mo[i], No[i], colsum ? colsum[i] : nullptr, nullptr,
                                   workspace, workspace_floats, accumulate, stream);
      if (rc) return rc;
    }
    return VLB_OK;
  }
  // the members' slab reduces as ONE launch (blockIdx.y = member): 4 small launches per encoder layer were 48 of the step's launches
  SplitkReduceGroup rg;
  int nr = 0;
  long max_blocks = 1;
  for (int i = 0; i < n; ++i) {
    if (slices[i] <= 1) continue;
    const long ldw = (No[i] + 3) / 4 * 4;
    long blocks = ((long)Mo[i] * (ldw / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks > max_blocks) max_blocks = blocks;
    rg.slabs[nr] = workspace + ws_off[i]; rg.slab_stride[nr] = (long)Mo[i] * ldw; rg.splits[nr] = slices[i];
    rg.C[nr] = C[i]; rg.ldc[nr] = ldc[i]; rg.M[nr] = Mo[i]; rg.N[nr] = No[i]; rg.ldw[nr] = (int)ldw;
    ++nr;
  }
  if (nr) {
    hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((int)max_blocks, nr), dim3(256), 0, stream, rg, accumulate);
    VLB_CHECK_LAUNCH("vlb_wgrad_tn_group_bf16(reduce)");
  }
  return VLB_OK;
}

// same product with every output ROW m multiplied by rowscale[m] before it is written / accumulated (the gradient of a weight
// whose frozen-BatchNorm scale was folded into the bf16 operand); needs workspace >= splits * Mo * round4(No) floats.
extern "C" int vlb_wgrad_tn_rowscale_bf16(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int R, int Mo, int No,
                                          const float* rowscale, float* workspace, long workspace_floats, int accumulate,
                                          hipStream_t stream) {
  VLB_CHECK_ARG(rowscale, "vlb_wgrad_tn_rowscale_bf16: null rowscale");
  return wgrad_tn_impl(A, lda, B, ldb, C, ldc, R, Mo, No, nullptr, rowscale, workspace, workspace_floats, accumulate, stream);
}

// dW[O, 9C] (fp32) (+)= dy[R, O]^T . im2col(x)[R, 9C] for a 3x3 / stride 1 / padding = dilation convolution on the NHWC
// activation x [R = N*H*W, C], C % 128 == 0 -- the TN kernel gathers the shifted pixels itself (no im2col image).
extern "C" int vlb_conv3x3_wgrad_tn_bf16(const void* dy, long lddy, const void* x, int N, int H, int W, int C, int dil, float* dW,
                                         long lddw, int O, const float* rowscale, float* workspace, long workspace_floats, int accumulate,
                                         hipStream_t stream) {
  if (N <= 0 || O <= 0) return VLB_OK;
  VLB_CHECK_ARG(dy && x && dW, "vlb_conv3x3_wgrad_tn_bf16: null operand");
  VLB_CHECK_ARG(C > 0 && (C % 128) == 0, "vlb_conv3x3_wgrad_tn_bf16: C=%d must be a multiple of 128", C);
  VLB_CHECK_ARG((lddy % 8) == 0 && lddy >= O && (lddw % 4) == 0 && lddw >= 9L * C, "vlb_conv3x3_wgrad_tn_bf16: bad leading dimensions");
  VLB_CHECK_ARG(H > 0 && W > 0 && dil >= 1 && (long)N * H * W < (1L << 24), "vlb_conv3x3_wgrad_tn_bf16: bad geometry (rows must be < 2^24)");
  const int R = N * H * W, Mo = O, No = 9 * C;
  const int Rp = vlb_cdiv(R, 64) * 64;
  const int splits = wgrad_pick_splits(Mo, No, Rp, workspace ? workspace_floats : 0);
  const int ktiles = Rp / 64;
  const int per = vlb_cdiv(ktiles, splits);
  const int nsp = vlb_cdiv(ktiles, per);
  const long ldw = No;
  const bool slab = nsp > 1 || rowscale != nullptr;
  VLB_CHECK_ARG(!rowscale || (workspace && workspace_floats >= (long)nsp * Mo * ldw), "vlb_conv3x3_wgrad_tn_bf16: rowscale needs a workspace of %ld floats",
                (long)nsp * Mo * ldw);
  GemmParams p = {};
  p.A = (const bf16_t*)dy; p.lda = lddy; p.B = (const bf16_t*)x; p.ldb = C;
  p.M = Mo; p.N = No; p.K = R; p.k_per_split = per * 64;
  p.bias = nullptr; p.act = 0; p.aux = nullptr; p.ldaux = 0; p.pre = nullptr; p.ldpre = 0; p.res = nullptr; p.ldres = 0;
  p.drop_thr = 0; p.drop_scale = 1.f; p.seed = nullptr; p.tag = 0;
  p.conv_C = C; p.conv_H = H; p.conv_W = W; p.conv_dil = dil; p.zero = nullptr;
  if (!slab) {
    p.C = dW; p.ldc = lddw; p.out_f32 = accumulate ? 3 : 1; p.c_split_stride = 0;
  } else {
    p.C = workspace; p.ldc = ldw; p.out_f32 = 1; p.c_split_stride = (long)Mo * ldw;
  }
  p.ntm = vlb_cdiv(Mo, 128); p.ntn = No / 128;
  {
    const double per_xcd = (double)p.ntm * p.ntn / 8.0;
    int gm = (2 * p.ntn >= p.ntm) ? (int)(sqrt(per_xcd) + 0.5) : 1;
    if (gm > p.ntm) gm = p.ntm;
    if (gm < 1) gm = 1;
    p.tile_group = gm;
  }
  constexpr int smem = 2 * 2 * 64 * 128 * 2;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<4, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    (void)hipFuncSetAttribute((const void*)gemm_tn_bf16_kernel<4, 3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    attr_set = true;
  }
  const dim3 grid(p.ntm * p.ntn, nsp);
  if (p.out_f32 == 3) hipLaunchKernelGGL((gemm_tn_bf16_kernel<4, 3, true>), grid, dim3(512), smem, stream, p, (float*)nullptr);
  else hipLaunchKernelGGL((gemm_tn_bf16_kernel<4, 1, true>), grid, dim3(512), smem, stream, p, (float*)nullptr);
  VLB_CHECK_LAUNCH("vlb_conv3x3_wgrad_tn_bf16");
  if (slab) {
    long blocks = ((long)Mo * (ldw / 4) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)blocks), dim3(256), 0, stream, workspace, (long)Mo * ldw, nsp, dW, lddw, Mo, No,
                       (int)ldw, (bf16_t*)nullptr, 0L, accumulate, rowscale);
    VLB_CHECK_LAUNCH("vlb_conv3x3_wgrad_tn_bf16(reduce)");
  }
  return VLB_OK;
}

extern "C" int vlb_transpose_bf16(const void* in, long ldi, void* out, long ldo, int R, int C, float* colsum,
                                  hipStream_t stream) {
  if (R <= 0 || C <= 0) return VLB_OK;
  VLB_CHECK_ARG(in && out && ldo >= R && ldi >= C, "vlb_transpose_bf16: bad arguments");
  dim3 grid(vlb_cdiv(C, 64), vlb_cdiv(R, 64));
  const int vec = ((ldi % 8) == 0 && (ldo % 8) == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0) ? 1 : 0;
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, R, C,
                     colsum, vec);
  VLB_CHECK_LAUNCH("vlb_transpose_bf16");
  return VLB_OK;
}

extern "C" int vlb_transpose_batched_bf16(const int64_t* desc, const int32_t* tile_start, int n, int total_tiles, hipStream_t stream) {
  if (n <= 0 || total_tiles <= 0) return VLB_OK;
  VLB_CHECK_ARG(desc && tile_start, "vlb_transpose_batched_bf16: null descriptor table");
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(total_tiles), dim3(256), 0, stream, (const long*)desc, (const int*)tile_start, n);
  VLB_CHECK_LAUNCH("vlb_transpose_batched_bf16");
  return VLB_OK;
}
