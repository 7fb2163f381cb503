// bf16 "NT" GEMM for gfx950 (CDNA4), large-tile core:  C[M,N] = epilogue(A[M,K] . B[N,K]^T)   (fp32 accumulate, bf16 out)
//
// The 128x128 kernels of gemm.hip stream 1/64 B of operands per FLOP through L2 -> LDS and park half their wave cycles on the
// s_waitcnt / barrier that drains the LDS-DMA queue at every K tile (profiles/r01_gemm_pmc.txt: MFMA pipe 30 % busy).  This
// kernel is the structure the CDNA4 guide measures at 1.3-1.5 PFLOP/s (cdna_hip_programming.md §5, "8-phase" schedule):
//
//   * ONE 8-wave workgroup per CU, 256 (or 320) x 256 output tile, BK = 64: 1/128 B of operands per FLOP.  Waves are 2 (M) x 4 (N);
//     a wave owns rows {half h} x [wm*16*FMH, +16*FMH) and columns {half h} x [wn*32, +32) of BOTH tile halves, so an operand
//     half-image (AH rows x 64 k, 128-B rows) is consumed by all waves in ONE phase and can be re-staged right after it.
//   * LDS = two K-tile buffers of four half-images [A0 | A1 | B0 | B1] (2 x 64 KiB) + a 32 KiB epilogue staging slab.  Operands
//     arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`, one SRD per operand, k offset in an SGPR) ONE HALF-IMAGE PER PHASE, two
//     K tiles ahead of their use, and are retired by a COUNTED `s_waitcnt vmcnt(4)` once per K tile -- the queue is never
//     drained inside the stream, which runs across output tiles (persistent workgroups: the next tile's operands land under
//     the epilogue of the current one).
//   * A K tile is 4 phases = the 4 quadrants (A-half x B-half) of the wave tile, 16 MFMAs each, in the order (0,0) (0,1) (1,1)
//     (1,0) so that a phase reads ONE new operand sub-tile.  Every phase is {LOAD: fragment reads + one half-image of DMA} |
//     barrier | {COMPUTE: 16 x v_mfma_f32_16x16x32_bf16 under s_setprio 1} | barrier, and the wm = 1 waves run ONE segment
//     behind the wm = 0 waves (they take one extra barrier at the start): on every SIMD one wave is in its MFMA segment
//     while its partner reads LDS / issues DMA -- "matrix beside memory", enforced, not statistical.
//   * Rules that make the counted waits sound (guide, "Read a staged buffer one phase AFTER the wait that retires it"):
//     the vmcnt wait sits in the LOAD segment of phase 4, in front of that segment's barrier, and the buffer is first read in
//     phase 1 of the next K tile (two barriers later for the lagging wave group); a half-image is re-staged no earlier than two
//     phases after its last read.  Fragment reads are inline asm (a compiler-visible LDS read that may alias a pending LDS-DMA
//     gets `s_waitcnt vmcnt(0)` from hipcc), released to the MFMAs by an explicit lgkmcnt(0) + sched_barrier.
//   * Epilogue in TWO passes over LDS-staged fp32 (the single-pass fused epilogues of the 128-register kernels spilled): the
//     raw accumulators go to the staging slab 32 rows at a time (ds_write_b128, XOR-swizzled), then every thread owns 8
//     CONSECUTIVE columns of a row: bias / GELU (+ GELU' saved) / x aux / dropout (counter RNG) / + residual / ReLU run on
//     whole 16-B groups with coalesced 16-B global accesses and one rounding to bf16.
//   * Tile-count quantisation: a 256-row tile gives 303 tiles for N = 768 at M = 25856 (2 rounds of 256 CUs at 59 %); the
//     FMH = 5 instantiation (320-row tiles, 160 accumulator registers) gives 243 (one round at 95 %).  The launcher picks the
//     tile height from a cost model over {256, 320}.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "gemm_params.h"

#ifndef VLB_P8_LN_SB
#define VLB_P8_LN_SB 2       // side rows (+ their LayerNorm statistics) requested at a time by the slab drain of the 320-row LayerNorm-residual forms
#endif
#ifndef VLB_P8_SIDE_DEPTH
#define VLB_P8_SIDE_DEPTH 1  // units of lead of the aux loads in the wave-private drain (EPI 2 / 10)
#endif
#ifndef VLB_P8_WDRAIN
#define VLB_P8_WDRAIN 1      // wave-private epilogue drain (p8_drain_w); 0: the shared-slab drain with workgroup barriers (p8_drain)
#endif

namespace {

template <int OFF>
__device__ __forceinline__ void p8_lds_read(bf16x8& dst, uint32_t vaddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF));
}

template <int OFF>
__device__ __forceinline__ void p8_lds_write_f4(uint32_t vaddr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(vaddr), "v"(v), "n"(OFF) : "memory");
}

template <int OFF>
__device__ __forceinline__ void p8_lds_read_f4(f32x4& dst, uint32_t vaddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF));
}

template <int OFF>   // two 16-B reads of the staging slab (8 consecutive fp32 of one row), waited for in the same statement
__device__ __forceinline__ void p8_stage_read(f32x4& x0, f32x4& x1, uint32_t a0, uint32_t a1) {
  asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(x0), "=&v"(x1) : "v"(a0), "v"(a1), "n"(OFF) : "memory");
}

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N) -- `#pragma unroll` is a request the optimiser may decline
// (it did, for the 20-round epilogues), and a run-time index into the accumulator array sends the whole array to scratch memory
template <int I, int N, typename F>
__device__ __forceinline__ void p8_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    p8_static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ void p8_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// 8 consecutive output columns of one row: the whole fused epilogue on fp32 values, one rounding to bf16.
// EPI: 0 bias | 1 bias + GELU, GELU'(x) -> pre | 2 x aux | 3 bias + dropout + residual | 4 bias + residual | 5 bias + ReLU |
//      6 bias + dropout + LayerNorm-residual | 7 bias + LayerNorm-residual | 8 relu(bias + residual) | 10 acc where aux > 0  (residual = LN output re-materialised in fp32 from the
//      fp16 pre-LN rows in `side`, the row's (mean, rstd) in `ms` and gamma / beta of the thread's 8 columns in g8 / be8)
template <int EPI>
__device__ __forceinline__ void p8_epilogue8(const GemmParams& p, float (&v)[8], const float (&b8)[8], int m, int n, uint32_t seed,
                                             const uint4& side, const float2 ms = make_float2(0.f, 0.f), const float* g8 = nullptr,
                                             const float* be8 = nullptr) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += b8[e];      // (zeros when there is no bias)
  if (EPI == 1) {
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; e += 2) {      // packed fp32: two elements per VALU issue
      vlb_f2 g2, d2;
      gelu_both2((vlb_f2){v[e], v[e + 1]}, g2, d2);
      v[e] = g2.x; v[e + 1] = g2.y;
      d[e] = d2.x; d[e + 1] = d2.y;
    }
    if (p.pre) {
      const uint4 dv = make_uint4(pack2bf(d[0], d[1]), pack2bf(d[2], d[3]), pack2bf(d[4], d[5]), pack2bf(d[6], d[7]));
      vlb_store_nt((uint4*)(p.pre + (long)m * p.ldpre + n), dv);      // GELU' is next read in the backward pass
    }
  } else if (EPI == 2) {
    v[0] *= bflo(side.x); v[1] *= bfhi(side.x); v[2] *= bflo(side.y); v[3] *= bfhi(side.y);
    v[4] *= bflo(side.z); v[5] *= bfhi(side.z); v[6] *= bflo(side.w); v[7] *= bfhi(side.w);
  } else if (EPI == 5) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (EPI == 3 || EPI == 6) {
    const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
    if ((idx & 1u) == 0) {   // element pairs share one 32-bit hash (vlb_common.h)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t h = vlb_rng_pair(seed, p.tag, (idx >> 1) + q);
        v[2 * q] = ((h & 0xffffu) >= p.drop_thr) ? v[2 * q] * p.drop_scale : 0.f;
        v[2 * q + 1] = ((h >> 16) >= p.drop_thr) ? v[2 * q + 1] * p.drop_scale : 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = vlb_keep(seed, p.tag, idx + e, p.drop_thr) ? v[e] * p.drop_scale : 0.f;
    }
  }
  if (EPI == 3 || EPI == 4 || EPI == 8) {
    v[0] += bflo(side.x); v[1] += bfhi(side.x); v[2] += bflo(side.y); v[3] += bfhi(side.y);
    v[4] += bflo(side.z); v[5] += bfhi(side.z); v[6] += bflo(side.w); v[7] += bfhi(side.w);
  }
  if (EPI == 8) {            // Bottleneck tail (common/backbone/resnet/resnet.py:112-116): relu(conv + shift + residual)
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (EPI == 10) {           // ReLU backward on a gradient: acc where the saved activation aux > 0
    const float a[8] = {bflo(side.x), bfhi(side.x), bflo(side.y), bfhi(side.y), bflo(side.z), bfhi(side.z), bflo(side.w), bfhi(side.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f;
  }
  if (EPI == 6 || EPI == 7) {
    const float z[8] = {hlo(side.x), hhi(side.x), hlo(side.y), hhi(side.y), hlo(side.z), hhi(side.z), hlo(side.w), hhi(side.w)};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += fmaf((z[e] - ms.x) * ms.y, g8[e], be8[e]);
  }
}

// scalar tail of the same (last, partial 8-column group of a row when N % 8 != 0)
template <int EPI>
__device__ __forceinline__ float p8_epilogue1(const GemmParams& p, float v, int m, int n, uint32_t seed) {
  if (p.bias) v += p.bias[n];
  if (EPI == 1) {
    float g, d;
    gelu_both(v, g, d);
    if (p.pre) p.pre[(long)m * p.ldpre + n] = f2bf(d);
    v = g;
  } else if (EPI == 2) {
    v *= bf2f(p.aux[(long)m * p.ldaux + n]);
  } else if (EPI == 5) {
    v = fmaxf(v, 0.f);
  }
  if (EPI == 3 || EPI == 6) v = vlb_keep(seed, p.tag, (uint32_t)m * (uint32_t)p.N + (uint32_t)n, p.drop_thr) ? v * p.drop_scale : 0.f;
  if (EPI == 3 || EPI == 4 || EPI == 8) v += bf2f(p.res[(long)m * p.ldres + n]);
  if (EPI == 8) v = fmaxf(v, 0.f);
  if (EPI == 10) v = bf2f(p.aux[(long)m * p.ldaux + n]) > 0.f ? v : 0.f;
  if (EPI == 6 || EPI == 7)
    v += fmaf((h2f(p.res[(long)m * p.ldres + n]) - p.res_stats[2 * (long)m]) * p.res_stats[2 * (long)m + 1], p.res_gamma[n], p.res_beta[n]);
  return v;
}

// Drain one output tile: raw fp32 accumulators -> LDS slab -> fused epilogue on 8-column groups -> bf16 rows.
// KF = accumulator row fragments per wave per round: the slab holds 32*KF rows of 256 fp32 (both wave groups write KF fragments
// each); KF = 0: a 16-row slab, the two wave groups alternate (320-row tiles in mid-stream: 16 KiB of LDS are left beside the
// operand ring).  The 128-row form (KF = 4) borrows the operand ring itself and is used for the LAST tile of a workgroup, when no
// LDS-DMA is in flight any more -- every launch with <= 256 tiles (the N = 768 GEMMs with 320-row tiles) drains this way.
// Latency discipline: the bias of a thread's 8 columns is loaded once per tile; the residual / aux 16-B groups of the NEXT round
// (KF <= 1) or of the whole round (KF = 4: 8 independent loads per thread) are requested before the slab is touched, so the
// L2 round trip overlaps the slab writes, the barrier and the previous round's math instead of sitting in every round.
template <int FMH, int EPI, int KF>
__device__ __forceinline__ void p8_drain(const GemmParams& p, f32x4 (&acc)[2 * FMH][4], uint32_t slab, int m0, int n0, int wm, int wn,
                                         int lane, int tid, uint32_t seed) {
  constexpr int AH = 32 * FMH, NF = 2 * FMH;
  constexpr bool ALT = (KF == 0);
  constexpr bool LNRES = (EPI == 6 || EPI == 7);
  constexpr int FPR = ALT ? 1 : KF;
  constexpr int ROUNDS = ALT ? 2 * NF : (NF + FPR - 1) / FPR;
  constexpr int PASSES = ALT ? 1 : 2 * FPR;
  constexpr bool SIDE = (EPI == 2 || EPI == 3 || EPI == 4 || EPI == 6 || EPI == 7 || EPI == 8 || EPI == 10);
  constexpr bool PRE_NEXT = SIDE && (PASSES <= 2);      // request the next round's side data one round ahead
  // side rows requested at a time (register budget: 160 accumulators + gamma / beta / bias vectors leave room for 2-4 of them)
  constexpr int SB = PRE_NEXT ? PASSES : (FMH == 5 ? (LNRES ? VLB_P8_LN_SB : 4) : PASSES);      // (320-row tile + LayerNorm residual: 2 rows + their statistics)
  // Every lane-dependent constant below is derived from an OPAQUE copy of the thread index: the optimiser would otherwise hoist
  // these loop-invariant address computations above the K loop of the persistent tile loop, where their live ranges cost the
  // main loop the registers it needs (the 320-row instantiation spilled a DMA offset and drained the queue to reload it).
  asm volatile("" : "+v"(tid));
  lane = tid & 63;
  const int frow = lane & 15;
  const int wrow = (ALT ? 0 : wm * 16) + frow;
  uint32_t wr[4];
#pragma unroll
  for (int cj = 0; cj < 4; ++cj) {
#if VLB_P8_WDRAIN
    const int c = wn * 16 + (cj >> 1) * 8 + (cj & 1) * 4 + (lane >> 4);      // a wave's two 32-column halves are neighbours (see p8_drain_w)
#else
    const int c = (cj >> 1) * 32 + wn * 8 + (cj & 1) * 4 + (lane >> 4);
#endif
    wr[cj] = slab + (uint32_t)(wrow * 1024 + ((c ^ (frow & 7)) << 4));
  }
  const int q = tid & 31, rho = tid >> 5;
  const uint32_t rd0 = slab + (uint32_t)(rho * 1024 + (((2 * q) ^ (rho & 7)) << 4));
  const uint32_t rd1 = slab + (uint32_t)(rho * 1024 + (((2 * q + 1) ^ (rho & 7)) << 4));
  const int n = n0 + q * 8;
  const bool full8 = n + 8 <= p.N;
  bf16_t* const C = (bf16_t*)p.C;
  float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.bias && full8) {
    const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
    b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
  }
  // gamma / beta of the thread's 8 columns: resident for the whole tile, except in the instantiation with the least register
  // headroom (320-row tile + dropout hash), which re-reads them per row group from L1
  constexpr bool HOIST_GB = LNRES;
  float g8[8], be8[8];
  if constexpr (LNRES) {
#pragma unroll
    for (int e = 0; e < 8; ++e) g8[e] = be8[e] = 0.f;
    if (HOIST_GB && full8) {
      const float4 a0 = *(const float4*)(p.res_gamma + n), a1 = *(const float4*)(p.res_gamma + n + 4);
      const float4 c0 = *(const float4*)(p.res_beta + n), c1 = *(const float4*)(p.res_beta + n + 4);
      g8[0] = a0.x; g8[1] = a0.y; g8[2] = a0.z; g8[3] = a0.w; g8[4] = a1.x; g8[5] = a1.y; g8[6] = a1.z; g8[7] = a1.w;
      be8[0] = c0.x; be8[1] = c0.y; be8[2] = c0.z; be8[3] = c0.w; be8[4] = c1.x; be8[5] = c1.y; be8[6] = c1.z; be8[7] = c1.w;
    }
  }
  const bool f16out = p.c_f16 != 0;
  // tile row held by slab rows [pp*16, +16) in round r
  auto row_of = [&](int r, int pp) {
    const int f = ALT ? 0 : (pp >> 1), g = ALT ? (r & 1) : (pp & 1);
    const int R = ALT ? (r >> 1) : (r * FPR + f);
    const int ha = R / FMH, i = R - ha * FMH;
    return m0 + ha * AH + g * FMH * 16 + i * 16 + rho;
  };
  auto frag_valid = [&](int r, int pp) { return ALT ? true : (r * FPR + (pp >> 1) < NF); };
  auto load_side = [&](int r, int pp) {
    uint4 sd = make_uint4(0, 0, 0, 0);
    if constexpr (SIDE) {
      const int m = row_of(r, pp);
      if (frag_valid(r, pp) && m < p.M && full8) {
        if (EPI == 2 || EPI == 10) sd = vlb_load_nt((const uint4*)(p.aux + (long)m * p.ldaux + n));      // read exactly once
        else sd = *(const uint4*)(p.res + (long)m * p.ldres + n);
      }
    }
    return sd;
  };
  // the row's LayerNorm statistics (mean, rstd) travel with its side row: loaded at the point of use (round 3) they were a dependent
  // L2 round trip in EVERY pass -- measured in-kernel (tools/p8_phase_probe.py): 24 us of epilogue per 320 x 256 tile for the dropout +
  // LayerNorm-residual form against 13.5 us for bias + residual through the same slab
  auto load_ms = [&](int r, int pp) {
    float2 ms = make_float2(0.f, 0.f);
    if constexpr (LNRES) {
      const int m = min(max(row_of(r, pp), 0), p.M - 1);      // unconditional load from a clamped row: no branch, no drained queue
      ms = *(const float2*)(p.res_stats + 2 * (long)m);
    }
    return ms;
  };
  uint4 side[2][SB];
  float2 mss[2][SB];
  if constexpr (PRE_NEXT) {
#pragma unroll
    for (int pp = 0; pp < PASSES; ++pp) {
      side[0][pp] = load_side(0, pp);
      mss[0][pp] = load_ms(0, pp);
    }
  }
  p8_static_for<0, ROUNDS>([&](auto r_c) {
    constexpr int r = decltype(r_c)::value;
    if constexpr (SIDE && !PRE_NEXT) {      // first batch of this round: in flight under the slab writes and the barrier
#pragma unroll
      for (int pp = 0; pp < SB; ++pp) {
        side[0][pp] = load_side(r, pp);
        mss[0][pp] = load_ms(r, pp);
      }
    }
    if (!ALT || wm == (r & 1)) {
      p8_static_for<0, FPR>([&](auto f_c) {
        constexpr int f = decltype(f_c)::value;
        constexpr int R = ALT ? (r >> 1) : (r * FPR + f);
        if constexpr (R < NF) {
          const uint32_t o = (uint32_t)(f * 32 * 1024);
          p8_lds_write_f4<0>(wr[0] + o, acc[R][0]);
          p8_lds_write_f4<0>(wr[1] + o, acc[R][1]);
          p8_lds_write_f4<0>(wr[2] + o, acc[R][2]);
          p8_lds_write_f4<0>(wr[3] + o, acc[R][3]);
        }
      });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    p8_barrier();
    if constexpr (PRE_NEXT) {
      if (r + 1 < ROUNDS) {
#pragma unroll
        for (int pp = 0; pp < PASSES; ++pp) {
          side[(r + 1) & 1][pp] = load_side(r + 1, pp);
          mss[(r + 1) & 1][pp] = load_ms(r + 1, pp);
        }
      }
    }
#pragma unroll
    for (int pp = 0; pp < PASSES; ++pp) {
      if constexpr (SIDE && !PRE_NEXT) {
        if (pp > 0 && pp % SB == 0) {       // next batch of side rows
#pragma unroll
          for (int k = 0; k < SB; ++k) {
            side[0][k] = load_side(r, pp + k);
            mss[0][k] = load_ms(r, pp + k);
          }
        }
      }
      if (!frag_valid(r, pp)) continue;
      const int m = row_of(r, pp);
      f32x4 x0, x1;
      p8_stage_read<0>(x0, x1, rd0 + (uint32_t)(pp * 16 * 1024), rd1 + (uint32_t)(pp * 16 * 1024));
      if (m < p.M && n < p.N) {
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (full8) {
          const uint4 sd = SIDE ? (PRE_NEXT ? side[r & 1][pp] : side[0][pp % SB]) : make_uint4(0, 0, 0, 0);
          if constexpr (LNRES) {
            const float2 ms = PRE_NEXT ? mss[r & 1][pp] : mss[0][pp % SB];
            if constexpr (!HOIST_GB) {
              const float4 a0 = *(const float4*)(p.res_gamma + n), a1 = *(const float4*)(p.res_gamma + n + 4);
              const float4 c0 = *(const float4*)(p.res_beta + n), c1 = *(const float4*)(p.res_beta + n + 4);
              g8[0] = a0.x; g8[1] = a0.y; g8[2] = a0.z; g8[3] = a0.w; g8[4] = a1.x; g8[5] = a1.y; g8[6] = a1.z; g8[7] = a1.w;
              be8[0] = c0.x; be8[1] = c0.y; be8[2] = c0.z; be8[3] = c0.w; be8[4] = c1.x; be8[5] = c1.y; be8[6] = c1.z; be8[7] = c1.w;
            }
            p8_epilogue8<EPI>(p, v, b8, m, n, seed, sd, ms, g8, be8);
          } else {
            p8_epilogue8<EPI>(p, v, b8, m, n, seed, sd);
          }
          const uint4 o4 = make_uint4(pack2o(v[0], v[1], f16out), pack2o(v[2], v[3], f16out), pack2o(v[4], v[5], f16out), pack2o(v[6], v[7], f16out));
          if (p.ablate != 2) *(uint4*)(C + (long)m * p.ldc + n) = o4;
          else if (o4.x == 0x12345678u && o4.y == 0x9abcdef0u) C[0] = 1;      // (timing ablation: keeps the math alive)
        } else {
          for (int e = 0; e < 8 && n + e < p.N; ++e) {
            const float o = p8_epilogue1<EPI>(p, v[e], m, n + e, seed);
            C[(long)m * p.ldc + n + e] = f16out ? f2h(o) : f2bf(o);
          }
        }
      }
    }
    if (r + 1 < ROUNDS) p8_barrier();      // the slab is free for the next round
  });
  asm volatile("" ::"v"(b8[0]), "v"(b8[4]));      // the bias loads are consumed on every path (see the end of p8_drain_w)
}

// Drain WITHOUT workgroup barriers (default; -DVLB_P8_WDRAIN=0 selects the shared-slab drain above): every wave transposes its own
// accumulators through a PRIVATE 2-KiB region of the slab, one unit = (fragment row R, row half hf) = 8 rows x 64 columns of fp32 at a
// time -- the lanes that hold those rows in the MFMA layout (lane: row lane & 15, 4 consecutive columns per fragment) write their four
// fragments (ds_write_b128), then every lane reads 8 consecutive columns of one row back (lane = (row lane >> 3, columns (lane & 7) * 8)),
// runs the fused epilogue on them and stores 16 B: 8 lanes per row = whole 128-B lines for C and for the side tensors.  LDS operations
// of one wave execute in issue order, so the region needs no second buffer: the writes of unit u + 1 are issued right after the
// reads of unit u have returned and land under the math of unit u.  Measured on the FFN shape (tools/epi_probe.py): the shared
// slab's 16 barriers per tile cost 7 of the 8 us of a tile's "bias only" epilogue.  The B half-images are staged with the rows of
// a wave's two halves ADJACENT (tile column = wn * 64 + h * 32 + ...), so that a wave owns 64 consecutive columns of C.
// EDGE = false: the tile lies inside C (every row < M, every 8-column group whole): straight-line code without per-lane conditions, so
// that the side loads issued one unit ahead are retired by COUNTED vmcnt waits (a load or store under a lane condition makes hipcc
// drain the queue -- including the previous unit's store -- in front of every use).
template <int FMH, int EPI, bool EDGE>
__device__ __forceinline__ void p8_drain_w(const GemmParams& p, f32x4 (&acc)[2 * FMH][4], uint32_t slab, int m0, int n0, int wm, int wn,
                                           int tid, uint32_t seed) {
  constexpr int AH = 32 * FMH, NF = 2 * FMH, NU = 2 * NF;
  constexpr bool LNRES = (EPI == 6 || EPI == 7);
  constexpr bool SIDE = (EPI == 2 || EPI == 3 || EPI == 4 || EPI == 6 || EPI == 7 || EPI == 8 || EPI == 10);
  asm volatile("" : "+v"(tid));      // (see p8_drain: keeps the address arithmetic below out of the K loop's live ranges)
  const int lane = tid & 63, frow = lane & 15;
  const uint32_t mine = slab + (uint32_t)((tid >> 6) * 2048);
  uint32_t wr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wr[j] = mine + (uint32_t)((frow & 7) * 256 + (((j * 4 + (lane >> 4)) ^ (frow & 7)) << 4));
  const bool upper = (frow >> 3) != 0;
  const int rr = lane >> 3, cq = lane & 7;
  const uint32_t rd0 = mine + (uint32_t)(rr * 256 + (((2 * cq) ^ rr) << 4));
  const uint32_t rd1 = mine + (uint32_t)(rr * 256 + (((2 * cq + 1) ^ rr) << 4));
  const bool f16out = p.c_f16 != 0;
  const int n = n0 + wn * 64 + cq * 8;
  const bool full8 = EDGE ? (n + 8 <= p.N) : true;
  const int mw = m0 + wm * FMH * 16 + rr;           // + ha * AH + i * 16 + hf * 8
  auto uoff = [](int u) { return ((u >> 1) / FMH) * AH + ((u >> 1) % FMH) * 16 + (u & 1) * 8; };      // tile row of unit u, lane row 0
  // per-lane element offsets of unit 0; a unit adds a wave-uniform multiple of the row stride
  bf16_t* const Cl = (bf16_t*)p.C + (long)mw * p.ldc + n;
  const bf16_t* const Sl = !SIDE ? nullptr : ((EPI == 2 || EPI == 10) ? p.aux + (long)mw * p.ldaux + n : p.res + (long)mw * p.ldres + n);
  const long lds_ = (EPI == 2 || EPI == 10) ? p.ldaux : p.ldres;
  const float* const Ml = LNRES ? p.res_stats + 2 * (long)mw : nullptr;
  // bias / gamma / beta of the lane's 8 columns: resident for the whole tile
  float b8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, g8[8], be8[8];
  auto consts = [&]() {
    if (p.bias && full8) {
      const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
      b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
    }
    if constexpr (LNRES) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g8[e] = be8[e] = 0.f;
      if (full8) {
        const float4 a0 = *(const float4*)(p.res_gamma + n), a1 = *(const float4*)(p.res_gamma + n + 4);
        const float4 c0 = *(const float4*)(p.res_beta + n), c1 = *(const float4*)(p.res_beta + n + 4);
        g8[0] = a0.x; g8[1] = a0.y; g8[2] = a0.z; g8[3] = a0.w; g8[4] = a1.x; g8[5] = a1.y; g8[6] = a1.z; g8[7] = a1.w;
        be8[0] = c0.x; be8[1] = c0.y; be8[2] = c0.z; be8[3] = c0.w; be8[4] = c1.x; be8[5] = c1.y; be8[6] = c1.z; be8[7] = c1.w;
      }
    }
  };
  auto load_side = [&](int u, uint4& sd, float2& ms) {
    if constexpr (SIDE) {
      if (!EDGE || (mw + uoff(u) < p.M && full8)) {
        const uint4* src = (const uint4*)(Sl + uoff(u) * lds_);
        if (EPI == 2 || EPI == 10) sd = vlb_load_nt(src);      // read exactly once
        else sd = *src;
        if constexpr (LNRES) ms = *(const float2*)(Ml + 2 * uoff(u));
      }
    }
  };
  auto put = [&](auto u_c) {      // the lanes holding rows [hf * 8, +8) of fragment row R write their 64 columns
    constexpr int u = decltype(u_c)::value, R = u >> 1, hf = u & 1;
    if (upper == (hf != 0)) {
      p8_lds_write_f4<0>(wr[0], acc[R][0]);
      p8_lds_write_f4<0>(wr[1], acc[R][1]);
      p8_lds_write_f4<0>(wr[2], acc[R][2]);
      p8_lds_write_f4<0>(wr[3], acc[R][3]);
    }
  };
  auto finish = [&](int u, const f32x4& x0, const f32x4& x1, const uint4& sd, const float2& ms) {
    const int m = mw + uoff(u);
    if (!EDGE || (m < p.M && n < p.N)) {
      float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      if (full8) {
        if constexpr (LNRES) p8_epilogue8<EPI>(p, v, b8, m, n, seed, sd, ms, g8, be8);
        else p8_epilogue8<EPI>(p, v, b8, m, n, seed, sd);
        uint4 o4;
        if (f16out) {
          asm volatile("" ::: "memory");      // a real (wave-uniform) branch: if-converted, BOTH roundings ran in every unit (+40 VALU instructions)
          o4 = make_uint4(pack2h(v[0], v[1]), pack2h(v[2], v[3]), pack2h(v[4], v[5]), pack2h(v[6], v[7]));
        } else {
          o4 = make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
        }
        // (timing ablation 2: every unit overwrites the lane's first 16 B -- the store stays, its HBM traffic goes; no branch here:
        // a store under a condition would make the counted waits conservative)
        *(uint4*)(Cl + (p.ablate != 2 ? uoff(u) * (long)p.ldc : 0L)) = o4;
      } else {
        for (int e = 0; e < 8 && n + e < p.N; ++e) {
          const float o = p8_epilogue1<EPI>(p, v[e], m, n + e, seed);
          Cl[uoff(u) * (long)p.ldc + e] = f16out ? f2h(o) : f2bf(o);
        }
      }
    }
  };
  // side rows are requested DEPTH units ahead of their use.  The aux tensor of the x-aux / ReLU-mask epilogues (GELU' of the FFN2 data
  // gradient: 159 MB per launch, written in the forward pass, i.e. read from HBM, 16 B per lane per unit) is the one side tensor that
  // is as large as the output: one unit of lead keeps 8 KiB per CU in flight, which caps the read stream by latency, not bandwidth
  constexpr int DEPTH = (EPI == 2 || EPI == 10) ? VLB_P8_SIDE_DEPTH : 1, RING = DEPTH + 1;
  uint4 side[RING];
  float2 mst[RING];
#pragma unroll
  for (int d = 0; d < RING; ++d) {
    side[d] = make_uint4(0, 0, 0, 0);
    mst[d] = make_float2(0.f, 0.f);
  }
  consts();
  p8_static_for<0, (DEPTH < NU ? DEPTH : NU)>([&](auto d_c) {
    constexpr int d = decltype(d_c)::value;
    load_side(d, side[d % RING], mst[d % RING]);
  });
  put(std::integral_constant<int, 0>{});
  p8_static_for<0, NU>([&](auto u_c) {
    constexpr int u = decltype(u_c)::value;
    f32x4 x0, x1;
    p8_stage_read<0>(x0, x1, rd0, rd1);      // (waits for this wave's LDS queue: the writes of unit u and these reads)
    if constexpr (u + DEPTH < NU) load_side(u + DEPTH, side[(u + DEPTH) % RING], mst[(u + DEPTH) % RING]);
    if constexpr (u + 1 < NU) put(std::integral_constant<int, u + 1>{});
    finish(u, x0, x1, side[u % RING], mst[u % RING]);
  });
  // The bias loads of consts() are CONSUMED on every path: on an edge tile whose units are all masked nothing read b8[], the loads stayed
  // "in flight" in the compiler's scoreboard across the back edge of the tile loop, and hipcc protected the first fragment registers
  // the K loop writes (the same VGPRs) with `s_waitcnt vmcnt(1)` / `vmcnt(0)` at the TOP OF EVERY K ITERATION of the bias-only forms
  // (EPI 0 / 1 / 5) -- a drain of the operand ring every two K tiles, found by tests/test_isa_cpu.py in round 6.  Here the wait is a
  // counted one behind stores that are long issued.
  asm volatile("" ::"v"(b8[0]), "v"(b8[4]));
}

// FMH: 16-row accumulator fragments per wave per tile half (3 -> 192-row tiles, 4 -> 256-row tiles, 5 -> 320-row tiles)
// KEEPB: keep the B0 fragments in registers for the 4th quadrant (16 more VGPRs) instead of re-reading them
template <int FMH, int EPI, bool KEEPB>
__global__ __launch_bounds__(512, 2) void gemm_nt_p8_kernel(const GemmParams p) {
  constexpr int AH = 32 * FMH;                 // rows of an A half-image (2 wave rows x FMH fragments x 16)
  constexpr int BM = 2 * AH, BN = 256;
  constexpr int AHB = AH * 128, BHB = 128 * 128;
  constexpr int OFF_A0 = 0, OFF_A1 = AHB, OFF_B0 = 2 * AHB, OFF_B1 = 2 * AHB + BHB;
  constexpr int BUF = 2 * AHB + 2 * BHB;       // one K tile: 64 KiB (FMH 4) / 72 KiB (FMH 5)
  constexpr int STG = 2 * BUF;                 // epilogue staging slab
  constexpr int SR = (163840 - STG) / 1024 >= 32 ? 32 : (163840 - STG) / 1024;    // its rows of 256 fp32: 32 / 16
  constexpr int NLA = (AH * 8 + 511) / 512;    // LDS-DMA instructions per thread per A half-image: 2 / 3 (the last one: waves 0-3 only)
  // loads of the NEXT K tile that may still be in flight when the current one is retired: its A0 half-image + one B half-image,
  // counted for the waves that issue the fewest (the waves with one more in flight merely wait for their oldest a little early)
  constexpr int VM_AHEAD = (AH * 8) / 512 + 2;
  static_assert(SR == 32 || SR == 16, "staging slab");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = p.ntm * p.ntn;
  const int nk = p.K >> 6;                      // K tiles per output tile (even: K % 128 == 0)
  if ((int)blockIdx.x >= nt) return;
  // (measurement ablation 4, tools/clock_probe.py: workgroup b leaves {shader-clock counter, 100 MHz real-time counter} at entry and at
  // exit in the int64 table the caller passes through `pre` -- the effective shader clock UNDER this kernel = d(cycles) / d(real time))
  // (the GELU epilogue writes GELU' through `pre`: its stamps travel through `aux`, which that epilogue does not read)
  unsigned long long* const stamps = (unsigned long long*)(EPI == 1 ? (void*)p.aux : (void*)p.pre);
  const bool stamp = (p.ablate == 4) && stamps;      // wave-uniform
  unsigned long long t_main = 0, t_epi = 0, t_mark = 0;
  if (stamp && tid == 0) {
    unsigned long long* t = stamps + 8 * blockIdx.x;
    t[0] = __builtin_readcyclecounter();
    t[1] = __builtin_amdgcn_s_memrealtime();
  }

  auto tile_of = [&](int w, int& m0, int& n0) {   // XCD-aware grouped order (see gemm_nt_bf16_kernel)
    const int xcd = w & 7, q = nt >> 3, r = nt & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
    const int gm = p.tile_group, per_group = gm * p.ntn, gid = t / per_group, first = gid * gm;
    const int gsz = min(p.ntm - first, gm), rem = t - gid * per_group;
    m0 = (first + rem % gsz) * BM;
    n0 = (rem / gsz) * BN;
  };

  // ---------------- producer: a continuous stream of K tiles over this workgroup's work items ----------------
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, 0x7FFFFFFF, 0x00020000);
  // This thread's 16-B chunk P = it*512 + tid of a half-image lies in row (P >> 3) = it*64 + (tid >> 3), physical slot P & 7, which
  // holds k-chunk kc = (P & 7) ^ ((row >> 1) & 7) -- independent of `it`.  Byte offset inside the operand = (tile row origin + half
  // + it*64) * ld (wave-uniform, SALU) + rowX (per thread), clamped to the chunk of the operand's last row (edge tiles re-read
  // that row; the rows / columns it feeds are masked by the epilogue).  No per-tile VGPR state.
  const int ldaB = (int)p.lda * 2, ldbB = (int)p.ldb * 2;
  const int kcb = (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
  const int rowA = (tid >> 3) * ldaB + kcb, maxA = (p.M - 1) * ldaB + kcb;
#if VLB_P8_WDRAIN
  // B half-image h, LDS row r  <-  tile column (r >> 5) * 64 + h * 32 + (r & 31): the 32-column groups a wave owns in the two halves are
  // neighbours in C (p8_drain_w)
  const int rowB = ((tid >> 8) * 64 + ((tid >> 3) & 31)) * ldbB + kcb, maxB = (p.N - 1) * ldbB + kcb;
#else
  const int rowB = (tid >> 3) * ldbB + kcb, maxB = (p.N - 1) * ldbB + kcb;
#endif
  int w_p = blockIdx.x, kt_p = 0, pm0 = 0, pn0 = 0;
  auto setup = [&](int w) { tile_of(w, pm0, pn0); };
  // half-image WHICH (0 A0 | 1 A1 | 2 B0 | 3 B1) of the producer's current K tile -> buffer `buf`
  auto stage = [&](auto which_c, auto buf_c) {
    constexpr int WHICH = decltype(which_c)::value, B_ = decltype(buf_c)::value;
    // The producer never stops (round 6): past the end of its work it re-stages the last K tile of its last output tile into the slots
    // the schedule frees anyway.  No branch in the load segment of a phase and none around the counted waits -- through round 5 a
    // `live` flag put one in front of every stage() and a two-way wait into phase 4 (the vendor's K loop has no branch at all:
    // profiles/r06_tn8_vs_vendor.txt); the one epilogue that lays its slab over the ring retires the surplus loads first.
    const int koff = kt_p * 128;
    if constexpr (WHICH < 2) {
      char* dst = smem + B_ * BUF + (WHICH == 0 ? OFF_A0 : OFF_A1);
#pragma unroll
      for (int it = 0; it < NLA; ++it) {
        if (it * 512 + 511 < AH * 8 || wave < (AH * 8 - it * 512) / 64) {
          const int vo = min((pm0 + WHICH * AH + it * 64) * ldaB + rowA, maxA);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst + (it * 512 + wave * 64) * 16), 16, vo, koff, 0, 0);
        }
      }
    } else {
      char* dst = smem + B_ * BUF + (WHICH == 2 ? OFF_B0 : OFF_B1);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
#if VLB_P8_WDRAIN
        const int vo = min((pn0 + (WHICH - 2) * 32 + it * 128) * ldbB + rowB, maxB);
#else
        const int vo = min((pn0 + (WHICH - 2) * 128 + it * 64) * ldbB + rowB, maxB);
#endif
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(dst + (it * 512 + wave * 64) * 16), 16, vo, koff, 0, 0);
      }
    }
  };
  auto advance = [&]() {     // the producer moves on to the next K tile (called in front of its first half-image)
    if (++kt_p == nk) {
      if (w_p + (int)gridDim.x < nt) {
        kt_p = 0;
        w_p += gridDim.x;
        setup(w_p);
      } else {
        kt_p = nk - 1;      // out of work: stay on the last K tile
      }
    }
  };
  // staging order of the four half-images of a K tile: the first is always A0 (read first, free first)
  constexpr int ORD1 = KEEPB ? 2 : 3, ORD2 = KEEPB ? 3 : 1, ORD3 = KEEPB ? 1 : 2;   // KEEPB: A0 B0 B1 A1 | else: A0 B1 A1 B0
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using W0 = std::integral_constant<int, 0>;
  using W1 = std::integral_constant<int, ORD1>;
  using W2 = std::integral_constant<int, ORD2>;
  using W3 = std::integral_constant<int, ORD3>;

  // ---------------- consumer state ----------------
  f32x4 acc[2 * FMH][4];           // [tile half x fragment row][B half x fragment column]
  bf16x8 af[FMH][2], bfr[2][2], bkeep[2][2];
  const int frow = lane & 15;
  const uint32_t c0 = (uint32_t)((((lane >> 4) ^ (frow >> 1)) << 4));
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const uint32_t a_rd = lds0 + (uint32_t)((wm * FMH * 16 + frow) * 128) + c0;       // + buf*BUF + half*AHB + i*2048, ^64 for k-step 1
  const uint32_t b_rd = lds0 + (uint32_t)(OFF_B0 + (wn * 32 + frow) * 128) + c0;    // + buf*BUF + half*BHB + j*2048

  auto read_a = [&](auto buf_c, auto half_c) {
    constexpr int B_ = decltype(buf_c)::value, H_ = decltype(half_c)::value;
    const uint32_t v0 = a_rd + B_ * BUF, v1 = v0 ^ 64u;
    constexpr int O = H_ * AHB;
    p8_lds_read<O + 0 * 2048>(af[0][0], v0); p8_lds_read<O + 0 * 2048>(af[0][1], v1);
    p8_lds_read<O + 1 * 2048>(af[1][0], v0); p8_lds_read<O + 1 * 2048>(af[1][1], v1);
    p8_lds_read<O + 2 * 2048>(af[2][0], v0); p8_lds_read<O + 2 * 2048>(af[2][1], v1);
    if constexpr (FMH >= 4) {
      p8_lds_read<O + 3 * 2048>(af[3][0], v0); p8_lds_read<O + 3 * 2048>(af[3][1], v1);
    }
    if constexpr (FMH == 5) {
      p8_lds_read<O + 4 * 2048>(af[4][0], v0); p8_lds_read<O + 4 * 2048>(af[4][1], v1);
    }
  };
  auto read_b = [&](auto buf_c, auto half_c, bf16x8 (&dst)[2][2]) {
    constexpr int B_ = decltype(buf_c)::value, H_ = decltype(half_c)::value;
    const uint32_t v0 = b_rd + B_ * BUF, v1 = v0 ^ 64u;
    constexpr int O = H_ * BHB;
    p8_lds_read<O>(dst[0][0], v0); p8_lds_read<O>(dst[0][1], v1);
    p8_lds_read<O + 2048>(dst[1][0], v0); p8_lds_read<O + 2048>(dst[1][1], v1);
  };
  // release the fragments of this phase to the MFMAs (every asm read above is retired), then the quadrant's 16 (20) MFMAs
  auto compute = [&](auto ha_c, auto hb_c, bf16x8 (&bq)[2][2]) {
    constexpr int HA = decltype(ha_c)::value, HB = decltype(hb_c)::value;
    if constexpr (FMH == 3) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1]), "+v"(bq[0][0]),
                     "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]));
    } else if constexpr (FMH == 4) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]),
                     "+v"(af[3][1]), "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1]), "+v"(af[3][0]),
                     "+v"(af[3][1]), "+v"(af[4][0]), "+v"(af[4][1]), "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[1][0]), "+v"(bq[1][1]));
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < FMH; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[HA * FMH + i][HB * 2 + j] = VLB_MFMA_16x16x32(bq[j][ks], af[i][ks], acc[HA * FMH + i][HB * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---------------- prologue: K tile 0 complete, the first two half-images of K tile 1 in flight ----------------
  setup(w_p);
  stage(W0{}, I0{}); stage(W1{}, I0{}); stage(W2{}, I0{}); stage(W3{}, I0{});
  advance();
  stage(W0{}, I1{}); stage(W1{}, I1{});
  __builtin_amdgcn_s_waitcnt(0x0F70 | VM_AHEAD);
  p8_barrier();
  if (wm == 1) p8_barrier();       // the wm = 1 waves run one segment behind

  int g = 0;                       // K tiles consumed by this workgroup so far
  const uint32_t seed = (EPI == 3 || EPI == 6) ? *p.seed : 0u;
  for (int w = blockIdx.x; w < nt; w += gridDim.x) {
#pragma unroll
    for (int i = 0; i < 2 * FMH; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (stamp) t_mark = __builtin_readcyclecounter();
    for (int kt = 0; kt < nk; kt += 2, g += 2) {
      // ======== K tile g (buffer 0) ========
      // phase 1: A0 + B0 -> quadrant (0,0)
      if constexpr (KEEPB) read_b(I0{}, I0{}, bkeep); else read_b(I0{}, I0{}, bfr);
      __builtin_amdgcn_sched_barrier(0);
      read_a(I0{}, I0{});
      stage(W2{}, I1{});
      p8_barrier();
      if constexpr (KEEPB) compute(I0{}, I0{}, bkeep); else compute(I0{}, I0{}, bfr);
      p8_barrier();
      // phase 2: B1 -> (0,1)
      read_b(I0{}, I1{}, bfr);
      stage(W3{}, I1{});
      p8_barrier();
      compute(I0{}, I1{}, bfr);
      p8_barrier();
      // phase 3: A1 -> (1,1)
      read_a(I0{}, I1{});
      advance();
      stage(W0{}, I0{});
      p8_barrier();
      compute(I1{}, I1{}, bfr);
      p8_barrier();
      // phase 4: (B0 again) -> (1,0); K tile g+1 retired
      if constexpr (!KEEPB) read_b(I0{}, I0{}, bfr);
      stage(W1{}, I0{});
      __builtin_amdgcn_s_waitcnt(0x0F70 | VM_AHEAD);
      p8_barrier();
      if constexpr (KEEPB) compute(I1{}, I0{}, bkeep); else compute(I1{}, I0{}, bfr);
      p8_barrier();
      // ======== K tile g+1 (buffer 1) ========
      if constexpr (KEEPB) read_b(I1{}, I0{}, bkeep); else read_b(I1{}, I0{}, bfr);
      __builtin_amdgcn_sched_barrier(0);
      read_a(I1{}, I0{});
      stage(W2{}, I0{});
      p8_barrier();
      if constexpr (KEEPB) compute(I0{}, I0{}, bkeep); else compute(I0{}, I0{}, bfr);
      p8_barrier();
      read_b(I1{}, I1{}, bfr);
      stage(W3{}, I0{});
      p8_barrier();
      compute(I0{}, I1{}, bfr);
      p8_barrier();
      read_a(I1{}, I1{});
      advance();
      stage(W0{}, I1{});
      p8_barrier();
      compute(I1{}, I1{}, bfr);
      p8_barrier();
      if constexpr (!KEEPB) read_b(I1{}, I0{}, bfr);
      stage(W1{}, I1{});
      __builtin_amdgcn_s_waitcnt(0x0F70 | VM_AHEAD);
      p8_barrier();
      if constexpr (KEEPB) compute(I1{}, I0{}, bkeep); else compute(I1{}, I0{}, bfr);
      p8_barrier();
    }
    if (wm == 0) p8_barrier();     // let the lagging wave group finish its last quadrant: the epilogue runs aligned
    if (stamp) {
      const unsigned long long now = __builtin_readcyclecounter();
      t_main += now - t_mark;
      t_mark = now;
    }

    // ---------------- epilogue: raw fp32 accumulators -> staging slab -> fused math on 8-column groups -> bf16 ----------------
    int m0, n0;
    tile_of(w, m0, n0);
    if (p.ablate == 1) {
      // (timing ablation: no epilogue at all)
#if VLB_P8_WDRAIN
    } else if ((w + (int)gridDim.x < nt || EPI == 0 || (p.p8_flags & 2)) && !(FMH == 5 && (EPI == 6 || EPI == 7))) {
      // (a workgroup's LAST tile takes this path as well for the bias-only / plain epilogue -- measured at M = 25856: attention-output
      // data gradient 37.2 -> 33.5 us, QKV forward 103.1 -> 100.3; with side tensors the 128-row slab over the idle ring is faster:
      // QKV data gradient 84.7 vs 90.4 -- ; p8_flags bit 1 / VLB_GEMM_P8_LASTW=1 forces it for every epilogue)
      // mid-stream tile: wave-private drain beside the operand ring (not in the one instantiation without the registers for it: its
      // spills land in the K loop; that one keeps the shared slab)
      if (m0 + BM <= p.M && n0 + BN <= p.N) p8_drain_w<FMH, EPI, false>(p, acc, lds0 + STG, m0, n0, wm, wn, tid, seed);
      else p8_drain_w<FMH, EPI, true>(p, acc, lds0 + STG, m0, n0, wm, wn, tid, seed);
#endif
    } else if (w + (int)gridDim.x >= nt) {
      // last tile of this workgroup: every K tile the consumer needs has been consumed -> drain through a 128-row slab laid over the ring
      __builtin_amdgcn_s_waitcnt(0x0F70);      // (the producer kept staging: retire its surplus half-images before the slab reuses their slots)
      p8_barrier();
      p8_drain<FMH, EPI, (STG >= 131072 ? 4 : 3)>(p, acc, lds0, m0, n0, wm, wn, lane, tid, seed);      // 128 (96) slab rows of 1 KiB
    } else {
      p8_drain<FMH, EPI, (SR == 32 ? 1 : 0)>(p, acc, lds0 + STG, m0, n0, wm, wn, lane, tid, seed);
    }
    // No drain of the epilogue's stores (round 4; p8_flags bit 0 restores the vmcnt(0) of round 3).  The next output tile's K tile 0
    // was retired by the last in-loop wait, BEFORE the epilogue.  Its K tile 1 -- whose first half-images were staged in phases 3 / 4
    // of the last K iteration, i.e. they are OLDER than the epilogue's stores -- is retired by the next in-loop vmcnt(VM_AHEAD).
    // Loads and stores retire out of order RELATIVE TO EACH OTHER (only loads among loads and stores among stores are ordered), so
    // "the stores are older" proves nothing.  The invariant that makes the counted wait sound is this one: between the loads a wait
    // retires and the wait itself, at least VM_AHEAD YOUNGER LOADS are issued (here: K tile 2's W0 and W1 half-images, phases 1-2 of
    // the next tile).  Then, if a load L being retired were still outstanding at a point where vmcnt <= VM_AHEAD, the VM_AHEAD
    // younger loads would be outstanding as well (loads retire in order among themselves): VM_AHEAD + 1 operations outstanding, a
    // contradiction -- however many stores are in the queue and whenever they are acknowledged.  Every counted wait of this file
    // satisfies it by construction (one stage() per phase, VM_AHEAD of them between a K tile's last load and its wait); a wait that
    // leans on anything else -- the software-pipelined LayerNorm backward of round 4 relied on load/store order and produced NaN
    // gradients -- is unsound.  VLB_GEMM_P8_DRAIN=1 stays available for a cold-cache A/B.  What the missing drain buys: a wave no
    // longer sits out the L2 / HBM acknowledgement of its last stores behind every tile (all 256 CUs end their epilogues together:
    // 119-318 MB of stores per launch in bursts); they drain under the first three phases of the next tile instead.
    if (p.p8_flags & 1) __builtin_amdgcn_s_waitcnt(0x0F70);
    p8_barrier();
    if (wm == 1 && w + (int)gridDim.x < nt) p8_barrier();   // re-establish the one-segment lag for the next output tile
    if (stamp) t_epi += __builtin_readcyclecounter() - t_mark;
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // surplus half-images of the branch-free producer (s_endpgm waits for them as well)
  if (stamp && tid == 0) {      // + the shader cycles this workgroup's wave 0 spent in K loops / in epilogues (incl. the tile-end barriers), tiles done
    unsigned long long* t = stamps + 8 * blockIdx.x;
    t[2] = __builtin_readcyclecounter();
    t[3] = __builtin_amdgcn_s_memrealtime();
    t[4] = t_main;
    t[5] = t_epi;
    t[6] = (unsigned long long)((nt - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
  }
}

static int g_p8_wgs = 256;

template <int FMH, int EPI, bool KEEPB>
int p8_launch(GemmParams& p, int group, hipStream_t stream) {
  constexpr int smem = 163840;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_p8_kernel<FMH, EPI, KEEPB>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vlb_set_error("gemm_p8: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
      return VLB_ERR_HIP;
    }
    attr_set = true;
  }
  p.ntm = vlb_cdiv(p.M, 64 * FMH);
  p.ntn = vlb_cdiv(p.N, 256);
  p.tile_group = group;
  int gx = p.ntm * p.ntn;
  const int cap = (g_p8_wgs >= 8 && g_p8_wgs <= 256) ? (g_p8_wgs & ~7) : 256;
  if (gx > cap) gx = cap;          // one persistent workgroup per CU (a multiple of 8: work item w and block b share an XCD)
  hipLaunchKernelGGL((gemm_nt_p8_kernel<FMH, EPI, KEEPB>), dim3(gx), dim3(512), smem, stream, p);
  VLB_CHECK_LAUNCH("vlb_gemm_nt_bf16(p8)");
  return 1;
}

template <int FMH, bool KEEPB>
int p8_launch_epi(GemmParams& p, int epi, int group, hipStream_t stream) {
  switch (epi) {
    case 0: return p8_launch<FMH, 0, KEEPB>(p, group, stream);
    case 1: return p8_launch<FMH, 1, KEEPB>(p, group, stream);
    case 2: return p8_launch<FMH, 2, KEEPB>(p, group, stream);
    case 3: return p8_launch<FMH, 3, KEEPB>(p, group, stream);
    case 4: return p8_launch<FMH, 4, KEEPB>(p, group, stream);
    case 5: return p8_launch<FMH, 5, KEEPB>(p, group, stream);
    case 6: return p8_launch<FMH, 6, KEEPB>(p, group, stream);
    case 7: return p8_launch<FMH, 7, KEEPB>(p, group, stream);
    case 8: return p8_launch<FMH, 8, KEEPB>(p, group, stream);
    default: return p8_launch<FMH, 10, KEEPB>(p, group, stream);
  }
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

inline bool aligned16(const void* q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace

// tuning knobs (environment defaults, run-time override through vlb_gemm_set_option for A/B measurements inside one process)
// p8_mode: 0 off | 1 cost model (default) | 3 / 4 / 5: force the 192- / 256- / 320-row tile wherever the kernel applies
// p8_wgs: persistent workgroups per launch (<= 256 = one per CU).  Fewer leave CUs to a kernel running on another stream (the
// weight-gradient GEMMs of the side stream): an MFMA-bound kernel then fills the HBM-bound epilogue bursts of this one.
// p8_ablate (tools/p8_check.py ablate; results are WRONG when != 0): 1 no epilogue | 2 epilogue without its global stores | 4 (results
// correct) per-workgroup clock stamps into the table passed as `pre` (tools/clock_probe.py)
static int g_opt[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
static const char* const g_opt_name[8] = {"p8_mode", "p8_keepb", "p8_group", "p8_min_tiles", "p8_wgs", "p8_ablate", "p8_tile192", "p8_drain"};
static void p8_options_init() {
  if (g_opt[0] >= 0) return;
  g_opt[0] = env_int("VLB_GEMM_P8", 1);
  g_opt[1] = env_int("VLB_GEMM_P8_KEEPB", 1);
  g_opt[2] = env_int("VLB_GEMM_P8_GROUP", 2);
  g_opt[3] = env_int("VLB_GEMM_P8_MIN_TILES", 160);
  g_opt[4] = env_int("VLB_GEMM_P8_WGS", 256);
  g_opt[5] = 0;
  g_opt[6] = env_int("VLB_GEMM_P8_192", 1);
  g_opt[7] = env_int("VLB_GEMM_P8_DRAIN", 0);      // 1: vmcnt(0) behind every output tile (round-3 behaviour, A/B)
}

void vlb_ln_set_fwd_rows(int v);      // layernorm.hip
void vlb_ln_set_bwd4(int v);

extern "C" int vlb_gemm_set_option(const char* name, int value) {
  p8_options_init();
  VLB_CHECK_ARG(name && value >= 0, "vlb_gemm_set_option: null name / negative value");
  if (!strcmp(name, "tn8_mode")) {
    vlb_tn8_set_mode(value);
    return VLB_OK;
  }
  if (!strcmp(name, "nt_ring")) {
    vlb_nt_set_ring(value);
    return VLB_OK;
  }
  if (!strcmp(name, "nt_stagger")) {
    vlb_nt_set_stagger(value);
    return VLB_OK;
  }
  if (!strcmp(name, "tn8_wgs")) {
    vlb_tn8_set_wgs(value);
    return VLB_OK;
  }
  if (!strcmp(name, "ln_fwd_rows")) {      // (the LayerNorm kernels' variants ride on the same knob)
    vlb_ln_set_fwd_rows(value);
    return VLB_OK;
  }
  if (!strcmp(name, "ln_bwd4")) {
    vlb_ln_set_bwd4(value);
    return VLB_OK;
  }
  if (!strcmp(name, "tn8_uneven")) {
    vlb_tn8_set_uneven(value);
    return VLB_OK;
  }
  if (!strcmp(name, "tn8_m32")) {          // weight-gradient core: 1 = 32x32x16 matrix instructions, 0 = 16x16x32 (default)
    vlb_tn8_set_m32(value);
    return VLB_OK;
  }
  if (!strcmp(name, "tn8_ablate")) {       // measurement builds only (-DVLB_TN8_PROBE)
    vlb_tn8_set_ablate(value);
    return VLB_OK;
  }
  for (int i = 0; i < 8; ++i)
    if (!strcmp(name, g_opt_name[i])) {
      g_opt[i] = value;
      return VLB_OK;
    }
  vlb_set_error("vlb_gemm_set_option: unknown option %s", name);
  return VLB_ERR_ARG;
}

int vlb_gemm_p8_try(GemmParams& p, hipStream_t stream) {
  p8_options_init();
  const int mode = g_opt[0], keepb = g_opt[1], group = g_opt[2], min_tiles = g_opt[3];
  g_p8_wgs = g_opt[4];
  if (!mode || p.out_f32 != 0 || p.c_split_stride != 0) return 0;
  if ((p.K % 128) != 0 || p.k_per_split < p.K) return 0;
  int epi;
  if (p.act == 0 && p.res && p.res_stats) epi = p.drop_thr ? 6 : 7;
  else if (p.act == 0) epi = p.res ? (p.drop_thr ? 3 : 4) : (p.drop_thr ? -1 : 0);
  else if (p.act == 4) epi = 1;
  else if (p.act == 5) epi = 2;
  else if (p.act == 2) epi = 5;
  else if (p.act == 7 && p.res && !p.drop_thr) epi = 8;        // relu(acc + bias + res): the Bottleneck forward tail (vision path)
  else if (p.act == 8 && !p.res && !p.drop_thr) epi = 10;      // acc where aux > 0: ReLU backward in a dgrad epilogue
  else epi = -1;
  if (epi < 0) return 0;
  // 16-B vector accesses on every side tensor; 31-bit byte offsets inside A and B
  if ((p.ldc % 8) || !aligned16(p.C) || (p.lda % 8) || (p.ldb % 8) || !aligned16(p.A) || !aligned16(p.B)) return 0;
  if (p.res && ((p.ldres % 8) || !aligned16(p.res))) return 0;
  if ((epi == 2 || epi == 10) && ((p.ldaux % 8) || !aligned16(p.aux))) return 0;
  if (epi == 1 && p.pre && ((p.ldpre % 8) || !aligned16(p.pre))) return 0;
  if (p.bias && ((uintptr_t)p.bias & 15)) return 0;
  if (p.res_stats && (((uintptr_t)p.res_gamma & 15) || ((uintptr_t)p.res_beta & 15) || ((uintptr_t)p.res_stats & 7))) return 0;
  if ((long)p.M * p.lda * 2 >= (1L << 31) || (long)p.N * p.ldb * 2 >= (1L << 31)) return 0;
  // tile height: whole rounds of 256 resident workgroups, time per round ~ tile area (the 192-row tile pays ~15 % in the main loop -- 8192^3: 1006 vs 1216 TFLOP/s --:
  // 12 instead of 16 MFMAs per phase behind the same barriers).  A height qualifies with enough tiles for one workgroup per CU --
  // below that the 128x128 kernel fills the chip better -- except for a long K loop, where the 256-row tile's main loop still wins
  // with 60 % of the CUs busy (M = 12928, N = 768: 153 tiles; K = 3072: 84.7 -> 78.2 us, K = 2304: 64.1 -> 58.7 us measured; at
  // K = 768 the 128x128 kernel is faster: 24.9 vs 27.8 us)
  const long nn = vlb_cdiv(p.N, 256);
  const long t3 = (long)vlb_cdiv(p.M, 192) * nn, t4 = (long)vlb_cdiv(p.M, 256) * nn, t5 = (long)vlb_cdiv(p.M, 320) * nn;
  const double c3 = (double)((t3 + 255) / 256) * 192.0 * 1.15, c4 = (double)((t4 + 255) / 256) * 256.0, c5 = (double)((t5 + 255) / 256) * 320.0;
  const bool ok3 = t3 >= min_tiles, ok4 = t4 >= min_tiles || (t4 >= 128 && p.K >= 1536), ok5 = t5 >= min_tiles;
  int fmh = 0;
  double best = 1e30;
  if (ok4) { fmh = 4; best = c4; }
  if (ok5 && c5 < 0.97 * best) { fmh = 5; best = c5; }
  if (ok3 && g_opt[6] && c3 < 0.97 * best) { fmh = 3; best = c3; }
  if (mode == 3 || mode == 4 || mode == 5) fmh = mode;
  if (!fmh) return 0;
  const int g = group < 1 ? 1 : group;
  p.ablate = g_opt[5];
  p.p8_flags = (g_opt[7] & 1) | (env_int("VLB_GEMM_P8_LASTW", 0) ? 2 : 0);      // p8_drain bit 0; VLB_GEMM_P8_LASTW: wave-private drain for last tiles too
  if (fmh == 3) return p8_launch_epi<3, true>(p, epi, g, stream);
  if (fmh == 5) return p8_launch_epi<5, false>(p, epi, g, stream);
  return keepb ? p8_launch_epi<4, true>(p, epi, g, stream) : p8_launch_epi<4, false>(p, epi, g, stream);
}
