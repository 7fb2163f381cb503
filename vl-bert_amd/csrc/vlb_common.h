// Shared device/host helpers for libvlbert_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vlbert_hip.h"  // every extern "C" definition is checked against the public prototypes

// The 16-bit activation / working-weight / gradient type is a BUILD-TIME choice of the whole library:
//   default            bfloat16  -> libvlbert_hip.so      (BASELINE.json's headline precision; no loss scaling needed)
//   -DVLB_ACT_F16      IEEE fp16 -> libvlbert_hip_f16.so  (the reference's own mixed precision: Apex O2 fp16 + a static loss scale,
//                      pretrain/function/train.py:345-352; 3 more mantissa bits on every GEMM operand, same MFMA rate on gfx950)
// Every kernel converts through the helpers below and issues its matrix instruction through VLB_MFMA_16x16x32, so the sources
// are shared; names keep the "bf16" of the default build (bf16_t = "the 16-bit storage type").
typedef uint16_t bf16_t;  // raw 16-bit element in memory (bfloat16, or IEEE fp16 in the VLB_ACT_F16 build)
#ifdef VLB_ACT_F16
typedef _Float16 vlb_h16;
#define VLB_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define VLB_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_f16
#define VLB_ACT_IS_F16 1
#else
typedef __bf16 vlb_h16;
#define VLB_MFMA_16x16x32 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define VLB_MFMA_32x32x16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
#define VLB_ACT_IS_F16 0
#endif
typedef __attribute__((ext_vector_type(8))) vlb_h16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define VLB_WAVE 64

// ---- error plumbing (C ABI: 0 = ok, <0 = error, message via vlb_last_error) ----
extern "C" const char* vlb_last_error(void);
void vlb_set_error(const char* fmt, ...);
#define VLB_OK 0
#define VLB_ERR_ARG (-1)
#define VLB_ERR_HIP (-2)
#define VLB_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      vlb_set_error(__VA_ARGS__);                \
      return VLB_ERR_ARG;                        \
    }                                            \
  } while (0)
#define VLB_CHECK_LAUNCH(name)                                              \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      vlb_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return VLB_ERR_HIP;                                                   \
    }                                                                       \
  } while (0)

// ---- 16-bit storage type <-> f32 ----
#ifdef VLB_ACT_F16
__device__ __forceinline__ float bf2f(bf16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }   // RNE
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  const h2_t v = {(_Float16)lo, (_Float16)hi};                                                        // v_cvt_pk / two v_cvt_f16_f32
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bflo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float bfhi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }  // RNE (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {      // ONE v_cvt_pk_bf16_f32 (two scalar conversions + shift + or otherwise)
  typedef __attribute__((ext_vector_type(2))) __bf16 vlb_bf2_t;
  typedef __attribute__((ext_vector_type(2))) float vlb_fl2_t;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((vlb_fl2_t){lo, hi}, vlb_bf2_t));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif

// ---- fp16 (IEEE half) <-> f32: the pre-LayerNorm sums of the encoder (the residual stream) are kept in fp16 -- 3 more mantissa
// bits than bf16 at the same 2 bytes; their magnitudes are O(1..10), far inside the fp16 range (DESIGN.md "precision") ----
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }   // RNE
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {        // ONE v_cvt_pk_f16_f32
  typedef __attribute__((ext_vector_type(2))) _Float16 vlb_h2_t;
  typedef __attribute__((ext_vector_type(2))) float vlb_fl2h_t;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((vlb_fl2h_t){lo, hi}, vlb_h2_t));
}
__device__ __forceinline__ float hlo(uint32_t w) { return h2f((uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float hhi(uint32_t w) { return h2f((uint16_t)(w >> 16)); }
// 16-bit element pair decoded as fp16 or bf16
__device__ __forceinline__ float dec_lo(uint32_t w, bool f16) { return f16 ? hlo(w) : bflo(w); }
__device__ __forceinline__ float dec_hi(uint32_t w, bool f16) { return f16 ? hhi(w) : bfhi(w); }
__device__ __forceinline__ uint32_t pack2o(float lo, float hi, bool f16) { return f16 ? pack2h(lo, hi) : pack2bf(lo, hi); }

// ---- streaming (non-temporal) 16-byte accesses: for tensors that are written now and read much later (GELU' kept for the backward
// pass, weight gradients kept for the optimizer, Adam moments) or read exactly once -- they bypass the cache allocation, so the
// 256 MB Infinity Cache / the L2s keep what the NEXT kernel reads (measured on FFN1 -> FFN2: 311 -> 292 us for the pair).
// -DVLB_NO_NT turns them back into plain accesses (A/B builds). ----
typedef __attribute__((ext_vector_type(4))) unsigned int vlb_u32x4;
typedef __attribute__((ext_vector_type(4))) float vlb_f32x4;
__device__ __forceinline__ void vlb_store_nt(uint4* ptr, const uint4& v) {
#ifdef VLB_NO_NT
  *ptr = v;
#else
  __builtin_nontemporal_store(__builtin_bit_cast(vlb_u32x4, v), (vlb_u32x4*)ptr);
#endif
}
__device__ __forceinline__ void vlb_store_nt(float4* ptr, const float4& v) {
#ifdef VLB_NO_NT
  *ptr = v;
#else
  __builtin_nontemporal_store(__builtin_bit_cast(vlb_f32x4, v), (vlb_f32x4*)ptr);
#endif
}
__device__ __forceinline__ uint4 vlb_load_nt(const uint4* ptr) {
#ifdef VLB_NO_NT
  return *ptr;
#else
  return __builtin_bit_cast(uint4, __builtin_nontemporal_load((const vlb_u32x4*)ptr));
#endif
}
__device__ __forceinline__ float4 vlb_load_nt(const float4* ptr) {
#ifdef VLB_NO_NT
  return *ptr;
#else
  return __builtin_bit_cast(float4, __builtin_nontemporal_load((const vlb_f32x4*)ptr));
#endif
}

// ---- wave reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- counter-based dropout RNG ----
// keep(seed, tag, idx): one 32-bit hash per PAIR of elements, 16 bits each.
// p_eff = thr/65536; the same function is evaluated in forward and backward so no
// mask tensor is stored.  `seed` lives in device memory (graph replays re-read it).
__device__ __forceinline__ uint32_t vlb_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// 32 random bits for element PAIR pair_idx under a (wave-uniform) key: two 16-bit fields, one per element.
// (Round 2 tried a version built only from 24-bit multiplies -- v_mad_u32_u24 / v_mul_u32_u24 + xorshifts, 11 issue slots, statistically
// as good on the CPU study in tools/hash_quality.py -- on the theory that the three v_mul_lo_u32 here are quarter-rate.  Same-box A/B of
// the whole step: 21.10 / 21.24 ms with this hash, 21.32 / 21.33 ms with the 24-bit one at batch 256, 5.79 vs 5.77 ms at batch 32:
// no gain, so the simpler, bijective 32-bit form stays.)
__device__ __forceinline__ uint32_t vlb_pair_hash(uint32_t pair_idx, uint32_t key) { return vlb_hash32(pair_idx * 0x9E3779B1u + key); }
__device__ __forceinline__ uint32_t vlb_rng_key(uint32_t seed, uint32_t tag) { return vlb_hash32(seed ^ (tag * 0x85ebca6bu + 0x632be5abu)); }
__device__ __forceinline__ uint32_t vlb_rng_pair(uint32_t seed, uint32_t tag, uint32_t pair_idx) {
  return vlb_pair_hash(pair_idx, vlb_rng_key(seed, tag));
}
__device__ __forceinline__ bool vlb_keep(uint32_t seed, uint32_t tag, uint32_t idx, uint32_t thr) {
  uint32_t h = vlb_rng_pair(seed, tag, idx >> 1);
  uint32_t bits = (idx & 1u) ? (h >> 16) : (h & 0xffffu);
  return bits >= thr;
}
static inline uint32_t vlb_drop_thr(float p) {
  if (p <= 0.f) return 0u;
  double t = (double)p * 65536.0 + 0.5;
  if (t > 65535.0) t = 65535.0;
  return (uint32_t)t;
}
static inline float vlb_drop_scale(uint32_t thr) { return thr ? 65536.0f / (65536.0f - (float)thr) : 1.0f; }

// ---- erf-GELU and its derivative (external/pytorch_pretrained_bert/modeling.py:114-120) ----
// erf by Abramowitz & Stegun 7.1.26 (|abs err| < 1.5e-7, i.e. fp32-roundoff class and far below the bf16
// output rounding): 1 rcp + 1 exp + 6 fma instead of libm's branchy ~40-instruction erff -- the GELU
// epilogue runs 64 times per lane per GEMM tile, so this is on the critical path of the FFN GEMMs.
__device__ __forceinline__ float erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp), not the IEEE divide sequence
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return x * 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// gelu(x) and gelu'(x) together: both need the same exp(-x^2/2) (erf's Gaussian factor and the pdf), so producing the
// derivative in the FORWARD epilogue costs two extra fmas and lets the backward epilogue be a plain multiply.
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-0.5f * x * x);
  const float cdf = 0.5f * (1.0f + copysignf(1.0f - p * t * e, x));
  g = x * cdf;
  dg = fmaf(x * 0.39894228040143268f, e, cdf);
}

// two elements at a time on the packed-fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32: 2 lanes-elements per issue slot).  The GELU
// epilogue of the FFN GEMM is VALU-bound (measured: 65 us of vector issue per 25856 x 3072 output with the scalar form);
// this form needs 11 instructions per element instead of ~25.  Same formulas as gelu_both.
typedef float vlb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_both2(vlb_f2 x, vlb_f2& g, vlb_f2& dg) {
  const vlb_f2 ax = __builtin_elementwise_abs(x) * 0.70710678118654752f;
  const vlb_f2 den = __builtin_elementwise_fma(ax, (vlb_f2)(0.3275911f), (vlb_f2)(1.0f));
  const vlb_f2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
  vlb_f2 p = __builtin_elementwise_fma(t, (vlb_f2)(1.061405429f), (vlb_f2)(-1.453152027f));
  p = __builtin_elementwise_fma(p, t, (vlb_f2)(1.421413741f));
  p = __builtin_elementwise_fma(p, t, (vlb_f2)(-0.284496736f));
  p = __builtin_elementwise_fma(p, t, (vlb_f2)(0.254829592f));
  const vlb_f2 xx = x * x * (-0.5f * 1.44269504088896341f);
  const vlb_f2 e = {__builtin_amdgcn_exp2f(xx.x), __builtin_amdgcn_exp2f(xx.y)};      // exp(-x^2 / 2)
  const vlb_f2 erfa = (vlb_f2)(1.0f) - p * t * e;                                        // erf(|x| / sqrt 2)
  const vlb_f2 sg = {__builtin_copysignf(erfa.x, x.x), __builtin_copysignf(erfa.y, x.y)};
  const vlb_f2 cdf = __builtin_elementwise_fma(sg, (vlb_f2)(0.5f), (vlb_f2)(0.5f));
  g = x * cdf;
  dg = __builtin_elementwise_fma(x * 0.39894228040143268f, e, cdf);
}

static inline int vlb_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
