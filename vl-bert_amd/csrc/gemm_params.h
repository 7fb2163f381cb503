// Parameter block shared by the GEMM kernels of libvlbert_hip.so (gemm.hip, gemm_p8.hip).
#pragma once
#include "vlb_common.h"

struct GemmParams {
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb;
  int M, N, K;
  int k_per_split;           // K range handled by one blockIdx.y slice (multiple of 64)
  const float* bias;         // [N] fp32 or null
  int act;                   // 0 none, 1 gelu, 2 relu, 3 multiply by gelu'(aux), 4 gelu with gelu'(x) -> pre, 5 multiply by aux, 6 tanh
  const bf16_t* aux; long ldaux;
  bf16_t* pre; long ldpre;   // optional pre-activation output (act==1)
  const bf16_t* res; long ldres;
  // "LayerNorm residual": res holds the fp16 PRE-LayerNorm rows of the producing sublayer and the residual that is added is the
  // LayerNorm output re-materialised in fp32, (res - mean[m]) * rstd[m] * gamma[n] + beta[n] (res_stats = [M][2] (mean, rstd)).
  // The encoder's residual stream then never passes through a bf16 rounding (DESIGN.md "precision").
  const float* res_stats; const float* res_gamma; const float* res_beta;
  int c_f16;                 // bf16-output kernels write C as fp16 instead (the pre-LayerNorm sums)
  int ablate;                // gemm_p8 timing ablations (results are WRONG when != 0): 1 no epilogue | 2 epilogue without global stores |
                             // 3 epilogue without the LDS slab round trip
  int p8_flags;              // gemm_p8 run-time variants: bit 0 = keep the vmcnt(0) drain behind every output tile's epilogue (round-3 behaviour)
  int stagger;               // 128x128 kernel: second-resident workgroups start `stagger` x ~3.4 us late (phase offset between the two
                             // workgroups of a CU, so one runs its epilogue under the other's MFMA loop)
  uint32_t drop_thr; float drop_scale; const uint32_t* seed; uint32_t tag;
  void* C; long ldc;
  long c_split_stride;       // elements between the outputs of consecutive K splits (slab split-K), 0 otherwise
  int out_f32;               // 0: bf16 store, 1: fp32 store, 2: fp32 atomicAdd
  int ntm, ntn;
  int tile_group;            // tile-rows per L2 group (see the kernel's tile order)
  // implicit 3x3 convolution (CONV kernels): A = NHWC activation [rows = n*H*W, conv_C], K = 9*conv_C, K tile kt reads tap
  // kt / (conv_C/64), channels (kt % (conv_C/64))*64.. of pixel (y + (tap/3-1)*dil, x + (tap%3-1)*dil); out-of-image taps read `zero`
  int conv_C, conv_H, conv_W, conv_dil;
  const bf16_t* zero;        // >= 16 B of zeros
};

#define GLDS_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// gemm_p8.hip: 256x256 / 320x256 tiles, 8-phase schedule.  Returns 1 when it took the call, 0 when the shape / epilogue /
// alignment is outside what it covers (the caller then uses the 128x128 kernel), < 0 on error.
int vlb_gemm_p8_try(GemmParams& p, hipStream_t stream);

// gemm_tn8.hip: weight gradients C[Mo,No] (fp32) (+)= A[R,Mo]^T B[R,No] with 256x256 tiles, 8-phase schedule, slab split-K.
// Returns the number of K slices it used (>= 1; > 1 or force_slab: the partial tiles are in `workspace`, slice stride Mo * round4(No),
// and the caller runs the slab reduce), 0 when the shape is outside what it covers, < 0 on error.
int vlb_gemm_tn8_try(GemmParams& p, float* C, long ldc, float* colsum, float* workspace, long workspace_floats, int accumulate,
                     bool force_slab, hipStream_t stream);
int vlb_tn8_pick_splits(int Mo, int No, int R);
int vlb_gemm_tn8_group(int n, const void* const* A, const long* lda, const void* const* B, const long* ldb, float* const* C,
                       const long* ldc, int R, const int* Mo, const int* No, float* const* colsum, float* workspace,
                       long workspace_floats, int accumulate, int* slices, long* ws_off, hipStream_t stream);
void vlb_nt_set_stagger(int v);
void vlb_nt_set_ring(int v);
void vlb_tn8_set_mode(int v);
void vlb_tn8_set_wgs(int v);
void vlb_tn8_set_uneven(int v);
void vlb_tn8_set_m32(int v);
void vlb_tn8_set_ablate(int v);
