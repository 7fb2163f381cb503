// bf16 "TN" GEMM for weight gradients on gfx950 (CDNA4), large-tile core:
//     C[Mo, No] (fp32) (+)= A[R, Mo]^T . B[R, No]        (reduction over the R rows; autograd's grad_output.t().mm(input))
// Same machine as gemm_p8.hip -- one persistent 8-wave workgroup per CU, 256 x 256 output tile, 64 reduction rows per K tile,
// two K-tile buffers of four 16-KiB half-images [A0 | A1 | B0 | B1] filled by LDS-DMA one half-image per phase and retired by a
// counted vmcnt, 4 quadrant phases per K tile with the two wave groups one segment apart -- with the two differences a
// row-reduction needs:
//   * operands are consumed as the forward / backward passes left them (row-major [R, cols]); a half-image is [64 rows][128
//     columns] (256-B rows), and the MFMA fragments (8 consecutive REDUCTION rows of one column per lane) come out of it through
//     the LDS transpose read ds_read_b64_tr_b16 (two per fragment), with the 32-B column blocks XOR-swizzled by
//     f(row) = (row & 3) | ((row >> 3) & 1) << 2 on the DMA source address (layout and conflict analysis: gemm_tn_bf16_kernel);
//   * few output tiles and a very long reduction: the work items of a launch are (tile, K slice) pairs, every slice writes its fp32
//     partial tile to a slab (plain 16-B stores) and a streaming reduce adds the slabs (gemm.hip: splitk_reduce_kernel).  Items
//     are ordered slice-major (all tiles of a slice are concurrent and share their operand rows in L2).
// The column sums of A (bias gradients) are taken from the A fragments with v_dot2c_f32_bf16 behind the MFMAs of the phase that holds
// them.  The waves that hold the same A columns take turns: the ntn tiles of a tile row x the 4 wave columns wn of a workgroup are
// ntn * 4 slots, slot tile_n * 4 + wn sums the K-tile PAIRS p with p % (ntn * 4) == slot (p counted from row 0 of the operand, so the K
// slices of one gradient partition the pairs too), and every wave adds its partials to colsum[] with fp32 atomics at the end of the
// item.  (Through round 5 the wn = 0 waves of the tile_n = 0 tiles did all of it: with one item per workgroup those 24 of 216
// workgroups ran 17 % longer than the rest -- 3430 vs 2920 cycles per K tile -- and the launch ended with them: 364 us with the
// sums, 331 us without; with the turns 341-345 us.  Splitting each phase's 32 v_dot2c over the four wn waves instead -- 8 each, in
// the turns of the tile only -- measured the same launch time with four copies of the code, and its first form sent csum[] to
// scratch: the compiler merged the four cases over a dynamic index.)
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "gemm_params.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) vlb_h16 bf16x2_t;
#ifdef VLB_ACT_F16
#define VLB_FDOT2 __builtin_amdgcn_fdot2
#else
#define VLB_FDOT2 __builtin_amdgcn_fdot2_f32_bf16
#endif

template <int OFF>
__device__ __forceinline__ void tn8_tr_read(s16x4& dst, uint32_t vaddr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vaddr), "n"(OFF));
}

__device__ __forceinline__ void tn8_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ bf16x8 tn8_frag(s16x4 lo, s16x4 hi) {
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// One weight gradient of a grouped launch.  Up to TN8_MAX_GROUP gradients (the four Linear layers of a transformer block) share ONE
// launch: their tiles x K slices form one item list, so the K slices can be few (2 for a block at batch 256: 108 tiles x 2 = 216
// items for 256 CUs) -- the fp32 slab traffic of slab split-K shrinks by the same factor -- and one launch replaces four.
struct Tn8Desc {
  const bf16_t* A; const bf16_t* B;     // dY [R, Mo], X [R, No]
  float* C;                             // output (direct, or the slab of slice 0)
  float* colsum;                        // [Mo] += column sums of A, or null
  long c_split_stride;                  // floats between consecutive slices' slabs
  int lda, ldb, ldc;
  int Mo, No, R;
  int ntm, ntn, tile_group;
  int nsplit, kt_per_split;             // K slices and 64-row K tiles per slice (even)
  int item0, nitems;                    // position in the launch's item list
  int short0;                           // uneven mode: position of this gradient's remainder slices in the short list
  int accumulate;                       // direct output (nsplit == 1): C += instead of C =
  const float* rowscale;                // direct output: row m of the product is multiplied by rowscale[m] first (frozen-BN fold), or null
};
constexpr int TN8_MAX_GROUP = 4;
// Uneven mode (nlong > 0): with T tiles and 2 T <= 256 < 3 T, two equal K slices leave 256 - 2 T CUs idle (216 items for 256 CUs at
// the encoder's shapes: 16 % of the chip).  Every tile's K range is then cut in THREE: two long slices of kt_per_split K tiles --
// the first nlong = 2 T items, one per workgroup -- and a short remainder; the T remainders are shared by the other workgroups
// (k = ceil(T / (grid - nlong)) each).  With L chosen so that L ~ k x remainder the makespan drops from 101 to 87 K-tile pairs at
// batch 256 (-14 %) for one more slab per tile.  OFF by default: on the GPU the launch is not faster (the 40 idle CUs leave the
// other 216 more fabric bandwidth and clock; the third slab costs 57 MB of extra traffic per layer) -- see DESIGN.md.
struct Tn8Group {
  Tn8Desc d[TN8_MAX_GROUP];
  int n, nitems;
  int nlong;
  unsigned long long* stamps;           // measurement only (tools/tn8_probe.py): workgroup b leaves {cycles, 100 MHz ticks} x {entry, exit}
};

// TABLE = false: up to TN8_MAX_GROUP descriptors travel in the kernel argument (the encoder's per-layer group).  TABLE = true (round 4):
// any number of descriptors in a device table `tab` (grp.n / grp.nitems valid, grp.d unused) -- the deferred weight gradients of a
// whole ResNet stage (vision.py): ~46 small products that would each be a 100-300-tile launch + a slab reduce become one launch of
// full-K 256 x 256 items; the descriptor of an item is found by binary search over item0.
// M32 (round 6): the quadrant's products as 8 x v_mfma_f32_32x32x16 instead of 16 x v_mfma_f32_16x16x32 -- half the matrix
// instructions and half the operand-register reads per FLOP (the chip is power-limited under these kernels), and the 32x32 form's
// issue ceiling is 15 % above the 16x16 form's (2382 vs 2075 TFLOP/s micro-benchmarked).  A wave's 64 x 32 share of a quadrant is two
// 32 x 32 blocks; a fragment of k-step s (16 reduction rows) is rows 16 s + 8 (lane >> 5) + [0, 8) of column lane & 31, i.e. the
// 16-lane groups of a transpose read take (k half, 16-column half) = (lane >> 5, (lane >> 4) & 1); the 32-B column units are swizzled
// by f(row) = (row & 3) << 1, which spreads the 4 rows x 2 units of a 32-lane read group over all 64 banks.
// Measured (tools/tn8_probe.py, profiles/r06_tn8_probe.txt): 11 % fewer shader cycles per K tile (2900 vs 3270) at an 11 % lower clock
// (1.99 vs 2.23 GHz) -- the chip is power-limited under this kernel and the launch takes the same 355-365 us either way; the whole step
// was 0.1 ms slower with it.  Default: the 16x16x32 form (VLB_GEMM_TN8_M32=1 / option "tn8_m32" selects this one).
// ABL (measurement builds only, -DVLB_TN8_PROBE): bit 0 no LDS-DMA, bit 1 no fragment reads, bit 2 no MFMAs, bit 3 every workgroup
// streams the operand panels of work item 0 (an all-L2-hit operand stream) -- results WRONG.
template <bool TABLE, bool M32, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_tn8_kernel(const Tn8Group grp, const Tn8Desc* __restrict__ tab) {
  auto D = [&](int gi) -> Tn8Desc {
    if constexpr (TABLE) return tab[gi];
    else return grp.d[gi];
  };
  constexpr int HB = 64 * 256;                  // bytes of a half-image: 64 reduction rows x 128 columns
  constexpr int OFF_A0 = 0, OFF_A1 = HB, OFF_B0 = 2 * HB, OFF_B1 = 3 * HB;
  constexpr int BUF = 4 * HB;                   // one K tile = 64 KiB
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nitems = grp.nitems;
  if ((int)blockIdx.x >= nitems) return;
  if (grp.stamps && tid == 0) {
    unsigned long long* t = grp.stamps + 4 * blockIdx.x;
    t[0] = __builtin_readcyclecounter();
    t[1] = __builtin_amdgcn_s_memrealtime();
  }

  // item -> (gradient g, tile origin, K-tile range): per gradient slice-major (all tiles of a slice are concurrent and share their
  // operand rows in L2); inside a slice the tiles are walked in groups of `tile_group` tile rows, column-major inside a group
  // Work item w (block b takes w = b, b + grid, ...; block b runs on XCD b % 8) -> position t in the item list: every XCD owns a
  // CONTIGUOUS run of the list, i.e. tiles of the same gradient and K slice that share operand panels in that XCD's L2.  (The plain
  // w -> t = w order spread neighbouring tiles over all 8 L2s: the profile showed 3.1x the algorithmic bytes on the fabric,
  // 6.6 TB/s -- the kernel was memory-bound.)
  const int nlong = grp.nlong;
  auto xcd_order = [](int w, int n) {      // position of work item w (w & 7 = its XCD) when every XCD owns a contiguous run of n items
    const int xcd = w & 7, q = n >> 3, r = n & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (w >> 3);
  };
  auto item_of = [&](int w, int& gi, int& m0, int& n0, int& kt0, int& nk) {
    int sp, t;
    gi = 0;
    if (nlong > 0 && w >= nlong) {          // a remainder slice (uneven mode): short list, tile-major per gradient
      const int vs = xcd_order(w - nlong, nitems - nlong);
#pragma unroll
      for (int q = 1; q < TN8_MAX_GROUP; ++q)
        if (q < grp.n && vs >= grp.d[q].short0) gi = q;
      sp = 2;
      t = vs - grp.d[gi].short0;
    } else {
      w = xcd_order(w, nlong > 0 ? nlong : nitems);
      if constexpr (TABLE) {
        int lo = 0, hi = grp.n;
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (w >= tab[mid].item0) lo = mid; else hi = mid;
        }
        gi = lo;
      } else {
#pragma unroll
        for (int q = 1; q < TN8_MAX_GROUP; ++q)
          if (q < grp.n && w >= grp.d[q].item0) gi = q;
      }
      const Tn8Desc dd = D(gi);
      const int wl = w - dd.item0, ntile = dd.ntm * dd.ntn;
      sp = wl / ntile;
      t = wl - sp * ntile;
    }
    const Tn8Desc d = D(gi);
    const int gm = d.tile_group, per_group = gm * d.ntn, gid = t / per_group, first = gid * gm;
    const int gsz = min(d.ntm - first, gm), rem = t - gid * per_group;
    m0 = (first + rem % gsz) * 256;
    n0 = (rem / gsz) * 256;
    kt0 = sp * d.kt_per_split;
    nk = min((d.R >> 6) - kt0, d.kt_per_split);       // even, >= 2 (host guarantees)
  };
  // this workgroup's walk through the item list: w = b, b + step, ...  (uneven mode: a long item is the only one of its workgroup,
  // the remainders are dealt round-robin to the workgroups behind the long ones)
  const int step = (nlong > 0) ? ((int)blockIdx.x < nlong ? (1 << 30) : (int)gridDim.x - nlong) : (int)gridDim.x;

  // ---------------- producer ----------------
  // chunk P = it*512 + tid of a half-image: row = P >> 4 = it*32 + (tid >> 4), physical 16-B chunk c = P & 15; physical 32-B block
  // c >> 1 holds logical block (c >> 1) ^ f(row), f(row) = (row & 3) | ((row >> 3) & 1) << 2 -- independent of `it`
  const int srow = tid >> 4, sc = tid & 15;
  const int sf = M32 ? ((srow & 3) << 1) : ((srow & 3) | (((srow >> 3) & 1) << 2));
  const int lc = ((((sc >> 1) ^ sf) << 1) | (sc & 1)) * 8;      // logical column offset inside the 128-wide half-image
  __amdgpu_buffer_rsrc_t rsA, rsB;
  int ldaB = 0, ldbB = 0, clampA = 0, clampB = 0, rowA = 0, rowB = 0;
  int w_p = blockIdx.x, kt_p = 0, nk_p = 0, kt0_p = 0, pm0 = 0, pn0 = 0;
  auto setup = [&](int w) {
    int gi;
    item_of(w, gi, pm0, pn0, kt0_p, nk_p);
    if constexpr ((ABL & 8) != 0) { gi = 0; pm0 = 0; pn0 = 0; kt0_p = 0; }      // every workgroup streams the SAME panels: L2 hits only
    const Tn8Desc d = D(gi);
    rsA = __builtin_amdgcn_make_buffer_rsrc((void*)d.A, 0, 0x7FFFFFFF, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc((void*)d.B, 0, 0x7FFFFFFF, 0x00020000);
    ldaB = d.lda * 2; ldbB = d.ldb * 2;
    clampA = d.lda - 8; clampB = d.ldb - 8;
    rowA = srow * ldaB; rowB = srow * ldbB;
  };
  auto stage = [&](auto which_c, auto buf_c) {
    constexpr int WHICH = decltype(which_c)::value, B_ = decltype(buf_c)::value;
    // The producer never stops (round 6): past the end of its work it keeps re-staging the last K tile of its last item into the slots
    // the schedule frees anyway (nothing reads them; the per-item drain retires them before the workgroup exits).  Through round 5 a
    // `live` flag put a branch in front of every phase's load segment and a two-way wait into phase 4; the vendor's 4-wave kernel has
    // no branch in its K loop (profiles/r06_tn8_vs_vendor.txt).  Measured: 380 -> 365 us for the layer group, step -0.2 ms.
    if constexpr ((ABL & 1) != 0) return;
    const int kt = kt0_p + kt_p;
    if constexpr (WHICH < 2) {
      char* dst = smem + B_ * BUF + (WHICH == 0 ? OFF_A0 : OFF_A1);
      // columns beyond the operand are clamped (their products land in output rows the epilogue masks)
      const int vo = rowA + 2 * min(pm0 + WHICH * 128 + lc, clampA);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(dst + (it * 512 + wave * 64) * 16), 16, vo, (kt * 64 + it * 32) * ldaB, 0, 0);
    } else {
      char* dst = smem + B_ * BUF + (WHICH == 2 ? OFF_B0 : OFF_B1);
      const int vo = rowB + 2 * min(pn0 + (WHICH - 2) * 128 + lc, clampB);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(dst + (it * 512 + wave * 64) * 16), 16, vo, (kt * 64 + it * 32) * ldbB, 0, 0);
    }
  };
  auto advance = [&]() {
    if (++kt_p == nk_p) {
      const int w_next = (step >= (1 << 30)) ? nitems : w_p + step;
      if (w_next < nitems) {
        kt_p = 0;
        w_p = w_next;
        setup(w_p);
      } else {
        kt_p = nk_p - 1;      // out of work: stay on the last K tile
      }
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using W0 = std::integral_constant<int, 0>;      // staging order A0 B1 A1 B0 (B0 is re-read for the 4th quadrant)
  using W1 = std::integral_constant<int, 3>;
  using W2 = std::integral_constant<int, 1>;
  using W3 = std::integral_constant<int, 2>;

  // ---------------- consumer ----------------
  // fragments: A index i * KS + s (i: 16- / 32-column block of the wave's 64 A columns, s: k-step), B index j * KS + s
  constexpr int NI = M32 ? 2 : 4, NJ = M32 ? 1 : 2, KS = M32 ? 4 : 2, KROWS = M32 ? 16 : 32;
  typedef __attribute__((ext_vector_type(16))) float f32x16;
  typedef typename std::conditional<M32, f32x16, f32x4>::type acc_t;
  acc_t acc[2 * NI][2 * NJ];
  s16x4 alo[NI * KS], ahi[NI * KS], blo[NJ * KS], bhi[NJ * KS];
  float csum[2 * NI];
  const int L = lane & 15, g = lane >> 4;
  const int kh = M32 ? (g >> 1) : g;           // which 8-row group of a k-step this lane's fragment elements come from
  const int e16 = M32 ? (g & 1) : 0;           // M32: which 16-column half of the 32-column block
  const int fl = M32 ? ((L >> 2) << 1) : ((L >> 2) | ((g & 1) << 2));
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const uint32_t lane_off = lds0 + (uint32_t)((8 * kh + (L >> 2)) * 256 + (L & 3) * 8);   // + KROWS*256*s, + 4*256 for the high half
  uint32_t a_fo[NI], b_fo[NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
    a_fo[i] = lane_off + (uint32_t)(((M32 ? 2 * (wm * 2 + i) + e16 : wm * 4 + i) ^ fl) << 5);
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    b_fo[j] = lane_off + (uint32_t)(OFF_B0 + (((M32 ? 2 * wn + e16 : wn * 2 + j) ^ fl) << 5));

  auto read_a = [&](auto buf_c, auto half_c) {
    if constexpr ((ABL & 2) != 0) return;
    constexpr int O = decltype(buf_c)::value * BUF + decltype(half_c)::value * HB;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const uint32_t v = a_fo[i] + (uint32_t)O;
      tn8_tr_read<0>(alo[i * KS + 0], v);
      tn8_tr_read<4 * 256>(ahi[i * KS + 0], v);
      tn8_tr_read<KROWS * 256>(alo[i * KS + 1], v);
      tn8_tr_read<KROWS * 256 + 4 * 256>(ahi[i * KS + 1], v);
      if constexpr (M32) {
        tn8_tr_read<2 * KROWS * 256>(alo[i * KS + 2], v);
        tn8_tr_read<2 * KROWS * 256 + 4 * 256>(ahi[i * KS + 2], v);
        tn8_tr_read<3 * KROWS * 256>(alo[i * KS + 3], v);
        tn8_tr_read<3 * KROWS * 256 + 4 * 256>(ahi[i * KS + 3], v);
      }
    }
  };
  auto read_b = [&](auto buf_c, auto half_c) {
    if constexpr ((ABL & 2) != 0) return;
    constexpr int O = decltype(buf_c)::value * BUF + decltype(half_c)::value * HB;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint32_t v = b_fo[j] + (uint32_t)O;
      tn8_tr_read<0>(blo[j * KS + 0], v);
      tn8_tr_read<4 * 256>(bhi[j * KS + 0], v);
      tn8_tr_read<KROWS * 256>(blo[j * KS + 1], v);
      tn8_tr_read<KROWS * 256 + 4 * 256>(bhi[j * KS + 1], v);
      if constexpr (M32) {
        tn8_tr_read<2 * KROWS * 256>(blo[j * KS + 2], v);
        tn8_tr_read<2 * KROWS * 256 + 4 * 256>(bhi[j * KS + 2], v);
        tn8_tr_read<3 * KROWS * 256>(blo[j * KS + 3], v);
        tn8_tr_read<3 * KROWS * 256 + 4 * 256>(bhi[j * KS + 3], v);
      }
    }
  };
  bool do_colsum = false;
  auto compute = [&](auto ha_c, auto hb_c, auto cs_c) {
    constexpr int HA = decltype(ha_c)::value, HB_ = decltype(hb_c)::value;
    constexpr bool CS = decltype(cs_c)::value;      // this phase holds freshly read A fragments: column sums are taken here
    // (both forms hold 8 A and 4 B fragment register pairs x {lo, hi})
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(alo[0]), "+v"(ahi[0]), "+v"(alo[1]), "+v"(ahi[1]), "+v"(alo[2]), "+v"(ahi[2]), "+v"(alo[3]), "+v"(ahi[3]),
                   "+v"(alo[4]), "+v"(ahi[4]), "+v"(alo[5]), "+v"(ahi[5]), "+v"(alo[6]), "+v"(ahi[6]), "+v"(alo[7]), "+v"(ahi[7]),
                   "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]), "+v"(bhi[3]));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((ABL & 4) == 0) {
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if constexpr (M32)
              acc[HA * NI + i][HB_ * NJ + j] = VLB_MFMA_32x32x16(tn8_frag(blo[j * KS + s], bhi[j * KS + s]), tn8_frag(alo[i * KS + s], ahi[i * KS + s]),
                                                                 acc[HA * NI + i][HB_ * NJ + j], 0, 0, 0);
            else
              acc[HA * NI + i][HB_ * NJ + j] = VLB_MFMA_16x16x32(tn8_frag(blo[j * KS + s], bhi[j * KS + s]), tn8_frag(alo[i * KS + s], ahi[i * KS + s]),
                                                                 acc[HA * NI + i][HB_ * NJ + j], 0, 0, 0);
          }
      __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (CS) {
      if (do_colsum) {      // wave-uniform; 32 v_dot2c behind the MFMAs of this quadrant
        const bf16x2_t one = {(vlb_h16)1.0f, (vlb_h16)1.0f};
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const uint2 l = __builtin_bit_cast(uint2, alo[i * KS + ks]), h = __builtin_bit_cast(uint2, ahi[i * KS + ks]);
            float s = csum[HA * NI + i];
            s = VLB_FDOT2(__builtin_bit_cast(bf16x2_t, l.x), one, s, false);
            s = VLB_FDOT2(__builtin_bit_cast(bf16x2_t, l.y), one, s, false);
            s = VLB_FDOT2(__builtin_bit_cast(bf16x2_t, h.x), one, s, false);
            s = VLB_FDOT2(__builtin_bit_cast(bf16x2_t, h.y), one, s, false);
            csum[HA * NI + i] = s;
          }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  using T = std::true_type;
  using F = std::false_type;

  // ---------------- prologue ----------------
  setup(w_p);
  stage(W0{}, I0{}); stage(W1{}, I0{}); stage(W2{}, I0{}); stage(W3{}, I0{});
  advance();
  stage(W0{}, I1{}); stage(W1{}, I1{});
  __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
  tn8_barrier();
  if (wm == 1) tn8_barrier();

  int gk = 0;      // K tiles consumed by this workgroup so far
  for (int w = blockIdx.x; w < nitems; w = (step >= (1 << 30)) ? nitems : w + step) {
    int gi, m0, n0, kt0, nk;
    item_of(w, gi, m0, n0, kt0, nk);
    float* const colsum = D(gi).colsum;
    // column sums: this wave's turn comes every cs_slots K-tile pairs (see the header)
    const int cs_slots = D(gi).ntn * 4;
    int cs_wait = ((n0 >> 8) * 4 + wn - (kt0 >> 1)) % cs_slots;
    if (cs_wait < 0) cs_wait += cs_slots;
#pragma unroll
    for (int i = 0; i < 2 * NI; ++i) {
      csum[i] = 0.f;
#pragma unroll
      for (int j = 0; j < 2 * NJ; ++j) {
        if constexpr (M32) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        }
      }
    }
    for (int kt = 0; kt < nk; kt += 2, gk += 2) {
      do_colsum = (colsum != nullptr) && (cs_wait == 0);
      cs_wait = (cs_wait == 0 ? cs_slots : cs_wait) - 1;
      // ======== K tile gk (buffer 0): quadrants (0,0) (0,1) (1,1) (1,0) ========
      read_b(I0{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      read_a(I0{}, I0{});
      stage(W2{}, I1{});
      tn8_barrier();
      compute(I0{}, I0{}, T{});
      tn8_barrier();
      read_b(I0{}, I1{});
      stage(W3{}, I1{});
      tn8_barrier();
      compute(I0{}, I1{}, F{});
      tn8_barrier();
      read_a(I0{}, I1{});
      advance();
      stage(W0{}, I0{});
      tn8_barrier();
      compute(I1{}, I1{}, T{});
      tn8_barrier();
      read_b(I0{}, I0{});
      stage(W1{}, I0{});
      __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
      tn8_barrier();
      compute(I1{}, I0{}, F{});
      tn8_barrier();
      // ======== K tile gk+1 (buffer 1) ========
      read_b(I1{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      read_a(I1{}, I0{});
      stage(W2{}, I0{});
      tn8_barrier();
      compute(I0{}, I0{}, T{});
      tn8_barrier();
      read_b(I1{}, I1{});
      stage(W3{}, I0{});
      tn8_barrier();
      compute(I0{}, I1{}, F{});
      tn8_barrier();
      read_a(I1{}, I1{});
      advance();
      stage(W0{}, I1{});
      tn8_barrier();
      compute(I1{}, I1{}, T{});
      tn8_barrier();
      read_b(I1{}, I0{});
      stage(W1{}, I1{});
      __builtin_amdgcn_s_waitcnt(0x0F70 | 4);
      tn8_barrier();
      compute(I1{}, I0{}, F{});
      tn8_barrier();
    }
    // ---------------- epilogue: fp32 fragments straight to the slab slice / C (16 B per lane) ----------------
    // (no LDS involved: the operand stream of the next item keeps landing; the wave groups need no re-alignment)
    {
      const Tn8Desc d = D(gi);
      const int sp = kt0 / d.kt_per_split;      // kt0 = slice x kt_per_split in both item lists
      float* const C = d.C + (long)sp * d.c_split_stride;
      const bool accum = d.accumulate != 0;
      const int Mo = d.Mo, No = d.No;
      const long ldc = d.ldc;
      const float* const rowscale = d.rowscale;
      // output fragment (fi, cj, q): row m (one per lane), 4 consecutive columns from n
      //   16x16 form: fi < 8 (tile half x 16-row fragment), cj < 4, q = 0:      m = .. + (fi & 3) * 16 + L,       n = .. + (cj & 1) * 16 + 4 g
      //   32x32 form: fi < 4 (tile half x 32-row block),    cj < 2, q < 4:      m = .. + (fi & 1) * 32 + lane&31, n = .. + 8 q + 4 (lane >> 5)
      constexpr int NQ = M32 ? 4 : 1;
#pragma unroll
      for (int fi = 0; fi < 2 * NI; ++fi) {
        const int m = M32 ? m0 + (fi >> 1) * 128 + wm * 64 + (fi & 1) * 32 + (lane & 31) : m0 + (fi >> 2) * 128 + wm * 64 + (fi & 3) * 16 + L;
        if (m >= Mo) continue;
        const float rs = rowscale ? rowscale[m] : 1.0f;
#pragma unroll
        for (int cj = 0; cj < 2 * NJ; ++cj) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) {
            const int n = M32 ? n0 + cj * 128 + wn * 32 + 8 * q + 4 * (lane >> 5) : n0 + (cj >> 1) * 128 + wn * 32 + (cj & 1) * 16 + 4 * g;
            if (n >= No) continue;
            float* c = C + (long)m * ldc + n;
            f32x4 v = (f32x4){acc[fi][cj][4 * q + 0], acc[fi][cj][4 * q + 1], acc[fi][cj][4 * q + 2], acc[fi][cj][4 * q + 3]};
            if (rowscale) v = (f32x4){v[0] * rs, v[1] * rs, v[2] * rs, v[3] * rs};
            if (n + 3 < No) {
              float4 o4 = make_float4(v[0], v[1], v[2], v[3]);
              if (accum) {
                const float4 o = *(const float4*)c;
                o4 = make_float4(o.x + v[0], o.y + v[1], o.z + v[2], o.w + v[3]);
              }
              if (d.c_split_stride == 0) vlb_store_nt((float4*)c, o4);      // the gradient itself (next read by the optimizer): streaming store
              else *(float4*)c = o4;                                          // a K-slice slab: re-read by the reduce kernel right away
            } else {
              for (int r = 0; r < 4 && n + r < No; ++r) c[r] = accum ? c[r] + v[r] : v[r];
            }
          }
        }
      }
      if (colsum != nullptr) {      // per-lane partial sums cover 8 of a k-step's rows: reduce over the lanes that hold the other rows of the same column
#pragma unroll
        for (int fi = 0; fi < 2 * NI; ++fi) {
          float v = csum[fi];
          if constexpr (M32) {
            v += __shfl_xor(v, 32, 64);
          } else {
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
          }
          const int m = M32 ? m0 + (fi >> 1) * 128 + wm * 64 + (fi & 1) * 32 + (lane & 31) : m0 + (fi >> 2) * 128 + wm * 64 + (fi & 3) * 16 + L;
          if (kh == 0 && m < D(gi).Mo) atomicAdd(colsum + m, v);
        }
      }
    }
    // stores / atomics and LDS-DMA share vmcnt: one full drain per item keeps the counted waits of the next item sound
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  if (wm == 0) tn8_barrier();      // pairs with the extra barrier the lagging wave group took at the start
  if (grp.stamps && tid == 0) {
    unsigned long long* t = grp.stamps + 4 * blockIdx.x;
    t[2] = __builtin_readcyclecounter();
    t[3] = __builtin_amdgcn_s_memrealtime();
  }
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

static int g_tn8_mode = -1;      // VLB_GEMM_TN8 (0: 128x128 TN kernel only); run-time override: vlb_gemm_set_option("tn8_mode", v)
void vlb_tn8_set_mode(int v) { g_tn8_mode = v; }
static int g_tn8_uneven = -1;   // VLB_GEMM_TN8_UNEVEN: 1 = the uneven three-slice cut of grouped launches; 0 (default) equal slices --
                                 // measured: no gain at batch 256 (21.07 vs 21.02 ms / step), 5-10 % slower launches at batch 64 / 32
static int g_tn8_wgs = -1;       // VLB_GEMM_TN8_WGS: persistent workgroups per launch (default 256 = one per CU)
void vlb_tn8_set_wgs(int v) { g_tn8_wgs = v; }
void vlb_tn8_set_uneven(int v) { g_tn8_uneven = v; }
static int g_tn8_m32 = -1;       // VLB_GEMM_TN8_M32: 1 = v_mfma_f32_32x32x16 form of the quadrant products, 0 (default) 16x16x32 (run-time: "tn8_m32")
void vlb_tn8_set_m32(int v) { g_tn8_m32 = v; }
static int g_tn8_ablate = 0;     // measurement builds (-DVLB_TN8_PROBE) only: see the kernel's ABL parameter
void vlb_tn8_set_ablate(int v) { g_tn8_ablate = v; }
static unsigned long long* g_tn8_stamps = nullptr;      // tools/tn8_probe.py: 4 x uint64 per workgroup, or null
extern "C" int vlb_tn8_set_stamps(void* dev) {
  g_tn8_stamps = (unsigned long long*)dev;
  return VLB_OK;
}

// K slices for a [Mo, No] gradient over R rows: fill the 256 CUs (one workgroup each) in whole rounds; a slice is a whole number of
// 128-row units.  Cost model: rounds x (slice length + fixed cost per item) + slab traffic.
int vlb_tn8_pick_splits(int Mo, int No, int R) {
  const long tiles = (long)vlb_cdiv(Mo, 256) * vlb_cdiv(No, 256);
  const int pairs = R / 128;
  const long ldw = (No + 3) / 4 * 4;
  double best = -1.0;
  int best_sp = 1;
  for (int sp = 1; sp <= pairs && sp <= 64; ++sp) {
    const long items = tiles * sp;
    const long rounds = (items + 255) / 256;
    const int per = vlb_cdiv(pairs, sp);
    if ((long)vlb_cdiv(pairs, per) != sp) continue;               // (sp must be realisable with equal slices)
    const double t = (double)rounds * (per * 2 * 1.6 + 6.0) + (sp > 1 ? 2.0 * sp * Mo * (double)ldw * 4.0 / 4.0e6 : 0.0);
    if (best < 0 || t < best) { best = t; best_sp = sp; }
  }
  return best_sp;
}

static int tile_group_for(int ntm, int ntn) {
  static const int group = env_int("VLB_GEMM_TN8_GROUP", 0);
  int gm = group;
  if (gm <= 0) {
    gm = 1;
    if (2 * ntn >= ntm) gm = (int)(sqrt((double)ntm * ntn / 8.0) + 0.5);
  }
  if (gm > ntm) gm = ntm;
  return gm < 1 ? 1 : gm;
}

template <bool TABLE, bool M32, int ABL>
static int tn8_launch_k(Tn8Group& grp, const Tn8Desc* tab, int gx, hipStream_t stream) {
  constexpr int smem = 131072;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_tn8_kernel<TABLE, M32, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) {
      vlb_set_error("gemm_tn8: cannot reserve %d bytes of LDS: %s", smem, hipGetErrorString(e));
      return VLB_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_tn8_kernel<TABLE, M32, ABL>), dim3(gx), dim3(512), smem, stream, grp, tab);
  VLB_CHECK_LAUNCH("vlb_wgrad_tn_bf16(tn8)");
  return VLB_OK;
}

template <bool TABLE>
static int tn8_launch_t(Tn8Group& grp, const Tn8Desc* tab, hipStream_t stream) {
  if (g_tn8_wgs < 0) g_tn8_wgs = env_int("VLB_GEMM_TN8_WGS", 256);
  if (g_tn8_m32 < 0) g_tn8_m32 = env_int("VLB_GEMM_TN8_M32", 0);
  const int cap = (g_tn8_wgs >= 8 && g_tn8_wgs <= 256) ? g_tn8_wgs : 256;
  const int gx = grp.nitems > cap ? cap : grp.nitems;
  grp.stamps = g_tn8_stamps;
#ifdef VLB_TN8_PROBE
  if (!TABLE && g_tn8_ablate) {
#define TN8_ABL(M, A) \
  if ((g_tn8_m32 != 0) == (M != 0) && (g_tn8_ablate & 15) == A) return tn8_launch_k<false, M != 0, A>(grp, tab, gx, stream);
#define TN8_ABLS(M) TN8_ABL(M, 1) TN8_ABL(M, 2) TN8_ABL(M, 3) TN8_ABL(M, 4) TN8_ABL(M, 5) TN8_ABL(M, 6) TN8_ABL(M, 8) TN8_ABL(M, 12) TN8_ABL(M, 14)
    TN8_ABLS(0) TN8_ABLS(1)
#undef TN8_ABLS
#undef TN8_ABL
  }
#endif
  return g_tn8_m32 ? tn8_launch_k<TABLE, true, 0>(grp, tab, gx, stream) : tn8_launch_k<TABLE, false, 0>(grp, tab, gx, stream);
}

static int tn8_launch(Tn8Group& grp, hipStream_t stream) { return tn8_launch_t<false>(grp, nullptr, stream); }

static bool tn8_shape_ok(long lda, long ldb, long ldc, int R) {
  if ((R % 128) != 0 || R < 256) return false;
  if ((lda % 8) || (ldb % 8) || lda < 8 || ldb < 8 || (ldc % 4)) return false;
  if ((long)R * lda * 2 >= (1L << 31) || (long)R * ldb * 2 >= (1L << 31)) return false;
  return true;
}

// Weight gradient through the large-tile core.  Returns the number of K slices used (>= 1; the caller then runs the slab reduce when
// it is > 1 or a row scale is pending), 0 when the shape is outside what this kernel covers, < 0 on error.
// p: A = dY [R, Mo], B = X [R, No], M = Mo, N = No, K = R.
int vlb_gemm_tn8_try(GemmParams& p, float* C, long ldc, float* colsum, float* workspace, long workspace_floats, int accumulate,
                     bool force_slab, hipStream_t stream) {
  if (g_tn8_mode < 0) g_tn8_mode = env_int("VLB_GEMM_TN8", 1);
  if (!g_tn8_mode) return 0;
  const int R = p.K, Mo = p.M, No = p.N;
  if (!tn8_shape_ok(p.lda, p.ldb, ldc, R)) return 0;
  const int ntm = vlb_cdiv(Mo, 256), ntn = vlb_cdiv(No, 256);
  const long tiles = (long)ntm * ntn;
  const int pairs = R / 128;                       // 128-row units of the reduction
  const long ldw = (No + 3) / 4 * 4;
  int splits = vlb_tn8_pick_splits(Mo, No, R);
  if (tiles * splits < 128) return 0;              // cannot fill the chip with 256x256 tiles: the 128x128 kernel covers it
  const int per = vlb_cdiv(pairs, splits);
  splits = vlb_cdiv(pairs, per);
  const bool slab = splits > 1 || force_slab;
  if (slab && (!workspace || workspace_floats < (long)splits * Mo * ldw)) return 0;
  Tn8Group grp = {};
  Tn8Desc& d = grp.d[0];
  d.A = p.A; d.B = p.B; d.colsum = colsum;
  d.lda = (int)p.lda; d.ldb = (int)p.ldb; d.Mo = Mo; d.No = No; d.R = R;
  d.ntm = ntm; d.ntn = ntn; d.tile_group = tile_group_for(ntm, ntn);
  d.nsplit = splits; d.kt_per_split = per * 2; d.item0 = 0; d.nitems = (int)(tiles * splits);
  if (slab) { d.C = workspace; d.ldc = (int)ldw; d.c_split_stride = (long)Mo * ldw; d.accumulate = 0; }
  else { d.C = C; d.ldc = (int)ldc; d.c_split_stride = 0; d.accumulate = accumulate ? 1 : 0; }
  grp.n = 1; grp.nitems = d.nitems;
  const int rc = tn8_launch(grp, stream);
  return rc < 0 ? rc : splits;
}

// Grouped form: n <= 4 gradients over the SAME number of rows R in one launch.  slices[i] (out) = K slices used for gradient i: when
// > 1 its partial tiles lie in `workspace` at float offset ws_off[i] (slice stride Mo[i] * round4(No[i])) and the caller reduces
// them; 1: written (or accumulated) straight into C[i].  Returns 1 when launched, 0 when the group is outside what the kernel covers
// (nothing launched), < 0 on error.
int vlb_gemm_tn8_group(int n, const void* const* A, const long* lda, const void* const* B, const long* ldb, float* const* C,
                       const long* ldc, int R, const int* Mo, const int* No, float* const* colsum, float* workspace,
                       long workspace_floats, int accumulate, int* slices, long* ws_off, hipStream_t stream) {
  if (g_tn8_mode < 0) g_tn8_mode = env_int("VLB_GEMM_TN8", 1);
  if (!g_tn8_mode || n < 1 || n > TN8_MAX_GROUP) return 0;
  long tiles = 0;
  for (int i = 0; i < n; ++i) {
    if (!tn8_shape_ok(lda[i], ldb[i], ldc[i], R)) return 0;
    tiles += (long)vlb_cdiv(Mo[i], 256) * vlb_cdiv(No[i], 256);
  }
  const int pairs = R / 128;
  // one slice count for the whole group (equal item lengths): as many slices as keep the item list within one round of 256 CUs
  int splits = (int)(256 / tiles);
  if (splits < 1) splits = 1;
  if (splits > pairs) splits = pairs;
  if (tiles * splits < 128) return 0;
  int per = vlb_cdiv(pairs, splits);
  splits = vlb_cdiv(pairs, per);
  // uneven three-way cut (see Tn8Group): two long slices of L pairs per tile, one workgroup each, and a remainder of pairs - 2 L
  // shared by the 256 - 2 T spare workgroups, k = ceil(T / spare) remainders each; L minimises max(L, k x remainder)
  if (g_tn8_uneven < 0) g_tn8_uneven = env_int("VLB_GEMM_TN8_UNEVEN", 0);
  int nlong = 0;
  long slab_floats = 0;
  for (int i = 0; i < n; ++i) slab_floats += (long)Mo[i] * ((No[i] + 3) / 4 * 4);
  if (g_tn8_wgs < 0) g_tn8_wgs = env_int("VLB_GEMM_TN8_WGS", 256);
  if (g_tn8_uneven && splits == 2 && 2 * tiles < 256 && workspace && 3 * slab_floats <= workspace_floats && g_tn8_wgs == 256) {
    const int spare = 256 - 2 * (int)tiles, k = vlb_cdiv((int)tiles, spare);
    const int L = vlb_cdiv(k * pairs, 2 * k + 1), rem = pairs - 2 * L;
    const int makespan = L > k * rem ? L : k * rem;
    if (rem >= 1 && L >= 1 && makespan * 100 <= per * 95) {      // worth a third slab per tile only for >= 5 %
      nlong = 2 * (int)tiles;
      per = L;
      splits = 3;
    }
  }
  Tn8Group grp = {};
  long off = 0;
  int item = 0, sitem = 0;
  for (int i = 0; i < n; ++i) {
    Tn8Desc& d = grp.d[i];
    const long ldw = (No[i] + 3) / 4 * 4;
    d.A = (const bf16_t*)A[i]; d.B = (const bf16_t*)B[i]; d.colsum = colsum ? colsum[i] : nullptr;
    d.lda = (int)lda[i]; d.ldb = (int)ldb[i]; d.Mo = Mo[i]; d.No = No[i]; d.R = R;
    d.ntm = vlb_cdiv(Mo[i], 256); d.ntn = vlb_cdiv(No[i], 256); d.tile_group = tile_group_for(d.ntm, d.ntn);
    d.nsplit = splits; d.kt_per_split = per * 2;
    const int ntile = d.ntm * d.ntn;
    d.item0 = item; d.nitems = ntile * (nlong ? 2 : splits);      // (uneven: the two long slices; the remainders follow all of them)
    d.short0 = sitem;
    item += d.nitems;
    sitem += ntile;
    if (splits > 1) {
      d.C = workspace + off; d.ldc = (int)ldw; d.c_split_stride = (long)Mo[i] * ldw; d.accumulate = 0;
      ws_off[i] = off;
      off += (long)splits * Mo[i] * ldw;
    } else {
      d.C = C[i]; d.ldc = (int)ldc[i]; d.c_split_stride = 0; d.accumulate = accumulate ? 1 : 0;
      ws_off[i] = 0;
    }
    slices[i] = splits;
  }
  if (nlong) item += sitem;
  grp.nlong = nlong;
  if (splits > 1 && (!workspace || off > workspace_floats)) return 0;
  grp.n = n; grp.nitems = item;
  const int rc = tn8_launch(grp, stream);
  return rc < 0 ? rc : 1;
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Table-driven form (round 4): n weight gradients  C[i] (+)= rowscale[i] . (A[i]^T B[i])  with their own row counts, in ONE launch of
// full-K 256 x 256 items (no K slices, no slabs, no reduce pass).  The descriptor table is built once on the host for a fixed set of
// buffers (vlb_wgrad_tn_table_pack), uploaded by the caller, and replayed every step (vlb_wgrad_tn_table_launch).
// Replaces the per-convolution `dW = dY^T X` products of torch's autograd in the ResNet stages (common/backbone/resnet/resnet.py:98-118).
extern "C" long vlb_wgrad_tn_table_desc_bytes(void) { return (long)sizeof(Tn8Desc); }

// host_out: n * vlb_wgrad_tn_table_desc_bytes() bytes.  Returns the number of work items (> 0), 0 when a product is outside what the
// kernel covers (R % 128, R < 256, unaligned leading dimensions), < 0 on error.
extern "C" int vlb_wgrad_tn_table_pack(int n, const void* const* A, const long* lda, const void* const* B, const long* ldb, float* const* C,
                                       const long* ldc, const int* R, const int* Mo, const int* No, float* const* colsum,
                                       const float* const* rowscale, int accumulate, void* host_out, long host_bytes) {
  VLB_CHECK_ARG(n >= 1 && A && B && C && lda && ldb && ldc && R && Mo && No && host_out, "vlb_wgrad_tn_table_pack: null argument");
  VLB_CHECK_ARG(host_bytes >= (long)n * (long)sizeof(Tn8Desc), "vlb_wgrad_tn_table_pack: output buffer too small");
  Tn8Desc* out = (Tn8Desc*)host_out;
  int item = 0;
  for (int i = 0; i < n; ++i) {
    if (!A[i] || !B[i] || !C[i] || Mo[i] < 1 || No[i] < 1) {
      vlb_set_error("vlb_wgrad_tn_table_pack: bad product %d", i);
      return VLB_ERR_ARG;
    }
    if (!tn8_shape_ok(lda[i], ldb[i], ldc[i], R[i])) return 0;
    Tn8Desc d = {};
    d.A = (const bf16_t*)A[i]; d.B = (const bf16_t*)B[i]; d.C = C[i]; d.colsum = colsum ? colsum[i] : nullptr;
    d.rowscale = rowscale ? rowscale[i] : nullptr;
    d.lda = (int)lda[i]; d.ldb = (int)ldb[i]; d.ldc = (int)ldc[i]; d.Mo = Mo[i]; d.No = No[i]; d.R = R[i];
    d.ntm = vlb_cdiv(Mo[i], 256); d.ntn = vlb_cdiv(No[i], 256); d.tile_group = tile_group_for(d.ntm, d.ntn);
    d.nsplit = 1; d.kt_per_split = R[i] / 64; d.c_split_stride = 0; d.accumulate = accumulate ? 1 : 0;
    d.item0 = item; d.nitems = d.ntm * d.ntn; d.short0 = 0;
    item += d.nitems;
    out[i] = d;
  }
  return item;
}

extern "C" int vlb_wgrad_tn_table_launch(const void* desc_dev, int n, int nitems, hipStream_t stream) {
  VLB_CHECK_ARG(desc_dev && n >= 1 && nitems >= 1, "vlb_wgrad_tn_table_launch: bad argument");
  Tn8Group grp = {};
  grp.n = n; grp.nitems = nitems; grp.nlong = 0;
  return tn8_launch_t<true>(grp, (const Tn8Desc*)desc_dev, stream);
}
