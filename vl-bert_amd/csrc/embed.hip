// Embedding-side kernels of the VL-BERT hot path (gfx950, HBM-bound gather/scatter work):
//   * vlb_seq_layout        - per-sample [text || objects || END || pad] layout from the masks
//   * vlb_obj_prep_fwd      - FastRCNN precomputed branch input: coordinate sin/cos embedding ||
//                             2048-d region feature (or the mask embedding) -> dropout -> bf16
//   * vlb_embed_fwd / _bwd  - VisualLinguisticBert.embedding: word/type/position/visual sum,
//                             seamless concatenation, embedding LayerNorm, dropout
//   * vlb_gather_rows, vlb_head_grad_combine, vlb_relu_bwd_cast, vlb_masked_colsum - glue
// Reference semantics: common/visual_linguistic_bert.py:173-241, common/fast_rcnn.py:136-187,
// common/utils/bbox.py:33-65, pretrain/modules/resnet_vlbert_for_pretraining.py:106-142.
#include "vlb_common.h"

#define EMB_MAX_IT 8  // H <= 2048, 4 elements per lane per iteration

enum { KIND_PAD = 0, KIND_TEXT = 1, KIND_OBJ = 2, KIND_END = 3 };

// ------------------------------------------------------------------------------------------
// layout: one thread block per sample; a single lane walks the (<= ~1k) mask entries.
//   code[b,s] = kind<<16 | source index (t for text, r for object)
//   text_rows[b,t] = b*S+t (VisualLinguisticBert.forward :152-154 slices [:, :T]),
//   obj_rows[b,r]  = row of the r-th object if object_mask[b,r] else -1 (:155-157)
// ------------------------------------------------------------------------------------------
__global__ void seq_layout_kernel(const uint8_t* __restrict__ text_mask, const uint8_t* __restrict__ obj_mask, int B, int T, int R,
                                  int S, int32_t* __restrict__ code, int32_t* __restrict__ text_len, int32_t* __restrict__ nobj,
                                  int32_t* __restrict__ text_rows, int32_t* __restrict__ obj_rows, float* __restrict__ attn_mask) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  int n = 0;
  for (int t = 0; t < T; ++t)
    if (text_mask[b * T + t]) code[b * S + n++] = (KIND_TEXT << 16) | t;
  const int tl = n;
  for (int r = 0; r < R; ++r) {
    if (obj_mask[b * R + r]) {
      obj_rows[b * R + r] = b * S + n;
      code[b * S + n++] = (KIND_OBJ << 16) | r;
    } else {
      obj_rows[b * R + r] = -1;
    }
  }
  const int no = n - tl;
  code[b * S + n++] = (KIND_END << 16);
  for (int s = n; s < S; ++s) code[b * S + s] = (KIND_PAD << 16);
  for (int s = 0; s < S; ++s) attn_mask[b * S + s] = (s < n) ? 1.f : 0.f;
  for (int t = 0; t < T; ++t) text_rows[b * T + t] = b * S + t;
  text_len[b] = tl;
  nobj[b] = no;
}

// ------------------------------------------------------------------------------------------
// obj_prep: out[b*R+r][0:2048] = coordinate_embeddings(box, im w/h) ; [2048:4096] = feature
// (common/fast_rcnn.py:165-175 + common/utils/bbox.py:33-65); mvrc_ops==1 rows take
// object_mask_visual_embedding instead of the precomputed feature
// (resnet_vlbert_for_pretraining.py:114-117); hard-coded Dropout(p) of obj_downsample.
// One 256-thread block per region: 8 coord + 8 feature elements per thread, 16/32-B accesses.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void obj_prep_fwd_kernel(const float* __restrict__ boxes, long ldbox, const float* __restrict__ im_info, long ldinfo,
                                                           const int64_t* __restrict__ mvrc_ops, const float* __restrict__ mask_emb,
                                                           bf16_t* __restrict__ out, int B, int R, uint32_t drop_thr, float drop_scale,
                                                           const uint32_t* __restrict__ seedp, uint32_t tag) {
  const int row = blockIdx.x;  // b*R + r
  const int b = row / R;
  const float* bx = boxes + (long)row * ldbox;
  bf16_t* o = out + (long)row * 4096;
  const int tid = threadIdx.x;
  const uint32_t seed = (drop_thr && seedp) ? *seedp : 0u;
  if (!(bx[0] > -1.5f)) {  // padded box (box_mask false): zero row, never consumed
    *(uint4*)(o + tid * 8) = make_uint4(0, 0, 0, 0);
    *(uint4*)(o + 2048 + tid * 8) = make_uint4(0, 0, 0, 0);
    return;
  }
  const float W = im_info[b * ldinfo + 0], Hh = im_info[b * ldinfo + 1];
  const float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  float pos[4];
  pos[0] = (x1 + x2) / 2 / W * 100;
  pos[1] = (y1 + y2) / 2 / Hh * 100;
  pos[2] = (x2 - x1) / W * 100;
  pos[3] = (y2 - y1) / Hh * 100;
  const bool masked = mvrc_ops && mvrc_ops[row] == 1;
  const float* feat = masked ? mask_emb : (bx + 4);
  // thread t owns elements [4t,4t+4) and [1024+4t, ..+4) of both the coordinate half and the
  // feature half: every wave-level access is a contiguous 1 KiB (fp32 in) / 512 B (bf16 out) run.
#pragma unroll
  for (int part = 0; part < 2; ++part) {
    const int e0 = part * 1024 + tid * 4;  // element within a 2048-wide half
    float cv[4], fv[4];
    {
      // coordinate half: e -> which = e/512, j = e%512, sin for j<256 else cos, frequency index j%256
      const int which = e0 >> 9, j0 = e0 & 511;
      const bool is_cos = j0 >= 256;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dim = powf(1000.0f, (float)((j0 & 255) + k) / 256.0f);
        const float a = pos[which] / dim;
        cv[k] = is_cos ? cosf(a) : sinf(a);
      }
      const float4 f = *(const float4*)(feat + e0);
      fv[0] = f.x; fv[1] = f.y; fv[2] = f.z; fv[3] = f.w;
    }
    if (drop_thr) {
      const uint32_t ic = (uint32_t)row * 4096u + (uint32_t)e0, ifeat = ic + 2048u;
      const uint32_t h0 = vlb_rng_pair(seed, tag, ic >> 1), h1 = vlb_rng_pair(seed, tag, (ic >> 1) + 1);
      const uint32_t g0 = vlb_rng_pair(seed, tag, ifeat >> 1), g1 = vlb_rng_pair(seed, tag, (ifeat >> 1) + 1);
      cv[0] = ((h0 & 0xffffu) >= drop_thr) ? cv[0] * drop_scale : 0.f;
      cv[1] = ((h0 >> 16) >= drop_thr) ? cv[1] * drop_scale : 0.f;
      cv[2] = ((h1 & 0xffffu) >= drop_thr) ? cv[2] * drop_scale : 0.f;
      cv[3] = ((h1 >> 16) >= drop_thr) ? cv[3] * drop_scale : 0.f;
      fv[0] = ((g0 & 0xffffu) >= drop_thr) ? fv[0] * drop_scale : 0.f;
      fv[1] = ((g0 >> 16) >= drop_thr) ? fv[1] * drop_scale : 0.f;
      fv[2] = ((g1 & 0xffffu) >= drop_thr) ? fv[2] * drop_scale : 0.f;
      fv[3] = ((g1 >> 16) >= drop_thr) ? fv[3] * drop_scale : 0.f;
    }
    *(uint2*)(o + e0) = make_uint2(pack2bf(cv[0], cv[1]), pack2bf(cv[2], cv[3]));
    *(uint2*)(o + 2048 + e0) = make_uint2(pack2bf(fv[0], fv[1]), pack2bf(fv[2], fv[3]));
  }
}

// x[r, :] = 0 for the rows of padded boxes (boxes[r*ldbox] <= -1.5): obj_downsample is only applied to valid boxes in the reference and
// pad_sequence fills the rest with zeros (common/fast_rcnn.py:176-186); the GEMM over the zeroed input row leaves relu(bias) there,
// which matters when box 0 of a sample is padding (its row feeds every text token's visual embedding, :132-135).
__global__ __launch_bounds__(256) void zero_padded_rows_kernel(bf16_t* __restrict__ x, long ld, const float* __restrict__ boxes, long ldbox,
                                                               int rows, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows || boxes[(long)row * ldbox] > -1.5f) return;
  for (int c = lane * 8; c < H; c += 512) *(uint4*)(x + (long)row * ld + c) = make_uint4(0, 0, 0, 0);
}

extern "C" int vlb_zero_padded_rows_bf16(void* x, long ld, const float* boxes, long ldbox, int rows, int H, hipStream_t stream) {
  if (rows <= 0) return VLB_OK;
  VLB_CHECK_ARG(x && boxes && H > 0 && (H % 8) == 0 && (ld % 8) == 0, "vlb_zero_padded_rows_bf16: bad argument");
  hipLaunchKernelGGL(zero_padded_rows_kernel, dim3(vlb_cdiv(rows, 4)), dim3(256), 0, stream, (bf16_t*)x, ld, boxes, ldbox, rows, H);
  VLB_CHECK_LAUNCH("vlb_zero_padded_rows_bf16");
  return VLB_OK;
}

// colsum over rows r with sel[r]==1 of src[r][c] * dropmask(row, col_off + c):  dst[c] += ...
// (gradient of object_mask_visual_embedding: sum over masked regions of dA[:, 2048:]).
__global__ __launch_bounds__(256) void masked_colsum_kernel(const bf16_t* __restrict__ src, long lds, const int64_t* __restrict__ sel,
                                                            int rows, int C, float* __restrict__ dst, uint32_t drop_thr,
                                                            float drop_scale, const uint32_t* __restrict__ seedp, uint32_t tag,
                                                            uint32_t row_elems, uint32_t col_off) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const uint32_t seed = (drop_thr && seedp) ? *seedp : 0u;
  float s = 0.f;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    if (sel[r] != 1) continue;
    float v = bf2f(src[(long)r * lds + c]);
    if (drop_thr) v = vlb_keep(seed, tag, (uint32_t)r * row_elems + col_off + (uint32_t)c, drop_thr) ? v * drop_scale : 0.f;
    s += v;
  }
  atomicAdd(dst + c, s);
}

// ------------------------------------------------------------------------------------------
// embedding forward: one wave per output row (b,s).
// ------------------------------------------------------------------------------------------
struct EmbedParams {
  const int32_t* code;        // [B,S]
  const int32_t* text_len;    // [B]
  const int64_t* text_ids;    // [B,T]
  const int64_t* text_type;   // [B,T] or null (all 0)
  const bf16_t* word_emb;     // [V,H]
  const bf16_t* pos_emb;      // [P,H]
  const bf16_t* type_emb;     // [3,H]
  const bf16_t* end_emb;      // [1,H]
  const bf16_t* text_vis; long tv_sb, tv_st;   // visual_ln_text output, strides (elements)
  const bf16_t* obj_vis; long ov_sb, ov_sr;    // visual_ln_object output
  const bf16_t* obj_ling; long ol_sb, ol_sr;   // dense linguistic part, or
  const int64_t* obj_ling_idx;                 // [B,R] row selector into obj_ling (table mode, strides ignored)
  const float* gamma; const float* beta;
  bf16_t* pre;                // [M,H] pre-LN sum (saved for backward)
  float* stats;               // [M,2]
  bf16_t* out;                // [M,H]
  int B, T, R, S, H, V, P;
  float eps;
  uint32_t drop_thr; float drop_scale; const uint32_t* seed; uint32_t tag;
};

__device__ __forceinline__ void add_row_bf16(const bf16_t* src, int H, int lane, float (*acc)[4]) {
#pragma unroll
  for (int i = 0; i < EMB_MAX_IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const uint2 w = *(const uint2*)(src + c);
      acc[i][0] += bflo(w.x); acc[i][1] += bfhi(w.x); acc[i][2] += bflo(w.y); acc[i][3] += bfhi(w.y);
    }
  }
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(const EmbedParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.B * p.S) return;
  const int b = row / p.S, s = row % p.S;
  const int code = p.code[row], kind = code >> 16, idx = code & 0xffff;
  const int tl = p.text_len[b];
  const int H = p.H;
  float acc[EMB_MAX_IT][4];
#pragma unroll
  for (int i = 0; i < EMB_MAX_IT; ++i) acc[i][0] = acc[i][1] = acc[i][2] = acc[i][3] = 0.f;
  int type_id = 0, pos_id = s;  // pad rows: zeros + position s + type 0 (visual_linguistic_bert.py:210-227)
  if (kind == KIND_TEXT) {
    long id = p.text_ids[b * p.T + idx];
    id = id < 0 ? 0 : (id >= p.V ? p.V - 1 : id);
    add_row_bf16(p.word_emb + id * H, H, lane, acc);
    add_row_bf16(p.text_vis + b * p.tv_sb + idx * p.tv_st, H, lane, acc);
    type_id = p.text_type ? (int)p.text_type[b * p.T + idx] : 0;
  } else if (kind == KIND_OBJ) {
    add_row_bf16(p.obj_vis + b * p.ov_sb + idx * p.ov_sr, H, lane, acc);
    if (p.obj_ling_idx)
      add_row_bf16(p.obj_ling + (long)p.obj_ling_idx[b * p.R + idx] * H, H, lane, acc);
    else
      add_row_bf16(p.obj_ling + b * p.ol_sb + idx * p.ol_sr, H, lane, acc);
    type_id = 2;
    pos_id = tl;
  } else if (kind == KIND_END) {
    add_row_bf16(p.end_emb, H, lane, acc);
    type_id = 2;
    pos_id = tl + 1;
  }
  pos_id = min(pos_id, p.P - 1);
  add_row_bf16(p.pos_emb + (long)pos_id * H, H, lane, acc);
  add_row_bf16(p.type_emb + (long)type_id * H, H, lane, acc);

  // the bf16-rounded sum is what backward re-reads, so normalise exactly that
  bf16_t* pre = p.pre + (long)row * H;
  float s1 = 0.f;
#pragma unroll
  for (int i = 0; i < EMB_MAX_IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      uint2 w = {pack2bf(acc[i][0], acc[i][1]), pack2bf(acc[i][2], acc[i][3])};
      *(uint2*)(pre + c) = w;
      acc[i][0] = bflo(w.x); acc[i][1] = bfhi(w.x); acc[i][2] = bflo(w.y); acc[i][3] = bfhi(w.y);
      s1 += (acc[i][0] + acc[i][1]) + (acc[i][2] + acc[i][3]);
    }
  }
  const float mean = wave_sum(s1) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < EMB_MAX_IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = acc[i][k] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + p.eps);
  if (lane == 0) {
    p.stats[2 * (long)row] = mean;
    p.stats[2 * (long)row + 1] = rstd;
  }
  const uint32_t seed = (p.drop_thr && p.seed) ? *p.seed : 0u;
  bf16_t* o = p.out + (long)row * H;
#pragma unroll
  for (int i = 0; i < EMB_MAX_IT; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < H) {
      const float4 g = *(const float4*)(p.gamma + c);
      const float4 be = *(const float4*)(p.beta + c);
      float y[4] = {(acc[i][0] - mean) * rstd * g.x + be.x, (acc[i][1] - mean) * rstd * g.y + be.y,
                    (acc[i][2] - mean) * rstd * g.z + be.z, (acc[i][3] - mean) * rstd * g.w + be.w};
      if (p.drop_thr) {
        const uint32_t e = (uint32_t)row * (uint32_t)H + (uint32_t)c;
        const uint32_t h0 = vlb_rng_pair(seed, p.tag, e >> 1), h1 = vlb_rng_pair(seed, p.tag, (e >> 1) + 1);
        y[0] = ((h0 & 0xffffu) >= p.drop_thr) ? y[0] * p.drop_scale : 0.f;
        y[1] = ((h0 >> 16) >= p.drop_thr) ? y[1] * p.drop_scale : 0.f;
        y[2] = ((h1 & 0xffffu) >= p.drop_thr) ? y[2] * p.drop_scale : 0.f;
        y[3] = ((h1 >> 16) >= p.drop_thr) ? y[3] * p.drop_scale : 0.f;
      }
      uint2 w = {pack2bf(y[0], y[1]), pack2bf(y[2], y[3])};
      *(uint2*)(o + c) = w;
    }
  }
}

// ------------------------------------------------------------------------------------------
// embedding backward: one block per sample, waves stride over its S rows.
// Per row: dropout-mask dy, LayerNorm backward -> d (fp32, in registers) and then
//   word_emb[id] / pos_emb[pos] / end_emb   : global fp32 atomics (few duplicates)
//   type_emb[0..2], position row `text_len` shared by all objects, broadcast text-visual row,
//   2-row linguistic table                     : accumulated in LDS, one flush per block
//   d(visual parts)                            : d_text_vis / d_obj_vis fp32 rows
// ------------------------------------------------------------------------------------------
struct EmbedBwdParams {
  const bf16_t* dy;  // [M,H]
  const bf16_t* pre; const float* stats; const float* gamma;
  const int32_t* code; const int32_t* text_len;
  const int64_t* text_ids; const int64_t* text_type; const int64_t* obj_ling_idx;
  float* d_word; float* d_pos; float* d_type; float* d_end; float* d_gamma; float* d_beta;
  float* d_text_vis; long dtv_sb, dtv_st;  // dtv_st==0: per-sample sum (broadcast text_visual)
  float* d_obj_vis; long dov_sb, dov_sr;   // plain stores
  float* d_obj_ling; long dol_sb, dol_sr;  // table mode (obj_ling_idx): d_obj_ling is the [2,H] table grad
  int B, T, R, S, H, V, P;
  uint32_t drop_thr; float drop_scale; const uint32_t* seed; uint32_t tag;
};

// NIT = ceil(H/256) (register footprint follows H); 512 threads = 8 waves per sample.  The block-shared accumulators
// live in LDS in LANE-MAJOR order (element (i,k) of lane l at (i*4+k)*64 + l): ds_add_f32 from consecutive lanes hits
// consecutive banks (the natural column order puts lanes 4 floats apart = 8-way conflicts on every one of the
// ~40 LDS atomics a row issues).  LWD = NIT*256 floats per accumulator.
template <int NIT>
__global__ __launch_bounds__(512) void embed_bwd_kernel(const EmbedBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int LWD = NIT * 256;
  const int H = p.H;
  float* l_type = lds;               // [3][LWD]
  float* l_objpos = lds + 3 * LWD;   // [LWD]
  float* l_tv = lds + 4 * LWD;       // [LWD]
  float* l_tab = lds + 5 * LWD;      // [2][LWD]
  float* l_g = lds + 7 * LWD;        // [LWD]
  float* l_b = lds + 8 * LWD;        // [LWD]
  float* rowbuf = lds + 9 * LWD + (threadIdx.x >> 6) * LWD;   // per-wave row image (natural column order)
  for (int i = threadIdx.x; i < 9 * LWD; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int tl = p.text_len[b];
  const uint32_t seed = (p.drop_thr && p.seed) ? *p.seed : 0u;
  float gs[NIT][4], bs[NIT][4];
  // Round 6: the block-shared sums of the row loop are kept in REGISTERS per wave and go to LDS once, behind the loop.  Every row used to
  // issue 24-36 LDS float atomics per lane (type row + object position / per-sample text-visual sum / linguistic table row); measured
  // (VLB_EMBED_ABLATE build, batch 256): 306 us with them, 167 us without -- the LDS atomic unit, not the global atomics, was the bound
  // (SQ_LDS_IDX_ACTIVE = 100 % of the kernel in profiles/r05_gemm_pmc.txt).  t0 / t1: text rows by token type (their sum is the
  // per-sample text-visual gradient); ob: object rows (type 2, the shared object position row, and -- minus l0 -- table row 1);
  // l0: object rows that select linguistic table row 0.  The single end row of a sample keeps its LDS atomics.
  float t0[NIT][4], t1[NIT][4], ob[NIT][4], l0[NIT][4];
#pragma unroll
  for (int i = 0; i < NIT; ++i)
#pragma unroll
    for (int k = 0; k < 4; ++k) gs[i][k] = bs[i][k] = t0[i][k] = t1[i][k] = ob[i][k] = l0[i][k] = 0.f;

  // gridDim.y workgroups share one sample: each takes a contiguous chunk of its S rows
  const int rows_per = (p.S + gridDim.y - 1) / gridDim.y, s_lo = blockIdx.y * rows_per, s_hi = min(p.S, s_lo + rows_per);
  // gamma of the lane's columns is loop-invariant (columns beyond H: clamped load, masked use)
  float gam[NIT][4];
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const float4 gm = *(const float4*)(p.gamma + min((lane + 64 * i) * 4, H - 4));
    gam[i][0] = gm.x; gam[i][1] = gm.y; gam[i][2] = gm.z; gam[i][3] = gm.w;
  }
  for (int s = s_lo + wave; s < s_hi; s += nwave) {
    const long row = (long)b * p.S + s;
    // every load of the row -- its code, its statistics, the saved pre-LN row, the gradient row -- is issued before the first use
    // (unconditional, clamped columns: with loads under `if (c < H)` hipcc waited vmcnt(0) behind each of them, ~8 dependent round
    // trips per row)
    const int code = p.code[row];
    const float2 st = *(const float2*)(p.stats + 2 * row);
    uint2 wxr[NIT], wdr[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int cc = min((lane + 64 * i) * 4, H - 4);
      wxr[i] = *(const uint2*)(p.pre + row * H + cc);
      wdr[i] = *(const uint2*)(p.dy + row * H + cc);
    }
    const int kind = code >> 16, idx = code & 0xffff;
    if (kind == KIND_PAD) continue;  // pad rows never reach a loss; their dy is exactly zero
    const float mean = st.x, rstd = st.y;
    float xh[NIT][4], g[NIT][4];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = (lane + 64 * i) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) xh[i][k] = g[i][k] = 0.f;
      if (c < H) {
        const uint2 wx = wxr[i], wd = wdr[i];
        float x[4] = {bflo(wx.x), bfhi(wx.x), bflo(wx.y), bfhi(wx.y)};
        float d[4] = {bflo(wd.x), bfhi(wd.x), bflo(wd.y), bfhi(wd.y)};
        if (p.drop_thr) {
          const uint32_t e = (uint32_t)row * (uint32_t)H + (uint32_t)c;
          const uint32_t h0 = vlb_rng_pair(seed, p.tag, e >> 1), h1 = vlb_rng_pair(seed, p.tag, (e >> 1) + 1);
          d[0] = ((h0 & 0xffffu) >= p.drop_thr) ? d[0] * p.drop_scale : 0.f;
          d[1] = ((h0 >> 16) >= p.drop_thr) ? d[1] * p.drop_scale : 0.f;
          d[2] = ((h1 & 0xffffu) >= p.drop_thr) ? d[2] * p.drop_scale : 0.f;
          d[3] = ((h1 >> 16) >= p.drop_thr) ? d[3] * p.drop_scale : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xh[i][k] = (x[k] - mean) * rstd;
          gs[i][k] += d[k] * xh[i][k];
          bs[i][k] += d[k];
          g[i][k] = d[k] * gam[i][k];
          s1 += g[i][k];
          s2 += g[i][k] * xh[i][k];
        }
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
    // destinations
    int type_id = 0, pos_id = s;
    float* w_dst = nullptr;   // global atomic destination for the "linguistic" row
    float* v_dst = nullptr;   // plain-store destination for the visual part
    bool pos_lds = false;
    if (kind == KIND_TEXT) {
      long id = p.text_ids[b * p.T + idx];
      id = id < 0 ? 0 : (id >= p.V ? p.V - 1 : id);
      w_dst = p.d_word + id * H;
      type_id = p.text_type ? (int)p.text_type[b * p.T + idx] : 0;
      if (p.d_text_vis && p.dtv_st != 0) v_dst = p.d_text_vis + b * p.dtv_sb + idx * p.dtv_st;      // (dtv_st == 0: per-sample sum, below)
    } else if (kind == KIND_OBJ) {
      type_id = 2;
      pos_lds = true;
      if (p.d_obj_vis) v_dst = p.d_obj_vis + b * p.dov_sb + idx * p.dov_sr;
    } else {  // END
      type_id = 2;
      pos_id = tl + 1;
      w_dst = p.d_end;
    }
    pos_id = min(pos_id, p.P - 1);
    float* pos_dst = p.d_pos + (long)pos_id * H;
    // (wave-uniform: one row per wave -- made provably so for scalar branches)
    const int ukind = __builtin_amdgcn_readfirstlane(kind), utype = __builtin_amdgcn_readfirstlane(type_id);
    const bool ling0 = (ukind == KIND_OBJ) && p.obj_ling_idx && __builtin_amdgcn_readfirstlane((int)(p.obj_ling_idx[b * p.R + idx] == 0));
    float* dl_dst = (kind == KIND_OBJ && !p.obj_ling_idx && p.d_obj_ling) ? p.d_obj_ling + b * p.dol_sb + idx * p.dol_sr : nullptr;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int c = (lane + 64 * i) * 4;
      if (c < H) {
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d[k] = rstd * (g[i][k] - s1 - xh[i][k] * s2);
          if (ukind == KIND_TEXT) {
            if (utype == 0) t0[i][k] += d[k];
            else if (utype == 1) t1[i][k] += d[k];
            else {                                                                       // (a text token of type 2: no caller produces one)
              atomicAdd(l_type + 2 * LWD + (i * 4 + k) * 64 + lane, d[k]);
              if (p.d_text_vis && p.dtv_st == 0) atomicAdd(l_tv + (i * 4 + k) * 64 + lane, d[k]);
            }
          } else if (ukind == KIND_OBJ) {
            ob[i][k] += d[k];
            if (ling0) l0[i][k] += d[k];
          } else {                                                                       // the end row
            atomicAdd(l_type + 2 * LWD + (i * 4 + k) * 64 + lane, d[k]);
          }
        }
        if (v_dst) *(float4*)(v_dst + c) = make_float4(d[0], d[1], d[2], d[3]);
        if (dl_dst) *(float4*)(dl_dst + c) = make_float4(d[0], d[1], d[2], d[3]);
        *(float4*)(rowbuf + c) = make_float4(d[0], d[1], d[2], d[3]);
      }
    }
    // Global atomics go out LANE-CONSECUTIVE (64 lanes = 256 contiguous bytes per instruction) through a per-wave LDS
    // row image: in the 4-floats-per-lane register layout one atomic instruction touches every 4th float of 1 KB,
    // i.e. four times the cache-line requests for the same data, and this kernel is bound by exactly that rate.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the row image is wave-private: order the writes above before the reads below
    __builtin_amdgcn_wave_barrier();
    if (!pos_lds || w_dst) {
      for (int c = lane; c < H; c += 64) {
        const float d = rowbuf[c];
        if (!pos_lds) atomicAdd(pos_dst + c, d);
        if (w_dst) atomicAdd(w_dst + c, d);
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the waves add their register sums into the block's accumulators ONE WAVE AT A TIME with plain LDS reads / writes (a float atomic in
  // LDS costs ~170 cycles per wave instruction on this chip and the unit is shared by the CU's 16 waves: 108 of them per wave were
  // still 2/3 of what the row loop's used to cost)
  __syncthreads();      // (the end rows' atomics of the loop above are complete)
  for (int r = 0; r < nwave; ++r) {
    if (wave == r) {
#pragma unroll
      for (int i = 0; i < NIT; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = (i * 4 + k) * 64 + lane;
          l_g[q] += gs[i][k];
          l_b[q] += bs[i][k];
          l_type[q] += t0[i][k];
          l_type[LWD + q] += t1[i][k];
          l_type[2 * LWD + q] += ob[i][k];
          l_objpos[q] += ob[i][k];
          if (p.d_text_vis && p.dtv_st == 0) l_tv[q] += t0[i][k] + t1[i][k];
          if (p.obj_ling_idx) {
            l_tab[q] += l0[i][k];
            l_tab[LWD + q] += ob[i][k] - l0[i][k];
          }
        }
    }
    __syncthreads();
  }
  const int pos_obj = min(tl, p.P - 1);
  for (int c = threadIdx.x; c < H; c += blockDim.x) {
    const int chunk = c >> 2, q = (((chunk >> 6) * 4 + (c & 3)) << 6) + (chunk & 63);   // natural column -> lane-major slot
#pragma unroll
    for (int t = 0; t < 3; ++t)
      if (l_type[t * LWD + q] != 0.f) atomicAdd(p.d_type + t * H + c, l_type[t * LWD + q]);
    if (l_objpos[q] != 0.f) atomicAdd(p.d_pos + (long)pos_obj * H + c, l_objpos[q]);
    if (p.d_text_vis && p.dtv_st == 0) {   // per-sample sum over the text rows: plain store when one workgroup owns the sample
      if (gridDim.y == 1) p.d_text_vis[b * p.dtv_sb + c] = l_tv[q];
      else if (l_tv[q] != 0.f) atomicAdd(p.d_text_vis + b * p.dtv_sb + c, l_tv[q]);
    }
    if (p.obj_ling_idx && p.d_obj_ling) {
      if (l_tab[q] != 0.f) atomicAdd(p.d_obj_ling + c, l_tab[q]);
      if (l_tab[LWD + q] != 0.f) atomicAdd(p.d_obj_ling + H + c, l_tab[LWD + q]);
    }
    atomicAdd(p.d_gamma + c, l_g[q]);
    atomicAdd(p.d_beta + c, l_b[q]);
  }
}

// out[i] = src[idx[i]] (idx<0 -> zeros); rows of H bf16.
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ idx, bf16_t* __restrict__ out,
                                                          int n, int H) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const int r = idx[i];
  for (int c = lane * 8; c < H; c += 512) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r >= 0) v = *(const uint4*)(src + (long)r * H + c);
    *(uint4*)(out + (long)i * H + c) = v;
  }
}

// out[idx[i]] = src[i] for idx[i] >= 0 (rows of H bf16; rows of `out` that no index names keep their contents): the inverse of a
// gather through the same index list (gradient of the compacted MLM rows back to their text positions).
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ idx, bf16_t* __restrict__ out,
                                                           int n, int H) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= n) return;
  const int r = idx[i];
  if (r < 0) return;
  for (int c = lane * 8; c < H; c += 512) *(uint4*)(out + (long)r * H + c) = *(const uint4*)(src + (long)i * H + c);
}

// dX[b,s] = (s<T ? d_text[b,s] : 0) + (row (b,s) is the j-th object ? d_obj[b,j] : 0)
// (inverse of the text/object split at common/visual_linguistic_bert.py:146-166).
__global__ __launch_bounds__(256) void head_grad_combine_kernel(const bf16_t* __restrict__ d_text, const bf16_t* __restrict__ d_obj,
                                                                const int32_t* __restrict__ code, bf16_t* __restrict__ dx, int B, int T,
                                                                int R, int S, int H) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * S) return;
  const int b = row / S, s = row % S;
  const int c_ = code[row], kind = c_ >> 16, idx = c_ & 0xffff;
  const bf16_t* t = (s < T && d_text) ? d_text + ((long)b * T + s) * H : nullptr;
  const bf16_t* o = (kind == KIND_OBJ && d_obj) ? d_obj + ((long)b * R + idx) * H : nullptr;
  for (int c = lane * 4; c < H; c += 256) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (t) {
      const uint2 w = *(const uint2*)(t + c);
      v[0] += bflo(w.x); v[1] += bfhi(w.x); v[2] += bflo(w.y); v[3] += bfhi(w.y);
    }
    if (o) {
      const uint2 w = *(const uint2*)(o + c);
      v[0] += bflo(w.x); v[1] += bfhi(w.x); v[2] += bflo(w.y); v[3] += bfhi(w.y);
    }
    uint2 w = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
    *(uint2*)(dx + (long)row * H + c) = w;
  }
}

// out_bf16[i] = (y[i] > 0) ? g_f32[i] : 0      (ReLU backward of obj_downsample + fp32->bf16)
__global__ void relu_bwd_cast_kernel(const float* __restrict__ g, const bf16_t* __restrict__ y, bf16_t* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 gv = *(const float4*)(g + i);
  const uint2 w = *(const uint2*)(y + i);
  uint2 o = {pack2bf(bflo(w.x) > 0.f ? gv.x : 0.f, bfhi(w.x) > 0.f ? gv.y : 0.f),
             pack2bf(bflo(w.y) > 0.f ? gv.z : 0.f, bfhi(w.y) > 0.f ? gv.w : 0.f)};
  *(uint2*)(out + i) = o;
}

// out = dG * gelu'(U)   (GELU backward where no GEMM epilogue is available: MLM transform, whose
// LayerNorm sits between the activation and the next GEMM -- modeling.py:448-452)
__global__ void dgelu_mul_kernel(const bf16_t* __restrict__ dg, const bf16_t* __restrict__ u, bf16_t* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const uint2 a = *(const uint2*)(dg + i), b = *(const uint2*)(u + i);
  uint2 o = {pack2bf(bflo(a.x) * dgelu_f(bflo(b.x)), bfhi(a.x) * dgelu_f(bfhi(b.x))),
             pack2bf(bflo(a.y) * dgelu_f(bflo(b.y)), bfhi(a.y) * dgelu_f(bfhi(b.y)))};
  *(uint2*)(out + i) = o;
}

// out = a * b  (GELU backward with the derivative tile saved by the forward GEMM epilogue, act 4)
__global__ void mul_bf16_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, bf16_t* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i >= n) return;
  const uint4 x = *(const uint4*)(a + i), y = *(const uint4*)(b + i);
  uint4 o;
  o.x = pack2bf(bflo(x.x) * bflo(y.x), bfhi(x.x) * bfhi(y.x));
  o.y = pack2bf(bflo(x.y) * bflo(y.y), bfhi(x.y) * bfhi(y.y));
  o.z = pack2bf(bflo(x.z) * bflo(y.z), bfhi(x.z) * bfhi(y.z));
  o.w = pack2bf(bflo(x.w) * bflo(y.w), bfhi(x.w) * bfhi(y.w));
  *(uint4*)(out + i) = o;
}

// out = dy * (1 - y^2)   (backward of y = tanh(x), BertPooler)
__global__ void tanh_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ y, bf16_t* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const uint2 a = *(const uint2*)(dy + i), b = *(const uint2*)(y + i);
  const float y0 = bflo(b.x), y1 = bfhi(b.x), y2 = bflo(b.y), y3 = bfhi(b.y);
  uint2 o = {pack2bf(bflo(a.x) * (1.f - y0 * y0), bfhi(a.x) * (1.f - y1 * y1)),
             pack2bf(bflo(a.y) * (1.f - y2 * y2), bfhi(a.y) * (1.f - y3 * y3))};
  *(uint2*)(out + i) = o;
}

// ---------------------------------------------------------------------------------- C ABI
extern "C" int vlb_seq_layout(const uint8_t* text_mask, const uint8_t* obj_mask, int B, int T, int R, int S, int32_t* code,
                              int32_t* text_len, int32_t* nobj, int32_t* text_rows, int32_t* obj_rows, float* attn_mask,
                              hipStream_t stream) {
  VLB_CHECK_ARG(B > 0 && T > 0 && R >= 0 && S >= T + R + 1, "vlb_seq_layout: need S >= T+R+1 (B=%d T=%d R=%d S=%d)", B, T, R, S);
  VLB_CHECK_ARG(T < 65536 && R < 65536, "vlb_seq_layout: T/R too large");
  hipLaunchKernelGGL(seq_layout_kernel, dim3(B), dim3(64), 0, stream, text_mask, obj_mask, B, T, R, S, code, text_len, nobj,
                     text_rows, obj_rows, attn_mask);
  VLB_CHECK_LAUNCH("vlb_seq_layout");
  return VLB_OK;
}

extern "C" int vlb_obj_prep_fwd(const float* boxes, long ldbox, const float* im_info, long ldinfo, const int64_t* mvrc_ops,
                                const float* mask_emb, void* out, int B, int R, float drop_p, const uint32_t* seed, uint32_t tag,
                                hipStream_t stream) {
  if (B * R <= 0) return VLB_OK;
  VLB_CHECK_ARG(ldbox >= 4 + 2048 && (ldbox % 4) == 0, "vlb_obj_prep_fwd: ldbox=%ld must be >= 2052 and a multiple of 4", ldbox);
  VLB_CHECK_ARG(boxes && im_info && out && ldinfo >= 2, "vlb_obj_prep_fwd: null argument or im_info rows narrower than (width, height): ldinfo=%ld",
                ldinfo);
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_obj_prep_fwd: dropout needs a device seed pointer");
  const uint32_t thr = vlb_drop_thr(drop_p);
  hipLaunchKernelGGL(obj_prep_fwd_kernel, dim3(B * R), dim3(256), 0, stream, boxes, ldbox, im_info, ldinfo, mvrc_ops, mask_emb, (bf16_t*)out,
                     B, R, thr, vlb_drop_scale(thr), seed, tag);
  VLB_CHECK_LAUNCH("vlb_obj_prep_fwd");
  return VLB_OK;
}

extern "C" int vlb_masked_colsum(const void* src, long lds_, const int64_t* sel, int rows, int C, float* dst, float drop_p,
                                 const uint32_t* seed, uint32_t tag, uint32_t row_elems, uint32_t col_off, hipStream_t stream) {
  if (rows <= 0 || C <= 0) return VLB_OK;
  const uint32_t thr = vlb_drop_thr(drop_p);
  int gy = rows < 64 ? rows : 64;
  hipLaunchKernelGGL(masked_colsum_kernel, dim3(vlb_cdiv(C, 256), gy), dim3(256), 0, stream, (const bf16_t*)src, lds_, sel, rows, C,
                     dst, thr, vlb_drop_scale(thr), seed, tag, row_elems, col_off);
  VLB_CHECK_LAUNCH("vlb_masked_colsum");
  return VLB_OK;
}

extern "C" int vlb_embed_fwd(const int32_t* code, const int32_t* text_len, const int64_t* text_ids, const int64_t* text_type,
                             const void* word_emb, const void* pos_emb, const void* type_emb, const void* end_emb,
                             const void* text_vis, long tv_sb, long tv_st, const void* obj_vis, long ov_sb, long ov_sr,
                             const void* obj_ling, long ol_sb, long ol_sr, const int64_t* obj_ling_idx, const float* gamma,
                             const float* beta, void* pre, float* stats, void* out, int B, int T, int R, int S, int H, int V,
                             int P, float eps, float drop_p, const uint32_t* seed, uint32_t tag, hipStream_t stream) {
  VLB_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= 256 * EMB_MAX_IT, "vlb_embed_fwd: unsupported H=%d", H);
  VLB_CHECK_ARG(code && text_len && text_ids && word_emb && pos_emb && type_emb && end_emb && text_vis && obj_vis && obj_ling,
                "vlb_embed_fwd: null input");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_embed_fwd: dropout needs a device seed pointer");
  EmbedParams p;
  p.code = code; p.text_len = text_len; p.text_ids = text_ids; p.text_type = text_type;
  p.word_emb = (const bf16_t*)word_emb; p.pos_emb = (const bf16_t*)pos_emb; p.type_emb = (const bf16_t*)type_emb;
  p.end_emb = (const bf16_t*)end_emb;
  p.text_vis = (const bf16_t*)text_vis; p.tv_sb = tv_sb; p.tv_st = tv_st;
  p.obj_vis = (const bf16_t*)obj_vis; p.ov_sb = ov_sb; p.ov_sr = ov_sr;
  p.obj_ling = (const bf16_t*)obj_ling; p.ol_sb = ol_sb; p.ol_sr = ol_sr; p.obj_ling_idx = obj_ling_idx;
  p.gamma = gamma; p.beta = beta; p.pre = (bf16_t*)pre; p.stats = stats; p.out = (bf16_t*)out;
  p.B = B; p.T = T; p.R = R; p.S = S; p.H = H; p.V = V; p.P = P; p.eps = eps;
  p.drop_thr = vlb_drop_thr(drop_p); p.drop_scale = vlb_drop_scale(p.drop_thr); p.seed = seed; p.tag = tag;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(vlb_cdiv((long)B * S, 4)), dim3(256), 0, stream, p);
  VLB_CHECK_LAUNCH("vlb_embed_fwd");
  return VLB_OK;
}

extern "C" int vlb_embed_bwd(const void* dy, const void* pre, const float* stats, const float* gamma, const int32_t* code,
                             const int32_t* text_len, const int64_t* text_ids, const int64_t* text_type,
                             const int64_t* obj_ling_idx, float* d_word, float* d_pos, float* d_type, float* d_end,
                             float* d_gamma, float* d_beta, float* d_text_vis, long dtv_sb, long dtv_st, float* d_obj_vis,
                             long dov_sb, long dov_sr, float* d_obj_ling, long dol_sb, long dol_sr, int B, int T, int R, int S,
                             int H, int V, int P, float drop_p, const uint32_t* seed, uint32_t tag, int text_vis_zeroed,
                             hipStream_t stream) {
  VLB_CHECK_ARG(H > 0 && (H % 4) == 0 && H <= 256 * EMB_MAX_IT, "vlb_embed_bwd: unsupported H=%d", H);
  VLB_CHECK_ARG(dy && pre && stats && gamma && code && text_len && text_ids && d_word && d_pos && d_type && d_end && d_gamma &&
                    d_beta, "vlb_embed_bwd: null input");
  VLB_CHECK_ARG(!(drop_p > 0.f) || seed, "vlb_embed_bwd: dropout needs a device seed pointer");
  EmbedBwdParams p;
  p.dy = (const bf16_t*)dy; p.pre = (const bf16_t*)pre; p.stats = stats; p.gamma = gamma; p.code = code; p.text_len = text_len;
  p.text_ids = text_ids; p.text_type = text_type; p.obj_ling_idx = obj_ling_idx;
  p.d_word = d_word; p.d_pos = d_pos; p.d_type = d_type; p.d_end = d_end; p.d_gamma = d_gamma; p.d_beta = d_beta;
  p.d_text_vis = d_text_vis; p.dtv_sb = dtv_sb; p.dtv_st = dtv_st;
  p.d_obj_vis = d_obj_vis; p.dov_sb = dov_sb; p.dov_sr = dov_sr;
  p.d_obj_ling = d_obj_ling; p.dol_sb = dol_sb; p.dol_sr = dol_sr;
  p.B = B; p.T = T; p.R = R; p.S = S; p.H = H; p.V = V; p.P = P;
  p.drop_thr = vlb_drop_thr(drop_p); p.drop_scale = vlb_drop_scale(p.drop_thr); p.seed = seed; p.tag = tag;
  VLB_CHECK_ARG((dtv_sb % 4) == 0 && (dtv_st % 4) == 0 && (dov_sb % 4) == 0 && (dov_sr % 4) == 0 && (dol_sb % 4) == 0 &&
                    (dol_sr % 4) == 0, "vlb_embed_bwd: output strides must be multiples of 4 floats");
  const int nit = vlb_cdiv(H, 256);
  // small batches: several workgroups per sample so that >= ~512 are in flight (needs d_text_vis zeroed by the caller
  // in the broadcast mode, where the per-sample sum is then accumulated with atomics)
  int split = 1;
  if (d_text_vis == nullptr || dtv_st != 0 || text_vis_zeroed) {
    split = vlb_cdiv(512, B);
    if (split > 8) split = 8;
    if (split > S) split = S;
    if (split < 1) split = 1;
  }
#define EMB_BWD(NIT)                                                                                                         \
  do {                                                                                                                       \
    constexpr int smem = (9 + 8) * NIT * 256 * (int)sizeof(float);                                                                 \
    if (smem > 48 * 1024)                                                                                                    \
      (void)hipFuncSetAttribute((const void*)embed_bwd_kernel<NIT>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);       \
    hipLaunchKernelGGL(embed_bwd_kernel<NIT>, dim3(B, split), dim3(512), smem, stream, p);                                   \
  } while (0)
  if (nit <= 1) EMB_BWD(1); else if (nit == 2) EMB_BWD(2); else if (nit == 3) EMB_BWD(3); else if (nit == 4) EMB_BWD(4); else EMB_BWD(8);
#undef EMB_BWD
  VLB_CHECK_LAUNCH("vlb_embed_bwd");
  return VLB_OK;
}

extern "C" int vlb_gather_rows(const void* src, const int32_t* idx, void* out, int n, int H, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG((H % 8) == 0, "vlb_gather_rows: H must be a multiple of 8");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(vlb_cdiv(n, 4)), dim3(256), 0, stream, (const bf16_t*)src, idx, (bf16_t*)out, n, H);
  VLB_CHECK_LAUNCH("vlb_gather_rows");
  return VLB_OK;
}

extern "C" int vlb_scatter_rows(const void* src, const int32_t* idx, void* out, int n, int H, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG(src && idx && out && (H % 8) == 0, "vlb_scatter_rows: bad argument (H must be a multiple of 8)");
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(vlb_cdiv(n, 4)), dim3(256), 0, stream, (const bf16_t*)src, idx, (bf16_t*)out, n, H);
  VLB_CHECK_LAUNCH("vlb_scatter_rows");
  return VLB_OK;
}

extern "C" int vlb_head_grad_combine(const void* d_text, const void* d_obj, const int32_t* code, void* dx, int B, int T, int R,
                                     int S, int H, hipStream_t stream) {
  VLB_CHECK_ARG((H % 4) == 0, "vlb_head_grad_combine: H must be a multiple of 4");
  hipLaunchKernelGGL(head_grad_combine_kernel, dim3(vlb_cdiv((long)B * S, 4)), dim3(256), 0, stream, (const bf16_t*)d_text,
                     (const bf16_t*)d_obj, code, (bf16_t*)dx, B, T, R, S, H);
  VLB_CHECK_LAUNCH("vlb_head_grad_combine");
  return VLB_OK;
}

extern "C" int vlb_relu_bwd_cast(const float* g, const void* y, void* out, long n, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG((n % 4) == 0, "vlb_relu_bwd_cast: n must be a multiple of 4");
  hipLaunchKernelGGL(relu_bwd_cast_kernel, dim3(vlb_cdiv(n / 4, 256)), dim3(256), 0, stream, g, (const bf16_t*)y, (bf16_t*)out, n);
  VLB_CHECK_LAUNCH("vlb_relu_bwd_cast");
  return VLB_OK;
}

extern "C" int vlb_dgelu_mul(const void* dg, const void* u, void* out, long n, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG((n % 4) == 0, "vlb_dgelu_mul: n must be a multiple of 4");
  hipLaunchKernelGGL(dgelu_mul_kernel, dim3(vlb_cdiv(n / 4, 256)), dim3(256), 0, stream, (const bf16_t*)dg, (const bf16_t*)u,
                     (bf16_t*)out, n);
  VLB_CHECK_LAUNCH("vlb_dgelu_mul");
  return VLB_OK;
}

extern "C" int vlb_mul_bf16(const void* a, const void* b, void* out, long n, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG((n % 8) == 0, "vlb_mul_bf16: n must be a multiple of 8");
  hipLaunchKernelGGL(mul_bf16_kernel, dim3(vlb_cdiv(n / 8, 256)), dim3(256), 0, stream, (const bf16_t*)a, (const bf16_t*)b,
                     (bf16_t*)out, n);
  VLB_CHECK_LAUNCH("vlb_mul_bf16");
  return VLB_OK;
}

extern "C" int vlb_tanh_bwd(const void* dy, const void* y, void* out, long n, hipStream_t stream) {
  if (n <= 0) return VLB_OK;
  VLB_CHECK_ARG((n % 4) == 0, "vlb_tanh_bwd: n must be a multiple of 4");
  hipLaunchKernelGGL(tanh_bwd_kernel, dim3(vlb_cdiv(n / 4, 256)), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)y,
                     (bf16_t*)out, n);
  VLB_CHECK_LAUNCH("vlb_tanh_bwd");
  return VLB_OK;
}
