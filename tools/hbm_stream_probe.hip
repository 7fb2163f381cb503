// What streaming rate does one MI355X actually give the HBM-bound kernels of the step (AdamW, LayerNorm, casts), and does the access pattern
// matter?  Stand-alone probe (own main, no torch): hipcc --offload-arch=gfx950 -O3 tools/hbm_stream_probe.hip -o /tmp/hbm && /tmp/hbm
//
// Patterns, each a template instantiation of ONE kernel:
//   op      copy (1 read + 1 write, 16 B per lane per access) | read (sum, 1 read) | fill (1 write) |
//           adamw (the optimizer's stream mix: 4 fp32 reads + 3 fp32 writes + 1 bf16 write per element, the real arithmetic)
//   NT      plain accesses | non-temporal loads and stores
//   UNROLL  independent 16-B accesses per lane and stream in flight before the first use (1 / 2 / 4)
//   order   grid-stride (block b touches 4 KiB x UNROLL at b, b + grid, ...) | chunked (block b streams ONE contiguous range) |
//           xcd (chunked, the eight XCDs -- blockIdx % 8 -- each own one contiguous eighth of the buffer)
//   grid    blocks of 256 threads: 2048 (what optim.hip launches) / 4096 / 8192 / 16384 / one access per lane (one-shot)
// Buffers are far larger than the 256 MB MALL (adamw: 115 M elements = the model; copy: 1 GiB each way).  Reported: GB/s of
// algorithmic bytes, median of 15 launches after 3 warm-ups, HIP events around each launch.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) uint32_t u2;

enum { OP_COPY = 0, OP_READ = 1, OP_FILL = 2, OP_ADAMW = 3 };
enum { ORD_STRIDE = 0, ORD_CHUNK = 1, ORD_XCD = 2 };

template <bool NT>
__device__ __forceinline__ f4 ld(const f4* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st(f4* p, f4 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

__device__ __forceinline__ uint32_t pack2bf(float a, float b) {
  const uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
  return ((x + 0x7fffu + ((x >> 16) & 1u)) >> 16) | ((y + 0x7fffu + ((y >> 16) & 1u)) & 0xffff0000u);
}

struct Bufs {
  f4 *p, *g, *m, *v;      // adamw: p, g, m, v ; copy: p -> g ; read: p ; fill: p
  u2* p16;
  float* sink;
  long n4;                // 16-B units per stream
};

// units [u0, u1) of 16 B are this block's share; inside it the lanes walk 256 x UNROLL units at a time
template <int OP, bool NT, int UNROLL, int ORDER>
__global__ __launch_bounds__(256) void stream_kernel(Bufs b) {
  const long n4 = b.n4;
  long base, step, end;
  if (ORDER == ORD_STRIDE) {
    base = (long)blockIdx.x * 256 * UNROLL;
    step = (long)gridDim.x * 256 * UNROLL;
    end = n4;
  } else {
    long blk = blockIdx.x;
    if (ORDER == ORD_XCD) blk = (blk & 7) * (gridDim.x >> 3) + (blk >> 3);      // XCD x = blockIdx % 8 streams the x-th eighth
    const long per = ((n4 + gridDim.x - 1) / gridDim.x + 256 * UNROLL - 1) / (256 * UNROLL) * (256 * UNROLL);
    base = blk * per;
    end = base + per < n4 ? base + per : n4;
    step = 256 * UNROLL;
  }
  float acc = 0.f;
  for (long u = base; u < end; u += step) {
    f4 a[UNROLL], g[UNROLL], m[UNROLL], v[UNROLL];
    bool ok[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const long i = u + k * 256 + threadIdx.x;
      ok[k] = i < end;
      const long j = ok[k] ? i : 0;
      if (OP != OP_FILL) a[k] = ld<NT>(b.p + j);
      if (OP == OP_ADAMW) {
        g[k] = ld<false>(b.g + j);      // (the gradient was just written by the backward pass: plain load, as in optim.hip)
        m[k] = ld<NT>(b.m + j);
        v[k] = ld<NT>(b.v + j);
      }
    }
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      const long i = u + k * 256 + threadIdx.x;
      if (!ok[k]) continue;
      if (OP == OP_COPY) st<NT>(b.g + i, a[k]);
      if (OP == OP_READ) acc += (a[k].x + a[k].y) + (a[k].z + a[k].w);
      if (OP == OP_FILL) st<NT>(b.p + i, (f4){1.f, 2.f, 3.f, 4.f});
      if (OP == OP_ADAMW) {
        f4 pp = a[k], mm = m[k], vv = v[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gg = g[k][e] * 0.5f;
          mm[e] = mm[e] * 0.9f + 0.1f * gg;
          vv[e] = vv[e] * 0.999f + 0.001f * gg * gg;
          pp[e] -= 1e-4f * (mm[e] / (sqrtf(vv[e]) + 1e-6f));
          pp[e] -= 1e-6f * pp[e];
        }
        st<NT>(b.p + i, pp);
        st<NT>(b.m + i, mm);
        st<NT>(b.v + i, vv);
        b.p16[i] = (u2){pack2bf(pp.x, pp.y), pack2bf(pp.z, pp.w)};
      }
    }
  }
  if (OP == OP_READ && acc == 123.456f) *b.sink = acc;
}

template <int OP, bool NT, int UNROLL, int ORDER>
static double run(const Bufs& b, int grid, double bytes, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  if (grid <= 0) grid = (int)((b.n4 + 256L * UNROLL - 1) / (256L * UNROLL));      // one-shot: every lane touches UNROLL units
  std::vector<float> ms;
  for (int it = 0; it < 18; ++it) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((stream_kernel<OP, NT, UNROLL, ORDER>), dim3(grid), dim3(256), 0, 0, b);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (it >= 3) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2];
  printf("%-8s %-3s unroll %d  %-7s grid %7d : %8.1f us  %7.1f GB/s\n", name, NT ? "nt" : "-", UNROLL,
         ORDER == ORD_STRIDE ? "stride" : ORDER == ORD_CHUNK ? "chunk" : "xcd", grid, med * 1e3, bytes / (med * 1e-3) / 1e9);
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  return med;
}

template <int OP>
static void sweep(const Bufs& b, double bytes, const char* name) {
  const int grids[] = {2048, 4096, 8192, 16384, 0};
  for (int g : grids) {
    run<OP, true, 1, ORD_STRIDE>(b, g, bytes, name);
    run<OP, true, 2, ORD_STRIDE>(b, g, bytes, name);
    run<OP, true, 4, ORD_STRIDE>(b, g, bytes, name);
  }
  run<OP, false, 1, ORD_STRIDE>(b, 2048, bytes, name);
  run<OP, false, 4, ORD_STRIDE>(b, 2048, bytes, name);
  run<OP, false, 4, ORD_STRIDE>(b, 0, bytes, name);
  for (int g : {2048, 8192}) {
    run<OP, true, 2, ORD_CHUNK>(b, g, bytes, name);
    run<OP, true, 4, ORD_CHUNK>(b, g, bytes, name);
    run<OP, true, 2, ORD_XCD>(b, g, bytes, name);
    run<OP, true, 4, ORD_XCD>(b, g, bytes, name);
  }
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# %s, %d CUs, memory clock %d kHz, bus %d bit\n", prop.name, prop.multiProcessorCount, prop.memoryClockRate, prop.memoryBusWidth);
  const long N_ADAM = 115L * 1000 * 1000 / 4 * 4, N_COPY = 1L << 28;      // elements (fp32)
  const long n = N_COPY > N_ADAM ? N_COPY : N_ADAM;
  Bufs b;
  CK(hipMalloc(&b.p, n * 4));
  CK(hipMalloc(&b.g, n * 4));
  CK(hipMalloc(&b.m, N_ADAM * 4));
  CK(hipMalloc(&b.v, N_ADAM * 4));
  CK(hipMalloc(&b.p16, N_ADAM * 2));
  CK(hipMalloc(&b.sink, 4));
  CK(hipMemset(b.p, 0, n * 4));
  CK(hipMemset(b.g, 0, n * 4));
  CK(hipMemset(b.m, 0, N_ADAM * 4));
  CK(hipMemset(b.v, 0, N_ADAM * 4));
  b.n4 = N_COPY / 4;
  sweep<OP_COPY>(b, 2.0 * N_COPY * 4, "copy");
  sweep<OP_READ>(b, 1.0 * N_COPY * 4, "read");
  sweep<OP_FILL>(b, 1.0 * N_COPY * 4, "fill");
  b.n4 = N_ADAM / 4;
  sweep<OP_ADAMW>(b, 30.0 * N_ADAM, "adamw");
  return 0;
}
