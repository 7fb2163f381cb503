// Are NON-TEMPORAL loads (global_load ... nt) on MI355X coherent with data another agent wrote since the same lines were last
// NT-loaded?  (Found the hard way: a squared-norm kernel that NT-loaded the gradient read STALE values after a host-to-device copy
// had rewritten it -- two data-parallel ranks derived different clip coefficients.)  The chip has eight XCDs with private L2s; plain
// loads are made coherent at kernel boundaries.  This probe pins down which (reader, writer) pairs are safe:
//   round k:   writer stores the value k into every element  (kernel with plain stores | kernel with NT stores | hipMemcpy H2D | hipMemset)
//              -- writer KERNELS use a rotated block -> data mapping, so a line is written by a different workgroup (XCD) than reads it
//              reader loads every element (plain | NT) and counts the elements that are not k  (stale)
// Buffer sizes: 8 MiB (fits the L2s), 64 MiB, 512 MiB (beyond the 256 MiB MALL).
// Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/nt_coherence_probe.hip -o /tmp/ntp && /tmp/ntp
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u4;

template <bool NT>
__global__ __launch_bounds__(256) void reader(const u4* __restrict__ buf, long n4, uint32_t expect, unsigned long long* __restrict__ stale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const u4 v = NT ? __builtin_nontemporal_load(buf + i) : buf[i];
  const int bad = (v.x != expect) + (v.y != expect) + (v.z != expect) + (v.w != expect);
  if (bad) atomicAdd(stale, (unsigned long long)bad);
}

template <bool NT>
__global__ __launch_bounds__(256) void writer(u4* __restrict__ buf, long n4, uint32_t value, int rotate) {
  const long blk = ((long)blockIdx.x + rotate) % gridDim.x;      // block b writes the data block b + rotate reads: another XCD (rotate % 8 != 0)
  const long i = blk * 256 + threadIdx.x;
  if (i >= n4) return;
  const u4 v = {value, value, value, value};
  if (NT) __builtin_nontemporal_store(v, buf + i);
  else buf[i] = v;
}

int main() {
  const long sizes[] = {8L << 20, 64L << 20, 512L << 20};
  const char* wname[] = {"kernel, plain stores", "kernel, NT stores", "hipMemcpy H2D", "hipMemsetD32"};
  unsigned long long* stale;
  CK(hipMalloc(&stale, 8));
  for (long bytes : sizes) {
    const long n4 = bytes / 16;
    const int grid = (int)((n4 + 255) / 256);
    u4* buf;
    CK(hipMalloc(&buf, bytes));
    std::vector<uint32_t> host(bytes / 4);
    for (int rd = 0; rd < 2; ++rd)
      for (int wr = 0; wr < 4; ++wr) {
        unsigned long long total = 0, worst = 0;
        const int rounds = 12;
        for (int k = 1; k <= rounds; ++k) {
          const uint32_t val = 1000u * (wr + 1) + k + 100000u * rd;
          if (wr == 0) hipLaunchKernelGGL(writer<false>, dim3(grid), dim3(256), 0, 0, buf, n4, val, 3 + (k % 5));
          else if (wr == 1) hipLaunchKernelGGL(writer<true>, dim3(grid), dim3(256), 0, 0, buf, n4, val, 3 + (k % 5));
          else if (wr == 2) {
            for (auto& x : host) x = val;
            CK(hipMemcpyAsync(buf, host.data(), bytes, hipMemcpyHostToDevice, 0));
          } else CK(hipMemsetD32Async((hipDeviceptr_t)buf, (int)val, bytes / 4, 0));
          CK(hipMemsetAsync(stale, 0, 8, 0));
          if (rd) hipLaunchKernelGGL(reader<true>, dim3(grid), dim3(256), 0, 0, buf, n4, val, stale);
          else hipLaunchKernelGGL(reader<false>, dim3(grid), dim3(256), 0, 0, buf, n4, val, stale);
          unsigned long long h = 0;
          CK(hipMemcpy(&h, stale, 8, hipMemcpyDeviceToHost));
          total += h;
          if (h > worst) worst = h;
        }
        printf("%4ld MiB  reader %-5s  writer %-22s : %s  (stale 4-byte elements over %d rounds: %llu, worst round %llu of %ld)\n", bytes >> 20,
               rd ? "NT" : "plain", wname[wr], total ? "STALE READS" : "coherent", rounds, total, worst, bytes / 4);
      }
    CK(hipFree(buf));
  }
  return 0;
}
