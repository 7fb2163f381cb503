// Development tool (not part of libvlbert_hip.so): a kernel that holds `blocks` workgroups of 256 threads resident for ~`us`
// microseconds each -- a stand-in for an RCCL collective kernel (one long-running workgroup per channel) -- so that the effect of a
// few occupied CUs on the persistent one-workgroup-per-CU GEMM kernels can be measured on a single GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void spin_kernel(long ticks, unsigned* sink) {
  const long t0 = wall_clock64();
  unsigned x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) {
    x = x * 1664525u + 1013904223u;
    __builtin_amdgcn_s_sleep(8);
  }
  if (x == 0xdeadbeefu) *sink = x;
}

extern "C" int contention_spin(int blocks, int us, unsigned* sink, hipStream_t stream) {
  // wall_clock64 ticks at 100 MHz
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, stream, (long)us * 100, sink);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
