// Hardware-semantics probe used while developing the LDS-transpose-read kernels.  A development tool: built on the fly into its own
// shared object by tools/probe_tr.py; NOT part of libvlbert_hip.so or of the product ABI (include/vlbert_hip.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) short s16x4;

// LDS[i] = in[i] (u16, n_elems <= 8192); every lane issues ONE ds_read_b64_tr_b16 at byte address addr[lane];
// out[lane*4 + j] = element j of the lane's result.
__global__ void probe_tr_read_kernel(const uint16_t* in, int n_elems, const int* addr, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < n_elems; i += blockDim.x) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)((char*)lds + addr[lane]));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

extern "C" int probe_tr_read(const uint16_t* in, int n_elems, const int* addr, uint16_t* out, hipStream_t stream) {
  if (!in || !addr || !out || n_elems <= 0 || n_elems > 8192) return -1;
  hipLaunchKernelGGL(probe_tr_read_kernel, dim3(1), dim3(64), 0, stream, in, n_elems, addr, out);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
