// Error plumbing + library identification for libvlbert_hip.so.
#include <stdarg.h>
#include <string.h>

#include "vlb_common.h"

static thread_local char g_err[512] = "";

void vlb_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vlb_last_error(void) { return g_err; }

extern "C" int vlb_version(void) { return 100; }  // 0.1.0

// 16-bit storage type this build of the library computes in (vlb_common.h): 0 = bfloat16 (libvlbert_hip.so), 1 = IEEE fp16
// (libvlbert_hip_f16.so, -DVLB_ACT_F16).  Same entry points, same layouts; the host allocates its tensors accordingly.
extern "C" int vlb_act_dtype(void) { return VLB_ACT_IS_F16; }

// Fills name (<= cap bytes) with the device's gcnArchName; returns the number of compute units,
// or a negative code.  Used by the host to fail loudly when not running on gfx950.
extern "C" int vlb_device_info(int device, char* name, int cap) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    vlb_set_error("vlb_device_info: %s", hipGetErrorString(e));
    return VLB_ERR_HIP;
  }
  if (name && cap > 0) {
    strncpy(name, prop.gcnArchName, cap - 1);
    name[cap - 1] = 0;
  }
  return prop.multiProcessorCount;
}
