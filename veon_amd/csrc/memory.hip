// Device allocations the lift's output volume can ask for (veon_amd/placement.py).
//
// The channels-first pool kernel writes C planes 4*Z*Y*X bytes apart per
// workgroup, so its speed depends on how few page-table fragments cover the
// volume.  hipDeviceMallocContiguous asks the driver for
// physically contiguous VRAM, i.e. the largest fragments it can map.
#include <hip/hip_runtime.h>

#include "veon_hip.h"

extern "C" {

int veon_alloc_contiguous(void** ptr, int64_t bytes) {
  if (!ptr || bytes <= 0) return VEON_ERR_BAD_ARG;
  *ptr = nullptr;
  if (hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocContiguous) != hipSuccess) {
    (void)hipGetLastError();  // not sticky: the caller falls back to hipMalloc'ed memory
    *ptr = nullptr;
    return VEON_ERR_LAUNCH;
  }
  return VEON_OK;
}

int veon_alloc_device_flags(void** ptr, int64_t bytes, unsigned flags) {
  // any hipExtMallocWithFlags flag (hipDeviceMallocDefault 0, Finegrained 1,
  // Uncached 3, Contiguous 4): tools/addr_probe.py compares them
  if (!ptr || bytes <= 0) return VEON_ERR_BAD_ARG;
  *ptr = nullptr;
  if (hipExtMallocWithFlags(ptr, (size_t)bytes, flags) != hipSuccess) {
    (void)hipGetLastError();
    *ptr = nullptr;
    return VEON_ERR_LAUNCH;
  }
  return VEON_OK;
}

int veon_free_device(void* ptr) {
  if (!ptr) return VEON_OK;
  return hipFree(ptr) == hipSuccess ? VEON_OK : VEON_ERR_LAUNCH;
}

}  // extern "C"
