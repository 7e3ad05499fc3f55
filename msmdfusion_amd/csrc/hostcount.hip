// hostcount.hip -- a count the host is waiting for, without the runtime's round trip.
//
// The index pass reads 13 counts back per step (rows of a strided conv's output set, size
// of a sparse_add union, ...), and next to the feature pass each `.item()` -- a device->host
// copy on the stream plus a stream synchronisation through the runtime's signal / interrupt
// path -- costs ~0.28 ms of latency on the step's critical path.  Here the counting kernel
// writes its total straight into a slot of pinned, device-mapped host memory (the `total`
// pointer the scan kernels already take), and the host spins on that slot: no copy, no
// stream synchronisation, no interrupt.
//
// Measured on the LC step: no gain -- 134.2-134.3 samples/s against 134.4
// with .item().  What a read waits for is the counting kernels getting their turn on a GPU
// the feature pass keeps full, not the runtime's copy + synchronise.  The Python mirror keeps
// the mechanism as an option (MSMD_HOST_COUNTS=1), off by default.
#include <chrono>

#include "common.hpp"

MSMD_EXPORT int msmd_host_device_pointer(void* host_ptr, void** device_ptr) {
  if (!host_ptr || !device_ptr) return MSMD_ERR_INVALID_ARG;
  return hipHostGetDevicePointer(device_ptr, host_ptr, 0) == hipSuccess ? MSMD_OK
                                                                       : MSMD_ERR_INVALID_ARG;
}

// Spin until *slot != sentinel (the kernel's store) or the timeout; -> value in *value.
MSMD_EXPORT int msmd_host_wait_i32(const int32_t* slot, int32_t sentinel, int64_t timeout_us,
                                   int32_t* value) {
  if (!slot || !value) return MSMD_ERR_INVALID_ARG;
  const volatile int32_t* p = slot;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; ++spins) {
    const int32_t v = *p;
    if (v != sentinel) {
      *value = v;
      return MSMD_OK;
    }
    __builtin_ia32_pause();
    if ((spins & 0xfff) == 0xfff &&
        std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() -
                                                              t0).count() > timeout_us)
      return MSMD_ERR_LAUNCH;      // the producing kernel never ran / failed
  }
}
