/*
 * per_device.h -- "once per device" guard for launch-side set-up (HIP-internal, C++ only).
 * hipFuncSetAttribute (the > 64 KiB dynamic-LDS opt-in) is a PER-DEVICE property of a kernel: a process that drives
 * several GPUs (runtime.hip) must set it on each of them, and several threads may reach the same launch helper.
 */
#pragma once

#include <atomic>
#include <cstdint>

#include "qnnp_hip.h"

namespace qnnp {

struct PerDeviceOnce {
  std::atomic<uint32_t> done{0};
  /* true for the first caller on the active device; later callers (and concurrent ones) get false -- the attribute
   * call is idempotent, so a racing second thread that proceeds to launch a moment early is at worst a launch
   * the runtime rejects and reports (QNNP_HIP_ELAUNCH), never silent corruption */
  bool first()
  {
    const int device = qnnp_hip_device();
    const uint32_t bit = 1u << (static_cast<uint32_t>(device < 0 ? 0 : device) & 31u);
    return (done.fetch_or(bit, std::memory_order_acq_rel) & bit) == 0;
  }
};

}  // namespace qnnp
