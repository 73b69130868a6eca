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
  /* `if (auto once = guard.begin()) { hipFuncSetAttribute(...); }` -- the body runs until ONE caller has finished it on
   * the active device: the done bit is published by the scope's destructor, i.e. AFTER the attribute call returned, so a
   * second thread launching the same kernel at the same moment either sees the bit (the opt-in exists) or makes the
   * idempotent call itself -- never a launch with > 64 KiB of dynamic LDS ahead of the opt-in. */
  struct Scope {
    std::atomic<uint32_t>* word;
    uint32_t bit;
    explicit operator bool() const { return word != nullptr; }
    ~Scope() { if (word != nullptr) word->fetch_or(bit, std::memory_order_release); }
  };
  Scope begin()
  {
    const int device = qnnp_hip_device();
    const uint32_t bit = 1u << (static_cast<uint32_t>(device < 0 ? 0 : device) & 31u);
    if ((done.load(std::memory_order_acquire) & bit) != 0) return Scope{nullptr, 0u};
    return Scope{&done, bit};
  }
};

}  // namespace qnnp
