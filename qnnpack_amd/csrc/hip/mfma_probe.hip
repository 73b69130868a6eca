/*
 * mfma_probe.hip -- qnnp_gfx950_mfma_probe(): the rate the whole chip sustains on a bare
 * v_mfma_i32_32x32x32_i8 loop (no LDS, no global traffic). bench.py reports it beside the GEMM roofline: on
 * MI355X the figure depends on the operand DATA (zero operands hold 2.4 GHz, random ones make the power
 * management drop the clock), so it is the practical ceiling the GEMM kernel's fraction should be read against.
 * The reference has no counterpart. Same loop as tools/ubench_mfma.hip.
 *
 * Measurement code: built into libqnnpack_gfx950_dbg.so, NOT into the product library. Self-contained -- it runs
 * on the calling thread's current HIP device and default stream and needs no library state.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <vector>

#include "qnnp_hip.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_probe_kernel(const v4i* in, int* out, int iters)
{
  v4i a[4], b[4];   // four different operand pairs in rotation: the multiplier inputs toggle like in a real GEMM
#pragma unroll
  for (int j = 0; j < 4; j++) {
    a[j] = in[threadIdx.x + j * 1024];
    b[j] = in[threadIdx.x + j * 1024 + 512];
  }
  v16i acc[8];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

}  // namespace

/* returns 0 on success, a negative QNNP_HIP_* code otherwise; *tops_out = TOP/s over all compute units */
extern "C" int qnnp_gfx950_mfma_probe(int random_operands, int iters, float* tops_out)
{
  if (tops_out == nullptr || iters <= 0) return QNNP_HIP_EINVAL;
  int device = 0;
  hipDeviceProp_t props;
  if (hipGetDevice(&device) != hipSuccess || hipGetDeviceProperties(&props, device) != hipSuccess) return QNNP_HIP_ENODEV;
  const int compute_units = props.multiProcessorCount;
  if (compute_units <= 0) return QNNP_HIP_ENODEV;
  std::vector<uint32_t> host(4096 * 4, 0u);
  if (random_operands) {
    uint32_t x = 0x9E3779B9u;
    for (auto& v : host) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v = x; }   // xorshift32
  }
  v4i* d_in = nullptr;
  int* d_out = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d_in), host.size() * 4) != hipSuccess) return QNNP_HIP_ENOMEM;
  if (hipMalloc(reinterpret_cast<void**>(&d_out), static_cast<size_t>(compute_units) * 512 * 4) != hipSuccess) {
    (void) hipFree(d_in);
    return QNNP_HIP_ENOMEM;
  }
  hipStream_t stream = nullptr;
  hipEvent_t e0, e1;
  bool ok = hipMemcpy(d_in, host.data(), host.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
            hipEventCreate(&e0) == hipSuccess;
  if (ok && hipEventCreate(&e1) != hipSuccess) { (void) hipEventDestroy(e0); ok = false; }
  float ms = 0.0f;
  const int reps = 10;
  if (ok) {
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(compute_units), dim3(512), 0, stream, d_in, d_out, iters);   // warm-up
    (void) hipEventRecord(e0, stream);
    for (int r = 0; r < reps; r++) {
      hipLaunchKernelGGL(mfma_probe_kernel, dim3(compute_units), dim3(512), 0, stream, d_in, d_out, iters);
    }
    ok = hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
         hipEventElapsedTime(&ms, e0, e1) == hipSuccess && hipGetLastError() == hipSuccess;
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
  }
  (void) hipFree(d_in);
  (void) hipFree(d_out);
  if (!ok || ms <= 0.0f) return QNNP_HIP_ELAUNCH;
  const double ops = static_cast<double>(compute_units) * 8.0 * iters * 8.0 * 65536.0 * reps;   // 8 waves x 8 MFMAs
  *tops_out = static_cast<float>(ops / (ms * 1e-3) / 1e12);
  return QNNP_HIP_OK;
}
