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

/*
 * qnnp_gfx950_copy_probe(): the HBM bandwidth this very chip gives a plain streaming kernel -- 16 bytes per lane,
 * grid-stride, read + write -- over buffers far larger than the 256 MiB Infinity Cache. SURVEY section 8(d) asks
 * for HBM fractions against "our own in-process copy kernel" beside the 8 TB/s specification and the guide's
 * 6.3 TB/s; bench.py reports it as roofline.hbm_copy_gbs / extra.*.frac_of_copy_kernel.
 */
namespace {

__global__ __launch_bounds__(256) void copy_probe_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  // four independent 16-byte loads in flight per lane
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

/* the same with the streaming (nt) policy on loads and stores: what a kernel that touches every line once can reach
 * (tools/ubench_copy.hip: +12 % over the plain form); bench.py reports BOTH and prices `frac_of_copy_kernel` against this one */
typedef unsigned int probe_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_probe_nt_kernel(const probe_u4* __restrict__ src, probe_u4* __restrict__ dst, size_t n16)
{
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const probe_u4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                   c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(a, dst + i); __builtin_nontemporal_store(b, dst + i + stride);
    __builtin_nontemporal_store(c, dst + i + 2 * stride); __builtin_nontemporal_store(d, dst + i + 3 * stride);
  }
  for (; i < n16; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

__global__ __launch_bounds__(256) void read_probe_kernel(const uint4* __restrict__ src, uint32_t* __restrict__ sink, size_t n16)
{
  const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
  size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  uint32_t x = 0;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    x ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; i < n16; i += stride) { const uint4 a = src[i]; x ^= a.x ^ a.y ^ a.z ^ a.w; }
  if (x == 0x12345678u) sink[0] = x;     // never true for the random fill; keeps the loads alive
}

}  // namespace

/* mode 0: copy (bytes read + bytes written counted), mode 1: read only, mode 2: copy with streaming (nt) loads and stores. `mbytes` per buffer (>= 512 recommended).
 * *gbs_out = GB/s of the best of `reps` event-timed launches. */
extern "C" int qnnp_gfx950_copy_probe(int mode, int mbytes, int reps, float* gbs_out)
{
  if (gbs_out == nullptr || mbytes <= 0 || reps <= 0) return QNNP_HIP_EINVAL;
  int device = 0;
  hipDeviceProp_t props;
  if (hipGetDevice(&device) != hipSuccess || hipGetDeviceProperties(&props, device) != hipSuccess) return QNNP_HIP_ENODEV;
  const size_t bytes = static_cast<size_t>(mbytes) << 20;
  const size_t n16 = bytes / 16;
  uint4* src = nullptr;
  uint4* dst = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&src), bytes) != hipSuccess) return QNNP_HIP_ENOMEM;
  if (hipMalloc(reinterpret_cast<void**>(&dst), mode != 1 ? bytes : 256) != hipSuccess) { (void) hipFree(src); return QNNP_HIP_ENOMEM; }
  (void) hipMemset(src, 0x5A, bytes);
  hipEvent_t e0, e1;
  bool ok = hipEventCreate(&e0) == hipSuccess;
  if (ok && hipEventCreate(&e1) != hipSuccess) { (void) hipEventDestroy(e0); ok = false; }
  float best = 0.0f;
  if (ok) {
    const dim3 grid(static_cast<unsigned>(props.multiProcessorCount) * 8u), block(256);
    for (int r = 0; r < reps + 1 && ok; r++) {
      (void) hipEventRecord(e0, nullptr);
      if (mode == 0) hipLaunchKernelGGL(copy_probe_kernel, grid, block, 0, nullptr, src, dst, n16);
      else if (mode == 2) hipLaunchKernelGGL(copy_probe_nt_kernel, grid, block, 0, nullptr, reinterpret_cast<const probe_u4*>(src), reinterpret_cast<probe_u4*>(dst), n16);
      else hipLaunchKernelGGL(read_probe_kernel, grid, block, 0, nullptr, src, reinterpret_cast<uint32_t*>(dst), n16);
      float ms = 0.0f;
      ok = hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
           hipEventElapsedTime(&ms, e0, e1) == hipSuccess && hipGetLastError() == hipSuccess && ms > 0.0f;
      if (ok && r > 0) {       // the first launch is the warm-up
        const float gbs = static_cast<float>((mode != 1 ? 2.0 : 1.0) * static_cast<double>(bytes) / (ms * 1e-3) / 1e9);
        if (gbs > best) best = gbs;
      }
    }
    (void) hipEventDestroy(e0);
    (void) hipEventDestroy(e1);
  }
  (void) hipFree(src);
  (void) hipFree(dst);
  if (!ok || best <= 0.0f) return QNNP_HIP_ELAUNCH;
  *gbs_out = best;
  return QNNP_HIP_OK;
}

/*
 * qnnp_gfx950_launch_floor_probe(): what a dependent chain of kernel launches costs on this box with NOTHING in the
 * kernels -- `kernels` empty launches of `blocks` x 256 threads captured as one hipGraph (stream order = a dependency
 * edge between consecutive nodes, as in the sweep / network graphs of bench.py), replayed `replays` times; microseconds
 * per launch of the best replay. The review of round 3 asked for this number beside the sweep's sum of layers.
 */
namespace {
__global__ __launch_bounds__(256) void empty_probe_kernel(uint32_t* sink)
{
  if (sink != nullptr && threadIdx.x == 0xFFFFu) sink[0] = 1u;       // never
}
}  // namespace

extern "C" int qnnp_gfx950_launch_floor_probe(int kernels, int blocks, int replays, float* us_per_launch)
{
  if (us_per_launch == nullptr || kernels <= 0 || kernels > 4096 || blocks <= 0 || replays <= 0) return QNNP_HIP_EINVAL;
  hipStream_t stream = nullptr;
  if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) return QNNP_HIP_ELAUNCH;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (ok) {
    for (int k = 0; k < kernels; k++) hipLaunchKernelGGL(empty_probe_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, nullptr);
    ok = hipStreamEndCapture(stream, &graph) == hipSuccess && graph != nullptr;
  }
  ok = ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
  ok = ok && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess;
  float best = 0.0f;
  for (int r = 0; ok && r < replays + 2; r++) {
    float ms = 0.0f;
    ok = hipEventRecord(e0, stream) == hipSuccess && hipGraphLaunch(exec, stream) == hipSuccess &&
         hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
         hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.0f;
    if (ok && r >= 2 && (best == 0.0f || ms < best)) best = ms;     // two warm-up replays
  }
  if (e0 != nullptr) (void) hipEventDestroy(e0);
  if (e1 != nullptr) (void) hipEventDestroy(e1);
  if (exec != nullptr) (void) hipGraphExecDestroy(exec);
  if (graph != nullptr) (void) hipGraphDestroy(graph);
  (void) hipStreamDestroy(stream);
  (void) hipGetLastError();
  if (!ok || best <= 0.0f) return QNNP_HIP_ELAUNCH;
  *us_per_launch = best * 1e3f / static_cast<float>(kernels);
  return QNNP_HIP_OK;
}
