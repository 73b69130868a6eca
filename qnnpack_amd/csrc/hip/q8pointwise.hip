/*
 * q8pointwise.hip -- the two byte-streaming operators between MobileNetV2's convolutions: quantized
 * element-wise add (residual connections) and global average pooling. Both are HBM-bound by construction
 * (3 bytes moved per add output; width bytes per pooled output), so the kernels are about coalescing:
 * 16-byte vectors per lane along the channel axis, grid-stride loops, no LDS.
 *
 * add          replaces q8vadd_ukernel__sse2 (reference src/q8vadd/sse2.c) and the add case of
 *              qnnp_run_operator; arithmetic = qnnp_add_quantize (src/qnnpack/requantization.h:500-522).
 * global avg   replaces q8gavgpool_ukernel_up8x7/mp8x7p7q/up8xm__sse2 (reference src/q8gavgpool/) and
 *              compute_global_average_pooling_unipass/multipass (src/operator-run.c:404-452); arithmetic =
 *              qnnp_avgpool_quantize (src/qnnpack/requantization.h:482-498). The reference needs three
 *              microkernels and a scratch row because an SSE2 register holds 8 channels of one pixel; here a
 *              lane keeps its channels' int32 sums in registers for the whole image.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "add_math.hip.h"
#include "qnnp_hip.h"

namespace qnnp {

namespace {

constexpr int kThreads = 256;

/* dense, 16-byte aligned tensors: one flat run of `vectors` uint4 */
__global__ __launch_bounds__(kThreads)
void q8_vadd_flat_kernel(const uint4* __restrict__ a, const uint4* b, uint4* sum,       // (b may BE sum: the in-place residual add, operator-run.c)
                         const uint64_t vectors, const qnnp_hip_add_params q, const uint32_t streaming)
{
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
  for (uint64_t v = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; v < vectors; v += stride) {
    // a pure stream -- every line read or written exactly once: streaming hints (a plain copy kernel gains 12 % from
    // them, tools/ubench_copy.hip; kernels that touch their lines again lose)
    typedef unsigned int nt_v4u __attribute__((ext_vector_type(4)));
    const nt_v4u x = __builtin_nontemporal_load(reinterpret_cast<const nt_v4u*>(a + v));
    const nt_v4u y = __builtin_nontemporal_load(reinterpret_cast<const nt_v4u*>(b + v));
    nt_v4u r;
    r.x = add_quantize4(x.x, y.x, q);
    r.y = add_quantize4(x.y, y.y, q);
    r.z = add_quantize4(x.z, y.z, q);
    r.w = add_quantize4(x.w, y.w, q);
    if (streaming) {                                                                        // ("streaming_stores")
      // (as an instruction: with the builtin, hipcc merges the two stores of this branch and drops the hint)
      asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(sum + v), "v"(r) : "memory");
    } else {
      *reinterpret_cast<nt_v4u*>(sum + v) = r;
    }
  }
}

/* any strides / alignment: one byte per lane, channel fastest */
__global__ __launch_bounds__(kThreads)
void q8_vadd_strided_kernel(const qnnp_hip_vadd_args p)
{
  const uint64_t total = p.rows * p.channels;
  const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kThreads;
  for (uint64_t e = static_cast<uint64_t>(blockIdx.x) * kThreads + threadIdx.x; e < total; e += stride) {
    const uint64_t row = e / p.channels;
    const uint64_t c = e - row * p.channels;
    p.sum[row * p.sum_stride + c] =
        static_cast<uint8_t>(add_quantize(p.a[row * p.a_stride + c], p.b[row * p.b_stride + c], p.params));
  }
}

/* src/qnnpack/requantization.h:482-498 */
__device__ __forceinline__ uint32_t avgpool_quantize(int32_t n, const qnnp_hip_avgpool_params& q)
{
  const int64_t product = static_cast<int64_t>(n) * static_cast<int64_t>(q.multiplier);
  const int64_t adjusted = product - static_cast<int64_t>(n < 0);
  int32_t y = static_cast<int32_t>((adjusted + q.rounding) >> q.right_shift);
  y = y < q.output_min_less_zero_point ? q.output_min_less_zero_point : y;
  y = y > q.output_max_less_zero_point ? q.output_max_less_zero_point : y;
  return static_cast<uint32_t>(y + q.output_zero_point) & 0xFFu;
}

/*
 * Global average pooling, VEC channels per lane (4: dword loads, channels % 4 == 0 and 4-byte aligned
 * strides / pointers; 1: anything). Workgroup = (image, chunk of kThreads*VEC channels) x kSplit pixel
 * slices reduced through LDS when the image has many pixels: a 7x7x1280 image gives only 320 lanes of work
 * per image otherwise.
 */
template <int VEC>
__global__ __launch_bounds__(kThreads)
void q8_gavgpool_kernel(const qnnp_hip_gavgpool_args p, const uint32_t chunks, const uint32_t lanes_per_chunk,
                        const uint32_t split)
{
  __shared__ int32_t partial[kThreads * 4];
  const uint32_t lane_c = threadIdx.x % lanes_per_chunk;          // channel vector inside the chunk
  const uint32_t slice = threadIdx.x / lanes_per_chunk;           // pixel slice
  const uint64_t image = blockIdx.x / chunks;
  const uint32_t chunk = blockIdx.x - image * chunks;
  const uint32_t c = (chunk * lanes_per_chunk + lane_c) * VEC;
  const bool live = c < p.channels && slice < split;

  int32_t acc[VEC];
#pragma unroll
  for (int i = 0; i < VEC; i++) acc[i] = 0;
  if (live) {
    const uint8_t* px = p.input + image * p.width * p.input_stride + c;
    uint64_t w = slice;
    if constexpr (VEC == 4) {
      // four pixels in flight per lane: the loop is a chain of dependent-latency loads otherwise
      for (; w + 3 * static_cast<uint64_t>(split) < p.width; w += 4 * static_cast<uint64_t>(split)) {
        uint32_t x[4];
#pragma unroll
        for (int u = 0; u < 4; u++) x[u] = *reinterpret_cast<const uint32_t*>(px + (w + u * static_cast<uint64_t>(split)) * p.input_stride);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          acc[0] += x[u] & 0xFFu;
          acc[1] += (x[u] >> 8) & 0xFFu;
          acc[2] += (x[u] >> 16) & 0xFFu;
          acc[3] += x[u] >> 24;
        }
      }
    }
    for (; w < p.width; w += split) {
      if constexpr (VEC == 4) {
        const uint32_t x = *reinterpret_cast<const uint32_t*>(px + w * p.input_stride);
        acc[0] += x & 0xFFu;
        acc[1] += (x >> 8) & 0xFFu;
        acc[2] += (x >> 16) & 0xFFu;
        acc[3] += x >> 24;
      } else {
        acc[0] += px[w * p.input_stride];
      }
    }
  }
  if (split > 1) {
#pragma unroll
    for (int i = 0; i < VEC; i++) partial[threadIdx.x * VEC + i] = acc[i];
    __syncthreads();
    if (live && slice == 0) {
      for (uint32_t s = 1; s < split; s++) {
#pragma unroll
        for (int i = 0; i < VEC; i++) acc[i] += partial[(s * lanes_per_chunk + lane_c) * VEC + i];
      }
    }
  }
  if (live && slice == 0) {
    uint8_t* out = p.output + image * p.output_stride + c;
    if constexpr (VEC == 4) {
      uint32_t r = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) r |= avgpool_quantize(acc[i] + p.params.bias, p.params) << (8 * i);
      *reinterpret_cast<uint32_t*>(out) = r;
    } else {
      out[0] = static_cast<uint8_t>(avgpool_quantize(acc[0] + p.params.bias, p.params));
    }
  }
}

inline uint32_t grid_for(uint64_t items, uint32_t cus)
{
  const uint64_t blocks = (items + kThreads - 1) / kThreads;
  const uint64_t cap = static_cast<uint64_t>(cus) * 8u;
  return static_cast<uint32_t>(blocks < cap ? (blocks ? blocks : 1) : cap);
}

inline uint32_t device_cus()
{
  const int cus = qnnp_hip_compute_units();   // of the active device context
  return cus > 0 ? static_cast<uint32_t>(cus) : 256u;
}

inline bool aligned(const void* p, uintptr_t a) { return reinterpret_cast<uintptr_t>(p) % a == 0; }

}  // namespace

}  // namespace qnnp

extern "C" int qnnp_hip_vadd_run(const struct qnnp_hip_vadd_args* a, const char** kernel_name)
{
  using namespace qnnp;
  if (a == nullptr || a->a == nullptr || a->b == nullptr || a->sum == nullptr || a->channels == 0) return QNNP_HIP_EINVAL;
  if (a->rows == 0) return QNNP_HIP_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(qnnp_hip_get_stream());
  const bool dense = a->a_stride == a->channels && a->b_stride == a->channels && a->sum_stride == a->channels;
  const uint64_t bytes = a->rows * a->channels;
  if (dense && bytes % 16 == 0 && aligned(a->a, 16) && aligned(a->b, 16) && aligned(a->sum, 16)) {
    const uint64_t vectors = bytes / 16;
    hipLaunchKernelGGL(q8_vadd_flat_kernel, dim3(grid_for(vectors, device_cus())), dim3(kThreads), 0, stream,
                       reinterpret_cast<const uint4*>(a->a), reinterpret_cast<const uint4*>(a->b),
                       reinterpret_cast<uint4*>(a->sum), vectors, a->params,
                       a->streaming_mode == 0 ? (qnnp_hip_streaming_stores() != 0 ? 1u : 0u) : (a->streaming_mode == 2 ? 1u : 0u));
    if (kernel_name != nullptr) *kernel_name = "q8_vadd_flat";
  } else {
    hipLaunchKernelGGL(q8_vadd_strided_kernel, dim3(grid_for(bytes, device_cus())), dim3(kThreads), 0, stream, *a);
    if (kernel_name != nullptr) *kernel_name = "q8_vadd_strided";
  }
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

extern "C" int qnnp_hip_gavgpool_run(const struct qnnp_hip_gavgpool_args* a, const char** kernel_name)
{
  using namespace qnnp;
  if (a == nullptr || a->input == nullptr || a->output == nullptr || a->channels == 0 || a->width == 0) return QNNP_HIP_EINVAL;
  if (a->batch == 0) return QNNP_HIP_OK;
  hipStream_t stream = reinterpret_cast<hipStream_t>(qnnp_hip_get_stream());
  const bool vec4 = a->channels % 4 == 0 && a->input_stride % 4 == 0 && a->output_stride % 4 == 0 &&
      aligned(a->input, 4) && aligned(a->output, 4);
  const uint32_t vec = vec4 ? 4u : 1u;
  const uint32_t cvecs = (a->channels + vec - 1) / vec;              // channel vectors per image
  // lanes of a workgroup: [split pixel slices][lanes_per_chunk channel vectors], split * lanes_per_chunk <= kThreads
  uint32_t lanes_per_chunk = kThreads;
  while (lanes_per_chunk > 32 && lanes_per_chunk / 2 >= cvecs) lanes_per_chunk /= 2;
  if (cvecs > static_cast<uint32_t>(kThreads)) lanes_per_chunk = 64;   // wide images: more, narrower workgroups
  uint32_t split = kThreads / lanes_per_chunk;
  if (split > a->width) split = static_cast<uint32_t>(a->width);
  const uint32_t chunks = (cvecs + lanes_per_chunk - 1) / lanes_per_chunk;
  const uint64_t blocks = a->batch * chunks;
  if (blocks > 0x7FFFFFFFull) return QNNP_HIP_EINVAL;
  if (vec4) {
    hipLaunchKernelGGL(q8_gavgpool_kernel<4>, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), 0, stream, *a,
                       chunks, lanes_per_chunk, split);
    if (kernel_name != nullptr) *kernel_name = "q8_gavgpool_x4";
  } else {
    hipLaunchKernelGGL(q8_gavgpool_kernel<1>, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), 0, stream, *a,
                       chunks, lanes_per_chunk, split);
    if (kernel_name != nullptr) *kernel_name = "q8_gavgpool_x1";
  }
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}
