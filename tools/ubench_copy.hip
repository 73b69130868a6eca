// Measurement tool (not part of the product): what the cache-policy bits of gfx950 loads / stores do to a plain
// streaming copy (16 bytes per lane, grid-stride over 512 MiB) and to a read-only / write-only stream.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_copy.hip -o tools/ubench_copy && tools/ubench_copy
// Variants: load policy x store policy, each in {plain, nt (__builtin_nontemporal_*)}; plus the raw-buffer forms with
// the aux bits spelled out (bit 0 = sc0/glc, bit 1 = slc/nt, bit 4 = sc1 on gfx940+).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int LD, int ST>
__global__ __launch_bounds__(256) void copy_k(const u4* __restrict__ src, u4* __restrict__ dst, size_t n)
{
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u4 v;
    if (LD == 1) v = __builtin_nontemporal_load(src + i); else v = src[i];
    if (ST == 1) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

template <int AUXL, int AUXS>
__global__ __launch_bounds__(256) void copy_buf_k(const u4* src, u4* dst, size_t n)
{
  // 512 MiB do not fit one 32-bit-offset descriptor range comfortably with a grid-stride walk: each block owns a
  // contiguous 1 MiB-aligned slab and rebases its descriptors on it
  const size_t per_block = n / gridDim.x;                 // vectors per block (launcher: divisible)
  const u4* s0 = src + (size_t) blockIdx.x * per_block;
  u4* d0 = dst + (size_t) blockIdx.x * per_block;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4*>(s0), 0, (int) (per_block * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(d0, 0, (int) (per_block * 16), 0x00020000);
  for (uint32_t i = threadIdx.x; i < per_block; i += blockDim.x) {
    const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, i * 16u, 0, AUXL);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, i * 16u, 0, AUXS);
  }
}

template <int LD>
__global__ __launch_bounds__(256) void read_k(const u4* __restrict__ src, unsigned* out, size_t n)
{
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  u4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    u4 v;
    if (LD == 1) v = __builtin_nontemporal_load(src + i); else v = src[i];
    acc ^= v;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int ST>
__global__ __launch_bounds__(256) void write_k(u4* __restrict__ dst, size_t n)
{
  const size_t stride = (size_t) gridDim.x * blockDim.x;
  const u4 v = {1u, 2u, 3u, (unsigned) threadIdx.x};
  for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (ST == 1) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

template <typename F>
static double time_ms(F&& launch, int reps)
{
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> t;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); t.push_back(ms);
  }
  std::sort(t.begin(), t.end());
  return t[t.size() / 2];
}

int main()
{
  const size_t bytes = 512ull << 20, n = bytes / 16;
  u4 *src, *dst; unsigned* out;
  hipMalloc(&src, bytes); hipMalloc(&dst, bytes); hipMalloc(&out, 64);
  hipMemset(src, 0x5a, bytes); hipMemset(dst, 0, bytes);
  for (int blocks : {2048, 8192}) {
    printf("== %d blocks of 256 threads, 512 MiB in + 512 MiB out\n", blocks);
#define RUN(name, expr, traffic) do { for (int w = 0; w < 3; w++) { expr; } hipDeviceSynchronize(); \
    const double ms = time_ms([&] { expr; }, 15); printf("  %-40s %8.3f ms  %7.1f GB/s\n", name, ms, (traffic) / ms / 1e6); } while (0)
    RUN("copy  plain load, plain store", (copy_k<0, 0><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  nt load,    plain store", (copy_k<1, 0><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  plain load, nt store", (copy_k<0, 1><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  nt load,    nt store", (copy_k<1, 1><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  buffer aux 0 / 0", (copy_buf_k<0, 0><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  buffer aux 2 / 2 (nt)", (copy_buf_k<2, 2><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  buffer aux 3 / 3 (sc0 nt)", (copy_buf_k<3, 3><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  buffer aux 16 / 16 (sc1)", (copy_buf_k<16, 16><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  buffer aux 18 / 18 (sc1 nt)", (copy_buf_k<18, 18><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("copy  buffer aux 0 / 18", (copy_buf_k<0, 18><<<blocks, 256>>>(src, dst, n)), 2.0 * bytes);
    RUN("read  plain", (read_k<0><<<blocks, 256>>>(src, out, n)), 1.0 * bytes);
    RUN("read  nt", (read_k<1><<<blocks, 256>>>(src, out, n)), 1.0 * bytes);
    RUN("write plain", (write_k<0><<<blocks, 256>>>(dst, n)), 1.0 * bytes);
    RUN("write nt", (write_k<1><<<blocks, 256>>>(dst, n)), 1.0 * bytes);
  }
  return 0;
}
