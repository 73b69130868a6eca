// Measurement tool (not part of the product): what each kind of filler instruction costs when it is placed
// between v_mfma_i32_32x32x32_i8 instructions of ONE wave per SIMD (the 4-wave GEMM flavour) or two.
// Each variant runs `iters` "phases" of 16 MFMAs plus its fillers; the print-out is shader cycles per phase
// (512 = the MFMA-bound floor). Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_gap.hip -o tools/ubench_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define PIN() __builtin_amdgcn_sched_barrier(0)

__device__ unsigned long long g_cycles[2];

__device__ __forceinline__ void dma16(const void* src, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*) src,
                                   (__attribute__((address_space(3))) void*) lds_dst, 16, 0, 0);
}

// V bits: 1 = 16 v_xor (4 per gap over the last 4 gaps)   2 = 16 v_dot4c in 4 chains   4 = 8 ds_read_b128 at the start
//         8 = 4 LDS-DMA pieces, one every 4 MFMAs          16 = s_barrier at the end     32 = dot4c in ONE chain
//         64 = xor spread 1 per gap over 16 gaps           128 = the DMA pieces use a 64-bit VGPR address add + 2 cndmask each
template <int V, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k(const v4i* in, const uint8_t* gsrc, int* out, int iters) {
  extern __shared__ uint8_t lds[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  v4i a[4], b[4], ra[4], rb[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { a[j] = in[threadIdx.x + j * 1024]; b[j] = in[threadIdx.x + j * 1024 + 512]; ra[j] = a[j]; rb[j] = b[j]; }
  for (uint32_t i = threadIdx.x; i < 16384; i += THREADS) reinterpret_cast<v4i*>(lds)[i % 4096] = in[i % 4096];
  __syncthreads();
  v16i acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0;
  int rs[4] = {0, 0, 0, 0};
  const uint8_t* g0 = gsrc + (size_t) blockIdx.x * 65536 + threadIdx.x * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    PIN();
    if (V & 4) {
      const uint8_t* base = lds + ((it & 3) * 16384) + wave * 1024 + lane * 16;
#pragma unroll
      for (int j = 0; j < 4; j++) ra[j] = *reinterpret_cast<const v4i*>(base + j * 4096);
#pragma unroll
      for (int j = 0; j < 4; j++) rb[j] = *reinterpret_cast<const v4i*>(base + j * 4096 + 2048);
    }
    PIN();
#pragma unroll
    for (int i = 0; i < 16; i++) {
      acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
      PIN();
      if ((V & 8) && (i & 3) == 0) {
        const uint8_t* src = g0 + ((it & 7) * 4 + (i >> 2)) * 4096;
        if (V & 128) {
          const uint32_t kk = it * 64 + lane;
          src = kk < 0x7fffffffu - (uint32_t) iters ? src : gsrc;
        }
        if (V & 1024) {  // uniform (SGPR) base + constant 32-bit lane offset
          const uint8_t* ub = gsrc + (size_t) blockIdx.x * 65536 + ((it & 7) * 4 + (i >> 2)) * 4096;
          src = ub + (uint32_t) (threadIdx.x * 16);
        }
        dma16(src, lds + 65536 + (i >> 2) * 4096 + wave * 1024);
        PIN();
      }
      if ((V & 64)) {
        const int j = i >> 2, c = i & 3;
        a[j][c] = ra[j][c] ^ 0x80808080; asm volatile("" : "+v"(a[j][c]));
        PIN();
      }
      if (i >= 12) {
        const int j = i - 12;
        if (V & 1) {
          a[j].x = ra[j].x ^ 0x80808080; a[j].y = ra[j].y ^ 0x80808080; a[j].z = ra[j].z ^ 0x80808080; a[j].w = ra[j].w ^ 0x80808080;
          asm volatile("" : "+v"(a[j]));
        }
        if (V & 2) {
          rs[0] = __builtin_amdgcn_sdot4(a[j].x, 0x01010101, rs[0], false);
          rs[1] = __builtin_amdgcn_sdot4(a[j].y, 0x01010101, rs[1], false);
          rs[2] = __builtin_amdgcn_sdot4(a[j].z, 0x01010101, rs[2], false);
          rs[3] = __builtin_amdgcn_sdot4(a[j].w, 0x01010101, rs[3], false);
        }
        if (V & 256) {   // row sum of the RAW bytes with v_sad_u8 (|a - 0| summed over 4 bytes + accumulator)
          rs[0] = __builtin_amdgcn_sad_u8(ra[j].x, 0, rs[0]);
          rs[1] = __builtin_amdgcn_sad_u8(ra[j].y, 0, rs[1]);
          rs[2] = __builtin_amdgcn_sad_u8(ra[j].z, 0, rs[2]);
          rs[3] = __builtin_amdgcn_sad_u8(ra[j].w, 0, rs[3]);
        }
        if (V & 512) {   // row sum with plain VOP2: (x & m) + ((x >> 8) & m) on packed halves, 4 ops per dword
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const uint32_t x = (uint32_t) ra[j][c];
            rs[c] += (int) ((x & 0x00FF00FFu) + ((x >> 8) & 0x00FF00FFu));
          }
        }
        if (V & 32) {
          rs[0] = __builtin_amdgcn_sdot4(a[j].x, 0x01010101, rs[0], false);
          rs[0] = __builtin_amdgcn_sdot4(a[j].y, 0x01010101, rs[0], false);
          rs[0] = __builtin_amdgcn_sdot4(a[j].z, 0x01010101, rs[0], false);
          rs[0] = __builtin_amdgcn_sdot4(a[j].w, 0x01010101, rs[0], false);
        }
        PIN();
      }
    }
    if (V & 4) { asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3])); 
#pragma unroll
      for (int j = 0; j < 4; j++) b[j] = rb[j]; }
    if ((V & 8) && !(V & 2048)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if ((V & 8) && (V & 2048)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // three phases of slack
    if (V & 16) __builtin_amdgcn_s_barrier();
    PIN();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  int s = rs[0] + rs[1] + rs[2] + rs[3];
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 7 && threadIdx.x == 0) g_cycles[0] = t1 - t0;
}

template <int V>
void run(const char* name, const v4i* d_in, const uint8_t* d_g, int* d_out) {
  const int iters = 256;
  k<V, 256><<<256, 256, 81920>>>(d_in, d_g, d_out, 8);
  hipDeviceSynchronize();
  for (int r = 0; r < 3; r++) k<V, 256><<<256, 256, 81920>>>(d_in, d_g, d_out, iters);
  hipDeviceSynchronize();
  unsigned long long t[2]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_cycles), sizeof(t));
  printf("%-58s %7.1f cycles per 16-MFMA phase (+%.0f over 512)\n", name, double(t[0]) / iters, double(t[0]) / iters - 512.0);
}

int main() {
  std::vector<int> h(4096 * 4);
  for (auto& x : h) x = (int) (rand() * 2654435761u);
  v4i* d_in; int* d_out; uint8_t* d_g;
  hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_g, 256 * 65536 + 65536);
  hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemset(d_g, 1, 256 * 65536 + 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
#define RUN(V, NAME) hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920); run<V>(NAME, d_in, d_g, d_out);
  RUN(0, "bare MFMAs")
  RUN(1, "+16 v_xor (4 per gap, last 4 gaps)")
  RUN(64, "+16 v_xor (1 per gap)")
  RUN(1 | 2, "+16 v_xor +16 v_dot4c in 4 chains")
  RUN(1 | 32, "+16 v_xor +16 v_dot4c in 1 chain")
  RUN(1 | 256, "+16 v_xor +16 v_sad_u8")
  RUN(1 | 512, "+16 v_xor +row sum with and/shift/add (64 VOP2)")
  RUN(8 | 1024, "+4 LDS-DMA pieces, SGPR base + 32-bit lane offset")
  RUN(4, "+8 ds_read_b128 at the phase start")
  RUN(8, "+4 LDS-DMA pieces (1 per 4 MFMAs)")
  RUN(8 | 128, "+4 LDS-DMA pieces with per-lane address select")
  RUN(16, "+s_barrier")
  RUN(4 | 16, "+8 ds_read +barrier")
  RUN(1 | 4 | 8 | 16, "+xor +ds_read +DMA +barrier")
  RUN(1 | 2 | 4 | 8 | 16, "+xor +dot4c(4 chains) +ds_read +DMA +barrier")
  RUN(1 | 256 | 4 | 8 | 16 | 1024, "+xor +sad_u8 +ds_read +DMA(sgpr base) +barrier")
  RUN(4 | 8, "+ds_read +DMA")
  RUN(4 | 8 | 2048, "+ds_read +DMA (vmcnt 12)")
  RUN(8 | 2048, "+DMA (vmcnt 12)")
  RUN(8 | 16 | 2048, "+DMA (vmcnt 12) +barrier")
  RUN(4 | 8 | 16 | 2048, "+ds_read +DMA (vmcnt 12) +barrier")
  RUN(1 | 4 | 8 | 16 | 2048, "+xor +ds_read +DMA (vmcnt 12) +barrier")
  RUN(1 | 256 | 4 | 8 | 16 | 2048, "+xor +sad +ds_read +DMA (vmcnt 12) +barrier")
  return 0;
}
