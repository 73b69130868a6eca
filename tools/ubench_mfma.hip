// Measurement tool (not part of the product): sustained v_mfma_i32_32x32x32_i8 rate of the whole chip with
// nothing else in the loop -- the practical ceiling the GEMM kernel is priced against (clock under MFMA load).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma.hip -o /tmp/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_ticks[2];

template <int NACC>
__global__ __launch_bounds__(512) void k(const v4i* in, int* out, int iters) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  v4i a[4], b[4];   // four different operand pairs in rotation: the multiplier inputs toggle like in a real GEMM
#pragma unroll
  for (int j = 0; j < 4; j++) { a[j] = in[threadIdx.x + j * 1024]; b[j] = in[threadIdx.x + j * 1024 + 512]; }
  v16i acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) acc[i][r] = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_ticks[0] = __builtin_readcyclecounter() - t0;   // s_memtime
    g_ticks[1] = wall_clock64() - w0;                 // s_memrealtime (100 MHz)
  }
}

template <int NACC>
void run(const char* name, int threads, const v4i* d_in, int* d_out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<256, threads>>>(d_in, d_out, 4);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; r++) k<NACC><<<256, threads>>>(d_in, d_out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  const double ops = 256.0 * (threads / 64) * iters * NACC * 65536.0;
  const double cyc = ms * 1e-3 * 2.4e9 / (double(iters) * NACC * (threads / 256));
  unsigned long long t[2]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), sizeof(t));
  printf("%-28s iters=%6d  %8.3f us  %7.1f TOPS  (%.1f cycles@2.4GHz per MFMA per SIMD)  s_memtime %.3f GHz (vs s_memrealtime@100MHz)\n", name, iters, ms * 1e3, ops / (ms * 1e-3) / 1e12, cyc, double(t[0]) / double(t[1]) * 0.1);
}

int main() {
  std::vector<int> h(4096 * 4);
  v4i* d_in; int* d_out;
  hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, 256 * 512 * 4);
  for (int pass = 0; pass < 2; pass++) {
    for (auto& x : h) x = pass == 0 ? 0 : (int) (rand() * 2654435761u);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("--- operands: %s\n", pass == 0 ? "zero" : "random");
    for (int iters : {128, 1280, 12800}) {
      run<8>("8 waves/CU x 8 acc", 512, d_in, d_out, iters);
      run<4>("8 waves/CU x 4 acc", 512, d_in, d_out, iters * 2);
    }
  }
  return 0;
}
