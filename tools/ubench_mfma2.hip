// Measurement tool (not part of the product), round 6: what the int8 matrix pipe of a power-limited MI355X gives for
// DIFFERENT ways of feeding it the same random operands -- the q8gemm kernel is energy-bound, so the
// question is energy per MAC, read off as sustained TOP/s of the whole chip:
//   shape   : v_mfma_i32_32x32x32_i8 against v_mfma_i32_16x16x64_i8 (half the accumulator traffic per MAC, twice the operand reads)
//   accinit : accumulators that start at 0 (partial sums cross zero: the upper accumulator bits toggle) against 2^31-offset ones
//             (the offset forms of requant_math.h: what q8gemm256c.hip runs)
//   waves   : 8 waves / CU x 8 accumulator tiles against 4 waves / CU x 16 tiles (one wave per SIMD: 128 x 128 wave tiles)
//   order   : operand B fixed for 4 consecutive MFMAs (the GEMM loop's order) against both operands changing every MFMA
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma2.hip -o /tmp/ubench_mfma2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ unsigned long long g_ticks[2];

// operand rotation: ORDER 0 = b held for 4 MFMAs, a rotates; 1 = both change every MFMA; 2 = snake (b held for 4, a walks 0123 3210 ...:
// only ONE operand changes per step); 3 = b held for 8 MFMAs, a rotates
template <int ORDER> __device__ constexpr int order_a(int i) { return ORDER == 2 ? (((i >> 2) & 1) ? 3 - (i & 3) : (i & 3)) : (i & 3); }
template <int ORDER> __device__ constexpr int order_b(int i) { return ORDER == 1 ? (i + (i >> 2)) & 3 : (ORDER == 3 ? (i >> 3) & 3 : (i >> 2) & 3); }

// SHAPE 0: 32x32x32 (16 acc regs), 1: 16x16x64 (4 acc regs; NACC counts 32x32-equivalents: 4 small tiles each, 2 MFMAs per tile per step
// so that one "step" is the same 32768 MACs). ORDER 0: b fixed over 4 MFMAs, 1: both rotate.
template <int NACC, int SHAPE, int ORDER, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const v4i* in, int* out, int iters, int acc0) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long w0 = wall_clock64();
  v4i a[4], b[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { a[j] = in[threadIdx.x + j * 1024]; b[j] = in[threadIdx.x + j * 1024 + 512]; }
  int s = 0;
  if constexpr (SHAPE == 0) {
    v16i acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][r] = acc0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) {
        const int ia = order_a<ORDER>(i), ib = order_b<ORDER>(i);
        acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[ia], b[ib], acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NACC; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) s += acc[i][r];
  } else {
    v4i acc[NACC * 4];
#pragma unroll
    for (int i = 0; i < NACC * 4; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][r] = acc0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC * 4; i++) {
        // a 32x32 block over K = 32 == four 16x16 tiles over K = 32; with K = 64 per instruction: two instructions per tile per
        // TWO steps -> per step, 2 instructions for each of ... keep it simple: NACC*4 tiles x 1 instruction x K=64 = NACC x 2 x 32768 MACs / 2
        const int ia = order_a<ORDER>(i), ib = order_b<ORDER>(i);
        acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[ia], b[ib], acc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NACC * 4; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) s += acc[i][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    g_ticks[0] = __builtin_readcyclecounter() - t0;
    g_ticks[1] = wall_clock64() - w0;
  }
}

template <int NACC, int SHAPE, int ORDER, int THREADS>
void run(const char* name, const v4i* d_in, int* d_out, int iters, int acc0) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC, SHAPE, ORDER, THREADS><<<256, THREADS>>>(d_in, d_out, 4, acc0);
  hipDeviceSynchronize();
  // sustained state: ~0.6 s of back-to-back launches before the timed ones
  for (int r = 0; r < 150; r++) k<NACC, SHAPE, ORDER, THREADS><<<256, THREADS>>>(d_in, d_out, iters, acc0);
  const int reps = 100;
  hipEventRecord(e0);
  for (int r = 0; r < reps; r++) k<NACC, SHAPE, ORDER, THREADS><<<256, THREADS>>>(d_in, d_out, iters, acc0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  // MACs per instruction: 32768 (32x32x32) or 16384 (16x16x64); instructions per iteration and wave: NACC or NACC * 4
  const double macs = 256.0 * (THREADS / 64) * double(iters) * (SHAPE == 0 ? NACC * 32768.0 : NACC * 4 * 16384.0);
  unsigned long long t[2]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), sizeof(t));
  printf("%-64s acc0=%08x  %9.3f us  %7.1f TOPS  clock %.3f GHz\n", name, (unsigned) acc0, ms * 1e3, 2.0 * macs / (ms * 1e-3) / 1e12,
         double(t[0]) / double(t[1]) * 0.1);
  fflush(stdout);
}

int main() {
  std::vector<int> h(4096 * 4);
  v4i* d_in; int* d_out;
  hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_out, 256 * 512 * 4);
  for (int pass = 0; pass < 2; pass++) {
    srand(12345);
    for (auto& x : h) x = pass == 0 ? 0 : (int) (rand() * 2654435761u);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("--- operands: %s\n", pass == 0 ? "zero" : "random");
    const int it = 3200;   // ~50 us per launch at the random-operand rate
    for (int round = 0; round < 2; round++) {
      for (int acc0 : {(int) 0x80000000u}) {
        run<8, 0, 0, 512>("32x32x32  8 waves/CU x 8 tiles   b fixed over 4", d_in, d_out, it, acc0);
        run<8, 0, 1, 512>("32x32x32  8 waves/CU x 8 tiles   both operands rotate", d_in, d_out, it, acc0);
        run<16, 0, 0, 256>("32x32x32  4 waves/CU x 16 tiles  b fixed over 4", d_in, d_out, it, acc0);
        run<8, 1, 0, 512>("16x16x64  8 waves/CU x 32 tiles  b fixed over 4", d_in, d_out, it / 2, acc0);
        run<8, 1, 1, 512>("16x16x64  8 waves/CU x 32 tiles  both operands rotate", d_in, d_out, it / 2, acc0);
        run<8, 1, 2, 512>("16x16x64  8 waves/CU x 32 tiles  snake: one operand changes per step", d_in, d_out, it / 2, acc0);
        run<8, 1, 3, 512>("16x16x64  8 waves/CU x 32 tiles  b fixed over 8", d_in, d_out, it / 2, acc0);
      }
    }
  }
  return 0;
}
