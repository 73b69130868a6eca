// Measurement tool (not part of the product): issue cost of the integer VALU instructions the Q31
// requantization can be built from, on gfx950. Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(int* out, int iters, int a0, int b0) {
  int a = a0 + threadIdx.x, b = b0, c = threadIdx.x;
  long long acc = c;
  int lo = c, hi = c + 1;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP16(asm volatile("v_add_u32 %0, %1, %0" : "+v"(lo) : "v"(a));) }
    if (OP == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(lo) : "v"(a));) }
    if (OP == 2) { REP16(asm volatile("v_mul_hi_i32 %0, %1, %0" : "+v"(lo) : "v"(a));) }
    if (OP == 3) { REP16(asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");) }
    if (OP == 4) { REP16(asm volatile("v_mul_i32_i24 %0, %1, %0" : "+v"(lo) : "v"(a));) }
    if (OP == 5) { REP16(asm volatile("v_mul_hi_i32_i24 %0, %1, %0" : "+v"(lo) : "v"(a));) }
    if (OP == 6) { REP16(asm volatile("v_ashrrev_i64 %0, 5, %0" : "+v"(acc));) }
    if (OP == 7) { REP16(asm volatile("v_alignbit_b32 %0, %1, %0, 31" : "+v"(lo) : "v"(hi));) }
    if (OP == 8) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");) }
    if (OP == 9) { REP16(asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));) }
    if (OP == 10) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc) : "v"(acc));) }
    if (OP == 11) { REP16(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(acc) : "v"(lo));) }
    if (OP == 12) { double d; REP16(asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(acc));) }
    if (OP == 13) { REP16(asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));) }
    if (OP == 14) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(lo) : "v"(a), "v"(b));) }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = lo + hi + (int) acc + (int) (acc >> 32);
}
template <int OP> void run(const char* name, int* d) {
  const int iters = 2000, blocks = 256 * 8, threads = 256;   // 8 waves/SIMD worth of independent chains
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(d, 10, 3, 5);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, threads>>>(d, iters, 3, 5);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // 32 waves per CU (8 per SIMD); each wave issues iters*16 dependent ops; SIMD time-slices the waves.
  double wave_insts_per_simd = 8.0 * iters * 16;
  double ns_per_inst = ms * 1e6 / wave_insts_per_simd;
  printf("%-18s %8.3f ms  %6.2f ns per wave-instruction per SIMD (= %5.1f cycles @2.4GHz)\n", name, ms, ns_per_inst, ns_per_inst * 2.4);
}
int main() {
  int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  run<0>("v_add_u32", d); run<1>("v_mul_lo_u32", d); run<2>("v_mul_hi_i32", d); run<3>("v_mad_i64_i32", d);
  run<8>("v_mad_u64_u32", d); run<4>("v_mul_i32_i24", d); run<5>("v_mul_hi_i32_i24", d); run<6>("v_ashrrev_i64", d);
  run<7>("v_alignbit_b32", d); run<9>("v_med3_i32", d); run<10>("v_lshl_add_u64", d); run<11>("v_cvt_f64_i32", d);
  run<12>("v_fma_f64", d); run<13>("v_dot4c_i32_i8", d); run<14>("v_perm_b32", d);
  return 0;
}
