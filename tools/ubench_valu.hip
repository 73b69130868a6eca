// measurement tool: issue cost (cycles per wave-instruction) of the integer / fp64 instructions a Q31
// requantization could be built from. 1 and 4 waves per SIMD; 8 independent chains per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP 64
template <int OP>
__global__ void k(uint32_t* out, int iters, uint32_t seed)
{
  uint32_t a[8]; uint64_t w[8]; double d[8];
  for (int i = 0; i < 8; i++) { a[i] = seed * (threadIdx.x + 1) + i * 77; w[i] = a[i]; d[i] = a[i]; }
  uint32_t m = seed | 0x40000001u;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < REP / 8; r++) {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (OP == 0) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(m) : "vcc");
        if (OP == 1) asm volatile("v_mul_hi_i32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 2) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 3) asm volatile("v_mul_hi_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 4) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
        if (OP == 6) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        if (OP == 7) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 9) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(m) : "vcc");
        if (OP == 10) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 11) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
        if (OP == 12) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n\tv_subb_co_u32 %2, vcc, %2, %1, vcc" : "+v"(a[i]), "+v"(m), "+v"(a[(i + 1) & 7]) : : "vcc");
        if (OP == 13) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 14) asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(m));
        if (OP == 15) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(w[i]));
        if (OP == 16) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
        // ---- round 2: what a depthwise tap loop could be built from ----
        if (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 18) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 19) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 20) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 21) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(w[i]) : "v"(w[(i + 1) & 7]));
        if (OP == 22) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 23) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[i]) : "v"(m));
        if (OP == 24) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
        if (OP == 25) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 26) asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 27) asm volatile("v_pk_mad_i16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 28) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 29) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 30) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a[i]));
        if (OP == 31) asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(a[i]) : "v"(m));
        if (OP == 32) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(a[i]) : "v"(m));
        if (OP == 33) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 34) asm volatile("v_pk_fma_f16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
        if (OP == 35) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 36) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
        if (OP == 37) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
        if (OP == 38) asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(a[(i + 1) & 7]));
      }
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  uint32_t acc = 0;
  for (int i = 0; i < 8; i++) acc ^= a[i] ^ (uint32_t) w[i] ^ (uint32_t) (w[i] >> 32) ^ (uint32_t) d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[1 << 20] = (uint32_t) (t1 - t0); }
}

template <int OP>
void run(const char* name, uint32_t* d_out)
{
  for (int waves_per_simd : {1, 4}) {
    const int threads = 256;                      // 4 waves = 1 per SIMD
    const int blocks = 256 * waves_per_simd;      // one CU gets waves_per_simd blocks (roughly)
    const int iters = 200;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 12345u);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint32_t cyc; hipMemcpy(&cyc, d_out + (1 << 20), 4, hipMemcpyDeviceToHost);
    const double n_instr = (double) iters * REP * (OP == 12 ? 2 : 1);
    printf("%-22s waves/SIMD=%d  cycles/instr (block 0, own clock) = %6.2f   wall %.3f ms\n", name, waves_per_simd,
           cyc / n_instr, ms);
  }
}

int main()
{
  uint32_t* d_out; hipMalloc(&d_out, ((1 << 20) + 16) * 4);
  run<8>("v_add_u32", d_out);
  run<0>("v_mad_i64_i32", d_out);
  run<9>("v_mad_u64_u32", d_out);
  run<1>("v_mul_hi_i32", d_out);
  run<2>("v_mul_lo_u32", d_out);
  run<3>("v_mul_hi_i32_i24", d_out);
  run<4>("v_mul_i32_i24", d_out);
  run<5>("v_fma_f64", d_out);
  run<6>("v_cvt_f64_i32", d_out);
  run<7>("v_cvt_i32_f64", d_out);
  run<10>("v_add3_u32", d_out);
  run<11>("v_ashrrev_i32", d_out);
  run<12>("sub_co+subb_co (x2)", d_out);
  run<13>("v_cvt_pk_i16_i32", d_out);
  run<14>("v_pk_add_i16 clamp", d_out);
  run<15>("v_lshlrev_b64", d_out);
  run<16>("v_fma_f32", d_out);
  run<22>("v_fmac_f32 (VOP2)", d_out);
  run<36>("v_add_f32", d_out);
  run<21>("v_pk_fma_f32", d_out);
  run<34>("v_pk_fma_f16", d_out);
  run<35>("v_dot2_f32_f16", d_out);
  run<17>("v_perm_b32", d_out);
  run<18>("v_dot2_i32_i16", d_out);
  run<37>("v_dot2c_i32_i16", d_out);
  run<19>("v_dot4_i32_i8", d_out);
  run<38>("v_dot4c_i32_i8", d_out);
  run<20>("v_dot4_u32_u8", d_out);
  run<23>("v_cvt_f32_ubyte1", d_out);
  run<24>("v_cvt_i32_f32", d_out);
  run<25>("v_sad_u8", d_out);
  run<26>("v_mad_i32_i24", d_out);
  run<27>("v_pk_mad_i16", d_out);
  run<28>("v_and_b32", d_out);
  run<29>("v_lshl_add_u32", d_out);
  run<30>("v_bfe_u32", d_out);
  run<31>("v_mul_u32_u24_sdwa", d_out);
  run<32>("v_add_u32_sdwa", d_out);
  run<33>("v_med3_i32", d_out);
  return 0;
}
