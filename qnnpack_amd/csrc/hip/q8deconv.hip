/*
 * q8deconv.hip -- stride-2 deconvolution (transposed convolution) with 3x3 / 4x4 kernels as ONE streaming MFMA
 * kernel over the input pixels. Replaces, for these shapes, the reference's deconvolution run path
 * (src/operator-run.c:805-844 with the indirection buffer of src/indirection.c:129-208 and the q8conv microkernel,
 * src/deconvolution.c:213-277): there every output pixel gathers KH*KW taps through the indirection buffer, three in
 * four of them padding at stride 2.
 *
 * Geometry (deconvolution.c, "phases"): output pixel oy belongs to phase py = (oy + pad_top) % 2 and sees only the
 * taps ky = py + 2j; its tap j reads input row (oy + pad_top - py)/2 - j. With the BASE position
 *     by = (oy + pad_top - py) / 2        (likewise bx)
 * the four output pixels (2by + py - pad_top, 2bx + px - pad_left), py, px in {0, 1}, read nothing but the 2x2 input
 * neighbourhood {by, by-1} x {bx, bx-1}. So:
 *   unit   = 32 consecutive base positions of the flattened (image, by, bx) grid, one per MFMA column (B operand);
 *   loads  = the 2x2 neighbourhood's pixels, C bytes each, ONCE per unit (pixels outside the image read the input
 *            zero point: a padding tap, as the reference's zero buffer);
 *   math   = per phase an implicit GEMM over its (1 | 2 | 4 taps) x C reduction against the phase's own packed
 *            sub-kernel (deconvolution.c packs one per phase; all of them sit in LDS), Q31 requantization
 *            (igemm_epilogue.hip.h), one 16-byte store per lane per 32 channels -- at most four output pixels per base.
 * Against the phase-table GEMMs of the generic kernel: the input is read once instead of once per tap and phase, no
 * offset / output-row tables are read at all, and the four phases share one launch's ramp and tail.
 * Bytes are identical to the generic path by construction: the same packed weights, folded biases and epilogue.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "igemm_epilogue.hip.h"
#include "igemm_params.h"
#include "per_device.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kWaves = 8;                     // per workgroup: the sub-kernels are staged once per eight units
constexpr int kThreads = kWaves * 64;
// waves per SIMD the kernel is compiled for (hipcc: the second __launch_bounds__ figure): four (<= 128 VGPRs, two
// workgroups per CU) up to 64 channels, where the layers this kernel exists for (2x upsampling of a few thousand units)
// then fit the chip in ONE round of waves -- 841 workgroups of four waves at three per CU ran two rounds, 13.8 us
// against the 6-7 us of a single pass
constexpr int resident(int cb) { return cb <= 2 ? 4 : 2; }
constexpr uint32_t kFlip = 0x80808080u;
constexpr uint32_t kMaxLds = 64 * 1024;

struct DeconvParams {
  const uint8_t* input;
  uint8_t* output;
  const int8_t* w[4];          // phase py*2 + px: packed sub-kernel (MFMA fragment panels, k = tap-major)
  const int32_t* bias[4];      // folded bias of the phase, [n_pad]
  uint32_t kblocks[4];         // fragment blocks per channel block in the phase's packed image (k_pad / 32)
  uint32_t lds_w[4];           // byte offset of the phase's fragments in LDS: [nb][taps * CB] KiB
  uint32_t lds_bias;           // then 4 x n_pad int32
  uint32_t batch, H, W, OH, OW, BH, BW;
  uint32_t pad_top, pad_left;
  uint32_t n, n_pad, in_stride, out_stride;
  int32_t row_coeff;
  uint32_t store_mode;
  const uint8_t* fill;         // 16 bytes of the input zero point
  RequantDev rq;
  qnnp_requant_lane lane;      // lane forms of the requantization (requant.hip.h)
};

constexpr int taps_of(int k, int phase) { return (k - phase + 1) / 2; }     // taps ky = phase, phase + 2, ... < k

/* CB = 32-channel blocks of the input pixel (C = 32 * CB); KH, KW in {3, 4}; SEQ / FULL: the requantization flavour
 * (requant.hip.h), chosen on the host -- one kernel per flavour, so that the common ones are not charged the registers
 * of the rare ones */
template <int CB, int KH, int KW, int SEQ, bool FULL>
__global__ __launch_bounds__(kThreads, resident(CB))
void q8_deconv_s2_stream_kernel(const DeconvParams p)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t col = lane & 31u;
  const uint32_t khalf = lane >> 5;
  const uint32_t nblocks = p.n_pad / 32;

  // ---- once per workgroup: the four sub-kernels and their biases -> LDS (LDS-DMA, all in flight at once) ----
#pragma unroll
  for (int ph = 0; ph < 4; ph++) {
    const uint32_t kbp = static_cast<uint32_t>(taps_of(KH, ph >> 1) * taps_of(KW, ph & 1) * CB);
    const uint32_t frags = nblocks * kbp;
    for (uint32_t f = wave; f < frags; f += kWaves) {
      const uint32_t nb = f / kbp;
      const uint32_t kb = f - nb * kbp;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (p.w[ph] + (static_cast<uint64_t>(nb) * p.kblocks[ph] + kb) * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*) (lds + p.lds_w[ph] + f * 1024), 16, 0, 0);
    }
    const uint32_t bias_chunks = p.n_pad / 4;
    // (lane forms of the requantization: bias + 2^31, the second half of the phase's pair table)
    const int32_t* bias_src = p.bias[ph] + (rq_is_lane<SEQ>() ? p.n_pad : 0u);
    for (uint32_t c0 = wave * 64; c0 < bias_chunks; c0 += kThreads) {
      // (only the lanes that have a chunk: an LDS-DMA lane writes its 16 bytes at base + lane * 16 whatever it read, and
      //  a phase's bias line is n_pad * 4 bytes -- the surplus lanes of a clamped load spilled into the next phase's line)
      if (c0 + lane < bias_chunks) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*) (reinterpret_cast<const uint8_t*>(bias_src) + (c0 + lane) * 16),
            (__attribute__((address_space(3))) void*) (lds + p.lds_bias + ph * p.n_pad * 4 + c0 * 16), 16, 0, 0);
      }
    }
  }

  IgemmParams sp{};                         // what igemm_store_tile reads
  sp.lane = p.lane;
  sp.rq = p.rq;
  sp.n = p.n;
  sp.store_mode = p.store_mode;

  const uint32_t per_image = p.BH * p.BW;
  const uint32_t total = p.batch * per_image;
  const uint32_t units = (total + 31u) / 32u;
  const uint32_t unit_stride = gridDim.x * kWaves;

  // the 2x2 neighbourhood of a unit: a[j][i][c] = 16 bytes (this lane's K half) of channel block c of input pixel
  // (by - j, bx - i)
  auto load_unit = [&](uint32_t unit, v4i (&a)[2][2][CB], uint32_t& img, uint32_t& by, uint32_t& bx, bool& valid)
      __attribute__((always_inline)) {
    uint32_t m = unit * 32u + col;
    valid = m < total;
    if (!valid) m = total - 1u;
    img = m / per_image;
    const uint32_t r = m - img * per_image;
    by = r / p.BW;
    bx = r - by * p.BW;
#pragma unroll
    for (int j = 0; j < 2; j++) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const uint32_t iy = by - j, ix = bx - i;                         // (wraps below zero: fails the range test)
        const bool inside = iy < p.H && ix < p.W;
        const uint8_t* px = p.input + (((img * p.H + (inside ? iy : 0u)) * p.W + (inside ? ix : 0u)) * p.in_stride + khalf * 16);
#pragma unroll
        for (int c = 0; c < CB; c++) {
          a[j][i][c] = *reinterpret_cast<const v4i*>(inside ? px + c * 32 : p.fill);
        }
      }
    }
  };

  uint32_t unit = blockIdx.x * kWaves + wave;
  v4i a[2][2][CB];
  uint32_t img = 0, by = 0, bx = 0;
  bool valid = false;
  if (unit < units) load_unit(unit, a, img, by, bx, valid);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // weights + biases are in LDS (and the first pixels landed)
  __syncthreads();

  {
    while (unit < units) {
      // per-pixel byte sums (the kernel-zero-point row term), then re-centre at 128
      uint32_t rs[2][2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
          uint32_t s = 0;
#pragma unroll
          for (int c = 0; c < CB; c++) {
            s = __builtin_amdgcn_sad_u8(a[j][i][c].x, 0u, s);
            s = __builtin_amdgcn_sad_u8(a[j][i][c].y, 0u, s);
            s = __builtin_amdgcn_sad_u8(a[j][i][c].z, 0u, s);
            s = __builtin_amdgcn_sad_u8(a[j][i][c].w, 0u, s);
            a[j][i][c].x ^= static_cast<int>(kFlip);
            a[j][i][c].y ^= static_cast<int>(kFlip);
            a[j][i][c].z ^= static_cast<int>(kFlip);
            a[j][i][c].w ^= static_cast<int>(kFlip);
          }
          rs[j][i] = s + __shfl_xor(s, 32);                   // the other K half of the same pixel
        }
      }
      const uint32_t cur_img = img, cur_by = by, cur_bx = bx;
      const bool cur_valid = valid;

#pragma unroll
      for (int ph = 0; ph < 4; ph++) {
        constexpr int dummy = 0; (void) dummy;
        const int py = ph >> 1, px = ph & 1;
        const int ny = taps_of(KH, py), nx = taps_of(KW, px);
        const uint32_t oy = 2u * cur_by + py - p.pad_top;      // (wraps below zero: fails the range test)
        const uint32_t ox = 2u * cur_bx + px - p.pad_left;
        const bool ok = cur_valid && oy < p.OH && ox < p.OW;
        // (launcher: the output tensor is addressable with 32-bit byte offsets)
        uint8_t* out_row = p.output + ((cur_img * p.OH + (ok ? oy : 0u)) * p.OW + (ok ? ox : 0u)) * p.out_stride;
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int i = 0; i < 2; i++)
            if (j < ny && i < nx) sum += rs[j][i];
        const int32_t rowterm = with_rq_offset<SEQ>(
            p.row_coeff * static_cast<int32_t>(sum - 128u * 32u * static_cast<uint32_t>(CB * ny * nx)));
        uint64_t row_addend = 0;                 // lane forms: the row term rides in the multiply-add's addend
        if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
        const uint8_t* wf = lds + p.lds_w[ph] + lane * 16;
        const int4* bias4p = reinterpret_cast<const int4*>(lds + p.lds_bias + ph * p.n_pad * 4);
        for (uint32_t nb = 0; nb < nblocks; nb++) {
          int4 bias4[4];
#pragma unroll
          for (int rg = 0; rg < 4; rg++) bias4[rg] = bias4p[nb * 8 + rg * 2 + khalf];
          v16i acc;
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            acc[rg * 4 + 0] = bias4[rg].x; acc[rg * 4 + 1] = bias4[rg].y;
            acc[rg * 4 + 2] = bias4[rg].z; acc[rg * 4 + 3] = bias4[rg].w;
          }
          const uint8_t* wnb = wf + nb * (ny * nx * CB) * 1024;
#pragma unroll
          for (int j = 0; j < 2; j++) {
#pragma unroll
            for (int i = 0; i < 2; i++) {
              if (j < ny && i < nx) {
#pragma unroll
                for (int c = 0; c < CB; c++) {
                  const v4i w = *reinterpret_cast<const v4i*>(wnb + ((j * nx + i) * CB + c) * 1024);
                  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a[j][i][c], acc, 0, 0, 0);
                }
                // (one pixel's fragments in flight at a time: hoisting all of a phase's weight reads above its MFMAs
                //  costs 32 registers and, at the 128 this kernel is compiled for, spills)
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          if constexpr (rq_is_lane<SEQ>()) {
            igemm_store_tile_lane<SEQ, FULL>(acc, row_addend, out_row, nb * 32, khalf, ok, sp);
          } else {
            igemm_store_tile<SEQ, FULL, false, 2>(acc, bias4, rowterm, out_row, nb * 32, khalf, ok, sp);
          }
          __builtin_amdgcn_sched_barrier(0);              // (one accumulator tile alive at a time: the other waves of the SIMD fill the gaps)
        }
      }
      unit += unit_stride;
      if (unit < units) load_unit(unit, a, img, by, bx, valid);
    }
  }
}

template <int CB, int KH, int KW, int SEQ, bool FULL>
int launch_flavour(const DeconvParams& p, uint32_t lds_bytes, hipStream_t stream)
{
  static PerDeviceOnce attr_once;
  if (auto once_scope = attr_once.begin()) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_deconv_s2_stream_kernel<CB, KH, KW, SEQ, FULL>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kMaxLds));
  }
  const uint32_t total = p.batch * p.BH * p.BW;
  const uint32_t units = (total + 31u) / 32u;
  // one unit per wave while the waves are resident all at once; persistent beyond
  const int cus = qnnp_hip_compute_units();
  const uint32_t max_blocks = static_cast<uint32_t>(cus > 0 ? cus : 256) * static_cast<uint32_t>(resident(CB) * 4 / kWaves);
  uint32_t blocks = (units + kWaves - 1) / kWaves;
  if (blocks > max_blocks) blocks = max_blocks;
  hipLaunchKernelGGL((q8_deconv_s2_stream_kernel<CB, KH, KW, SEQ, FULL>), dim3(blocks), dim3(kThreads), lds_bytes, stream, p);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int CB, int KH, int KW>
int launch_as(const DeconvParams& p, uint32_t lds_bytes, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    rc = launch_flavour<CB, KH, KW, decltype(seq)::value, decltype(full)::value>(p, lds_bytes, stream);
  });
  return rc;
}

template <int KH, int KW>
int launch_cb(const DeconvParams& p, uint32_t cb, uint32_t lds_bytes, hipStream_t stream)
{
  switch (cb) {
    case 1: return launch_as<1, KH, KW>(p, lds_bytes, stream);
    case 2: return launch_as<2, KH, KW>(p, lds_bytes, stream);
    case 3: return launch_as<3, KH, KW>(p, lds_bytes, stream);
    case 4: return launch_as<4, KH, KW>(p, lds_bytes, stream);
    default: return QNNP_HIP_EINVAL;
  }
}

}  // namespace

}  // namespace qnnp

extern "C" int qnnp_hip_deconv_s2_run(const struct qnnp_hip_deconv_s2_args* a, const char** kernel_name)
{
  using namespace qnnp;
  if (a == nullptr || a->batch == 0) return QNNP_HIP_EINVAL;
  const bool k33 = a->kernel_height == 3 && a->kernel_width == 3;
  const bool k44 = a->kernel_height == 4 && a->kernel_width == 4;
  if (!k33 && !k44) return QNNP_HIP_EINVAL;
  if (a->channels == 0 || a->channels % 32 != 0 || a->channels > 128) return QNNP_HIP_EINVAL;
  if (a->n_pad % 32 != 0 || a->n == 0 || a->n > a->n_pad) return QNNP_HIP_EINVAL;
  const uintptr_t in_addr = reinterpret_cast<uintptr_t>(a->input);
  if (in_addr % 16 != 0 || a->input_stride % 16 != 0) return QNNP_HIP_EINVAL;
  const uint32_t cb = a->channels / 32;
  const uint32_t nblocks = a->n_pad / 32;

  DeconvParams p;
  p.input = a->input;
  p.output = a->output;
  uint32_t lds = 0;
  for (int ph = 0; ph < 4; ph++) {
    const uint32_t taps = static_cast<uint32_t>(taps_of(static_cast<int>(a->kernel_height), ph >> 1) *
                                                taps_of(static_cast<int>(a->kernel_width), ph & 1));
    if (a->packed_w[ph] == nullptr || a->bias2[ph] == nullptr || a->k_pad[ph] < taps * a->channels) return QNNP_HIP_EINVAL;
    p.w[ph] = a->packed_w[ph];
    p.bias[ph] = a->bias2[ph];
    p.kblocks[ph] = a->k_pad[ph] / 32;
    p.lds_w[ph] = lds;
    lds += nblocks * taps * cb * 1024;
  }
  p.lds_bias = lds;
  lds += 4u * a->n_pad * 4u;
  if (lds > kMaxLds) return QNNP_HIP_EINVAL;
  p.batch = a->batch; p.H = a->input_height; p.W = a->input_width; p.OH = a->output_height; p.OW = a->output_width;
  if (p.H == 0 || p.W == 0 || p.OH == 0 || p.OW == 0) return QNNP_HIP_EINVAL;
  // base positions: by = (oy + pad_top) / 2 over the output rows
  p.BH = (p.OH - 1u + a->pad_top) / 2u + 1u;
  p.BW = (p.OW - 1u + a->pad_left) / 2u + 1u;
  p.pad_top = a->pad_top; p.pad_left = a->pad_left;
  p.n = a->n; p.n_pad = a->n_pad; p.in_stride = a->input_stride; p.out_stride = a->output_stride;
  p.row_coeff = a->row_coeff;
  // 32-bit byte offsets into both tensors, 32-bit indices in the base grid
  const uint64_t in_pixels = static_cast<uint64_t>(p.batch) * p.H * p.W;
  const uint64_t out_pixels = static_cast<uint64_t>(p.batch) * p.OH * p.OW;
  const uint64_t bases = static_cast<uint64_t>(p.batch) * p.BH * p.BW;
  if (in_pixels * p.in_stride >= (UINT64_C(1) << 32) || out_pixels * p.out_stride >= (UINT64_C(1) << 32) ||
      bases + 32u >= (UINT64_C(1) << 31)) return QNNP_HIP_EINVAL;
  const uintptr_t out_addr = reinterpret_cast<uintptr_t>(a->output);
  p.store_mode = 0;
  if (a->n % 16 == 0 && a->output_stride % 16 == 0 && out_addr % 16 == 0) {
    p.store_mode = 2;
  } else if (a->n % 4 == 0 && a->output_stride % 4 == 0 && out_addr % 4 == 0) {
    p.store_mode = 1;
  }
  const uint8_t* table = qnnp_hip_fill_table();
  if (table == nullptr) return QNNP_HIP_EINVAL;
  p.fill = table + (a->input_zero_point & 0xFFu) * 16u;
  p.rq = make_requant_dev(a->rq);
  p.lane = make_requant_lane(a->rq);
  if (a->bias2_pair == 0) p.lane.kind = 0;      // no pair tables to start from: the offset forms
  hipStream_t stream = reinterpret_cast<hipStream_t>(qnnp_hip_get_stream());
  if (kernel_name != nullptr) *kernel_name = k33 ? "q8_deconv_s2_stream_3x3" : "q8_deconv_s2_stream_4x4";
  return k33 ? launch_cb<3, 3>(p, cb, lds, stream) : launch_cb<4, 4>(p, cb, lds, stream);
}
