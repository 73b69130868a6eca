/*
 * q8convlds.hip -- LDS-tiled direct convolution on the matrix cores (dense, single group).
 *
 * Replaces q8conv_ukernel_4x4c2__sse2 (src/q8conv/4x4c2-sse2.c:14-273) + compute_q8conv
 * (src/operator-run.c:183-217, 837-842) + the indirection buffer (src/indirection.c:18-79) for
 * convolutions whose input channel count is 32/64/128/256 -- BASELINE.json configs[2]
 * (3x3 s1, 56x56x64 -> 64, batch 128).
 *
 * An implicit GEMM that gathers every tap from global memory moves each input byte KH*KW times
 * through L2. Here a workgroup stages the input rows its 256 output positions need ONCE into LDS
 * (coalesced 16-byte NHWC reads, padding materialised as the zero point, bytes already re-centred
 * a' = a ^ 0x80) together with the whole weight image (MFMA fragments, pack.h), and the K loop
 * (tap-major, then 32-channel blocks) reads shifted fragments of that LDS tile:
 *   MFMA B operand (activations): lane l -> output position (l & 31), 16 channels (l >> 5)*16..+15 of
 *     input pixel (y*sh + ky*dh, x*sw + kx*dw): one ds_read_b128; the 16-byte chunk index of a pixel
 *     is XOR-swizzled with the pixel index so consecutive positions hit different banks.
 *   MFMA A operand (weights): linear 1 KiB fragment reads.
 * HBM traffic = input once + output once (the algorithmic bytes); the 9x tap re-reads stay in LDS.
 * Arithmetic and epilogue are those of q8igemm.hip (same packed weights, same folded bias, per-position
 * row sum of a' by v_dot4, Q31 requantization, 16-byte stores).
 *
 * Workgroup = 4 waves, PERSISTENT: the weight image is staged once, then the workgroup walks work items of
 * 256 consecutive output positions of one image (each wave two groups of 32 positions, so a weight
 * fragment feeds two MFMAs) x all output channels (<= 128).
 */
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <stdint.h>

#include "igemm_epilogue.hip.h"
#include "igemm_params.h"
#include "per_device.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int kPosPerBlock = 256;      // 4 waves x 2 groups x 32 positions
constexpr uint32_t kFlip = 0x80808080u;
constexpr uint32_t kLdsLimit = 160 * 1024;  // LDS per CU; two workgroups are co-resident when a plan needs <= half

struct Plan {
  uint32_t blocks_per_image;
  uint32_t ir_max;       // input rows staged per workgroup (upper bound)
  uint32_t ic;           // staged columns: (OW-1)*sw + (KW-1)*dw + 1, starting at -pad_left
  uint32_t w_bytes;      // packed weight image
  uint32_t lds_bytes;
};

inline Plan make_plan(const IgemmParams& p, const ConvGeom& g)
{
  Plan pl;
  const uint32_t ohw = g.OH * g.OW;
  pl.blocks_per_image = (ohw + kPosPerBlock - 1) / kPosPerBlock;
  const uint32_t rows_spanned = (kPosPerBlock + g.OW - 1) / g.OW + 1;
  const uint32_t rows = rows_spanned < g.OH ? rows_spanned : g.OH;
  pl.ir_max = (rows - 1) * g.sh + (g.KH - 1) * g.dh + 1;
  pl.ic = (g.OW - 1) * g.sw + (g.KW - 1) * g.dw + 1;
  pl.w_bytes = p.n_pad * p.k_pad;
  pl.lds_bytes = pl.w_bytes + pl.ir_max * pl.ic * p.kc;
  return pl;
}

/* PIPE: the next work item's input band is loaded into registers (<= kPipeVec vectors per thread) while the
 * current one is multiplied -- staging was 40 % of an item's time as a separate phase (in-kernel stamps). */
constexpr int kPipeVec = 8;

/* ABL: measurement-only ablation mask (builds with -DQNNP_ENABLE_ABLATION, env QNNP_CONV_ABL; scripts/gpu_convabl.sh):
 * 1 no row-sum v_dot4, 2 linear (conflict-free, address-free) A reads, 4 no weight reads, 8 no MFMA, 16 no stores,
 * 32 no staging. Results are wrong by design; only the time is read. */
template <int TN, bool PIPE, int ABL = 0>
__global__ __launch_bounds__(kThreads, 2)
void q8_conv_lds_mfma_kernel(const IgemmParams p, const ConvGeom g, const uint32_t blocks_per_image,
                             const uint32_t total_items, const uint32_t ic, const uint32_t w_bytes)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [weight fragments][input band]
  uint8_t* w_lds = lds;
  uint8_t* in_lds = lds + w_bytes;

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t ohw = g.OH * g.OW;

  // ---- weights: staged once per (persistent) workgroup ----
  {
    const uint4* src = reinterpret_cast<const uint4*>(p.packed_w);
    uint4* dst = reinterpret_cast<uint4*>(w_lds);
    for (uint32_t i = tid; i < (w_bytes >> 4); i += kThreads) dst[i] = src[i];
  }

  // ---- pipelined staging (PIPE): global -> registers for one item, registers -> LDS (re-centred, swizzled) ----
  uint4 st_val[PIPE ? kPipeVec : 1];
  uint32_t st_dst[PIPE ? kPipeVec : 1];                             // 0xFFFFFFFF = nothing held
  auto stage_load = [&](uint32_t item) __attribute__((always_inline)) {
    const uint32_t cin = p.kc;
    const uint32_t log_cin = 31u - __builtin_clz(cin);
    const uint32_t cpp = cin >> 4;
    const uint32_t log_ppr = 4u - (log_cin - 4u);
    const uint32_t img = item / blocks_per_image;
    const uint32_t blk = item - img * blocks_per_image;
    const uint32_t p0 = blk * kPosPerBlock;
    const uint32_t p_end = min(p0 + kPosPerBlock, ohw);
    const uint32_t y_first = p0 / g.OW;
    const uint32_t y_last = (p_end - 1) / g.OW;
    const int32_t iy0 = static_cast<int32_t>(y_first * g.sh) - static_cast<int32_t>(g.pad_top);
    const uint32_t ir = (y_last - y_first) * g.sh + (g.KH - 1) * g.dh + 1;
    const uint32_t nvec = ir * ic * cpp;
    const uint8_t* image = p.input + static_cast<uint64_t>(img) * p.image_stride;
    const uint32_t raw_fill = (p.izp_fill & 0xFFu) * 0x01010101u;
#pragma unroll
    for (int u = 0; u < (PIPE ? kPipeVec : 1); u++) {
      const uint32_t v = tid + u * kThreads;
      const bool live = v < nvec;
      const uint32_t c = v & (cpp - 1);
      const uint32_t q = v >> (log_cin - 4);               // pixel index inside the band
      const uint32_t iyl = q / ic;
      const uint32_t ixl = q - iyl * ic;
      const int32_t iy = iy0 + static_cast<int32_t>(iyl);
      const int32_t ix = static_cast<int32_t>(ixl) - static_cast<int32_t>(g.pad_left);
      const bool inb = live && iy >= 0 && iy < static_cast<int32_t>(g.H) && ix >= 0 && ix < static_cast<int32_t>(g.W);
      st_val[u] = make_uint4(raw_fill, raw_fill, raw_fill, raw_fill);
      if (inb) {
        st_val[u] = *reinterpret_cast<const uint4*>(
            image + (static_cast<uint64_t>(static_cast<uint32_t>(iy)) * g.W + static_cast<uint32_t>(ix)) * p.input_stride + c * 16);
      }
      const uint32_t swz = (q >> log_ppr) & (cpp - 1);
      st_dst[u] = live ? (q << log_cin) + ((c ^ swz) << 4) : 0xFFFFFFFFu;
    }
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < (PIPE ? kPipeVec : 1); u++) {
      if (st_dst[u] != 0xFFFFFFFFu) {
        uint4 x = st_val[u];
        x.x ^= kFlip; x.y ^= kFlip; x.z ^= kFlip; x.w ^= kFlip;
        *reinterpret_cast<uint4*>(in_lds + st_dst[u]) = x;
      }
    }
  };
  if constexpr (PIPE) {
    if (blockIdx.x < total_items) stage_load(blockIdx.x);
  }

  // persistent: work item = (image, 256-position block); items of one image are consecutive
  uint32_t item_no = 0;
  for (uint32_t item = blockIdx.x; item < total_items; item += gridDim.x, item_no++) {
  QNNP_TRACE(p, blockIdx.x, item_no, 0);
  const uint32_t img = item / blocks_per_image;
  const uint32_t blk = item - img * blocks_per_image;
  const uint32_t p0 = blk * kPosPerBlock;
  const uint32_t p_end = min(p0 + kPosPerBlock, ohw);

  const uint32_t cin = p.kc;                       // 32, 64, 128 or 256
  const uint32_t log_cin = 31u - __builtin_clz(cin);
  const uint32_t cpp = cin >> 4;                   // 16-byte chunks per pixel (2..16)
  const uint32_t log_ppr = 4u - (log_cin - 4u);    // log2(pixels per 256-byte bank row) = log2(16 / cpp)

  // input rows this workgroup needs
  const uint32_t y_first = p0 / g.OW;
  const uint32_t y_last = (p_end - 1) / g.OW;
  const int32_t iy0 = static_cast<int32_t>(y_first * g.sh) - static_cast<int32_t>(g.pad_top);
  const uint32_t ir = (y_last - y_first) * g.sh + (g.KH - 1) * g.dh + 1;

  // ---- stage the input band (re-centred, swizzled) ----
  // Batches of kStageBatch vectors per thread: all global loads of a batch are issued before the first
  // LDS write, so a thread keeps kStageBatch loads in flight instead of one (the loop is latency-bound).
  if constexpr (PIPE) {
    if constexpr (!(ABL & 32)) stage_store();                                   // this item's band was loaded during the previous item
  } else {
    constexpr int kStageBatch = 8;
    const uint32_t nvec = ir * ic * cpp;
    const uint8_t* image = p.input + static_cast<uint64_t>(img) * p.image_stride;
    const uint32_t fill = ((p.izp_fill & 0xFFu) ^ 0x80u) * 0x01010101u;
    for (uint32_t v0 = tid; v0 < nvec; v0 += kThreads * kStageBatch) {
      uint4 val[kStageBatch];
      uint32_t dst[kStageBatch];
      bool live[kStageBatch];
#pragma unroll
      for (int u = 0; u < kStageBatch; u++) {
        const uint32_t v = v0 + u * kThreads;
        live[u] = v < nvec;
        const uint32_t c = v & (cpp - 1);
        const uint32_t q = v >> (log_cin - 4);             // pixel index inside the band
        const uint32_t iyl = q / ic;
        const uint32_t ixl = q - iyl * ic;
        const int32_t iy = iy0 + static_cast<int32_t>(iyl);
        const int32_t ix = static_cast<int32_t>(ixl) - static_cast<int32_t>(g.pad_left);
        const bool inb = live[u] && iy >= 0 && iy < static_cast<int32_t>(g.H) && ix >= 0 && ix < static_cast<int32_t>(g.W);
        val[u] = make_uint4(fill ^ kFlip, fill ^ kFlip, fill ^ kFlip, fill ^ kFlip);   // re-centred below
        if (inb) {
          val[u] = *reinterpret_cast<const uint4*>(
              image + (static_cast<uint64_t>(static_cast<uint32_t>(iy)) * g.W + static_cast<uint32_t>(ix)) * p.input_stride + c * 16);
        }
        const uint32_t swz = (q >> log_ppr) & (cpp - 1);
        dst[u] = (q << log_cin) + ((c ^ swz) << 4);
      }
#pragma unroll
      for (int u = 0; u < kStageBatch; u++) {
        if (live[u]) {
          uint4 x = val[u];
          x.x ^= kFlip; x.y ^= kFlip; x.z ^= kFlip; x.w ^= kFlip;
          *reinterpret_cast<uint4*>(in_lds + dst[u]) = x;
        }
      }
    }
  }

  QNNP_TRACE(p, blockIdx.x, item_no, 1);
  // ---- this lane's two output positions ----
  const uint32_t khalf = lane >> 5;
  uint32_t pos[2], qbase[2];
  bool valid[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    pos[j] = p0 + wave * 64 + j * 32 + (lane & 31u);
    valid[j] = pos[j] < p_end;
    const uint32_t pp = valid[j] ? pos[j] : p0;
    const uint32_t y = pp / g.OW;
    const uint32_t x = pp - y * g.OW;
    qbase[j] = ((y - y_first) * g.sh) * ic + x * g.sw;
  }

  // bias of this lane's 4-channel groups
  int4 bias4[TN][4];
#pragma unroll
  for (int tn = 0; tn < TN; tn++)
#pragma unroll
    for (int rg = 0; rg < 4; rg++)
      bias4[tn][rg] = *reinterpret_cast<const int4*>(p.bias2 + tn * 32 + rg * 8 + khalf * 4);

  v16i acc[2][TN];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int tn = 0; tn < TN; tn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[j][tn][r] = 0;
  int32_t rs[2] = {0, 0};

  __syncthreads();
  QNNP_TRACE(p, blockIdx.x, item_no, 2);
  if constexpr (PIPE) {
    if constexpr (!(ABL & 32)) if (item + gridDim.x < total_items) stage_load(item + gridDim.x);   // flies under the K loop and the epilogue
  }

  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t cblocks = cin >> 5;
  const uint8_t* w_lane = w_lds + lane * 16;
  uint32_t kb = 0;
  for (uint32_t ky = 0; ky < g.KH; ky++) {
    for (uint32_t kx = 0; kx < g.KW; kx++) {
      const uint32_t tap_off = ky * g.dh * ic + kx * g.dw;         // wave-uniform
      uint32_t abase[2], aswz[2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const uint32_t q = qbase[j] + tap_off;
        abase[j] = q << log_cin;
        aswz[j] = (q >> log_ppr) & (cpp - 1);
      }
      for (uint32_t cb = 0; cb < cblocks; cb++, kb++) {
        v4i af[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if constexpr (ABL & 2) af[j] = *reinterpret_cast<const v4i*>(in_lds + lane * 16 + j * 1024 + cb * 2048);
          else af[j] = *reinterpret_cast<const v4i*>(in_lds + abase[j] + ((((cb << 1) | khalf) ^ aswz[j]) << 4));
        }
        v4i wf[TN];
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
          if constexpr (ABL & 4) wf[tn] = v4i{static_cast<int>(lane), static_cast<int>(kb), tn, 3};
          else wf[tn] = *reinterpret_cast<const v4i*>(w_lane + (tn * kblocks + kb) * 1024);
        }
        if constexpr (!(ABL & 1))
#pragma unroll
        for (int j = 0; j < 2; j++) {
          int32_t s = rs[j];
          s = __builtin_amdgcn_sdot4(af[j].x, 0x01010101, s, false);
          s = __builtin_amdgcn_sdot4(af[j].y, 0x01010101, s, false);
          s = __builtin_amdgcn_sdot4(af[j].z, 0x01010101, s, false);
          s = __builtin_amdgcn_sdot4(af[j].w, 0x01010101, s, false);
          rs[j] = s;
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++)
            if constexpr (ABL & 8) acc[j][tn][0] += wf[tn].x ^ af[j].x ^ wf[tn].w ^ af[j].w;
            else acc[j][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[tn], af[j], acc[j][tn], 0, 0, 0);
      }
    }
  }

  QNNP_TRACE(p, blockIdx.x, item_no, 3);
  // ---- fused epilogue ----
  requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      int32_t s = rs[j];
      s += __shfl_xor(s, 32);                     // the two K halves of a position live in lanes l, l+32
      // (+ 2^31 for the offset rounding sequences, requant.hip.h)
      const int32_t rowterm = with_rq_offset<decltype(shift0)::value>(p.row_coeff * s);
      uint8_t* out_row = p.output + (static_cast<uint64_t>(img) * ohw + pos[j]) * p.output_stride;
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        igemm_store_tile<decltype(shift0)::value, decltype(full)::value>(
            acc[j][tn], bias4[tn], rowterm, out_row, tn * 32, khalf, (ABL & 16) ? false : valid[j], p);
      }
    }
  });
  QNNP_TRACE(p, blockIdx.x, item_no, 4);
  __syncthreads();    // every wave is done with this band before the next item overwrites it
  QNNP_TRACE(p, blockIdx.x, item_no, 5);
  }
}

template <int TN, bool PIPE, int ABL = 0>
int launch_one(const IgemmParams& p, const ConvGeom& g, const Plan& pl, uint32_t batch, hipStream_t stream)
{
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_lds_mfma_kernel<TN, PIPE, ABL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void) hipGetLastError();
    }
  }
  const uint32_t total_items = batch * pl.blocks_per_image;
  const uint32_t resident = p.cu_count * (kLdsLimit >= 2 * pl.lds_bytes ? 2u : 1u);
  const uint32_t grid = total_items < resident ? total_items : resident;
  hipLaunchKernelGGL((q8_conv_lds_mfma_kernel<TN, PIPE, ABL>), dim3(grid), dim3(kThreads),
                     pl.lds_bytes, stream, p, g, pl.blocks_per_image, total_items, pl.ic, pl.w_bytes);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int TN>
int launch(const IgemmParams& p, const ConvGeom& g, const Plan& pl, uint32_t batch, hipStream_t stream)
{
  // the pipelined flavour needs the whole band of an item in kPipeVec 16-byte vectors per thread
  const uint32_t band_vectors = pl.ir_max * pl.ic * (p.kc >> 4);
  // (TN >= 3: 96-128 accumulator registers leave no room for the held band -- the pipelined flavour spills)
  if constexpr (TN <= 2) {
#ifdef QNNP_ENABLE_ABLATION
    if constexpr (TN == 2) {
      static const int abl = getenv("QNNP_CONV_ABL") ? atoi(getenv("QNNP_CONV_ABL")) : 0;
      switch (abl) {
        case 1: return launch_one<TN, true, 1>(p, g, pl, batch, stream);
        case 2: return launch_one<TN, true, 2>(p, g, pl, batch, stream);
        case 4: return launch_one<TN, true, 4>(p, g, pl, batch, stream);
        case 7: return launch_one<TN, true, 7>(p, g, pl, batch, stream);
        case 8: return launch_one<TN, true, 8>(p, g, pl, batch, stream);
        case 16: return launch_one<TN, true, 16>(p, g, pl, batch, stream);
        case 32: return launch_one<TN, true, 32>(p, g, pl, batch, stream);
        case 48: return launch_one<TN, true, 48>(p, g, pl, batch, stream);
        case 55: return launch_one<TN, true, 55>(p, g, pl, batch, stream);
        case 63: return launch_one<TN, true, 63>(p, g, pl, batch, stream);
        default: break;
      }
    }
#endif
    if (band_vectors <= static_cast<uint32_t>(kPipeVec * kThreads)) return launch_one<TN, true>(p, g, pl, batch, stream);
  }
  return launch_one<TN, false>(p, g, pl, batch, stream);
}

}  // namespace

bool convlds_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec)
{
  if (p.offsets == nullptr || groups != 1 || vec != 16) return false;
  if (!(p.kc == 32 || p.kc == 64 || p.kc == 128 || p.kc == 256)) return false;
  if (p.n % 32 != 0 || p.n > 128 || p.n_pad != p.n) return false;
  if (p.k_total != g.KH * g.KW * p.kc) return false;
  const Plan pl = make_plan(p, g);
  return pl.lds_bytes <= kLdsLimit;
}

int convlds_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name)
{
  const Plan pl = make_plan(p, g);
  *name = "q8_conv_lds_mfma";
  switch (p.n / 32) {
    case 1: return launch<1>(p, g, pl, batch, stream);
    case 2: return launch<2>(p, g, pl, batch, stream);
    case 3: return launch<3>(p, g, pl, batch, stream);
    default: return launch<4>(p, g, pl, batch, stream);
  }
}

}  // namespace qnnp
