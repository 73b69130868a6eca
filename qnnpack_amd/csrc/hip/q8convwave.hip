/*
 * q8convwave.hip -- direct convolution on the matrix cores, one WAVE per 8x8 block of output positions.
 *
 * Same operator and arithmetic as q8convlds.hip (replaces q8conv_ukernel_4x4c2__sse2,
 * src/q8conv/4x4c2-sse2.c:14-273, + compute_q8conv, src/operator-run.c:183-217, 837-842, + the indirection
 * buffer, src/indirection.c:18-79) for BASELINE.json configs[2]-like layers: dense, single group, 32 or 64
 * input channels, <= 64 output channels, a small window (3x3 stride 1).
 *
 * Why a second kernel: in q8convlds.hip a workgroup's four waves walk the phases of an item together
 * (stage band | barrier | K loop | epilogue | barrier), and ablation shows the phases ADD UP -- removing the
 * stores, the staging or the LDS operand reads each shortens the kernel by its full cost -- because two
 * co-resident workgroups are all the overlap there is. Here nothing is shared but the weights:
 *   - ONE workgroup of 12 waves per CU; the packed weight image and the bias go to LDS once;
 *   - each wave owns a private LDS patch and processes UNITS = 8x8 output positions x all channels on its own,
 *     with no workgroup barrier after the weights: global -> registers (next unit's input patch, issued before
 *     the K loop of the current one) -> LDS patch (re-centred bytes a ^ 0x80, chunk-swizzled by patch row, plus
 *     per-pixel channel sums by v_sad_u8) -> K loop over (tap, 32-channel block) reading shifted B fragments
 *     from the patch and A fragments from the shared weights -> row term from the pixel sums -> Q31
 *     requantization into the (now free) patch as a 64 x n output image -> 16-byte stores of whole 8-position
 *     runs;
 *   - units are handed out inside the workgroup by an LDS counter, so a wave that waits (loads, store
 *     acknowledgements) never holds up another one, and three waves per SIMD fill each other's gaps.
 * MEASURED (configs[2], batch 128, same box, production builds): 41.9-42.6 us against 37.0-38.0 us for the
 * LDS-tiled kernel, so this kernel is OPT-IN ("gemm_kernel" = 8) and never selected automatically. In-kernel
 * stamps say why: a wave gets only ~2 units (6272 units over 3072 waves), so there is no steady state -- the
 * first two patches of every wave are 98 % of the input, requested at once (store_patch waits 3.4-4.5 k
 * cycles, the first load issue stalls 10 k), then the K loops run (480 cycles per K block per wave, the MFMA
 * pipe 80 % busy while three waves are in it), then the stores. Kept as the tested starting point for larger
 * batches / images, where units per wave grow.
 * An 8x8 block needs a (7*s + (K-1)*d + 1)^2 patch: 10x10 pixels for 3x3/s1, 1.56x the block's own pixels,
 * re-read from L2 (HBM sees the input about once).
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

constexpr int kWaves = 12;
constexpr int kThreads = kWaves * 64;
constexpr int kPatchVec = 7;            // 16-byte vectors of a unit's input patch per lane (<= 448 per patch)
constexpr uint32_t kFlip = 0x80808080u;
constexpr uint32_t kLdsLimit = 160 * 1024;

struct WaveArgs {
  uint32_t PH, PW;          // patch rows / columns
  uint32_t inv_pw;          // ceil(65536 / PW): q / PW == (q * inv_pw) >> 16 for the small q used here
  uint32_t tiles_x, tiles_y;
  uint32_t units;           // batch * tiles_y * tiles_x
  uint32_t w_bytes;         // packed weight image
  uint32_t head_bytes;      // weights + bias + counter, 256-aligned: offset of the first wave region
  uint32_t patch_bytes;     // per wave: patch / output image (the larger of the two), 256-aligned
  uint32_t wave_bytes;      // per wave: patch_bytes + pixel sums
};

inline bool make_args(const IgemmParams& p, const ConvGeom& g, uint32_t batch, WaveArgs* a, uint32_t* lds_bytes)
{
  a->PH = 7u * g.sh + (g.KH - 1u) * g.dh + 1u;
  a->PW = 7u * g.sw + (g.KW - 1u) * g.dw + 1u;
  a->inv_pw = (65536u + a->PW - 1u) / a->PW;
  a->tiles_x = (g.OW + 7u) / 8u;
  a->tiles_y = (g.OH + 7u) / 8u;
  a->units = batch * a->tiles_x * a->tiles_y;
  a->w_bytes = p.n_pad * p.k_pad;
  a->head_bytes = (a->w_bytes + p.n * 4u + 16u + 255u) & ~255u;
  const uint32_t patch = a->PH * a->PW * p.kc;
  const uint32_t image = 64u * p.n;
  a->patch_bytes = ((patch > image ? patch : image) + 255u) & ~255u;
  a->wave_bytes = a->patch_bytes + ((a->PH * a->PW * 4u + 255u) & ~255u);
  *lds_bytes = a->head_bytes + kWaves * a->wave_bytes;
  if (a->PH * a->PW * (p.kc >> 4) > static_cast<uint32_t>(kPatchVec) * 64u) return false;
  if (a->PH * a->PW > 4096u) return false;   // inv_pw exactness
  return *lds_bytes <= kLdsLimit;
}

template <int TN>
__global__ __launch_bounds__(kThreads)
void q8_conv_wave_mfma_kernel(const IgemmParams p, const ConvGeom g, const WaveArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [weights][bias][counter][12 x (patch | pixel sums)]
  uint8_t* w_lds = lds;
  int32_t* bias_lds = reinterpret_cast<int32_t*>(lds + a.w_bytes);
  uint32_t* counter = reinterpret_cast<uint32_t*>(lds + a.w_bytes + p.n * 4u);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* patch = lds + a.head_bytes + wave * a.wave_bytes;
  int32_t* pix = reinterpret_cast<int32_t*>(patch + a.patch_bytes);

  // contiguous unit range of this workgroup (neighbouring blocks share halos in L2)
  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * a.units / gridDim.x);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * a.units / gridDim.x);

  {
    const uint4* src = reinterpret_cast<const uint4*>(p.packed_w);
    uint4* dst = reinterpret_cast<uint4*>(w_lds);
    for (uint32_t i = tid; i < (a.w_bytes >> 4); i += kThreads) dst[i] = src[i];
    for (uint32_t i = tid; i < p.n; i += kThreads) bias_lds[i] = p.bias2[i];
    if (tid == 0) *counter = lo + kWaves;
  }
  __syncthreads();          // the only workgroup barrier

  const uint32_t cin = p.kc;                       // 32 or 64
  const uint32_t log_cin = 31u - __builtin_clz(cin);
  const uint32_t cpp = cin >> 4;                   // 16-byte chunks per pixel
  const uint32_t log_cpp = log_cin - 4u;
  const uint32_t sh_log = g.sh >> 1;               // strides 1 / 2
  const uint32_t pvec = a.PH * a.PW * cpp;
  const uint32_t tiles = a.tiles_x * a.tiles_y;
  const uint32_t raw_fill = (p.izp_fill & 0xFFu) * 0x01010101u;
  const uint32_t khalf = lane >> 5;

  // this lane's two output positions inside a unit (fixed): i = j*32 + (lane & 31) -> (i >> 3, i & 7)
  uint32_t ty[2], tx[2], qbase[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const uint32_t i = j * 32u + (lane & 31u);
    ty[j] = i >> 3;
    tx[j] = i & 7u;
    qbase[j] = ty[j] * g.sh * a.PW + tx[j] * g.sw;
  }

  uint4 st_val[kPatchVec];
#pragma unroll
  for (int u = 0; u < kPatchVec; u++) st_val[u] = make_uint4(0, 0, 0, 0);

  // global -> registers: the input patch of unit `unit` (padding = the raw zero point)
  auto load_patch = [&](uint32_t unit) __attribute__((always_inline)) {
    const uint32_t img = unit / tiles;
    const uint32_t r = unit - img * tiles;
    const uint32_t tyi = r / a.tiles_x;
    const uint32_t txi = r - tyi * a.tiles_x;
    const int32_t iy0 = static_cast<int32_t>(tyi * 8u * g.sh) - static_cast<int32_t>(g.pad_top);
    const int32_t ix0 = static_cast<int32_t>(txi * 8u * g.sw) - static_cast<int32_t>(g.pad_left);
    const uint8_t* image = p.input + static_cast<uint64_t>(img) * p.image_stride;
#pragma unroll
    for (int u = 0; u < kPatchVec; u++) {
      const uint32_t v = lane + u * 64u;
      const uint32_t c = v & (cpp - 1u);
      const uint32_t q = v >> log_cpp;
      const uint32_t py = (q * a.inv_pw) >> 16;
      const uint32_t px = q - py * a.PW;
      const int32_t iy = iy0 + static_cast<int32_t>(py);
      const int32_t ix = ix0 + static_cast<int32_t>(px);
      const bool inb = v < pvec && iy >= 0 && iy < static_cast<int32_t>(g.H) && ix >= 0 && ix < static_cast<int32_t>(g.W);
      st_val[u] = make_uint4(raw_fill, raw_fill, raw_fill, raw_fill);
      if (inb) {
        st_val[u] = *reinterpret_cast<const uint4*>(
            image + (static_cast<uint64_t>(static_cast<uint32_t>(iy)) * g.W + static_cast<uint32_t>(ix)) * p.input_stride + c * 16u);
      }
    }
  };
  // registers -> LDS patch: re-centred bytes at the swizzled slot, per-pixel channel sums (of a') beside it
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kPatchVec; u++) {
      const uint32_t v = lane + u * 64u;
      const uint32_t c = v & (cpp - 1u);
      const uint32_t q = v >> log_cpp;
      const uint32_t py = (q * a.inv_pw) >> 16;
      const uint32_t swz = (py >> sh_log) & (cpp - 1u);
      uint4 x = st_val[u];
      uint32_t sum = __builtin_amdgcn_sad_u8(x.x, 0u, 0u);
      sum = __builtin_amdgcn_sad_u8(x.y, 0u, sum);
      sum = __builtin_amdgcn_sad_u8(x.z, 0u, sum);
      sum = __builtin_amdgcn_sad_u8(x.w, 0u, sum);
      sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0xB1, 0xF, 0xF, false));        // lane ^ 1
      if (cpp > 2) sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0x4E, 0xF, 0xF, false));   // lane ^ 2
      if (v < pvec) {
        x.x ^= kFlip; x.y ^= kFlip; x.z ^= kFlip; x.w ^= kFlip;
        *reinterpret_cast<uint4*>(patch + (q << log_cin) + ((c ^ swz) << 4)) = x;
        if (c == 0) pix[q] = static_cast<int32_t>(sum) - 128 * static_cast<int32_t>(cin);
      }
    }
  };

  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t cblocks = cin >> 5;
  const uint8_t* w_lane = w_lds + lane * 16;
  const uint32_t cpr = p.n >> 4;                   // 16-byte pieces per output position (2 or 4)
  const uint32_t log_cpr = 31u - __builtin_clz(cpr);


  uint32_t cur = lo + wave;
  if (cur < hi) load_patch(cur);

  uint32_t unit_no = 0;
  (void) unit_no;
#define CW_STAMP(slot) do { if (wave == 0) { QNNP_TRACE(p, blockIdx.x, unit_no, slot); } } while (0)
  while (cur < hi) {
    CW_STAMP(0);
    store_patch();                                  // (waits for the patch loads)
    // next unit: claimed now so that its loads fly under this unit's K loop
    uint32_t claimed = 0;
    if (lane == 0) claimed = atomicAdd(counter, 1u);
    const uint32_t nxt = __builtin_amdgcn_readfirstlane(claimed);
    CW_STAMP(1);
    if (nxt < hi) load_patch(nxt);
    CW_STAMP(2);

    const uint32_t img = cur / tiles;
    const uint32_t r = cur - img * tiles;
    const uint32_t tyi = r / a.tiles_x;
    const uint32_t oy0 = tyi * 8u;
    const uint32_t ox0 = (r - tyi * a.tiles_x) * 8u;

    // accumulators start at the folded bias
    v16i acc[2][TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const v4i b = *reinterpret_cast<const v4i*>(bias_lds + tn * 32 + rg * 8 + khalf * 4);
#pragma unroll
        for (int j = 0; j < 2; j++) {
          acc[j][tn][rg * 4 + 0] = b.x;
          acc[j][tn][rg * 4 + 1] = b.y;
          acc[j][tn][rg * 4 + 2] = b.z;
          acc[j][tn][rg * 4 + 3] = b.w;
        }
      }

    int32_t rs[2] = {0, 0};
    uint32_t kb = 0, t = 0;
    for (uint32_t ky = 0; ky < g.KH; ky++) {
      for (uint32_t kx = 0; kx < g.KW; kx++, t++) {
        const uint32_t tap_off = ky * g.dh * a.PW + kx * g.dw;       // wave-uniform
        uint32_t abase[2], aswz[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const uint32_t q = qbase[j] + tap_off;
          const uint32_t py = ty[j] * g.sh + ky * g.dh;
          abase[j] = q << log_cin;
          aswz[j] = (py >> sh_log) & (cpp - 1u);
          if ((t & 1u) == khalf) rs[j] += pix[q];                    // this lane: every second tap; partner: the others
        }
        for (uint32_t cb = 0; cb < cblocks; cb++, kb++) {
          v4i af[2];
#pragma unroll
          for (int j = 0; j < 2; j++) {
            af[j] = *reinterpret_cast<const v4i*>(patch + abase[j] + ((((cb << 1) | khalf) ^ aswz[j]) << 4));
          }
          v4i wf[TN];
#pragma unroll
          for (int tn = 0; tn < TN; tn++) {
            wf[tn] = *reinterpret_cast<const v4i*>(w_lane + (tn * kblocks + kb) * 1024);
          }
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int tn = 0; tn < TN; tn++)
              acc[j][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[tn], af[j], acc[j][tn], 0, 0, 0);
        }
      }
    }

    CW_STAMP(3);
    // ---- fused epilogue: row term, Q31 requantization into the patch (now an output image), 16-byte stores ----
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // patch reads done before it is overwritten
    requant_dispatch(p.rq, [&](auto shift0, auto full) {
      const int4 no_bias[4] = {};
#pragma unroll
      for (int j = 0; j < 2; j++) {
        int32_t s = rs[j];
        s += __shfl_xor(s, 32);
        const int32_t rowterm = p.row_coeff * s;
        uint8_t* img_row = patch + (j * 32u + (lane & 31u)) * p.n;
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
          igemm_stage_tile<decltype(shift0)::value, decltype(full)::value, false, 2>(
              acc[j][tn], no_bias, rowterm, img_row, tn * 32, khalf, p);
        }
      }
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // image complete before it is read back
    CW_STAMP(4);
    {
      const uint32_t pieces = 64u * cpr;
      uint8_t* out_img = p.output + static_cast<uint64_t>(img) * g.OH * g.OW * p.n;
#pragma unroll
      for (int tt = 0; tt < 4; tt++) {
        const uint32_t idx = lane + tt * 64u;
        const uint32_t i = idx >> log_cpr;
        const uint32_t ch = idx & (cpr - 1u);
        const uint32_t oy = oy0 + (i >> 3);
        const uint32_t ox = ox0 + (i & 7u);
        if (idx < pieces && oy < g.OH && ox < g.OW) {
          *reinterpret_cast<uint4*>(out_img + (static_cast<uint64_t>(oy) * g.OW + ox) * p.n + ch * 16u) =
              *reinterpret_cast<const uint4*>(patch + idx * 16u);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // read back before the next patch lands
    CW_STAMP(5);
    unit_no++;
    cur = nxt;
  }
#undef CW_STAMP
}

template <int TN>
int launch(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, uint32_t lds_bytes, hipStream_t stream)
{
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (attr_once.first()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_wave_mfma_kernel<TN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void) hipGetLastError();
    }
  }
  const uint32_t want = (a.units + kWaves - 1) / kWaves;
  const uint32_t grid = want < p.cu_count ? want : p.cu_count;
  hipLaunchKernelGGL((q8_conv_wave_mfma_kernel<TN>), dim3(grid), dim3(kThreads), lds_bytes, stream, p, g, a);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

}  // namespace

bool convwave_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch)
{
  if (groups != 1 || vec != 16) return false;
  if (!(p.kc == 32 || p.kc == 64)) return false;
  if (!(p.n == 32 || p.n == 64) || p.n_pad != p.n || p.output_stride != p.n) return false;
  if (p.k_total != g.KH * g.KW * p.kc) return false;
  if (g.sh > 2 || g.sw > 2) return false;
  WaveArgs a;
  uint32_t lds_bytes = 0;
  return make_args(p, g, batch, &a, &lds_bytes);
}

int convwave_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name)
{
  WaveArgs a;
  uint32_t lds_bytes = 0;
  if (!make_args(p, g, batch, &a, &lds_bytes)) return QNNP_HIP_EINVAL;
  *name = "q8_conv_wave_mfma";
  return p.n == 32 ? launch<1>(p, g, a, lds_bytes, stream) : launch<2>(p, g, a, lds_bytes, stream);
}

}  // namespace qnnp
