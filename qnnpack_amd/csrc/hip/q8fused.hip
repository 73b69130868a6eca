/*
 * q8fused.hip -- one launch for a whole inverted-residual block:
 *     [1x1 expand ->] 3x3 depthwise (pad 1, stride 1 | 2) -> 1x1 project [-> + block input]
 *
 * Why: the MobileNetV2 sweep is HBM-bound and the expanded tensor (6x the block's input) is written by one
 * operator and read back by the next twice over. Here a workgroup owns a TH x TW tile of the block's OUTPUT and
 * keeps everything between the block input and the block output in LDS:
 *   (once)   the expand / project weight fragments and folded biases -> LDS, the depthwise weights -> registers
 *   stage 0  input tile ((TH-1)s+3) x ((TW-1)s+3) pixels -> LDS (raw bytes, channel padding 0x80)
 *   stage 1  expand: MFMA GEMM (pixels x Cin) x (Cin x Ch), both operands from LDS; Q31 requantization -> uint8
 *            hidden tile in LDS;
 *            pixels outside the image hold the hidden zero point (the depthwise stage's padding)
 *   stage 2  depthwise 3x3 over the hidden tile (v_perm + v_dot2 on int16 tap pairs, as q8dwconv.hip kernel A),
 *            requantization -> uint8 tile in LDS
 *   stage 3  project: MFMA GEMM (TH*TW pixels x Ch) x (Ch x Cout), requantization, optional quantized residual add
 *            with the block input (still in LDS), stores
 * Every intermediate is requantized to uint8 with the stand-alone operator's own parameters, so the output is
 * bit-identical to running the operators one after another (tests/test_gpu_fused.py compares against exactly that
 * and against the oracle). The reference has no such operator; its three microkernel families
 * (src/q8gemm, src/q8dwconv, src/q8vadd) are what one block costs there.
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
typedef short v2s __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kMaxKb1 = 5;                 // input channels <= 160
constexpr uint32_t kFlip = 0x80808080u;
constexpr uint32_t kLdsLimit = 160 * 1024;

struct FusedParams {
  const uint8_t* input;
  uint8_t* output;
  uint32_t batch, H, W, OH, OW, cin, ch, cout, in_stride, out_stride, stride;
  uint32_t TH, TW, tiles_x, tiles_y, IH, IW, pin, inv_iw;
  uint32_t in_pitch, hid_pitch, dw_pitch, in_off, hid_off, dw_off, dw_rows;
  uint32_t w1_off, w3_off, b1_off, b3_off;                    // LDS: weight fragments and folded biases, staged once
  uint32_t has_expand, has_res, hid_zp, store_mode;
  const int8_t* w1; const int32_t* b1; uint32_t kblocks1, kb1, nb1; int32_t rowc1;
  const int16_t* wdw; const int32_t* bdw; uint32_t c_pad;
  const int8_t* w3; const int32_t* b3; uint32_t kblocks3, kb3, nb3; int32_t rowc3;
  RequantDev rq1, rq2, rq3;
  qnnp_hip_add_params add;
};

/*
 * Persistent workgroups: the expand / project weight fragments and biases (a few KB for the early blocks this kernel
 * is selected for) are staged into LDS ONCE, the depthwise weights of a thread's channel group live in registers,
 * and the workgroup then walks output tiles; inside a tile nothing waits on global memory except the tile's own
 * input pixels and output stores.
 */
template <bool RESIDUAL>
__global__ __launch_bounds__(kThreads)
void q8_fused_block_kernel(const FusedParams p)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* in_lds = lds + p.in_off;        // [pin][in_pitch] raw block input (expand operand / residual)
  uint8_t* hid = lds + p.hid_off;          // [pin][hid_pitch] hidden tensor (depthwise input)
  uint8_t* dwb = lds + p.dw_off;           // [dw_rows][dw_pitch] depthwise output (project operand)
  const uint8_t* w1_lds = lds + p.w1_off;  // [nb1][kb1] fragments of 1 KiB
  const uint8_t* w3_lds = lds + p.w3_off;  // [nb3][kb3]
  const int32_t* b1_lds = reinterpret_cast<const int32_t*>(lds + p.b1_off);
  const int32_t* b3_lds = reinterpret_cast<const int32_t*>(lds + p.b3_off);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t col = lane & 31u;
  const uint32_t khalf = lane >> 5;

  // ---- once per workgroup: weights, biases, constant padding ----
  {
    if (p.has_expand) {
      for (uint32_t f = wave; f < p.nb1 * p.kb1; f += kWaves) {
        const uint32_t nb = f / p.kb1, kb = f - nb * p.kb1;
        *reinterpret_cast<uint4*>(lds + p.w1_off + f * 1024 + lane * 16) =
            *reinterpret_cast<const uint4*>(p.w1 + (static_cast<uint64_t>(nb) * p.kblocks1 + kb) * 1024 + lane * 16);
      }
      for (uint32_t i = tid; i < p.nb1 * 32u; i += kThreads) reinterpret_cast<int32_t*>(lds + p.b1_off)[i] = p.b1[i];
    }
    for (uint32_t f = wave; f < p.nb3 * p.kb3; f += kWaves) {
      const uint32_t nb = f / p.kb3, kb = f - nb * p.kb3;
      *reinterpret_cast<uint4*>(lds + p.w3_off + f * 1024 + lane * 16) =
          *reinterpret_cast<const uint4*>(p.w3 + (static_cast<uint64_t>(nb) * p.kblocks3 + kb) * 1024 + lane * 16);
    }
    for (uint32_t i = tid; i < p.nb3 * 32u; i += kThreads) reinterpret_cast<int32_t*>(lds + p.b3_off)[i] = p.b3[i];
    // K / channel padding reads as a' == 0 and is never overwritten: expand operand rows, project operand rows
    if (p.has_expand || p.has_res) {
      for (uint32_t i = tid; i < p.pin * (p.in_pitch >> 2); i += kThreads) reinterpret_cast<uint32_t*>(in_lds)[i] = kFlip;
    }
    for (uint32_t i = tid; i < p.dw_rows * (p.dw_pitch >> 2); i += kThreads) reinterpret_cast<uint32_t*>(dwb)[i] = kFlip;
  }
  // depthwise weights of this thread's 4-channel group: (tap 2i, tap 2i+1) int16 pairs
  const uint32_t q4 = p.ch >> 2;
  const uint32_t nslots = kThreads / q4;
  const uint32_t c4 = tid % q4;
  const uint32_t slot = tid / q4;
  const uint32_t cg = c4 * 4;
  uint32_t wpair[5][4];
  int4 bv = make_int4(0, 0, 0, 0);
  if (slot < nslots) {
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const uint2 lo = *reinterpret_cast<const uint2*>(p.wdw + (2 * i) * p.c_pad + cg);     // 4 x int16
      uint2 hi = make_uint2(0u, 0u);
      if (2 * i + 1 < 9) hi = *reinterpret_cast<const uint2*>(p.wdw + (2 * i + 1) * p.c_pad + cg);
      wpair[i][0] = (lo.x & 0xFFFFu) | (hi.x << 16);
      wpair[i][1] = (lo.x >> 16) | (hi.x & 0xFFFF0000u);
      wpair[i][2] = (lo.y & 0xFFFFu) | (hi.y << 16);
      wpair[i][3] = (lo.y >> 16) | (hi.y & 0xFFFF0000u);
    }
    bv = *reinterpret_cast<const int4*>(p.bdw + cg);
  }
  __syncthreads();

  IgemmParams sp{};                         // what igemm_store_tile reads: requantization, width, store flavour
  sp.rq = p.rq3;
  sp.n = p.cout;
  sp.store_mode = p.store_mode;

  const uint32_t tiles = p.tiles_x * p.tiles_y;
  const uint32_t total = p.batch * tiles;
  const uint32_t cdw_in = p.cin >> 2;
  const uint32_t cdw_hid = p.ch >> 2;
  const uint32_t zp4 = (p.hid_zp & 0xFFu) * 0x01010101u;

  for (uint32_t tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const uint32_t img = tile / tiles;
    const uint32_t t = tile - img * tiles;
    const uint32_t ty = t / p.tiles_x;
    const uint32_t oy0 = ty * p.TH;
    const uint32_t ox0 = (t - ty * p.tiles_x) * p.TW;
    const int32_t iy0 = static_cast<int32_t>(oy0 * p.stride) - 1;
    const int32_t ix0 = static_cast<int32_t>(ox0 * p.stride) - 1;
    const uint8_t* image = p.input + static_cast<uint64_t>(img) * p.H * p.W * p.in_stride;
    // (wave-uniform) does the input tile stick out of the image?
    const bool border = iy0 < 0 || ix0 < 0 || iy0 + static_cast<int32_t>(p.IH) > static_cast<int32_t>(p.H) ||
        ix0 + static_cast<int32_t>(p.IW) > static_cast<int32_t>(p.W);

    // ---- stage 0: block input -> LDS (without an expand stage the hidden tensor IS the block input) ----
    {
      uint8_t* dst = p.has_expand ? in_lds : hid;
      const uint32_t dst_pitch = p.has_expand ? p.in_pitch : p.hid_pitch;
      const uint32_t fill = p.has_expand ? kFlip : zp4;
      for (uint32_t i = tid; i < p.pin * cdw_in; i += kThreads) {
        const uint32_t px = i / cdw_in;
        const uint32_t d = i - px * cdw_in;
        const uint32_t py = (px * p.inv_iw) >> 16;
        const int32_t iy = iy0 + static_cast<int32_t>(py);
        const int32_t ix = ix0 + static_cast<int32_t>(px - py * p.IW);
        uint32_t v = fill;
        if (iy >= 0 && iy < static_cast<int32_t>(p.H) && ix >= 0 && ix < static_cast<int32_t>(p.W)) {
          v = *reinterpret_cast<const uint32_t*>(
              image + (static_cast<uint64_t>(static_cast<uint32_t>(iy)) * p.W + static_cast<uint32_t>(ix)) * p.in_stride + d * 4);
        }
        *reinterpret_cast<uint32_t*>(dst + px * dst_pitch + d * 4) = v;
      }
      if (RESIDUAL && !p.has_expand) {
        // (a block without an expand stage but with a residual: keep a raw copy for the add)
        for (uint32_t i = tid; i < p.pin * cdw_in; i += kThreads) {
          const uint32_t px = i / cdw_in;
          const uint32_t d = i - px * cdw_in;
          *reinterpret_cast<uint32_t*>(in_lds + px * p.in_pitch + d * 4) =
              *reinterpret_cast<const uint32_t*>(hid + px * p.hid_pitch + d * 4);
        }
      }
    }
    __syncthreads();

    // ---- stage 1: expand (pixels x Cin) x (Cin x Ch) -> hidden tile ----
    if (p.has_expand) {
      requant_dispatch(p.rq1, [&](auto shift0, auto full) {
        const uint32_t nrb = (p.pin + 31u) / 32u;
        // work items (row block, channel block) dealt round-robin to the waves
        for (uint32_t item = wave; item < nrb * p.nb1; item += kWaves) {
          const uint32_t rb = item / p.nb1;
          const uint32_t nb = item - rb * p.nb1;
          const uint32_t px = rb * 32u + col;
          const uint32_t pxc = px < p.pin ? px : p.pin - 1u;
          v16i acc;
#pragma unroll
          for (int r = 0; r < 16; r++) acc[r] = 0;
          uint32_t rs = 0;
#pragma unroll
          for (int kb = 0; kb < kMaxKb1; kb++) {
            if (static_cast<uint32_t>(kb) < p.kb1) {
              v4i a = *reinterpret_cast<const v4i*>(in_lds + pxc * p.in_pitch + kb * 32 + khalf * 16);
              const v4i w = *reinterpret_cast<const v4i*>(w1_lds + (nb * p.kb1 + kb) * 1024 + lane * 16);
              rs = __builtin_amdgcn_sad_u8(a.x, 0u, rs);
              rs = __builtin_amdgcn_sad_u8(a.y, 0u, rs);
              rs = __builtin_amdgcn_sad_u8(a.z, 0u, rs);
              rs = __builtin_amdgcn_sad_u8(a.w, 0u, rs);
              a.x ^= static_cast<int>(kFlip);
              a.y ^= static_cast<int>(kFlip);
              a.z ^= static_cast<int>(kFlip);
              a.w ^= static_cast<int>(kFlip);
              acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a, acc, 0, 0, 0);
            }
          }
          rs += __shfl_xor(rs, 32);
          const int32_t rowterm = p.rowc1 * static_cast<int32_t>(rs - 128u * 32u * p.kb1);
          int4 bias4[4];
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            bias4[rg] = *reinterpret_cast<const int4*>(b1_lds + nb * 32 + rg * 8 + khalf * 4);
          }
          igemm_stage_tile_rq<decltype(shift0)::value, decltype(full)::value>(
              acc, bias4, rowterm, hid + pxc * p.hid_pitch, nb * 32, khalf, p.rq1,
              px < p.pin && nb * 32 + khalf * 16 < p.ch);
        }
      });
      __syncthreads();
      // pixels outside the image are the depthwise stage's padding: its input zero point, not expand(anything)
      if (border) {
        for (uint32_t i = tid; i < p.pin * cdw_hid; i += kThreads) {
          const uint32_t px = i / cdw_hid;
          const uint32_t d = i - px * cdw_hid;
          const uint32_t py = (px * p.inv_iw) >> 16;
          const int32_t iy = iy0 + static_cast<int32_t>(py);
          const int32_t ix = ix0 + static_cast<int32_t>(px - py * p.IW);
          if (!(iy >= 0 && iy < static_cast<int32_t>(p.H) && ix >= 0 && ix < static_cast<int32_t>(p.W))) {
            *reinterpret_cast<uint32_t*>(hid + px * p.hid_pitch + d * 4) = zp4;
          }
        }
        __syncthreads();
      }
    }

    // ---- stage 2: depthwise 3x3 over the hidden tile ----
    if (slot < nslots) {
      const uint32_t npos = p.TH * p.TW;
      requant_dispatch(p.rq2, [&](auto shift0, auto full) {
        for (uint32_t pos = slot; pos < npos; pos += nslots) {
          const uint32_t oyl = pos / p.TW;
          const uint32_t oxl = pos - oyl * p.TW;
          const uint8_t* base = hid + ((oyl * p.stride) * p.IW + oxl * p.stride) * p.hid_pitch + cg;
          uint32_t in[9];
#pragma unroll
          for (int ky = 0; ky < 3; ky++)
#pragma unroll
            for (int kx = 0; kx < 3; kx++)
              in[ky * 3 + kx] = *reinterpret_cast<const uint32_t*>(base + (ky * p.IW + kx) * p.hid_pitch);
          int32_t acc0 = bv.x, acc1 = bv.y, acc2 = bv.z, acc3 = bv.w;
#pragma unroll
          for (int i = 0; i < 5; i++) {
            const uint32_t in0 = in[2 * i];
            const uint32_t in1 = (2 * i + 1 < 9) ? in[2 * i + 1] : 0u;
            const uint32_t p0 = __builtin_amdgcn_perm(in1, in0, 0x0c040c00u);
            const uint32_t p1 = __builtin_amdgcn_perm(in1, in0, 0x0c050c01u);
            const uint32_t p2 = __builtin_amdgcn_perm(in1, in0, 0x0c060c02u);
            const uint32_t p3 = __builtin_amdgcn_perm(in1, in0, 0x0c070c03u);
            acc0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p0), __builtin_bit_cast(v2s, wpair[i][0]), acc0, false);
            acc1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p1), __builtin_bit_cast(v2s, wpair[i][1]), acc1, false);
            acc2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p2), __builtin_bit_cast(v2s, wpair[i][2]), acc2, false);
            acc3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p3), __builtin_bit_cast(v2s, wpair[i][3]), acc3, false);
          }
          *reinterpret_cast<uint32_t*>(dwb + pos * p.dw_pitch + cg) =
              q31_requantize_pack4<decltype(shift0)::value, decltype(full)::value>(acc0, acc1, acc2, acc3, p.rq2);
        }
      });
    }
    __syncthreads();

    // ---- stage 3: project (TH*TW pixels x Ch) x (Ch x Cout) [+ residual] -> global ----
    {
      const uint32_t npos = p.TH * p.TW;
      const uint32_t nrb = (npos + 31u) / 32u;
      requant_dispatch(p.rq3, [&](auto shift0, auto full) {
        for (uint32_t item = wave; item < nrb * p.nb3; item += kWaves) {
          const uint32_t rb = item / p.nb3;
          const uint32_t nb = item - rb * p.nb3;
          const uint32_t pos = rb * 32u + col;
          const uint32_t posc = pos < npos ? pos : npos - 1u;
          const uint32_t oyl = posc / p.TW;
          const uint32_t oxl = posc - oyl * p.TW;
          const uint32_t oy = oy0 + oyl, ox = ox0 + oxl;
          const bool ok = pos < npos && oy < p.OH && ox < p.OW;
          uint8_t* out_row = p.output + ((static_cast<uint64_t>(img) * p.OH + (ok ? oy : 0u)) * p.OW + (ok ? ox : 0u)) * p.out_stride;
          const uint8_t* res_row = in_lds + ((oyl * p.stride + 1u) * p.IW + oxl * p.stride + 1u) * p.in_pitch;
          const uint8_t* arow = dwb + posc * p.dw_pitch + khalf * 16;
          const uint8_t* wf = w3_lds + nb * p.kb3 * 1024 + lane * 16;
          v16i acc;
#pragma unroll
          for (int r = 0; r < 16; r++) acc[r] = 0;
          uint32_t rs = 0;
          for (uint32_t kb = 0; kb < p.kb3; kb++) {
            v4i a = *reinterpret_cast<const v4i*>(arow + kb * 32);
            const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
            rs = __builtin_amdgcn_sad_u8(a.x, 0u, rs);
            rs = __builtin_amdgcn_sad_u8(a.y, 0u, rs);
            rs = __builtin_amdgcn_sad_u8(a.z, 0u, rs);
            rs = __builtin_amdgcn_sad_u8(a.w, 0u, rs);
            a.x ^= static_cast<int>(kFlip);
            a.y ^= static_cast<int>(kFlip);
            a.z ^= static_cast<int>(kFlip);
            a.w ^= static_cast<int>(kFlip);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a, acc, 0, 0, 0);
          }
          rs += __shfl_xor(rs, 32);
          const int32_t rowterm = p.rowc3 * static_cast<int32_t>(rs - 128u * 32u * p.kb3);
          int4 bias4[4];
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            bias4[rg] = *reinterpret_cast<const int4*>(b3_lds + nb * 32 + rg * 8 + khalf * 4);
          }
          igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 0, RESIDUAL>(
              acc, bias4, rowterm, out_row, nb * 32, khalf, ok, sp, res_row, &p.add);
        }
      });
    }
    __syncthreads();      // the tile buffers are reused by the next tile
  }
}

/* tile and LDS plan; false when the block does not fit (then the stand-alone operators run it) */
bool plan(const qnnp_hip_fused_args& a, FusedParams* p, uint32_t* lds_bytes)
{
  if (a.stride != 1 && a.stride != 2) return false;
  if (a.input_channels % 4 != 0 || a.hidden_channels % 16 != 0 || a.output_channels % 4 != 0) return false;
  if (a.hidden_channels / 4 > static_cast<uint32_t>(kThreads)) return false;
  if (a.has_expand && a.input_channels > 32u * kMaxKb1) return false;
  if (!a.has_expand && a.hidden_channels != a.input_channels) return false;
  if (a.has_residual && (a.stride != 1 || a.input_channels != a.output_channels)) return false;
  if (a.input_stride % 4 != 0 || reinterpret_cast<uintptr_t>(a.input) % 4 != 0) return false;
  p->kb1 = (a.input_channels + 31u) / 32u;
  p->kb3 = (a.hidden_channels + 31u) / 32u;
  p->nb1 = a.has_expand ? a.expand_n_pad / 32u : 0u;
  p->nb3 = a.project_n_pad / 32u;
  p->in_pitch = p->kb1 * 32u + 16u;
  p->hid_pitch = a.hidden_channels + 16u;
  p->dw_pitch = p->kb3 * 32u + 16u;
  // weights + biases resident in LDS: this kernel is for the blocks where they are small (the early, large-image
  // blocks, which are the HBM-bound ones); later blocks have 50-300 KB of weights and little activation traffic
  const uint32_t w1_bytes = p->nb1 * p->kb1 * 1024u, w3_bytes = p->nb3 * p->kb3 * 1024u;
  const uint32_t bias_bytes = (p->nb1 + p->nb3) * 32u * 4u;
  const uint32_t weights = w1_bytes + w3_bytes + ((bias_bytes + 255u) & ~255u);
  if (weights > 64u * 1024u) return false;
  // candidate tiles, largest first; 64 output positions fill two MFMA row blocks exactly
  const uint32_t cand[][2] = {{8, 8}, {7, 7}, {4, 8}, {4, 7}, {4, 4}, {2, 8}, {2, 7}, {2, 4}, {1, 8}, {1, 7}};
  for (const auto& c : cand) {
    uint32_t th = c[0], tw = c[1];
    if (a.output_width % tw != 0 && !(tw == 8 && a.output_width % 7 != 0)) continue;   // 7-wide tiles only where they divide
    if (th > a.output_height) th = a.output_height;
    if (tw > a.output_width) tw = a.output_width;
    const uint32_t ih = (th - 1) * a.stride + 3, iw = (tw - 1) * a.stride + 3;
    const uint32_t pin = ih * iw;
    if (pin > 4096u) continue;
    const uint32_t dw_rows = ((th * tw + 31u) / 32u) * 32u;
    const uint32_t in_bytes = (a.has_expand || a.has_residual) ? ((pin * p->in_pitch + 255u) & ~255u) : 0u;
    const uint32_t hid_bytes = (pin * p->hid_pitch + 255u) & ~255u;
    const uint32_t dw_bytes = (dw_rows * p->dw_pitch + 255u) & ~255u;
    const uint32_t need = weights + in_bytes + hid_bytes + dw_bytes;
    if (need > kLdsLimit / 2) continue;        // two workgroups per CU at least
    p->TH = th; p->TW = tw; p->IH = ih; p->IW = iw; p->pin = pin; p->dw_rows = dw_rows;
    p->inv_iw = (65536u + iw - 1u) / iw;
    p->w1_off = 0; p->w3_off = w1_bytes; p->b1_off = w1_bytes + w3_bytes; p->b3_off = p->b1_off + p->nb1 * 128u;
    p->in_off = weights; p->hid_off = weights + in_bytes; p->dw_off = weights + in_bytes + hid_bytes;
    *lds_bytes = need;
    p->tiles_x = (a.output_width + tw - 1) / tw;
    p->tiles_y = (a.output_height + th - 1) / th;
    return true;
  }
  return false;
}

}  // namespace

}  // namespace qnnp

extern "C" int qnnp_hip_fused_block_supported(const struct qnnp_hip_fused_args* a)
{
  qnnp::FusedParams p{};
  uint32_t lds_bytes = 0;
  return a != nullptr && qnnp::plan(*a, &p, &lds_bytes) ? 1 : 0;
}

extern "C" int qnnp_hip_fused_block_run(const struct qnnp_hip_fused_args* a, const char** kernel_name)
{
  using namespace qnnp;
  if (a == nullptr || a->input == nullptr || a->output == nullptr) return QNNP_HIP_EINVAL;
  if (a->batch == 0) return QNNP_HIP_OK;
  FusedParams p{};
  uint32_t lds_bytes = 0;
  if (!plan(*a, &p, &lds_bytes)) return QNNP_HIP_EINVAL;
  p.input = a->input; p.output = a->output;
  p.H = a->input_height; p.W = a->input_width; p.OH = a->output_height; p.OW = a->output_width;
  p.cin = a->input_channels; p.ch = a->hidden_channels; p.cout = a->output_channels;
  p.in_stride = a->input_stride; p.out_stride = a->output_stride; p.stride = a->stride;
  p.has_expand = a->has_expand; p.has_res = a->has_residual;
  p.hid_zp = a->dw_input_zero_point;
  const uintptr_t out_addr = reinterpret_cast<uintptr_t>(a->output);
  p.store_mode = 0;
  if (a->output_channels % 16 == 0 && a->output_stride % 16 == 0 && out_addr % 16 == 0) p.store_mode = 2;
  else if (a->output_channels % 4 == 0 && a->output_stride % 4 == 0 && out_addr % 4 == 0) p.store_mode = 1;
  p.batch = a->batch;
  p.w1 = a->expand_w; p.b1 = a->expand_bias2; p.kblocks1 = a->expand_k_pad / 32;
  p.rowc1 = a->expand_row_coeff;
  p.wdw = a->dw_wadj; p.bdw = a->dw_bias1; p.c_pad = a->dw_c_pad;
  p.w3 = a->project_w; p.b3 = a->project_bias2; p.kblocks3 = a->project_k_pad / 32;
  p.rowc3 = a->project_row_coeff;
  if (a->has_expand) p.rq1 = make_requant_dev(a->expand_rq);
  p.rq2 = make_requant_dev(a->dw_rq);
  p.rq3 = make_requant_dev(a->project_rq);
  p.add = a->add;
  if (p.kb3 > p.kblocks3 || (a->has_expand && p.kb1 > p.kblocks1)) return QNNP_HIP_EINVAL;

  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_fused_block_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_fused_block_kernel<true>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit);
    (void) hipGetLastError();
  }
  const uint64_t total_tiles = static_cast<uint64_t>(a->batch) * p.tiles_x * p.tiles_y;
  if (total_tiles > 0x7FFFFFFFull) return QNNP_HIP_EINVAL;
  int cus = qnnp_hip_compute_units();
  if (cus <= 0) cus = 256;
  uint32_t per_cu = kLdsLimit / lds_bytes;
  if (per_cu > 4u) per_cu = 4u;
  uint64_t blocks = static_cast<uint64_t>(cus) * per_cu;
  if (blocks > total_tiles) blocks = total_tiles;
  hipStream_t stream = reinterpret_cast<hipStream_t>(qnnp_hip_get_stream());
  if (a->has_residual) {
    hipLaunchKernelGGL(q8_fused_block_kernel<true>, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), lds_bytes, stream, p);
  } else {
    hipLaunchKernelGGL(q8_fused_block_kernel<false>, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), lds_bytes, stream, p);
  }
  if (kernel_name != nullptr) *kernel_name = "q8_fused_block";
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}
