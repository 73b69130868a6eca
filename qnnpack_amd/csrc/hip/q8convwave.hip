/*
 * q8convwave.hip -- direct convolution on the matrix cores, one WAVE per 8x8 block of output positions, input
 * patches streamed by LDS-DMA two units ahead of the multiplies.
 *
 * Same operator and arithmetic as q8convlds.hip (replaces q8conv_ukernel_4x4c2__sse2,
 * src/q8conv/4x4c2-sse2.c:14-273, + compute_q8conv, src/operator-run.c:183-217, 837-842, + the indirection
 * buffer, src/indirection.c:18-79) for BASELINE.json configs[2]-like layers: dense, single group, 32 or 64
 * input channels, <= 64 output channels, a small window (3x3 stride 1).
 *
 * Why a second kernel: in q8convlds.hip a workgroup's four waves walk the phases of an item together
 * (stage band | barrier | K loop | epilogue | barrier). In-kernel stamps (tools/trace_dump.py 99): the K loop takes
 * 10.4 k cycles per item for 2 x 72 MFMAs x 32 cycles = 4.6 k cycles of matrix-pipe work per SIMD -- 44 % busy,
 * each (tap, channel block) step waits for its own LDS reads -- and staging, epilogue and the two barriers add
 * 4 k more during which the pipe idles. Here nothing is shared but the weights, and every latency has something
 * queued behind it:
 *   - ONE workgroup of 8 waves per CU (two per SIMD); the packed weight image and the bias go to LDS once (LDS-DMA);
 *   - each wave owns TWO private LDS patch buffers and processes UNITS = 8x8 output positions x all channels on its
 *     own, no workgroup barrier after the weights. The input patch of the NEXT unit is gathered by LDS-DMA
 *     (global_load_lds, 16 bytes per lane, no registers: per-lane source = pixel chunk or the zero-point line of
 *     the fill table, destination lane-linear, chunk-swizzled by patch row on the SOURCE side) while the current
 *     unit is multiplied; a counted vmcnt leaves it in flight;
 *   - one pass over a landed patch re-centres its bytes in place (a ^ 0x80) and leaves per-pixel channel sums beside
 *     it (v_sad_u8 + DPP), so that the K loop carries NOTHING but fragment reads and MFMAs: a wave issues roughly
 *     one instruction per four cycles, i.e. eight per 32-cycle MFMA, and the first version of this loop -- sixteen
 *     v_sad_u8, sixteen v_xor and the address arithmetic per eight MFMAs -- ran at half the matrix-pipe rate;
 *   - K loop software-pipelined in registers: the fragments of tap t+1 (activations from the patch, weights from
 *     the shared image) are read while the MFMAs of tap t issue; for the 3x3 / stride 1 / dilation 1 window every
 *     fragment address is a per-unit register plus an immediate;
 *   - the LDS-DMA instructions are inline asm: hipcc guards every LDS access behind a global_load_lds it knows of
 *     with s_waitcnt vmcnt(0) (it cannot prove the two do not alias), which would drain the gather of the next patch
 *     and wait for the previous unit's store acknowledgements several times per unit (see dma16);
 *   - epilogue: row term, Q31 requantization into the (now free) patch buffer as a 64 x n output image, whole
 *     8-position runs stored with 16-byte pieces (full 128-byte lines);
 *   - units are handed out inside the workgroup by an LDS counter.
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

constexpr int kWaves = 9;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxDma = 8;              // 1 KiB LDS-DMA pieces per patch (<= 512 chunks of 16 bytes)
constexpr uint32_t kFlip = 0x80808080u;
constexpr uint32_t kLdsLimit = 160 * 1024;

struct WaveArgs {
  uint32_t PH, PW;          // patch rows / columns
  uint32_t inv_pw;          // ceil(65536 / PW): q / PW == (q * inv_pw) >> 16 for the small q used here
  uint32_t tiles_x, tiles_y;
  uint32_t units;           // batch * tiles_y * tiles_x
  uint32_t inv_tiles;       // ceil(2^32 / (tiles_x * tiles_y)), 0 when the divisor is 1: unit / tiles = mulhi(unit, inv_tiles)
  uint32_t inv_tiles_x;     // the same for tiles_x (both exact while units * divisor < 2^32: make_args checks)
  uint32_t w_bytes;         // packed weight image
  uint32_t head_bytes;      // weights + bias + counter, 1024-aligned: offset of the first wave region
  uint32_t ndma;            // LDS-DMA pieces per patch
  uint32_t patch_bytes;     // one patch buffer (256-aligned; the last DMA piece is lane-masked to the patch)
  uint32_t stage_bytes;     // per wave: output staging image of 32 positions x n bytes
  uint32_t pix_bytes;       // per wave: per-pixel channel sums of the current patch (int32)
  uint32_t wave_bytes;      // per wave: 2 patch buffers + staging + pixel sums
};

inline bool make_args(const IgemmParams& p, const ConvGeom& g, uint32_t batch, WaveArgs* a, uint32_t* lds_bytes)
{
  a->PH = 7u * g.sh + (g.KH - 1u) * g.dh + 1u;
  a->PW = 7u * g.sw + (g.KW - 1u) * g.dw + 1u;
  a->inv_pw = (65536u + a->PW - 1u) / a->PW;
  a->tiles_x = (g.OW + 7u) / 8u;
  a->tiles_y = (g.OH + 7u) / 8u;
  a->units = batch * a->tiles_x * a->tiles_y;
  {
    // q = floor(n * M / 2^32) with M = ceil(2^32 / d) is exact while n * d < 2^32 (error term n*e/(d*2^32) < 1/d)
    const uint64_t tiles = static_cast<uint64_t>(a->tiles_x) * a->tiles_y;
    if (static_cast<uint64_t>(batch) * tiles * tiles >= (UINT64_C(1) << 32)) return false;
    a->inv_tiles = tiles > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + tiles - 1) / tiles) : 0u;
    a->inv_tiles_x = a->tiles_x > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + a->tiles_x - 1) / a->tiles_x) : 0u;
  }
  a->w_bytes = p.n_pad * p.k_pad;
  a->head_bytes = (a->w_bytes + p.n * 4u + 16u + 1023u) & ~1023u;
  const uint32_t patch = a->PH * a->PW * p.kc;
  a->ndma = (patch + 1023u) / 1024u;
  a->patch_bytes = (patch + 255u) & ~255u;
  a->stage_bytes = 32u * p.n;
  a->pix_bytes = (a->PH * a->PW * 4u + 255u) & ~255u;
  // (the staging image lives in the patch buffer of the unit just multiplied: its K loop is over, only its pixel sums
  //  are still needed and they have their own region)
  a->wave_bytes = 2u * a->patch_bytes + a->pix_bytes;
  if (a->stage_bytes > a->patch_bytes) return false;
  *lds_bytes = a->head_bytes + kWaves * a->wave_bytes;
  if (a->ndma > static_cast<uint32_t>(kMaxDma)) return false;
  if (a->ndma * 64u > 4096u * (p.kc >> 4)) return false;   // inv_pw exactness: pixel index < 4096
  if (a->w_bytes % 1024u != 0) return false;
  return *lds_bytes <= kLdsLimit;
}

// byte offset of an LDS pointer inside the workgroup's LDS allocation (what DS instructions and M0 address)
__device__ __forceinline__ uint32_t lds_off(const void* p)
{
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*) p));
}

/* LDS-DMA, 16 bytes per lane: LDS destination = wave-uniform base (M0) + lane * 16, source address per lane.
 * Issued as inline asm ON PURPOSE: with the builtin, hipcc (ROCm 7.2) remembers that a global_load_lds is in flight
 * and puts s_waitcnt vmcnt(0) in front of every later LDS access it cannot prove disjoint -- inside this kernel's unit
 * loop that is every ds_read / ds_write / LDS atomic, each draining the gather of the next patch (and waiting for
 * the previous unit's store acknowledgements on the way). The asm form is invisible to that bookkeeping; the waits
 * this kernel needs are the explicit s_waitcnt vmcnt in the unit loop, nothing else orders a ds_read behind a DMA
 * (cdna_hip_programming.md section 5.7). M0 is restored: the compiler owns it. */
__device__ __forceinline__ void dma16(const uint8_t* src, uint8_t* lds_wave_base)
{
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_off(lds_wave_base));
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

// 16-byte / 4-byte LDS stores the compiler does not see as LDS accesses (see the header: no vmcnt(0) in front)
__device__ __forceinline__ void ds_write16_raw(uint32_t off, uint4 v)
{
  typedef int raw_v4i __attribute__((ext_vector_type(4)));
  const raw_v4i x = {static_cast<int>(v.x), static_cast<int>(v.y), static_cast<int>(v.z), static_cast<int>(v.w)};
  asm volatile("ds_write_b128 %0, %1" :: "v"(off), "v"(x) : "memory");
}
__device__ __forceinline__ void ds_write4_raw(uint32_t off, int32_t v)
{
  asm volatile("ds_write_b32 %0, %1" :: "v"(off), "v"(v) : "memory");
}

// n / d through the host-made reciprocal (WaveArgs): one scalar multiply instead of the ~40-instruction sequence hipcc
// emits for a division by a run-time value -- four of those per unit sat on every wave's critical path
__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t inv)
{
  return inv != 0u ? __umulhi(n, inv) : n;
}

// KS: 3 = 3x3 window, stride 1, dilation 1 (10x10 patch; every fragment address = per-unit register + immediate),
//     0 = any geometry make_args accepts (addresses computed per tap)
// SEQ / FULL: the requantization flavour (requant.hip.h), chosen on the host: one kernel per flavour
template <int TN, int CB, int KS, int SEQ, bool FULL>
__global__ __launch_bounds__(kThreads, 2)
void q8_conv_wave_mfma_kernel(const IgemmParams p, const ConvGeom g, const WaveArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [weights][bias][counter][8 x (2 x patch | staging)]
  uint8_t* w_lds = lds;
  int32_t* bias_lds = reinterpret_cast<int32_t*>(lds + a.w_bytes);
  uint32_t* counter = reinterpret_cast<uint32_t*>(lds + a.w_bytes + p.n * 4u);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* patch0 = lds + a.head_bytes + wave * a.wave_bytes;
  int32_t* pix = reinterpret_cast<int32_t*>(patch0 + 2u * a.patch_bytes);

  // contiguous unit range of this workgroup (neighbouring blocks share halos in L2)
  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * a.units / gridDim.x);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * a.units / gridDim.x);

  constexpr uint32_t cin = CB * 32u;
  constexpr uint32_t log_cin = CB == 1 ? 5u : (CB == 2 ? 6u : 7u);
  constexpr uint32_t cpp = cin >> 4;               // 16-byte chunks per pixel
  constexpr uint32_t log_cpp = log_cin - 4u;
  const uint32_t sh_log = g.sh >> 1;               // strides 1 / 2
  const uint32_t pvec = a.PH * a.PW * cpp;
  const uint32_t tiles = a.tiles_x * a.tiles_y;
  const uint8_t* fill_line = p.fill_table + (p.izp_fill & 0xFFu) * 16u;   // sixteen bytes of the raw zero point
  const uint32_t khalf = lane >> 5;

  // ---- LDS-DMA gather of the input patch of `unit` into `dst` (a.ndma pieces of 1 KiB) ----
  // LDS image: [pixel q][cin bytes]; 16-byte slot s of pixel q holds chunk s ^ swz(row of q): the destination is
  // lane-linear by construction, so the swizzle is applied to the SOURCE chunk (and again on the fragment reads).
  // The gather pattern is the same for every unit: per piece u, this lane's patch pixel (py, px) and its byte offset
  // from the patch's first pixel are fixed (kept in registers), only the patch origin moves.
  uint32_t g_rel[kMaxDma], g_pyx[kMaxDma];
#pragma unroll
  for (int u = 0; u < kMaxDma; u++) {
    const uint32_t v = lane + u * 64u;
    const uint32_t s = v & (cpp - 1u);
    const uint32_t q = v >> log_cpp;
    const uint32_t py = (q * a.inv_pw) >> 16;
    const uint32_t px = q - py * a.PW;
    const uint32_t c = s ^ ((py >> sh_log) & (cpp - 1u));
    g_rel[u] = (py * g.W + px) * p.input_stride + c * 16u;
    g_pyx[u] = (py << 16) | px;
  }
  auto dma_patch = [&](uint32_t unit, uint8_t* dst) __attribute__((always_inline)) {
    const uint32_t img = div_magic(unit, a.inv_tiles);
    const uint32_t r = unit - img * tiles;
    const uint32_t tyi = div_magic(r, a.inv_tiles_x);
    const uint32_t txi = r - tyi * a.tiles_x;
    const int32_t iy0 = static_cast<int32_t>(tyi * 8u * g.sh) - static_cast<int32_t>(g.pad_top);
    const int32_t ix0 = static_cast<int32_t>(txi * 8u * g.sw) - static_cast<int32_t>(g.pad_left);
    // byte offset of the patch origin inside the tensor (may be "negative" for border units: wraps consistently in
    // 64-bit arithmetic below, and is only dereferenced for in-bounds pixels)
    const int64_t origin = static_cast<int64_t>(img) * static_cast<int64_t>(p.image_stride) +
        (static_cast<int64_t>(iy0) * static_cast<int64_t>(g.W) + ix0) * static_cast<int64_t>(p.input_stride);
    const uint8_t* base = p.input + origin;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + static_cast<int32_t>(a.PH) <= static_cast<int32_t>(g.H) &&
                          ix0 + static_cast<int32_t>(a.PW) <= static_cast<int32_t>(g.W);      // wave-uniform
#pragma unroll
    for (int u = 0; u < kMaxDma; u++) {
      if (static_cast<uint32_t>(u) < a.ndma) {
        const uint32_t v = lane + u * 64u;
        const uint8_t* src = base + g_rel[u];
        if (!interior) {
          const int32_t iy = iy0 + static_cast<int32_t>(g_pyx[u] >> 16);
          const int32_t ix = ix0 + static_cast<int32_t>(g_pyx[u] & 0xFFFFu);
          const bool inb = iy >= 0 && iy < static_cast<int32_t>(g.H) && ix >= 0 && ix < static_cast<int32_t>(g.W);
          src = inb ? src : fill_line;
        }
        if (v < pvec) dma16(src, dst + u * 1024u);        // lanes past the patch write nothing (the buffer ends there)
      }
    }
  };

  // ---- weights + bias + counter, once per workgroup; the first patch rides along ----
  {
    const uint32_t pieces = a.w_bytes >> 10;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(p.packed_w) + lane * 16u;
    for (uint32_t i = wave; i < pieces; i += kWaves) dma16(src + i * 1024u, w_lds + i * 1024u);
    for (uint32_t i = tid; i < p.n; i += kThreads) bias_lds[i] = p.bias2[i];
  }
  uint32_t cur = lo + wave;
  if (tid == 0) *counter = lo + kWaves;
  if (cur < hi) dma_patch(cur, patch0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's weight pieces and its first patch have landed
  __syncthreads();                                        // ... and everybody else's weights: the only workgroup barrier
  // (Holding the second half of the workgroup back by half a unit, so that the two waves of a SIMD alternate between
  //  multiplying and requantizing, was measured: s_sleep 0 / 56 / 110 -> 30.8 / 32.2 / 32.8 us on configs[2]. Not kept.)

  // this lane's two output positions inside a unit (fixed): i = j*32 + (lane & 31) -> (i >> 3, i & 7)
  uint32_t ty[2], rowbase[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const uint32_t i = j * 32u + (lane & 31u);
    ty[j] = i >> 3;
    rowbase[j] = (ty[j] * g.sh * a.PW + (i & 7u) * g.sw) << log_cin;
  }

  const uint32_t kblocks = p.k_pad / 32;
  const uint8_t* w_lane = w_lds + lane * 16;
  const uint32_t cpr = p.n >> 4;                   // 16-byte pieces per output position (2 or 4)
  const uint32_t log_cpr = 31u - __builtin_clz(cpr);
  const uint32_t taps = g.KH * g.KW;

  struct Frags {
    v4i a[2][CB];
    v4i w[TN][CB];
  };
  uint32_t buf = 0;
  uint32_t unit_no = 0;
  (void) unit_no;
#define CW_STAMP(slot) do { if (wave == 0) { QNNP_TRACE(p, blockIdx.x, unit_no, slot); } } while (0)
  while (cur < hi) {
    CW_STAMP(0);
    uint8_t* patch = patch0 + buf * a.patch_bytes;
    uint8_t* stage = patch;                  // (after the K loop, see make_args)
    // next unit: claimed now, its patch gathered into the other buffer while this unit is multiplied. (The patch of
    // `cur` has landed: its gather was waited for after the K loop of the previous unit, or before the barrier.)
    // (Claiming one unit further ahead, so that the LDS atomic's round trip leaves the critical path, measured
    //  SLOWER -- 28.9 against 27.8 us: a busy wave then sits on a unit an idle one could have taken in the last round.)
    uint32_t nxt = hi;
    {
      uint32_t claimed = 0;
      if (lane == 0) claimed = atomicAdd(counter, 1u);
      nxt = __builtin_amdgcn_readfirstlane(claimed);
    }
    if (nxt < hi) dma_patch(nxt, patch0 + (buf ^ 1u) * a.patch_bytes);
    CW_STAMP(1);
    const uint32_t img = div_magic(cur, a.inv_tiles);
    const uint32_t r = cur - img * tiles;
    const uint32_t tyi = div_magic(r, a.inv_tiles_x);
    const uint32_t oy0 = tyi * 8u;
    const uint32_t ox0 = (r - tyi * a.tiles_x) * 8u;

    // ---- the landed patch: re-centre the bytes in place, per-pixel channel sums (of a') beside it ----
    // Software-pipelined by one piece: one piece per trip (read -> wait -> compute -> write, the raw writes being
    // memory barriers to the compiler) cost an LDS round trip per piece, 2.4 k cycles per unit (stamps).
    {
      const uint32_t patch_off = lds_off(patch);
      const uint32_t pix_off = lds_off(pix);
      // (two pieces per trip, the read of the next piece issued before a piece is processed: two pieces live)
      auto read_piece = [&](uint32_t u) __attribute__((always_inline)) -> uint4 {
        const uint32_t v = lane + u * 64u;
        return *reinterpret_cast<const uint4*>(patch + min(v, pvec - 1u) * 16u);    // (past the patch: its last chunk, unused)
      };
      auto fix_piece = [&](uint32_t u, const uint4& x) __attribute__((always_inline)) {
        const uint32_t v = lane + u * 64u;
        if (u < a.ndma && v < pvec) {                                // whole pixels: pvec is a multiple of cpp
          uint32_t sum = __builtin_amdgcn_sad_u8(x.x, 0u, 0u);
          sum = __builtin_amdgcn_sad_u8(x.y, 0u, sum);
          sum = __builtin_amdgcn_sad_u8(x.z, 0u, sum);
          sum = __builtin_amdgcn_sad_u8(x.w, 0u, sum);
          sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0xB1, 0xF, 0xF, false));        // lane ^ 1
          if (cpp > 2) sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0x4E, 0xF, 0xF, false));   // lane ^ 2
          uint4 y = x;
          y.x ^= kFlip; y.y ^= kFlip; y.z ^= kFlip; y.w ^= kFlip;
          ds_write16_raw(patch_off + v * 16u, y);
          if ((v & (cpp - 1u)) == 0) ds_write4_raw(pix_off + (v >> log_cpp) * 4u, static_cast<int32_t>(sum) - 128 * static_cast<int32_t>(cin));
        }
      };
      uint4 xa = read_piece(0);
      for (uint32_t u = 0; u < a.ndma; u += 2) {
        const uint4 xb = read_piece(u + 1);
        fix_piece(u, xa);
        xa = read_piece(u + 2);
        fix_piece(u + 1, xb);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    CW_STAMP(2);

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

    auto mma = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
      for (int cb = 0; cb < CB; cb++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++)
            acc[j][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[tn][cb], f.a[j][cb], acc[j][tn], 0, 0, 0);
    };

    if constexpr (KS == 3) {
      // 3x3, stride 1, dilation 1: patch row pitch 10 pixels. Fragment address of (ky, kx, j, cb) =
      //   patch + rowbase[j] + ky * 10 * cin + slot(ky, j, cb) * 16   [one register per (ky, j, cb)]   + kx * cin [immediate]
      const uint8_t* abase[3][2][CB];
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const uint32_t swz = (ty[j] + ky) & (cpp - 1u);
#pragma unroll
          for (int cb = 0; cb < CB; cb++) {
            abase[ky][j][cb] = patch + rowbase[j] + ky * 10 * cin + ((((cb << 1) | khalf) ^ swz) << 4);
          }
        }
      auto read3 = [&](auto t_c, Frags& f) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        constexpr int ky = t / 3, kx = t % 3;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int cb = 0; cb < CB; cb++) f.a[j][cb] = *reinterpret_cast<const v4i*>(abase[ky][j][cb] + kx * cin);
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
          for (int cb = 0; cb < CB; cb++)
            f.w[tn][cb] = *reinterpret_cast<const v4i*>(w_lane + (tn * kblocks + t * CB + cb) * 1024u);
      };
      Frags f0, f1;
      read3(std::integral_constant<int, 0>{}, f0);
      read3(std::integral_constant<int, 1>{}, f1); mma(f0);
      read3(std::integral_constant<int, 2>{}, f0); mma(f1);
      read3(std::integral_constant<int, 3>{}, f1); mma(f0);
      read3(std::integral_constant<int, 4>{}, f0); mma(f1);
      read3(std::integral_constant<int, 5>{}, f1); mma(f0);
      read3(std::integral_constant<int, 6>{}, f0); mma(f1);
      read3(std::integral_constant<int, 7>{}, f1); mma(f0);
      read3(std::integral_constant<int, 8>{}, f0); mma(f1);
      mma(f0);
    } else {
      // any window: addresses per tap
      auto read_frags = [&](uint32_t t, uint32_t ky, uint32_t kx, Frags& f) __attribute__((always_inline)) {
        const uint32_t tap_off = (ky * g.dh * a.PW + kx * g.dw) << log_cin;       // wave-uniform
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const uint32_t swz = ((ty[j] * g.sh + ky * g.dh) >> sh_log) & (cpp - 1u);
          const uint8_t* base = patch + rowbase[j] + tap_off;
#pragma unroll
          for (int cb = 0; cb < CB; cb++) {
            f.a[j][cb] = *reinterpret_cast<const v4i*>(base + ((((cb << 1) | khalf) ^ swz) << 4));
          }
        }
        const uint8_t* wt = w_lane + (t * CB) * 1024u;
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
          for (int cb = 0; cb < CB; cb++) {
            f.w[tn][cb] = *reinterpret_cast<const v4i*>(wt + (tn * kblocks + cb) * 1024u);
          }
      };
      // two taps per trip, two register sets: the reads of tap t+1 are issued before the multiplies of tap t
      Frags f0, f1;
      read_frags(0, 0, 0, f0);
      uint32_t ky = 0, kx = 0;
      auto advance = [&]() __attribute__((always_inline)) { if (++kx == g.KW) { kx = 0; ky++; } };
      uint32_t t = 0;
      while (t + 2 <= taps) {
        advance();
        read_frags(t + 1, ky, kx, f1);
        mma(f0);
        advance();
        if (t + 2 < taps) read_frags(t + 2, ky, kx, f0);
        mma(f1);
        t += 2;
      }
      if (t < taps) mma(f0);
    }
    CW_STAMP(3);
    // Everything this wave has in flight -- the gather of `nxt` and the output stores of the previous unit -- was
    // issued before the K loop: by now it has landed, so this wait is (almost) free; it is what makes `nxt`'s patch
    // safe to read at the top of the next iteration without waiting for THIS unit's stores.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    CW_STAMP(4);

    // ---- fused epilogue, 32 positions at a time: row term from the pixel sums, Q31 requantization into the staging
    //      image, whole 4-row x 8-position runs stored with 16-byte pieces ----
    uint8_t* out_img = p.output + static_cast<uint64_t>(img) * g.OH * g.OW * p.n;
    const uint32_t stage_off = lds_off(stage);
    {
      const std::integral_constant<int, SEQ> shift0{};
      const std::integral_constant<bool, FULL> full{};
      (void) shift0; (void) full;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        // sum of a' over this position's window: the window's pixel sums
        int32_t s = 0;
        const int32_t* pq = pix + (ty[j] * g.sh * a.PW + ((j * 32u + (lane & 31u)) & 7u) * g.sw);
        if constexpr (KS == 3) {
#pragma unroll
          for (int t = 0; t < 9; t++) s += pq[(t / 3) * 10 + (t % 3)];
        } else {
          for (uint32_t ky = 0; ky < g.KH; ky++)
            for (uint32_t kx = 0; kx < g.KW; kx++) s += pq[ky * g.dh * a.PW + kx * g.dw];
        }
        // (+ 2^31 for the offset rounding sequences, requant.hip.h: free here)
        const int32_t rowterm = with_rq_offset<decltype(shift0)::value>(p.row_coeff * s);
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
          // (igemm_stage_tile, with the LDS store as a raw ds_write: see the header)
          uint32_t pk[4];
#pragma unroll
          for (int rg = 0; rg < 4; rg++) {
            pk[rg] = q31_requantize_pack4<decltype(shift0)::value, decltype(full)::value, false>(
                add_wrap(acc[j][tn][rg * 4 + 0], rowterm), add_wrap(acc[j][tn][rg * 4 + 1], rowterm),
                add_wrap(acc[j][tn][rg * 4 + 2], rowterm), add_wrap(acc[j][tn][rg * 4 + 3], rowterm), p.rq);
          }
          const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
          ds_write16_raw(stage_off + (lane & 31u) * p.n + tn * 32 + khalf * 16, make_uint4(s02[0], s02[1], s13[0], s13[1]));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // image complete before it is read back
        const uint32_t pieces = 32u * cpr;
#pragma unroll
        for (int tt = 0; tt < 2; tt++) {
          const uint32_t idx = lane + tt * 64u;
          const uint32_t i = j * 32u + (idx >> log_cpr);
          const uint32_t ch = idx & (cpr - 1u);
          const uint32_t oy = oy0 + (i >> 3);
          const uint32_t ox = ox0 + (i & 7u);
          if (idx < pieces && oy < g.OH && ox < g.OW) {
            *reinterpret_cast<uint4*>(out_img + (static_cast<uint64_t>(oy) * g.OW + ox) * p.n + ch * 16u) =
                *reinterpret_cast<const uint4*>(stage + idx * 16u);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // read back before the next half overwrites it
      }
    }
    CW_STAMP(5);
    unit_no++;
    cur = nxt;
    buf ^= 1u;
  }
#undef CW_STAMP
}

/*
 * 3x3 / stride 1 / dilation 1 with the patch going through REGISTERS instead of LDS-DMA. Stamps on the kernel above
 * put the issue of a patch's seven LDS-DMA pieces at 1.7-2.4 k cycles per unit -- 250+ cycles each on a CU this busy,
 * wherever they are placed (between the taps of the K loop they lengthen it by the same 2 k) -- and the fix-up pass
 * reads the landed patch back from LDS only to re-centre it. Here the next unit's patch is fetched with plain 16-byte
 * global loads issued between the K loop and the epilogue (28 registers, held while the epilogue works), and the
 * fix-up pass of that unit re-centres them on their way INTO the (single) patch buffer: no LDS-DMA, no read-back, half
 * the patch LDS -- twelve waves per workgroup, three per SIMD. The epilogue's four stores are unconditional buffer
 * stores (positions outside the image get an out-of-range offset), so the wait for the fetched patch is a counted
 * vmcnt that does not cover them.
 */
/* TM: 32-position groups per unit -- 2: 8x8 positions (10x10-pixel patch), twelve waves per workgroup; 1: 4 rows x 8
 * positions (6x10 patch), sixteen waves. The small unit reads half again as many fragments per MFMA, but it halves
 * what the last, partly filled round of units costs: with 24.5 units of 8x8 per CU and twelve waves that round --
 * one unit on every other CU -- was a quarter of the launch. */
template <int TM> constexpr int reg_waves() { return TM == 2 ? 12 : 16; }

/* the unit geometry of the register-path kernel (its own tiling of the output in TM x 4 rows) */
template <int TM>
inline WaveArgs reg_args(const WaveArgs& a0, const IgemmParams& p, const ConvGeom& g, uint32_t batch, bool* ok)
{
  WaveArgs a = a0;
  a.PH = TM * 4u + 2u;
  a.PW = 10u;
  a.tiles_x = (g.OW + 7u) / 8u;
  a.tiles_y = (g.OH + TM * 4u - 1u) / (TM * 4u);
  a.units = batch * a.tiles_x * a.tiles_y;
  const uint64_t tiles = static_cast<uint64_t>(a.tiles_x) * a.tiles_y;
  *ok = static_cast<uint64_t>(batch) * tiles * tiles < (UINT64_C(1) << 32);
  a.inv_tiles = tiles > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + tiles - 1) / tiles) : 0u;
  a.inv_tiles_x = a.tiles_x > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + a.tiles_x - 1) / a.tiles_x) : 0u;
  const uint32_t patch = a.PH * a.PW * p.kc;
  a.patch_bytes = (patch + 255u) & ~255u;
  a.pix_bytes = (a.PH * a.PW * 4u + 255u) & ~255u;
  if (a.stage_bytes > a.patch_bytes) *ok = false;
  return a;
}

template <int TM>
inline uint32_t reg_lds_bytes(const WaveArgs& a) { return a.head_bytes + reg_waves<TM>() * (a.patch_bytes + a.pix_bytes); }

template <int TM, int TN, int CB, int SEQ, bool FULL>
__global__ __launch_bounds__(reg_waves<TM>() * 64, reg_waves<TM>() / 4)
void q8_conv_wave_reg_kernel(const IgemmParams p, const ConvGeom g, const WaveArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [weights][bias][counter][waves x (patch | pixel sums)]
  constexpr int kRegWaves = reg_waves<TM>();
  constexpr uint32_t kPR = TM * 4u + 2u;           // patch rows (10 columns)
  constexpr int NP = (kPR * 10u * CB * 2u + 63u) / 64u;   // 1 KiB pieces of the patch
  uint8_t* w_lds = lds;
  int32_t* bias_lds = reinterpret_cast<int32_t*>(lds + a.w_bytes);
  uint32_t* counter = reinterpret_cast<uint32_t*>(lds + a.w_bytes + p.n * 4u);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave == 0) { QNNP_TRACE(p, blockIdx.x, 3, 0); }       // (measurement builds: kernel entry)
  uint8_t* patch = lds + a.head_bytes + wave * (a.patch_bytes + a.pix_bytes);
  uint8_t* stage = patch;                          // the staging image lives in the patch once its K loop is over
  int32_t* pix = reinterpret_cast<int32_t*>(patch + a.patch_bytes);

  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * a.units / gridDim.x);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * a.units / gridDim.x);

  constexpr uint32_t cin = CB * 32u;
  constexpr uint32_t log_cin = CB == 1 ? 5u : 6u;
  constexpr uint32_t cpp = cin >> 4;               // 16-byte chunks per pixel
  constexpr uint32_t log_cpp = log_cin - 4u;
  const uint32_t pvec = kPR * 10u * cpp;           // chunks of the patch
  const uint32_t tiles = a.tiles_x * a.tiles_y;
  const uint8_t* fill_line = p.fill_table + (p.izp_fill & 0xFFu) * 16u;   // sixteen bytes of the raw zero point
  const uint32_t khalf = lane >> 5;

  // ---- fetch of a unit's patch: NP x 16 bytes per lane, chunk v = lane + 64 u -> pixel v / cpp (10 per row), source
  //      chunk (v % cpp) ^ swz(row); pixels outside the image read the zero-point line; chunks past the patch re-read
  //      its last one (never written to LDS). Every load is unconditional.
  uint32_t lane_now = lane;      // refreshed through an empty asm once per unit: keeps the per-piece address values from
                                 // being hoisted out of the unit loop and held in registers across the K loop
  struct Raw { v4i x[NP]; };
  struct Src { const uint8_t* base; int32_t iy0, ix0; };
  auto patch_source = [&](uint32_t unit) __attribute__((always_inline)) -> Src {
    const uint32_t img = div_magic(unit, a.inv_tiles);
    const uint32_t rr = unit - img * tiles;
    const uint32_t tyi = div_magic(rr, a.inv_tiles_x);
    const uint32_t txi = rr - tyi * a.tiles_x;
    Src sc;
    sc.iy0 = static_cast<int32_t>(tyi * (TM * 4u)) - static_cast<int32_t>(g.pad_top);
    sc.ix0 = static_cast<int32_t>(txi * 8u) - static_cast<int32_t>(g.pad_left);
    const int64_t origin = static_cast<int64_t>(img) * static_cast<int64_t>(p.image_stride) +
        (static_cast<int64_t>(sc.iy0) * static_cast<int64_t>(g.W) + sc.ix0) * static_cast<int64_t>(p.input_stride);
    sc.base = p.input + origin;
    return sc;
  };
  auto fetch_piece = [&](const Src& sc, int u, Raw& r) __attribute__((always_inline)) {
    const uint32_t v = min(lane_now + u * 64u, pvec - 1u);
    const uint32_t s = v & (cpp - 1u);
    const uint32_t q = v >> log_cpp;
    const uint32_t py = (q * 6554u) >> 16;                 // q / 10 for q < 100
    const uint32_t px = q - py * 10u;
    const uint32_t c = s ^ (py & (cpp - 1u));
    const int32_t iy = sc.iy0 + static_cast<int32_t>(py);
    const int32_t ix = sc.ix0 + static_cast<int32_t>(px);
    const bool inb = iy >= 0 && iy < static_cast<int32_t>(g.H) && ix >= 0 && ix < static_cast<int32_t>(g.W);
    const uint8_t* src = inb ? sc.base + ((py * g.W + px) * p.input_stride + c * 16u) : fill_line;
    r.x[u] = *reinterpret_cast<const v4i*>(src);
  };
  auto fetch_patch = [&](uint32_t unit, Raw& r) __attribute__((always_inline)) {
    const Src sc = patch_source(unit);
#pragma unroll
    for (int u = 0; u < NP; u++) fetch_piece(sc, u, r);
  };

  // ---- the first patch is requested first (its HBM round trip is the longest thing in the prologue: 6.5 k cycles from
  //      kernel entry to "everything landed" by the stamps), then weights + bias + counter, once per workgroup. The
  //      wait for the weights and the workgroup's only barrier sit AFTER the first unit's fix-up pass, which needs
  //      the patch only. ----
  uint32_t cur = lo + wave;
  Raw raw;
  fetch_patch(min(cur, a.units - 1u), raw);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>(a.units / tiles * g.OH * g.OW * p.n), 0x00020000);   // (launcher: < 2^31)
  // Stores to nowhere (out-of-range offset: the hardware drops them), as many as a unit's epilogue issues. hipcc sizes
  // the vmcnt waits of the fix-up pass for the smaller of the counts outstanding on the two ways into the loop; coming
  // from here that was "NP loads", from the loop's end "NP loads + 2 TM stores", so the waits came out as
  // vmcnt(NP-1..0) and covered the previous unit's stores. With the same sequence on both ways in they are counted
  // past the stores. (Distinct offsets and a constant payload: identical stores are merged, a fetched register as
  // payload is a wait. They come BEFORE the weight pieces, which the compiler does not see: its counts then never
  // reach into them.)
#pragma unroll
  for (int i = 0; i < 2 * TM; i++) {
    const v4i nothing = {0, 0, 0, 0};
    __builtin_amdgcn_raw_buffer_store_b128(
        __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, nothing), out_rsrc,
        0xFFFFFF00u + static_cast<uint32_t>(i) * 16u, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    const uint32_t pieces = a.w_bytes >> 10;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(p.packed_w) + lane * 16u;
    for (uint32_t i = wave; i < pieces; i += kRegWaves) dma16(src + i * 1024u, w_lds + i * 1024u);
    // (the bias the same way, 16 bytes per lane of one wave: a load + LDS store would put a compiler-made vmcnt(0)
    //  -- patch, weights and all -- in front of the first unit)
    if (wave == kRegWaves - 1 && lane < p.n / 4u) {
      dma16(reinterpret_cast<const uint8_t*>(p.bias2) + lane * 16u, reinterpret_cast<uint8_t*>(bias_lds));
    }
  }
  // (a wave's first TWO units are fixed -- lo + wave and lo + waves + wave -- so that its first claim on the counter
  //  comes after the barrier that publishes it)
  if (tid == 0) *counter = lo + 2u * kRegWaves;
  bool weights_pending = true;                            // (wave-uniform)

  uint32_t ty[TM], rowbase[TM];
#pragma unroll
  for (int j = 0; j < TM; j++) {
    const uint32_t i = j * 32u + (lane & 31u);
    ty[j] = i >> 3;
    rowbase[j] = (ty[j] * 10u + (i & 7u)) << log_cin;
  }
  const uint32_t kblocks = p.k_pad / 32;
  const uint8_t* w_lane = w_lds + lane * 16;
  const uint32_t cpr = p.n >> 4;                   // 16-byte pieces per output position (2 or 4)
  const uint32_t log_cpr = 31u - __builtin_clz(cpr);
  struct Frags {
    v4i a[TM][CB];
    v4i w[TN][CB];
  };
  uint32_t unit_no = 0;
  (void) unit_no;
#define CR_STAMP(slot) do { if (wave == 0) { QNNP_TRACE(p, blockIdx.x, unit_no, slot); } } while (0)
  while (cur < hi) {
    CR_STAMP(0);
    asm volatile("" : "+v"(lane_now));
    uint32_t claimed = cur + kRegWaves;                     // first unit: the next one is fixed (see above)
    if (!weights_pending) {
      claimed = 0;
      if (lane == 0) claimed = atomicAdd(counter, 1u);      // the unit after this one (read before the K loop)
    }
    const uint32_t img = div_magic(cur, a.inv_tiles);
    const uint32_t r = cur - img * tiles;
    const uint32_t tyi = div_magic(r, a.inv_tiles_x);
    const uint32_t oy0 = tyi * (TM * 4u);
    const uint32_t ox0 = (r - tyi * a.tiles_x) * 8u;

    // ---- the fetched patch: re-centred into LDS, per-pixel channel sums (of a') beside it ----
    {
      const uint32_t patch_off = lds_off(patch);
      const uint32_t pix_off = lds_off(pix);
#pragma unroll
      for (int u = 0; u < NP; u++) {
        const uint32_t v = lane_now + u * 64u;             // (lane_now: recomputed per unit, not held across the K loop)
        if (v < pvec) {                                    // whole pixels: pvec is a multiple of cpp
          const v4i x = raw.x[u];
          uint32_t sum = __builtin_amdgcn_sad_u8(x.x, 0u, 0u);
          sum = __builtin_amdgcn_sad_u8(x.y, 0u, sum);
          sum = __builtin_amdgcn_sad_u8(x.z, 0u, sum);
          sum = __builtin_amdgcn_sad_u8(x.w, 0u, sum);
          sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0xB1, 0xF, 0xF, false));        // lane ^ 1
          if (cpp > 2) sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0x4E, 0xF, 0xF, false));   // lane ^ 2
          ds_write16_raw(patch_off + v * 16u, make_uint4(x.x ^ kFlip, x.y ^ kFlip, x.z ^ kFlip, x.w ^ kFlip));
          if ((v & (cpp - 1u)) == 0) ds_write4_raw(pix_off + (v >> log_cpp) * 4u, static_cast<int32_t>(sum) - 128 * static_cast<int32_t>(cin));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (weights_pending) {                                  // first unit only: the weights, for everybody
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      weights_pending = false;
    }
    CR_STAMP(1);
    // next unit (claimed at the top: the atomic's round trip passed under the fix-up); past the range: some valid unit
    const uint32_t nxt = __builtin_amdgcn_readfirstlane(claimed);
    const Src nsrc = patch_source(min(nxt, a.units - 1u));

    // accumulators start at the folded bias
    v16i acc[TM][TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const v4i b = *reinterpret_cast<const v4i*>(bias_lds + tn * 32 + rg * 8 + khalf * 4);
#pragma unroll
        for (int j = 0; j < TM; j++) {
          acc[j][tn][rg * 4 + 0] = b.x;
          acc[j][tn][rg * 4 + 1] = b.y;
          acc[j][tn][rg * 4 + 2] = b.z;
          acc[j][tn][rg * 4 + 3] = b.w;
        }
      }
    auto mma = [&](const Frags& f) __attribute__((always_inline)) {
#pragma unroll
      for (int cb = 0; cb < CB; cb++)
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++)
            acc[j][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[tn][cb], f.a[j][cb], acc[j][tn], 0, 0, 0);
    };
    {
      const uint8_t* abase[3][TM][CB];
#pragma unroll
      for (int ky = 0; ky < 3; ky++)
#pragma unroll
        for (int j = 0; j < TM; j++) {
          const uint32_t swz = (ty[j] + ky) & (cpp - 1u);
#pragma unroll
          for (int cb = 0; cb < CB; cb++) {
            abase[ky][j][cb] = patch + rowbase[j] + ky * 10 * cin + ((((cb << 1) | khalf) ^ swz) << 4);
          }
        }
      auto read3 = [&](auto t_c, Frags& f) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        constexpr int ky = t / 3, kx = t % 3;
#pragma unroll
        for (int j = 0; j < TM; j++)
#pragma unroll
          for (int cb = 0; cb < CB; cb++) f.a[j][cb] = *reinterpret_cast<const v4i*>(abase[ky][j][cb] + kx * cin);
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
#pragma unroll
          for (int cb = 0; cb < CB; cb++)
            f.w[tn][cb] = *reinterpret_cast<const v4i*>(w_lane + (tn * kblocks + t * CB + cb) * 1024u);
      };
      // the next unit's patch is requested piece by piece between the taps: twelve waves asking for their seven
      // pieces at the same moment queued on the CU's address unit for 1.4-2.9 k cycles (stamps); behind a tap's
      // eight queued MFMAs the wait for the address unit costs nothing
      auto piece = [&](int u) __attribute__((always_inline)) { if (u < NP) fetch_piece(nsrc, u, raw); };
      Frags f0, f1;
      read3(std::integral_constant<int, 0>{}, f0);
      read3(std::integral_constant<int, 1>{}, f1); mma(f0);
      read3(std::integral_constant<int, 2>{}, f0); mma(f1); piece(0);
      read3(std::integral_constant<int, 3>{}, f1); mma(f0); piece(1);
      read3(std::integral_constant<int, 4>{}, f0); mma(f1); piece(2);
      read3(std::integral_constant<int, 5>{}, f1); mma(f0); piece(3);
      read3(std::integral_constant<int, 6>{}, f0); mma(f1); piece(4);
      read3(std::integral_constant<int, 7>{}, f1); mma(f0); piece(5);
      read3(std::integral_constant<int, 8>{}, f0); mma(f1); piece(6);
      mma(f0);
    }
    CR_STAMP(2);
    CR_STAMP(3);
    __builtin_amdgcn_sched_barrier(0);
    CR_STAMP(4);

    // ---- fused epilogue, 32 positions at a time (as the kernel above), stores through the buffer descriptor ----
    const uint32_t out_img = img * g.OH * g.OW * p.n;
    const uint32_t stage_off = lds_off(stage);
#pragma unroll
    for (int j = 0; j < TM; j++) {
      int32_t s = 0;
      const int32_t* pq = pix + (ty[j] * 10u + ((j * 32u + (lane & 31u)) & 7u));
#pragma unroll
      for (int t = 0; t < 9; t++) s += pq[(t / 3) * 10 + (t % 3)];
      const int32_t rowterm = with_rq_offset<SEQ>(p.row_coeff * s);
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        uint32_t pk[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          pk[rg] = q31_requantize_pack4<SEQ, FULL, false>(
              add_wrap(acc[j][tn][rg * 4 + 0], rowterm), add_wrap(acc[j][tn][rg * 4 + 1], rowterm),
              add_wrap(acc[j][tn][rg * 4 + 2], rowterm), add_wrap(acc[j][tn][rg * 4 + 3], rowterm), p.rq);
        }
        const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
        ds_write16_raw(stage_off + (lane & 31u) * p.n + tn * 32 + khalf * 16, make_uint4(s02[0], s02[1], s13[0], s13[1]));
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // image complete before it is read back
      const uint32_t pieces = 32u * cpr;
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const uint32_t idx = lane + tt * 64u;
        const uint32_t i = j * 32u + (idx >> log_cpr);
        const uint32_t ch = idx & (cpr - 1u);
        const uint32_t oy = oy0 + (i >> 3);
        const uint32_t ox = ox0 + (i & 7u);
        const bool ok = idx < pieces && oy < g.OH && ox < g.OW;
        const v4i v = *reinterpret_cast<const v4i*>(stage + min(idx, pieces - 1u) * 16u);
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), out_rsrc,
            ok ? out_img + (oy * g.OW + ox) * p.n + ch * 16u : 0xFFFFFFF0u, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");         // read back before the next half overwrites it
    }
    CR_STAMP(5);
    unit_no++;
    cur = nxt;
  }
#undef CR_STAMP
  if (weights_pending) {                                    // a wave without a unit still owes the barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (wave == 0) { QNNP_TRACE(p, blockIdx.x, 3, 3); QNNP_TRACE(p, blockIdx.x, 3, 4); QNNP_TRACE(p, blockIdx.x, 3, 5); }
}


/*
 * The WEIGHT-STATIONARY flavour (round 3; default for 3x3 / stride 1): a wave keeps EVERY weight fragment of the layer
 * in registers -- 9 taps x CB x TN fragments of 16 bytes per lane, 144 VGPRs for 64 -> 64 channels -- for its whole
 * life, so that the K loop of a unit (4 rows x 8 positions x all channels, 36 MFMAs) reads nothing from LDS but the two
 * activation fragments of a tap: 18 ds_read_b128 per unit instead of 54. What the counters said about the register-path
 * kernel above (profiles/r03/pmc_conv3x3_*: first PMC passes this kernel family ever got): matrix pipe 36 % busy, the
 * LDS array 49 % busy of which a sixth bank conflicts, waves stalled on instruction issue 45 % of their cycles with
 * four of them per SIMD walking fetch / fix-up / K loop / epilogue as one chain each. Here: two waves per SIMD
 * (the register file allows no more), a third of the LDS traffic, and per unit a chain of K loop (matrix pipe) and
 * epilogue + fix-up (VALU) of about the same length, so the two waves of a SIMD settle into opposite phases.
 * The weights go through LDS once per workgroup (LDS-DMA, 36 KiB), every wave then reads its copy with 36 ds_read_b128.
 */
constexpr int kWsWaves = 8;
constexpr uint32_t kWsPixBytes = 1024;             // per wave: one dword per 16-byte patch chunk (the pixel's sum, replicated)

inline uint32_t ws_lds_bytes(const WaveArgs& a) { return a.head_bytes + kWsWaves * (a.patch_bytes + kWsPixBytes); }

/* What the first two versions of this kernel taught (stamps + PMC, profiles/r03): with weights in registers, with two
 * patches in flight, with the epilogue software-pipelined into the next K loop -- always 4.1-4.3 k cycles per unit and
 * wave, two waves per SIMD. 2 x ~520 instructions in 4.15 k cycles is ONE INSTRUCTION PER FOUR CYCLES PER SIMD, and the
 * round-2 kernel (four waves of 572 instructions per 9.4 k cycles) sits on the same line: a SIMD issues one instruction
 * every four cycles whatever the number of waves and whatever their mix. A unit's 36 MFMAs occupy the matrix pipe for
 * 1152 cycles = 288 issue slots: the kernel is bound by its INSTRUCTION COUNT, and the budget is 8 instructions per
 * MFMA. This version is written against that budget:
 *   - the patch is fetched through a buffer descriptor with per-lane offsets that are constants of the kernel (one
 *     v_add per piece; the 64-bit address arithmetic, bounds tests and selects of the flat-address version were ~100
 *     instructions per unit); pixels outside the image are fetched from wherever the offset lands (in range: some other
 *     pixel; out of range: the hardware returns zeros) and REPLACED by the zero point in the fix-up pass of border
 *     units only (wave-uniform branch; 61 % of the units of a 56 x 56 image are interior);
 *   - unit coordinates in 32-bit scalar arithmetic (the tensor is < 2^31 bytes: launcher);
 *   - every lane writes its pixel sum (no exec-mask dance around a quarter-populated store). */
/* CEN (round 4): the zero-point-centred image of q8gemm256c.hip / pack.h -- kernel zero point 128 (the standard image, whose
 * row coefficient is zero) or 127 (weights w ^ 0x7F, activations ^ 0x7F): no kernel-zero-point row term, hence no pixel
 * sums in the fix-up pass (4 v_sad_u8 + 2 DPP adds + a store per piece) and no window sum in the epilogue (9 LDS reads +
 * 8 adds + the addend): ~55 of a unit's ~420 instructions. p.a_flip carries the mask. */
template <int TN, int CB, int SEQ, bool FULL, bool CEN = false>
__global__ __launch_bounds__(kWsWaves * 64, 2)
void q8_conv_wave_ws_kernel(const IgemmParams p, const ConvGeom g, const WaveArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [weights][bias][counter][waves x (patch | pixel sums)]
  // (measurement builds: the prologue's stamps go to "workgroup" blockIdx.x + 256 of the trace buffer -- item 0: cycles at
  //  entry / weights requested / first patch requested / patch in LDS / barrier passed / weights in registers / loop done,
  //  item 1: the 100 MHz wall clock at entry and exit. tools/trace_conv33.py prints them.)
#define WS_PRO(slot) QNNP_TRACE(p, blockIdx.x + 256u, 0, slot)
  WS_PRO(0);
  QNNP_TRACE_WALL(p, blockIdx.x + 256u, 1, 0);
  constexpr uint32_t kPR = 6u;                     // patch rows (10 columns) of a 4 x 8 unit
  constexpr int NP = (kPR * 10u * CB * 2u + 63u) / 64u;   // 1 KiB pieces of the patch
  uint8_t* w_lds = lds;
  int32_t* bias_lds = reinterpret_cast<int32_t*>(lds + a.w_bytes);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* patch = lds + a.head_bytes + wave * (a.patch_bytes + kWsPixBytes);
  int32_t* pix = reinterpret_cast<int32_t*>(patch + a.patch_bytes);

  // ---- prologue, first half: weights + bias by LDS-DMA, once per workgroup. FIRST (round 5): nothing but the argument
  //      block stands in front of these requests, while the first patch's addresses take ~1.8 k cycles of cold-start
  //      arithmetic (profiles/r05/conv3x3_prologue_and_unit_stamps_r05trace.txt) -- and the barrier waits for the weights
  {
    const uint32_t pieces = a.w_bytes >> 10;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(p.packed_w) + lane * 16u;
    for (uint32_t i = wave; i < pieces; i += kWsWaves) dma16(src + i * 1024u, w_lds + i * 1024u);
    if (wave == kWsWaves - 1 && lane < p.n / 4u) {
      // (lane forms of the requantization: bias + 2^31, the second half of the pair table)
      dma16(reinterpret_cast<const uint8_t*>(rq_is_lane<SEQ>() ? p.bias2u : p.bias2) + lane * 16u, reinterpret_cast<uint8_t*>(bias_lds));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  WS_PRO(1);

  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * a.units / gridDim.x);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * a.units / gridDim.x);

  constexpr uint32_t cin = CB * 32u;
  constexpr uint32_t log_cin = CB == 1 ? 5u : 6u;
  constexpr uint32_t cpp = cin >> 4;               // 16-byte chunks per pixel
  constexpr uint32_t log_cpp = log_cin - 4u;
  constexpr uint32_t pvec = kPR * 10u * cpp;       // chunks of the patch
  const uint32_t tiles = a.tiles_x * a.tiles_y;
  const uint32_t fill4 = p.izp_fill;               // the raw zero point in every byte
  const uint32_t khalf = lane >> 5;

  // ---- the gather pattern of a patch is the same for every unit: per piece u this lane's chunk v = lane + 64 u is pixel
  //      (py, px) of the patch, source chunk (v % cpp) ^ swz(py); only the patch origin moves ----
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(a.units / tiles * p.image_stride), 0x00020000);   // (launcher: < 2^31)
  uint32_t rel[NP], pyx[NP];
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const uint32_t v = min(lane + u * 64u, pvec - 1u);
    const uint32_t sl = v & (cpp - 1u);
    const uint32_t q = v >> log_cpp;
    const uint32_t py = (q * 6554u) >> 16;                 // q / 10 for q < 100
    const uint32_t px = q - py * 10u;
    rel[u] = (py * g.W + px) * p.input_stride + ((sl ^ (py & (cpp - 1u))) << 4);
    pyx[u] = (py << 16) | px;
  }
  struct Raw { v4i x[NP]; };
  struct Where { uint32_t origin; int32_t iy0, ix0; uint32_t out_img, oy0, ox0; bool border; };   // (wave-uniform)
  auto locate = [&](uint32_t unit) __attribute__((always_inline)) -> Where {
    const uint32_t img = div_magic(unit, a.inv_tiles);
    const uint32_t rr = unit - img * tiles;
    const uint32_t tyi = div_magic(rr, a.inv_tiles_x);
    const uint32_t txi = rr - tyi * a.tiles_x;
    Where w;
    w.oy0 = tyi * 4u;
    w.ox0 = txi * 8u;
    w.iy0 = static_cast<int32_t>(w.oy0) - static_cast<int32_t>(g.pad_top);
    w.ix0 = static_cast<int32_t>(w.ox0) - static_cast<int32_t>(g.pad_left);
    // byte offset of the patch's first pixel inside the tensor, modulo 2^32: negative for the first rows / columns --
    // those lanes then see an offset past the descriptor's range (zeros) or some other pixel, and the fix-up replaces them
    w.origin = img * static_cast<uint32_t>(p.image_stride) +
        static_cast<uint32_t>(w.iy0 * static_cast<int32_t>(g.W) + w.ix0) * p.input_stride;
    w.out_img = img * g.OH * g.OW * p.n;
    w.border = w.iy0 < 0 || w.ix0 < 0 || w.iy0 + static_cast<int32_t>(kPR) > static_cast<int32_t>(g.H) ||
               w.ix0 + 10 > static_cast<int32_t>(g.W);
    return w;
  };
  auto fetch_piece = [&](const Where& w, int u, Raw& r) __attribute__((always_inline)) {
    r.x[u] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, rel[u] + w.origin, 0, 0));
  };
  // the fetched patch: (border units: pixels outside the image become the zero point,) re-centred into LDS, per-pixel
  // channel sums (of a') beside it
  auto fix_up = [&](Raw& r, const Where& w) __attribute__((always_inline)) {
    const uint32_t patch_off = lds_off(patch);
    const uint32_t pix_off = lds_off(pix);
    if (w.border) {
#pragma unroll
      for (int u = 0; u < NP; u++) {
        const int32_t iy = w.iy0 + static_cast<int32_t>(pyx[u] >> 16);
        const int32_t ix = w.ix0 + static_cast<int32_t>(pyx[u] & 0xFFFFu);
        const bool inb = static_cast<uint32_t>(iy) < g.H && static_cast<uint32_t>(ix) < g.W;
        r.x[u].x = inb ? r.x[u].x : static_cast<int>(fill4);
        r.x[u].y = inb ? r.x[u].y : static_cast<int>(fill4);
        r.x[u].z = inb ? r.x[u].z : static_cast<int>(fill4);
        r.x[u].w = inb ? r.x[u].w : static_cast<int>(fill4);
      }
    }
#pragma unroll
    for (int u = 0; u < NP; u++) {
      const uint32_t v = lane + u * 64u;
      if ((u + 1) * 64u <= pvec || v < pvec) {             // (only the last piece is partly populated)
        const v4i x = r.x[u];
        if constexpr (CEN) {
          const uint32_t flip = p.a_flip;
          ds_write16_raw(patch_off + v * 16u, make_uint4(x.x ^ flip, x.y ^ flip, x.z ^ flip, x.w ^ flip));
        } else {
        uint32_t sum = __builtin_amdgcn_sad_u8(x.x, 0u, 0u);
        sum = __builtin_amdgcn_sad_u8(x.y, 0u, sum);
        sum = __builtin_amdgcn_sad_u8(x.z, 0u, sum);
        sum = __builtin_amdgcn_sad_u8(x.w, 0u, sum);
        sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0xB1, 0xF, 0xF, false));        // lane ^ 1
        if (cpp > 2) sum += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(sum), 0x4E, 0xF, 0xF, false));   // lane ^ 2
        ds_write16_raw(patch_off + v * 16u, make_uint4(x.x ^ kFlip, x.y ^ kFlip, x.z ^ kFlip, x.w ^ kFlip));
        ds_write4_raw(pix_off + v * 4u, static_cast<int32_t>(sum) - 128 * static_cast<int32_t>(cin));   // (all cpp lanes of the pixel)
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  // ---- prologue, second half: the first patch (an HBM round trip) behind the weights
  uint32_t cur = lo + wave;                        // round-robin walk: every unit costs the same, a counter buys no balance
  Raw raw;
  Where here = locate(min(cur, a.units - 1u));
#pragma unroll
  for (int u = 0; u < NP; u++) fetch_piece(here, u, raw);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>(a.units / tiles * g.OH * g.OW * p.n), 0x00020000);   // (launcher: < 2^31)
  __builtin_amdgcn_sched_barrier(0);
  WS_PRO(2);
  fix_up(raw, here);                                // needs the patch only
  WS_PRO(3);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WS_PRO(4);

  // ---- every weight fragment into registers, for good ----
  const uint32_t kblocks = p.k_pad / 32;
  v4i wreg[9][CB][TN];
  {
    const uint8_t* w_lane = w_lds + lane * 16;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int cb = 0; cb < CB; cb++)
#pragma unroll
        for (int tn = 0; tn < TN; tn++)
          wreg[t][cb][tn] = *reinterpret_cast<const v4i*>(w_lane + (tn * kblocks + t * CB + cb) * 1024u);
  }

#ifdef QNNP_ENABLE_ABLATION
  if (p.trace != nullptr) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (so that the stamp means "weights in registers")
#endif
  WS_PRO(5);
  const uint32_t i0 = lane & 31u;                  // position of this lane inside the unit
  const uint32_t ty = i0 >> 3;
  const uint32_t rowbase = (ty * 10u + (i0 & 7u)) << log_cin;
  struct AF { v4i a[CB]; };
  uint32_t unit_no = 0;
  (void) unit_no;
#define WS_STAMP(slot) do { if (wave == 0) { QNNP_TRACE(p, blockIdx.x, unit_no, slot); } } while (0)
  while (cur < hi) {
    WS_STAMP(0);
    const Where next = locate(min(cur + kWsWaves, a.units - 1u));

    // accumulators start at the folded bias
    v16i acc[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const v4i b = *reinterpret_cast<const v4i*>(bias_lds + tn * 32 + rg * 8 + khalf * 4);
        acc[tn][rg * 4 + 0] = b.x;
        acc[tn][rg * 4 + 1] = b.y;
        acc[tn][rg * 4 + 2] = b.z;
        acc[tn][rg * 4 + 3] = b.w;
      }
    {
      const uint8_t* abase[3][CB];
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const uint32_t swz = (ty + ky) & (cpp - 1u);
#pragma unroll
        for (int cb = 0; cb < CB; cb++) abase[ky][cb] = patch + rowbase + ky * 10 * cin + ((((cb << 1) | khalf) ^ swz) << 4);
      }
      auto read_a = [&](auto t_c, AF& f) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        constexpr int ky = t / 3, kx = t % 3;
#pragma unroll
        for (int cb = 0; cb < CB; cb++) f.a[cb] = *reinterpret_cast<const v4i*>(abase[ky][cb] + kx * cin);
      };
      auto mma = [&](auto t_c, const AF& f) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++)
            acc[tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wreg[t][cb][tn], f.a[cb], acc[tn], 0, 0, 0);
      };
      // the next unit's patch is requested piece by piece between the taps
      auto piece = [&](int u) __attribute__((always_inline)) { if (u < NP) fetch_piece(next, u, raw); };
#define QNNP_T(n) std::integral_constant<int, n>{}
      AF f0, f1;
      WS_STAMP(1);
      read_a(QNNP_T(0), f0);
      read_a(QNNP_T(1), f1); mma(QNNP_T(0), f0);
      read_a(QNNP_T(2), f0); mma(QNNP_T(1), f1); piece(0);
      read_a(QNNP_T(3), f1); mma(QNNP_T(2), f0); piece(1);
      read_a(QNNP_T(4), f0); mma(QNNP_T(3), f1); piece(2);
      read_a(QNNP_T(5), f1); mma(QNNP_T(4), f0); piece(3);
      read_a(QNNP_T(6), f0); mma(QNNP_T(5), f1); piece(4);
      read_a(QNNP_T(7), f1); mma(QNNP_T(6), f0); piece(5);
      read_a(QNNP_T(8), f0); mma(QNNP_T(7), f1); piece(6);
      mma(QNNP_T(8), f0);
#undef QNNP_T
    }
    WS_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);

    // ---- fused epilogue: row term, requantization into the (now free) patch buffer, 16-byte stores ----
    {
      int32_t s = 0;
      if constexpr (!CEN) {
        const int32_t* pq = pix + (ty * 10u + (i0 & 7u)) * cpp;
#pragma unroll
        for (int t = 0; t < 9; t++) s += pq[((t / 3) * 10 + (t % 3)) * cpp];
      }
      const int32_t rowterm = with_rq_offset<SEQ>(CEN ? 0 : p.row_coeff * s);       // (CEN: a constant of the launch)
      uint64_t row_addend = 0;                       // lane forms: the row term rides in the multiply-add's addend
      if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
      v4i outv[TN];
#pragma unroll
      for (int tn = 0; tn < TN; tn++) {
        uint32_t pk[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          if constexpr (rq_is_lane<SEQ>()) {
            pk[rg] = q31_requantize_pack4_lane<SEQ, FULL>(
                static_cast<uint32_t>(acc[tn][rg * 4 + 0]), static_cast<uint32_t>(acc[tn][rg * 4 + 1]),
                static_cast<uint32_t>(acc[tn][rg * 4 + 2]), static_cast<uint32_t>(acc[tn][rg * 4 + 3]), row_addend, p.lane, p.rq);
          } else {
            pk[rg] = q31_requantize_pack4<SEQ, FULL, false>(
                add_wrap(acc[tn][rg * 4 + 0], rowterm), add_wrap(acc[tn][rg * 4 + 1], rowterm),
                add_wrap(acc[tn][rg * 4 + 2], rowterm), add_wrap(acc[tn][rg * 4 + 3], rowterm), p.rq);
          }
        }
        const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
        // this lane now holds 16 consecutive channels of ITS position: tn * 32 + khalf * 16 .. + 15
        outv[tn] = v4i{static_cast<int>(s02[0]), static_cast<int>(s02[1]), static_cast<int>(s13[0]), static_cast<int>(s13[1])};
      }
      if (TN == 2 && p.stream_out != 0) {
        // WHOLE-LINE stores (round 5): stored as they are, the two pieces are halves of 64-byte pixels -- every store instruction
        // writes 32-byte runs, and the streaming hint on such pieces measured 4 % slower (profiles/r04/streaming_stores_*).
        // Four v_permlane16_swap move positions 16-31's first piece into the second register and positions 0-15's second piece
        // into the first: register S then holds ALL FOUR 16-byte chunks of positions 16 S .. 16 S + 15 (lane L: position
        // 16 S + (L & 15), chunk 2 * ((L >> 4) & 1) + (L >> 5)), i.e. two output rows of eight 64-byte pixels = 4 + 4 whole
        // 128-byte lines per instruction, written exactly once: streaming stores, no read-back, no LDS.
        int* a0 = reinterpret_cast<int*>(&outv[0]);
        int* a1 = reinterpret_cast<int*>(&outv[TN - 1]);
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const auto sw = __builtin_amdgcn_permlane16_swap(static_cast<uint32_t>(a0[d]), static_cast<uint32_t>(a1[d]), false, false);
          a0[d] = static_cast<int>(sw[0]);
          a1[d] = static_cast<int>(sw[1]);
        }
        const uint32_t lx = lane & 7u, lr = (lane >> 3) & 1u;
        const uint32_t chunk = ((lane >> 4) & 1u) * 2u + khalf;
#pragma unroll
        for (int half = 0; half < 2; half++) {
          const uint32_t oy = here.oy0 + lr + 2u * half;
          const uint32_t ox = here.ox0 + lx;
          const bool ok = oy < g.OH && ox < g.OW;
          __builtin_amdgcn_raw_buffer_store_b128(
              __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, outv[half == 0 ? 0 : TN - 1]), out_rsrc,
              ok ? here.out_img + (oy * g.OW + ox) * p.n + chunk * 16u : 0xFFFFFFF0u, 0, 2);
        }
      } else {
        // stored directly -- no staging image, no read-back: the round trip through LDS cost ~370 cycles per unit of pure
        // latency (two dependent ds_read -> store pairs), and with two waves per SIMD nothing hides it
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
          const uint32_t oy = here.oy0 + ty;
          const uint32_t ox = here.ox0 + (i0 & 7u);
          const bool ok = oy < g.OH && ox < g.OW;
          __builtin_amdgcn_raw_buffer_store_b128(
              __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, outv[tn]), out_rsrc,
              ok ? here.out_img + (oy * g.OW + ox) * p.n + tn * 32 + khalf * 16 : 0xFFFFFFF0u, 0, 0);
        }
      }
      WS_STAMP(3);
    }
    WS_STAMP(4);
    // ---- the next unit's patch (fetched between the taps) into the patch buffer ----
    fix_up(raw, next);
    WS_STAMP(5);
    unit_no++;
    here = next;
    cur += kWsWaves;
  }
  WS_PRO(6);
  QNNP_TRACE_WALL(p, blockIdx.x + 256u, 1, 1);
#undef WS_PRO
#undef WS_STAMP
}

/*
 * The weight-stationary kernel on v_mfma_i32_16x16x64_i8 (round 6; 64 input channels, the zero-point-centred image): same unit
 * (4 rows x 8 positions x all channels), same patch path (buffer loads between the taps, re-centred on their way into the wave's
 * patch buffer), same weights-in-registers idea -- with the matrix shape that costs the least energy per MAC on this chip
 * (tools/ubench_mfma2.hip: 4.08 against 3.45 PetaOP/s sustained on random operands; this kernel's loop holds the matrix pipe
 * ~73 % busy at a power-limited clock, profiles/r05/conv3x3_prologue_and_unit_stamps_r05trace.txt). One instruction covers a tap's
 * 64 channels: a unit is 9 taps x 2 position tiles x TN16 channel tiles of 16 x 16. Operand lane l = (position or channel l & 15,
 * 16-byte K chunk g = l >> 4); result lane l, register r: position l & 15, channel 4 g + r.
 *   - weight fragments from pack.h's 32 x 32 image with other lane addresses (q8gemm256x.hip): tile tn, tap t -> fragment
 *     (tn >> 1, 2 t + (g >> 1)), position (16 (tn & 1) + (l & 15) + 32 (g & 1)) * 16; 9 x TN16 x 4 registers for the wave's life;
 *   - activation fragment of (position tile tm, tap): lane -> patch pixel (2 tm + ((l & 15) >> 3) + ky, (l & 7) + kx), chunk
 *     g ^ f(patch row), f(py) = 2 (py & 1): the four 4-lane runs a ds_read_b128 lane group touches land on four different bank
 *     quads (the swizzle py & 3 of the 32 x 32 kernel would pair them two by two);
 *   - epilogue: requantize 4 -> 1 dword, one 4 x 4 lane transpose per position tile: lane (position, g) holds channels
 *     16 g .. 16 g + 15 of its position, and a store instruction writes two output rows of eight whole 64-byte pixels.
 */
template <int TN16, int SEQ, bool FULL>
__global__ __launch_bounds__(kWsWaves * 64, 2)
void q8_conv_wave_ws16_kernel(const IgemmParams p, const ConvGeom g, const WaveArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [weights][bias][counter][waves x (patch | unused pixel sums)]
  constexpr uint32_t kPR = 6u;                     // patch rows (10 columns) of a 4 x 8 unit
  constexpr uint32_t cin = 64u, cpp = 4u;
  constexpr uint32_t pvec = kPR * 10u * cpp;       // 16-byte chunks of the patch
  constexpr int NP = (pvec + 63u) / 64u;           // 1 KiB pieces of the patch
  uint8_t* w_lds = lds;
  int32_t* bias_lds = reinterpret_cast<int32_t*>(lds + a.w_bytes);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* patch = lds + a.head_bytes + wave * (a.patch_bytes + kWsPixBytes);

  // ---- prologue, first half: weights + bias by LDS-DMA, once per workgroup, in front of everything else ----
  {
    const uint32_t pieces = a.w_bytes >> 10;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(p.packed_w) + lane * 16u;
    for (uint32_t i = wave; i < pieces; i += kWsWaves) dma16(src + i * 1024u, w_lds + i * 1024u);
    if (wave == kWsWaves - 1 && lane < p.n / 4u) {
      dma16(reinterpret_cast<const uint8_t*>(rq_is_lane<SEQ>() ? p.bias2u : p.bias2) + lane * 16u, reinterpret_cast<uint8_t*>(bias_lds));
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x) * a.units / gridDim.x);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(blockIdx.x + 1) * a.units / gridDim.x);
  const uint32_t tiles = a.tiles_x * a.tiles_y;
  const uint32_t fill4 = p.izp_fill;               // the raw zero point in every byte
  const uint32_t fpos = lane & 15u;                // position inside a 16-position tile: row fpos >> 3, column fpos & 7
  const uint32_t fg = lane >> 4;                   // K chunk of an operand; channel quad of a result

  // ---- the gather pattern of a patch: per piece u this lane's chunk v = lane + 64 u is pixel (py, px) of the patch, source
  //      chunk (v & 3) ^ 2 (py & 1); only the patch origin moves ----
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(a.units / tiles * p.image_stride), 0x00020000);   // (launcher: < 2^31)
  uint32_t rel[NP], pyx[NP];
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const uint32_t v = min(lane + u * 64u, pvec - 1u);
    const uint32_t sl = v & (cpp - 1u);
    const uint32_t q = v >> 2;
    const uint32_t py = (q * 6554u) >> 16;                 // q / 10 for q < 100
    const uint32_t px = q - py * 10u;
    rel[u] = (py * g.W + px) * p.input_stride + ((sl ^ ((py & 1u) << 1)) << 4);
    pyx[u] = (py << 16) | px;
  }
  struct Raw { v4i x[NP]; };
  struct Where { uint32_t origin; int32_t iy0, ix0; uint32_t out_img, oy0, ox0; bool border; };   // (wave-uniform)
  auto locate = [&](uint32_t unit) __attribute__((always_inline)) -> Where {
    const uint32_t img = div_magic(unit, a.inv_tiles);
    const uint32_t rr = unit - img * tiles;
    const uint32_t tyi = div_magic(rr, a.inv_tiles_x);
    const uint32_t txi = rr - tyi * a.tiles_x;
    Where w;
    w.oy0 = tyi * 4u;
    w.ox0 = txi * 8u;
    w.iy0 = static_cast<int32_t>(w.oy0) - static_cast<int32_t>(g.pad_top);
    w.ix0 = static_cast<int32_t>(w.ox0) - static_cast<int32_t>(g.pad_left);
    w.origin = img * static_cast<uint32_t>(p.image_stride) +
        static_cast<uint32_t>(w.iy0 * static_cast<int32_t>(g.W) + w.ix0) * p.input_stride;
    w.out_img = img * g.OH * g.OW * p.n;
    w.border = w.iy0 < 0 || w.ix0 < 0 || w.iy0 + static_cast<int32_t>(kPR) > static_cast<int32_t>(g.H) ||
               w.ix0 + 10 > static_cast<int32_t>(g.W);
    return w;
  };
  auto fetch_piece = [&](const Where& w, int u, Raw& r) __attribute__((always_inline)) {
    r.x[u] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, rel[u] + w.origin, 0, 0));
  };
  // the fetched patch: (border units: pixels outside the image become the zero point,) re-centred into LDS
  auto fix_up = [&](Raw& r, const Where& w) __attribute__((always_inline)) {
    const uint32_t patch_off = lds_off(patch);
    if (w.border) {
#pragma unroll
      for (int u = 0; u < NP; u++) {
        const int32_t iy = w.iy0 + static_cast<int32_t>(pyx[u] >> 16);
        const int32_t ix = w.ix0 + static_cast<int32_t>(pyx[u] & 0xFFFFu);
        const bool inb = static_cast<uint32_t>(iy) < g.H && static_cast<uint32_t>(ix) < g.W;
        r.x[u].x = inb ? r.x[u].x : static_cast<int>(fill4);
        r.x[u].y = inb ? r.x[u].y : static_cast<int>(fill4);
        r.x[u].z = inb ? r.x[u].z : static_cast<int>(fill4);
        r.x[u].w = inb ? r.x[u].w : static_cast<int>(fill4);
      }
    }
    const uint32_t flip = p.a_flip;
#pragma unroll
    for (int u = 0; u < NP; u++) {
      const uint32_t v = lane + u * 64u;
      if ((u + 1) * 64u <= pvec || v < pvec) {             // (only the last piece is partly populated)
        const v4i x = r.x[u];
        ds_write16_raw(patch_off + v * 16u, make_uint4(x.x ^ flip, x.y ^ flip, x.z ^ flip, x.w ^ flip));
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  // ---- prologue, second half: the first patch (an HBM round trip) behind the weights
  uint32_t cur = lo + wave;
  Raw raw;
  Where here = locate(min(cur, a.units - 1u));
#pragma unroll
  for (int u = 0; u < NP; u++) fetch_piece(here, u, raw);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>(a.units / tiles * g.OH * g.OW * p.n), 0x00020000);   // (launcher: < 2^31)
  __builtin_amdgcn_sched_barrier(0);
  fix_up(raw, here);                                // needs the patch only
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- every weight fragment into registers, for good ----
  const uint32_t kblocks = p.k_pad / 32;
  v4i wreg[9][TN16];
  {
    const uint8_t* w_lane = w_lds + (fpos + 32u * (fg & 1u)) * 16u;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int tn = 0; tn < TN16; tn++)
        wreg[t][tn] = *reinterpret_cast<const v4i*>(w_lane + ((tn >> 1) * kblocks + 2 * t + (fg >> 1)) * 1024u + (tn & 1) * 256u);
  }

  // fragment bases: position tile tm, kernel row ky (the swizzle depends on the parity of the patch row)
  const uint32_t tyl = fpos >> 3;
  struct AF { v4i a[2]; };
  while (cur < hi) {
    const Where next = locate(min(cur + kWsWaves, a.units - 1u));

    // accumulators start at the folded bias: register r of tile tn = channel 16 tn + 4 g + r
    v4i acc[2][TN16];
#pragma unroll
    for (int tn = 0; tn < TN16; tn++) {
      const v4i b = *reinterpret_cast<const v4i*>(bias_lds + tn * 16 + fg * 4);
      acc[0][tn] = b;
      acc[1][tn] = b;
    }
    {
      const uint8_t* abase[3][2];
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
          const uint32_t py = tm * 2u + tyl + ky;
          abase[ky][tm] = patch + ((py * 10u + (fpos & 7u)) << 6) + ((fg ^ ((py & 1u) << 1)) << 4);
        }
      }
      auto read_a = [&](auto t_c, AF& f) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
        constexpr int ky = t / 3, kx = t % 3;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) f.a[tm] = *reinterpret_cast<const v4i*>(abase[ky][tm] + kx * cin);
      };
      auto mma = [&](auto t_c, const AF& f) __attribute__((always_inline)) {
        constexpr int t = decltype(t_c)::value;
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
          for (int j = 0; j < TN16; j++) {
            const int tn = tm == 0 ? j : TN16 - 1 - j;       // snake: one operand changes per MFMA (q8gemm256x.hip)
            acc[tm][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wreg[t][tn], f.a[tm], acc[tm][tn], 0, 0, 0);
          }
      };
      // the next unit's patch is requested piece by piece between the taps
      auto piece = [&](int u) __attribute__((always_inline)) { if (u < NP) fetch_piece(next, u, raw); };
#define QNNP_T(n) std::integral_constant<int, n>{}
      AF f0, f1;
      read_a(QNNP_T(0), f0);
      read_a(QNNP_T(1), f1); mma(QNNP_T(0), f0);
      read_a(QNNP_T(2), f0); mma(QNNP_T(1), f1); piece(0);
      read_a(QNNP_T(3), f1); mma(QNNP_T(2), f0); piece(1);
      read_a(QNNP_T(4), f0); mma(QNNP_T(3), f1); piece(2);
      read_a(QNNP_T(5), f1); mma(QNNP_T(4), f0); piece(3);
      read_a(QNNP_T(6), f0); mma(QNNP_T(5), f1); piece(4);
      read_a(QNNP_T(7), f1); mma(QNNP_T(6), f0); piece(5);
      read_a(QNNP_T(8), f0); mma(QNNP_T(7), f1); piece(6);
      mma(QNNP_T(8), f0);
#undef QNNP_T
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- fused epilogue: requantization, one lane transpose per position tile, 16-byte stores of whole pixels ----
    {
      const int32_t rowterm = with_rq_offset<SEQ>(0);
      uint64_t row_addend = 0;                       // lane forms: the (constant) row term rides in the multiply-add's addend
      if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
#pragma unroll
      for (int tm = 0; tm < 2; tm++) {
        uint32_t q[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int tn = 0; tn < TN16; tn++) {
          if constexpr (rq_is_lane<SEQ>()) {
            q[tn] = q31_requantize_pack4_lane<SEQ, FULL>(
                static_cast<uint32_t>(acc[tm][tn][0]), static_cast<uint32_t>(acc[tm][tn][1]),
                static_cast<uint32_t>(acc[tm][tn][2]), static_cast<uint32_t>(acc[tm][tn][3]), row_addend, p.lane, p.rq);
          } else {
            q[tn] = q31_requantize_pack4<SEQ, FULL, false>(
                add_wrap(acc[tm][tn][0], rowterm), add_wrap(acc[tm][tn][1], rowterm),
                add_wrap(acc[tm][tn][2], rowterm), add_wrap(acc[tm][tn][3], rowterm), p.rq);
          }
        }
        // 4 x 4 dword transpose over the four 16-lane rows: lane (position, g) then holds channels 16 g .. 16 g + 15
        const auto s02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
        const auto tlo = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const auto thi = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        const v4i outv = {static_cast<int>(tlo[0]), static_cast<int>(tlo[1]), static_cast<int>(thi[0]), static_cast<int>(thi[1])};
        const uint32_t oy = here.oy0 + tm * 2u + tyl;
        const uint32_t ox = here.ox0 + (fpos & 7u);
        const bool ok = oy < g.OH && ox < g.OW && fg < static_cast<uint32_t>(TN16);
        const uint32_t off = ok ? here.out_img + (oy * g.OW + ox) * p.n + fg * 16u : 0xFFFFFFF0u;
        const auto bits = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, outv);
        // (64 output channels: a store instruction writes two runs of eight whole 64-byte pixels, every line exactly once)
        if (TN16 == 4 && p.stream_out != 0) __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, off, 0, 0);
      }
    }
    // ---- the next unit's patch (fetched between the taps) into the patch buffer ----
    fix_up(raw, next);
    here = next;
    cur += kWsWaves;
  }
}

template <int TN16, int SEQ, bool FULL>
int launch_ws16_as(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, hipStream_t stream)
{
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_wave_ws16_kernel<TN16, SEQ, FULL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void) hipGetLastError();
    }
  }
  const uint32_t want = (a.units + kWsWaves - 1) / kWsWaves;
  const uint32_t grid = want < p.cu_count ? want : p.cu_count;
  hipLaunchKernelGGL((q8_conv_wave_ws16_kernel<TN16, SEQ, FULL>), dim3(grid), dim3(kWsWaves * 64), ws_lds_bytes(a), stream, p, g, a);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int TN16>
int launch_ws16(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    rc = launch_ws16_as<TN16, decltype(seq)::value, decltype(full)::value>(p, g, a, stream);
  });
  return rc;
}


template <int TN, int CB, int SEQ, bool FULL, bool CEN>
int launch_ws_as(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, hipStream_t stream)
{
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_wave_ws_kernel<TN, CB, SEQ, FULL, CEN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void) hipGetLastError();
    }
  }
  const uint32_t want = (a.units + kWsWaves - 1) / kWsWaves;
  const uint32_t grid = want < p.cu_count ? want : p.cu_count;
  hipLaunchKernelGGL((q8_conv_wave_ws_kernel<TN, CB, SEQ, FULL, CEN>), dim3(grid), dim3(kWsWaves * 64), ws_lds_bytes(a), stream, p, g, a);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int TN, int CB>
int launch_ws(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    if (p.a_flip != 0) rc = launch_ws_as<TN, CB, decltype(seq)::value, decltype(full)::value, true>(p, g, a, stream);
    else rc = launch_ws_as<TN, CB, decltype(seq)::value, decltype(full)::value, false>(p, g, a, stream);
  });
  return rc;
}


template <int TM, int TN, int CB, int SEQ, bool FULL>
int launch_reg_as(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, hipStream_t stream)
{
  constexpr int kRegWaves = reg_waves<TM>();
  constexpr int kRegThreads = kRegWaves * 64;
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_wave_reg_kernel<TM, TN, CB, SEQ, FULL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void) hipGetLastError();
    }
  }
  const uint32_t want = (a.units + kRegWaves - 1) / kRegWaves;
  const uint32_t grid = want < p.cu_count ? want : p.cu_count;
  hipLaunchKernelGGL((q8_conv_wave_reg_kernel<TM, TN, CB, SEQ, FULL>), dim3(grid), dim3(kRegThreads), reg_lds_bytes<TM>(a), stream, p, g, a);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int TM, int TN, int CB>
int launch_reg(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    rc = launch_reg_as<TM, TN, CB, decltype(seq)::value, decltype(full)::value>(p, g, a, stream);
  });
  return rc;
}

template <int TN, int CB, int KS, int SEQ, bool FULL>
int launch_as(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, uint32_t lds_bytes, hipStream_t stream)
{
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_wave_mfma_kernel<TN, CB, KS, SEQ, FULL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      (void) hipGetLastError();
    }
  }
  const uint32_t want = (a.units + kWaves - 1) / kWaves;
  const uint32_t grid = want < p.cu_count ? want : p.cu_count;
  hipLaunchKernelGGL((q8_conv_wave_mfma_kernel<TN, CB, KS, SEQ, FULL>), dim3(grid), dim3(kThreads), lds_bytes, stream, p, g, a);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int TN, int CB, int KS>
int launch(const IgemmParams& p, const ConvGeom& g, const WaveArgs& a, uint32_t lds_bytes, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    rc = launch_as<TN, CB, KS, decltype(seq)::value, decltype(full)::value>(p, g, a, lds_bytes, stream);
  });
  return rc;
}

}  // namespace

bool convwave_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch)
{
  if (groups != 1 || vec != 16 || p.fill_table == nullptr) return false;
  if (!(p.kc == 32 || p.kc == 64)) return false;
  if (!(p.n == 32 || p.n == 64) || p.n_pad != p.n || p.output_stride != p.n) return false;
  if (p.k_total != g.KH * g.KW * p.kc) return false;
  if (g.sh > 2 || g.sw > 2) return false;
  WaveArgs a;
  uint32_t lds_bytes = 0;
  return make_args(p, g, batch, &a, &lds_bytes);
}

/* `centred`: the same launch with the zero-point-centred weight image, its bias pair and a_flip (q8igemm.hip), or null:
 * the weight-stationary kernel takes it, the other flavours keep `p`. */
int convwave_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name, int flavour,
                    const IgemmParams* centred)
{
  WaveArgs a;
  uint32_t lds_bytes = 0;
  if (!make_args(p, g, batch, &a, &lds_bytes)) return QNNP_HIP_EINVAL;
  *name = "q8_conv_wave_mfma";
  const bool k33 = g.KH == 3 && g.KW == 3 && g.sh == 1 && g.sw == 1 && g.dh == 1 && g.dw == 1;
  // 3x3 / stride 1: the register-path kernel when its LDS fits and the output is addressable with 32-bit offsets
  const uint64_t out_bytes = static_cast<uint64_t>(batch) * g.OH * g.OW * p.n;
  if (k33 && out_bytes < (UINT64_C(1) << 31) && flavour != 1) {
    // weights in registers (q8_conv_wave_ws_kernel): the default; "gemm_kernel" = 12 keeps the round-2 register-path kernel
    bool ok = false;
    const WaveArgs ar = reg_args<1>(a, p, g, batch, &ok);
    const uint64_t in_bytes = static_cast<uint64_t>(batch) * p.image_stride;    // (32-bit buffer offsets)
    if (ok && in_bytes < (UINT64_C(1) << 31) && ws_lds_bytes(ar) <= kLdsLimit) {
      const IgemmParams& pw = centred != nullptr ? *centred : p;
      // round 6: 64 input channels with a centred image on the 16x16x64 shape; flavour 2 ("gemm_kernel" 27) keeps the 32x32x32 one
      if (centred != nullptr && p.kc == 64 && flavour != 2) {
        *name = "q8_conv_wave_ws_c16_mfma";
        return p.n == 32 ? launch_ws16<2>(pw, g, ar, stream) : launch_ws16<4>(pw, g, ar, stream);
      }
      *name = centred != nullptr ? "q8_conv_wave_ws_c_mfma" : "q8_conv_wave_ws_mfma";
      if (p.kc == 32) return p.n == 32 ? launch_ws<1, 1>(pw, g, ar, stream) : launch_ws<2, 1>(pw, g, ar, stream);
      return p.n == 32 ? launch_ws<1, 2>(pw, g, ar, stream) : launch_ws<2, 2>(pw, g, ar, stream);
    }
  }
  if (k33 && out_bytes < (UINT64_C(1) << 31)) {
    int tm = 1;                                    // 4x8-position units (see reg_waves)
#ifdef QNNP_ENABLE_ABLATION
    if (const char* env = getenv("QNNP_CONV_REG")) tm = atoi(env);       // 0: LDS-DMA kernel, 1 / 2: unit size
#endif
    bool ok = false;
    if (tm == 1) {
      const WaveArgs ar = reg_args<1>(a, p, g, batch, &ok);
      if (ok && reg_lds_bytes<1>(ar) <= kLdsLimit) {
        if (p.kc == 32) return p.n == 32 ? launch_reg<1, 1, 1>(p, g, ar, stream) : launch_reg<1, 2, 1>(p, g, ar, stream);
        return p.n == 32 ? launch_reg<1, 1, 2>(p, g, ar, stream) : launch_reg<1, 2, 2>(p, g, ar, stream);
      }
    }
#ifdef QNNP_ENABLE_ABLATION
    else if (tm == 2) {                              // (measurement builds only: the 8x8-position flavour)
      const WaveArgs ar = reg_args<2>(a, p, g, batch, &ok);
      if (ok && reg_lds_bytes<2>(ar) <= kLdsLimit) {
        if (p.kc == 32) return p.n == 32 ? launch_reg<2, 1, 1>(p, g, ar, stream) : launch_reg<2, 2, 1>(p, g, ar, stream);
        return p.n == 32 ? launch_reg<2, 1, 2>(p, g, ar, stream) : launch_reg<2, 2, 2>(p, g, ar, stream);
      }
    }
#endif
  }
  if (p.kc == 32) {
    if (k33) return p.n == 32 ? launch<1, 1, 3>(p, g, a, lds_bytes, stream) : launch<2, 1, 3>(p, g, a, lds_bytes, stream);
    return p.n == 32 ? launch<1, 1, 0>(p, g, a, lds_bytes, stream) : launch<2, 1, 0>(p, g, a, lds_bytes, stream);
  }
  if (k33) return p.n == 32 ? launch<1, 2, 3>(p, g, a, lds_bytes, stream) : launch<2, 2, 3>(p, g, a, lds_bytes, stream);
  return p.n == 32 ? launch<1, 2, 0>(p, g, a, lds_bytes, stream) : launch<2, 2, 0>(p, g, a, lds_bytes, stream);
}

}  // namespace qnnp
