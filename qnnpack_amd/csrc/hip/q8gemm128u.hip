/*
 * q8gemm128u.hip -- the uint8 GEMM on v_mfma_i32_16x16x64_i8 for the shapes NOTHING aligned takes (round 6): 1x1 convolutions and
 * fully connected layers with ANY channel counts, any number of groups, any kernel zero point, any pixel stride and alignment.
 *
 * Role. What reached the generic tile kernel (q8_igemm_mfma_128x{32,64,128}, q8igemm.hip) for three rounds at 0.04-0.13 of its bounds:
 * ShuffleNet's grouped 1x1 convolutions (bench/convolution.cc:108-301: 2 / 3 / 4 / 8 groups of 12 ... 400 channels), ShuffleNet v2's
 * pointwise layers with 58 / 116 / 122 / 232 / 244 / 488 channels (:303-426) -- pixels of 50 or 58 bytes, rows that start at any byte.
 * The reference runs them through the same q8gemm microkernel as every GEMM (src/q8gemm/4x4c2-sse2.c:14-318, whose loads are byte
 * granular anyway), one group after the other (src/operator-run.c:770-804). The generic kernel gathered every 16-byte operand piece
 * through the offset table with 64-bit address arithmetic, byte loads where rows were unaligned, and byte stores.
 *
 * Here the matrix side is q8gemm128x.hip's (128 x 128 tile, four waves of 64 x 64, 16x16x64 MFMAs, two-phase fragment schedule,
 * snake order), the weights come from pack.h's STANDARD image by LDS-DMA (always aligned: the library packed them), and only the
 * activation tile is staged by the threads themselves:
 *   - thread (row = tid >> 1, half = tid & 1) owns 32 bytes of its row per 64-byte K tile: nine dwords from the 4-byte-aligned
 *     address below the piece (one buffer descriptor over the tensor, 32-bit offsets; what lies past the tensor reads 0), eight
 *     v_alignbyte to shift them into place, bytes at k >= K replaced by 0x80 (last tile only), eight v_sad_u8 for the row sum of the
 *     kernel-zero-point term, eight v_xor to re-centre, two ds_write_b128 into the swizzled image q8gemm128x.hip reads;
 *   - tile kt + 1 is fetched into registers before the MFMAs of tile kt and written behind them: one barrier per K tile, two
 *     16-KiB stages (weights: LDS-DMA one tile ahead), 33 KiB of LDS -- four workgroups per CU;
 *   - epilogue: + (128 - kzp) * row sum, requantize four accumulators -> one dword; dword stores when the group's channel run is
 *     dword-aligned in every row, byte stores otherwise (valid channels only).
 * K need not be a multiple of anything (the image pads K to 32 with zero weights, the staged bytes past K are a' = 0); N neither.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <type_traits>

#include "igemm_params.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;
constexpr int kBK = 64;
constexpr int kATile = kBM * kBK;              // 8 KiB
constexpr int kThreads = 256;
constexpr int kTM = 4;

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ const uint8_t* scalar_ptr(const uint8_t* ptr)
{
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}

__device__ __forceinline__ uint32_t lds_address(uint8_t* lds_ptr)
{
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*) lds_ptr));
}

__device__ __forceinline__ void dma16_saddr(const uint8_t* base, uint32_t lane_offset, uint8_t* lds_wave_base)
{
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(lane_offset), "s"(base), "s"(lds_address(lds_wave_base)));
}

__device__ __forceinline__ uint32_t a_swizzle(uint32_t row) { return (row & 8u) != 0 ? 3u : 0u; }

#define QNNP_PIN() __builtin_amdgcn_sched_barrier(0)

/* OUT: 2 = every row's channel run of the group starts dword-aligned and N % 4 == 0; 1 = the same for 2 bytes (58 / 122 channels); 0 = no
 * alignment at all. 2 and 1 transpose the requantized dwords over the four lanes of a row (v_permlane32_swap / v_permlane16_swap) so that
 * a lane holds SIXTEEN consecutive channels of its row and writes them with one 16-byte store (2; rows of 2 mod 4: a short, three dwords
 * shifted by v_alignbyte, a short) -- 16 TN contiguous bytes per row and instruction instead of 16-byte pieces (one dword per lane and
 * tile, the first build) or single bytes; 0 keeps the byte stores. 3 = FLAT: dense pixels, one group, one channel tile -- the workgroup's 128
 * rows are one contiguous run of memory: staged in LDS as memory holds them, written as whole 16-byte chunks (any N).
 * TN: 16-channel MFMA tiles per wave: 4 / 2 / 1 = 128- / 64- / 32-channel workgroup tiles (groups of 12 ... 62 channels would leave
 * most of a wide tile empty: ShuffleNet v1's 8 groups of 48 -> 12). */
template <int SEQ, int CLAMP, int OUT, int TN>
__global__ __launch_bounds__(kThreads, 2)
void q8_gemm_mfma_128xN_u16_kernel(const IgemmParams p)
{
  constexpr int kTN = TN;
  constexpr int kBN = 32 * TN;
  constexpr int kWTile = kBN * kBK;
  constexpr int kStage = kATile + kWTile;
  constexpr int kFrags = (kBN / 32) * 2;                                              // 1-KiB weight fragments per K tile
  constexpr int kWPieces = (kFrags + 3) / 4;                                        // LDS-DMA instructions per wave (some waves idle)
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * kStage + 4 * 256 + kBM * 8];   // two stages, bias lines, row sums

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave >> 1, wn = wave & 1u;
  const uint32_t g = blockIdx.y;

  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  uint32_t m_tile, n_tile;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    const uint32_t logical = xcd * q + min(xcd, r) + idx;
    m_tile = p.tiles_n_magic != 0 ? __umulhi(logical, p.tiles_n_magic) : logical;
    n_tile = logical - m_tile * tiles_n;
  }
  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t K = p.k_total;
  const uint32_t ktiles = (K + kBK - 1) / kBK;
  const uint32_t nb0 = n_tile * (kBN / 32);

  // ---- weights: LDS-DMA, piece i of wave w = fragment (channel block nb0 + 2 i + (w >> 1), K block 2 kt + (w & 1)); blocks past the
  //      image's last channel block / K block re-read a valid one (unstored channels; K positions whose activations are staged as 0) ----
  const uint8_t* w_group = reinterpret_cast<const uint8_t*>(p.packed_w) + static_cast<uint64_t>(g) * nblocks * kblocks * 1024;
  uint32_t w_nb[kWPieces];
#pragma unroll
  for (int i = 0; i < kWPieces; i++) w_nb[i] = min(nb0 + 2u * i + (wave >> 1), nblocks - 1u);
  auto stage_w = [&](uint32_t kt, uint32_t slot) __attribute__((always_inline)) {
    const uint32_t kb = min(2u * kt + (wave & 1u), kblocks - 1u);
#pragma unroll
    for (int i = 0; i < kWPieces; i++) {
      if (i * 4 + static_cast<int>(wave) < kFrags) {                                 // (wave-uniform: 32-channel tiles have two fragments)
        dma16_saddr(scalar_ptr(w_group + (static_cast<uint64_t>(w_nb[i]) * kblocks + kb) * 1024), lane * 16,
                    lds + slot * kStage + kATile + (i * 4 + wave) * 1024);
      }
    }
  };

  // ---- activations: this thread's 32 bytes of row (tid >> 1) per K tile, through registers ----
  const uint32_t lrow = tid >> 1, lhalf = tid & 1u;
  uint32_t m_l = m_tile * kBM + lrow;
  if (m_l >= p.rows) m_l = p.rows - 1;                       // clamp: results of those rows are never stored
  // (the extent is rounded up to whole dwords: the dword that holds the tensor's last bytes must not read as out of range -- up to three
  //  bytes past the tensor are fetched with it and masked or multiplied by zero weights)
  const uint64_t in_bytes = (static_cast<uint64_t>(p.input_end - p.input) + 3u) & ~static_cast<uint64_t>(3);
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(in_bytes), 0x00020000);       // (launcher: < 2^31 bytes)
  const uint32_t a_byte0 = m_l * p.input_stride + g * p.kc + lhalf * 32u;            // byte offset of the piece at kt = 0
  const uint32_t a_wr = lrow * kBK + (((lhalf * 2u) ^ a_swizzle(lrow)) << 4);          // chunk 2 half; chunk 2 half + 1 sits at a_wr ^ 16
  uint32_t rs = 0;                                                                     // sum of this piece's raw bytes over all K tiles
  struct Raw { uint32_t d[9]; };
  auto fetch = [&](uint32_t kt) __attribute__((always_inline)) -> Raw {
    const uint32_t off = a_byte0 + kt * kBK;
    const uint32_t al = off & ~3u;
    Raw r;
    const auto q0 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, al, 0, 0);
    const auto q1 = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, al + 16u, 0, 0);
    r.d[0] = q0[0]; r.d[1] = q0[1]; r.d[2] = q0[2]; r.d[3] = q0[3];
    r.d[4] = q1[0]; r.d[5] = q1[1]; r.d[6] = q1[2]; r.d[7] = q1[3];
    r.d[8] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, al + 32u, 0, 0);
    return r;
  };
  const uint32_t a_shift = a_byte0 & 3u;                                               // (kt * 64 keeps it)
  auto settle = [&](const Raw& r, uint32_t kt, uint32_t slot) __attribute__((always_inline)) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = __builtin_amdgcn_alignbyte(r.d[i + 1], r.d[i], a_shift);
    const uint32_t k0 = kt * kBK + lhalf * 32u;
    if (kt * kBK + kBK > K) {                                                          // the last tile of a K that is no multiple of 64
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int32_t valid = static_cast<int32_t>(K) - static_cast<int32_t>(k0 + 4u * i);     // real bytes of this dword
        const uint32_t mask = valid >= 4 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : (1u << (8 * valid)) - 1u);
        e[i] = (e[i] & mask) | (0x80808080u & ~mask);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) rs = __builtin_amdgcn_sad_u8(e[i], 0u, rs);
    uint8_t* dst = lds + slot * kStage + a_wr;
    *reinterpret_cast<v4i*>(dst) = v4i{static_cast<int>(e[0] ^ 0x80808080u), static_cast<int>(e[1] ^ 0x80808080u),
                                      static_cast<int>(e[2] ^ 0x80808080u), static_cast<int>(e[3] ^ 0x80808080u)};
    *reinterpret_cast<v4i*>(lds + slot * kStage + (a_wr ^ 16u)) =
        v4i{static_cast<int>(e[4] ^ 0x80808080u), static_cast<int>(e[5] ^ 0x80808080u),
            static_cast<int>(e[6] ^ 0x80808080u), static_cast<int>(e[7] ^ 0x80808080u)};
  };

  const uint32_t frow = lane & 15u, fg = lane >> 4;
  const uint32_t n0 = n_tile * kBN + wn * (16u * kTN);

  // ---- prologue: weights of tile 0 + the wave's 64 biases by LDS-DMA, activations of tile 0 through registers ----
  stage_w(0u, 0u);
  uint8_t* bias_line = lds + 2 * kStage + wave * 256;
  if (lane < 4 * kTN) {
    const int32_t* bias_tab = (SEQ == kRqGeneral ? p.bias2 : p.bias2u) + static_cast<uint64_t>(g) * p.n_pad;
    dma16_saddr(scalar_ptr(reinterpret_cast<const uint8_t*>(bias_tab)), min(n0 + lane * 4u, p.n_pad - 4u) * 4u, bias_line);
  }
  {
    const Raw r0 = fetch(0u);
    settle(r0, 0u, 0u);
  }

  const uint32_t a_off = (wm * 64 + frow) * kBK + ((fg ^ a_swizzle(frow)) << 4);                       // + tm * 1024
  // weight fragment of the wave's tile tn: 16-channel tile t = wn * TN + tn of the workgroup tile -> block t >> 1, half t & 1
  const uint32_t w_off = kATile + (fg >> 1) * 1024 + (frow + 32 * (fg & 1u)) * 16;                   // + (t >> 1) * 2048 + (t & 1) * 256
  v4i acc[kTM][kTN];
  bool acc_init = false;
  (void) acc_init;

  for (uint32_t kt = 0; kt < ktiles; kt++) {
    const uint32_t slot = kt & 1u, nxt = slot ^ 1u;
    // tile kt is complete: its weights were requested a tile ago (the only vector-memory operations in flight here), its
    // activations written behind the previous tile's MFMAs
    wait_vmcnt<0>();
    __syncthreads();
    QNNP_PIN();
    if (kt == 0) {
      // accumulators start from the bias: register r of tile tn = channel n0 + 16 tn + 4 (lane >> 4) + r
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) {
        const v4i b = *reinterpret_cast<const v4i*>(bias_line + tn * 64 + fg * 16);
#pragma unroll
        for (int tm = 0; tm < kTM; tm++) acc[tm][tn] = b;
      }
    }
    const bool more = kt + 1u < ktiles;
    Raw rn;
    if (more) {
      stage_w(kt + 1u, nxt);
      rn = fetch(kt + 1u);
    }
    QNNP_PIN();
    const uint8_t* st = lds + slot * kStage;
    v4i fa[kTM], fw[kTN];
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) fa[tm] = *reinterpret_cast<const v4i*>(st + a_off + tm * 1024);
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      const uint32_t t16 = wn * kTN + tn;
      fw[tn] = *reinterpret_cast<const v4i*>(st + w_off + (t16 >> 1) * 2048 + (t16 & 1u) * 256);
    }
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
#pragma unroll
      for (int j = 0; j < kTN; j++) {
        const int tn = (tm & 1) != 0 ? kTN - 1 - j : j;              // snake: one operand changes per MFMA
        acc[tm][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fw[tn], fa[tm], acc[tm][tn], 0, 0, 0);
      }
    }
    QNNP_PIN();
    if (more) settle(rn, kt + 1u, nxt);
  }

  // ---- row sums: the two pieces of a row, through LDS ----
  int32_t* rowsum = reinterpret_cast<int32_t*>(lds + 2 * kStage + 4 * 256);
  rowsum[lrow * 2 + lhalf] = static_cast<int32_t>(rs);
  __syncthreads();

  // ---- epilogue ----
  const uint32_t m0 = m_tile * kBM + wm * 64;
  const uint32_t out_group = g * p.n;
  const int32_t row_coeff = p.row_coeff;
  const int32_t staged = static_cast<int32_t>(128u * ktiles * kBK);          // sum over the staged bytes of 128
  auto transpose4 = [&](uint32_t (&q)[4]) __attribute__((always_inline)) {
    const auto s02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    const auto lo = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    q[0] = lo[0]; q[1] = lo[1]; q[2] = hi[0]; q[3] = hi[1];
  };
  typedef int v4i_a4 __attribute__((ext_vector_type(4), aligned(4)));
  typedef int v3i_a4 __attribute__((ext_vector_type(3), aligned(4)));
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
    const uint32_t r = tm * 16 + frow;
    const int2 s2 = *reinterpret_cast<const int2*>(rowsum + (wm * 64 + r) * 2);
    const uint32_t term = static_cast<uint32_t>(row_coeff) * static_cast<uint32_t>(s2.x + s2.y - staged);
    const bool row_ok = m0 + r < p.rows;
    uint8_t* out_row = p.output + static_cast<uint64_t>(m0 + r) * p.output_stride + out_group;
    uint32_t q[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      q[tn] = q31_requantize_pack4_clamp<SEQ, CLAMP>(
          static_cast<int>(static_cast<uint32_t>(acc[tm][tn][0]) + term), static_cast<int>(static_cast<uint32_t>(acc[tm][tn][1]) + term),
          static_cast<int>(static_cast<uint32_t>(acc[tm][tn][2]) + term), static_cast<int>(static_cast<uint32_t>(acc[tm][tn][3]) + term), p.rq);
    }
    if constexpr (OUT == 3) {
      // FLAT: dense pixels (stride == N) and one channel tile -- the workgroup's 128 output rows are ONE contiguous run of 128 N bytes
      // that starts on a 128-byte boundary. The requantized bytes go to LDS where memory will hold them (the stage buffers are free:
      // every wave is past the barrier behind the K loop) and leave as whole 16-byte chunks below, whatever N is a multiple of.
      const uint32_t rflat = (wm * 64u + r) * p.n;
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) {
        const uint32_t n = n0 + tn * 16 + fg * 4;
#pragma unroll
        for (int b = 0; b < 4; b++) {
          if (n + b < p.n) lds[rflat + n + b] = static_cast<uint8_t>(q[tn] >> (8 * b));
        }
      }
    } else if constexpr (OUT == 0 || (OUT == 2 && kTN == 1)) {
      if (!row_ok) continue;
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) {
        const uint32_t n = n0 + tn * 16 + fg * 4;
        if constexpr (OUT == 2) {
          if (n < p.n) *reinterpret_cast<uint32_t*>(out_row + n) = q[tn];
        } else {
#pragma unroll
          for (int b = 0; b < 4; b++) {
            if (n + b < p.n) out_row[n + b] = static_cast<uint8_t>(q[tn] >> (8 * b));
          }
        }
      }
    } else {
      transpose4(q);               // lane (row, g): channels 16 g .. 16 g + 15 of its row within the wave's 16 TN (g < TN)
      const uint32_t n = n0 + fg * 16;
      if (!row_ok || fg >= static_cast<uint32_t>(kTN) || n >= p.n) continue;
      uint8_t* dst = out_row + n;
      if (n + 16 <= p.n) {
        if (OUT == 2 || (reinterpret_cast<uintptr_t>(dst) & 2u) == 0) {
          *reinterpret_cast<v4i_a4*>(dst) = v4i_a4{static_cast<int>(q[0]), static_cast<int>(q[1]), static_cast<int>(q[2]), static_cast<int>(q[3])};
        } else {
          *reinterpret_cast<uint16_t*>(dst) = static_cast<uint16_t>(q[0]);
          *reinterpret_cast<v3i_a4*>(dst + 2) = v3i_a4{static_cast<int>(__builtin_amdgcn_alignbyte(q[1], q[0], 2)),
                                                       static_cast<int>(__builtin_amdgcn_alignbyte(q[2], q[1], 2)),
                                                       static_cast<int>(__builtin_amdgcn_alignbyte(q[3], q[2], 2))};
          *reinterpret_cast<uint16_t*>(dst + 14) = static_cast<uint16_t>(q[3] >> 16);
        }
      } else {                     // the group's last channels: what is left of the piece, dword by dword (2) or byte by byte
#pragma unroll
        for (int d = 0; d < 4; d++) {
          if constexpr (OUT == 2) {
            if (n + 4 * d < p.n) *reinterpret_cast<uint32_t*>(dst + 4 * d) = q[d];
          } else {
#pragma unroll
            for (int b = 0; b < 4; b++) {
              if (n + 4 * d + b < p.n) dst[4 * d + b] = static_cast<uint8_t>(q[d] >> (8 * b));
            }
          }
        }
      }
    }
  }
  if constexpr (OUT == 3) {
    __syncthreads();
    const uint32_t row0 = m_tile * kBM;
    const uint32_t rows_here = min(static_cast<uint32_t>(kBM), p.rows - row0);
    const uint32_t total = rows_here * p.n;
    uint8_t* dst = p.output + static_cast<uint64_t>(row0) * p.n;                     // (launcher: a multiple of 16 bytes from a 16-byte aligned base)
    for (uint32_t c = tid * 16u; c + 16u <= total; c += kThreads * 16u) {
      *reinterpret_cast<v4i*>(dst + c) = *reinterpret_cast<const v4i*>(lds + c);
    }
    const uint32_t tail = total & 15u;                                                // (the last tile of a tensor whose size is no multiple of 16)
    if (tid < tail) dst[(total & ~15u) + tid] = lds[(total & ~15u) + tid];
  }
}
#undef QNNP_PIN

}  // namespace

/* any 1x1 / fully connected problem on the standard image with a bias pair table; tensors below 2^31 bytes */
bool gemm128u_supported(const IgemmParams& p)
{
  if (p.offsets != nullptr || p.ks != 1 || p.residual != nullptr || p.d2s_sh != 0) return false;
  if (p.bias2u == nullptr || p.packed_w == nullptr || p.rows < 1 || p.k_total < 1 || p.n < 1) return false;
  if (p.k_pad % 32 != 0 || p.n_pad % 32 != 0 || p.k_pad < p.k_total || p.kc != p.k_total) return false;
  const uint64_t in_bytes = static_cast<uint64_t>(p.input_end - p.input);
  if (in_bytes >= (1ull << 31) || static_cast<uint64_t>(p.rows) * p.input_stride + 64 >= (1ull << 32)) return false;
  return true;
}

namespace {
template <int TN>
int launch_u(const IgemmParams& p, uint32_t groups, hipStream_t stream)
{
  constexpr uint32_t kBN = 32 * TN;
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  if (static_cast<uint64_t>(tiles_m) * tiles_n * tiles_n >= (1ull << 32)) return QNNP_HIP_EINVAL;
  const dim3 grid(tiles_m * tiles_n, groups, 1);
  IgemmParams pm = p;
  pm.tiles_n_magic = tiles_n == 1 ? 0u : static_cast<uint32_t>((1ull << 32) / tiles_n) + 1u;
  const bool dword_out = p.n % 4 == 0 && p.output_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(p.output) & 3u) == 0;
  const bool even_out = p.n % 2 == 0 && p.output_stride % 2 == 0 && (reinterpret_cast<uintptr_t>(p.output) & 1u) == 0;
  // (flat rows: dense pixels, one group, one channel tile, rows that are not whole 16-byte pieces anyway)
  const bool flat_out = groups == 1 && tiles_n == 1 && p.output_stride == p.n && p.n % 16 != 0 &&
      (reinterpret_cast<uintptr_t>(p.output) & 15u) == 0 && static_cast<uint64_t>(p.rows) * p.n < (1ull << 32);
  int rc = QNNP_HIP_EINVAL;
  auto launch = [&](auto seq, auto clamp) {
    constexpr int kSeq = decltype(seq)::value;
    constexpr int kClamp = decltype(clamp)::value;
    if (flat_out) hipLaunchKernelGGL((q8_gemm_mfma_128xN_u16_kernel<kSeq, kClamp, 3, TN>), grid, dim3(kThreads), 0, stream, pm);
    else if (dword_out) hipLaunchKernelGGL((q8_gemm_mfma_128xN_u16_kernel<kSeq, kClamp, 2, TN>), grid, dim3(kThreads), 0, stream, pm);
    else if (even_out) hipLaunchKernelGGL((q8_gemm_mfma_128xN_u16_kernel<kSeq, kClamp, 1, TN>), grid, dim3(kThreads), 0, stream, pm);
    else hipLaunchKernelGGL((q8_gemm_mfma_128xN_u16_kernel<kSeq, kClamp, 0, TN>), grid, dim3(kThreads), 0, stream, pm);
    rc = hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  using C2 = std::integral_constant<int, 2>;
  if (p.rq.f.shift != 0 && p.rq.f.bounded && p.rq.f.ofs_kind == 2 && !p.rq.full_range) {
    if (p.rq.zp_late == 0) launch(std::integral_constant<int, kRqBoundedOfs>{}, C1{});
    else launch(std::integral_constant<int, kRqBoundedOfs>{}, C2{});
    return rc;
  }
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    if constexpr (decltype(full)::value) launch(seq, C0{});
    else if (p.rq.zp_late == 0) launch(seq, C1{});
    else launch(seq, C2{});
  });
  return rc;
}
}  // namespace

/* tile_n: 0 = by the shape, else forced. Every channel tile stages the activation tile again (cost ~ K) and pays an epilogue
 * (cost ~ its width): with a short reduction (K <= 64) the width with the fewest padded columns wins (14 x 14, 3 groups of 40 -> 160:
 * 12.6 us on 32-wide tiles against 14.3 on 128-wide), beyond that the FEWEST tiles, the narrowest width among those
 * (160 -> 80: 10.7 against 12.5; 25 -> 88: 13.5 against 14.5; profiles/r06/ugemm_by_forced_kernel_r06p.txt). */
int gemm128u_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, uint32_t tile_n)
{
  uint32_t bn = tile_n;
  if (bn == 0) {
    const uint32_t t128 = (p.n + 127u) / 128u, t64 = (p.n + 63u) / 64u, t32 = (p.n + 31u) / 32u;
    if (p.k_total <= 64u) {
      const uint32_t c128 = t128 * 128u, c64 = t64 * 64u, c32 = t32 * 32u;
      bn = c32 < c64 ? (c32 < c128 ? 32u : 128u) : (c64 < c128 ? 64u : 128u);
    } else {
      bn = t32 == t128 ? 32u : (t64 == t128 ? 64u : 128u);
    }
  }
  if (bn == 32u) { *name = "q8_gemm_mfma_128x32_u16"; return launch_u<1>(p, groups, stream); }
  if (bn == 64u) { *name = "q8_gemm_mfma_128x64_u16"; return launch_u<2>(p, groups, stream); }
  *name = "q8_gemm_mfma_128x128_u16";
  return launch_u<4>(p, groups, stream);
}

}  // namespace qnnp
