/*
 * pack.h -- create-time weight/bias packing into the gfx950 device layouts.
 *
 * Role-equivalent to the reference's src/qnnpack/pack.h, whose packers interleave
 * weights for 4x4c2 / 8x8 CPU register tiles and fold the input zero point into
 * the bias (pack_q8gemm_w :12-49, pack_q8conv_w :51-91, pack_q8dw_w :135-167).
 * Here the targets are (a) MFMA operand fragments for
 * v_mfma_i32_32x32x32_i8 and (b) a tap-major int16 image for the depthwise
 * kernels. All arithmetic is exact int32 (wraps mod 2^32 like the reference).
 */
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t qnnp_round_up_u32(uint32_t x, uint32_t q) {
  return (x + q - 1) / q * q;
}

/*
 * igemm weight image.
 *
 * Source (one group): kernel[n][kk], kk = tap * kc + channel in [0, k_total) --
 * exactly the reference's [oc][ky][kx][ic] tensor flattened
 * (test/convolution-operator-tester.h:393), or [N][K] for fully connected.
 *
 * Destination: for group g, 32-column block nb, 32-deep K block kb, one
 * 1024-byte MFMA fragment: lane l (0..63) owns 16 consecutive bytes
 *     w'(n = nb*32 + (l & 31), kk = kb*32 + (l >> 5)*16 + j),  j = 0..15
 * at byte offset ((((g*NB + nb)*KB + kb)*64 + l)*16 + j), NB = n_pad/32,
 * KB = k_pad/32. w' = w - 128 (uint8 -> int8 by flipping the top bit);
 * positions with n >= N or kk >= k_total hold 0 so they add nothing.
 * A wave loads its whole fragment with one coalesced 16-B-per-lane read.
 *
 * bias2[g*n_pad + n] = bias + (128 - izp) * sum_kk w'(n,kk)
 *                           + k_total * (128 - izp) * (128 - kzp)
 * so that, with a' = a - 128,
 *   bias + sum (a - izp)(w - kzp) = bias2 + (128 - kzp) * sum a' + sum a' w'.
 * This is the same zero-point algebra the reference uses for its ARM "XZP"
 * GEMM (src/operator-run.c:727-743, pack.h:204-256), re-centred at 128 because
 * the MFMA multiplies signed int8.
 */
static inline size_t qnnp_igemm_packed_weights_size(uint32_t groups, uint32_t n_pad, uint32_t k_pad) {
  return (size_t) groups * n_pad * k_pad;
}

/*
 * Channel slots: a tap normally occupies `kc` consecutive K positions. For 3-channel inputs (first
 * layers) the device kernel fetches each tap with ONE unaligned 4-byte load, so a tap occupies a
 * 4-wide slot whose last position is K padding (w' = 0 here, a' forced to 0 in the kernel):
 * packed K index = tap * kc_slot + channel, k_total (real) = ks * kc.
 */
static inline void qnnp_pack_igemm_w_slots(
    uint32_t groups, uint32_t n, uint32_t ks, uint32_t kc, uint32_t kc_slot,
    uint32_t n_pad, uint32_t k_pad,
    uint8_t izp, uint8_t kzp,
    const uint8_t* kernel, const int32_t* bias,
    int8_t* packed, int32_t* bias2)
{
  const uint32_t k_total = ks * kc;
  const uint32_t nblocks = n_pad / 32;
  const uint32_t kblocks = k_pad / 32;
  memset(packed, 0, qnnp_igemm_packed_weights_size(groups, n_pad, k_pad));
  const uint32_t a_off = (uint32_t) (128 - (int32_t) izp);
  const uint32_t w_off = (uint32_t) (128 - (int32_t) kzp);
  for (uint32_t g = 0; g < groups; g++) {
    for (uint32_t col = 0; col < n_pad; col++) {
      uint32_t b2 = 0;
      if (col < n) {
        const uint8_t* src = kernel + ((size_t) g * n + col) * k_total;
        const uint32_t nb = col / 32;
        const uint32_t lane_lo = col % 32;
        uint32_t wsum = 0;
        for (uint32_t kk = 0; kk < k_total; kk++) {
          const int32_t ws = (int32_t) src[kk] - 128;
          wsum += (uint32_t) ws;
          const uint32_t kp = (kk / kc) * kc_slot + (kk % kc);   /* packed K position */
          const uint32_t kb = kp / 32;
          const uint32_t lane = lane_lo + 32 * ((kp % 32) / 16);
          const size_t dst = ((((size_t) g * nblocks + nb) * kblocks + kb) * 64 + lane) * 16 + (kp % 16);
          packed[dst] = (int8_t) ws;
        }
        b2 = (uint32_t) bias[(size_t) g * n + col] + a_off * wsum + k_total * a_off * w_off;
      }
      bias2[(size_t) g * n_pad + col] = (int32_t) b2;
    }
  }
}

/*
 * Row-slot image for 3-channel first layers (hip/q8convc3.hip): packed K index = ky * 16 + kx * 3 + c -- one 16-byte
 * slot per kernel ROW (the kx * 3 + c bytes of a row are contiguous in a dense 3-byte-pixel image, so the device kernel
 * fetches a slot with one 16-byte load); the unused bytes of a slot and the slots beyond the kernel are zero weights.
 * k_pad = 64 (kernel_height <= 4, kernel_width * 3 <= 16). Same fragment geometry as the images above; the folded bias
 * is the one qnnp_pack_igemm_w_slots makes (it depends on the real taps only).
 * Source kernel layout: [oc][ky][kx][ic] (single group).
 */
static inline void qnnp_pack_conv_rows16(
    uint32_t n, uint32_t kh, uint32_t kw, uint32_t kc, uint32_t n_pad,
    const uint8_t* kernel, int8_t* packed /* [n_pad / 32][2][64][16] */)
{
  memset(packed, 0, (size_t) n_pad * 64);
  for (uint32_t col = 0; col < n; col++) {
    const uint32_t nb = col / 32;
    const uint32_t lane_lo = col % 32;
    for (uint32_t ky = 0; ky < kh; ky++) {
      for (uint32_t kx = 0; kx < kw; kx++) {
        for (uint32_t c = 0; c < kc; c++) {
          const int32_t ws = (int32_t) kernel[(((size_t) col * kh + ky) * kw + kx) * kc + c] - 128;
          const uint32_t kp = ky * 16 + kx * kc + c;
          const uint32_t kb = kp / 32;
          const uint32_t lane = lane_lo + 32 * ((kp % 32) / 16);
          packed[((((size_t) nb * 2 + kb) * 64) + lane) * 16 + (kp % 16)] = (int8_t) ws;
        }
      }
    }
  }
}

/* ... centred on kernel zero point 127 (element 127 - w, activations ^ 0x7F, no row term; qnnp_pack_conv_rows32_centred127 below has the algebra) */
static inline void qnnp_pack_conv_rows16_centred127(
    uint32_t n, uint32_t kh, uint32_t kw, uint32_t kc, uint32_t n_pad, uint8_t izp,
    const uint8_t* kernel, const int32_t* bias, int8_t* packed /* [n_pad / 32][2][64][16] */, int32_t* biasc /* [n_pad] */)
{
  memset(packed, 0, (size_t) n_pad * 64);
  const uint32_t a_off = (uint32_t) (127 - (int32_t) izp);
  for (uint32_t col = 0; col < n_pad; col++) {
    uint32_t b = 0;
    if (col < n) {
      const uint32_t nb = col / 32;
      const uint32_t lane_lo = col % 32;
      uint32_t wsum = 0;                                   /* sum (w - 127), mod 2^32 */
      for (uint32_t ky = 0; ky < kh; ky++) {
        for (uint32_t kx = 0; kx < kw; kx++) {
          for (uint32_t c = 0; c < kc; c++) {
            const int32_t w = (int32_t) kernel[(((size_t) col * kh + ky) * kw + kx) * kc + c];
            wsum += (uint32_t) (w - 127);
            const uint32_t kp = ky * 16 + kx * kc + c;
            const uint32_t kb = kp / 32;
            const uint32_t lane = lane_lo + 32 * ((kp % 32) / 16);
            packed[((((size_t) nb * 2 + kb) * 64) + lane) * 16 + (kp % 16)] = (int8_t) (127 - w);
          }
        }
      }
      b = (uint32_t) bias[col] + a_off * wsum;
    }
    biasc[col] = (int32_t) b;
  }
}

/*
 * The same with 32-byte row slots (hip/q8convc3.hip, q8_conv_c3rows32_kernel): windows whose rows are 17 .. 32 bytes of a dense
 * 3-channel image -- ResNet's 7x7 entry layer, bench/convolution.cc:646: 21 bytes -- or have more than four rows. Packed K index
 * = ky * 32 + kx * 3 + c: K block ky is kernel row ky, its two 16-byte halves are the two fetches of that row; the slot's
 * unused bytes are zero weights. kernel_height K blocks per 32 channels.
 */
static inline size_t qnnp_conv_rows32_size(uint32_t n_pad, uint32_t kh) { return (size_t) (n_pad / 32) * kh * 1024; }
static inline void qnnp_pack_conv_rows32(
    uint32_t n, uint32_t kh, uint32_t kw, uint32_t kc, uint32_t n_pad,
    const uint8_t* kernel, int8_t* packed /* [n_pad / 32][kh][64][16] */)
{
  memset(packed, 0, qnnp_conv_rows32_size(n_pad, kh));
  for (uint32_t col = 0; col < n; col++) {
    const uint32_t nb = col / 32;
    const uint32_t lane_lo = col % 32;
    for (uint32_t ky = 0; ky < kh; ky++) {
      for (uint32_t kx = 0; kx < kw; kx++) {
        for (uint32_t c = 0; c < kc; c++) {
          const int32_t ws = (int32_t) kernel[(((size_t) col * kh + ky) * kw + kx) * kc + c] - 128;
          const uint32_t kp = kx * kc + c;             /* byte of the row slot: < 32 */
          const uint32_t lane = lane_lo + 32 * (kp / 16);
          packed[((((size_t) nb * kh + ky) * 64) + lane) * 16 + (kp % 16)] = (int8_t) ws;
        }
      }
    }
  }
}

/*
 * The same image centred on kernel zero point 127 (hip/q8convc3.hip with IgemmParams.a_flip = 0x7F7F7F7F): element 127 - w
 * (= w ^ 0x7F read as int8), the kernel re-centres the activations with the same mask (a ^ 0x7F = 127 - a; padding pixels hold
 * izp ^ 0x7F), so that
 *   bias + sum (a - izp)(w - 127) = biasc + sum (127 - a)(127 - w),   biasc[n] = bias[n] + (127 - izp) * sum_k (w(n, k) - 127)
 * and the per-pixel row term -- a third of the kernel's matrix instructions -- disappears (qnnp_pack_igemm_w_centred127 below is
 * the same algebra for the GEMM image).
 */
static inline void qnnp_pack_conv_rows32_centred127(
    uint32_t n, uint32_t kh, uint32_t kw, uint32_t kc, uint32_t n_pad, uint8_t izp,
    const uint8_t* kernel, const int32_t* bias, int8_t* packed /* [n_pad / 32][kh][64][16] */, int32_t* biasc /* [n_pad] */)
{
  memset(packed, 0, qnnp_conv_rows32_size(n_pad, kh));
  const uint32_t a_off = (uint32_t) (127 - (int32_t) izp);
  for (uint32_t col = 0; col < n_pad; col++) {
    uint32_t b = 0;
    if (col < n) {
      const uint32_t nb = col / 32;
      const uint32_t lane_lo = col % 32;
      uint32_t wsum = 0;                                   /* sum (w - 127), mod 2^32 */
      for (uint32_t ky = 0; ky < kh; ky++) {
        for (uint32_t kx = 0; kx < kw; kx++) {
          for (uint32_t c = 0; c < kc; c++) {
            const int32_t w = (int32_t) kernel[(((size_t) col * kh + ky) * kw + kx) * kc + c];
            wsum += (uint32_t) (w - 127);
            const uint32_t kp = kx * kc + c;
            const uint32_t lane = lane_lo + 32 * (kp / 16);
            packed[((((size_t) nb * kh + ky) * 64) + lane) * 16 + (kp % 16)] = (int8_t) (127 - w);
          }
        }
      }
      b = (uint32_t) bias[col] + a_off * wsum;
    }
    biasc[col] = (int32_t) b;
  }
}

static inline void qnnp_pack_igemm_w(
    uint32_t groups, uint32_t n, uint32_t k_total,
    uint32_t n_pad, uint32_t k_pad,
    uint8_t izp, uint8_t kzp,
    const uint8_t* kernel, const int32_t* bias,
    int8_t* packed, int32_t* bias2)
{
  const uint32_t nblocks = n_pad / 32;
  const uint32_t kblocks = k_pad / 32;
  memset(packed, 0, qnnp_igemm_packed_weights_size(groups, n_pad, k_pad));
  const uint32_t a_off = (uint32_t) (128 - (int32_t) izp);
  const uint32_t w_off = (uint32_t) (128 - (int32_t) kzp);
  for (uint32_t g = 0; g < groups; g++) {
    for (uint32_t col = 0; col < n_pad; col++) {
      uint32_t b2 = 0;
      if (col < n) {
        const uint8_t* src = kernel + ((size_t) g * n + col) * k_total;
        const uint32_t nb = col / 32;
        const uint32_t lane_lo = col % 32;
        uint32_t wsum = 0;
        for (uint32_t kk = 0; kk < k_total; kk++) {
          const int32_t ws = (int32_t) src[kk] - 128;
          wsum += (uint32_t) ws;
          const uint32_t kb = kk / 32;
          const uint32_t lane = lane_lo + 32 * ((kk % 32) / 16);
          const size_t dst = ((((size_t) g * nblocks + nb) * kblocks + kb) * 64 + lane) * 16 + (kk % 16);
          packed[dst] = (int8_t) ws;
        }
        b2 = (uint32_t) bias[(size_t) g * n + col] + a_off * wsum + k_total * a_off * w_off;
      }
      bias2[(size_t) g * n_pad + col] = (int32_t) b2;
    }
  }
}

/*
 * Zero-point-centred GEMM image (hip/q8gemm256c.hip) for kernel zero point 127: same fragment geometry as
 * qnnp_pack_igemm_w, element w'' = 127 - w (= w ^ 0x7F read as int8: always in [-128, 127]). The kernel recentres
 * the activations with the same mask, a'' = a ^ 0x7F = 127 - a, so a'' w'' = (a - 127)(w - 127) and
 *   bias + sum (a - izp)(w - 127) = biasc + sum a'' w'',   biasc[n] = bias[n] + (127 - izp) * sum_k (w(n,k) - 127)
 * -- no per-row kernel-zero-point term (the reference folds only the INPUT zero point into its packed bias,
 * src/qnnpack/pack.h:24-43; here both zero points end up in the weights' image and the bias).
 * Single group; padding columns and K padding hold zero weights (and a zero bias).
 * (Kernel zero point 128 needs nothing of its own: qnnp_pack_igemm_w's image w ^ 0x80 = w - 128 is the centred one and
 *  its bias2 has no kernel-zero-point part.)
 */
static inline void qnnp_pack_igemm_w_centred127(
    uint32_t n, uint32_t k_total, uint32_t k_pad, uint32_t n_pad,
    uint8_t izp,
    const uint8_t* kernel, const int32_t* bias,
    int8_t* packed, int32_t* biasc)
{
  const uint32_t kblocks = k_pad / 32;                     /* (K padding, if any: zero weights, as padding columns) */
  memset(packed, 0, (size_t) n_pad * k_pad);
  const uint32_t a_off = (uint32_t) (127 - (int32_t) izp);
  for (uint32_t col = 0; col < n_pad; col++) {
    uint32_t b = 0;
    if (col < n) {
      const uint8_t* src = kernel + (size_t) col * k_total;
      const uint32_t nb = col / 32;
      const uint32_t lane_lo = col % 32;
      uint32_t wsum = 0;                                   /* sum (w - 127), mod 2^32 */
      for (uint32_t kk = 0; kk < k_total; kk++) {
        wsum += (uint32_t) ((int32_t) src[kk] - 127);
        const uint32_t kb = kk / 32;
        const uint32_t lane = lane_lo + 32 * ((kk % 32) / 16);
        packed[((((size_t) nb * kblocks + kb) * 64) + lane) * 16 + (kk % 16)] = (int8_t) (127 - (int32_t) src[kk]);
      }
      b = (uint32_t) bias[col] + a_off * wsum;
    }
    biasc[col] = (int32_t) b;
  }
}

/*
 * Images of the fused-block strip kernel (hip/q8fusedstrip.hip), derived from the STANDARD images above (fused-block.c
 * reads those back from the device: the caller's kernel and bias arrays are gone by then). Centred element: w ^ flip
 * with flip = 0x80 for kernel zero point 128 (= w' = w - 128) and 0x7F for 127 (= ~w' = 127 - w); padding positions
 * stay zero weights. With
 *   bias     = bias2 - (128 - izp) * sum w' - K (128 - izp)(128 - kzp)          (undoing qnnp_pack_igemm_w's folding)
 *   biasc[n] = bias + (kzp - izp) * sum_k (w - kzp)   [+ 2^31 where the stage's rounding sequence is an offset form]
 * bias + sum (a - izp)(w - kzp) = biasc + sum (a ^ flip)(w ^ flip), both factors valid int8 (kzp in {127, 128}).
 * Output fragments: [ceil(n / 32)][ceil(k / 32)] of 1 KiB, the geometry of qnnp_pack_igemm_w without its K padding to 64.
 */
static inline void qnnp_strip_pointwise_images(
    const int8_t* std_w, const int32_t* std_bias2, uint32_t k_pad_std,
    uint32_t n, uint32_t k, uint8_t izp8, uint8_t kzp8, int offset_form,
    int8_t* frags, int32_t* biasc)
{
  const uint32_t nb = (n + 31u) / 32u, kb = (k + 31u) / 32u;
  const uint32_t kblocks_std = k_pad_std / 32u;
  const uint32_t izp = izp8, kzp = kzp8;
  const uint32_t offset = offset_form ? UINT32_C(0x80000000) : 0u;
  memset(frags, 0, (size_t) nb * kb * 1024u);
  memset(biasc, 0, sizeof(int32_t) * nb * 32u);
  for (uint32_t col = 0; col < n; col++) {
    const uint32_t b = col / 32u, lane_lo = col % 32u;
    uint32_t wsum = 0;                                   /* sum of w' = w - 128, mod 2^32 */
    for (uint32_t kk = 0; kk < k; kk++) {
      const uint32_t lane = lane_lo + 32u * ((kk % 32u) / 16u);
      const int8_t ws = std_w[(((size_t) b * kblocks_std + kk / 32u) * 64u + lane) * 16u + (kk % 16u)];
      wsum += (uint32_t) (int32_t) ws;
      frags[(((size_t) b * kb + kk / 32u) * 64u + lane) * 16u + (kk % 16u)] = kzp == 128 ? ws : (int8_t) ~ws;
    }
    const uint32_t bias = (uint32_t) std_bias2[col] - (128u - izp) * wsum - k * (128u - izp) * (128u - kzp);
    const uint32_t wsum_c = wsum + k * (128u - kzp);     /* sum (w - kzp) */
    biasc[col] = (int32_t) (bias + (kzp - izp) * wsum_c + offset);
  }
}

/* the depthwise stage's: int8 [9][hidden_pad] = w - 128 or 127 - w from the int16 image w - kzp of qnnp_pack_dwconv_w,
 * biasc[c] = bias + (kzp - izp) * sum_taps (w - kzp) with bias recovered from bias1 = bias + 9 izp kzp - izp sum w */
static inline void qnnp_strip_depthwise_images(
    const int16_t* wadj, const int32_t* bias1, uint32_t c_pad,
    uint32_t ch, uint32_t hidden_pad, uint8_t izp8, uint8_t kzp8, int offset_form,
    int8_t* w2, int32_t* biasc)
{
  const uint32_t izp = izp8, kzp = kzp8;
  const uint32_t offset = offset_form ? UINT32_C(0x80000000) : 0u;
  memset(w2, 0, (size_t) 9 * hidden_pad);
  memset(biasc, 0, sizeof(int32_t) * hidden_pad);
  for (uint32_t c = 0; c < ch; c++) {
    uint32_t sum_adj = 0;                                /* sum (w - kzp) */
    for (uint32_t t = 0; t < 9; t++) {
      const int32_t x = wadj[(size_t) t * c_pad + c];
      sum_adj += (uint32_t) x;
      w2[(size_t) t * hidden_pad + c] = (int8_t) (kzp == 128 ? x : -x);     /* w - 128, or 127 - w */
    }
    const uint32_t sum_w = sum_adj + 9u * kzp;
    const uint32_t bias = (uint32_t) bias1[c] - 9u * izp * kzp + izp * sum_w;
    biasc[c] = (int32_t) (bias + (kzp - izp) * sum_adj + offset);
  }
}

/*
 * depthwise image: wadj[tap][c] = w[c][ky][kx] - kzp as int16, tap = ky*kw + kx,
 * row length c_pad (zero padded); bias1[c] = bias[c] + taps*izp*kzp - izp*sum_taps w
 * -- the folding of pack_q8dw_w (src/qnnpack/pack.h:146,151,159), so the kernel
 * accumulates sum a*(w - kzp) over raw uint8 activations.
 * Source kernel layout: [c][ky][kx] (groups = c, one input/output channel each).
 */
static inline void qnnp_pack_dwconv_w(
    uint32_t channels, uint32_t c_pad, uint32_t kh, uint32_t kw,
    uint8_t izp, uint8_t kzp,
    const uint8_t* kernel, const int32_t* bias,
    int16_t* wadj, int32_t* bias1)
{
  const uint32_t taps = kh * kw;
  memset(wadj, 0, sizeof(int16_t) * (size_t) taps * c_pad);
  memset(bias1, 0, sizeof(int32_t) * (size_t) c_pad);
  for (uint32_t c = 0; c < channels; c++) {
    uint32_t wsum = 0;
    for (uint32_t t = 0; t < taps; t++) {
      const uint8_t w = kernel[(size_t) c * taps + t];
      wsum += w;
      wadj[(size_t) t * c_pad + c] = (int16_t) ((int32_t) w - (int32_t) kzp);
    }
    bias1[c] = (int32_t) ((uint32_t) bias[c] + taps * (uint32_t) izp * (uint32_t) kzp - (uint32_t) izp * wsum);
  }
}

/*
 * Range class of a depthwise weight image x = w - kzp (int16, as packed above) for the int8 dot-product flavour of the
 * 3x3 column kernel: 1 = every x fits int8 as it is (kzp == 128 with full-range weights, i.e. what PyTorch's
 * symmetric qint8 weights become), 2 = every -x does (kzp == 127, the zero point of the reference's benchmarks,
 * bench/convolution.cc:71-74), 0 = neither (the int16 pair kernel serves those).
 */
static inline uint32_t qnnp_dwconv_weight_range(const int16_t* wadj, size_t count)
{
  int32_t lo = 0, hi = 0;
  for (size_t i = 0; i < count; i++) {
    const int32_t x = wadj[i];
    if (x < lo) lo = x;
    if (x > hi) hi = x;
  }
  if (lo >= -128 && hi <= 127) return 1;
  if (lo >= -127 && hi <= 128) return 2;
  return 0;
}

/*
 * Third image, 3x3 depthwise operators whose range class is 1 or 2: the tap weights exactly as the int8 dot-product
 * walk of the column kernel keeps them in registers -- image[r][c] = (x_r0, x_r1, x_r2, 0) as int8, x = w - kzp
 * (class 1) or kzp - w (class 2) -- followed by image[3][c] = bias1[c] + (128 | 127) * sum_taps (w - kzp), the bias
 * that goes with activations re-centred as a ^ 0x80 (= a - 128) or a ^ 0x7f (= 127 - a). Four 16-byte loads per
 * thread instead of ten loads and ~100 unpacking instructions.
 */
static inline void qnnp_pack_dwconv_dot4(
    uint32_t c_pad, uint32_t range, const int16_t* wadj /* [9][c_pad] */, const int32_t* bias1 /* [c_pad] */,
    uint32_t* image /* [4][c_pad] */)
{
  for (uint32_t c = 0; c < c_pad; c++) {
    uint32_t xsum = 0;
    for (uint32_t r = 0; r < 3; r++) {
      uint32_t q = 0;
      for (uint32_t k = 0; k < 3; k++) {
        const uint32_t x = (uint32_t) (int32_t) wadj[(size_t) (r * 3 + k) * c_pad + c];
        xsum += x;
        q |= ((range == 2 ? 0u - x : x) & 0xFFu) << (8 * k);
      }
      image[(size_t) r * c_pad + c] = q;
    }
    image[(size_t) 3 * c_pad + c] = (uint32_t) bias1[c] + (range == 2 ? 127u : 128u) * xsum;
  }
}

/*
 * The same for 5x5 depthwise operators (q8dwconv.hip, kernel H), eight rows of c_pad words:
 *   image[ky][c], ky = 0..4 = (x_ky0, x_ky1, x_ky2, x_ky3)   columns 0..3 of kernel row ky, met by the transposed quads
 *   image[5][c]             = (x_04, x_14, x_24, x_34)       column 4 of kernel rows 0..3, met by the sliding column
 *   image[6][c]             = x_44 at byte c % 4, else 0     column 4 of the newest row: one-hot against the raw dword
 *   image[7][c]             = bias1[c] + (128 | 127) * sum_taps (w - kzp)
 * x = w - kzp (class 1) or kzp - w (class 2) as int8.
 */
static inline void qnnp_pack_dwconv_dot4_5x5(
    uint32_t c_pad, uint32_t range, const int16_t* wadj /* [25][c_pad] */, const int32_t* bias1 /* [c_pad] */,
    uint32_t* image /* [8][c_pad] */)
{
  for (uint32_t c = 0; c < c_pad; c++) {
    uint32_t xsum = 0, col4 = 0;
    for (uint32_t r = 0; r < 5; r++) {
      uint32_t q = 0;
      for (uint32_t k = 0; k < 5; k++) {
        const uint32_t x = (uint32_t) (int32_t) wadj[(size_t) (r * 5 + k) * c_pad + c];
        xsum += x;
        const uint32_t b = (range == 2 ? 0u - x : x) & 0xFFu;
        if (k < 4) {
          q |= b << (8 * k);
        } else if (r < 4) {
          col4 |= b << (8 * r);
        } else {
          image[(size_t) 6 * c_pad + c] = b << (8 * (c & 3u));
        }
      }
      image[(size_t) r * c_pad + c] = q;
    }
    image[(size_t) 5 * c_pad + c] = col4;
    image[(size_t) 7 * c_pad + c] = (uint32_t) bias1[c] + (range == 2 ? 127u : 128u) * xsum;
  }
}

/*
 * depthwise image for the MFMA kernel (q8dwconv.hip, kernel D): the signed tap weights
 *     x[t][c] = w[c][t] - kzp   in [-255, 255]
 * do not fit int8, so they are split into up to three int8 parts x = x0 + x1 + x2
 * (x0 = clamp(x, -128, 127), x1 = clamp(x - x0, -128, 127), x2 = the rest; x2 != 0 only for x = 255),
 * stored as xparts[part][tap][c_pad32]; the kernel multiplies a' = a - 128 (uint8 -> int8 by flipping
 * the top bit) by each part on the matrix cores (one diagonal 32x32 operand per tap and part) and
 *     biasm[c] = bias[c] + (128 - izp) * sum_t x[t][c]
 * makes  biasm + sum_t a'_t * x_t  ==  bias + sum_t (a_t - izp) * (w_t - kzp)   (padding taps read a = izp).
 * Returns the number of parts actually needed (1 when every x fits int8, e.g. kzp == 128).
 * Source kernel layout: [c][ky][kx], as qnnp_pack_dwconv_w.
 */
static inline uint32_t qnnp_pack_dwconv_mfma(
    uint32_t channels, uint32_t c_pad32, uint32_t kh, uint32_t kw,
    uint8_t izp, uint8_t kzp,
    const uint8_t* kernel, const int32_t* bias,
    int8_t* xparts /* [3][taps][c_pad32] */, int32_t* biasm /* [c_pad32] */)
{
  const uint32_t taps = kh * kw;
  memset(xparts, 0, (size_t) 3 * taps * c_pad32);
  memset(biasm, 0, sizeof(int32_t) * (size_t) c_pad32);
  uint32_t parts = 1;
  for (uint32_t c = 0; c < channels; c++) {
    int32_t xsum = 0;
    for (uint32_t t = 0; t < taps; t++) {
      int32_t x = (int32_t) kernel[(size_t) c * taps + t] - (int32_t) kzp;
      xsum += x;
      for (uint32_t part = 0; part < 3; part++) {
        int32_t piece = x;
        if (piece > 127) piece = 127;
        if (piece < -128) piece = -128;
        xparts[((size_t) part * taps + t) * c_pad32 + c] = (int8_t) piece;
        x -= piece;
        if (piece != 0 && part + 1 > parts) parts = part + 1;
      }
    }
    biasm[c] = (int32_t) ((uint32_t) bias[c] + (uint32_t) (128 - (int32_t) izp) * (uint32_t) xsum);
  }
  return parts;
}
