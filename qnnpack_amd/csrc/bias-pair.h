/*
 * bias-pair.h -- the folded bias table of an implicit-GEMM operator as it lives on the device: `count` int32 of
 * bias2 (pack.h) followed by the same `count` values + 2^31 (mod 2^32). The second copy is what the accumulators of
 * the kernels that use the lane forms of the requantization start from (hip/requant_math.h: the unsigned multiply-add
 * wants a + 2^31, and a table costs nothing per value where an add per value is what those forms remove).
 * Kernels find it at bias2 + groups * n_pad when the argument block says `bias2_pair`.
 */
#pragma once

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hip/qnnp_hip.h"

/* device copy of {bias2, bias2 + 2^31}; NULL on failure */
static inline int32_t* qnnp_upload_bias_pair(const int32_t* host_bias, size_t count)
{
  const size_t bytes = sizeof(int32_t) * count;
  int32_t* pair = (int32_t*) malloc(2 * bytes);
  if (pair == NULL) return NULL;
  memcpy(pair, host_bias, bytes);
  for (size_t i = 0; i < count; i++) {
    pair[count + i] = (int32_t) ((uint32_t) host_bias[i] ^ UINT32_C(0x80000000));
  }
  int32_t* d = (int32_t*) qnnp_hip_alloc(2 * bytes);
  if (d != NULL && qnnp_hip_h2d(d, pair, 2 * bytes, 0) != QNNP_HIP_OK) {
    qnnp_hip_free(d);
    d = NULL;
  }
  free(pair);
  return d;
}
