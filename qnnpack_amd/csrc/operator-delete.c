/*
 * operator-delete.c -- qnnp_delete_operator.
 * Replaces reference src/operator-delete.c:15-28: frees everything the
 * operator owns -- here device allocations (weights, folded bias, offset table,
 * staging buffers) instead of host packed weights / indirection / zero buffers.
 * delete(NULL) -> invalid_parameter, as in the reference (:17-19).
 */
#include <stdlib.h>

#include <qnnpack.h>

#include "hip/qnnp_hip.h"
#include "operator.h"

enum qnnp_status qnnp_delete_operator(qnnp_operator_t op)
{
  if (op == NULL) {
    return qnnp_status_invalid_parameter;
  }
  /* the allocations belong to the operator's device (a negative token = the library was deinitialized: the
   * runtime then frees by pointer on whatever device is current, which HIP accepts) */
  const int token = qnnp_hip_enter(op->device);
  qnnp_hip_free(op->d_weights);
  qnnp_hip_free(op->d_weights_rows16);
  qnnp_hip_free(op->d_bias_rows);
  qnnp_hip_free(op->d_weights_dense);
  qnnp_hip_free(op->d_bias_dense);
  qnnp_hip_free(op->d_bias);
  qnnp_hip_free(op->d_weights_centred);
  qnnp_hip_free(op->d_strip);
  qnnp_hip_free(op->d_bias_centred);
  qnnp_hip_free(op->d_dwm_x);
  qnnp_hip_free(op->d_dwm_bias);
  qnnp_hip_free(op->d_dw_dot4);
  qnnp_hip_free(op->d_offsets);
  qnnp_hip_free(op->d_phase_table);
  for (uint32_t i = 0; i < op->deconv_phases && i < QNNP_MAX_DECONV_PHASES; i++) {
    qnnp_hip_free(op->phase[i].d_weights);
    qnnp_hip_free(op->phase[i].d_bias);
    qnnp_hip_free(op->phase[i].d_offsets);
    qnnp_hip_free(op->phase[i].d_out_rows);
  }
  qnnp_hip_free(op->d_stage_in);
  qnnp_hip_free(op->d_stage_in2);
  qnnp_hip_free(op->d_stage_out);
  free(op);
  qnnp_hip_leave(token);
  return qnnp_status_success;
}
