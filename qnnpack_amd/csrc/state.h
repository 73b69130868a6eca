/*
 * state.h -- process-wide library state. Role-equivalent to the reference's
 * global `qnnp_params` (src/qnnpack/params.h:520-538): there it is the per-ISA
 * microkernel table filled by init(); here it records that a gfx950 device is
 * bound plus the A/B knobs of qnnpack_gfx950.h, which are read at SETUP time only
 * (an operator keeps the variant it was set up with). Everything the launch path
 * needs -- stream, asynchrony, capture -- lives in the per-device contexts and
 * thread-local state of hip/runtime.hip, not here.
 */
#pragma once

#include <stdbool.h>

struct qnnp_state {
  bool initialized;
  int requested_device;   /* primary device asked for before qnnp_initialize; -1: env / current */
  int opt_gemm_kernel;    /* 0 auto, 1 generic, 2 big-tile */
  int opt_dwconv_kernel;  /* 0 auto, 1 generic, 2 LDS-tiled */
  int opt_timing_graph;   /* 1: qnnp_gfx950_time_operator* replay a hipGraph of the launches (default) */
  int opt_fused_kernel;   /* fused blocks: 0 auto (strip kernel where it applies), 1 tile kernel only, 2 strip kernel only */
  int opt_fused_rows;     /* strip kernel: output rows per strip, 0 = its own choice */
  int opt_fused_weights;  /* strip kernel: 0 = its own choice, 1 = chunk weights staged in LDS or refuse, 2 = weights from L2 */
};

extern struct qnnp_state qnnp_state;
