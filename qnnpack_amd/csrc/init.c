/*
 * init.c -- qnnp_initialize / qnnp_deinitialize and the gfx950 extension knobs.
 *
 * Replaces reference src/init.c:244-263. The reference probes the CPU with
 * cpuinfo and fills a microkernel table once (pthread_once, :40, :251); this
 * build probes for a gfx950 GPU through the HIP shim and binds to it once.
 * No GPU => qnnp_status_unsupported_hardware (reference: same status when no
 * ISA table could be filled, init.c:253-257). There is no CPU fallback.
 */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include <qnnpack.h>
#include <qnnpack_gfx950.h>
#include <qnnpack_gfx950_test.h>

#include "hip/qnnp_hip.h"
#include "log.h"
#include "operator.h"
#include "state.h"

struct qnnp_state qnnp_state = {
  .initialized = false,
  .requested_device = -1,
  .opt_gemm_kernel = 0,
  .opt_dwconv_kernel = 0,
  .opt_timing_graph = 1,
};

static pthread_mutex_t init_lock = PTHREAD_MUTEX_INITIALIZER;

enum qnnp_status qnnp_initialize(void)
{
  enum qnnp_status status = qnnp_status_success;
  pthread_mutex_lock(&init_lock);
  if (!qnnp_state.initialized) {
    int device = qnnp_state.requested_device;
    if (device < 0) {
      const char* env = getenv("QNNP_GFX950_DEVICE");
      if (env != NULL && env[0] != '\0') device = atoi(env);
    }
    const int rc = qnnp_hip_init(device);
    if (rc == QNNP_HIP_OK) {
      qnnp_state.initialized = true;
    } else if (rc == QNNP_HIP_ENOMEM) {
      qnnp_log_error("qnnp_initialize: out of device memory while binding the gfx950 device");
      status = qnnp_status_out_of_memory;
    } else {
      qnnp_log_error("qnnp_initialize: no usable gfx950 (MI355X) device; this build has no CPU path");
      status = qnnp_status_unsupported_hardware;
    }
  }
  pthread_mutex_unlock(&init_lock);
  return status;
}

enum qnnp_status qnnp_deinitialize(void)
{
  pthread_mutex_lock(&init_lock);
  if (qnnp_state.initialized) {
    qnnp_hip_shutdown();
    qnnp_state.initialized = false;
  }
  pthread_mutex_unlock(&init_lock);
  return qnnp_status_success;
}

/* ---- qnnpack_gfx950.h ---- */

/* run `expr` (an int QNNP_HIP_* code or a value) inside the calling thread's selected device context */
#define QNNP_WITH_SELECTED_DEVICE(token) \
  const int token = qnnp_hip_enter(qnnp_hip_device())

enum qnnp_status qnnp_gfx950_set_device(int device)
{
  enum qnnp_status status = qnnp_status_success;
  if (device < 0) {
    return qnnp_status_invalid_parameter;
  }
  pthread_mutex_lock(&init_lock);
  if (!qnnp_state.initialized) {
    qnnp_state.requested_device = device;          /* the primary device qnnp_initialize will bind */
  } else {
    /* after initialization: bind the device if this is its first use and make it the calling THREAD's device --
     * operators created from this thread live there (one host thread per GPU drives a whole node) */
    if (qnnp_hip_bind(device) != QNNP_HIP_OK || qnnp_hip_select(device) != QNNP_HIP_OK) {
      qnnp_log_error("qnnp_gfx950_set_device: device %d is not a usable gfx950 GPU", device);
      status = qnnp_status_invalid_parameter;
    }
  }
  pthread_mutex_unlock(&init_lock);
  return status;
}

int qnnp_gfx950_get_device(void)
{
  return qnnp_state.initialized ? qnnp_hip_device() : -1;
}

int qnnp_gfx950_device_count(void)
{
  return qnnp_hip_device_count();
}

enum qnnp_status qnnp_gfx950_set_stream(void* hip_stream)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  qnnp_hip_set_stream(hip_stream);
  return qnnp_status_success;
}

enum qnnp_status qnnp_gfx950_set_async(int async)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  qnnp_hip_set_async(async);
  return qnnp_status_success;
}

enum qnnp_status qnnp_gfx950_synchronize(void)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  QNNP_WITH_SELECTED_DEVICE(token);
  const int rc = qnnp_hip_stream_sync();
  qnnp_hip_leave(token);
  return rc == QNNP_HIP_OK ? qnnp_status_success : qnnp_status_unsupported_hardware;
}

void* qnnp_gfx950_malloc(size_t bytes)
{
  if (!qnnp_state.initialized) return NULL;
  QNNP_WITH_SELECTED_DEVICE(token);
  void* p = qnnp_hip_alloc(bytes);
  qnnp_hip_leave(token);
  return p;
}

void qnnp_gfx950_free(void* device_ptr)
{
  if (qnnp_state.initialized) qnnp_hip_free(device_ptr);
}

enum qnnp_status qnnp_gfx950_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  QNNP_WITH_SELECTED_DEVICE(token);
  const int rc = qnnp_hip_h2d(dst_device, src_host, bytes, 0);
  qnnp_hip_leave(token);
  return rc == QNNP_HIP_OK ? qnnp_status_success : qnnp_status_invalid_parameter;
}

enum qnnp_status qnnp_gfx950_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  QNNP_WITH_SELECTED_DEVICE(token);
  const int rc = qnnp_hip_d2h(dst_host, src_device, bytes, 0);
  qnnp_hip_leave(token);
  return rc == QNNP_HIP_OK ? qnnp_status_success : qnnp_status_invalid_parameter;
}

enum qnnp_status qnnp_gfx950_memset(void* dst_device, int value, size_t bytes)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  QNNP_WITH_SELECTED_DEVICE(token);
  const int rc = qnnp_hip_memset(dst_device, value, bytes);
  qnnp_hip_leave(token);
  return rc == QNNP_HIP_OK ? qnnp_status_success : qnnp_status_invalid_parameter;
}

enum qnnp_status qnnp_gfx950_set_option(const char* key, int value)
{
  if (key == NULL) return qnnp_status_invalid_parameter;
  if (strcmp(key, "timing_graph") == 0 && (value == 0 || value == 1)) {
    qnnp_state.opt_timing_graph = value;
    return qnnp_status_success;
  }
  if (strcmp(key, "streaming_stores") == 0 && (value == 0 || value == 1)) {
    qnnp_hip_set_streaming_stores(value);
    return qnnp_status_success;
  }
  return qnnp_status_invalid_parameter;
}

/* include/qnnpack_gfx950_test.h: which kernel the operators set up from now on run on (tests, A/B tools) */
enum qnnp_status qnnp_gfx950_test_force_kernel(const char* key, int value)
{
  if (key == NULL) return qnnp_status_invalid_parameter;
  if (strcmp(key, "gemm_kernel") == 0 && value >= 0 && value <= 32 && !(value >= 17 && value <= 19)) {
    qnnp_state.opt_gemm_kernel = value;
    return qnnp_status_success;
  }
  if (strcmp(key, "fused_kernel") == 0 && value >= 0 && value <= 2) {
    qnnp_state.opt_fused_kernel = value;
    return qnnp_status_success;
  }
  if (strcmp(key, "fused_weights") == 0 && value >= 0 && value <= 2) {
    qnnp_state.opt_fused_weights = value;
    return qnnp_status_success;
  }
  if (strcmp(key, "fused_rows") == 0 && value >= 0 && value <= 4096) {
    qnnp_state.opt_fused_rows = value;
    return qnnp_status_success;
  }
  if (strcmp(key, "dwconv_kernel") == 0 && value >= 0 && value <= 9) {
    qnnp_state.opt_dwconv_kernel = value;
    return qnnp_status_success;
  }
  return qnnp_status_invalid_parameter;
}

/* include/qnnpack_gfx950_test.h */
int qnnp_gfx950_test_operator_ran_dense(qnnp_operator_t op)
{
  return op != NULL && op->ran_dense != 0;
}

enum qnnp_status qnnp_gfx950_operator_set_streaming_stores(qnnp_operator_t op, int value)
{
  if (op == NULL || value < -1 || value > 1) return qnnp_status_invalid_parameter;
  op->streaming_mode = value < 0 ? 0u : (uint32_t) value + 1u;      /* 0 = the process default, 1 = off, 2 = on */
  return qnnp_status_success;
}

const char* qnnp_gfx950_operator_kernel(qnnp_operator_t op)
{
  return op == NULL ? NULL : op->kernel_name;
}

enum qnnp_status qnnp_gfx950_device_info(
    char* arch, size_t arch_len, int* compute_units, int* clock_khz, size_t* hbm_bytes)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  return qnnp_hip_device_info(arch, arch_len, compute_units, clock_khz, hbm_bytes) == QNNP_HIP_OK ?
      qnnp_status_success : qnnp_status_unsupported_hardware;
}
