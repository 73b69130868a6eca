/*
 * hybrid_seam.c -- the reference tree with its q8 conv / GEMM hot path re-routed to the gfx950 build, as ONE library.
 * TEST / INTEGRATION INFRASTRUCTURE (built by oracle/Makefile target `hybrid` into oracle/_ref/libqnnpack_hybrid.so).
 *
 * What is linked together:
 *   - every object of the UNMODIFIED reference (oracle/_ref/obj, the O2 build) -- its CPU operators stay what they are;
 *     the entry points of the hot path (create / setup of convolution and fully connected, initialize, deinitialize,
 *     delete) are renamed qnnp_ref_* with objcopy so that both implementations can live in one image;
 *   - the reference's src/operator-run.c compiled from a generated copy that differs from the original by ONE inserted
 *     statement at the seam INTEGRATION.md section 2 names (the top of qnnp_run_operator, in front of the
 *     `switch (op->ukernel_type)` of src/operator-run.c:646):
 *         if (qnnp_hybrid_owns(op)) return qnnp_gfx950_run_operator(op, threadpool);
 *   - the product's host C objects and HIP kernels (qnnpack_amd/csrc/build), their public names renamed qnnp_gfx950_*;
 *   - this file: the public entry points of the hot path, which create DEVICE operators and remember them, and the
 *     ownership test the patched dispatch asks.
 * A caller sees the reference's API and nothing else: a convolution or fully connected operator runs on the MI355X, a
 * max-pooling / sigmoid / clamp / average-pooling operator runs the reference's SSE2 kernels, through the same
 * qnnp_run_operator and qnnp_delete_operator (tests/test_gpu_hybrid_seam.py).
 */
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <qnnpack.h>

/* the reference's own entry points (renamed at link time) */
enum qnnp_status qnnp_ref_initialize(void);
enum qnnp_status qnnp_ref_deinitialize(void);
enum qnnp_status qnnp_ref_delete_operator(qnnp_operator_t op);

/* the product's entry points (renamed at link time) */
enum qnnp_status qnnp_gfx950_initialize(void);
enum qnnp_status qnnp_gfx950_deinitialize(void);
enum qnnp_status qnnp_gfx950_delete_operator(qnnp_operator_t op);
enum qnnp_status qnnp_gfx950_run_operator(qnnp_operator_t op, pthreadpool_t threadpool);
enum qnnp_status qnnp_gfx950_create_convolution2d_nhwc_q8(
    uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, size_t, size_t,
    uint8_t, float, uint8_t, float, const uint8_t*, const int32_t*, uint8_t, float, uint8_t, uint8_t, uint32_t, qnnp_operator_t*);
enum qnnp_status qnnp_gfx950_setup_convolution2d_nhwc_q8(
    qnnp_operator_t, size_t, size_t, size_t, const uint8_t*, size_t, uint8_t*, size_t, pthreadpool_t);
enum qnnp_status qnnp_gfx950_create_fully_connected_nc_q8(
    size_t, size_t, uint8_t, float, uint8_t, float, const uint8_t*, const int32_t*, uint8_t, float, uint8_t, uint8_t, uint32_t,
    qnnp_operator_t*);
enum qnnp_status qnnp_gfx950_setup_fully_connected_nc_q8(qnnp_operator_t, size_t, const uint8_t*, size_t, uint8_t*, size_t);

/* ---- which operators live on the device: a small open-addressing set of handles ---- */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static qnnp_operator_t* g_slots = NULL;
static size_t g_capacity = 0, g_count = 0;
#define TOMBSTONE ((qnnp_operator_t) (uintptr_t) 1)

static size_t slot_of(qnnp_operator_t op, size_t capacity) { return (size_t) (((uintptr_t) op >> 4) * 0x9E3779B97F4A7C15ull) & (capacity - 1); }

static int grow_locked(void)
{
  const size_t capacity = g_capacity ? g_capacity * 2 : 64;
  qnnp_operator_t* slots = (qnnp_operator_t*) calloc(capacity, sizeof(qnnp_operator_t));
  if (slots == NULL) return 0;
  for (size_t i = 0; i < g_capacity; i++) {
    qnnp_operator_t op = g_slots[i];
    if (op == NULL || op == TOMBSTONE) continue;
    size_t s = slot_of(op, capacity);
    while (slots[s] != NULL) s = (s + 1) & (capacity - 1);
    slots[s] = op;
  }
  free(g_slots);
  g_slots = slots;
  g_capacity = capacity;
  return 1;
}

static int remember(qnnp_operator_t op)
{
  pthread_mutex_lock(&g_lock);
  int ok = 1;
  if ((g_count + 1) * 2 > g_capacity) ok = grow_locked();
  if (ok) {
    size_t s = slot_of(op, g_capacity);
    while (g_slots[s] != NULL && g_slots[s] != TOMBSTONE) s = (s + 1) & (g_capacity - 1);
    g_slots[s] = op;
    g_count++;
  }
  pthread_mutex_unlock(&g_lock);
  return ok;
}

static int find_locked(qnnp_operator_t op, size_t* where)
{
  if (g_capacity == 0 || op == NULL) return 0;
  size_t s = slot_of(op, g_capacity);
  for (size_t probes = 0; probes < g_capacity && g_slots[s] != NULL; probes++, s = (s + 1) & (g_capacity - 1)) {
    if (g_slots[s] == op) {
      if (where != NULL) *where = s;
      return 1;
    }
  }
  return 0;
}

/* asked by the patched dispatch of the reference's src/operator-run.c */
int qnnp_hybrid_owns(qnnp_operator_t op)
{
  pthread_mutex_lock(&g_lock);
  const int owned = find_locked(op, NULL);
  pthread_mutex_unlock(&g_lock);
  return owned;
}

static int forget(qnnp_operator_t op)
{
  pthread_mutex_lock(&g_lock);
  size_t where = 0;
  const int owned = find_locked(op, &where);
  if (owned) {
    g_slots[where] = TOMBSTONE;     /* (count keeps the tombstone: the table only grows, it is tiny) */
  }
  pthread_mutex_unlock(&g_lock);
  return owned;
}

/* ---- the public entry points of the hot path ---- */
enum qnnp_status qnnp_initialize(void)
{
  const enum qnnp_status ref = qnnp_ref_initialize();
  if (ref != qnnp_status_success) return ref;
  return qnnp_gfx950_initialize();         /* unsupported_hardware without a gfx950 device: there is no CPU fallback for the hot path */
}

enum qnnp_status qnnp_deinitialize(void)
{
  (void) qnnp_gfx950_deinitialize();
  return qnnp_ref_deinitialize();
}

enum qnnp_status qnnp_create_convolution2d_nhwc_q8(
    uint32_t input_padding_top, uint32_t input_padding_right, uint32_t input_padding_bottom, uint32_t input_padding_left,
    uint32_t kernel_height, uint32_t kernel_width, uint32_t subsampling_height, uint32_t subsampling_width,
    uint32_t dilation_height, uint32_t dilation_width, uint32_t groups, size_t group_input_channels, size_t group_output_channels,
    uint8_t input_zero_point, float input_scale, uint8_t kernel_zero_point, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias, uint8_t output_zero_point, float output_scale,
    uint8_t output_min, uint8_t output_max, uint32_t flags, qnnp_operator_t* convolution)
{
  qnnp_operator_t op = NULL;
  const enum qnnp_status status = qnnp_gfx950_create_convolution2d_nhwc_q8(
      input_padding_top, input_padding_right, input_padding_bottom, input_padding_left, kernel_height, kernel_width,
      subsampling_height, subsampling_width, dilation_height, dilation_width, groups, group_input_channels,
      group_output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias,
      output_zero_point, output_scale, output_min, output_max, flags, &op);
  if (status != qnnp_status_success) return status;
  if (!remember(op)) {
    (void) qnnp_gfx950_delete_operator(op);
    return qnnp_status_out_of_memory;
  }
  *convolution = op;
  return qnnp_status_success;
}

enum qnnp_status qnnp_setup_convolution2d_nhwc_q8(
    qnnp_operator_t convolution, size_t batch_size, size_t input_height, size_t input_width,
    const uint8_t* input, size_t input_pixel_stride, uint8_t* output, size_t output_pixel_stride, pthreadpool_t threadpool)
{
  if (!qnnp_hybrid_owns(convolution)) return qnnp_status_invalid_parameter;
  return qnnp_gfx950_setup_convolution2d_nhwc_q8(convolution, batch_size, input_height, input_width, input,
                                                 input_pixel_stride, output, output_pixel_stride, threadpool);
}

enum qnnp_status qnnp_create_fully_connected_nc_q8(
    size_t input_channels, size_t output_channels, uint8_t input_zero_point, float input_scale,
    uint8_t kernel_zero_point, float kernel_scale, const uint8_t* kernel, const int32_t* bias,
    uint8_t output_zero_point, float output_scale, uint8_t output_min, uint8_t output_max, uint32_t flags,
    qnnp_operator_t* fully_connected)
{
  qnnp_operator_t op = NULL;
  const enum qnnp_status status = qnnp_gfx950_create_fully_connected_nc_q8(
      input_channels, output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias,
      output_zero_point, output_scale, output_min, output_max, flags, &op);
  if (status != qnnp_status_success) return status;
  if (!remember(op)) {
    (void) qnnp_gfx950_delete_operator(op);
    return qnnp_status_out_of_memory;
  }
  *fully_connected = op;
  return qnnp_status_success;
}

enum qnnp_status qnnp_setup_fully_connected_nc_q8(
    qnnp_operator_t fully_connected, size_t batch_size, const uint8_t* input, size_t input_stride, uint8_t* output,
    size_t output_stride)
{
  if (!qnnp_hybrid_owns(fully_connected)) return qnnp_status_invalid_parameter;
  return qnnp_gfx950_setup_fully_connected_nc_q8(fully_connected, batch_size, input, input_stride, output, output_stride);
}

enum qnnp_status qnnp_delete_operator(qnnp_operator_t op)
{
  if (op == NULL) return qnnp_status_invalid_parameter;      /* reference src/operator-delete.c:17-19 */
  if (forget(op)) return qnnp_gfx950_delete_operator(op);
  return qnnp_ref_delete_operator(op);
}
