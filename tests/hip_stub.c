/*
 * hip_stub.c -- TEST INFRASTRUCTURE: a host-memory stand-in for the HIP seam (qnnpack_amd/csrc/hip/qnnp_hip.h) so the
 * plain-C host code of the product (create / setup / run plumbing / delete, weight packers, offset tables) can be
 * exercised under AddressSanitizer + UndefinedBehaviorSanitizer on a machine without a GPU (SURVEY.md section 5:
 * host-sanitizer cleanliness). "Device" memory is malloc, copies are memcpy, kernel launches validate their argument
 * block and touch the first and last byte of every tensor / table they were given (so ASan sees an undersized
 * allocation), nothing is computed. Never linked into the product.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "hip/qnnp_hip.h"

static int g_bound = 0;
static int g_async = 0;
static void* g_stream = NULL;
static volatile uint8_t g_sink;

static void touch(const void* p, size_t bytes)
{
  if (p != NULL && bytes != 0) {
    g_sink ^= ((const volatile uint8_t*) p)[0];
    g_sink ^= ((const volatile uint8_t*) p)[bytes - 1];
  }
}

int qnnp_hip_init(int device) { (void) device; g_bound = 1; return QNNP_HIP_OK; }
int qnnp_hip_bind(int device) { return device == 0 && g_bound ? QNNP_HIP_OK : QNNP_HIP_ENODEV; }
int qnnp_hip_shutdown(void) { g_bound = 0; return QNNP_HIP_OK; }
int qnnp_hip_device_count(void) { return 1; }
int qnnp_hip_select(int device) { return device == 0 && g_bound ? QNNP_HIP_OK : QNNP_HIP_ENODEV; }
int qnnp_hip_device(void) { return g_bound ? 0 : -1; }
int qnnp_hip_enter(int device) { return device == 0 && g_bound ? 0 : -1; }
void qnnp_hip_leave(int token) { (void) token; }
int qnnp_hip_device_info(char* arch, size_t arch_len, int* cus, int* clock_khz, size_t* mem_bytes)
{
  if (arch != NULL && arch_len != 0) { strncpy(arch, "gfx950-stub", arch_len - 1); arch[arch_len - 1] = '\0'; }
  if (cus != NULL) *cus = 256;
  if (clock_khz != NULL) *clock_khz = 2400000;
  if (mem_bytes != NULL) *mem_bytes = (size_t) 1 << 30;
  return QNNP_HIP_OK;
}
int qnnp_hip_compute_units(void) { return 256; }
static int g_streaming = 1;
void qnnp_hip_set_streaming_stores(int on) { g_streaming = on != 0; }
int qnnp_hip_streaming_stores(void) { return g_streaming; }
void qnnp_hip_set_stream(void* stream) { g_stream = stream; }
void* qnnp_hip_get_stream(void) { return g_stream; }
void qnnp_hip_set_async(int async) { g_async = async != 0; }
int qnnp_hip_get_async(void) { return g_async; }
int qnnp_hip_stream_sync(void) { return QNNP_HIP_OK; }
const uint8_t* qnnp_hip_fill_table(void) { return NULL; }

/* every "device" allocation carries a tag in front so that is_device_pointer can tell it from caller memory */
#define STUB_MAGIC UINT64_C(0x51AB51AB51AB51AB)
void* qnnp_hip_alloc(size_t bytes)
{
  uint64_t* p = (uint64_t*) malloc(bytes + 16);
  if (p == NULL) return NULL;
  p[0] = STUB_MAGIC;
  p[1] = bytes;
  return p + 2;
}
void qnnp_hip_free(void* p) { if (p != NULL) free((uint64_t*) p - 2); }
int qnnp_hip_h2d(void* dst, const void* src, size_t bytes, int async) { (void) async; memcpy(dst, src, bytes); return QNNP_HIP_OK; }
int qnnp_hip_d2h(void* dst, const void* src, size_t bytes, int async) { (void) async; memcpy(dst, src, bytes); return QNNP_HIP_OK; }
int qnnp_hip_memset(void* dst, int value, size_t bytes) { memset(dst, value, bytes); return QNNP_HIP_OK; }
int qnnp_hip_is_device_pointer(const void* p) { (void) p; return 0; }   /* the test hands host tensors: staged path */

int qnnp_hip_timer_create(void** timer) { *timer = malloc(1); return *timer != NULL ? QNNP_HIP_OK : QNNP_HIP_ENOMEM; }
int qnnp_hip_timer_start(void* timer) { (void) timer; return QNNP_HIP_OK; }
int qnnp_hip_timer_stop_ms(void* timer, float* ms) { (void) timer; *ms = 1.0f; return QNNP_HIP_OK; }
void qnnp_hip_timer_destroy(void* timer) { free(timer); }

int qnnp_hip_graph_capturing(void) { return 0; }
int qnnp_hip_graph_begin(void) { return QNNP_HIP_EINVAL; }   /* no graphs in the stub: timing falls back to the plain loop */
int qnnp_hip_graph_end(void** graph) { (void) graph; return QNNP_HIP_EINVAL; }
int qnnp_hip_graph_device(void* graph) { (void) graph; return 0; }
int qnnp_hip_graph_launch(void* graph) { (void) graph; return QNNP_HIP_EINVAL; }
int qnnp_hip_graph_time(void* graph, int warmup, int iters, float* avg_ms) { (void) graph; (void) warmup; (void) iters; (void) avg_ms; return QNNP_HIP_EINVAL; }
int qnnp_hip_graph_time_median(void* graph, int warmup, int iters, int samples, float* avg_ms)
{ (void) graph; (void) warmup; (void) iters; (void) samples; (void) avg_ms; return QNNP_HIP_EINVAL; }
int qnnp_hip_graph_sync(void* graph) { (void) graph; return QNNP_HIP_EINVAL; }
void qnnp_hip_graph_destroy(void* graph) { (void) graph; }

/* ---- "kernels": check the argument block against the documented layout, touch every buffer end to end ---- */
int qnnp_hip_deconv_s2_run(const struct qnnp_hip_deconv_s2_args* a, const char** kernel_name)
{
  (void) kernel_name;
  if (a == NULL || a->batch == 0) return QNNP_HIP_EINVAL;
  for (int ph = 0; ph < 4; ph++) {
    touch(a->packed_w[ph], (size_t) a->n_pad * a->k_pad[ph]);
    touch(a->bias2[ph], (size_t) a->n_pad * 4);
  }
  return QNNP_HIP_EINVAL;   /* "outside the streaming kernel's range": the host then takes the phase-table path */
}

int qnnp_hip_igemm_run(const struct qnnp_hip_igemm_args* a, const char** kernel_name)
{
  if (a == NULL || a->rows == 0 || a->n == 0 || a->n_pad % 32 != 0 || a->k_pad % 64 != 0 || a->n_pad < a->n) return QNNP_HIP_EINVAL;
  if (kernel_name != NULL) *kernel_name = "stub_igemm";
  if (a->phases == NULL) {
    touch(a->packed_w, (size_t) a->groups * a->n_pad * a->k_pad);
    touch(a->bias2, (size_t) a->groups * a->n_pad * 4);
    if (a->offsets != NULL) touch(a->offsets, (size_t) a->rows_per_image * a->ks * 4);
  } else {
    touch(a->phases, (size_t) a->nphases * sizeof(struct qnnp_hip_igemm_phase));
    for (uint32_t i = 0; i < a->nphases; i++) {
      const struct qnnp_hip_igemm_phase* ph = &a->phases[i];
      touch(ph->packed_w, (size_t) a->n_pad * ph->k_pad);
      touch(ph->bias2, (size_t) a->n_pad * 4);
      touch(ph->offsets, (size_t) ph->rows_per_image * ph->ks * 4);
      touch(ph->out_rows, (size_t) ph->rows_per_image * 4);
    }
  }
  touch(a->input, a->input_bytes);
  return QNNP_HIP_OK;
}

int qnnp_hip_dwconv_run(const struct qnnp_hip_dwconv_args* a, const char** kernel_name)
{
  if (a == NULL || a->batch == 0 || a->channels == 0 || a->c_pad < a->channels) return QNNP_HIP_EINVAL;
  if (kernel_name != NULL) *kernel_name = "stub_dwconv";
  const size_t taps = (size_t) a->kernel_height * a->kernel_width;
  touch(a->wadj, taps * a->c_pad * 2);
  touch(a->bias1, (size_t) a->c_pad * 4);
  if (a->dwm_x != NULL) touch(a->dwm_x, (size_t) a->dwm_parts * taps * a->c_pad32);
  if (a->dwm_bias != NULL) touch(a->dwm_bias, (size_t) a->c_pad32 * 4);
  touch(a->input, ((size_t) a->batch * a->input_height * a->input_width - 1) * a->input_stride + a->channels);
  touch(a->output, ((size_t) a->batch * a->output_height * a->output_width - 1) * a->output_stride + a->channels);
  if (a->plan != NULL) a->plan->key = 0x80000000u;
  return QNNP_HIP_OK;
}

int qnnp_hip_vadd_run(const struct qnnp_hip_vadd_args* a, const char** kernel_name)
{
  if (a == NULL || a->rows == 0 || a->channels == 0) return QNNP_HIP_EINVAL;
  if (kernel_name != NULL) *kernel_name = "stub_vadd";
  touch(a->a, (size_t) (a->rows - 1) * a->a_stride + a->channels);
  touch(a->b, (size_t) (a->rows - 1) * a->b_stride + a->channels);
  touch(a->sum, (size_t) (a->rows - 1) * a->sum_stride + a->channels);
  return QNNP_HIP_OK;
}

int qnnp_hip_gavgpool_run(const struct qnnp_hip_gavgpool_args* a, const char** kernel_name)
{
  if (a == NULL || a->batch == 0 || a->width == 0 || a->channels == 0) return QNNP_HIP_EINVAL;
  if (kernel_name != NULL) *kernel_name = "stub_gavgpool";
  touch(a->input, (size_t) (a->batch * a->width - 1) * a->input_stride + a->channels);
  touch(a->output, (size_t) (a->batch - 1) * a->output_stride + a->channels);
  return QNNP_HIP_OK;
}

int qnnp_hip_fused_strip_bias_offset(const struct qnnp_hip_requant* rq) { return rq != NULL && rq->shift == 0; }
int qnnp_hip_fused_strip_supported(const struct qnnp_hip_fused_strip_args* a) { return a != NULL && a->hidden_channels % 32 == 0; }
int qnnp_hip_fused_strip_run(const struct qnnp_hip_fused_strip_args* a, const char** kernel_name)
{
  if (!qnnp_hip_fused_strip_supported(a)) return QNNP_HIP_EINVAL;
  if (kernel_name != NULL) *kernel_name = "stub_fused_strip";
  return QNNP_HIP_OK;
}
int qnnp_hip_fused_block_supported(const struct qnnp_hip_fused_args* a) { return a != NULL && a->hidden_channels % 16 == 0; }
int qnnp_hip_fused_block_run(const struct qnnp_hip_fused_args* a, const char** kernel_name)
{
  if (!qnnp_hip_fused_block_supported(a)) return QNNP_HIP_EINVAL;
  if (kernel_name != NULL) *kernel_name = "stub_fused";
  touch(a->dw_wadj, (size_t) 9 * a->dw_c_pad * 2);
  touch(a->project_w, (size_t) a->project_n_pad * a->project_k_pad);
  if (a->has_expand) touch(a->expand_w, (size_t) a->expand_n_pad * a->expand_k_pad);
  return QNNP_HIP_OK;
}
