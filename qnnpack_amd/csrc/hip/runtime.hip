/*
 * runtime.hip -- device binding, memory, stream and event plumbing behind the
 * C-ABI seam of qnnp_hip.h. The reference has no counterpart (its runtime is
 * cpuinfo + pthreadpool, src/init.c:244-263); this is what "initialize" means
 * on an MI355X: bind one gfx950 device, keep one launch stream.
 */
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "qnnp_hip.h"

namespace {

struct Runtime {
  bool bound = false;
  int device = -1;
  hipStream_t stream = nullptr;  // nullptr = default stream
  hipDeviceProp_t props;
  uint8_t* fill_table = nullptr;  // [256][16]: entry v = sixteen bytes of value v (LDS-DMA padding sources)
  // hipGraph capture of a sequence of operator launches (qnnp_hip_graph_*)
  bool capturing = false;
  hipStream_t saved_stream = nullptr;    // the library stream while a capture redirects launches
  hipStream_t private_stream = nullptr;  // capture / replay stream when the library stream is the default stream
};

Runtime g_rt;

struct Timer {
  hipEvent_t start;
  hipEvent_t stop;
};

inline bool ok(hipError_t e) { return e == hipSuccess; }

}  // namespace

extern "C" {

int qnnp_hip_init(int device)
{
  int count = 0;
  if (!ok(hipGetDeviceCount(&count)) || count <= 0) {
    (void) hipGetLastError();
    return QNNP_HIP_ENODEV;
  }
  if (device < 0) {
    if (!ok(hipGetDevice(&device))) device = 0;
  }
  if (device >= count) return QNNP_HIP_ENODEV;
  if (!ok(hipSetDevice(device))) return QNNP_HIP_ENODEV;
  if (!ok(hipGetDeviceProperties(&g_rt.props, device))) return QNNP_HIP_ENODEV;
  // Only CDNA4: the kernels use v_mfma_i32_32x32x32_i8 and are built for gfx950 alone.
  if (std::strncmp(g_rt.props.gcnArchName, "gfx950", 6) != 0) {
    return QNNP_HIP_ENODEV;
  }
  g_rt.device = device;
  g_rt.stream = nullptr;
  if (g_rt.fill_table == nullptr) {
    uint8_t host[256 * 16];
    for (int v = 0; v < 256; v++) std::memset(host + v * 16, v, 16);
    if (!ok(hipMalloc(reinterpret_cast<void**>(&g_rt.fill_table), sizeof(host)))) return QNNP_HIP_ENOMEM;
    if (!ok(hipMemcpy(g_rt.fill_table, host, sizeof(host), hipMemcpyHostToDevice))) return QNNP_HIP_ENOMEM;
  }
  g_rt.bound = true;
  return QNNP_HIP_OK;
}

const uint8_t* qnnp_hip_fill_table(void) { return g_rt.fill_table; }

static unsigned long long* g_trace = nullptr;
static const size_t kTraceWords = 4096 * 4 * 8;
void* qnnp_hip_trace_buffer(void)
{
  if (g_trace == nullptr && getenv("QNNP_GFX950_TRACE") != nullptr) {
    if (hipMalloc(reinterpret_cast<void**>(&g_trace), kTraceWords * 8) != hipSuccess) g_trace = nullptr;
    else (void) hipMemset(g_trace, 0, kTraceWords * 8);
  }
  return g_trace;
}
int qnnp_hip_trace_dump(unsigned long long* host, size_t count)
{
  if (g_trace == nullptr) return QNNP_HIP_EINVAL;
  if (count > kTraceWords) count = kTraceWords;
  (void) hipDeviceSynchronize();
  return hipMemcpy(host, g_trace, count * 8, hipMemcpyDeviceToHost) == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int qnnp_hip_shutdown(void)
{
  if (g_rt.fill_table != nullptr) {
    (void) hipFree(g_rt.fill_table);
    g_rt.fill_table = nullptr;
  }
  g_rt.bound = false;
  g_rt.stream = nullptr;
  return QNNP_HIP_OK;
}

int qnnp_hip_device(void) { return g_rt.bound ? g_rt.device : -1; }

int qnnp_hip_device_info(char* arch, size_t arch_len, int* cus, int* clock_khz, size_t* mem_bytes)
{
  if (!g_rt.bound) return QNNP_HIP_ENODEV;
  if (arch != nullptr && arch_len > 0) {
    std::strncpy(arch, g_rt.props.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = '\0';
  }
  if (cus != nullptr) *cus = g_rt.props.multiProcessorCount;
  if (clock_khz != nullptr) *clock_khz = g_rt.props.clockRate;
  if (mem_bytes != nullptr) *mem_bytes = g_rt.props.totalGlobalMem;
  return QNNP_HIP_OK;
}

void qnnp_hip_set_stream(void* stream) { g_rt.stream = reinterpret_cast<hipStream_t>(stream); }
void* qnnp_hip_get_stream(void) { return reinterpret_cast<void*>(g_rt.stream); }

int qnnp_hip_stream_sync(void)
{
  return ok(hipStreamSynchronize(g_rt.stream)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

void* qnnp_hip_alloc(size_t bytes)
{
  if (!g_rt.bound) return nullptr;
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  if (!ok(hipMalloc(&p, bytes))) {
    (void) hipGetLastError();
    return nullptr;
  }
  return p;
}

void qnnp_hip_free(void* p)
{
  if (p != nullptr) (void) hipFree(p);
}

int qnnp_hip_h2d(void* dst, const void* src, size_t bytes, int async)
{
  if (bytes == 0) return QNNP_HIP_OK;
  const hipError_t e = async ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g_rt.stream)
                             : hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  return ok(e) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int qnnp_hip_d2h(void* dst, const void* src, size_t bytes, int async)
{
  if (bytes == 0) return QNNP_HIP_OK;
  const hipError_t e = async ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, g_rt.stream)
                             : hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost);
  return ok(e) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int qnnp_hip_memset(void* dst, int value, size_t bytes)
{
  if (bytes == 0) return QNNP_HIP_OK;
  if (!ok(hipMemsetAsync(dst, value, bytes, g_rt.stream))) return QNNP_HIP_ELAUNCH;
  return ok(hipStreamSynchronize(g_rt.stream)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int qnnp_hip_is_device_pointer(const void* p)
{
  if (p == nullptr || !g_rt.bound) return 0;
  hipPointerAttribute_t attr;
  if (!ok(hipPointerGetAttributes(&attr, p))) {
    (void) hipGetLastError();  // plain host memory is "invalid value" to the runtime
    return 0;
  }
  return (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) ? 1 : 0;
}

int qnnp_hip_timer_create(void** timer)
{
  Timer* t = new (std::nothrow) Timer;
  if (t == nullptr) return QNNP_HIP_ENOMEM;
  if (!ok(hipEventCreate(&t->start)) || !ok(hipEventCreate(&t->stop))) {
    delete t;
    return QNNP_HIP_ENOMEM;
  }
  *timer = t;
  return QNNP_HIP_OK;
}

int qnnp_hip_timer_start(void* timer)
{
  Timer* t = static_cast<Timer*>(timer);
  return ok(hipEventRecord(t->start, g_rt.stream)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int qnnp_hip_timer_stop_ms(void* timer, float* ms)
{
  Timer* t = static_cast<Timer*>(timer);
  if (!ok(hipEventRecord(t->stop, g_rt.stream))) return QNNP_HIP_ELAUNCH;
  if (!ok(hipEventSynchronize(t->stop))) return QNNP_HIP_ELAUNCH;
  return ok(hipEventElapsedTime(ms, t->start, t->stop)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/* ---- hipGraph capture: a run of operator launches replayed as ONE submission (no per-launch gaps) ---- */

int qnnp_hip_graph_capturing(void) { return g_rt.capturing ? 1 : 0; }

int qnnp_hip_graph_begin(void)
{
  if (!g_rt.bound || g_rt.capturing) return QNNP_HIP_EINVAL;
  hipStream_t s = g_rt.stream;
  if (s == nullptr) {                    // the legacy default stream cannot be captured
    if (g_rt.private_stream == nullptr && !ok(hipStreamCreateWithFlags(&g_rt.private_stream, hipStreamNonBlocking))) {
      return QNNP_HIP_ENOMEM;
    }
    if (!ok(hipDeviceSynchronize())) return QNNP_HIP_ELAUNCH;   // order after everything already enqueued
    s = g_rt.private_stream;
  }
  if (!ok(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) {
    (void) hipGetLastError();
    return QNNP_HIP_ELAUNCH;
  }
  g_rt.saved_stream = g_rt.stream;
  g_rt.stream = s;
  g_rt.capturing = true;
  return QNNP_HIP_OK;
}

struct Graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipStream_t stream;                    // replay stream (the one it was captured on)
};

int qnnp_hip_graph_end(void** out)
{
  if (!g_rt.capturing || out == nullptr) return QNNP_HIP_EINVAL;
  hipStream_t s = g_rt.stream;
  g_rt.stream = g_rt.saved_stream;
  g_rt.capturing = false;
  hipGraph_t graph = nullptr;
  if (!ok(hipStreamEndCapture(s, &graph)) || graph == nullptr) {
    (void) hipGetLastError();
    return QNNP_HIP_ELAUNCH;
  }
  hipGraphExec_t exec = nullptr;
  if (!ok(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0))) {
    (void) hipGetLastError();
    (void) hipGraphDestroy(graph);
    return QNNP_HIP_ENOMEM;
  }
  Graph* g = new (std::nothrow) Graph{graph, exec, s};
  if (g == nullptr) {
    (void) hipGraphExecDestroy(exec);
    (void) hipGraphDestroy(graph);
    return QNNP_HIP_ENOMEM;
  }
  *out = g;
  return QNNP_HIP_OK;
}

int qnnp_hip_graph_launch(void* graph)
{
  Graph* g = static_cast<Graph*>(graph);
  if (g == nullptr) return QNNP_HIP_EINVAL;
  return ok(hipGraphLaunch(g->exec, g->stream)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/* average milliseconds of one replay over `iters` replays after `warmup` untimed ones (events on the replay stream) */
int qnnp_hip_graph_time(void* graph, int warmup, int iters, float* avg_ms)
{
  Graph* g = static_cast<Graph*>(graph);
  if (g == nullptr || iters <= 0 || avg_ms == nullptr) return QNNP_HIP_EINVAL;
  hipEvent_t e0, e1;
  if (!ok(hipEventCreate(&e0))) return QNNP_HIP_ENOMEM;
  if (!ok(hipEventCreate(&e1))) { (void) hipEventDestroy(e0); return QNNP_HIP_ENOMEM; }
  bool good = true;
  for (int i = 0; i < warmup && good; i++) good = ok(hipGraphLaunch(g->exec, g->stream));
  good = good && ok(hipEventRecord(e0, g->stream));
  for (int i = 0; i < iters && good; i++) good = ok(hipGraphLaunch(g->exec, g->stream));
  good = good && ok(hipEventRecord(e1, g->stream)) && ok(hipEventSynchronize(e1));
  float ms = 0.0f;
  good = good && ok(hipEventElapsedTime(&ms, e0, e1));
  (void) hipEventDestroy(e0);
  (void) hipEventDestroy(e1);
  if (!good) { (void) hipGetLastError(); return QNNP_HIP_ELAUNCH; }
  *avg_ms = ms / static_cast<float>(iters);
  return QNNP_HIP_OK;
}

int qnnp_hip_graph_sync(void* graph)
{
  Graph* g = static_cast<Graph*>(graph);
  if (g == nullptr) return QNNP_HIP_EINVAL;
  return ok(hipStreamSynchronize(g->stream)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

void qnnp_hip_graph_destroy(void* graph)
{
  Graph* g = static_cast<Graph*>(graph);
  if (g == nullptr) return;
  (void) hipGraphExecDestroy(g->exec);
  (void) hipGraphDestroy(g->graph);
  delete g;
}

void qnnp_hip_timer_destroy(void* timer)
{
  Timer* t = static_cast<Timer*>(timer);
  if (t == nullptr) return;
  (void) hipEventDestroy(t->start);
  (void) hipEventDestroy(t->stop);
  delete t;
}

}  // extern "C"
