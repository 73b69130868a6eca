/*
 * runtime.hip -- device contexts, memory, stream and event plumbing behind the
 * C-ABI seam of qnnp_hip.h. The reference has no counterpart (its runtime is
 * cpuinfo + pthreadpool, src/init.c:244-263); this is what "initialize" means
 * on an MI355X node.
 *
 * One context per gfx950 device of the node (stream, asynchrony flag, padding
 * fill table, device properties). qnnp_initialize binds the PRIMARY device;
 * further devices are bound on demand (qnnp_gfx950_set_device after init), so a
 * process may drive all 8 GPUs of a node from one thread per device, or one
 * process per GPU -- the batch shards without a collective either way
 * (reference src/operator-run.c:675-679, 797-802, 837-842: batch is an
 * independent grid dimension).
 *
 * Threading (the reference's contract, src/operator-run.c:783-795: run contexts
 * are stack-local, distinct operators may run from different threads): nothing
 * here is per-process mutable state on the launch path. A thread's SELECTED
 * device, its ACTIVE context (entered through an operator) and its hipGraph
 * capture state are thread-local; a context's stream / async flag are atomics.
 */
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "qnnp_hip.h"

namespace {

constexpr int kMaxDevices = 16;

struct DeviceCtx {
  std::atomic<bool> bound{false};
  int device = -1;
  std::atomic<hipStream_t> stream{nullptr};   // nullptr = the device's default stream
  std::atomic<int> async{0};
  hipDeviceProp_t props;
  uint8_t* fill_table = nullptr;  // [256][16]: entry v = sixteen bytes of value v (LDS-DMA padding sources)
};

DeviceCtx g_dev[kMaxDevices];
std::mutex g_bind_lock;
std::atomic<int> g_primary{-1};
// Initialization generation: bumped by every shutdown. Thread-local selections and capture state remember the
// generation they were made in; a stale one (another thread deinitialized and re-initialized the library since)
// reads as "none", so that thread falls back to the primary device instead of to a context that no longer exists.
std::atomic<uint32_t> g_generation{1};

// per-thread state
thread_local int t_selected = -1;        // device chosen with qnnp_hip_select (-1: the primary device)
thread_local uint32_t t_selected_gen = 0;
thread_local int t_active = -1;          // context entered through qnnp_hip_enter (-1: the selected device)
struct Capture {
  bool on = false;
  int device = -1;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  uint32_t generation = 0;
};
thread_local Capture t_cap;

inline bool capture_live()
{
  return t_cap.on && t_cap.generation == g_generation.load(std::memory_order_acquire);
}

struct Timer {
  hipEvent_t start;
  hipEvent_t stop;
};

inline bool ok(hipError_t e) { return e == hipSuccess; }

inline int current_index()
{
  if (t_active >= 0) return t_active;
  if (t_selected >= 0 && t_selected_gen == g_generation.load(std::memory_order_acquire)) return t_selected;
  return g_primary.load(std::memory_order_acquire);
}

inline DeviceCtx* ctx()
{
  const int d = current_index();
  if (d < 0 || d >= kMaxDevices || !g_dev[d].bound.load(std::memory_order_acquire)) return nullptr;
  return &g_dev[d];
}

// the stream launches of the calling thread go to: its capture stream while it records a graph on this device
inline hipStream_t launch_stream(const DeviceCtx* c)
{
  if (capture_live() && t_cap.device == c->device) return t_cap.stream;
  return c->stream.load(std::memory_order_acquire);
}

// the calling thread is recording a hipGraph on THIS context's device
inline bool capturing_on(const DeviceCtx* c)
{
  return capture_live() && t_cap.device == c->device;
}

int bind_locked(int device)
{
  int count = 0;
  if (!ok(hipGetDeviceCount(&count)) || count <= 0) {
    (void) hipGetLastError();
    return QNNP_HIP_ENODEV;
  }
  if (device < 0 || device >= count || device >= kMaxDevices) return QNNP_HIP_ENODEV;
  DeviceCtx& c = g_dev[device];
  if (c.bound.load(std::memory_order_acquire)) return QNNP_HIP_OK;
  int previous = -1;
  (void) hipGetDevice(&previous);
  if (!ok(hipSetDevice(device))) return QNNP_HIP_ENODEV;
  int rc = QNNP_HIP_OK;
  if (!ok(hipGetDeviceProperties(&c.props, device))) {
    rc = QNNP_HIP_ENODEV;
  } else if (std::strncmp(c.props.gcnArchName, "gfx950", 6) != 0) {
    // Only CDNA4: the kernels use v_mfma_i32_32x32x32_i8 and are built for gfx950 alone.
    rc = QNNP_HIP_ENODEV;
  } else if (c.fill_table == nullptr) {
    uint8_t host[256 * 16];
    for (int v = 0; v < 256; v++) std::memset(host + v * 16, v, 16);
    if (!ok(hipMalloc(reinterpret_cast<void**>(&c.fill_table), sizeof(host)))) {
      rc = QNNP_HIP_ENOMEM;
    } else if (!ok(hipMemcpy(c.fill_table, host, sizeof(host), hipMemcpyHostToDevice))) {
      (void) hipFree(c.fill_table);
      c.fill_table = nullptr;
      rc = QNNP_HIP_ENOMEM;
    }
  }
  if (rc == QNNP_HIP_OK) {
    c.device = device;
    c.stream.store(nullptr);
    c.async.store(0);
    c.bound.store(true, std::memory_order_release);
  } else {
    (void) hipGetLastError();
  }
  if (previous >= 0 && previous != device) (void) hipSetDevice(previous);
  return rc;
}

}  // namespace

extern "C" {

int qnnp_hip_init(int device)
{
  std::lock_guard<std::mutex> guard(g_bind_lock);
  if (device < 0) {
    if (!ok(hipGetDevice(&device))) {
      (void) hipGetLastError();
      device = 0;
    }
  }
  const int rc = bind_locked(device);
  if (rc != QNNP_HIP_OK) return rc;
  g_primary.store(device, std::memory_order_release);
  return QNNP_HIP_OK;
}

int qnnp_hip_bind(int device)
{
  std::lock_guard<std::mutex> guard(g_bind_lock);
  if (g_primary.load() < 0) return QNNP_HIP_ENODEV;   // not initialized
  return bind_locked(device);
}

int qnnp_hip_shutdown(void)
{
  std::lock_guard<std::mutex> guard(g_bind_lock);
  int previous = -1;
  (void) hipGetDevice(&previous);
  for (int d = 0; d < kMaxDevices; d++) {
    DeviceCtx& c = g_dev[d];
    if (!c.bound.load()) continue;
    c.bound.store(false, std::memory_order_release);
    if (c.fill_table != nullptr) {
      (void) hipSetDevice(d);
      (void) hipFree(c.fill_table);
      c.fill_table = nullptr;
    }
    c.stream.store(nullptr);
    c.async.store(0);
  }
  if (previous >= 0) (void) hipSetDevice(previous);
  g_primary.store(-1, std::memory_order_release);
  g_generation.fetch_add(1, std::memory_order_acq_rel);   // every thread's selection / capture state is now stale
  t_selected = -1;
  t_active = -1;
  t_cap = Capture();
  return QNNP_HIP_OK;
}

int qnnp_hip_device_count(void)
{
  int count = 0;
  if (!ok(hipGetDeviceCount(&count))) {
    (void) hipGetLastError();
    return 0;
  }
  return count;
}

int qnnp_hip_select(int device)
{
  if (device < 0 || device >= kMaxDevices || !g_dev[device].bound.load(std::memory_order_acquire)) return QNNP_HIP_ENODEV;
  t_selected = device;
  t_selected_gen = g_generation.load(std::memory_order_acquire);
  return QNNP_HIP_OK;
}

int qnnp_hip_device(void)
{
  const DeviceCtx* c = ctx();
  return c != nullptr ? c->device : -1;
}

/* Make `device`'s context the calling thread's active one and its HIP device current. The token restores both. */
int qnnp_hip_enter(int device)
{
  if (device < 0 || device >= kMaxDevices || !g_dev[device].bound.load(std::memory_order_acquire)) return -1;
  int hip_prev = -1;
  if (!ok(hipGetDevice(&hip_prev))) {
    (void) hipGetLastError();
    hip_prev = -1;
  }
  int restore = 0;                       // 0: the HIP device was already right
  if (hip_prev != device) {
    if (!ok(hipSetDevice(device))) {
      (void) hipGetLastError();
      return -1;
    }
    restore = hip_prev + 1;
  }
  const int token = ((t_active + 1) << 8) | restore;
  t_active = device;
  return token;
}

void qnnp_hip_leave(int token)
{
  if (token < 0) return;
  t_active = (token >> 8) - 1;
  const int restore = token & 0xFF;
  if (restore != 0) (void) hipSetDevice(restore - 1);
}

int qnnp_hip_device_info(char* arch, size_t arch_len, int* cus, int* clock_khz, size_t* mem_bytes)
{
  const DeviceCtx* c = ctx();
  if (c == nullptr) return QNNP_HIP_ENODEV;
  if (arch != nullptr && arch_len > 0) {
    std::strncpy(arch, c->props.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = '\0';
  }
  if (cus != nullptr) *cus = c->props.multiProcessorCount;
  if (clock_khz != nullptr) *clock_khz = c->props.clockRate;
  if (mem_bytes != nullptr) *mem_bytes = c->props.totalGlobalMem;
  return QNNP_HIP_OK;
}

int qnnp_hip_compute_units(void)
{
  const DeviceCtx* c = ctx();
  return c != nullptr ? c->props.multiProcessorCount : 0;
}

const uint8_t* qnnp_hip_fill_table(void)
{
  const DeviceCtx* c = ctx();
  return c != nullptr ? c->fill_table : nullptr;
}

/* "streaming_stores" (qnnp_gfx950_set_option): 1 (default) = kernels that write whole lines exactly once mark them
 * as streaming; 0 = plain stores, for callers that chain operators and want each output cacheable for its consumer */
static std::atomic<int> g_streaming_stores{1};
void qnnp_hip_set_streaming_stores(int on) { g_streaming_stores.store(on != 0 ? 1 : 0, std::memory_order_relaxed); }
int qnnp_hip_streaming_stores(void) { return g_streaming_stores.load(std::memory_order_relaxed); }

void qnnp_hip_set_stream(void* stream)
{
  DeviceCtx* c = ctx();
  if (c != nullptr) c->stream.store(reinterpret_cast<hipStream_t>(stream), std::memory_order_release);
}

void* qnnp_hip_get_stream(void)
{
  const DeviceCtx* c = ctx();
  return c != nullptr ? reinterpret_cast<void*>(launch_stream(c)) : nullptr;
}

void qnnp_hip_set_async(int async)
{
  DeviceCtx* c = ctx();
  if (c != nullptr) c->async.store(async != 0 ? 1 : 0, std::memory_order_release);
}

int qnnp_hip_get_async(void)
{
  const DeviceCtx* c = ctx();
  return c != nullptr ? c->async.load(std::memory_order_acquire) : 0;
}

int qnnp_hip_stream_sync(void)
{
  const DeviceCtx* c = ctx();
  if (c == nullptr) return QNNP_HIP_ENODEV;
  return ok(hipStreamSynchronize(launch_stream(c))) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

void* qnnp_hip_alloc(size_t bytes)
{
  if (ctx() == nullptr) return nullptr;
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

/* async = 0: the copy is complete on return AND ordered behind everything already enqueued on the library
 * stream (a blocking null-stream copy would not be, for a non-blocking library stream: a table re-uploaded
 * by setup could land under a kernel of the previous run that is still in flight).
 * While the calling thread records a hipGraph on this device every copy / fill is REFUSED: it would become a graph
 * node whose host source (a table setup frees right afterwards) dangles at replay, and the synchronizing forms
 * would invalidate the capture. create / setup / the memcpy helpers therefore fail with invalid_parameter inside
 * qnnp_gfx950_graph_begin ... graph_end; only operator launches on device pointers are recordable. */
int qnnp_hip_h2d(void* dst, const void* src, size_t bytes, int async)
{
  if (bytes == 0) return QNNP_HIP_OK;
  const DeviceCtx* c = ctx();
  if (c == nullptr) return QNNP_HIP_ENODEV;
  if (capturing_on(c)) return QNNP_HIP_EINVAL;
  hipStream_t s = c->stream.load(std::memory_order_acquire);
  if (!ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s))) {
    (void) hipGetLastError();
    return QNNP_HIP_ELAUNCH;
  }
  if (!async) return ok(hipStreamSynchronize(s)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  return QNNP_HIP_OK;
}

int qnnp_hip_d2h(void* dst, const void* src, size_t bytes, int async)
{
  if (bytes == 0) return QNNP_HIP_OK;
  const DeviceCtx* c = ctx();
  if (c == nullptr) return QNNP_HIP_ENODEV;
  if (capturing_on(c)) return QNNP_HIP_EINVAL;
  hipStream_t s = c->stream.load(std::memory_order_acquire);
  if (!ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s))) {
    (void) hipGetLastError();
    return QNNP_HIP_ELAUNCH;
  }
  if (!async) return ok(hipStreamSynchronize(s)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  return QNNP_HIP_OK;
}

int qnnp_hip_memset(void* dst, int value, size_t bytes)
{
  if (bytes == 0) return QNNP_HIP_OK;
  const DeviceCtx* c = ctx();
  if (c == nullptr) return QNNP_HIP_ENODEV;
  if (capturing_on(c)) return QNNP_HIP_EINVAL;
  hipStream_t s = c->stream.load(std::memory_order_acquire);
  if (!ok(hipMemsetAsync(dst, value, bytes, s))) return QNNP_HIP_ELAUNCH;
  return ok(hipStreamSynchronize(s)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/* 1 = memory of the active device (or managed), 0 = host memory, -1 = memory of ANOTHER device */
int qnnp_hip_is_device_pointer(const void* p)
{
  const DeviceCtx* c = ctx();
  if (p == nullptr || c == nullptr) return 0;
  hipPointerAttribute_t attr;
  if (!ok(hipPointerGetAttributes(&attr, p))) {
    (void) hipGetLastError();  // plain host memory is "invalid value" to the runtime
    return 0;
  }
  if (attr.type == hipMemoryTypeManaged) return 1;
  if (attr.type == hipMemoryTypeDevice) return attr.device == c->device ? 1 : -1;
  return 0;
}

int qnnp_hip_timer_create(void** timer)
{
  Timer* t = new (std::nothrow) Timer;
  if (t == nullptr) return QNNP_HIP_ENOMEM;
  if (!ok(hipEventCreate(&t->start))) {
    delete t;
    return QNNP_HIP_ENOMEM;
  }
  if (!ok(hipEventCreate(&t->stop))) {
    (void) hipEventDestroy(t->start);
    delete t;
    return QNNP_HIP_ENOMEM;
  }
  *timer = t;
  return QNNP_HIP_OK;
}

int qnnp_hip_timer_start(void* timer)
{
  Timer* t = static_cast<Timer*>(timer);
  const DeviceCtx* c = ctx();
  if (c == nullptr) return QNNP_HIP_ENODEV;
  return ok(hipEventRecord(t->start, launch_stream(c))) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int qnnp_hip_timer_stop_ms(void* timer, float* ms)
{
  Timer* t = static_cast<Timer*>(timer);
  const DeviceCtx* c = ctx();
  if (c == nullptr) return QNNP_HIP_ENODEV;
  if (!ok(hipEventRecord(t->stop, launch_stream(c)))) return QNNP_HIP_ELAUNCH;
  if (!ok(hipEventSynchronize(t->stop))) return QNNP_HIP_ELAUNCH;
  return ok(hipEventElapsedTime(ms, t->start, t->stop)) ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

void qnnp_hip_timer_destroy(void* timer)
{
  Timer* t = static_cast<Timer*>(timer);
  if (t == nullptr) return;
  (void) hipEventDestroy(t->start);
  (void) hipEventDestroy(t->stop);
  delete t;
}

/* ---- hipGraph capture: a run of operator launches replayed as ONE submission (no per-launch gaps) ----
 * Capture state belongs to the capturing THREAD: its launches on that device go to the capture stream, every
 * other thread keeps launching on the context's stream. */

/* 1 only while the calling thread records on the ACTIVE context's device: an operator of another GPU run from a
 * thread that captures on GPU A is launched live on its own device's stream (launch_stream), so its caller must take
 * the live path too -- synchronize as usual, never report "recorded". */
int qnnp_hip_graph_capturing(void)
{
  const DeviceCtx* c = ctx();
  return c != nullptr && capturing_on(c) ? 1 : 0;
}

int qnnp_hip_graph_begin(void)
{
  const DeviceCtx* c = ctx();
  if (c == nullptr || capture_live()) return QNNP_HIP_EINVAL;
  hipStream_t s = c->stream.load(std::memory_order_acquire);
  bool owns = false;
  if (s == nullptr) {                    // the legacy default stream cannot be captured: the graph gets its own
    if (!ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking))) {
      (void) hipGetLastError();
      return QNNP_HIP_ENOMEM;
    }
    owns = true;
    if (!ok(hipDeviceSynchronize())) {   // order after everything already enqueued
      (void) hipStreamDestroy(s);
      return QNNP_HIP_ELAUNCH;
    }
  }
  if (!ok(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal))) {
    (void) hipGetLastError();
    if (owns) (void) hipStreamDestroy(s);
    return QNNP_HIP_ELAUNCH;
  }
  t_cap.on = true;
  t_cap.generation = g_generation.load(std::memory_order_acquire);
  t_cap.device = c->device;
  t_cap.stream = s;
  t_cap.owns_stream = owns;
  return QNNP_HIP_OK;
}

struct Graph {
  hipGraph_t graph;
  hipGraphExec_t exec;
  hipStream_t stream;                    // replay stream (the one it was captured on)
  bool owns_stream;
  int device;
};

int qnnp_hip_graph_end(void** out)
{
  if (!capture_live() || out == nullptr) return QNNP_HIP_EINVAL;
  const Capture cap = t_cap;
  t_cap = Capture();
  hipGraph_t graph = nullptr;
  const int token = qnnp_hip_enter(cap.device);
  int rc = QNNP_HIP_OK;
  hipGraphExec_t exec = nullptr;
  if (!ok(hipStreamEndCapture(cap.stream, &graph)) || graph == nullptr) {
    (void) hipGetLastError();
    rc = QNNP_HIP_ELAUNCH;
  } else if (!ok(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0))) {
    (void) hipGetLastError();
    (void) hipGraphDestroy(graph);
    rc = QNNP_HIP_ENOMEM;
  } else {
    Graph* g = new (std::nothrow) Graph{graph, exec, cap.stream, cap.owns_stream, cap.device};
    if (g == nullptr) {
      (void) hipGraphExecDestroy(exec);
      (void) hipGraphDestroy(graph);
      rc = QNNP_HIP_ENOMEM;
    } else {
      *out = g;
    }
  }
  if (rc != QNNP_HIP_OK && cap.owns_stream) (void) hipStreamDestroy(cap.stream);
  qnnp_hip_leave(token);
  return rc;
}

int qnnp_hip_graph_device(void* graph)
{
  Graph* g = static_cast<Graph*>(graph);
  return g != nullptr ? g->device : -1;
}

int qnnp_hip_graph_launch(void* graph)
{
  Graph* g = static_cast<Graph*>(graph);
  if (g == nullptr) return QNNP_HIP_EINVAL;
  const int token = qnnp_hip_enter(g->device);
  const bool good = ok(hipGraphLaunch(g->exec, g->stream));
  qnnp_hip_leave(token);
  return good ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/* Milliseconds of one replay: `samples` event-bracketed batches of `iters` back-to-back replays each after
 * `warmup` untimed replays; *avg_ms = the MEDIAN batch / iters. */
int qnnp_hip_graph_time_median(void* graph, int warmup, int iters, int samples, float* avg_ms)
{
  Graph* g = static_cast<Graph*>(graph);
  if (g == nullptr || iters <= 0 || samples <= 0 || samples > 64 || avg_ms == nullptr) return QNNP_HIP_EINVAL;
  const int token = qnnp_hip_enter(g->device);
  hipEvent_t e0, e1;
  if (!ok(hipEventCreate(&e0))) { qnnp_hip_leave(token); return QNNP_HIP_ENOMEM; }
  if (!ok(hipEventCreate(&e1))) { (void) hipEventDestroy(e0); qnnp_hip_leave(token); return QNNP_HIP_ENOMEM; }
  bool good = true;
  for (int i = 0; i < warmup && good; i++) good = ok(hipGraphLaunch(g->exec, g->stream));
  float batch_ms[64];
  for (int s = 0; s < samples && good; s++) {
    good = good && ok(hipEventRecord(e0, g->stream));
    for (int i = 0; i < iters && good; i++) good = ok(hipGraphLaunch(g->exec, g->stream));
    good = good && ok(hipEventRecord(e1, g->stream)) && ok(hipEventSynchronize(e1));
    float ms = 0.0f;
    good = good && ok(hipEventElapsedTime(&ms, e0, e1));
    batch_ms[s] = ms;
  }
  (void) hipEventDestroy(e0);
  (void) hipEventDestroy(e1);
  qnnp_hip_leave(token);
  if (!good) { (void) hipGetLastError(); return QNNP_HIP_ELAUNCH; }
  for (int i = 1; i < samples; i++) {          // insertion sort, <= 64 entries
    const float v = batch_ms[i];
    int j = i - 1;
    while (j >= 0 && batch_ms[j] > v) { batch_ms[j + 1] = batch_ms[j]; j--; }
    batch_ms[j + 1] = v;
  }
  const float median = (samples & 1) ? batch_ms[samples / 2] : 0.5f * (batch_ms[samples / 2 - 1] + batch_ms[samples / 2]);
  *avg_ms = median / static_cast<float>(iters);
  return QNNP_HIP_OK;
}

int qnnp_hip_graph_time(void* graph, int warmup, int iters, float* avg_ms)
{
  return qnnp_hip_graph_time_median(graph, warmup, iters, 1, avg_ms);
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
  const int token = qnnp_hip_enter(g->device);
  (void) hipStreamSynchronize(g->stream);
  (void) hipGraphExecDestroy(g->exec);
  (void) hipGraphDestroy(g->graph);
  if (g->owns_stream) (void) hipStreamDestroy(g->stream);
  qnnp_hip_leave(token);
  delete g;
}

#ifdef QNNP_ENABLE_ABLATION
/* measurement builds only: device buffer for in-kernel cycle stamps */
static unsigned long long* g_trace = nullptr;
static const size_t kTraceWords = 4096 * 4 * 8;
void* qnnp_hip_trace_buffer(void)
{
  static const bool enabled = getenv("QNNP_GFX950_TRACE") != nullptr;
  if (g_trace == nullptr && enabled) {
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
#endif

}  // extern "C"
