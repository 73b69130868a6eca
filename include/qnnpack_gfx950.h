/*
 * qnnpack_gfx950.h -- MI355X-specific extensions that sit BESIDE the unchanged
 * qnnpack.h API. Nothing here is required by a drop-in caller; these entry
 * points exist for callers that own device memory and streams (frameworks,
 * bench.py, the parity tests).
 *
 * The reference has no counterpart for any of these: its only execution
 * resource is the pthreadpool argument of qnnp_run_operator
 * (include/qnnpack.h:327-329, src/operator-run.c:639), which this build ignores.
 *
 * All functions are plain C ABI: pointers and sizes only, no HIP types. A HIP
 * stream is passed as the opaque `void*` value of a hipStream_t.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "qnnpack.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Devices. The library keeps one context (launch stream, asynchrony flag) per gfx950 GPU of the node.
 *   BEFORE qnnp_initialize: names the PRIMARY device qnnp_initialize binds (default: env QNNP_GFX950_DEVICE,
 *     else the calling thread's current HIP device).
 *   AFTER qnnp_initialize: binds `device` on first use and makes it the CALLING THREAD's device: operators the
 *     thread creates from now on live there, and set_stream / set_async / synchronize / malloc / memcpy /
 *     graph_begin act on it. Threads that never call it use the primary device.
 * An operator remembers its device: setup / run / delete may be called from any thread, whatever that thread's
 * current HIP device is (it is restored on return). Tensors handed to setup must be host memory or memory of the
 * operator's device (another GPU's memory -> invalid_parameter). Batches shard across the GPUs of a node without
 * a collective -- every output pixel depends on one image (reference src/operator-run.c:675-679, 797-802,
 * 837-842) -- so both "one process per GPU" (bench.py) and "one process, one host thread per GPU" work.
 * invalid_parameter: negative ordinal, or not a usable gfx950 device. */
enum qnnp_status qnnp_gfx950_set_device(int device);
int qnnp_gfx950_get_device(void);      /* the calling thread's device, -1 before qnnp_initialize */
int qnnp_gfx950_device_count(void);    /* HIP devices visible to the process */

/* Threads: as in the reference (run contexts are stack-local, src/operator-run.c:783-795) DISTINCT operators may
 * be created, set up, run and deleted from different threads concurrently, on the same device or different ones.
 * One operator must not be used by two threads at once. set_option is configuration, read at setup time. */

/* Stream all subsequent qnnp_run_operator launches (and host-pointer staging
 * copies) on the calling thread's device are enqueued on. NULL = the device's default stream. */
enum qnnp_status qnnp_gfx950_set_stream(void* hip_stream);

/* async = 1: qnnp_run_operator only enqueues work and returns; the caller
 * synchronises (qnnp_gfx950_synchronize or its own stream sync). Requires device
 * pointers for input/output (host pointers force a synchronous staged run).
 * async = 0 (default): reference semantics, outputs complete on return.
 * A property of the calling thread's device context. */
enum qnnp_status qnnp_gfx950_set_async(int async);
enum qnnp_status qnnp_gfx950_synchronize(void);

/* Device memory helpers for C callers without HIP headers. */
void* qnnp_gfx950_malloc(size_t bytes);
void qnnp_gfx950_free(void* device_ptr);
enum qnnp_status qnnp_gfx950_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes);
enum qnnp_status qnnp_gfx950_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes);
enum qnnp_status qnnp_gfx950_memset(void* dst_device, int value, size_t bytes);

/* Time `iters` back-to-back qnnp_run_operator launches of one operator with hipEvents, after `warmup`
 * untimed launches; writes the AVERAGE milliseconds per launch. Device pointers only. By default the
 * launches are recorded into a hipGraph and its replay is timed (option "timing_graph"), so the figure is
 * kernel time -- what rocprofv3 reports per kernel -- not the host's per-launch dispatch gap. The replay is
 * timed five times, each with its own event pair, and the MEDIAN is reported. */
enum qnnp_status qnnp_gfx950_time_operator(
    qnnp_operator_t op, int warmup, int iters, float* avg_ms_out);

/* Same, but rotates through `nsets` (input, output) device buffer pairs between
 * launches so the working set exceeds the 256 MiB Infinity Cache when an HBM
 * bandwidth figure is claimed. inputs/outputs: arrays of nsets device pointers. */
enum qnnp_status qnnp_gfx950_time_operator_rotating(
    qnnp_operator_t op, size_t nsets, const void* const* inputs, void* const* outputs,
    int warmup, int iters, float* avg_ms_out);

/* hipGraph capture: between begin and end, qnnp_run_operator only RECORDS its launch (device pointers only; a
 * host-pointer operator returns invalid_parameter). The graph replays the whole sequence -- e.g. every layer of
 * a network -- as one submission: no per-launch dispatch gap, which on MI355X is as long as the small layers
 * themselves. Replays run on the stream given to qnnp_gfx950_set_stream at capture time (a private stream
 * stands in for the default stream, which cannot be captured): synchronous unless qnnp_gfx950_set_async(1),
 * then qnnp_gfx950_graph_synchronize. Operators and their buffers must outlive the graph.
 * The capture belongs to the calling THREAD (its launches are recorded, other threads keep running normally).
 * qnnp_gfx950_graph_time: milliseconds per replay, hipEvents on the replay stream -- five batches of `iters`
 * replays after `warmup` untimed ones, the median batch / iters. */
enum qnnp_status qnnp_gfx950_graph_begin(void);
enum qnnp_status qnnp_gfx950_graph_end(void** graph_out);
enum qnnp_status qnnp_gfx950_graph_launch(void* graph);
enum qnnp_status qnnp_gfx950_graph_synchronize(void* graph);
enum qnnp_status qnnp_gfx950_graph_time(void* graph, int warmup, int iters, float* avg_ms_out);
void qnnp_gfx950_graph_destroy(void* graph);

/* Fused inverted-residual block (no reference counterpart):
 *     [1x1 expand ->] 3x3 depthwise (padding 1, stride 1 | 2) -> 1x1 project [-> quantized add with the block input]
 * as ONE operator; the expanded tensors stay in LDS. Built FROM stand-alone operators created with qnnpack.h
 * (`expand` and `residual_add` may be NULL): it borrows their packed weights and quantization parameters, so the
 * result is bit-identical to running them in sequence, and they must outlive it. Run / delete with
 * qnnp_run_operator / qnnp_delete_operator. unsupported_parameter (from create or setup) = outside the fused
 * kernel's range (channel multiples, LDS) -- keep using the stand-alone operators for that block. */
enum qnnp_status qnnp_gfx950_create_fused_block(
    qnnp_operator_t expand, qnnp_operator_t depthwise, qnnp_operator_t project, qnnp_operator_t residual_add,
    qnnp_operator_t* fused);
enum qnnp_status qnnp_gfx950_setup_fused_block(
    qnnp_operator_t fused, size_t batch_size, size_t input_height, size_t input_width,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride);

/* Residual add folded into a convolution (no reference counterpart: there it is a convolution operator followed by
 * an add operator, src/add.c). After a successful qnnp_setup_convolution2d_nhwc_q8 / qnnp_setup_fully_connected_nc_q8
 * of `convolution` on DEVICE pointers, attach an add operator created with qnnp_create_add_nc_q8 whose channel count
 * is the convolution's output channel count: from then on qnnp_run_operator(convolution) writes
 *     output = add(a = residual pixel, b = convolution output pixel)
 * bit-identical to running `convolution` and then `add` set up with (a = residual, b = the convolution's output).
 * The add's parameters are copied (the add operator may be deleted); `residual` is caller-owned device memory,
 * residual_stride bytes between pixels, and may be the convolution's input tensor or any other tensor except its
 * output. The next setup of `convolution` detaches it. The add rides in the convolution kernel's epilogue where that
 * kernel carries it (pointwise layers: every MobileNetV2 project layer; needs residual_stride == output stride and
 * the output's alignment), otherwise the add kernel is launched in place behind the convolution.
 * qnnp_gfx950_operator_residual_folded: after a run, 1 = epilogue, 0 = separate launch, -1 = nothing attached.
 * NO OVERLAP: the residual range [residual, residual + (pixels - 1) * residual_stride + channels) must not intersect
 * the convolution's output range, not even exactly (residual == output): the kernels read the residual through
 * non-aliasing / streaming loads while other workgroups write the output. An in-place sum is what the stand-alone add
 * operator is for (qnnp_setup_add_nc_q8 accepts sum == a or sum == b).
 * Status: invalid_parameter (NULL / wrong operator kinds / channel mismatch / no valid setup / during capture /
 * residual overlapping the output), unsupported_parameter (host-memory endpoints -- answered before the overlap is
 * looked at, the addresses of a host-staged output mean nothing on the device). */
enum qnnp_status qnnp_gfx950_attach_residual_add(
    qnnp_operator_t convolution, qnnp_operator_t add, const uint8_t* residual, size_t residual_stride);
int qnnp_gfx950_operator_residual_folded(qnnp_operator_t op);

/* Process-wide options of the product. Keys:
 *   "timing_graph":  1 (default) = qnnp_gfx950_time_operator* time a hipGraph replay of the launches (kernel
 *                    time without per-launch dispatch gaps); 0 = a plain back-to-back launch loop
 *   "streaming_stores": 1 (default) = kernels whose waves write whole cache lines exactly once (pointwise / fully
 *                    connected outputs, the 4096^3-class GEMM, the element-wise add) mark those stores as streaming:
 *                    right for an operator that runs on its own (per-layer sweep +4-5 %, GEMM +2 %); 0 = plain stores,
 *                    for callers that chain operators -- a streamed tensor is not in the last-level cache when its
 *                    consumer starts (whole MobileNetV2: -1 % with the hint). Read at launch (or graph-capture) time.
 *                    This is the DEFAULT of every operator; qnnp_gfx950_operator_set_streaming_stores below sets it per
 *                    operator, which is what a caller with both kinds of operators in one process wants.
 * Unknown key -> invalid_parameter.
 * (Kernel-forcing codes -- which kernel an operator runs on, for A/B measurement and for the test tiers -- are NOT part of
 *  this interface: include/qnnpack_gfx950_test.h, qnnp_gfx950_test_force_kernel.) */
enum qnnp_status qnnp_gfx950_set_option(const char* key, int value);

/* The streaming-store hint ("streaming_stores" above) of ONE operator: value 1 / 0 = on / off for every later launch of
 * `op`, -1 = follow the process-wide option again (the default). An operator whose output the next operator reads at
 * once (a chained network) wants 0; an operator on its own, as the reference bench runs them, 1 -- both kinds can live
 * in one process, and no launch of another thread is affected.
 * Honoured by every kernel that has a streaming form: convolution / fully connected / depthwise operators, the
 * element-wise add, and deconvolutions that run as GEMMs (kernel == stride, and the per-phase GEMMs). Kernels WITHOUT a
 * streaming form write plain stores whatever the setting, and the call still answers success for them: the stride-2
 * 3x3 / 4x4 deconvolution stream kernel, global average pooling (its output is a few KB) and fused blocks (their output
 * feeds the next block). */
enum qnnp_status qnnp_gfx950_operator_set_streaming_stores(qnnp_operator_t op, int value);

/* Name of the HIP kernel the operator's last setup selected (static string), or
 * NULL. Lets tests assert that the intended kernel actually ran. */
const char* qnnp_gfx950_operator_kernel(qnnp_operator_t op);

/* Device properties as seen by the library: gcnArchName copied into `arch`
 * (NUL-terminated, truncated to arch_len), CU count, clock in kHz. */
enum qnnp_status qnnp_gfx950_device_info(
    char* arch, size_t arch_len, int* compute_units, int* clock_khz, size_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
