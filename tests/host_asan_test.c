/*
 * host_asan_test.c -- TEST INFRASTRUCTURE: drives the product's plain-C host code (compiled with
 * -fsanitize=address,undefined against tests/hip_stub.c) through the public API of include/qnnpack.h the way the
 * reference's operator testers do (test/convolution-operator-tester.h:415-449): create -> setup -> run -> re-setup with
 * another geometry -> run -> delete, over the operator types and the shapes that take different packing / table paths.
 * Any heap overflow, use-after-free, leak or undefined arithmetic in the packers, offset tables, phase splitting,
 * staging logic or error paths aborts the program. Prints "host-sanitizers-ok" on success.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <qnnpack.h>
#include <qnnpack_gfx950.h>

static uint32_t rng_state = 0x1234567u;
static uint8_t rnd8(void) { rng_state = rng_state * 1664525u + 1013904223u; return (uint8_t) (rng_state >> 24); }

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "%s:%d: check failed: %s\n", __FILE__, __LINE__, #cond); exit(1); } } while (0)

static uint8_t* random_bytes(size_t n)
{
  uint8_t* p = (uint8_t*) malloc(n ? n : 1);
  CHECK(p != NULL);
  for (size_t i = 0; i < n; i++) p[i] = rnd8();
  return p;
}

static int32_t* random_bias(size_t n)
{
  int32_t* p = (int32_t*) malloc(sizeof(int32_t) * (n ? n : 1));
  CHECK(p != NULL);
  for (size_t i = 0; i < n; i++) p[i] = (int32_t) (rnd8() * 79) - 10000;
  return p;
}

static size_t out_dim(size_t in, uint32_t pad, uint32_t k, uint32_t d, uint32_t s)
{
  const size_t eff = (size_t) (k - 1) * d + 1;
  return (in + pad - eff) / s + 1;
}

static void conv_case(uint32_t pad, uint32_t kh, uint32_t kw, uint32_t stride, uint32_t dil, uint32_t groups,
                      size_t gic, size_t goc, size_t batch, size_t h, size_t w, size_t extra_stride)
{
  const size_t cin = groups * gic, cout = groups * goc;
  uint8_t* kernel = random_bytes((size_t) groups * goc * kh * kw * gic);
  int32_t* bias = random_bias(cout);
  qnnp_operator_t op = NULL;
  CHECK(qnnp_create_convolution2d_nhwc_q8(pad, pad, pad, pad, kh, kw, stride, stride, dil, dil, groups, gic, goc,
      127, 0.5f, 121, 0.5f, kernel, bias, 130, 0.75f, 3, 250, 0, &op) == qnnp_status_success);
  free(kernel);   /* reference ownership: create copied (packed) them */
  free(bias);
  CHECK(qnnp_run_operator(op, NULL) == qnnp_status_invalid_parameter);   /* before setup */
  for (int round = 0; round < 3; round++) {
    const size_t hh = h + (size_t) round * 3, ww = w + (size_t) round;
    const size_t in_stride = cin + extra_stride, out_stride = cout + extra_stride;
    const size_t oh = out_dim(hh, 2 * pad, kh, dil, stride), ow = out_dim(ww, 2 * pad, kw, dil, stride);
    uint8_t* in = random_bytes((batch * hh * ww - 1) * in_stride + cin);
    uint8_t* out = random_bytes((batch * oh * ow - 1) * out_stride + cout);
    CHECK(qnnp_setup_convolution2d_nhwc_q8(op, batch, hh, ww, in, in_stride, out, out_stride, NULL) == qnnp_status_success);
    CHECK(qnnp_run_operator(op, NULL) == qnnp_status_success);
    /* invalid geometry must not disturb the operator */
    CHECK(qnnp_setup_convolution2d_nhwc_q8(op, batch, 0, ww, in, in_stride, out, out_stride, NULL) == qnnp_status_invalid_parameter);
    CHECK(qnnp_run_operator(op, NULL) == qnnp_status_success);
    free(in);
    free(out);
  }
  CHECK(qnnp_setup_convolution2d_nhwc_q8(op, 0, 5, 5, NULL, cin, NULL, cout, NULL) == qnnp_status_success);
  CHECK(qnnp_run_operator(op, NULL) == qnnp_status_success);   /* empty batch */
  CHECK(qnnp_delete_operator(op) == qnnp_status_success);
}

static void deconv_case(uint32_t pad, uint32_t adj, uint32_t k, uint32_t stride, uint32_t groups, size_t gic, size_t goc,
                        size_t batch, size_t h, size_t w)
{
  uint8_t* kernel = random_bytes((size_t) groups * gic * k * k * goc);
  int32_t* bias = random_bias(groups * goc);
  qnnp_operator_t op = NULL;
  CHECK(qnnp_create_deconvolution2d_nhwc_q8(pad, pad, pad, pad, adj, adj, k, k, stride, stride, 1, 1, groups, gic, goc,
      127, 0.5f, 127, 0.5f, kernel, bias, 127, 0.5f, 0, 255, 0, &op) == qnnp_status_success);
  free(kernel);
  free(bias);
  for (int round = 0; round < 2; round++) {
    const size_t hh = h + (size_t) round * 2, ww = w + (size_t) round * 3;
    const size_t oh = stride * (hh - 1) + adj + k - 2 * pad, ow = stride * (ww - 1) + adj + k - 2 * pad;
    uint8_t* in = random_bytes(batch * hh * ww * groups * gic);
    uint8_t* out = random_bytes(batch * oh * ow * groups * goc);
    CHECK(qnnp_setup_deconvolution2d_nhwc_q8(op, batch, hh, ww, in, groups * gic, out, groups * goc, NULL) == qnnp_status_success);
    CHECK(qnnp_run_operator(op, NULL) == qnnp_status_success);
    free(in);
    free(out);
  }
  CHECK(qnnp_delete_operator(op) == qnnp_status_success);
}

int main(void)
{
  qnnp_operator_t op = NULL;
  uint8_t k4[16] = {0};
  int32_t b4[4] = {0};
  /* before initialization everything answers uninitialized (reference convolution.c:69-72) */
  CHECK(qnnp_create_fully_connected_nc_q8(4, 4, 0, 1.0f, 0, 1.0f, k4, b4, 0, 2.0f, 0, 255, 0, &op) == qnnp_status_uninitialized);
  CHECK(qnnp_initialize() == qnnp_status_success);
  CHECK(qnnp_initialize() == qnnp_status_success);
  CHECK(qnnp_gfx950_set_device(0) == qnnp_status_success);
  CHECK(qnnp_gfx950_set_device(3) == qnnp_status_invalid_parameter);

  /* parameter validation (reference convolution.c:74-176) */
  CHECK(qnnp_create_convolution2d_nhwc_q8(0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 4, 4, 0, 1.0f, 0, 1.0f, k4, b4, 0, 2.0f, 0, 255, 0, &op) == qnnp_status_invalid_parameter);
  CHECK(qnnp_create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 4, 4, 0, 1.0f, 0, 1.0f, k4, b4, 0, 0.5f, 0, 255, 0, &op) == qnnp_status_unsupported_parameter);
  CHECK(qnnp_delete_operator(NULL) == qnnp_status_invalid_parameter);
  CHECK(qnnp_run_operator(NULL, NULL) == qnnp_status_invalid_parameter);

  /*        pad kh kw s  d  groups gic goc batch h   w  extra-stride */
  conv_case(0, 1, 1, 1, 1, 1,    23, 19, 2,   7,  9, 0);    /* pointwise GEMM, ragged channels */
  conv_case(0, 1, 1, 1, 1, 2,    17, 19, 1,   6,  5, 3);    /* grouped, padded pixel strides */
  conv_case(1, 3, 3, 1, 1, 1,    15, 17, 3,   10, 9, 0);    /* offset table */
  conv_case(1, 3, 3, 2, 1, 1,     3, 32, 2,   17, 19, 2);   /* 3-channel first layer (4-wide tap slots) */
  conv_case(2, 3, 3, 1, 2, 2,    14, 13, 1,   11, 12, 0);   /* dilated, grouped */
  conv_case(1, 3, 3, 1, 1, 27,    1,  1, 2,   15, 14, 0);   /* depthwise, ragged channels */
  conv_case(1, 3, 3, 2, 1, 96,    1,  1, 2,   15, 14, 0);   /* depthwise, matrix-core weight image */
  conv_case(2, 5, 5, 1, 1, 40,    1,  1, 1,   12, 13, 8);   /* depthwise 5x5 */
  conv_case(1, 3, 3, 1, 1, 1,    64, 64, 2,   14, 14, 0);   /* power-of-two channels (LDS-tiled kernel image) */

  /*          pad adj k stride groups gic goc batch h  w */
  deconv_case(0,  0,  2, 2,    1,     16, 8,  2,    5, 6);  /* kernel == stride: depth-to-space GEMM */
  deconv_case(1,  1,  3, 2,    1,     8,  12, 2,    5, 4);  /* phase split */
  deconv_case(1,  0,  3, 1,    2,     6,  5,  1,    6, 7);  /* single table, grouped */
  deconv_case(0,  1,  4, 3,    1,     4,  4,  1,    3, 3);  /* 9 phases, some without taps */

  /* fully connected */
  for (int round = 0; round < 2; round++) {
    const size_t kc = round ? 1024 : 37, n = round ? 1000 : 29, batch = round ? 1 : 5;
    uint8_t* kernel = random_bytes(n * kc);
    int32_t* bias = random_bias(n);
    CHECK(qnnp_create_fully_connected_nc_q8(kc, n, 127, 0.5f, 127, 0.5f, kernel, bias, 127, 0.5f, 0, 255, 0, &op) == qnnp_status_success);
    free(kernel);
    free(bias);
    uint8_t* in = random_bytes(batch * (kc + 3));
    uint8_t* out = random_bytes(batch * (n + 5));
    CHECK(qnnp_setup_fully_connected_nc_q8(op, batch, in, kc + 3, out, n + 5) == qnnp_status_success);
    CHECK(qnnp_run_operator(op, NULL) == qnnp_status_success);
    float ms = 0.0f;
    CHECK(qnnp_gfx950_time_operator(op, 1, 3, &ms) == qnnp_status_invalid_parameter);   /* host tensors cannot be timed */
    free(in);
    free(out);
    CHECK(qnnp_delete_operator(op) == qnnp_status_success);
  }

  /* add + global average pooling */
  {
    CHECK(qnnp_create_add_nc_q8(24, 121, 0.75f, 127, 1.25f, 133, 0.96875f, 0, 255, 0, &op) == qnnp_status_success);
    uint8_t* a = random_bytes(9 * 31), * b = random_bytes(9 * 29), * s = random_bytes(9 * 24);
    CHECK(qnnp_setup_add_nc_q8(op, 9, a, 31, b, 29, s, 24) == qnnp_status_success);
    CHECK(qnnp_run_operator(op, NULL) == qnnp_status_success);
    CHECK(qnnp_setup_add_nc_q8(op, 9, a, 3, b, 29, s, 24) == qnnp_status_invalid_parameter);   /* stride < channels */
    free(a); free(b); free(s);
    CHECK(qnnp_delete_operator(op) == qnnp_status_success);
    CHECK(qnnp_create_global_average_pooling_nwc_q8(77, 121, 1.0f, 133, 1.0f, 0, 255, 0, &op) == qnnp_status_success);
    uint8_t* x = random_bytes(3 * 49 * 80), * y = random_bytes(3 * 77);
    CHECK(qnnp_setup_global_average_pooling_nwc_q8(op, 3, 49, x, 80, y, 77) == qnnp_status_success);
    CHECK(qnnp_run_operator(op, NULL) == qnnp_status_success);
    free(x); free(y);
    CHECK(qnnp_delete_operator(op) == qnnp_status_success);
  }

  /* fused block built from stand-alone operators */
  {
    qnnp_operator_t expand = NULL, dw = NULL, project = NULL, add = NULL, fused = NULL;
    uint8_t* ke = random_bytes(96 * 16); int32_t* be = random_bias(96);
    uint8_t* kd = random_bytes(96 * 9); int32_t* bd = random_bias(96);
    uint8_t* kp = random_bytes(16 * 96); int32_t* bp = random_bias(16);
    CHECK(qnnp_create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 16, 96, 127, 0.5f, 127, 0.5f, ke, be, 127, 0.5f, 0, 255, 0, &expand) == qnnp_status_success);
    CHECK(qnnp_create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 96, 1, 1, 127, 0.5f, 127, 0.5f, kd, bd, 127, 0.5f, 0, 255, 0, &dw) == qnnp_status_success);
    CHECK(qnnp_create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 96, 16, 127, 0.5f, 127, 0.5f, kp, bp, 127, 0.5f, 0, 255, 0, &project) == qnnp_status_success);
    CHECK(qnnp_create_add_nc_q8(16, 127, 0.5f, 127, 0.5f, 127, 0.75f, 0, 255, 0, &add) == qnnp_status_success);
    free(ke); free(be); free(kd); free(bd); free(kp); free(bp);
    CHECK(qnnp_gfx950_create_fused_block(expand, dw, project, add, &fused) == qnnp_status_success);
    uint8_t* x = random_bytes(2 * 14 * 14 * 16), * y = random_bytes(2 * 14 * 14 * 16);
    CHECK(qnnp_gfx950_setup_fused_block(fused, 2, 14, 14, x, 16, y, 16) == qnnp_status_success);
    CHECK(qnnp_run_operator(fused, NULL) == qnnp_status_success);
    free(x); free(y);
    CHECK(qnnp_delete_operator(fused) == qnnp_status_success);
    CHECK(qnnp_delete_operator(expand) == qnnp_status_success);
    CHECK(qnnp_delete_operator(dw) == qnnp_status_success);
    CHECK(qnnp_delete_operator(project) == qnnp_status_success);
    CHECK(qnnp_delete_operator(add) == qnnp_status_success);
  }

  CHECK(qnnp_deinitialize() == qnnp_status_success);
  puts("host-sanitizers-ok");
  return 0;
}
