/*
 * consumer.c -- TEST INFRASTRUCTURE: a C caller written against the REFERENCE's public header and nothing else.
 *
 * It is compiled with -I<reference>/include (the upstream include/qnnpack.h, unmodified; <pthreadpool.h> from the
 * PyTorch wheel as for oracle/_ref) and linked to libqnnpack_gfx950.so: the translation unit a QNNPACK user already
 * has, re-linked. That is the drop-in claim of INTEGRATION.md section 1 in executable form -- ctypes proves the
 * symbols and calling convention, this proves the header (prototypes, enum values, the opaque qnnp_operator_t).
 * Flow and parameters follow the reference's own benchmark driver (bench/convolution.cc:59-98, bench/q8gemm.cc):
 * host tensors, NULL thread pool, create -> setup -> run -> delete.
 *
 *   consumer <case> <directory>
 * writes <directory>/<case>.{in,kernel,bias,out} (raw bytes / int32) and a one-line description to stdout; the
 * pytest side recomputes the expected output from the dumped tensors with the scalar oracle.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <qnnpack.h>

struct conv_case {
  const char* name;
  uint32_t pad, kh, kw, stride, dilation, groups;
  size_t gic, goc, batch, h, w;
  uint8_t izp, kzp, ozp;
  float iscale, kscale, oscale;
  uint8_t qmin, qmax;
  int fully_connected;   /* 1: qnnp_*_fully_connected_nc_q8 with batch rows, gic inputs, goc outputs */
};

static const struct conv_case cases[] = {
  /* bench/convolution.cc quantization (127, 0.5, 127, 0.5 -> 127, 0.5; clamp 0..255) */
  {"conv3x3", 1, 3, 3, 1, 1, 1, 16, 24, 2, 17, 13, 127, 127, 127, 0.5f, 0.5f, 0.5f, 0, 255, 0},
  {"conv3x3s2_rgb", 1, 3, 3, 2, 1, 1, 3, 32, 2, 32, 32, 127, 127, 127, 0.5f, 0.5f, 0.5f, 0, 255, 0},
  {"dw3x3", 1, 3, 3, 1, 1, 40, 1, 1, 2, 14, 15, 121, 133, 120, 0.5f, 0.25f, 0.625f, 5, 250, 0},
  {"grouped1x1", 0, 1, 1, 1, 1, 2, 17, 19, 1, 9, 8, 3, 250, 100, 0.75f, 0.5f, 0.5f, 0, 255, 0},
  /* BASELINE.json configs[0]: qnnp_fully_connected_nc_q8 M=1 K=1024 N=1000; bench/q8gemm.cc:103 quantization */
  {"fc_1x1024x1000", 0, 1, 1, 1, 1, 1, 1024, 1000, 1, 1, 1, 127, 127, 127, 0.75f, 1.0f, 1.0f, 1, 254, 1},
  {"fc_37x200x96", 0, 1, 1, 1, 1, 1, 200, 96, 37, 1, 1, 127, 127, 127, 0.75f, 1.0f, 1.0f, 1, 254, 1},
};

static uint32_t lcg = 0x2545F491u;
static uint32_t next_u32(void) { lcg = lcg * 1664525u + 1013904223u; return lcg; }

static int dump(const char* dir, const char* name, const char* ext, const void* data, size_t bytes)
{
  char path[1024];
  snprintf(path, sizeof(path), "%s/%s.%s", dir, name, ext);
  FILE* f = fopen(path, "wb");
  if (f == NULL) return -1;
  const size_t n = fwrite(data, 1, bytes, f);
  fclose(f);
  return n == bytes ? 0 : -1;
}

int main(int argc, char** argv)
{
  if (argc != 3) {
    fprintf(stderr, "usage: %s <case> <directory>\n", argv[0]);
    return 2;
  }
  const struct conv_case* c = NULL;
  for (size_t i = 0; i < sizeof(cases) / sizeof(cases[0]); i++) {
    if (strcmp(cases[i].name, argv[1]) == 0) c = &cases[i];
  }
  if (c == NULL) {
    fprintf(stderr, "unknown case %s\n", argv[1]);
    return 2;
  }
  const size_t cin = c->groups * c->gic, cout = c->groups * c->goc;
  const size_t eff_kh = (c->kh - 1) * c->dilation + 1, eff_kw = (c->kw - 1) * c->dilation + 1;
  const size_t oh = (c->h + 2 * c->pad - eff_kh) / c->stride + 1, ow = (c->w + 2 * c->pad - eff_kw) / c->stride + 1;
  const size_t in_bytes = c->batch * c->h * c->w * cin, out_bytes = c->batch * oh * ow * cout;
  const size_t kernel_bytes = (size_t) c->groups * c->goc * c->kh * c->kw * c->gic;
  uint8_t* in = (uint8_t*) malloc(in_bytes);
  uint8_t* kernel = (uint8_t*) malloc(kernel_bytes);
  int32_t* bias = (int32_t*) malloc(cout * sizeof(int32_t));
  uint8_t* out = (uint8_t*) malloc(out_bytes);
  if (in == NULL || kernel == NULL || bias == NULL || out == NULL) return 3;
  for (size_t i = 0; i < in_bytes; i++) in[i] = (uint8_t) (next_u32() >> 24);
  for (size_t i = 0; i < kernel_bytes; i++) kernel[i] = (uint8_t) (next_u32() >> 24);
  for (size_t i = 0; i < cout; i++) bias[i] = (int32_t) (next_u32() % 20001u) - 10000;
  memset(out, 0xA5, out_bytes);

  enum qnnp_status status = qnnp_initialize();
  if (status != qnnp_status_success) {
    fprintf(stderr, "qnnp_initialize -> %d\n", (int) status);
    return 4;
  }
  qnnp_operator_t op = NULL;
  if (c->fully_connected) {
    status = qnnp_create_fully_connected_nc_q8(c->gic, c->goc, c->izp, c->iscale, c->kzp, c->kscale, kernel, bias,
        c->ozp, c->oscale, c->qmin, c->qmax, 0 /* flags */, &op);
    if (status == qnnp_status_success) {
      status = qnnp_setup_fully_connected_nc_q8(op, c->batch, in, c->gic, out, c->goc);
    }
  } else {
    status = qnnp_create_convolution2d_nhwc_q8(c->pad, c->pad, c->pad, c->pad, c->kh, c->kw, c->stride, c->stride,
        c->dilation, c->dilation, c->groups, c->gic, c->goc, c->izp, c->iscale, c->kzp, c->kscale, kernel, bias,
        c->ozp, c->oscale, c->qmin, c->qmax, 0 /* flags */, &op);
    if (status == qnnp_status_success) {
      status = qnnp_setup_convolution2d_nhwc_q8(op, c->batch, c->h, c->w, in, cin, out, cout, NULL /* thread pool */);
    }
  }
  if (status != qnnp_status_success) {
    fprintf(stderr, "create/setup -> %d\n", (int) status);
    return 5;
  }
  status = qnnp_run_operator(op, NULL /* thread pool */);
  if (status != qnnp_status_success) {
    fprintf(stderr, "qnnp_run_operator -> %d\n", (int) status);
    return 6;
  }
  if (qnnp_delete_operator(op) != qnnp_status_success || qnnp_deinitialize() != qnnp_status_success) return 7;

  if (dump(argv[2], c->name, "in", in, in_bytes) != 0 || dump(argv[2], c->name, "kernel", kernel, kernel_bytes) != 0 ||
      dump(argv[2], c->name, "bias", bias, cout * sizeof(int32_t)) != 0 || dump(argv[2], c->name, "out", out, out_bytes) != 0) {
    return 8;
  }
  printf("%s fc=%d pad=%u k=%ux%u stride=%u dilation=%u groups=%u gic=%zu goc=%zu batch=%zu in=%zux%zu out=%zux%zu "
         "izp=%u kzp=%u ozp=%u scale=%.9g qmin=%u qmax=%u\n",
      c->name, c->fully_connected, c->pad, c->kh, c->kw, c->stride, c->dilation, c->groups, c->gic, c->goc, c->batch,
      c->h, c->w, oh, ow, c->izp, c->kzp, c->ozp, (double) (c->iscale * c->kscale / c->oscale), c->qmin, c->qmax);
  free(in); free(kernel); free(bias); free(out);
  return 0;
}
