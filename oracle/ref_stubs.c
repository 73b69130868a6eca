/*
 * ref_stubs.c -- link-time shims that let the UNMODIFIED reference sources under
 * /root/reference be compiled into oracle/_ref/libqnnpack_ref.so ("O2") without
 * its un-vendored dependencies (cpuinfo, pthreadpool), which CMake would fetch
 * from the network. This file is our own code; no reference source is copied.
 *
 * TEST INFRASTRUCTURE ONLY: O2 is the timed CPU baseline (bench.py cpu_baseline
 * kind "reference") and the generator of tests/golden/. Not part of the product.
 *
 * Provided:
 *   - cpuinfo_initialize / cpuinfo_deinitialize  (x86-64: SSE2 is baseline, so
 *     src/init.c:182-237 takes the SSE2 branch via the inline cpuinfo_has_x86_sse2()).
 *   - the five legacy pthreadpool_compute_* entry points src/operator-run.c uses,
 *     run over an OpenMP team. threadpool == NULL means "calling thread only",
 *     which is the pthreadpool contract every in-tree caller relies on.
 *   - pthreadpool_create / pthreadpool_destroy / pthreadpool_get_threads_count.
 */
#include <stdbool.h>
#include <stddef.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif

struct pthreadpool {
  int threads;
};
typedef struct pthreadpool* pthreadpool_t;

bool cpuinfo_initialize(void) { return true; }
void cpuinfo_deinitialize(void) {}

pthreadpool_t pthreadpool_create(size_t threads_count)
{
  pthreadpool_t p = (pthreadpool_t) malloc(sizeof(struct pthreadpool));
  if (p == NULL) return NULL;
#ifdef _OPENMP
  p->threads = threads_count == 0 ? omp_get_num_procs() : (int) threads_count;
#else
  p->threads = 1;
#endif
  return p;
}

void pthreadpool_destroy(pthreadpool_t p) { free(p); }

size_t pthreadpool_get_threads_count(pthreadpool_t p) { return p == NULL ? 1 : (size_t) p->threads; }

static inline int team(pthreadpool_t p) { return p == NULL ? 1 : p->threads; }
static inline size_t divup(size_t a, size_t b) { return (a + b - 1) / b; }
static inline size_t minsz(size_t a, size_t b) { return a < b ? a : b; }

typedef void (*fn_1d)(void*, size_t);
typedef void (*fn_1d_tiled)(void*, size_t, size_t);
typedef void (*fn_2d)(void*, size_t, size_t);
typedef void (*fn_3d_tiled)(void*, size_t, size_t, size_t, size_t, size_t, size_t);
typedef void (*fn_4d_tiled)(void*, size_t, size_t, size_t, size_t, size_t, size_t, size_t, size_t);

void pthreadpool_compute_1d(pthreadpool_t p, fn_1d f, void* arg, size_t range)
{
  const int nt = team(p);
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
  for (ptrdiff_t i = 0; i < (ptrdiff_t) range; i++) f(arg, (size_t) i);
}

void pthreadpool_compute_1d_tiled(pthreadpool_t p, fn_1d_tiled f, void* arg, size_t range, size_t tile)
{
  const int nt = team(p);
  const ptrdiff_t tiles = (ptrdiff_t) divup(range, tile);
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
  for (ptrdiff_t t = 0; t < tiles; t++) {
    const size_t start = (size_t) t * tile;
    f(arg, start, minsz(tile, range - start));
  }
}

void pthreadpool_compute_2d(pthreadpool_t p, fn_2d f, void* arg, size_t range_i, size_t range_j)
{
  const int nt = team(p);
  const ptrdiff_t total = (ptrdiff_t) (range_i * range_j);
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
  for (ptrdiff_t t = 0; t < total; t++) f(arg, (size_t) t / range_j, (size_t) t % range_j);
}

void pthreadpool_compute_3d_tiled(
    pthreadpool_t p, fn_3d_tiled f, void* arg,
    size_t range_i, size_t range_j, size_t range_k,
    size_t tile_i, size_t tile_j, size_t tile_k)
{
  const int nt = team(p);
  const size_t ti = divup(range_i, tile_i), tj = divup(range_j, tile_j), tk = divup(range_k, tile_k);
  const ptrdiff_t total = (ptrdiff_t) (ti * tj * tk);
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
  for (ptrdiff_t t = 0; t < total; t++) {
    const size_t k = ((size_t) t % tk) * tile_k;
    const size_t j = (((size_t) t / tk) % tj) * tile_j;
    const size_t i = ((size_t) t / (tk * tj)) * tile_i;
    f(arg, i, j, k, minsz(tile_i, range_i - i), minsz(tile_j, range_j - j), minsz(tile_k, range_k - k));
  }
}

void pthreadpool_compute_4d_tiled(
    pthreadpool_t p, fn_4d_tiled f, void* arg,
    size_t range_i, size_t range_j, size_t range_k, size_t range_l,
    size_t tile_i, size_t tile_j, size_t tile_k, size_t tile_l)
{
  const int nt = team(p);
  const size_t ti = divup(range_i, tile_i), tj = divup(range_j, tile_j);
  const size_t tk = divup(range_k, tile_k), tl = divup(range_l, tile_l);
  const ptrdiff_t total = (ptrdiff_t) (ti * tj * tk * tl);
#pragma omp parallel for num_threads(nt) schedule(static) if (nt > 1)
  for (ptrdiff_t t = 0; t < total; t++) {
    const size_t l = ((size_t) t % tl) * tile_l;
    const size_t k = (((size_t) t / tl) % tk) * tile_k;
    const size_t j = (((size_t) t / (tl * tk)) % tj) * tile_j;
    const size_t i = ((size_t) t / (tl * tk * tj)) * tile_i;
    f(arg, i, j, k, l,
      minsz(tile_i, range_i - i), minsz(tile_j, range_j - j),
      minsz(tile_k, range_k - k), minsz(tile_l, range_l - l));
  }
}
