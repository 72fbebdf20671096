/* c_host_smoke.c -- a plain C host of the library, the way Kvazaar itself (C99, no HIP headers, no C++) would use the
 * batched entries of include/kvz_hip.h: device memory and streams through the kvz_hip_* helpers, known-answer checks
 * taken from the reference's own unit tests, a frame graph, and the error channel.
 * Built by __graft_entry__.build() with gcc and linked against kvazaar_amd/libkvzhip.so; run by tests/test_gpu_c_host.py.
 * Exit code 0 = every check passed. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kvz_hip.h"

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); \
                                             fprintf(stderr, " (%s)\n", kvz_hip_last_error()); return 1; } } while (0)

int main(void)
{
  CHECK(kvz_hip_abi_version() == KVZ_HIP_ABI_VERSION, "library ABI %d, header %d", kvz_hip_abi_version(), KVZ_HIP_ABI_VERSION);
  CHECK(kvz_hip_init(-1) == KVZ_HIP_OK, "kvz_hip_init");
  printf("device: %s\n", kvz_hip_device_name());
  kvz_hip_stream st = kvz_hip_stream_create();
  CHECK(st != NULL, "stream");

  /* ---- satd_8x8 / sad_8x8 known answers: tests/satd_tests.c:109-146 (black vs white: 8x8 SATD of a flat
   * difference of 255 is its DC term, 255 * 64 * 8 / ... = (64 * 255 + 2) >> 2 per the 8x8 normalisation) ---- */
  enum { N = 1000 };
  unsigned char *h_a = malloc(N * 64), *h_b = malloc(N * 64);
  uint32_t *h_cost = malloc(N * sizeof(uint32_t));
  for (int i = 0; i < N; ++i) {
    memset(h_a + i * 64, (i & 1) ? 255 : 10, 64);
    memset(h_b + i * 64, (i & 1) ? 0 : 13, 64);
  }
  kvz_hip_pixel *d_a = kvz_hip_malloc(N * 64), *d_b = kvz_hip_malloc(N * 64);
  uint32_t *d_cost = kvz_hip_malloc(N * sizeof(uint32_t));
  CHECK(d_a && d_b && d_cost, "kvz_hip_malloc");
  CHECK(kvz_hip_memcpy_h2d(d_a, h_a, N * 64, st) == KVZ_HIP_OK, "h2d");
  CHECK(kvz_hip_memcpy_h2d(d_b, h_b, N * 64, st) == KVZ_HIP_OK, "h2d");
  CHECK(kvz_hip_sad_nxn_batch(8, d_a, d_b, N, d_cost, st) == KVZ_HIP_OK, "sad_nxn_batch");
  CHECK(kvz_hip_memcpy_d2h(h_cost, d_cost, N * sizeof(uint32_t), st) == KVZ_HIP_OK, "d2h");
  for (int i = 0; i < N; ++i) CHECK(h_cost[i] == ((i & 1) ? 64u * 255u : 64u * 3u), "sad_8x8[%d] = %u", i, h_cost[i]);
  CHECK(kvz_hip_satd_nxn_batch(8, d_a, d_b, N, d_cost, st) == KVZ_HIP_OK, "satd_nxn_batch");
  CHECK(kvz_hip_memcpy_d2h(h_cost, d_cost, N * sizeof(uint32_t), st) == KVZ_HIP_OK, "d2h");
  /* a flat difference d has one non-zero Hadamard coefficient, 64 d; satd_8x8 = (64 d + 2) >> 2 (picture-generic.c:240-328) */
  for (int i = 0; i < N; ++i) CHECK(h_cost[i] == (((i & 1) ? 64u * 255u : 64u * 3u) + 2) >> 2, "satd_8x8[%d] = %u", i, h_cost[i]);

  /* ---- dct_32x32 of a flat block: only the DC coefficient; 64 * 32 * v >> (shift 4 then 11) (dct-generic.c:600-640) ---- */
  enum { B = 16 };
  kvz_hip_coeff *h_res = malloc(B * 1024 * sizeof(kvz_hip_coeff)), *h_coef = malloc(B * 1024 * sizeof(kvz_hip_coeff));
  for (int i = 0; i < B; ++i)
    for (int k = 0; k < 1024; ++k) h_res[i * 1024 + k] = (kvz_hip_coeff)(i - 8);
  kvz_hip_coeff *d_res = kvz_hip_malloc(B * 2048), *d_coef = kvz_hip_malloc(B * 2048);
  CHECK(d_res && d_coef, "kvz_hip_malloc");
  CHECK(kvz_hip_memcpy_h2d(d_res, h_res, B * 2048, st) == KVZ_HIP_OK, "h2d");

  /* the launch sequence of a "frame" captured once and replayed: sad, satd, dct on one stream */
  kvz_hip_graph g = NULL;
  CHECK(kvz_hip_graph_begin(st) == KVZ_HIP_OK, "graph_begin");
  CHECK(kvz_hip_sad_nxn_batch(8, d_a, d_b, N, d_cost, st) == KVZ_HIP_OK, "sad in capture");
  CHECK(kvz_hip_transform_batch(0, 32, d_res, d_coef, B, st) == KVZ_HIP_OK, "dct in capture");
  CHECK(kvz_hip_graph_end(st, &g) == KVZ_HIP_OK && g, "graph_end");
  for (int rep = 0; rep < 3; ++rep) CHECK(kvz_hip_graph_launch(g, st) == KVZ_HIP_OK, "graph_launch");
  CHECK(kvz_hip_memcpy_d2h(h_coef, d_coef, B * 2048, st) == KVZ_HIP_OK, "d2h");
  for (int i = 0; i < B; ++i) {
    /* first pass: (64 * 32 * v + 8) >> 4 per column, second: (64 * 32 * that + 1024) >> 11 */
    const int v = i - 8, p1 = (64 * 32 * v + 8) >> 4, dc = (64 * 32 * p1 + 1024) >> 11;
    CHECK(h_coef[i * 1024] == (kvz_hip_coeff)dc, "dct32 DC[%d] = %d, want %d", i, h_coef[i * 1024], dc);
    for (int k = 1; k < 1024; ++k) CHECK(h_coef[i * 1024 + k] == 0, "dct32 AC[%d][%d] = %d", i, k, h_coef[i * 1024 + k]);
  }
  kvz_hip_graph_destroy(g);

  /* ---- the error channel: an unsupported size is refused with a message, nothing is launched ---- */
  CHECK(kvz_hip_sad_nxn_batch(7, d_a, d_b, N, d_cost, st) == KVZ_HIP_ERR_INVALID, "sad_nxn_batch(7) must be refused");
  CHECK(strstr(kvz_hip_last_error(), "kvz_hip_sad_nxn_batch") != NULL, "error text names the entry");

  kvz_hip_free(d_a); kvz_hip_free(d_b); kvz_hip_free(d_cost); kvz_hip_free(d_res); kvz_hip_free(d_coef);
  kvz_hip_stream_destroy(st);
  kvz_hip_shutdown();
  free(h_a); free(h_b); free(h_cost); free(h_res); free(h_coef);
  printf("c_host_smoke ok\n");
  return 0;
}
