/* c_host_smoke.c -- a plain C host of the library, the way Kvazaar itself (C99, no HIP headers, no C++) would use the
 * batched entries of include/kvz_hip.h: device memory and streams through the kvz_hip_* helpers, known-answer checks
 * taken from the reference's own unit tests, a frame graph, and the error channel.
 * Built by __graft_entry__.build() with gcc and linked against kvazaar_amd/libkvzhip.so; run by tests/test_gpu_c_host.py.
 * Exit code 0 = every check passed. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kvz_hip.h"

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); \
                                             fprintf(stderr, " (%s)\n", kvz_hip_last_error()); return 1; } } while (0)

/* One worker thread of a multi-device host (the reference's workers are pthreads, threadqueue.c:263): binds itself to
 * a device with kvz_hip_set_device and runs known-answer launches there with its own stream and buffers.  On a
 * 1-GPU box both workers get device 0 -- two threads, two streams, one context; with more devices each gets its own. */
typedef struct { int device; int seed; int failed; char msg[256]; } worker_arg;
#define WCHECK(cond, ...) do { if (!(cond)) { snprintf(a->msg, sizeof(a->msg), __VA_ARGS__); a->failed = 1; return NULL; } } while (0)
static void *worker(void *p)
{
  worker_arg *a = p;
  enum { N = 4096 };
  WCHECK(kvz_hip_set_device(a->device) == KVZ_HIP_OK, "set_device(%d): %s", a->device, kvz_hip_last_error());
  WCHECK(kvz_hip_get_device() == a->device, "get_device = %d, want %d", kvz_hip_get_device(), a->device);
  kvz_hip_stream st = kvz_hip_stream_create();
  unsigned char *h_a = malloc(N * 64), *h_b = malloc(N * 64);
  uint32_t *h_cost = malloc(N * sizeof(uint32_t));
  for (int i = 0; i < N; ++i) { memset(h_a + i * 64, (i * 7 + a->seed) & 255, 64); memset(h_b + i * 64, (i * 3) & 255, 64); }
  kvz_hip_pixel *d_a = kvz_hip_malloc(N * 64), *d_b = kvz_hip_malloc(N * 64);
  uint32_t *d_cost = kvz_hip_malloc(N * sizeof(uint32_t));
  WCHECK(st && d_a && d_b && d_cost, "alloc on device %d: %s", a->device, kvz_hip_last_error());
  for (int rep = 0; rep < 20; ++rep) {
    WCHECK(kvz_hip_memcpy_h2d(d_a, h_a, N * 64, st) == KVZ_HIP_OK && kvz_hip_memcpy_h2d(d_b, h_b, N * 64, st) == KVZ_HIP_OK, "h2d");
    WCHECK(kvz_hip_satd_nxn_batch(8, d_a, d_b, N, d_cost, st) == KVZ_HIP_OK, "satd: %s", kvz_hip_last_error());
    WCHECK(kvz_hip_memcpy_d2h(h_cost, d_cost, N * sizeof(uint32_t), st) == KVZ_HIP_OK, "d2h");
    for (int i = 0; i < N; ++i) {
      const int d = abs(((i * 7 + a->seed) & 255) - ((i * 3) & 255));
      WCHECK(h_cost[i] == (64u * (unsigned)d + 2) >> 2, "thread seed %d: satd_8x8[%d] = %u, flat difference %d", a->seed, i, h_cost[i], d);
    }
  }
  kvz_hip_free(d_a); kvz_hip_free(d_b); kvz_hip_free(d_cost); kvz_hip_stream_destroy(st);
  free(h_a); free(h_b); free(h_cost);
  return NULL;
}

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

  /* ---- the NULL stream is ordered like the legacy default stream: a memset queued there with no synchronisation is
   * seen by the entry, and the entry's result by a following default-stream copy (kvz_hip.h, kvz_hip_stream) ---- */
  CHECK(kvz_hip_memset(d_a, 9, N * 64, NULL) == KVZ_HIP_OK && kvz_hip_memset(d_b, 4, N * 64, NULL) == KVZ_HIP_OK, "memset");
  CHECK(kvz_hip_sad_nxn_batch(8, d_a, d_b, N, d_cost, NULL) == KVZ_HIP_OK, "sad on the NULL stream");
  CHECK(kvz_hip_memcpy_d2h(h_cost, d_cost, N * sizeof(uint32_t), NULL) == KVZ_HIP_OK, "d2h");
  for (int i = 0; i < N; ++i) CHECK(h_cost[i] == 64u * 5u, "NULL-stream sad_8x8[%d] = %u", i, h_cost[i]);

  /* ---- the glue between dependency fronts, on the device: intra references from a reconstruction plane, merge / AMVP
   * candidates from a CU array, both with answers known by construction ---- */
  {
    enum { PW = 64, PH = 64 };
    uint8_t *d_rec = kvz_hip_malloc(PW * PH);
    kvz_hip_intra_pos h_pos[2] = { { 0, 0 }, { 8, 8 } }, *d_pos = kvz_hip_malloc(sizeof(h_pos));
    kvz_hip_intra_ref h_refs[2], *d_refs = kvz_hip_malloc(sizeof(h_refs));
    CHECK(d_rec && d_pos && d_refs, "kvz_hip_malloc");
    CHECK(kvz_hip_memset(d_rec, 77, PW * PH, st) == KVZ_HIP_OK && kvz_hip_memcpy_h2d(d_pos, h_pos, sizeof(h_pos), st) == KVZ_HIP_OK, "plane");
    CHECK(kvz_hip_intra_build_reference_batch(3, 0, d_rec, PW, PW, PH, d_pos, 2, d_refs, st) == KVZ_HIP_OK, "intra_build_reference: %s", kvz_hip_last_error());
    CHECK(kvz_hip_memcpy_d2h(h_refs, d_refs, sizeof(h_refs), st) == KVZ_HIP_OK, "d2h");
    for (int k = 0; k <= 16; ++k) {
      CHECK(h_refs[0].left[k] == 128 && h_refs[0].top[k] == 128, "picture corner: mid grey, got %d / %d at %d", h_refs[0].left[k], h_refs[0].top[k], k);
      CHECK(h_refs[1].left[k] == 77 && h_refs[1].top[k] == 77, "inside a flat plane: its value, got %d / %d at %d", h_refs[1].left[k], h_refs[1].top[k], k);
    }
    /* a 64 x 64 picture whose left half is one inter CU with vector (12, -8): the 8x8 PU at (32, 0) sees it as A1 */
    kvz_hip_cu_info h_cus[16 * 16];
    memset(h_cus, 0, sizeof(h_cus));
    for (int y = 0; y < 16; ++y) for (int x = 0; x < 8; ++x) {
      kvz_hip_cu_info *c = &h_cus[y * 16 + x];
      c->type = 2; c->mv_dir = 1; c->mv[0][0] = 12; c->mv[0][1] = -8;
    }
    kvz_hip_cu_info *d_cus = kvz_hip_malloc(sizeof(h_cus));
    kvz_hip_me_pu h_pu, *d_pu = kvz_hip_malloc(sizeof(h_pu));
    kvz_hip_merge_cand h_mc[5], *d_mc = kvz_hip_malloc(sizeof(h_mc));
    memset(&h_pu, 0, sizeof(h_pu));
    h_pu.x = 32; h_pu.y = 0; h_pu.width = 8; h_pu.height = 8;
    kvz_hip_inter_params ip;
    memset(&ip, 0, sizeof(ip));
    ip.poc = 1; ip.num_refs = 1; ip.ref_pocs[0] = 0; ip.ref_LX_size[0] = 1;
    ip.pic_width = ip.in_width = PW; ip.pic_height = ip.in_height = PH; ip.cus_stride = ip.col_stride = 16;
    CHECK(d_cus && d_pu && d_mc, "kvz_hip_malloc");
    CHECK(kvz_hip_memcpy_h2d(d_cus, h_cus, sizeof(h_cus), st) == KVZ_HIP_OK && kvz_hip_memcpy_h2d(d_pu, &h_pu, sizeof(h_pu), st) == KVZ_HIP_OK, "h2d");
    CHECK(kvz_hip_inter_candidates_batch(d_cus, NULL, NULL, &ip, d_pu, 1, d_mc, st) == KVZ_HIP_OK, "inter_candidates: %s", kvz_hip_last_error());
    CHECK(kvz_hip_memcpy_d2h(&h_pu, d_pu, sizeof(h_pu), st) == KVZ_HIP_OK && kvz_hip_memcpy_d2h(h_mc, d_mc, sizeof(h_mc), st) == KVZ_HIP_OK, "d2h");
    CHECK(h_pu.num_merge_cand == 5 && h_mc[0].dir == 1 && h_mc[0].mv[0][0] == 12 && h_mc[0].mv[0][1] == -8 && h_mc[0].ref[0] == 0,
          "merge candidate 0 = A1's motion, got dir %d mv (%d, %d)", h_mc[0].dir, h_mc[0].mv[0][0], h_mc[0].mv[0][1]);
    CHECK(h_mc[1].dir == 1 && h_mc[1].mv[0][0] == 0 && h_mc[1].mv[0][1] == 0, "merge candidate 1 = the zero vector");
    CHECK(h_pu.mv_cand[0][0] == 12 && h_pu.mv_cand[0][1] == -8 && h_pu.mv_cand[1][0] == 0 && h_pu.mv_cand[1][1] == 0,
          "AMVP = (A1, zero), got (%d, %d) (%d, %d)", h_pu.mv_cand[0][0], h_pu.mv_cand[0][1], h_pu.mv_cand[1][0], h_pu.mv_cand[1][1]);
    CHECK(h_pu.merge[0].usable == 1 && h_pu.merge[0].same_ref == 1 && h_pu.merge[0].mv[0] == 12, "the search's view of merge candidate 0");
    /* the same PU through the several-pictures entry: one record, picture 0 */
    kvz_hip_inter_picture h_picrec, *d_picrec = kvz_hip_malloc(sizeof(h_picrec));
    memset(&h_picrec, 0, sizeof(h_picrec));
    h_picrec.cus = d_cus; h_picrec.params = ip;
    memset(&h_pu, 0, sizeof(h_pu));
    h_pu.x = 32; h_pu.y = 0; h_pu.width = 8; h_pu.height = 8;
    CHECK(d_picrec && kvz_hip_memcpy_h2d(d_picrec, &h_picrec, sizeof(h_picrec), st) == KVZ_HIP_OK && kvz_hip_memcpy_h2d(d_pu, &h_pu, sizeof(h_pu), st) == KVZ_HIP_OK, "h2d");
    CHECK(kvz_hip_inter_candidates_multi_batch(d_picrec, 1, d_pu, 1, NULL, st) == KVZ_HIP_OK, "inter_candidates_multi: %s", kvz_hip_last_error());
    CHECK(kvz_hip_memcpy_d2h(&h_pu, d_pu, sizeof(h_pu), st) == KVZ_HIP_OK, "d2h");
    CHECK(h_pu.num_merge_cand == 5 && h_pu.mv_cand[0][0] == 12 && h_pu.mv_cand[0][1] == -8 && h_pu.merge[0].mv[0] == 12, "several-pictures entry, picture 0");
    h_pu.pad = 1 << 2;                                    /* a picture the table does not have */
    CHECK(kvz_hip_memcpy_h2d(d_pu, &h_pu, sizeof(h_pu), st) == KVZ_HIP_OK, "h2d");
    CHECK(kvz_hip_inter_candidates_multi_batch(d_picrec, 1, d_pu, 1, NULL, st) == KVZ_HIP_OK, "inter_candidates_multi");
    CHECK(kvz_hip_memcpy_d2h(&h_pu, d_pu, sizeof(h_pu), st) == KVZ_HIP_OK && h_pu.num_merge_cand == -1, "unknown picture is flagged");
    kvz_hip_free(d_picrec);
    kvz_hip_free(d_rec); kvz_hip_free(d_pos); kvz_hip_free(d_refs); kvz_hip_free(d_cus); kvz_hip_free(d_pu); kvz_hip_free(d_mc);
  }

  /* ---- several contexts in one process: every visible device gets a context; a second init of another index is a
   * second context, an index beyond the device count is refused; two pthreads drive (up to) two devices at once ---- */
  const int ndev = kvz_hip_device_count();
  CHECK(ndev >= 1, "device count");
  CHECK(kvz_hip_init(ndev) == KVZ_HIP_ERR_INVALID, "init of a device index beyond the count must be refused");
  CHECK(strstr(kvz_hip_last_error(), "out of range") != NULL, "error text");
  const int home = kvz_hip_get_device();
  for (int d = 0; d < ndev; ++d) CHECK(kvz_hip_init(d) == KVZ_HIP_OK && kvz_hip_get_device() == d, "init(%d)", d);
  CHECK(kvz_hip_set_device(home) == KVZ_HIP_OK, "back to device %d", home);
  worker_arg wa[2] = { { 0, 1, 0, "" }, { ndev > 1 ? 1 : 0, 2, 0, "" } };
  pthread_t th[2];
  for (int i = 0; i < 2; ++i) CHECK(pthread_create(&th[i], NULL, worker, &wa[i]) == 0, "pthread_create");
  for (int i = 0; i < 2; ++i) pthread_join(th[i], NULL);
  for (int i = 0; i < 2; ++i) CHECK(!wa[i].failed, "worker %d on device %d: %s", i, wa[i].device, wa[i].msg);
  CHECK(kvz_hip_get_device() == home, "the main thread's device is untouched by the workers");
  /* shard-boundary exchange inside one process: rows of device A's plane into device B's halo (same device on a 1-GPU box) */
  {
    const int da = 0, db = ndev > 1 ? 1 : 0;
    enum { ROWS = 80, W = 3840 };
    CHECK(kvz_hip_set_device(da) == KVZ_HIP_OK, "set_device");
    unsigned char *src = kvz_hip_malloc((size_t)ROWS * W);
    CHECK(src && kvz_hip_memset(src, 0x5A, (size_t)ROWS * W, NULL) == KVZ_HIP_OK, "src rows");
    CHECK(kvz_hip_set_device(db) == KVZ_HIP_OK, "set_device");
    unsigned char *halo = kvz_hip_malloc((size_t)ROWS * W);
    CHECK(halo && kvz_hip_memset(halo, 0, (size_t)ROWS * W, NULL) == KVZ_HIP_OK, "halo rows");
    CHECK(kvz_hip_set_device(da) == KVZ_HIP_OK && kvz_hip_stream_sync(NULL) == KVZ_HIP_OK, "src ready");
    CHECK(kvz_hip_set_device(db) == KVZ_HIP_OK, "set_device");
    CHECK(kvz_hip_memcpy_peer(halo, db, src, da, (size_t)ROWS * W, NULL) == KVZ_HIP_OK, "memcpy_peer");
    unsigned char *back = malloc((size_t)ROWS * W);
    CHECK(kvz_hip_memcpy_d2h(back, halo, (size_t)ROWS * W, NULL) == KVZ_HIP_OK, "d2h");
    for (int i = 0; i < ROWS * W; i += 997) CHECK(back[i] == 0x5A, "halo byte %d = %d", i, back[i]);
    free(back); kvz_hip_free(halo);
    CHECK(kvz_hip_set_device(da) == KVZ_HIP_OK, "set_device"); kvz_hip_free(src);
    CHECK(kvz_hip_set_device(home) == KVZ_HIP_OK, "set_device");
  }
  /* the same exchange as one call per shard (kvz_hip_halo_exchange): two row shards with a halo of MARGIN rows towards each other */
  {
    const int da = 0, db = ndev > 1 ? 1 : 0;
    enum { MARGIN = 16, ROWS = 64, W = 256 };
    const size_t ext = (size_t)(ROWS + MARGIN) * W;        /* A: own rows then the halo below; B: the halo above then own rows */
    unsigned char *h = malloc(ext), *back = malloc(ext);
    CHECK(kvz_hip_set_device(da) == KVZ_HIP_OK, "set_device");
    unsigned char *a = kvz_hip_malloc(ext);
    for (size_t i = 0; i < ext; ++i) h[i] = i < (size_t)ROWS * W ? (unsigned char)(1 + i % 101) : 0;
    CHECK(a && kvz_hip_memcpy_h2d(a, h, ext, NULL) == KVZ_HIP_OK && kvz_hip_stream_sync(NULL) == KVZ_HIP_OK, "shard A");
    CHECK(kvz_hip_set_device(db) == KVZ_HIP_OK, "set_device");
    unsigned char *b = kvz_hip_malloc(ext);
    for (size_t i = 0; i < ext; ++i) h[i] = i >= (size_t)MARGIN * W ? (unsigned char)(7 + i % 89) : 0;
    CHECK(b && kvz_hip_memcpy_h2d(b, h, ext, NULL) == KVZ_HIP_OK && kvz_hip_stream_sync(NULL) == KVZ_HIP_OK, "shard B");
    const kvz_hip_shard_plane pa = { a, da, 0, ROWS }, pb = { b, db, MARGIN, ROWS };
    CHECK(kvz_hip_set_device(da) == KVZ_HIP_OK && kvz_hip_halo_exchange(&pa, NULL, &pb, W, MARGIN, NULL) == KVZ_HIP_OK &&
          kvz_hip_stream_sync(NULL) == KVZ_HIP_OK, "A pushes its last rows: %s", kvz_hip_last_error());
    CHECK(kvz_hip_set_device(db) == KVZ_HIP_OK && kvz_hip_halo_exchange(&pb, &pa, NULL, W, MARGIN, NULL) == KVZ_HIP_OK &&
          kvz_hip_stream_sync(NULL) == KVZ_HIP_OK, "B pushes its first rows: %s", kvz_hip_last_error());
    CHECK(kvz_hip_memcpy_d2h(back, b, ext, NULL) == KVZ_HIP_OK, "d2h");
    for (size_t i = 0; i < (size_t)MARGIN * W; ++i) CHECK(back[i] == (unsigned char)(1 + ((size_t)(ROWS - MARGIN) * W + i) % 101), "B's halo byte %zu", i);
    CHECK(kvz_hip_set_device(da) == KVZ_HIP_OK && kvz_hip_memcpy_d2h(back, a, ext, NULL) == KVZ_HIP_OK, "d2h");
    for (size_t i = 0; i < (size_t)MARGIN * W; ++i) CHECK(back[(size_t)ROWS * W + i] == (unsigned char)(7 + ((size_t)MARGIN * W + i) % 89), "A's halo byte %zu", i);
    CHECK(kvz_hip_halo_exchange(&pb, &pa, NULL, W, MARGIN, NULL) == KVZ_HIP_ERR_INVALID || da == db, "a shard of another device is refused");
    kvz_hip_free(a);
    CHECK(kvz_hip_set_device(db) == KVZ_HIP_OK, "set_device"); kvz_hip_free(b);
    CHECK(kvz_hip_set_device(home) == KVZ_HIP_OK, "set_device");
    free(h); free(back);
  }

  kvz_hip_free(d_a); kvz_hip_free(d_b); kvz_hip_free(d_cost); kvz_hip_free(d_res); kvz_hip_free(d_coef);
  kvz_hip_stream_destroy(st);
  kvz_hip_shutdown();
  free(h_a); free(h_b); free(h_cost); free(h_res); free(h_coef);
  printf("c_host_smoke ok\n");
  return 0;
}
