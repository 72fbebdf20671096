/* service_stress.c -- a plain C99 pthread host of the search service (include/kvz_hip.h "search service"), shaped like
 * the reference's thread pool (threadqueue.c:263: N workers, each posting one search at a time and blocking on it).
 *
 *   service_stress THREADS REQUESTS_PER_THREAD [WIDTH HEIGHT N_REFS [THINK_US [FULL_RANGE]]]
 *   (THINK_US: each worker computes for that long between two requests, like an encoder worker does; FULL_RANGE > 0: the
 *   exhaustive search of that range instead of hexbs)
 *
 * 1. a table of PUs (8x8 .. 64x64, random candidates) is searched ONCE through the service by one thread: the expected answers;
 * 2. THREADS workers then post the same PUs concurrently, in different orders, and every answer must equal the table's
 *    (the service must be deterministic under any interleaving, batching and ring wrap-around);
 * 3. prints requests/s, the mean time a worker waited per request and how the requests were batched.
 * Exit code 0 = every answer matched.  Built by __graft_entry__.build(); run by tests/test_gpu_c_host.py. */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "kvz_hip.h"

enum { N_PUS = 256 };

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static double now_s(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static kvz_hip_me_service *g_svc;
static kvz_hip_me_request g_req[N_PUS];
static kvz_hip_me_result g_want[N_PUS][KVZ_HIP_SERVICE_MAX_REFS];
static int g_per_thread;
static double g_think_s;

typedef struct { int id, failed; long done; char msg[200]; } worker_arg;

static void *worker(void *p)
{
  worker_arg *a = p;
  uint32_t s = 77u + 13u * (uint32_t)a->id;
  for (int k = 0; k < g_per_thread; ++k) {
    const int i = (int)(lcg(&s) % N_PUS);
    kvz_hip_me_result got[KVZ_HIP_SERVICE_MAX_REFS];
    if (kvz_hip_me_service_search(g_svc, &g_req[i], got) != KVZ_HIP_OK) {
      snprintf(a->msg, sizeof(a->msg), "request %d of thread %d: %s", k, a->id, kvz_hip_last_error());
      a->failed = 1;
      return NULL;
    }
    if (memcmp(got, g_want[i], sizeof(got[0]) * (size_t)g_req[i].n_refs) != 0) {
      snprintf(a->msg, sizeof(a->msg), "thread %d request %d (PU %d): answer differs from the single-threaded one", a->id, k, i);
      a->failed = 1;
      return NULL;
    }
    ++a->done;
    if (g_think_s > 0) { const double t = now_s(); while (now_s() - t < g_think_s) {} }
  }
  return NULL;
}

int main(int argc, char **argv)
{
  const int threads = argc > 1 ? atoi(argv[1]) : 16;
  g_per_thread = argc > 2 ? atoi(argv[2]) : 2000;
  const int w = argc > 3 ? atoi(argv[3]) : 640, h = argc > 4 ? atoi(argv[4]) : 384, n_refs = argc > 5 ? atoi(argv[5]) : 4;
  g_think_s = argc > 6 ? 1e-6 * atof(argv[6]) : 0.0;
  const int full_range = argc > 7 ? atoi(argv[7]) : 0;
  if (threads < 1 || threads > 512 || n_refs < 1 || n_refs > KVZ_HIP_SERVICE_MAX_REFS || w < 128 || h < 128 || (w & 7) || (h & 7)) return 2;
  if (kvz_hip_init(-1) != KVZ_HIP_OK) { fprintf(stderr, "kvz_hip_init: %s\n", kvz_hip_last_error()); return 1; }
  kvz_hip_me_service_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.width = w; cfg.height = h; cfg.max_pictures = 1 + n_refs; cfg.max_threads = threads + 1;
  g_svc = kvz_hip_me_service_create(&cfg);
  if (!g_svc) { fprintf(stderr, "kvz_hip_me_service_create: %s\n", kvz_hip_last_error()); return 1; }

  /* planes: a smooth texture; reference k is the picture moved by (2k + 1, -k) with a little noise */
  uint32_t s = 12345;
  unsigned char *tex = malloc((size_t)(w + 64) * (size_t)(h + 64)), *plane = malloc((size_t)w * (size_t)h);
  for (int y = 0; y < h + 64; ++y)
    for (int x = 0; x < w + 64; ++x)
      tex[(size_t)y * (size_t)(w + 64) + (size_t)x] = (unsigned char)(128 + ((x * 5 + y * 3) % 61) + ((x / 7 + y / 5) % 23) - 40 + (int)(lcg(&s) & 7));
  for (int k = 0; k <= n_refs; ++k) {
    const int dx = k ? 2 * k + 1 : 0, dy = k ? -k : 0;
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x)
        plane[(size_t)y * (size_t)w + (size_t)x] = (unsigned char)(tex[(size_t)(y + 32 + dy) * (size_t)(w + 64) + (size_t)(x + 32 + dx)] + (k ? (lcg(&s) & 3) : 0));
    if (kvz_hip_me_service_put_rect(g_svc, k, plane, (uint32_t)w, 0, 0, w, h) != KVZ_HIP_OK) { fprintf(stderr, "put_rect: %s\n", kvz_hip_last_error()); return 1; }
  }

  for (int i = 0; i < N_PUS; ++i) {
    kvz_hip_me_request *r = &g_req[i];
    memset(r, 0, sizeof(*r));
    const int size = 8 << (lcg(&s) % 4);
    r->pic_slot = 0; r->n_refs = 1 + (int)(lcg(&s) % (uint32_t)n_refs);
    r->cost_to_beat = 2147483647u;
    r->params.lambda_cost = 10 + (int)(lcg(&s) % 40); r->params.early_termination = 1; r->params.max_steps = 0xffffffffu;
    r->params.fme_level = 4; r->params.max_ref_lcu_down = 1; r->params.max_ref_lcu_right = 1;
    if (i % 5 == 0) { r->params.wpp_owf = 1; r->params.ref_delay_px = 10; }
    if (full_range > 0) { r->params.algorithm = 3; r->params.search_range = full_range; }
    const int x = (int)(lcg(&s) % (uint32_t)((w - size) / 8 + 1)) * 8, y = (int)(lcg(&s) % (uint32_t)((h - size) / 8 + 1)) * 8;
    for (int k = 0; k < r->n_refs; ++k) {
      kvz_hip_me_pu *pu = &r->pu[k];
      r->ref_slot[k] = 1 + k;
      pu->x = x; pu->y = y; pu->width = size; pu->height = size;
      for (int c = 0; c < 2; ++c) { pu->mv_cand[c][0] = (int16_t)((int)(lcg(&s) % 81) - 40); pu->mv_cand[c][1] = (int16_t)((int)(lcg(&s) % 81) - 40); }
      pu->extra_mv[0] = (int16_t)((int)(lcg(&s) % 33) - 16); pu->extra_mv[1] = (int16_t)((int)(lcg(&s) % 33) - 16);
      pu->num_merge_cand = (int16_t)(lcg(&s) % 6);
      for (int m = 0; m < pu->num_merge_cand; ++m) {
        pu->merge[m].mv[0] = (int16_t)((int)(lcg(&s) % 49) - 24); pu->merge[m].mv[1] = (int16_t)((int)(lcg(&s) % 49) - 24);
        pu->merge[m].usable = (uint8_t)(lcg(&s) % 4 != 0); pu->merge[m].same_ref = (uint8_t)(lcg(&s) % 2);
      }
    }
  }
  /* 1. the expected answers, one request at a time */
  const double t0 = now_s();
  for (int i = 0; i < N_PUS; ++i)
    if (kvz_hip_me_service_search(g_svc, &g_req[i], g_want[i]) != KVZ_HIP_OK) { fprintf(stderr, "search %d: %s\n", i, kvz_hip_last_error()); return 1; }
  const double t1 = now_s();
  kvz_hip_me_service_stats st0;
  kvz_hip_me_service_get_stats(g_svc, &st0);
  printf("single thread: %d requests, %.1f us per request (wait %.1f us)\n", N_PUS, (t1 - t0) * 1e6 / N_PUS, (double)st0.wait_ns / 1e3 / N_PUS);

  /* 2. the same PUs from THREADS workers at once */
  pthread_t *tid = malloc(sizeof(pthread_t) * (size_t)threads);
  worker_arg *arg = calloc((size_t)threads, sizeof(worker_arg));
  const double t2 = now_s();
  for (int i = 0; i < threads; ++i) { arg[i].id = i; pthread_create(&tid[i], NULL, worker, &arg[i]); }
  int failed = 0;
  long done = 0;
  for (int i = 0; i < threads; ++i) {
    pthread_join(tid[i], NULL);
    if (arg[i].failed) { fprintf(stderr, "FAIL: %s\n", arg[i].msg); failed = 1; }
    done += arg[i].done;
  }
  const double t3 = now_s();
  kvz_hip_me_service_stats st;
  kvz_hip_me_service_get_stats(g_svc, &st);
  const double reqs = (double)(st.requests - st0.requests), batches = (double)(st.batches - st0.batches);
  printf("%d threads: %ld requests in %.3f s = %.0f requests/s (%.0f units/s), %.2f requests per launch, largest batch %llu units, mean wait %.1f us\n",
         threads, done, t3 - t2, (double)done / (t3 - t2), (double)(st.units - st0.units) / (t3 - t2), reqs / (batches > 0 ? batches : 1),
         (unsigned long long)st.max_batch_units, (double)(st.wait_ns - st0.wait_ns) / 1e3 / (reqs > 0 ? reqs : 1));
  kvz_hip_me_service_destroy(g_svc);
  free(tex); free(plane); free(tid); free(arg);
  if (!failed) printf("service_stress ok\n");
  return failed;
}
