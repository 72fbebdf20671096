/* front_replay.c -- times kvz_hip_search_pu_batch driven the way a host that derives candidates itself has to drive it:
 * front by front in the encoder's dependency order.  Input: a binary file written by tools/front_replay.py from searches
 * RECORDED during real encodes of the reference encoder (oracle/ref_harness.c recorder): planes, the frame's PU records
 * sorted by front, the fronts' offsets, the decisions the reference took.  Plain C99 host of include/kvz_hip.h.
 *   per front: descriptors -> pinned buffer -> device, one kvz_hip_search_pu_batch, results -> host, stream sync
 *   (the next front's candidates depend on these results: inter.c:1209,1314)
 * With a 4th argument S > 1 the program instead runs S host threads, each replaying the frame front by front on its own
 * stream, planes and pinned buffers -- S encoder instances (or S frames in flight under --owf, S tiles) of one process sharing
 * the GPU through the per-thread contexts of the C ABI -- and reports the aggregate rate.
 * With a 5th argument M > 1 it runs ONE host thread that merges the same front of M sessions (M copies of the planes, as M
 * frames in flight / tiles / instances would have) into one kvz_hip_search_pu_multi_batch launch per front.
 * Prints one JSON line; every replayed result is compared with the recorded one.  NOT an encoder: candidate derivation,
 * mode decision, reconstruction are not here. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "kvz_hip.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
#define DIE(...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, " (%s)\n", kvz_hip_last_error()); return 1; } while (0)

/* ---- S sessions: one host thread each ---- */
typedef struct {
  int w, h, n, ng, repeats, hint, device;
  const uint8_t *pic, *ref;
  const kvz_hip_me_params *prm;
  const kvz_hip_me_pu *pus;
  const kvz_hip_me_result *want;
  const int32_t *off;
  pthread_barrier_t *start;
  double seconds;      /* out: wall time of this thread's `repeats` frames */
  long mismatches;     /* out */
  int failed;          /* out */
} session_t;

static void *session_main(void *arg)
{
  session_t *s = arg;
  const int w = s->w, h = s->h, n = s->n;
  s->failed = 1;
  uint8_t *d_pic = NULL, *d_ref = NULL;
  kvz_hip_me_pu *h_pus = NULL;
  kvz_hip_me_result *h_res = NULL;
  kvz_hip_stream st = NULL;
  int ready = kvz_hip_set_device(s->device) == KVZ_HIP_OK && (st = kvz_hip_stream_create()) != NULL &&
              (d_pic = kvz_hip_malloc((size_t)w * h)) && (d_ref = kvz_hip_malloc((size_t)w * h)) &&
              (h_pus = kvz_hip_malloc_host((size_t)n * sizeof(*h_pus))) && (h_res = kvz_hip_malloc_host((size_t)n * sizeof(*h_res))) &&
              !kvz_hip_memcpy_h2d(d_pic, s->pic, (size_t)w * h, st) && !kvz_hip_memcpy_h2d(d_ref, s->ref, (size_t)w * h, st) && !kvz_hip_stream_sync(st);
  for (int rep = 0; rep < s->repeats + 1; ++rep) {            /* the first pass warms up, the timed ones start together */
    if (rep == 1) pthread_barrier_wait(s->start);
    const double t0 = now_s();
    if (ready) memset(h_res, 0, (size_t)n * sizeof(*h_res));
    for (int g = 0; ready && g < s->ng; ++g) {
      const int a = s->off[g], c = s->off[g + 1] - s->off[g];
      memcpy(h_pus + a, s->pus + a, (size_t)c * sizeof(*h_pus));
      kvz_hip_me_params fp = *s->prm;
      int classes = 0;
      for (int i = a; i < a + c; ++i) {
        const int sz = s->pus[i].width > s->pus[i].height ? s->pus[i].width : s->pus[i].height;
        classes |= sz <= 16 ? 1 : (sz <= 32 ? 2 : 4);
      }
      fp.size_classes = s->hint ? classes : 0;
      if (kvz_hip_search_pu_batch(d_pic, (uint32_t)w, w, h, d_ref, (uint32_t)w, w, h, h_pus + a, (size_t)c, &fp, h_res + a, st) ||
          kvz_hip_stream_sync(st)) { fprintf(stderr, "session: %s\n", kvz_hip_last_error()); ready = 0; }
    }
    if (rep >= 1) s->seconds += now_s() - t0;
    for (int i = 0; ready && i < n; ++i)
      if (memcmp(&h_res[i], &s->want[i], 28) != 0) ++s->mismatches;
  }
  if (s->repeats < 1) pthread_barrier_wait(s->start);
  s->failed = !ready;
  kvz_hip_free(d_pic); kvz_hip_free(d_ref); kvz_hip_free_host(h_pus); kvz_hip_free_host(h_res);
  if (st) kvz_hip_stream_destroy(st);
  return NULL;
}

static int run_sessions(int S, session_t proto)
{
  pthread_t *th = calloc((size_t)S, sizeof(*th));
  session_t *ss = calloc((size_t)S, sizeof(*ss));
  pthread_barrier_t start;
  pthread_barrier_init(&start, NULL, (unsigned)S + 1);
  proto.start = &start;
  proto.device = kvz_hip_get_device();
  for (int i = 0; i < S; ++i) { ss[i] = proto; if (pthread_create(&th[i], NULL, session_main, &ss[i])) { perror("pthread_create"); return 2; } }
  pthread_barrier_wait(&start);
  const double t0 = now_s();
  for (int i = 0; i < S; ++i) pthread_join(th[i], NULL);
  const double wall = now_s() - t0;
  long mism = 0; int failed = 0; double slowest = 0;
  for (int i = 0; i < S; ++i) { mism += ss[i].mismatches; failed |= ss[i].failed; if (ss[i].seconds > slowest) slowest = ss[i].seconds; }
  const double frames = (double)S * proto.repeats;
  printf("{\"what\": \"search only, fronts, %d host threads each replaying the frame on its own stream (NOT an encoder)\", \"sessions\": %d, "
         "\"size_class_hint\": %d, \"frame\": \"%dx%d\", \"searches\": %d, \"fronts\": %d, \"frames_per_session\": %d, \"mismatches_vs_recorded\": %ld, "
         "\"wall_s\": %.4f, \"aggregate_frames_per_s\": %.2f, \"aggregate_searches_per_s\": %.0f, \"ms_per_frame_in_a_session\": %.3f, "
         "\"us_per_front_in_a_session\": %.2f, \"device\": \"%s\"}\n",
         S, S, proto.hint, proto.w, proto.h, proto.n, proto.ng, proto.repeats, mism, wall, frames / wall, frames * proto.n / wall,
         slowest * 1e3 / proto.repeats, slowest * 1e6 / proto.repeats / proto.ng, kvz_hip_device_name());
  free(th); free(ss);
  return (mism || failed) ? 1 : 0;
}

/* ---- M sessions merged into one launch per front ---- */
static int run_merged(int M, int w, int h, int n, int ng, int repeats, const uint8_t *pic, const uint8_t *ref, const kvz_hip_me_params *prm,
                      const kvz_hip_me_pu *pus, const kvz_hip_me_result *want, const int32_t *off)
{
  kvz_hip_stream st = kvz_hip_stream_create();
  const uint8_t **h_pics = malloc((size_t)M * sizeof(*h_pics)), **h_refs = malloc((size_t)M * sizeof(*h_refs));
  for (int m = 0; m < M; ++m) {
    uint8_t *a = kvz_hip_malloc((size_t)w * h), *b = kvz_hip_malloc((size_t)w * h);
    if (!a || !b || kvz_hip_memcpy_h2d(a, pic, (size_t)w * h, st) || kvz_hip_memcpy_h2d(b, ref, (size_t)w * h, st)) DIE("session planes");
    h_pics[m] = a; h_refs[m] = b;
  }
  const uint8_t **d_pics = kvz_hip_malloc((size_t)M * sizeof(*d_pics)), **d_refs = kvz_hip_malloc((size_t)M * sizeof(*d_refs));
  int max_front = 0;
  for (int g = 0; g < ng; ++g) if (off[g + 1] - off[g] > max_front) max_front = off[g + 1] - off[g];
  kvz_hip_me_pu *h_pus = kvz_hip_malloc_host((size_t)M * max_front * sizeof(*h_pus));
  kvz_hip_me_result *h_res = kvz_hip_malloc_host((size_t)M * max_front * sizeof(*h_res));
  if (!st || !d_pics || !d_refs || !h_pus || !h_res) DIE("allocation");
  if (kvz_hip_memcpy_h2d(d_pics, h_pics, (size_t)M * sizeof(*d_pics), st) || kvz_hip_memcpy_h2d(d_refs, h_refs, (size_t)M * sizeof(*d_refs), st) ||
      kvz_hip_stream_sync(st)) DIE("tables");
  double best = 1e30;
  long mismatches = 0;
  for (int rep = 0; rep < repeats + 1; ++rep) {
    const double t0 = now_s();
    for (int g = 0; g < ng; ++g) {
      const int a = off[g], c = off[g + 1] - off[g];
      int classes = 0;
      for (int m = 0; m < M; ++m)
        for (int i = 0; i < c; ++i) {
          kvz_hip_me_pu *u = &h_pus[(size_t)m * c + i];
          *u = pus[a + i];                                 /* the host "derives" every session's descriptors */
          u->pad = (int16_t)(m << 2);
          const int sz = u->width > u->height ? u->width : u->height;
          classes |= sz <= 16 ? 1 : (sz <= 32 ? 2 : 4);
        }
      kvz_hip_me_params fp = *prm;
      fp.size_classes = classes;
      if (kvz_hip_search_pu_multi_batch(d_pics, (uint32_t)w, w, h, d_refs, (uint32_t)w, w, h, M, h_pus, (size_t)M * c, &fp, h_res, st) ||
          kvz_hip_stream_sync(st)) DIE("search (merged)");
      for (int m = 0; m < M; ++m)
        for (int i = 0; i < c; ++i)
          if (memcmp(&h_res[(size_t)m * c + i], &want[a + i], 28) != 0) ++mismatches;      /* checked inside the loop: part of what a host does */
    }
    const double dt = now_s() - t0;
    if (rep > 0 && dt < best) best = dt;
  }
  printf("{\"what\": \"search only, fronts, the same front of %d sessions merged into one launch (NOT an encoder)\", \"merged_sessions\": %d, "
         "\"frame\": \"%dx%d\", \"searches_per_session\": %d, \"fronts\": %d, \"mismatches_vs_recorded\": %ld, \"ms_per_pass\": %.3f, "
         "\"us_per_front\": %.2f, \"aggregate_frames_per_s\": %.2f, \"aggregate_searches_per_s\": %.0f, \"device\": \"%s\"}\n",
         M, M, w, h, n, ng, mismatches, best * 1e3, best * 1e6 / ng, M / best, (double)M * n / best, kvz_hip_device_name());
  return mismatches ? 1 : 0;
}

int main(int argc, char **argv)
{
  if (argc < 2) { fprintf(stderr, "usage: front_replay FILE [repeats] [size-class hint 0/1] [sessions] [merged sessions]\n"); return 2; }
  const int repeats = argc > 2 ? atoi(argv[2]) : 3;
  const int hint = argc > 3 ? atoi(argv[3]) : 1;
  const int sessions = argc > 4 ? atoi(argv[4]) : 1;
  const int merged = argc > 5 ? atoi(argv[5]) : 1;
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 2; }
  int32_t hdr[4];                                     /* width, height, PUs, fronts */
  if (fread(hdr, 4, 4, f) != 4) return 2;
  const int w = hdr[0], h = hdr[1], n = hdr[2], ng = hdr[3];
  uint8_t *pic = malloc((size_t)w * h), *ref = malloc((size_t)w * h);
  kvz_hip_me_params prm;
  kvz_hip_me_pu *pus = malloc((size_t)n * sizeof(*pus));
  kvz_hip_me_result *want = malloc((size_t)n * sizeof(*want));
  int32_t *off = malloc((size_t)(ng + 1) * 4);
  if (fread(pic, 1, (size_t)w * h, f) != (size_t)w * h || fread(ref, 1, (size_t)w * h, f) != (size_t)w * h ||
      fread(&prm, sizeof(prm), 1, f) != 1 || fread(pus, sizeof(*pus), (size_t)n, f) != (size_t)n ||
      fread(want, sizeof(*want), (size_t)n, f) != (size_t)n || fread(off, 4, (size_t)(ng + 1), f) != (size_t)(ng + 1)) { fprintf(stderr, "short file\n"); return 2; }
  fclose(f);

  if (kvz_hip_init(-1) != KVZ_HIP_OK) DIE("kvz_hip_init");
  if (merged > 1) {
    if (merged > 2048 || repeats < 1) { fprintf(stderr, "merged sessions 2..2048, repeats >= 1\n"); return 2; }
    return run_merged(merged, w, h, n, ng, repeats, pic, ref, &prm, pus, want, off);
  }
  if (sessions > 1) {
    if (sessions > 64 || repeats < 1) { fprintf(stderr, "sessions 2..64, repeats >= 1\n"); return 2; }
    session_t proto = { w, h, n, ng, repeats, hint, 0, pic, ref, &prm, pus, want, off, NULL, 0.0, 0, 0 };
    return run_sessions(sessions, proto);
  }
  kvz_hip_stream st = kvz_hip_stream_create();
  uint8_t *d_pic = kvz_hip_malloc((size_t)w * h), *d_ref = kvz_hip_malloc((size_t)w * h);
  kvz_hip_me_pu *d_pus = kvz_hip_malloc((size_t)n * sizeof(*pus));
  kvz_hip_me_result *d_res = kvz_hip_malloc((size_t)n * sizeof(*want));
  kvz_hip_me_pu *h_pus = kvz_hip_malloc_host((size_t)n * sizeof(*pus));
  kvz_hip_me_result *h_res = kvz_hip_malloc_host((size_t)n * sizeof(*want));
  if (!st || !d_pic || !d_ref || !d_pus || !d_res || !h_pus || !h_res) DIE("allocation");
  if (kvz_hip_memcpy_h2d(d_pic, pic, (size_t)w * h, st) || kvz_hip_memcpy_h2d(d_ref, ref, (size_t)w * h, st) || kvz_hip_stream_sync(st)) DIE("planes");

  int max_front = 0;
  for (int g = 0; g < ng; ++g) if (off[g + 1] - off[g] > max_front) max_front = off[g + 1] - off[g];
  double best_fronts = 1e30, best_whole = 1e30, best_zc = 1e30;
  long mismatches = 0;
  for (int rep = 0; rep < repeats + 1; ++rep) {              /* the first pass warms up */
    /* (a) front by front */
    memset(h_res, 0, (size_t)n * sizeof(*want));
    double t0 = now_s();
    for (int g = 0; g < ng; ++g) {
      const int a = off[g], c = off[g + 1] - off[g];
      memcpy(h_pus + a, pus + a, (size_t)c * sizeof(*pus));                      /* the host "derives" the front's descriptors */
      if (kvz_hip_memcpy_h2d(d_pus + a, h_pus + a, (size_t)c * sizeof(*pus), st)) DIE("h2d");
      /* the searches of one front sit at the same place of their LCUs' quadtree walk, i.e. mostly have one size (the ragged last LCU
       * row walks a smaller tree): name the size classes present, so that one kernel is launched instead of three
       * (kvz_hip_me_params.size_classes) */
      kvz_hip_me_params fp = prm;
      int classes = 0;
      for (int i = a; i < a + c; ++i) {
        const int sz = pus[i].width > pus[i].height ? pus[i].width : pus[i].height;
        classes |= sz <= 16 ? 1 : (sz <= 32 ? 2 : 4);
      }
      fp.size_classes = hint ? classes : 0;
      if (kvz_hip_search_pu_batch(d_pic, (uint32_t)w, w, h, d_ref, (uint32_t)w, w, h, d_pus + a, (size_t)c, &fp, d_res + a, st)) DIE("search");
      if (kvz_hip_memcpy_d2h(h_res + a, d_res + a, (size_t)c * sizeof(*want), st)) DIE("d2h");     /* syncs the stream */
    }
    double dt = now_s() - t0;
    if (rep > 0 && dt < best_fronts) best_fronts = dt;
    for (int i = 0; i < n; ++i)
      if (memcmp(&h_res[i], &want[i], 28) != 0) ++mismatches;                    /* mv, cost, bitcost, merged, merge_idx, mv_cand */
    /* (a') the same loop with the kernels reading the descriptors from, and writing the results to, the pinned host buffers
     * directly (page-locked memory is device-visible): no copies, one launch + one stream sync per front */
    memset(h_res, 0, (size_t)n * sizeof(*want));
    t0 = now_s();
    for (int g = 0; g < ng; ++g) {
      const int a = off[g], c = off[g + 1] - off[g];
      memcpy(h_pus + a, pus + a, (size_t)c * sizeof(*pus));
      kvz_hip_me_params fp = prm;
      int classes = 0;
      for (int i = a; i < a + c; ++i) {
        const int sz = pus[i].width > pus[i].height ? pus[i].width : pus[i].height;
        classes |= sz <= 16 ? 1 : (sz <= 32 ? 2 : 4);
      }
      fp.size_classes = hint ? classes : 0;
      if (kvz_hip_search_pu_batch(d_pic, (uint32_t)w, w, h, d_ref, (uint32_t)w, w, h, h_pus + a, (size_t)c, &fp, h_res + a, st)) DIE("search (zero copy)");
      if (kvz_hip_stream_sync(st)) DIE("sync");
    }
    dt = now_s() - t0;
    if (rep > 0 && dt < best_zc) best_zc = dt;
    for (int i = 0; i < n; ++i)
      if (memcmp(&h_res[i], &want[i], 28) != 0) ++mismatches;
    /* (b) the whole frame in one launch (no dependency order: what the kernel can do when every candidate is known) */
    memcpy(h_pus, pus, (size_t)n * sizeof(*pus));
    t0 = now_s();
    if (kvz_hip_memcpy_h2d(d_pus, h_pus, (size_t)n * sizeof(*pus), st)) DIE("h2d");
    if (kvz_hip_search_pu_batch(d_pic, (uint32_t)w, w, h, d_ref, (uint32_t)w, w, h, d_pus, (size_t)n, &prm, d_res, st)) DIE("search");
    if (kvz_hip_memcpy_d2h(h_res, d_res, (size_t)n * sizeof(*want), st)) DIE("d2h");
    dt = now_s() - t0;
    if (rep > 0 && dt < best_whole) best_whole = dt;
    for (int i = 0; i < n; ++i)
      if (memcmp(&h_res[i], &want[i], 28) != 0) ++mismatches;
  }
  printf("{\"what\": \"search only, fronts (NOT an encoder)\", \"size_class_hint\": %d, \"frame\": \"%dx%d\", \"searches\": %d, \"fronts\": %d, \"largest_front\": %d, \"mismatches_vs_recorded\": %ld, "
         "\"fronts_ms_per_frame\": %.3f, \"fronts_searches_per_s\": %.0f, \"fronts_frames_per_s\": %.2f, \"us_per_front\": %.2f, "
         "\"zero_copy_fronts_ms_per_frame\": %.3f, \"zero_copy_fronts_frames_per_s\": %.2f, \"zero_copy_us_per_front\": %.2f, "
         "\"one_launch_ms_per_frame\": %.3f, \"one_launch_searches_per_s\": %.0f, \"device\": \"%s\"}\n",
         hint, w, h, n, ng, max_front, mismatches, best_fronts * 1e3, n / best_fronts, 1.0 / best_fronts, best_fronts * 1e6 / ng,
         best_zc * 1e3, 1.0 / best_zc, best_zc * 1e6 / ng,
         best_whole * 1e3, n / best_whole, kvz_hip_device_name());
  return mismatches ? 1 : 0;
}
