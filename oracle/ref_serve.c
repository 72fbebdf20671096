/* ref_serve.c -- TEST / MEASUREMENT INFRASTRUCTURE, part of oracle/_ref/libkvzref.so (the reference compiled from where it
 * lies under /root/reference by oracle/Makefile; nothing of it is copied here).  Not part of the product: kvazaar_amd/
 * neither links nor loads it; this file dlopen()s the product.
 *
 * The reference encoder's inter searches answered by the product's SEARCH SERVICE (include/kvz_hip.h,
 * kvazaar_amd/csrc/serve.hip) while the encoder runs with its own thread pool: every threadqueue worker
 * (threadqueue.c:263; one CTU job each, encoderstate.c:777-828, frames in flight under --owf) that reaches
 * kvz_search_cu_inter (search_inter.c:1587; -Wl,--wrap) posts ONE request -- the PU with the candidates of every
 * reference picture of search_pu_inter's loop (:1502-1507), derived by the encoder's own kvz_inter_get_merge_cand /
 * kvz_inter_get_mv_cand on the worker -- and blocks until the service answers; the decision is written into cur_cu exactly
 * where search_pu_inter_ref writes it (:1275-1290).  This is what the few lines of glue inside Kvazaar would do
 * (INTEGRATION.md section 6).
 *
 * Pictures.  The service keeps luma planes in numbered slots.  A source picture is uploaded whole by the first worker that
 * needs it.  A reference picture may still be under reconstruction (--owf with WPP): the reference lets a search read only
 * what fracmv_within_tile (search_inter.c:87-176) proves final -- for a PU of LCU (lx, ly): LCU row ly + 1 - k up to LCU
 * column lx + 1 + k, less the in-loop filters' delay -- because the CTU job depends on LCU (lx + 1, ly + 1) of the
 * previous frame (encoderstate.c:795-808) and WPP order implies the rest.  The worker uploads exactly that staircase
 * (what is not on the device yet) before its first request of a CTU: the same pixels the CPU search could have read. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "global.h"
#include "encoder.h"
#include "encoderstate.h"
#include "cu.h"
#include "image.h"
#include "inter.h"
#include "search.h"
#include "search_inter.h"
#include "videoframe.h"

#include "../include/kvz_hip.h"

void __real_kvz_search_cu_inter(encoder_state_t * const state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost);
void kvz_cu_cost_inter_rd2(encoder_state_t * const state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost);   /* search_inter.c:1545, not in a header */

enum { SVC_SLOTS = 96, SVC_MAX_BANDS = 80 };

typedef struct {
  const kvz_picture *key;
  int32_t poc, kind;                 /* kind 0: source picture, 1: reconstructed picture */
  int used;
  uint64_t last_use;
  int complete;                      /* the whole plane is on the device */
  int32_t cols[SVC_MAX_BANDS];       /* columns on the device, per band of rows (LCU rows shifted up by the filter delay) */
  pthread_mutex_t mu;
} svc_slot_t;

static struct {
  int on;
  void *lib;
  kvz_hip_me_service *svc;
  kvz_hip_me_service *(*create)(const kvz_hip_me_service_config *);
  void (*destroy)(kvz_hip_me_service *);
  int (*put_rect)(kvz_hip_me_service *, int, const kvz_hip_pixel *, uint32_t, int, int, int, int);
  int (*search)(kvz_hip_me_service *, const kvz_hip_me_request *, kvz_hip_me_result *);
  int (*get_stats)(kvz_hip_me_service *, kvz_hip_me_service_stats *);
  int (*init)(int);
  const char *(*last_error)(void);
  int w, h, min_size, shadow, probe;
  int upload_only;                      /* flags bit 5: the pictures travel as usual, every search stays with the reference (what the uploads alone cost an encode) */
  int table_range;                      /* > 0: SAD-table mode (the search stays with the reference, kvz_image_calc_sad is answered from tables) */
  const uint32_t *(*sad_tables)(kvz_hip_me_service *, int, int, const int32_t *, int, int, int);
  long long tab_hits, tab_range_misses, tab_other, tab_ns;
  long long probe_ns[4], probe_n[4];
  long long spec_n[4], spec_hit[4];     /* speculation probe */
  int spec_probe;    /* probe mode: the reference's own search timed per CU size 64, 32, 16, 8 */
  pthread_mutex_t table_mu;
  svc_slot_t slots[SVC_SLOTS];
  uint64_t clock;
  long served, passed_on, failed, shadow_mismatch, upload_rects;
  long long search_ns, upload_ns, cand_ns;
} g_svc;

static long long svc_now_ns(void)
{
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (long long)t.tv_sec * 1000000000ll + t.tv_nsec;
}

/* flags: bits 8..15 = SAD-table mode with that range (1..32): nothing of the search is served, but at the start of each CTU the
 * worker fetches kvz_hip_me_service_sad_tables for every reference picture and the reference's own kvz_image_calc_sad calls
 * (check_mv_cost, search_inter.c:200) are answered from them; a call the table does not hold runs the reference's function;
 * bit 1 = probe mode (nothing is served: the reference's own searches are timed per CU size);
 * bit 0 = shadow mode (every served search is repeated by the reference's own search and compared; the
 * reference's result is kept), min_size = smallest PU width that is served (smaller ones run the reference's search) */
int ref_service_begin(const char *lib_path, int w, int h, int max_threads, int min_size, int flags)
{
  memset(&g_svc, 0, sizeof(g_svc));
  if (flags & 16) {           /* host only: the timing and speculation probes, no device library, nothing served */
    g_svc.w = w; g_svc.h = h; g_svc.min_size = min_size; g_svc.probe = 1; g_svc.spec_probe = (flags >> 3) & 1;
    __atomic_store_n(&g_svc.on, 1, __ATOMIC_RELEASE);
    return 0;
  }
  void *l = dlopen(lib_path, RTLD_NOW | RTLD_GLOBAL);
  if (!l) { fprintf(stderr, "dlopen %s: %s\n", lib_path, dlerror()); return -1; }
  g_svc.lib = l;
  *(void **)&g_svc.create = dlsym(l, "kvz_hip_me_service_create");
  *(void **)&g_svc.destroy = dlsym(l, "kvz_hip_me_service_destroy");
  *(void **)&g_svc.put_rect = dlsym(l, "kvz_hip_me_service_put_rect");
  *(void **)&g_svc.search = dlsym(l, "kvz_hip_me_service_search");
  *(void **)&g_svc.get_stats = dlsym(l, "kvz_hip_me_service_get_stats");
  *(void **)&g_svc.sad_tables = dlsym(l, "kvz_hip_me_service_sad_tables");
  *(void **)&g_svc.init = dlsym(l, "kvz_hip_init");
  *(void **)&g_svc.last_error = dlsym(l, "kvz_hip_last_error");
  if (!g_svc.create || !g_svc.destroy || !g_svc.put_rect || !g_svc.search || !g_svc.get_stats || !g_svc.init || !g_svc.last_error || !g_svc.sad_tables) return -1;
  if (g_svc.init(-1) != KVZ_HIP_OK) { fprintf(stderr, "kvz_hip_init: %s\n", g_svc.last_error()); return -1; }
  if ((h + 63) / 64 + 1 > SVC_MAX_BANDS) return -1;
  kvz_hip_me_service_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.width = w; cfg.height = h; cfg.max_pictures = SVC_SLOTS; cfg.max_threads = max_threads + 8;
  g_svc.svc = g_svc.create(&cfg);
  if (!g_svc.svc) { fprintf(stderr, "kvz_hip_me_service_create: %s\n", g_svc.last_error()); return -1; }
  g_svc.w = w; g_svc.h = h; g_svc.min_size = min_size; g_svc.shadow = flags & 1; g_svc.probe = (flags >> 1) & 1;
  g_svc.upload_only = (flags >> 5) & 1;
  g_svc.table_range = (flags >> 8) & 0xff;
  g_svc.spec_probe = (flags >> 3) & 1;
  pthread_mutex_init(&g_svc.table_mu, NULL);
  for (int i = 0; i < SVC_SLOTS; ++i) pthread_mutex_init(&g_svc.slots[i].mu, NULL);
  __atomic_store_n(&g_svc.on, 1, __ATOMIC_RELEASE);
  return 0;
}

/* out[24..30]: SAD-table mode: lookups answered, lookups outside the range, other calls, ns in the wrapper's misses (unused),
 * tables fetched, their bytes, ns spent fetching them.  out[16..19] / out[20..23]: probe mode's summed nanoseconds / number of searches for CU sizes 64, 32, 16, 8.
 * out[0..15]: served, passed on, failed, shadow mismatches, upload rects, search wait ns, upload ns, candidate ns,
 * then the service's own statistics: requests, units, batches, launches, max_batch_units, rects, rect_bytes, wait_ns
 * (with the speculation probe and no service: out[8..11] searches compared, out[12..15] searches whose CTU-start candidates were the real ones). */
void ref_service_end(long long *out)
{
  __atomic_store_n(&g_svc.on, 0, __ATOMIC_RELEASE);
  if (out) {
    memset(out, 0, 32 * sizeof(out[0]));
    for (int i = 0; i < 4; ++i) { out[16 + i] = g_svc.probe_ns[i]; out[20 + i] = g_svc.probe_n[i]; }
    if (g_svc.spec_probe) for (int i = 0; i < 4; ++i) { out[8 + i] = g_svc.spec_n[i]; out[12 + i] = g_svc.spec_hit[i]; }   /* probe-only runs: in place of the service's statistics */
    out[24] = g_svc.tab_hits; out[25] = g_svc.tab_range_misses; out[26] = g_svc.tab_other; out[27] = g_svc.tab_ns;
    if (g_svc.svc) {
      kvz_hip_me_service_stats st2;
      if (g_svc.get_stats(g_svc.svc, &st2) == KVZ_HIP_OK) { out[28] = (long long)st2.tables; out[29] = (long long)st2.table_bytes; out[30] = (long long)st2.table_ns; }
    }
    out[0] = g_svc.served; out[1] = g_svc.passed_on; out[2] = g_svc.failed; out[3] = g_svc.shadow_mismatch; out[4] = g_svc.upload_rects;
    out[5] = g_svc.search_ns; out[6] = g_svc.upload_ns; out[7] = g_svc.cand_ns;
    kvz_hip_me_service_stats st;
    if (g_svc.svc && g_svc.get_stats(g_svc.svc, &st) == KVZ_HIP_OK) {
      out[8] = (long long)st.requests; out[9] = (long long)st.units; out[10] = (long long)st.batches; out[11] = (long long)st.launches;
      out[12] = (long long)st.max_batch_units; out[13] = (long long)st.rects; out[14] = (long long)st.rect_bytes; out[15] = (long long)st.wait_ns;
    }
  }
  if (g_svc.svc) g_svc.destroy(g_svc.svc);
  memset(&g_svc, 0, sizeof(g_svc));
}

/* the slot of a picture (created empty on first sight; the least recently used one is recycled) */
static svc_slot_t *svc_slot_of(const kvz_picture *pic, int32_t poc, int kind, int *index)
{
  pthread_mutex_lock(&g_svc.table_mu);
  int found = -1, lru = -1;
  for (int i = 0; i < SVC_SLOTS; ++i) {
    svc_slot_t *s = &g_svc.slots[i];
    if (s->used && s->key == pic && s->poc == poc && s->kind == kind) { found = i; break; }
    if (!s->used) { if (lru < 0 || g_svc.slots[lru].used) lru = i; }
    else if (lru < 0 || (g_svc.slots[lru].used && s->last_use < g_svc.slots[lru].last_use)) lru = i;
  }
  if (found < 0) {
    found = lru;
    svc_slot_t *s = &g_svc.slots[found];
    pthread_mutex_lock(&s->mu);
    s->key = pic; s->poc = poc; s->kind = kind; s->used = 1; s->complete = 0;
    memset(s->cols, 0, sizeof(s->cols));
    pthread_mutex_unlock(&s->mu);
  }
  g_svc.slots[found].last_use = ++g_svc.clock;
  pthread_mutex_unlock(&g_svc.table_mu);
  *index = found;
  return &g_svc.slots[found];
}

static int svc_upload_whole(svc_slot_t *s, int index, const kvz_picture *pic)
{
  int bad = 0;
  if (__atomic_load_n(&s->complete, __ATOMIC_ACQUIRE)) return 0;
  pthread_mutex_lock(&s->mu);
  if (!s->complete) {
    bad = g_svc.put_rect(g_svc.svc, index, pic->y, (uint32_t)pic->stride, 0, 0, g_svc.w, g_svc.h) != KVZ_HIP_OK;
    __atomic_add_fetch(&g_svc.upload_rects, 1, __ATOMIC_RELAXED);
    if (!bad) __atomic_store_n(&s->complete, 1, __ATOMIC_RELEASE);
  }
  pthread_mutex_unlock(&s->mu);
  return bad;
}

/* what a search of LCU (lx, ly) may read of a picture that is still being reconstructed (see the file header) */
static int svc_upload_staircase(svc_slot_t *s, int index, const kvz_picture *pic, int lx, int ly, int delay, int down, int right)
{
  if (__atomic_load_n(&s->complete, __ATOMIC_ACQUIRE)) return 0;
  const int W = g_svc.w, H = g_svc.h;
  const int lcu_cols = (W + 63) / 64, lcu_rows = (H + 63) / 64;
  int bad = 0;
  pthread_mutex_lock(&s->mu);
  int all = 1;
  for (int b = 0; b < lcu_rows; ++b) {
    /* band b = LCU row b less the delay: LCU row b done up to LCU column X makes columns < (X + 1) * 64 - delay of it final */
    const int up = ly + down - b;                         /* LCU rows above the lowest one allowed */
    int need = 0;
    if (up >= 0) {
      const int X = lx + right + up;                      /* mv_lcu.x + mv_lcu.y <= down + right */
      need = X >= lcu_cols - 1 ? W : (X + 1) * 64 - delay;
      if (need > W) need = W;
    }
    if (need > s->cols[b] && !bad) {
      const int y0 = b == 0 ? 0 : b * 64 - delay, y1 = b == lcu_rows - 1 ? H : (b + 1) * 64 - delay;
      if (y1 > y0) {
        bad = g_svc.put_rect(g_svc.svc, index, pic->y + (size_t)y0 * pic->stride + s->cols[b], (uint32_t)pic->stride, s->cols[b], y0,
                             need - s->cols[b], y1 - y0) != KVZ_HIP_OK;
        __atomic_add_fetch(&g_svc.upload_rects, 1, __ATOMIC_RELAXED);
      }
      if (!bad) s->cols[b] = need;
    }
    if (s->cols[b] < W) all = 0;
  }
  if (all && !bad) __atomic_store_n(&s->complete, 1, __ATOMIC_RELEASE);
  pthread_mutex_unlock(&s->mu);
  return bad;
}

static int svc_can_serve(const encoder_state_t *state, int width)
{
  const encoder_control_t *ctrl = state->encoder_control;
  const encoder_state_config_frame_t *fr = state->frame;
  const int nref = (int)fr->ref->used_size;
  if (nref < 1 || nref > KVZ_HIP_SERVICE_MAX_REFS || fr->slicetype == KVZ_SLICE_I || ctrl->cfg.mv_rdo) return 0;
  if (fr->slicetype == KVZ_SLICE_B && ctrl->cfg.bipred) return 0;       /* search_pu_inter_bipred stays with the reference */
  if (ctrl->in.width != g_svc.w || ctrl->in.height != g_svc.h) return 0;
  if (state->tile->offset_x != 0 || state->tile->offset_y != 0 || state->tile->frame->width != g_svc.w || state->tile->frame->height != g_svc.h) return 0;
  if (width < g_svc.min_size) return 0;
  return 1;
}

/* What search_pu_inter (:1492-1500) and search_pu_inter_ref (:1143-1206) derive for a 2Nx2N PU before the searches, for every reference
 * picture, with the encoder's own functions on the given lcu: the merge list as calc_mvd_cost sees it, the AMVP pair and the start
 * vector of each picture.  Returns non-zero when a picture is in neither list. */
static int svc_derive_pus(encoder_state_t *state, int x, int y, int width, lcu_t *lcu, kvz_hip_me_pu *pus, int8_t *ref_list_of, int8_t *lx_idx_of)
{
  const encoder_state_config_frame_t *fr = state->frame;
  const int nref = (int)fr->ref->used_size;
  cu_info_t *cur_cu = LCU_GET_CU_AT_PX(lcu, SUB_SCU(x), SUB_SCU(y));
  inter_merge_cand_t merge[MRG_MAX_NUM_CANDS];
  const int n_merge = kvz_inter_get_merge_cand(state, x, y, width, width, true, true, merge, lcu);
  CU_SET_MV_CAND(cur_cu, 0, 0);
  CU_SET_MV_CAND(cur_cu, 1, 0);
  const int8_t lx_max = MAX(fr->ref_LX_size[0], fr->ref_LX_size[1]);
  for (int ref_idx = 0; ref_idx < nref; ++ref_idx) {
    int8_t ref_list = -1, LX_idx;
    for (LX_idx = 0; LX_idx < lx_max; LX_idx++) {
      if (LX_idx < fr->ref_LX_size[0] && fr->ref_LX[0][LX_idx] == ref_idx) { ref_list = 0; break; }
      if (LX_idx < fr->ref_LX_size[1] && fr->ref_LX[1][LX_idx] == ref_idx) { ref_list = 1; break; }
    }
    if (ref_list < 0) return 1;
    ref_list_of[ref_idx] = ref_list; lx_idx_of[ref_idx] = LX_idx;
    kvz_hip_me_pu *pu = &pus[ref_idx];
    memset(pu, 0, sizeof(*pu));
    pu->x = x; pu->y = y; pu->width = width; pu->height = width;
    pu->num_merge_cand = (int16_t)n_merge;
    for (int i = 0; i < n_merge; ++i) {
      const int dir = merge[i].dir;
      pu->merge[i].usable = dir != 3;
      if (dir != 3) {
        pu->merge[i].mv[0] = merge[i].mv[dir - 1][0]; pu->merge[i].mv[1] = merge[i].mv[dir - 1][1];
        pu->merge[i].same_ref = fr->ref_LX[dir - 1][merge[i].ref[dir - 1]] == ref_idx;
      }
    }
    /* :1170-1187: the AMVP pair of this picture; cur_cu->inter.mv_ref is borrowed for the call and put back */
    const int8_t temp = cur_cu->inter.mv_ref[ref_list];
    cur_cu->inter.mv_ref[ref_list] = LX_idx;
    kvz_inter_get_mv_cand(state, x, y, width, width, pu->mv_cand, cur_cu, lcu, ref_list);
    cur_cu->inter.mv_ref[ref_list] = temp;
    /* :1190-1206 */
    const cu_info_t *ref_cu = kvz_cu_array_at_const(fr->ref->cu_arrays[ref_idx], state->tile->offset_x + x + (width >> 1),
                                                    state->tile->offset_y + y + (width >> 1));
    if (ref_cu->type == CU_INTER) {
      const int l = (ref_cu->inter.mv_dir & 1) ? 0 : 1;
      pu->extra_mv[0] = ref_cu->inter.mv[l][0]; pu->extra_mv[1] = ref_cu->inter.mv[l][1];
    }
  }
  return 0;
}

/* Speculation probe (flags bit 3; nothing is served or changed): how often would the candidates of a search have been known when its CTU
 * STARTED?  At the first search of a CTU the worker keeps a copy of the lcu (inside the CTU nothing is decided yet); at every search the
 * candidates are derived twice -- on the real lcu and on that copy -- and compared.  Equal = a search issued at CTU start would have been
 * the right one.  Counted per CU size. */
static __thread lcu_t t_spec_lcu;
static __thread struct { int32_t poc, lx, ly; int valid; } t_spec_ctu;
static void svc_spec_probe(encoder_state_t *state, int x, int y, int depth, lcu_t *lcu)
{
  const encoder_state_config_frame_t *fr = state->frame;
  const int nref = (int)fr->ref->used_size;
  if (fr->slicetype == KVZ_SLICE_I || nref < 1 || nref > KVZ_HIP_SERVICE_MAX_REFS) return;
  const int width = LCU_WIDTH >> depth, lx = x / LCU_WIDTH, ly = y / LCU_WIDTH;
  if (!t_spec_ctu.valid || t_spec_ctu.poc != fr->poc || t_spec_ctu.lx != lx || t_spec_ctu.ly != ly) {
    if (depth != 0) { t_spec_ctu.valid = 0; return; }     /* a CTU whose first search is not the 64x64 one (picture edge): not speculated */
    memcpy(&t_spec_lcu, lcu, sizeof(lcu_t));
    t_spec_ctu.poc = fr->poc; t_spec_ctu.lx = lx; t_spec_ctu.ly = ly; t_spec_ctu.valid = 1;
  }
  kvz_hip_me_pu real[KVZ_HIP_SERVICE_MAX_REFS], spec[KVZ_HIP_SERVICE_MAX_REFS];
  int8_t a[KVZ_HIP_SERVICE_MAX_REFS], b[KVZ_HIP_SERVICE_MAX_REFS];
  cu_info_t *cur_cu = LCU_GET_CU_AT_PX(lcu, SUB_SCU(x), SUB_SCU(y));
  const cu_info_t saved = *cur_cu;
  const int bad = svc_derive_pus(state, x, y, width, lcu, real, a, b);
  *cur_cu = saved;
  if (bad || svc_derive_pus(state, x, y, width, &t_spec_lcu, spec, a, b)) return;
  const int same = !memcmp(real, spec, sizeof(real[0]) * (size_t)nref);
  __atomic_add_fetch(&g_svc.spec_n[depth & 3], 1, __ATOMIC_RELAXED);
  if (same) __atomic_add_fetch(&g_svc.spec_hit[depth & 3], 1, __ATOMIC_RELAXED);
}

/* Is the picture with this POC possibly still being reconstructed?  The frames in flight are the ones the other encoder
 * states hold (kvazaar.c:115-125: a ring of owf + 1 states, each linked to the state of the frame coded before it). */
static int svc_in_flight(const encoder_state_t *state, int32_t poc)
{
  const encoder_state_t *s = state;
  for (int i = 0; i < state->encoder_control->cfg.owf; ++i) {
    s = s->previous_encoder_state;
    if (!s || s == state) break;
    if (s->frame->poc == poc) return 1;
  }
  return 0;
}

/* SAD-table mode: the calling worker's tables for the CTU it is searching */
static __thread struct {
  const uint32_t *tab;
  const kvz_picture *pic, *refs[KVZ_HIP_SERVICE_MAX_REFS];
  int n_refs, ctu_x, ctu_y, range, valid;
  long long hits, range_misses, other;
} t_tab;

unsigned __real_kvz_image_calc_sad(const kvz_picture *pic, const kvz_picture *ref, int pic_x, int pic_y, int ref_x, int ref_y, int block_width, int block_height);

/* kvz_image_calc_sad (image.c:455-486; -Wl,--wrap): the one call site is check_mv_cost (search_inter.c:200) */
unsigned __wrap_kvz_image_calc_sad(const kvz_picture *pic, const kvz_picture *ref, int pic_x, int pic_y, int ref_x, int ref_y, int block_width, int block_height)
{
  if (t_tab.valid && pic == t_tab.pic && block_width == block_height && (pic_x & ~63) == t_tab.ctu_x && (pic_y & ~63) == t_tab.ctu_y) {
    const int n = block_width, bx = pic_x & 63, by = pic_y & 63;
    int k = -1;
    if (!((bx | by) & (n - 1))) {
      if (n == 8) k = 21 + (by >> 3) * 8 + (bx >> 3);
      else if (n == 16) k = 5 + (by >> 4) * 4 + (bx >> 4);
      else if (n == 32) k = 1 + (by >> 5) * 2 + (bx >> 5);
      else if (n == 64) k = 0;
    }
    if (k >= 0) {
      const int R = t_tab.range, side = 2 * R + 1, dx = ref_x - pic_x, dy = ref_y - pic_y;
      if (dx >= -R && dx <= R && dy >= -R && dy <= R) {
        for (int i = 0; i < t_tab.n_refs; ++i)
          if (t_tab.refs[i] == ref) {
            const uint32_t v = t_tab.tab[(((size_t)i * side + (size_t)(dy + R)) * side + (size_t)(dx + R)) * KVZ_HIP_CTU_PUS + k];
            if (v != 0xffffffffu) { ++t_tab.hits; return v; }
            break;
          }
      } else {
        ++t_tab.range_misses;
        return __real_kvz_image_calc_sad(pic, ref, pic_x, pic_y, ref_x, ref_y, block_width, block_height);
      }
    }
  }
  ++t_tab.other;
  return __real_kvz_image_calc_sad(pic, ref, pic_x, pic_y, ref_x, ref_y, block_width, block_height);
}

static void svc_flush_table_counters(void)
{
  __atomic_add_fetch(&g_svc.tab_hits, t_tab.hits, __ATOMIC_RELAXED);
  __atomic_add_fetch(&g_svc.tab_range_misses, t_tab.range_misses, __ATOMIC_RELAXED);
  __atomic_add_fetch(&g_svc.tab_other, t_tab.other, __ATOMIC_RELAXED);
  t_tab.hits = t_tab.range_misses = t_tab.other = 0;
}

/* the last CTU this worker uploaded a staircase for */
static __thread struct { int32_t poc, lx, ly; int valid; int32_t pic_slot, ref_slot[KVZ_HIP_SERVICE_MAX_REFS]; } t_last_ctu;

/* kvz_search_cu_inter's 2Nx2N PU through the service.  Returns 1 when served. */
int svc_serve_cu_inter(encoder_state_t *state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost)
{
  if (!__atomic_load_n(&g_svc.on, __ATOMIC_ACQUIRE)) return 0;
  const int width = LCU_WIDTH >> depth;
  if (g_svc.probe) {
    const long long p0 = svc_now_ns();
    if (g_svc.spec_probe) svc_spec_probe(state, x, y, depth, lcu);
    __real_kvz_search_cu_inter(state, x, y, depth, lcu, inter_cost, inter_bitcost);
    __atomic_add_fetch(&g_svc.probe_ns[depth & 3], svc_now_ns() - p0, __ATOMIC_RELAXED);
    __atomic_add_fetch(&g_svc.probe_n[depth & 3], 1, __ATOMIC_RELAXED);
    return 1;
  }
  if (!svc_can_serve(state, g_svc.table_range > 0 ? 64 : width)) { t_tab.valid = 0; __atomic_add_fetch(&g_svc.passed_on, 1, __ATOMIC_RELAXED); return 0; }
  const encoder_control_t *ctrl = state->encoder_control;
  const encoder_state_config_frame_t *fr = state->frame;
  const int nref = (int)fr->ref->used_size;
  int bad = 0;
  long long t0 = svc_now_ns();

  /* ---- pictures ---- */
  kvz_hip_me_request req;
  memset(&req, 0, sizeof(req));
  const kvz_picture *src = state->tile->frame->source;
  req.n_refs = nref;
  const int wpp_owf = ctrl->cfg.owf && ctrl->cfg.wpp;
  const int delay = ctrl->cfg.sao_type ? SAO_DELAY_PX : (ctrl->cfg.deblock_enable ? DEBLOCK_DELAY_PX : 0);
  const int lx = x / LCU_WIDTH, ly = y / LCU_WIDTH;
  const int new_ctu = !t_last_ctu.valid || t_last_ctu.poc != fr->poc || t_last_ctu.lx != lx || t_last_ctu.ly != ly;
  if (new_ctu) {
    /* once per CTU and worker: where the pictures live on the device, and whatever of them has become final since (the slots are
     * remembered for the CTU's other searches: the table lookups take a lock) */
    int idx;
    svc_slot_t *s = svc_slot_of(src, fr->poc, 0, &idx);
    bad |= svc_upload_whole(s, idx, src);
    t_last_ctu.pic_slot = idx;
    for (int i = 0; i < nref && !bad; ++i) {
      const kvz_picture *ref = fr->ref->images[i];
      s = svc_slot_of(ref, fr->ref->pocs[i], 1, &idx);
      t_last_ctu.ref_slot[i] = idx;
      if (__atomic_load_n(&s->complete, __ATOMIC_ACQUIRE)) continue;
      if (!wpp_owf || !svc_in_flight(state, fr->ref->pocs[i])) bad |= svc_upload_whole(s, idx, ref);
      else bad |= svc_upload_staircase(s, idx, ref, lx, ly, delay, ctrl->max_inter_ref_lcu.down, ctrl->max_inter_ref_lcu.right);
    }
    t_last_ctu.poc = fr->poc; t_last_ctu.lx = lx; t_last_ctu.ly = ly; t_last_ctu.valid = !bad;
  }
  req.pic_slot = t_last_ctu.pic_slot;
  for (int i = 0; i < nref; ++i) req.ref_slot[i] = t_last_ctu.ref_slot[i];
  long long t1 = svc_now_ns();
  if (g_svc.upload_only) {
    if (new_ctu) __atomic_add_fetch(&g_svc.upload_ns, t1 - t0, __ATOMIC_RELAXED);
    __atomic_add_fetch(&g_svc.passed_on, 1, __ATOMIC_RELAXED);
    return 0;
  }
  if (g_svc.table_range > 0) {
    /* SAD-table mode: the search stays with the reference; a new CTU gets its tables (every reference picture, +-range) first */
    if (new_ctu || !t_tab.valid) {
      svc_flush_table_counters();
      t_tab.valid = 0;
      if (!bad) {
        const uint32_t *tab = g_svc.sad_tables(g_svc.svc, req.pic_slot, nref, req.ref_slot, lx * LCU_WIDTH, ly * LCU_WIDTH, g_svc.table_range);
        if (tab) {
          t_tab.tab = tab; t_tab.pic = src; t_tab.n_refs = nref; t_tab.ctu_x = lx * LCU_WIDTH; t_tab.ctu_y = ly * LCU_WIDTH; t_tab.range = g_svc.table_range;
          for (int i = 0; i < nref; ++i) t_tab.refs[i] = fr->ref->images[i];
          t_tab.valid = 1;
        } else if (__atomic_fetch_add(&g_svc.failed, 1, __ATOMIC_RELAXED) == 0) fprintf(stderr, "sad_tables: %s\n", g_svc.last_error());
      }
      __atomic_add_fetch(&g_svc.upload_ns, t1 - t0, __ATOMIC_RELAXED);
    }
    __atomic_add_fetch(&g_svc.passed_on, 1, __ATOMIC_RELAXED);
    return 0;
  }

  /* ---- the request: search_pu_inter :1492-1500, then per picture search_pu_inter_ref :1143-1206 ---- */
  cu_info_t *cur_cu = LCU_GET_CU_AT_PX(lcu, SUB_SCU(x), SUB_SCU(y));
  const cu_info_t saved = *cur_cu;
  int8_t ref_list_of[KVZ_HIP_SERVICE_MAX_REFS], lx_idx_of[KVZ_HIP_SERVICE_MAX_REFS];
  bad |= svc_derive_pus(state, x, y, width, lcu, req.pu, ref_list_of, lx_idx_of);
  kvz_hip_me_params *p = &req.params;
  p->lambda_cost = (int32_t)(state->lambda_sqrt + 0.5);
  p->early_termination = ctrl->cfg.me_early_termination;
  p->max_steps = ctrl->cfg.me_max_steps;
  p->fme_level = ctrl->cfg.fme_level;
  p->wpp_owf = wpp_owf;
  p->ref_delay_px = delay;
  p->max_ref_lcu_down = ctrl->max_inter_ref_lcu.down; p->max_ref_lcu_right = ctrl->max_inter_ref_lcu.right;
  switch (ctrl->cfg.ime_algorithm) {
    case KVZ_IME_DIA: p->algorithm = 1; break;
    case KVZ_IME_TZ: p->algorithm = 2; break;
    case KVZ_IME_FULL64: p->algorithm = 3; p->search_range = 64; break;
    case KVZ_IME_FULL32: case KVZ_IME_FULL: p->algorithm = 3; p->search_range = 32; break;
    case KVZ_IME_FULL16: p->algorithm = 3; p->search_range = 16; break;
    case KVZ_IME_FULL8: p->algorithm = 3; p->search_range = 8; break;
    default: p->algorithm = 0; break;
  }
  p->mv_constraint = ctrl->cfg.mv_constraint;
  req.cost_to_beat = (uint32_t)MAX_INT;
  long long t2 = svc_now_ns();

  kvz_hip_me_result res[KVZ_HIP_SERVICE_MAX_REFS];
  if (!bad) bad = g_svc.search(g_svc.svc, &req, res) != KVZ_HIP_OK;
  long long t3 = svc_now_ns();
  __atomic_add_fetch(&g_svc.upload_ns, t1 - t0, __ATOMIC_RELAXED);
  __atomic_add_fetch(&g_svc.cand_ns, t2 - t1, __ATOMIC_RELAXED);
  __atomic_add_fetch(&g_svc.search_ns, t3 - t2, __ATOMIC_RELAXED);
  if (bad) {
    if (__atomic_fetch_add(&g_svc.failed, 1, __ATOMIC_RELAXED) == 0) fprintf(stderr, "svc_serve_cu_inter: %s\n", g_svc.last_error());
    *cur_cu = saved;
    return 0;
  }
  /* ---- :1275-1290 for every picture in order ---- */
  double cost = MAX_INT;
  uint32_t bitcost = MAX_INT;
  for (int ref_idx = 0; ref_idx < nref; ++ref_idx) {
    const kvz_hip_me_result *r = &res[ref_idx];
    if (r->cost != 0xffffffffu && r->cost < cost) {
      const int ref_list = ref_list_of[ref_idx], LX_idx = lx_idx_of[ref_idx];
      cur_cu->inter.mv_dir = ref_list + 1;
      cur_cu->merged = (uint8_t)r->merged;
      cur_cu->merge_idx = (uint8_t)r->merge_idx;
      cur_cu->inter.mv_ref[ref_list] = LX_idx;
      cur_cu->inter.mv[ref_list][0] = (int16_t)r->mv[0];
      cur_cu->inter.mv[ref_list][1] = (int16_t)r->mv[1];
      CU_SET_MV_CAND(cur_cu, ref_list, r->mv_cand);
      cost = r->cost;
      bitcost = r->bitcost + cur_cu->inter.mv_dir - 1 + LX_idx;
    }
  }
  if (g_svc.shadow) {
    /* the reference's own search from the same state: must decide the same; its result is the one kept */
    const cu_info_t mine = *cur_cu;
    *cur_cu = saved;
    double c2; uint32_t b2;
    __real_kvz_search_cu_inter(state, x, y, depth, lcu, &c2, &b2);
    int same = c2 == cost && (cost >= MAX_INT || (b2 == bitcost && cur_cu->inter.mv_dir == mine.inter.mv_dir && cur_cu->merged == mine.merged &&
               (!mine.merged || cur_cu->merge_idx == mine.merge_idx) &&
               !memcmp(cur_cu->inter.mv[mine.inter.mv_dir - 1], mine.inter.mv[mine.inter.mv_dir - 1], 4) &&
               cur_cu->inter.mv_ref[mine.inter.mv_dir - 1] == mine.inter.mv_ref[mine.inter.mv_dir - 1]));
    if (!same && __atomic_fetch_add(&g_svc.shadow_mismatch, 1, __ATOMIC_RELAXED) < 20)
      fprintf(stderr, "shadow: poc %d PU (%d,%d) %dx%d served cost %.0f bits %u dir %d mv (%d,%d) | reference cost %.0f bits %u dir %d mv (%d,%d)\n",
              fr->poc, x, y, width, width, cost, bitcost, mine.inter.mv_dir, mine.inter.mv[(mine.inter.mv_dir - 1) & 1][0], mine.inter.mv[(mine.inter.mv_dir - 1) & 1][1],
              c2, b2, cur_cu->inter.mv_dir, cur_cu->inter.mv[(cur_cu->inter.mv_dir - 1) & 1][0], cur_cu->inter.mv[(cur_cu->inter.mv_dir - 1) & 1][1]);
    *inter_cost = c2; *inter_bitcost = b2;
    __atomic_add_fetch(&g_svc.served, 1, __ATOMIC_RELAXED);
    return 1;
  }
  *inter_cost = cost;
  *inter_bitcost = bitcost;
  __atomic_add_fetch(&g_svc.served, 1, __ATOMIC_RELAXED);
  if (ctrl->cfg.rdo >= 2) kvz_cu_cost_inter_rd2(state, x, y, depth, lcu, inter_cost, inter_bitcost);     /* search_inter.c:1600-1607 */
  return 1;
}
