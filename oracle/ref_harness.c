/*
 * ref_harness.c -- glue that exposes the COMPILED REFERENCE (Kvazaar built from
 * /root/reference by oracle/Makefile into oracle/_ref/) to the tests and to
 * bench.py's cpu_baseline leg.  TEST INFRASTRUCTURE ONLY; this file contains
 * no reference code, it only calls it through the reference's own headers and
 * strategy registry (src/strategyselector.h:86-87, tests/test_strategies.c:29-52
 * is the model for walking the registry).
 */
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "strategyselector.h"
#include "encoder.h"
#include "encoderstate.h"
#include "image.h"
#include "scalinglist.h"
#include "cu.h"
#include "search_inter.h"
#include "rdo.h"
#include "intra.h"
#include "strategies/strategies-intra.h"
#include "sao.h"
#include "strategies/strategies-sao.h"

static strategy_list_t g_list;
static int g_ready = 0;

static encoder_control_t g_ctrl;
static encoder_state_t g_state;
static encoder_state_config_frame_t g_frame;

int ref_init(void)
{
  if (g_ready) return 1;
  /* selects the best registered strategy into the kvz_* globals (avx2/sse41..) */
  if (!kvz_strategyselector_init(1, 8)) return 0;
  memset(&g_list, 0, sizeof(g_list));
  if (!kvz_strategy_register_picture(&g_list, 8)) return 0;
  if (!kvz_strategy_register_dct(&g_list, 8)) return 0;
  if (!kvz_strategy_register_quant(&g_list, 8)) return 0;
  if (!kvz_strategy_register_ipol(&g_list, 8)) return 0;
  if (!kvz_strategy_register_intra(&g_list, 8)) return 0;
  if (!kvz_strategy_register_sao(&g_list, 8)) return 0;

  memset(&g_ctrl, 0, sizeof(g_ctrl));
  memset(&g_state, 0, sizeof(g_state));
  memset(&g_frame, 0, sizeof(g_frame));
  g_ctrl.bitdepth = 8;
  kvz_scalinglist_init(&g_ctrl.scaling_list);
  kvz_scalinglist_process(&g_ctrl.scaling_list, 8);
  g_state.encoder_control = &g_ctrl;
  g_state.frame = &g_frame;
  g_ready = 1;
  return 1;
}

int ref_strategy_count(void) { return (int)g_list.count; }
const char *ref_strategy_type(int i) { return g_list.strategies[i].type; }
const char *ref_strategy_name(int i) { return g_list.strategies[i].strategy_name; }
int ref_strategy_priority(int i) { return (int)g_list.strategies[i].priority; }

/* function pointer of (type, name); name NULL/"best" => highest priority */
void *ref_strategy(const char *type, const char *name)
{
  void *best = NULL; unsigned best_prio = 0;
  for (unsigned i = 0; i < g_list.count; ++i) {
    const strategy_t *s = &g_list.strategies[i];
    if (strcmp(s->type, type)) continue;
    if (name && strcmp(name, "best")) { if (!strcmp(s->strategy_name, name)) return s->fptr; }
    else if (!best || s->priority >= best_prio) { best = s->fptr; best_prio = s->priority; }
  }
  return best;
}

/* register an external implementation (the hip strategy under test) into the
 * harness list, exactly as strategies-*.c would. */
strategy_list_t *ref_list(void) { return &g_list; }

/* ---- thin typed call-throughs (ctypes cannot call raw fptrs portably) ---- */
unsigned ref_reg_sad(const char *name, const kvz_pixel *a, const kvz_pixel *b, int w, int h, unsigned s1, unsigned s2)
{ return ((reg_sad_func *)ref_strategy("reg_sad", name))(a, b, w, h, s1, s2); }

unsigned ref_cost_nxn(const char *type, const char *name, const kvz_pixel *a, const kvz_pixel *b)
{ return ((cost_pixel_nxn_func *)ref_strategy(type, name))(a, b); }

void ref_cost_nxn_dual(const char *type, const char *name, const kvz_pixel *preds, const kvz_pixel *orig, unsigned *costs)
{ ((cost_pixel_nxn_multi_func *)ref_strategy(type, name))((pred_buffer)preds, orig, 2, costs); }

unsigned ref_satd_any_size(const char *name, int w, int h, const kvz_pixel *a, int s1, const kvz_pixel *b, int s2)
{ return ((cost_pixel_any_size_func *)ref_strategy("satd_any_size", name))(w, h, a, s1, b, s2); }

void ref_satd_any_size_quad(const char *name, int w, int h, const kvz_pixel *p0, const kvz_pixel *p1,
                            const kvz_pixel *p2, const kvz_pixel *p3, int stride,
                            const kvz_pixel *orig, int orig_stride, unsigned *costs)
{
  const kvz_pixel *preds[4] = { p0, p1, p2, p3 };
  int8_t valid[4] = { 1, 1, 1, 1 };
  ((cost_pixel_any_size_multi_func *)ref_strategy("satd_any_size_quad", name))(w, h, preds, stride, orig, orig_stride, 4, costs, valid);
}

unsigned ref_pixels_calc_ssd(const char *name, const kvz_pixel *a, const kvz_pixel *b, int s1, int s2, int w)
{ return ((pixels_calc_ssd_func *)ref_strategy("pixels_calc_ssd", name))(a, b, s1, s2, w); }

void ref_transform(const char *type, const char *name, const int16_t *in, int16_t *out)
{ ((dct_func *)ref_strategy(type, name))(8, in, out); }

uint32_t ref_coeff_abs_sum(const char *name, const coeff_t *c, size_t n)
{ return ((coeff_abs_sum_func *)ref_strategy("coeff_abs_sum", name))(c, n); }

static void set_state(int qp, int slice_is_intra, int signhide, int scaling_list_default)
{
  g_state.qp = (int8_t)qp;
  g_frame.slicetype = slice_is_intra ? KVZ_SLICE_I : KVZ_SLICE_P;
  g_ctrl.cfg.signhide_enable = signhide;
  g_ctrl.cfg.rdoq_enable = 0;
  if ((int)g_ctrl.scaling_list.enable != scaling_list_default) {
    kvz_scalinglist_destroy(&g_ctrl.scaling_list);
    kvz_scalinglist_init(&g_ctrl.scaling_list);
    if (scaling_list_default) { g_ctrl.scaling_list.enable = 1; g_ctrl.scaling_list.use_default_list = 1; }
    kvz_scalinglist_process(&g_ctrl.scaling_list, 8);
  }
}

/* expose the processed scaling-list tables so the flattened C-ABI can be fed
 * the same per-coefficient factors the reference uses */
const int32_t *ref_quant_coeff_table(int log2_tr, int list_type, int qp_rem)
{ return g_ctrl.scaling_list.quant_coeff[log2_tr - 2][list_type][qp_rem]; }
const int32_t *ref_dequant_coeff_table(int log2_tr, int list_type, int qp_rem)
{ return g_ctrl.scaling_list.de_quant_coeff[log2_tr - 2][list_type][qp_rem]; }

void ref_quant(const char *name, int qp, int slice_is_intra, int signhide, int sl,
               coeff_t *coef, coeff_t *q_coef, int w, int h, int type, int scan_idx, int block_type)
{
  set_state(qp, slice_is_intra, signhide, sl);
  ((quant_func *)ref_strategy("quant", name))(&g_state, coef, q_coef, w, h, (int8_t)type, (int8_t)scan_idx, (int8_t)block_type);
}

void ref_dequant(const char *name, int qp, int sl, coeff_t *q_coef, coeff_t *coef, int w, int h, int type, int block_type)
{
  set_state(qp, 0, 0, sl);
  ((dequant_func *)ref_strategy("dequant", name))(&g_state, q_coef, coef, w, h, (int8_t)type, (int8_t)block_type);
}

int ref_quantize_residual(const char *name, int qp, int slice_is_intra, int signhide, int sl,
                          int cu_is_intra, int width, int color, int scan_order, int use_trskip,
                          int in_stride, int out_stride, const kvz_pixel *ref_in, const kvz_pixel *pred_in,
                          kvz_pixel *rec_out, coeff_t *coeff_out)
{
  cu_info_t cu; memset(&cu, 0, sizeof(cu));
  cu.type = cu_is_intra ? CU_INTRA : CU_INTER;
  cu.part_size = SIZE_2Nx2N;
  set_state(qp, slice_is_intra, signhide, sl);
  return (int)((quant_residual_func *)ref_strategy("quantize_residual", name))(
      &g_state, &cu, width, (color_t)color, (coeff_scan_order_t)scan_order, use_trskip,
      in_stride, out_stride, ref_in, pred_in, rec_out, coeff_out);
}

void ref_sample_luma(const char *name, kvz_pixel *src, int src_stride, int w, int h, kvz_pixel *dst, int dst_stride, int mvx, int mvy)
{
  int16_t mv[2] = { (int16_t)mvx, (int16_t)mvy };
  ((kvz_sample_quarterpel_luma_func *)ref_strategy("sample_quarterpel_luma", name))(&g_ctrl, src, (int16_t)src_stride, w, h, dst, (int16_t)dst_stride, 0, 0, mv);
}
void ref_sample_luma_14bit(const char *name, kvz_pixel *src, int src_stride, int w, int h, int16_t *dst, int dst_stride, int mvx, int mvy)
{
  int16_t mv[2] = { (int16_t)mvx, (int16_t)mvy };
  ((kvz_sample_14bit_quarterpel_luma_func *)ref_strategy("sample_14bit_quarterpel_luma", name))(&g_ctrl, src, (int16_t)src_stride, w, h, dst, (int16_t)dst_stride, 0, 0, mv);
}
void ref_sample_chroma(const char *name, kvz_pixel *src, int src_stride, int w, int h, kvz_pixel *dst, int dst_stride, int mvx, int mvy)
{
  int16_t mv[2] = { (int16_t)mvx, (int16_t)mvy };
  ((kvz_sample_octpel_chroma_func *)ref_strategy("sample_octpel_chroma", name))(&g_ctrl, src, (int16_t)src_stride, w, h, dst, (int16_t)dst_stride, 0, 0, mv);
}
void ref_sample_chroma_14bit(const char *name, kvz_pixel *src, int src_stride, int w, int h, int16_t *dst, int dst_stride, int mvx, int mvy)
{
  int16_t mv[2] = { (int16_t)mvx, (int16_t)mvy };
  ((kvz_sample_14bit_octpel_chroma_func *)ref_strategy("sample_14bit_octpel_chroma", name))(&g_ctrl, src, (int16_t)src_stride, w, h, dst, (int16_t)dst_stride, 0, 0, mv);
}

/* kvz_image_calc_sad / _satd (image.c:455,488) on raw luma planes, with the
 * named reg_sad / satd_any_size strategy installed in the globals. */
static void wrap_pic(kvz_picture *p, kvz_pixel *y, int w, int h)
{
  memset(p, 0, sizeof(*p));
  p->y = y; p->data[0] = y; p->width = w; p->height = h; p->stride = w;
}
unsigned ref_image_calc_sad(const char *name, kvz_pixel *pic, int pw, int ph, kvz_pixel *ref, int rw, int rh,
                            int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh)
{
  kvz_picture a, b; wrap_pic(&a, pic, pw, ph); wrap_pic(&b, ref, rw, rh);
  kvz_reg_sad = (reg_sad_func *)ref_strategy("reg_sad", name);
  return kvz_image_calc_sad(&a, &b, pic_x, pic_y, ref_x, ref_y, bw, bh);
}
unsigned ref_image_calc_satd(const char *name, kvz_pixel *pic, int pw, int ph, kvz_pixel *ref, int rw, int rh,
                             int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh)
{
  kvz_picture a, b; wrap_pic(&a, pic, pw, ph); wrap_pic(&b, ref, rw, rh);
  kvz_satd_any_size = (cost_pixel_any_size_func *)ref_strategy("satd_any_size", name);
  kvz_get_extended_block = (epol_func *)ref_strategy("get_extended_block", "generic");
  return kvz_image_calc_satd(&a, &b, pic_x, pic_y, ref_x, ref_y, bw, bh);
}

/* epol_func (strategies-ipol.h:41-42) of the named strategy: copies the (height + filter_size) x (width + filter_size)
 * window it describes into win_out, reports malloc_used, the stride and -- for an in-plane window -- the offset of
 * .buffer and .orig_topleft inside the plane; frees what the strategy allocated, like the callers do */
int ref_get_extended_block(const char *name, int xpos, int ypos, int mv_x, int mv_y, int off_x, int off_y, kvz_pixel *ref, int ref_w, int ref_h,
                           int filter_size, int width, int height, kvz_pixel *win_out, long *info /* stride, buffer - ref, topleft - buffer */)
{
  epol_func *f = (epol_func *)ref_strategy("get_extended_block", name);
  if (!f) return -1;
  kvz_extended_block b = { 0, 0, 0, 0 };
  f(xpos, ypos, mv_x, mv_y, off_x, off_y, ref, ref_w, ref_h, filter_size, width, height, &b);
  const int half = filter_size >> 1;
  for (int y = 0; y < height + 2 * half; ++y) memcpy(win_out + (size_t)y * (width + 2 * half), b.buffer + (size_t)y * b.stride, (size_t)(width + 2 * half));
  info[0] = (long)b.stride;
  info[1] = b.malloc_used ? -1 : (long)(b.buffer - ref);
  info[2] = (long)(b.orig_topleft - b.buffer);
  const int used = (int)b.malloc_used;
  if (b.malloc_used) free(b.buffer);
  return used;
}

/* The filter + quad-SATD sequence of search_frac (search_inter.c:965-1128),
 * driven through the reference's strategy functions, without MV bit costs:
 * validates orc_search_frac_costs. */
void ref_search_frac_costs(const char *name, kvz_pixel *pic, int pic_stride, kvz_pixel *ref, int ref_w, int ref_h,
                           int x, int y, int w, int h, int mvx, int mvy, unsigned costs_out[17], int best_out[2])
{
  static const int sq[9][2] = { {0,0}, {-1,0}, {1,0}, {0,-1}, {0,1}, {-1,-1}, {1,-1}, {-1,1}, {1,1} };
  ipol_blocks_func *steps[4] = {
    (ipol_blocks_func *)ref_strategy("filter_hpel_blocks_hor_ver_luma", name),
    (ipol_blocks_func *)ref_strategy("filter_hpel_blocks_diag_luma", name),
    (ipol_blocks_func *)ref_strategy("filter_qpel_blocks_hor_ver_luma", name),
    (ipol_blocks_func *)ref_strategy("filter_qpel_blocks_diag_luma", name) };
  cost_pixel_any_size_func *satd = (cost_pixel_any_size_func *)ref_strategy("satd_any_size", name);
  cost_pixel_any_size_multi_func *quad = (cost_pixel_any_size_multi_func *)ref_strategy("satd_any_size_quad", name);
  epol_func *ext = (epol_func *)ref_strategy("get_extended_block", "generic");

  kvz_pixel (*filtered)[LCU_WIDTH * LCU_WIDTH] = aligned_alloc(64, 4 * LCU_WIDTH * LCU_WIDTH);
  int16_t (*inter)[(KVZ_EXT_BLOCK_W_LUMA + 1) * LCU_WIDTH] = aligned_alloc(64, sizeof(int16_t) * 5 * (KVZ_EXT_BLOCK_W_LUMA + 1) * LCU_WIDTH);
  int16_t cols[5][KVZ_EXT_BLOCK_W_LUMA + 1];
  memset(filtered, 0, 4 * LCU_WIDTH * LCU_WIDTH);
  memset(inter, 0, sizeof(int16_t) * 5 * (KVZ_EXT_BLOCK_W_LUMA + 1) * LCU_WIDTH);
  memset(cols, 0, sizeof(cols));

  const int iw = ((w + 7) >> 3) << 3, ih = ((h + 7) >> 3) << 3;
  kvz_extended_block src = { 0, 0, 0, 0 };
  ext(x, y, mvx - 1, mvy - 1, 0, 0, ref, ref_w, ref_h, KVZ_LUMA_FILTER_TAPS, iw + 1, ih + 1, &src);
  kvz_pixel *cur = pic + y * pic_stride + x;
  unsigned best = satd(w, h, cur, pic_stride, src.orig_topleft + src.stride + 1, src.stride);
  costs_out[0] = best;
  int best_index = 0, i = 1; int8_t offx = 0, offy = 0;
  for (int step = 0; step < 4; ++step) {
    unsigned c[4]; int8_t valid[4] = { 1, 1, 1, 1 };
    const kvz_pixel *fp[4] = { filtered[0], filtered[1], filtered[2], filtered[3] };
    steps[step](&g_ctrl, src.orig_topleft, (int16_t)src.stride, iw, ih, filtered, inter, 4, cols, offx, offy);
    quad(w, h, fp, LCU_WIDTH, cur, pic_stride, 4, c, valid);
    for (int j = 0; j < 4; ++j) {
      costs_out[(step >= 2 ? 8 : 0) + i + j] = c[j];
      if (c[j] < best) { best = c[j]; best_index = i + j; }
    }
    i += 4;
    if (step == 1) { best_out[0] = best_index; offx = (int8_t)sq[best_index][0]; offy = (int8_t)sq[best_index][1]; best_index = 0; i = 1; }
    else if (step == 3) best_out[1] = best_index;
  }
  if (src.malloc_used) free(src.buffer);
  free(filtered); free(inter);
}

/* One frac-search filter step through the reference (for block-level parity
 * of the 4 filtered blocks).  State arrays are caller provided so that the
 * four steps can be chained. */
void ref_filter_step(const char *name, int step, kvz_pixel *src, int src_stride, int w, int h,
                     kvz_pixel *filtered, int16_t *inter, int16_t *cols, int fme_level, int offx, int offy)
{
  static const char *types[4] = { "filter_hpel_blocks_hor_ver_luma", "filter_hpel_blocks_diag_luma",
                                  "filter_qpel_blocks_hor_ver_luma", "filter_qpel_blocks_diag_luma" };
  ((ipol_blocks_func *)ref_strategy(types[step], name))(&g_ctrl, src, (int16_t)src_stride, w, h,
      (kvz_pixel (*)[LCU_WIDTH * LCU_WIDTH])filtered,
      (int16_t (*)[(KVZ_EXT_BLOCK_W_LUMA + 1) * LCU_WIDTH])inter, (int8_t)fme_level,
      (int16_t (*)[KVZ_EXT_BLOCK_W_LUMA + 1])cols, (int8_t)offx, (int8_t)offy);
}

/* ---- CPU baseline timing loops (speed_tests.c:117-154 is the model: call a
 * strategy over a working set for a wall-clock budget, report calls/s) ---- */
static double now_s(void)
{
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* blocks/s of a cost_pixel_nxn_func over `count` contiguous block pairs */
double ref_bench_cost_nxn(const char *type, const char *name, int n, const kvz_pixel *a, const kvz_pixel *b,
                          size_t count, double budget_s, unsigned *checksum)
{
  cost_pixel_nxn_func *f = (cost_pixel_nxn_func *)ref_strategy(type, name);
  if (!f) return -1.0;
  size_t done = 0; unsigned acc = 0; const size_t bs = (size_t)n * n;
  double t0 = now_s(), t1;
  do {
    for (size_t i = 0; i < count; ++i) acc += f(a + i * bs, b + i * bs);
    done += count; t1 = now_s();
  } while (t1 - t0 < budget_s);
  if (checksum) *checksum = acc;
  return (double)done / (t1 - t0);
}

double ref_bench_transform(const char *type, const char *name, int n, const int16_t *in, int16_t *out,
                           size_t count, double budget_s)
{
  dct_func *f = (dct_func *)ref_strategy(type, name);
  if (!f) return -1.0;
  size_t done = 0; const size_t bs = (size_t)n * n;
  double t0 = now_s(), t1;
  do {
    for (size_t i = 0; i < count; ++i) f(8, in + i * bs, out + i * bs);
    done += count; t1 = now_s();
  } while (t1 - t0 < budget_s);
  return (double)done / (t1 - t0);
}

/* the reference's CABAC tables (cabac.c:28-75), for the test that pins the oracle's restatement of ITU-T H.265 Tables 9-46 / 9-47:
 * kind 0 kvz_g_auc_lpst_table[i >> 2][i & 3], 1 kvz_g_auc_next_state_mps[i], 2 kvz_g_auc_next_state_lps[i], 3 kvz_g_auc_renorm_table[i] */
#include "cabac.h"
int ref_cabac_table(int kind, int i)
{
  switch (kind) {
    case 0: return kvz_g_auc_lpst_table[i >> 2][i & 3];
    case 1: return kvz_g_auc_next_state_mps[i];
    case 2: return kvz_g_auc_next_state_lps[i];
    default: return kvz_g_auc_renorm_table[i];
  }
}

/* whole-launch parity checks: one strategy function over `count` contiguous blocks (a test gives each host thread a range) */
int ref_cost_nxn_many(const char *type, const char *name, int n, const kvz_pixel *a, const kvz_pixel *b, size_t count, unsigned *costs)
{
  cost_pixel_nxn_func *f = (cost_pixel_nxn_func *)ref_strategy(type, name);
  if (!f) return -1;
  const size_t bs = (size_t)n * n;
  for (size_t i = 0; i < count; ++i) costs[i] = f(a + i * bs, b + i * bs);
  return 0;
}
int ref_transform_many(const char *type, const char *name, int n, const int16_t *in, int16_t *out, size_t count)
{
  dct_func *f = (dct_func *)ref_strategy(type, name);
  if (!f) return -1;
  const size_t bs = (size_t)n * n;
  for (size_t i = 0; i < count; ++i) f(8, in + i * bs, out + i * bs);
  return 0;
}

double ref_bench_reg_sad(const char *name, const kvz_pixel *a, const kvz_pixel *b, int stride, int fw, int fh,
                         int bw, int bh, double budget_s, unsigned *checksum)
{
  reg_sad_func *f = (reg_sad_func *)ref_strategy("reg_sad", name);
  if (!f) return -1.0;
  size_t done = 0; unsigned acc = 0;
  double t0 = now_s(), t1;
  do {
    for (int y = 0; y + bh <= fh; y += bh)
      for (int x = 0; x + bw <= fw; x += bw) { acc += f(a + y * stride + x, b + y * stride + x, bw, bh, stride, stride); ++done; }
    t1 = now_s();
  } while (t1 - t0 < budget_s);
  if (checksum) *checksum = acc;
  return (double)done / (t1 - t0);
}

/* The reg_sad loop of the reference's own benchmark (tests/speed_tests.c:198-236, test_inter_speed): "a sparse full search on
 * the first CU of every LCU" of a 4K frame -- iteration i takes LCU (1 + i % (W/64 - 2), 1 + (i / (H/64 - 2)) % (H/64 - 2)) and
 * the 5 x 5 vectors {-6, -3, 0, 3, 6}^2, both blocks in the SAME frame -- run for budget_s seconds.  Returns calls per second. */
double ref_bench_speed_inter_sad(const char *name, const kvz_pixel *frame, int W, int H, int bw, int bh, double budget_s, unsigned long long *checksum)
{
  reg_sad_func *f = (reg_sad_func *)ref_strategy("reg_sad", name);
  if (!f) return -1.0;
  const int dx = W / 64 - 2, dy = H / 64 - 2;
  size_t done = 0; unsigned long long acc = 0;
  double t0 = now_s(), t1;
  uint64_t i = 0;
  do {
    for (int rep = 0; rep < 256; ++rep, ++i) {
      const int lx = 1 + (int)(i % (uint64_t)dx), ly = 1 + (int)((i / (uint64_t)dy) % (uint64_t)dy);
      const kvz_pixel *buf1 = frame + (size_t)ly * 64 * W + (size_t)lx * 64;
      for (int my = -6; my <= 6; my += 3)
        for (int mx = -6; mx <= 6; mx += 3) { acc += f(buf1, buf1 + my * W + mx, bw, bh, W, W); ++done; }
    }
    t1 = now_s();
  } while (t1 - t0 < budget_s);
  if (checksum) *checksum = acc;
  return (double)done / (t1 - t0);
}

/* ------------------------------------------------------------------------
 * Integration check of the drop-in boundary: load libkvzhip.so and let it
 * register its "hip" strategies into THIS process's reference registry through
 * the reference's own kvz_strategyselector_register -- exactly what the patched
 * strategies-*.c of INTEGRATION.md do.  The accessors below are the glue a
 * Kvazaar maintainer compiles inside the encoder (they need encoderstate.h).
 * ------------------------------------------------------------------------ */
/* ---- deblocking: kvz_filter_deblock_lcu (filter.c:770-779) for every LCU in raster order, on an encoder state
 * fabricated from the flat SCU map the oracle / the GPU entry take (same layout as orc_cu_info / orc_deblock_params) ---- */
#include "filter.h"
#include "videoframe.h"
typedef struct { uint8_t type, depth, part_size, tr_depth, cbf_y, mv_dir, qp, reserved; int16_t mv[2][2]; uint8_t mv_ref[2]; uint8_t pad[2]; } ref_cu_flat;
typedef struct { int32_t beta_offset_div2, tc_offset_div2, qp, frame_qp, per_cu_qp, slice_is_b, chroma, reserved; uint8_t ref_LX[2][16]; } ref_deblock_prm;

void ref_deblock_frame(kvz_pixel *y, int stride_y, kvz_pixel *u, kvz_pixel *v, int stride_c, int width, int height,
                       const ref_cu_flat *cus, const ref_deblock_prm *prm)
{
  static encoder_control_t ctrl;
  static encoder_state_t state;
  static encoder_state_config_frame_t frame;
  static encoder_state_config_tile_t tile;
  static videoframe_t vframe;
  memset(&ctrl, 0, sizeof(ctrl)); memset(&state, 0, sizeof(state)); memset(&frame, 0, sizeof(frame));
  memset(&tile, 0, sizeof(tile)); memset(&vframe, 0, sizeof(vframe));
  ctrl.bitdepth = 8;
  ctrl.cfg.deblock_beta = prm->beta_offset_div2;
  ctrl.cfg.deblock_tc = prm->tc_offset_div2;
  ctrl.max_qp_delta_depth = prm->per_cu_qp ? 0 : -1;
  ctrl.chroma_format = prm->chroma ? KVZ_CSP_420 : KVZ_CSP_400;
  kvz_picture rec;
  memset(&rec, 0, sizeof(rec));
  rec.y = y; rec.u = u; rec.v = v; rec.width = width; rec.height = height; rec.stride = stride_y;
  (void)stride_c;                                           /* the reference derives it: rec->stride >> 1 */
  vframe.rec = &rec; vframe.width = width; vframe.height = height;
  vframe.width_in_lcu = (width + 63) / 64; vframe.height_in_lcu = (height + 63) / 64;
  vframe.cu_array = kvz_cu_array_alloc(width, height);
  const int w4 = (width + 3) >> 2;
  for (int sy = 0; sy < height; sy += 4)
    for (int sx = 0; sx < width; sx += 4) {
      const ref_cu_flat *f = &cus[(sy >> 2) * w4 + (sx >> 2)];
      cu_info_t *cu = kvz_cu_array_at(vframe.cu_array, sx, sy);
      cu->type = f->type; cu->depth = f->depth; cu->part_size = f->part_size; cu->tr_depth = f->tr_depth;
      cu->cbf = 0;
      if (f->cbf_y) cbf_set(&cu->cbf, f->tr_depth, COLOR_Y);
      cu->qp = f->qp;
      if (f->type != CU_INTRA) {
        memcpy(cu->inter.mv, f->mv, sizeof(cu->inter.mv));
        cu->inter.mv_ref[0] = f->mv_ref[0]; cu->inter.mv_ref[1] = f->mv_ref[1];
        cu->inter.mv_dir = f->mv_dir;
      }
    }
  tile.frame = &vframe;
  frame.QP = prm->frame_qp;
  frame.slicetype = prm->slice_is_b ? KVZ_SLICE_B : KVZ_SLICE_P;
  memcpy(frame.ref_LX, prm->ref_LX, sizeof(frame.ref_LX));
  state.encoder_control = &ctrl; state.tile = &tile; state.frame = &frame;
  state.qp = prm->qp;
  for (int ly = 0; ly < height; ly += 64)
    for (int lx = 0; lx < width; lx += 64) kvz_filter_deblock_lcu(&state, lx, ly);
  kvz_cu_array_free(&vframe.cu_array);
}

#include <dlfcn.h>
#include "../include/kvz_hip.h"

static int acc_qp(const void *s) { return ((const encoder_state_t *)s)->qp; }
static int acc_slice_is_intra(const void *s) { return ((const encoder_state_t *)s)->frame->slicetype == KVZ_SLICE_I; }
static int acc_signhide(const void *s) { return ((const encoder_state_t *)s)->encoder_control->cfg.signhide_enable; }
static int acc_sl_enable(const void *s) { return ((const encoder_state_t *)s)->encoder_control->scaling_list.enable; }
static const int32_t *acc_quant_coeff(const void *s, int log2_tr, int list, int rem)
{ return ((const encoder_state_t *)s)->encoder_control->scaling_list.quant_coeff[log2_tr - 2][list][rem]; }
static const int32_t *acc_dequant_coeff(const void *s, int log2_tr, int list, int rem)
{ return ((const encoder_state_t *)s)->encoder_control->scaling_list.de_quant_coeff[log2_tr - 2][list][rem]; }
static int acc_rdoq(const void *s) { return ((const encoder_state_t *)s)->encoder_control->cfg.rdoq_enable; }
static int acc_cu_is_intra(const void *cu) { return ((const cu_info_t *)cu)->type == CU_INTRA; }
static int acc_rdoq_skip(const void *s) { return ((const encoder_state_t *)s)->encoder_control->cfg.rdoq_skip; }
static int acc_cu_rdoq_tr_depth(const void *cu)
{
  const cu_info_t *c = (const cu_info_t *)cu;
  return c->tr_depth - c->depth + (c->part_size == SIZE_NxN ? 1 : 0);
}
static int acc_cu_type(const void *cu) { return ((const cu_info_t *)cu)->type; }
static void acc_rdoq_fn(void *state, coeff_t *coef, coeff_t *dest, int32_t w, int32_t h, int8_t type, int8_t scan, int8_t block_type, int8_t tr_depth)
{
  kvz_rdoq((encoder_state_t *)state, coef, dest, w, h, type, scan, block_type, tr_depth);
}
static const int16_t *acc_hp_y(const void *b) { return ((const hi_prec_buf_t *)b)->y; }
static const int16_t *acc_hp_u(const void *b) { return ((const hi_prec_buf_t *)b)->u; }
static const int16_t *acc_hp_v(const void *b) { return ((const hi_prec_buf_t *)b)->v; }
static kvz_pixel *acc_rec_y(void *l) { return ((lcu_t *)l)->rec.y; }
static kvz_pixel *acc_rec_u(void *l) { return ((lcu_t *)l)->rec.u; }
static kvz_pixel *acc_rec_v(void *l) { return ((lcu_t *)l)->rec.v; }

/* returns the number of strategies the hip library registered, or -1 */
int ref_register_hip(const char *lib_path)
{
  void *h = dlopen(lib_path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) { fprintf(stderr, "dlopen %s: %s\n", lib_path, dlerror()); return -1; }
  void (*set_reg)(kvz_hip_register_fn) = (void (*)(kvz_hip_register_fn))dlsym(h, "kvz_hip_set_registrar");
  void (*set_acc)(const kvz_hip_state_accessors *) = (void (*)(const kvz_hip_state_accessors *))dlsym(h, "kvz_hip_set_state_accessors");
  int (*reg_pic)(void *, uint8_t) = (int (*)(void *, uint8_t))dlsym(h, "kvz_strategy_register_picture_hip");
  int (*reg_dct)(void *, uint8_t) = (int (*)(void *, uint8_t))dlsym(h, "kvz_strategy_register_dct_hip");
  int (*reg_quant)(void *, uint8_t) = (int (*)(void *, uint8_t))dlsym(h, "kvz_strategy_register_quant_hip");
  int (*reg_ipol)(void *, uint8_t) = (int (*)(void *, uint8_t))dlsym(h, "kvz_strategy_register_ipol_hip");
  int (*reg_intra)(void *, uint8_t) = (int (*)(void *, uint8_t))dlsym(h, "kvz_strategy_register_intra_hip");
  int (*reg_sao)(void *, uint8_t) = (int (*)(void *, uint8_t))dlsym(h, "kvz_strategy_register_sao_hip");
  if (!set_reg || !set_acc || !reg_pic || !reg_dct || !reg_quant || !reg_ipol || !reg_intra || !reg_sao) return -1;
  static const kvz_hip_state_accessors acc = { acc_qp, acc_slice_is_intra, acc_signhide, acc_sl_enable,
                                               acc_quant_coeff, acc_dequant_coeff, acc_rdoq, acc_cu_is_intra,
                                               acc_hp_y, acc_hp_u, acc_hp_v, acc_rec_y, acc_rec_u, acc_rec_v,
                                               acc_rdoq_skip, acc_cu_rdoq_tr_depth, acc_cu_type, acc_rdoq_fn };
  set_reg(kvz_strategyselector_register);
  set_acc(&acc);
  unsigned before = g_list.count;
  if (!reg_pic(&g_list, 8) || !reg_dct(&g_list, 8) || !reg_quant(&g_list, 8) || !reg_ipol(&g_list, 8) ||
      !reg_intra(&g_list, 8) || !reg_sao(&g_list, 8)) return -1;
  return (int)(g_list.count - before);
}

/* inter_recon_bipred (strategies-picture.h:117-130) of the named strategy on flat planes:
 * hp0/hp1 = 14-bit buffers (y 64x64, u/v 32x32), rec = lcu->rec planes (in/out), tmp = temp_lcu planes */
void ref_bipred(const char *name, int hi_l0, int hi_l1, int hi_c0, int hi_c1, int height, int width, int ypos, int xpos,
                int16_t *hp0_y, int16_t *hp0_u, int16_t *hp0_v, int16_t *hp1_y, int16_t *hp1_u, int16_t *hp1_v,
                kvz_pixel *rec_y, kvz_pixel *rec_u, kvz_pixel *rec_v, kvz_pixel *tmp_y, kvz_pixel *tmp_u, kvz_pixel *tmp_v)
{
  lcu_t *lcu = calloc(1, sizeof(lcu_t));
  hi_prec_buf_t b0, b1;
  memset(&b0, 0, sizeof(b0)); memset(&b1, 0, sizeof(b1));
  b0.y = hp0_y; b0.u = hp0_u; b0.v = hp0_v; b1.y = hp1_y; b1.u = hp1_u; b1.v = hp1_v;
  memcpy(lcu->rec.y, rec_y, 64 * 64); memcpy(lcu->rec.u, rec_u, 32 * 32); memcpy(lcu->rec.v, rec_v, 32 * 32);
  ((inter_recon_bipred_func *)ref_strategy("inter_recon_bipred", name))(hi_l0, hi_l1, hi_c0, hi_c1, height, width, ypos, xpos,
                                                                         /* as inter.c:455-458: no buffer without a fractional MV */
                                                                         (hi_l0 || hi_c0) ? &b0 : NULL, (hi_l1 || hi_c1) ? &b1 : NULL,
                                                                         lcu, tmp_y, tmp_u, tmp_v);
  memcpy(rec_y, lcu->rec.y, 64 * 64); memcpy(rec_u, lcu->rec.u, 32 * 32); memcpy(rec_v, lcu->rec.v, 32 * 32);
  free(lcu);
}

/* ------------------------------------------------------------------------
 * End-to-end check of the drop-in: run the REFERENCE ENCODER (kvz_api,
 * kvazaar.c:378-395) on raw 4:2:0 frames.  With strategy_name != NULL every
 * strategy type that `strategy_name` registered in the harness list is installed
 * into the encoder's global function pointers after encoder_open -- exactly the
 * table kvz_strategyselector_init fills (strategyselector.h:99-108), i.e. what the
 * selector does itself once the hooks of INTEGRATION.md are compiled in.
 * opts: "name=value,name=value" for kvz_config_parse.  Returns the bitstream
 * length (bytes copied to out, at most cap) or -1.
 * ------------------------------------------------------------------------ */
#include "kvazaar.h"

static long append_chunks(const kvz_api *api, kvz_data_chunk *chunks, uint8_t *out, long pos, long cap)
{
  for (kvz_data_chunk *c = chunks; c; c = c->next) {
    if (pos + (long)c->len <= cap) memcpy(out + pos, c->data, c->len);
    pos += c->len;
  }
  if (chunks) api->chunk_free(chunks);
  return pos;
}

long ref_encode(const uint8_t *yuv, int w, int h, int nframes, const char *opts, const char *strategy_name,
                uint8_t *out, long cap, int *installed_out)
{
  const kvz_api *api = kvz_api_get(8);
  if (!api) return -1;
  kvz_config *cfg = api->config_alloc();
  api->config_init(cfg);
  char num[32];
  snprintf(num, sizeof(num), "%d", w); api->config_parse(cfg, "width", num);
  snprintf(num, sizeof(num), "%d", h); api->config_parse(cfg, "height", num);
  char *dup = strdup(opts ? opts : "");
  for (char *tok = strtok(dup, ","); tok; tok = strtok(NULL, ",")) {
    char *eq = strchr(tok, '=');
    const char *val = "true";
    if (eq) { *eq = 0; val = eq + 1; }
    if (!api->config_parse(cfg, tok, val)) { fprintf(stderr, "ref_encode: bad option %s=%s\n", tok, val); free(dup); return -1; }
  }
  free(dup);
  kvz_encoder *enc = api->encoder_open(cfg);
  if (!enc) return -1;
  int installed = 0;
  if (strategy_name) {
    for (const strategy_to_select_t *s = strategies_to_select; s->fptr; ++s) {
      void *f = NULL;
      for (unsigned i = 0; i < g_list.count; ++i)
        if (!strcmp(g_list.strategies[i].type, s->strategy_type) && !strcmp(g_list.strategies[i].strategy_name, strategy_name))
          f = g_list.strategies[i].fptr;
      if (f) { *s->fptr = f; ++installed; }
    }
  }
  if (installed_out) *installed_out = installed;
  long pos = 0;
  const size_t fsz = (size_t)w * h * 3 / 2;
  int fed = 0, done = 0;
  while (!done) {
    kvz_picture *pic = NULL;
    if (fed < nframes) {
      pic = api->picture_alloc(w, h);
      const uint8_t *f = yuv + fsz * fed;
      for (int y = 0; y < h; ++y) memcpy(pic->y + (size_t)y * pic->stride, f + (size_t)y * w, w);
      for (int y = 0; y < h / 2; ++y) {
        memcpy(pic->u + (size_t)y * (pic->stride / 2), f + (size_t)w * h + (size_t)y * (w / 2), w / 2);
        memcpy(pic->v + (size_t)y * (pic->stride / 2), f + (size_t)w * h * 5 / 4 + (size_t)y * (w / 2), w / 2);
      }
      ++fed;
    }
    kvz_data_chunk *chunks = NULL; uint32_t len = 0; kvz_picture *rec = NULL, *src = NULL; kvz_frame_info info;
    if (!api->encoder_encode(enc, pic, &chunks, &len, &rec, &src, &info)) { api->picture_free(pic); pos = -1; break; }
    if (!chunks && !pic) done = 1;
    pos = append_chunks(api, chunks, out, pos, cap);
    api->picture_free(pic); api->picture_free(rec); api->picture_free(src);
  }
  api->encoder_close(enc);
  api->config_destroy(cfg);
  return pos;
}

/* ------------------------------------------------------------------------
 * intra group (strategies-intra.h:33-55, intra.c:281-331).  refs_in = the
 * kvz_intra_ref layout {left[65], top[65]} (intra.h:35-38).
 * ------------------------------------------------------------------------ */


void ref_angular_pred(const char *name, int log2_width, int mode, const kvz_pixel *above, const kvz_pixel *left, kvz_pixel *dst)
{
  ((angular_pred_func *)ref_strategy("angular_pred", name))(log2_width, mode, above, left, dst);
}

void ref_intra_pred_planar(const char *name, int log2_width, const kvz_pixel *top, const kvz_pixel *left, kvz_pixel *dst)
{
  ((intra_pred_planar_func *)ref_strategy("intra_pred_planar", name))(log2_width, top, left, dst);
}

/* kvz_intra_predict itself (intra.c:281), with the angular / planar strategies `name` installed in the globals */
void ref_intra_predict(const char *name, const uint8_t *refs_in /*130 bytes*/, int log2_width, int mode, int color,
                       int filter_boundary, kvz_pixel *dst)
{
  angular_pred_func *save_a = kvz_angular_pred;
  intra_pred_planar_func *save_p = kvz_intra_pred_planar;
  kvz_angular_pred = (angular_pred_func *)ref_strategy("angular_pred", name);
  kvz_intra_pred_planar = (intra_pred_planar_func *)ref_strategy("intra_pred_planar", name);
  kvz_intra_references refs;
  memset(&refs, 0, sizeof(refs));
  memcpy(refs.ref.left, refs_in, 65);
  memcpy(refs.ref.top, refs_in + 65, 65);
  refs.filtered_initialized = false;
  kvz_intra_predict(&refs, log2_width, mode, (color_t)color, dst, filter_boundary != 0);
  kvz_angular_pred = save_a;
  kvz_intra_pred_planar = save_p;
}

/* kvz_intra_build_reference (intra.c:574-588) on an LCU whose planes of `color` the caller filled
 * (rec 64x64 luma / 32x32 chroma, top / left = the 96 / 48 border pixels after entry 0 = top_left):
 * used to produce realistic reference arrays (unavailable neighbours, picture edges). */
void ref_intra_build_reference(int log2_width, int color, int luma_x, int luma_y, int pic_w, int pic_h,
                               const kvz_pixel *rec, const kvz_pixel *top, const kvz_pixel *left, int top_left,
                               uint8_t *refs_out /*130 bytes*/)
{
  lcu_t *lcu = calloc(1, sizeof(lcu_t));
  kvz_pixel *rec_p = color == 0 ? lcu->rec.y : color == 1 ? lcu->rec.u : lcu->rec.v;
  kvz_pixel *top_p = color == 0 ? lcu->top_ref.y : color == 1 ? lcu->top_ref.u : lcu->top_ref.v;
  kvz_pixel *left_p = color == 0 ? lcu->left_ref.y : color == 1 ? lcu->left_ref.u : lcu->left_ref.v;
  const int w = color ? LCU_WIDTH_C : LCU_WIDTH, nref = color ? LCU_REF_PX_WIDTH / 2 + 1 : LCU_REF_PX_WIDTH + 1;
  memcpy(rec_p, rec, w * w);
  memcpy(top_p, top, nref);
  memcpy(left_p, left, nref);
  top_p[0] = left_p[0] = (kvz_pixel)top_left;
  kvz_intra_references refs;
  memset(&refs, 0, sizeof(refs));
  vector2d_t luma_px = { luma_x, luma_y }, pic_px = { pic_w, pic_h };
  kvz_intra_build_reference(log2_width, (color_t)color, &luma_px, &pic_px, lcu, &refs);
  memcpy(refs_out, refs.ref.left, 65);
  memcpy(refs_out + 65, refs.ref.top, 65);
  free(lcu);
}

/* ------------------------------------------------------------------------
 * SAO group (strategies-sao.h:36-57).  sao14 = the 14 ints of the test-side
 * sao record {type, eo_class, band_position[2], offsets[10]}.
 * ------------------------------------------------------------------------ */

int ref_sao_edge_ddistortion(const char *name, const kvz_pixel *orig, const kvz_pixel *rec, int bw, int bh, int eo_class, int *offsets)
{
  return ((sao_edge_ddistortion_func *)ref_strategy("sao_edge_ddistortion", name))(orig, rec, bw, bh, eo_class, offsets);
}

void ref_calc_sao_edge_dir(const char *name, const kvz_pixel *orig, const kvz_pixel *rec, int eo_class, int bw, int bh, int *cat_sum_cnt)
{
  ((calc_sao_edge_dir_func *)ref_strategy("calc_sao_edge_dir", name))(orig, rec, eo_class, bw, bh, (int (*)[NUM_SAO_EDGE_CATEGORIES])cat_sum_cnt);
}

int ref_sao_band_ddistortion(const char *name, const kvz_pixel *orig, const kvz_pixel *rec, int bw, int bh, int band_pos, int *bands)
{
  set_state(27, 0, 0, 0);
  return ((sao_band_ddistortion_func *)ref_strategy("sao_band_ddistortion", name))(&g_state, orig, rec, bw, bh, band_pos, bands);
}

void ref_sao_reconstruct_color(const char *name, const kvz_pixel *rec, kvz_pixel *new_rec, const int32_t *sao14, int stride, int new_stride,
                               int bw, int bh, int color)
{
  sao_info_t sao;
  memset(&sao, 0, sizeof(sao));
  sao.type = (sao_type)sao14[0];
  sao.eo_class = (sao_eo_class)sao14[1];
  sao.band_position[0] = sao14[2]; sao.band_position[1] = sao14[3];
  for (int i = 0; i < 10; ++i) sao.offsets[i] = sao14[4 + i];
  ((sao_reconstruct_color_func *)ref_strategy("sao_reconstruct_color", name))(&g_ctrl, rec, new_rec, &sao, stride, new_stride, bw, bh, (color_t)color);
}

int ref_sizeof_sao_info(void) { return (int)sizeof(sao_info_t); }


/* ------------------------------------------------------------------------
 * Recorder of the reference's OWN inter searches during a real encode (the "driver" half of
 * SURVEY 8f row 1).  oracle/Makefile links this library with -Wl,--wrap=kvz_search_cu_inter, so that the call in
 * search.c reaches __wrap_kvz_search_cu_inter below, which -- when recording -- derives the PU's AMVP / merge
 * candidates and start vector with the encoder's own functions exactly as search_pu_inter / search_pu_inter_ref do
 * (search_inter.c:1492-1500, :1175-1206; pure functions of the state), runs the untouched reference search
 * (__real_kvz_search_cu_inter) and notes what it decided.  With ONE reference picture and no bi-prediction the
 * outputs of kvz_search_cu_inter are the outputs of the single search_pu_inter_ref call (:1275-1290): inter_cost =
 * best_cost, inter_bitcost = best_bitcost, cur_cu->inter.mv[0] = best_mv, merged / merge_idx / mv_cand.
 * Nothing here changes what the encoder does; without ref_record_begin the wrapper is a plain call-through.
 * ------------------------------------------------------------------------ */
#include "search_inter.h"
#include "search_intra.h"
#include "search.h"
#include "inter.h"

typedef struct { int16_t mv[2]; uint8_t usable; uint8_t same_ref; } rec_merge_t;
typedef struct {                              /* = kvz_hip_me_pu / orc_me_pu */
  int32_t x, y, width, height;
  int16_t mv_cand[2][2];
  int16_t extra_mv[2];
  int16_t num_merge_cand, reserved;
  rec_merge_t merge[5];
  int16_t pad;
} rec_pu_t;
typedef struct { int32_t mv[2]; uint32_t cost, bitcost; int32_t merged, merge_idx, mv_cand, reserved; } rec_result_t;
typedef struct { int32_t frame, lcu_x, lcu_y, seq, lambda_cost, depth; } rec_meta_t;
typedef struct {                              /* = kvz_hip_me_params / orc_me_params */
  int32_t lambda_cost, early_termination;
  uint32_t max_steps;
  int32_t fme_level, wpp_owf, ref_delay_px, max_ref_lcu_down, max_ref_lcu_right;
  int32_t algorithm, search_range, size_classes, mv_constraint;
  int32_t tile_x, tile_y, tile_w, tile_h;
  int32_t mv_rdo, ref_idx, refs_before, reserved;
  const void *cabac;
  const void *cost_to_beat;
} rec_params_t;

static struct {
  int on, max, count, skipped;
  rec_pu_t *pu; rec_result_t *res; rec_meta_t *meta;
  rec_params_t params;
  int n_frames, max_frames, w, h;
  int32_t *frame_poc;
  kvz_pixel *pic, *ref;                       /* [frame][h][w] luma: the source and the reference picture searched */
  int cur_lcu_x, cur_lcu_y, cur_frame, seq;
} g_rec;

void __real_kvz_search_cu_inter(encoder_state_t * const state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost);
/* optional snapshots of what the candidate derivation read (defined with the flat CU types below) */
static void rec_snapshot_search(const encoder_state_t *state, const lcu_t *lcu, int record_index);
static void rec_snapshot_frame(const encoder_state_t *state, int frame_index);
/* optional: the search served by the GPU chain instead of the reference's (defined at the end of the file) */
static int gpu_search_serve(encoder_state_t *state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost);

static void rec_copy_plane(kvz_pixel *dst, const kvz_picture *p, int w, int h)
{
  for (int r = 0; r < h; ++r) memcpy(dst + (size_t)r * w, p->y + (size_t)r * p->stride, (size_t)w);
}

/* optional: the search answered by the product's search service, from every threadqueue worker at once (ref_serve.c) */
int svc_serve_cu_inter(encoder_state_t *state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost);

void __wrap_kvz_search_cu_inter(encoder_state_t * const state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost)
{
  if (svc_serve_cu_inter(state, x, y, depth, lcu, inter_cost, inter_bitcost)) return;
  if (gpu_search_serve(state, x, y, depth, lcu, inter_cost, inter_bitcost)) return;
  const encoder_control_t *ctrl = state->encoder_control;
  const int usable = g_rec.on && state->frame->ref->used_size == 1 && state->frame->slicetype == KVZ_SLICE_P &&
                     !ctrl->cfg.mv_rdo && state->tile->offset_x == 0 && state->tile->offset_y == 0;
  if (!usable || g_rec.count >= g_rec.max) {
    if (g_rec.on) ++g_rec.skipped;
    __real_kvz_search_cu_inter(state, x, y, depth, lcu, inter_cost, inter_bitcost);
    return;
  }
  const int width = LCU_WIDTH >> depth;
  rec_pu_t pu; memset(&pu, 0, sizeof(pu));
  pu.x = x; pu.y = y; pu.width = width; pu.height = width;
  cu_info_t *cur_cu = LCU_GET_CU_AT_PX(lcu, SUB_SCU(x), SUB_SCU(y));
  rec_snapshot_search(state, lcu, g_rec.count);          /* before the derivation below touches the neighbours */
  /* search_pu_inter :1492-1500 */
  inter_merge_cand_t merge[MRG_MAX_NUM_CANDS];
  const int n_merge = kvz_inter_get_merge_cand(state, x, y, width, width, true, true, merge, lcu);
  pu.num_merge_cand = (int16_t)n_merge;
  for (int i = 0; i < n_merge; ++i) {
    const int dir = merge[i].dir;
    pu.merge[i].usable = dir != 3;
    if (dir != 3) {
      pu.merge[i].mv[0] = merge[i].mv[dir - 1][0]; pu.merge[i].mv[1] = merge[i].mv[dir - 1][1];
      pu.merge[i].same_ref = state->frame->ref_LX[dir - 1][merge[i].ref[dir - 1]] == 0;
    }
  }
  /* search_pu_inter_ref :1170-1187 (reference 0 = L0[0]) */
  const int8_t saved = cur_cu->inter.mv_ref[0];
  const uint8_t saved_cand0 = cur_cu->inter.mv_cand0, saved_cand1 = cur_cu->inter.mv_cand1;
  cur_cu->inter.mv_ref[0] = 0;
  kvz_inter_get_mv_cand(state, x, y, width, width, pu.mv_cand, cur_cu, lcu, 0);
  cur_cu->inter.mv_ref[0] = saved;
  cur_cu->inter.mv_cand0 = saved_cand0; cur_cu->inter.mv_cand1 = saved_cand1;
  /* :1190-1206 */
  {
    const cu_info_t *ref_cu = kvz_cu_array_at_const(state->frame->ref->cu_arrays[0], x + (width >> 1), y + (width >> 1));
    if (ref_cu->type == CU_INTER) {
      const int l = (ref_cu->inter.mv_dir & 1) ? 0 : 1;
      pu.extra_mv[0] = ref_cu->inter.mv[l][0]; pu.extra_mv[1] = ref_cu->inter.mv[l][1];
    }
  }
  /* planes of a new frame, the encoder settings, the position in the frame's dependency order */
  const int poc = state->frame->poc;
  if (g_rec.n_frames == 0 || g_rec.frame_poc[g_rec.n_frames - 1] != poc) {
    if (g_rec.n_frames >= g_rec.max_frames) { ++g_rec.skipped; __real_kvz_search_cu_inter(state, x, y, depth, lcu, inter_cost, inter_bitcost); return; }
    const size_t off = (size_t)g_rec.n_frames * g_rec.w * g_rec.h;
    rec_copy_plane(g_rec.pic + off, state->tile->frame->source, g_rec.w, g_rec.h);
    rec_copy_plane(g_rec.ref + off, state->frame->ref->images[0], g_rec.w, g_rec.h);
    g_rec.frame_poc[g_rec.n_frames++] = poc;
    rec_snapshot_frame(state, g_rec.n_frames - 1);
    g_rec.cur_lcu_x = g_rec.cur_lcu_y = -1;
    rec_params_t *p = &g_rec.params;
    memset(p, 0, sizeof(*p));
    p->early_termination = ctrl->cfg.me_early_termination;
    p->max_steps = ctrl->cfg.me_max_steps;
    p->fme_level = ctrl->cfg.fme_level;
    p->wpp_owf = ctrl->cfg.owf && ctrl->cfg.wpp;
    p->ref_delay_px = ctrl->cfg.sao_type ? SAO_DELAY_PX : (ctrl->cfg.deblock_enable ? DEBLOCK_DELAY_PX : 0);
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
  }
  const int lx = x / LCU_WIDTH, ly = y / LCU_WIDTH;
  if (lx != g_rec.cur_lcu_x || ly != g_rec.cur_lcu_y) { g_rec.cur_lcu_x = lx; g_rec.cur_lcu_y = ly; g_rec.seq = 0; }

  __real_kvz_search_cu_inter(state, x, y, depth, lcu, inter_cost, inter_bitcost);

  const int i = g_rec.count++;
  g_rec.pu[i] = pu;
  rec_result_t *r = &g_rec.res[i];
  memset(r, 0, sizeof(*r));
  if (*inter_cost < (double)MAX_INT) {
    r->mv[0] = cur_cu->inter.mv[0][0]; r->mv[1] = cur_cu->inter.mv[0][1];
    r->cost = (uint32_t)*inter_cost; r->bitcost = *inter_bitcost;
    r->merged = cur_cu->merged; r->merge_idx = cur_cu->merge_idx;
    r->mv_cand = cur_cu->merged ? 0 : cur_cu->inter.mv_cand0;
  } else {
    r->cost = 0xffffffffu;
  }
  rec_meta_t *m = &g_rec.meta[i];
  m->frame = g_rec.n_frames - 1; m->lcu_x = lx; m->lcu_y = ly; m->seq = g_rec.seq++;
  m->lambda_cost = (int32_t)(state->lambda_sqrt + 0.5); m->depth = depth;
}

/* start recording: room for max_records searches and max_frames frames of w x h luma */
int ref_record_begin(int max_records, int max_frames, int w, int h)
{
  memset(&g_rec, 0, sizeof(g_rec));
  g_rec.max = max_records; g_rec.max_frames = max_frames; g_rec.w = w; g_rec.h = h;
  g_rec.pu = calloc((size_t)max_records, sizeof(rec_pu_t));
  g_rec.res = calloc((size_t)max_records, sizeof(rec_result_t));
  g_rec.meta = calloc((size_t)max_records, sizeof(rec_meta_t));
  g_rec.frame_poc = calloc((size_t)max_frames, sizeof(int32_t));
  g_rec.pic = malloc((size_t)max_frames * w * h); g_rec.ref = malloc((size_t)max_frames * w * h);
  if (!g_rec.pu || !g_rec.res || !g_rec.meta || !g_rec.frame_poc || !g_rec.pic || !g_rec.ref) return -1;
  g_rec.on = 1;
  return 0;
}
/* stop; copies out what was recorded (any pointer may be NULL) and frees the recorder.  Returns the record count;
 * info[0] = frames, info[1] = searches that could not be recorded (not a single-reference P search, or no room) */
int ref_record_end(void *pus, void *results, void *meta, void *params, void *pic, void *ref, int *info)
{
  g_rec.on = 0;
  const int n = g_rec.count;
  if (pus) memcpy(pus, g_rec.pu, (size_t)n * sizeof(rec_pu_t));
  if (results) memcpy(results, g_rec.res, (size_t)n * sizeof(rec_result_t));
  if (meta) memcpy(meta, g_rec.meta, (size_t)n * sizeof(rec_meta_t));
  if (params) memcpy(params, &g_rec.params, sizeof(rec_params_t));
  if (pic) memcpy(pic, g_rec.pic, (size_t)g_rec.n_frames * g_rec.w * g_rec.h);
  if (ref) memcpy(ref, g_rec.ref, (size_t)g_rec.n_frames * g_rec.w * g_rec.h);
  if (info) { info[0] = g_rec.n_frames; info[1] = g_rec.skipped; }
  free(g_rec.pu); free(g_rec.res); free(g_rec.meta); free(g_rec.frame_poc); free(g_rec.pic); free(g_rec.ref);
  memset(&g_rec, 0, sizeof(g_rec));
  return n;
}

/* ------------------------------------------------------------------------
 * AMVP / merge candidate derivation (inter.c:1209-1446) on a state fabricated from flat SCU maps: the counterpart of
 * orc_inter_candidates (oracle/kvz_oracle.h) made of the reference's own kvz_inter_get_merge_cand /
 * kvz_inter_get_mv_cand calls, in the order and with the arguments search_pu_inter / search_pu_inter_ref use
 * (search_inter.c:1470-1500, :1143-1206).  The lcu_t of a PU is cut from the current picture's map the way lcu->cu is
 * laid out (cu.h:324-344: the LCU's 16x16 SCUs, one row above, one column left, the corner, the top-right SCU).
 * ------------------------------------------------------------------------ */
typedef struct {                              /* = orc_cu_info / kvz_hip_cu_info */
  uint8_t type, depth, part_size, tr_depth, cbf_y, mv_dir, qp, reserved;
  int16_t mv[2][2];
  uint8_t mv_ref[2], pad[2];
} flat_cu_t;
typedef struct {                              /* = orc_inter_params / kvz_hip_inter_params */
  int32_t poc, slice_is_b, tmvp_enable, num_refs;
  int32_t ref_pocs[16];
  uint8_t ref_LX[2][16];
  uint8_t ref_LX_size[2], pad[2];
  int32_t col_ref_pocs[16];
  uint8_t col_ref_LX[2][16];
  int32_t pic_width, pic_height, in_width, in_height, tile_x, tile_y, ref_idx, cus_stride, col_stride, reserved;
} flat_inter_params_t;
typedef struct { uint8_t dir, ref[2], pad; int16_t mv[2][2]; } flat_merge_t;

static void flat_to_cu(const flat_cu_t *f, cu_info_t *c)
{
  memset(c, 0, sizeof(*c));
  c->type = f->type; c->depth = f->depth; c->part_size = f->part_size; c->tr_depth = f->tr_depth; c->qp = f->qp;
  if (f->type == CU_INTER) {
    memcpy(c->inter.mv, f->mv, sizeof(c->inter.mv));
    c->inter.mv_ref[0] = f->mv_ref[0]; c->inter.mv_ref[1] = f->mv_ref[1];
    c->inter.mv_dir = f->mv_dir;
  }
}

static cu_array_t *flat_to_cua(const flat_cu_t *map, int stride, int rows)
{
  cu_array_t *a = calloc(1, sizeof(*a));
  a->data = calloc((size_t)stride * rows, sizeof(cu_info_t));
  a->width = stride * 4; a->height = rows * 4; a->stride = stride * 4; a->refcount = 1;
  for (int i = 0; i < stride * rows; ++i) flat_to_cu(&map[i], &a->data[i]);
  return a;
}

void ref_inter_candidates(const flat_cu_t *cus, const flat_cu_t *col_cus, const flat_cu_t *ref_cus, const flat_inter_params_t *p,
                          rec_pu_t *pus, size_t count, flat_merge_t *merge_out)
{
  static encoder_control_t ctrl;
  static encoder_state_t state;
  static encoder_state_config_frame_t frame;
  static encoder_state_config_tile_t tile;
  static videoframe_t vframe;
  static image_list_t refs;
  static kvz_picture pics[16];
  static kvz_picture *pic_ptr[16];
  static cu_array_t *cuas[16];
  static int32_t pocs[16];
  static uint8_t ref_LXs[16][2][16];
  memset(&ctrl, 0, sizeof(ctrl)); memset(&state, 0, sizeof(state)); memset(&frame, 0, sizeof(frame));
  memset(&tile, 0, sizeof(tile)); memset(&vframe, 0, sizeof(vframe)); memset(&refs, 0, sizeof(refs));
  memset(pics, 0, sizeof(pics)); memset(ref_LXs, 0, sizeof(ref_LXs));
  ctrl.cfg.tmvp_enable = p->tmvp_enable;
  ctrl.in.width = p->in_width; ctrl.in.height = p->in_height;
  state.encoder_control = &ctrl; state.frame = &frame; state.tile = &tile;
  tile.frame = &vframe; tile.offset_x = p->tile_x; tile.offset_y = p->tile_y;
  vframe.width = p->pic_width; vframe.height = p->pic_height;
  frame.poc = p->poc;
  frame.slicetype = p->slice_is_b ? KVZ_SLICE_B : KVZ_SLICE_P;
  memcpy(frame.ref_LX, p->ref_LX, sizeof(frame.ref_LX));
  frame.ref_LX_size[0] = p->ref_LX_size[0]; frame.ref_LX_size[1] = p->ref_LX_size[1];
  frame.ref = &refs;
  refs.images = pic_ptr; refs.cu_arrays = cuas; refs.pocs = pocs; refs.ref_LXs = ref_LXs;
  refs.size = 16; refs.used_size = (uint32_t)p->num_refs;
  const int col_rows = ((p->in_height + 63) / 64) * 16;
  const int col_pic = p->ref_LX_size[0] ? p->ref_LX[0][0] : -1;
  cu_array_t *col_a = col_cus ? flat_to_cua(col_cus, p->col_stride, col_rows) : NULL;
  cu_array_t *ref_a = ref_cus ? flat_to_cua(ref_cus, p->col_stride, col_rows) : NULL;
  cu_array_t *empty = calloc(1, sizeof(*empty));
  empty->data = calloc((size_t)p->col_stride * col_rows, sizeof(cu_info_t));
  empty->width = p->col_stride * 4; empty->height = col_rows * 4; empty->stride = empty->width;
  for (int i = 0; i < 16; ++i) {
    pic_ptr[i] = &pics[i]; pocs[i] = p->ref_pocs[i];
    cuas[i] = empty;
  }
  if (col_pic >= 0) {
    if (col_a) cuas[col_pic] = col_a;
    memcpy(pics[col_pic].ref_pocs, p->col_ref_pocs, sizeof(pics[col_pic].ref_pocs));
    memcpy(ref_LXs[col_pic], p->col_ref_LX, sizeof(ref_LXs[col_pic]));
  }
  if (ref_a && p->ref_idx >= 0 && p->ref_idx < 16 && !(p->ref_idx == col_pic && col_a)) cuas[p->ref_idx] = ref_a;

  /* search_pu_inter_ref :1143-1166 */
  int8_t ref_list = -1, LX_idx;
  const int8_t lx_max = MAX(frame.ref_LX_size[0], frame.ref_LX_size[1]);
  for (LX_idx = 0; LX_idx < lx_max; LX_idx++) {
    if (LX_idx < frame.ref_LX_size[0] && frame.ref_LX[0][LX_idx] == p->ref_idx) { ref_list = 0; break; }
    if (LX_idx < frame.ref_LX_size[1] && frame.ref_LX[1][LX_idx] == p->ref_idx) { ref_list = 1; break; }
  }

  lcu_t *lcu = calloc(1, sizeof(lcu_t));
  const int map_rows = (p->pic_height + 3) / 4;
  for (size_t n = 0; n < count; ++n) {
    rec_pu_t *u = &pus[n];
    const int x = u->x - p->tile_x, y = u->y - p->tile_y, w = u->width, h = u->height;    /* descriptors carry picture coordinates */
    const int ox = (x / LCU_WIDTH) * 16, oy = (y / LCU_WIDTH) * 16;     /* the LCU's first SCU */
    memset(lcu->cu, 0, sizeof(lcu->cu));
    for (int sy = -1; sy < 16; ++sy)
      for (int sx = -1; sx < 16; ++sx) {
        const int fx = ox + sx, fy = oy + sy;
        if (fx < 0 || fy < 0 || fx >= p->cus_stride || fy >= map_rows) continue;
        flat_to_cu(&cus[fy * p->cus_stride + fx], &lcu->cu[LCU_CU_OFFSET + sx + sy * LCU_T_CU_WIDTH]);
      }
    if (oy > 0 && ox + 16 < p->cus_stride) flat_to_cu(&cus[(oy - 1) * p->cus_stride + ox + 16], LCU_GET_TOP_RIGHT_CU(lcu));
    cu_info_t *cur_cu = LCU_GET_CU_AT_PX(lcu, SUB_SCU(x), SUB_SCU(y));

    inter_merge_cand_t merge[MRG_MAX_NUM_CANDS];
    memset(merge, 0, sizeof(merge));
    const int n_merge = kvz_inter_get_merge_cand(&state, x, y, w, h, !(u->pad & 1), !(u->pad & 2), merge, lcu);
    u->num_merge_cand = (int16_t)n_merge;
    memset(u->merge, 0, sizeof(u->merge));
    for (int i = 0; i < n_merge; ++i) {
      const int dir = merge[i].dir;
      u->merge[i].usable = dir != 3;
      if (dir != 3) {
        u->merge[i].mv[0] = merge[i].mv[dir - 1][0]; u->merge[i].mv[1] = merge[i].mv[dir - 1][1];
        u->merge[i].same_ref = frame.ref_LX[dir - 1][merge[i].ref[dir - 1]] == p->ref_idx;
      }
    }
    if (merge_out)
      for (int i = 0; i < 5; ++i) {
        flat_merge_t *o = &merge_out[5 * n + i];
        memset(o, 0, sizeof(*o));
        o->dir = merge[i].dir; o->ref[0] = merge[i].ref[0]; o->ref[1] = merge[i].ref[1];
        memcpy(o->mv, merge[i].mv, sizeof(o->mv));
      }
    memset(u->mv_cand, 0, sizeof(u->mv_cand));
    if (ref_list >= 0) {
      cur_cu->inter.mv_ref[ref_list] = LX_idx;
      kvz_inter_get_mv_cand(&state, x, y, w, h, u->mv_cand, cur_cu, lcu, ref_list);
    }
    u->extra_mv[0] = u->extra_mv[1] = 0;
    if (ref_cus && p->ref_idx >= 0 && p->ref_idx < p->num_refs) {
      const cu_info_t *ref_cu = kvz_cu_array_at_const(refs.cu_arrays[p->ref_idx], p->tile_x + x + (w >> 1), p->tile_y + y + (h >> 1));
      if (ref_cu->type == CU_INTER) {
        const int l = (ref_cu->inter.mv_dir & 1) ? 0 : 1;
        u->extra_mv[0] = ref_cu->inter.mv[l][0]; u->extra_mv[1] = ref_cu->inter.mv[l][1];
      }
    }
  }
  free(lcu);
  if (col_a) { free(col_a->data); free(col_a); }
  if (ref_a) { free(ref_a->data); free(ref_a); }
  free(empty->data); free(empty);
}


/* ------------------------------------------------------------------------
 * Snapshots for the recorder above: what kvz_inter_get_merge_cand / kvz_inter_get_mv_cand READ when the encoder derived the
 * recorded candidates -- lcu->cu as it stood (the 17 x 17 + 1 records of cu.h:324: decided neighbours, work-tree leftovers)
 * and, per frame, the collocated picture's CU array with the POC tables.  The candidate derivation of the oracle and of
 * the GPU entry is then checked against the candidates the encoder really used, on the states a real encode goes through.
 * ------------------------------------------------------------------------ */
static struct {
  int max, count;
  int32_t *index;                               /* record index of each snapshot */
  flat_cu_t *cu;                                /* [max][LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1] */
  int max_frames, col_stride, col_rows;
  flat_cu_t *col;                               /* [frames][col_rows][col_stride] */
  flat_inter_params_t *params;                  /* [frames] */
} g_snap;

static void cu_to_flat(const cu_info_t *c, flat_cu_t *f)
{
  memset(f, 0, sizeof(*f));
  f->type = c->type; f->depth = c->depth; f->part_size = c->part_size; f->tr_depth = c->tr_depth; f->qp = c->qp;
  if (c->type == CU_INTER) {
    f->mv_dir = c->inter.mv_dir;
    memcpy(f->mv, c->inter.mv, sizeof(f->mv));
    f->mv_ref[0] = c->inter.mv_ref[0]; f->mv_ref[1] = c->inter.mv_ref[1];
  }
}

int ref_record_snapshots(int max_snapshots)
{
  memset(&g_snap, 0, sizeof(g_snap));
  if (!g_rec.on) return -1;
  g_snap.max = max_snapshots;
  g_snap.max_frames = g_rec.max_frames;
  g_snap.col_stride = ((g_rec.w + 63) / 64) * 16; g_snap.col_rows = ((g_rec.h + 63) / 64) * 16;
  g_snap.index = calloc((size_t)max_snapshots, sizeof(int32_t));
  g_snap.cu = calloc((size_t)max_snapshots * (LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1), sizeof(flat_cu_t));
  g_snap.col = calloc((size_t)g_snap.max_frames * g_snap.col_rows * g_snap.col_stride, sizeof(flat_cu_t));
  g_snap.params = calloc((size_t)g_snap.max_frames, sizeof(flat_inter_params_t));
  return (g_snap.index && g_snap.cu && g_snap.col && g_snap.params) ? 0 : -1;
}

static void rec_snapshot_search(const encoder_state_t *state, const lcu_t *lcu, int record_index)
{
  (void)state;
  if (!g_snap.cu || g_snap.count >= g_snap.max) return;
  if (g_snap.count && g_snap.index[g_snap.count - 1] == record_index) --g_snap.count;      /* a search that ended unrecorded */
  const int n = LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1;
  g_snap.index[g_snap.count] = record_index;
  for (int i = 0; i < n; ++i) cu_to_flat(&lcu->cu[i], &g_snap.cu[(size_t)g_snap.count * n + i]);
  ++g_snap.count;
}

static void rec_snapshot_frame(const encoder_state_t *state, int f)
{
  if (!g_snap.col || f >= g_snap.max_frames) return;
  const encoder_state_config_frame_t *fr = state->frame;
  flat_inter_params_t *p = &g_snap.params[f];
  memset(p, 0, sizeof(*p));
  p->poc = fr->poc; p->slice_is_b = fr->slicetype == KVZ_SLICE_B; p->tmvp_enable = state->encoder_control->cfg.tmvp_enable;
  p->num_refs = (int32_t)fr->ref->used_size;
  for (int i = 0; i < p->num_refs && i < 16; ++i) p->ref_pocs[i] = fr->ref->pocs[i];
  memcpy(p->ref_LX, fr->ref_LX, sizeof(p->ref_LX));
  p->ref_LX_size[0] = fr->ref_LX_size[0]; p->ref_LX_size[1] = fr->ref_LX_size[1];
  p->pic_width = state->tile->frame->width; p->pic_height = state->tile->frame->height;
  p->in_width = state->encoder_control->in.width; p->in_height = state->encoder_control->in.height;
  p->tile_x = state->tile->offset_x; p->tile_y = state->tile->offset_y;
  p->ref_idx = 0;
  p->cus_stride = g_snap.col_stride; p->col_stride = g_snap.col_stride;
  if (fr->ref_LX_size[0] > 0) {
    const int c = fr->ref_LX[0][0];
    memcpy(p->col_ref_pocs, fr->ref->images[c]->ref_pocs, sizeof(p->col_ref_pocs));
    memcpy(p->col_ref_LX, fr->ref->ref_LXs[c], sizeof(p->col_ref_LX));
    const cu_array_t *a = fr->ref->cu_arrays[c];
    flat_cu_t *dst = g_snap.col + (size_t)f * g_snap.col_rows * g_snap.col_stride;
    for (int y = 0; y < g_snap.col_rows && y * 4 < a->height; ++y)
      for (int x = 0; x < g_snap.col_stride && x * 4 < a->width; ++x)
        cu_to_flat(&a->data[x + y * (a->stride >> 2)], &dst[y * g_snap.col_stride + x]);
  }
}

/* copies the snapshots out (call BEFORE ref_record_end); returns their count.  dims[0..1] = rows, stride of a col map */
int ref_record_snapshots_get(int32_t *index, void *cu, void *col, void *params, int *dims)
{
  const int n = g_snap.count;
  const size_t per = LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1;
  if (index) memcpy(index, g_snap.index, (size_t)n * sizeof(int32_t));
  if (cu) memcpy(cu, g_snap.cu, (size_t)n * per * sizeof(flat_cu_t));
  if (col) memcpy(col, g_snap.col, (size_t)g_rec.n_frames * g_snap.col_rows * g_snap.col_stride * sizeof(flat_cu_t));
  if (params) memcpy(params, g_snap.params, (size_t)g_rec.n_frames * sizeof(flat_inter_params_t));
  if (dims) { dims[0] = g_snap.col_rows; dims[1] = g_snap.col_stride; }
  free(g_snap.index); free(g_snap.cu); free(g_snap.col); free(g_snap.params);
  memset(&g_snap, 0, sizeof(g_snap));
  return n;
}


/* ------------------------------------------------------------------------
 * The encoder's 2Nx2N inter searches SERVED BY THE GPU CHAIN (tests only): with ref_gpu_search_begin, the wrapper of
 * kvz_search_cu_inter above does not run the reference's search_pu_inter at all for the searches it can express
 * (P and B slices with any number of reference pictures, any number of tiles, no mv-rdo) but
 *   1. copies what the candidate derivation would read -- lcu->cu -- into the picture's CU array on the device
 *      (the frame's planes and the reference pictures' CU arrays go up once per frame),
 *   2. for every reference picture in turn, as search_pu_inter does (search_inter.c:1502-1507), runs
 *      kvz_hip_inter_candidates_batch and kvz_hip_search_pu_batch back to back on one stream, the best cost so far
 *      as the cost to beat (search_inter.c:1239),
 *   3. writes the decision into cur_cu and the two cost outputs exactly where search_pu_inter_ref does
 *      (search_inter.c:1275-1290, :1497-1499).
 * The encode must then produce the bitstream of the untouched encoder: every later decision of the encoder consumes
 * these results.  One PU per launch -- a correctness path, the throughput form is a front of PUs per launch.
 * ------------------------------------------------------------------------ */
#define GPU_MAX_REFS 16
static struct {
  int on, w, h, stride, rows, poc_loaded;
  void *lib;
  int (*init)(int);
  void *(*dmalloc)(size_t);
  void (*dfree)(void *);
  int (*h2d)(void *, const void *, size_t, kvz_hip_stream);
  int (*d2h)(void *, const void *, size_t, kvz_hip_stream);
  int (*cand)(const kvz_hip_cu_info *, const kvz_hip_cu_info *, const kvz_hip_cu_info *, const kvz_hip_inter_params *, kvz_hip_me_pu *, size_t,
              kvz_hip_merge_cand *, kvz_hip_stream);
  int (*search)(const kvz_hip_pixel *, uint32_t, int, int, const kvz_hip_pixel *, uint32_t, int, int, const kvz_hip_me_pu *, size_t,
                const kvz_hip_me_params *, kvz_hip_me_result *, kvz_hip_stream);
  int (*bipred)(const kvz_hip_pixel *, uint32_t, int, int, const kvz_hip_pixel *, uint32_t, const kvz_hip_pixel *, uint32_t, int, int,
                const kvz_hip_bipred_cand *, size_t, uint32_t *, kvz_hip_stream);
  const char *(*last_error)(void);
  kvz_hip_merge_cand *d_merge;
  kvz_hip_bipred_cand *d_bcand;
  uint32_t *d_bcost;
  long bipred_pairs, bipred_launches;
  uint8_t *d_pic, *d_ref[GPU_MAX_REFS], *h_plane;
  kvz_hip_cu_info *d_cus, *d_refcus[GPU_MAX_REFS], *h_cus, *h_col;
  kvz_hip_me_pu *d_pu;
  kvz_hip_me_result *d_res;
  uint32_t *d_beat;
  kvz_hip_inter_params ip;
  kvz_hip_me_params mp;
  long served, passed_on, failed, launches;
  /* intra: the LCU's surroundings as a small picture, one PU's original block, its references and 35 + 35 costs */
  int (*build_ref)(int, int, const kvz_hip_pixel *, int, int, int, const kvz_hip_intra_pos *, size_t, kvz_hip_intra_ref *, kvz_hip_stream);
  int (*rough)(int, int, const kvz_hip_intra_ref *, const kvz_hip_pixel *, size_t, uint32_t *, uint32_t *, kvz_hip_stream);
  uint8_t *d_vplane, *d_orig, *h_vplane;
  kvz_hip_intra_pos *d_pos;
  kvz_hip_intra_ref *d_iref;
  uint32_t *d_icost;
  long intra_served, intra_passed_on;
} g_gpu;
enum { VPLANE_W = 192, VPLANE_H = 128 };

int ref_gpu_search_begin(const char *lib_path, int w, int h)
{
  memset(&g_gpu, 0, sizeof(g_gpu));
  void *l = dlopen(lib_path, RTLD_NOW | RTLD_GLOBAL);
  if (!l) { fprintf(stderr, "dlopen %s: %s\n", lib_path, dlerror()); return -1; }
  g_gpu.lib = l;
  *(void **)&g_gpu.init = dlsym(l, "kvz_hip_init");
  *(void **)&g_gpu.dmalloc = dlsym(l, "kvz_hip_malloc");
  *(void **)&g_gpu.dfree = dlsym(l, "kvz_hip_free");
  *(void **)&g_gpu.h2d = dlsym(l, "kvz_hip_memcpy_h2d");
  *(void **)&g_gpu.d2h = dlsym(l, "kvz_hip_memcpy_d2h");
  *(void **)&g_gpu.cand = dlsym(l, "kvz_hip_inter_candidates_batch");
  *(void **)&g_gpu.search = dlsym(l, "kvz_hip_search_pu_batch");
  *(void **)&g_gpu.last_error = dlsym(l, "kvz_hip_last_error");
  *(void **)&g_gpu.bipred = dlsym(l, "kvz_hip_bipred_cost_batch");
  *(void **)&g_gpu.build_ref = dlsym(l, "kvz_hip_intra_build_reference_batch");
  *(void **)&g_gpu.rough = dlsym(l, "kvz_hip_intra_rough_batch");
  if (!g_gpu.init || !g_gpu.dmalloc || !g_gpu.dfree || !g_gpu.h2d || !g_gpu.d2h || !g_gpu.cand || !g_gpu.search || !g_gpu.last_error ||
      !g_gpu.build_ref || !g_gpu.rough || !g_gpu.bipred) return -1;
  if (g_gpu.init(-1) != KVZ_HIP_OK) { fprintf(stderr, "kvz_hip_init: %s\n", g_gpu.last_error()); return -1; }
  g_gpu.w = w; g_gpu.h = h;
  g_gpu.stride = ((w + 63) / 64) * 16; g_gpu.rows = ((h + 63) / 64) * 16;
  const size_t map_bytes = (size_t)g_gpu.stride * g_gpu.rows * sizeof(kvz_hip_cu_info);
  g_gpu.d_pic = g_gpu.dmalloc((size_t)w * h);
  g_gpu.d_cus = g_gpu.dmalloc(map_bytes);
  g_gpu.d_pu = g_gpu.dmalloc(sizeof(kvz_hip_me_pu)); g_gpu.d_res = g_gpu.dmalloc(sizeof(kvz_hip_me_result));
  g_gpu.d_beat = g_gpu.dmalloc(sizeof(uint32_t));
  g_gpu.d_merge = g_gpu.dmalloc(5 * sizeof(kvz_hip_merge_cand)); g_gpu.d_bcand = g_gpu.dmalloc(12 * sizeof(kvz_hip_bipred_cand));
  g_gpu.d_bcost = g_gpu.dmalloc(12 * sizeof(uint32_t));
  g_gpu.h_plane = malloc((size_t)w * h);
  g_gpu.h_cus = calloc(1, map_bytes); g_gpu.h_col = calloc(1, map_bytes);
  g_gpu.d_vplane = g_gpu.dmalloc(VPLANE_W * VPLANE_H); g_gpu.h_vplane = calloc(1, VPLANE_W * VPLANE_H);
  g_gpu.d_orig = g_gpu.dmalloc(32 * 32); g_gpu.d_pos = g_gpu.dmalloc(sizeof(kvz_hip_intra_pos));
  g_gpu.d_iref = g_gpu.dmalloc(sizeof(kvz_hip_intra_ref)); g_gpu.d_icost = g_gpu.dmalloc(70 * sizeof(uint32_t));
  if (!g_gpu.d_merge || !g_gpu.d_bcand || !g_gpu.d_bcost) return -1;
  if (!g_gpu.d_pic || !g_gpu.d_cus || !g_gpu.d_pu || !g_gpu.d_res || !g_gpu.d_beat || !g_gpu.h_plane || !g_gpu.h_cus || !g_gpu.h_col ||
      !g_gpu.d_vplane || !g_gpu.h_vplane || !g_gpu.d_orig || !g_gpu.d_pos || !g_gpu.d_iref || !g_gpu.d_icost) return -1;
  g_gpu.poc_loaded = -1;
  g_gpu.on = 1;
  return 0;
}

/* out[0..7] (out[6] = bi-prediction candidate pairs scored by kvz_hip_bipred_cost_batch, out[7] = in that many calls); out[0..5] = inter searches served by the GPU chain, inter searches passed on to the reference, GPU calls that failed,
 * (candidates + search) launch pairs issued (one per reference picture of a served search), intra searches served, passed on */
void ref_gpu_search_end(long *out)
{
  if (out) { out[0] = g_gpu.served; out[1] = g_gpu.passed_on; out[2] = g_gpu.failed; out[3] = g_gpu.launches;
             out[4] = g_gpu.intra_served; out[5] = g_gpu.intra_passed_on; out[6] = g_gpu.bipred_pairs; out[7] = g_gpu.bipred_launches; }
  if (g_gpu.lib) {
    g_gpu.dfree(g_gpu.d_merge); g_gpu.dfree(g_gpu.d_bcand); g_gpu.dfree(g_gpu.d_bcost);
    g_gpu.dfree(g_gpu.d_vplane); g_gpu.dfree(g_gpu.d_orig); g_gpu.dfree(g_gpu.d_pos); g_gpu.dfree(g_gpu.d_iref); g_gpu.dfree(g_gpu.d_icost);
    g_gpu.dfree(g_gpu.d_pic); g_gpu.dfree(g_gpu.d_cus); g_gpu.dfree(g_gpu.d_pu); g_gpu.dfree(g_gpu.d_res); g_gpu.dfree(g_gpu.d_beat);
    for (int i = 0; i < GPU_MAX_REFS; ++i) { g_gpu.dfree(g_gpu.d_ref[i]); g_gpu.dfree(g_gpu.d_refcus[i]); }
  }
  free(g_gpu.h_plane); free(g_gpu.h_cus); free(g_gpu.h_col); free(g_gpu.h_vplane);
  memset(&g_gpu, 0, sizeof(g_gpu));
}

static void cu_to_hip(const cu_info_t *c, kvz_hip_cu_info *f)
{
  memset(f, 0, sizeof(*f));
  f->type = c->type; f->depth = c->depth; f->part_size = c->part_size; f->tr_depth = c->tr_depth; f->qp = c->qp;
  if (c->type == CU_INTER) {
    f->mv_dir = c->inter.mv_dir;
    memcpy(f->mv, c->inter.mv, sizeof(f->mv));
    f->mv_ref[0] = c->inter.mv_ref[0]; f->mv_ref[1] = c->inter.mv_ref[1];
  }
}

/* per frame: the luma planes, the reference pictures' CU arrays, the state both entries read */
static int gpu_load_frame(const encoder_state_t *state)
{
  const encoder_control_t *ctrl = state->encoder_control;
  const encoder_state_config_frame_t *fr = state->frame;
  const int w = g_gpu.w, h = g_gpu.h, nref = (int)fr->ref->used_size;
  const size_t map_bytes = (size_t)g_gpu.stride * g_gpu.rows * sizeof(kvz_hip_cu_info);
  int bad = 0;
  const kvz_picture *src = state->tile->frame->source->base_image;     /* a tile's source is a view of the whole picture */
  for (int r = 0; r < h; ++r) memcpy(g_gpu.h_plane + (size_t)r * w, src->y + (size_t)r * src->stride, (size_t)w);
  bad |= g_gpu.h2d(g_gpu.d_pic, g_gpu.h_plane, (size_t)w * h, NULL);
  for (int i = 0; i < nref; ++i) {
    if (!g_gpu.d_ref[i]) { g_gpu.d_ref[i] = g_gpu.dmalloc((size_t)w * h); g_gpu.d_refcus[i] = g_gpu.dmalloc(map_bytes); }
    if (!g_gpu.d_ref[i] || !g_gpu.d_refcus[i]) return 1;
    const kvz_picture *ref = fr->ref->images[i];
    for (int r = 0; r < h; ++r) memcpy(g_gpu.h_plane + (size_t)r * w, ref->y + (size_t)r * ref->stride, (size_t)w);
    bad |= g_gpu.h2d(g_gpu.d_ref[i], g_gpu.h_plane, (size_t)w * h, NULL);
    const cu_array_t *a = fr->ref->cu_arrays[i];
    memset(g_gpu.h_col, 0, map_bytes);
    for (int sy = 0; sy < g_gpu.rows && sy * 4 < a->height; ++sy)
      for (int sx = 0; sx < g_gpu.stride && sx * 4 < a->width; ++sx)
        cu_to_hip(&a->data[sx + sy * (a->stride >> 2)], &g_gpu.h_col[sy * g_gpu.stride + sx]);
    bad |= g_gpu.h2d(g_gpu.d_refcus[i], g_gpu.h_col, map_bytes, NULL);
  }
  kvz_hip_inter_params *ip = &g_gpu.ip;
  memset(ip, 0, sizeof(*ip));
  ip->poc = fr->poc; ip->slice_is_b = fr->slicetype == KVZ_SLICE_B; ip->tmvp_enable = ctrl->cfg.tmvp_enable; ip->num_refs = nref;
  for (int i = 0; i < nref; ++i) ip->ref_pocs[i] = fr->ref->pocs[i];
  memcpy(ip->ref_LX, fr->ref_LX, sizeof(ip->ref_LX));
  ip->ref_LX_size[0] = fr->ref_LX_size[0]; ip->ref_LX_size[1] = fr->ref_LX_size[1];
  if (fr->ref_LX_size[0] > 0) {
    const int c = fr->ref_LX[0][0];
    memcpy(ip->col_ref_pocs, fr->ref->images[c]->ref_pocs, sizeof(ip->col_ref_pocs));
    memcpy(ip->col_ref_LX, fr->ref->ref_LXs[c], sizeof(ip->col_ref_LX));
  }
  ip->in_width = ctrl->in.width; ip->in_height = ctrl->in.height;      /* the tile's own fields are set per search */
  ip->cus_stride = g_gpu.stride; ip->col_stride = g_gpu.stride;
  kvz_hip_me_params *p = &g_gpu.mp;
  memset(p, 0, sizeof(*p));
  p->early_termination = ctrl->cfg.me_early_termination;
  p->max_steps = ctrl->cfg.me_max_steps;
  p->fme_level = ctrl->cfg.fme_level;
  p->wpp_owf = ctrl->cfg.owf && ctrl->cfg.wpp;
  p->ref_delay_px = ctrl->cfg.sao_type ? SAO_DELAY_PX : (ctrl->cfg.deblock_enable ? DEBLOCK_DELAY_PX : 0);
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
  p->cost_to_beat = g_gpu.d_beat;
  return bad;
}

uint32_t refme_calc_mvd_cost(const encoder_state_t *state, int x, int y, int mv_shift, int16_t mv_cand[2][2], uint32_t *bitcost);
int refme_select_mv_cand(const encoder_state_t *state, int16_t mv_cand[2][2], int32_t mv_x, int32_t mv_y);
int refme_fracmv_within_tile(const encoder_state_t *state, int origin_x, int origin_y, int width, int height, int mv_x, int mv_y);

static void gpu_flush_deblock(const encoder_state_t *state);

/* which searches the two entries can answer */
static int gpu_can_serve_inter(const encoder_state_t *state)
{
  const encoder_control_t *ctrl = state->encoder_control;
  const encoder_state_config_frame_t *fr = state->frame;
  const int nref = (int)fr->ref->used_size;
  return nref >= 1 && nref <= GPU_MAX_REFS && fr->slicetype != KVZ_SLICE_I &&
         !ctrl->cfg.mv_rdo && ctrl->in.width == g_gpu.w && ctrl->in.height == g_gpu.h;
}

/* search_pu_inter (search_inter.c:1451-1520) for one PU of any shape, the two entries in place of search_pu_inter_ref's
 * middle.  Returns 0 when a GPU call failed (nothing of the encoder's state is changed then). */
static int gpu_serve_pu(encoder_state_t *state, int x, int y, int width, int height, int merge_a1, int merge_b1, lcu_t *lcu,
                        double *inter_cost, uint32_t *inter_bitcost)
{
  const encoder_state_config_frame_t *fr = state->frame;
  const int nref = (int)fr->ref->used_size;
  const int w = g_gpu.w, h = g_gpu.h;
  int bad = 0;
  if (g_gpu.poc_loaded != fr->poc) { bad |= gpu_load_frame(state); g_gpu.poc_loaded = fr->poc; }
  /* 1. lcu->cu into the picture's CU array: the LCU's 16 x 16 SCUs, the row above, the column to the left, the corner,
   *    the top-right SCU (cu.h:324-344); the rows that changed go to the device in one copy */
  const int ox = (x / LCU_WIDTH) * 16, oy = (y / LCU_WIDTH) * 16;
  for (int sy = -1; sy < 16; ++sy)
    for (int sx = -1; sx < 16; ++sx) {
      const int fx = ox + sx, fy = oy + sy;
      if (fx < 0 || fy < 0 || fx >= g_gpu.stride || fy >= g_gpu.rows) continue;
      cu_to_hip(&lcu->cu[LCU_CU_OFFSET + sx + sy * LCU_T_CU_WIDTH], &g_gpu.h_cus[fy * g_gpu.stride + fx]);
    }
  if (oy > 0 && ox + 16 < g_gpu.stride) cu_to_hip(LCU_GET_TOP_RIGHT_CU(lcu), &g_gpu.h_cus[(oy - 1) * g_gpu.stride + ox + 16]);
  const int r0 = oy > 0 ? oy - 1 : 0, r1 = oy + 16 < g_gpu.rows ? oy + 16 : g_gpu.rows;
  bad |= g_gpu.h2d(g_gpu.d_cus + (size_t)r0 * g_gpu.stride, g_gpu.h_cus + (size_t)r0 * g_gpu.stride,
                   (size_t)(r1 - r0) * g_gpu.stride * sizeof(kvz_hip_cu_info), NULL);
  /* 2. + 3.: every reference picture in turn (search_inter.c:1502-1507) */
  cu_info_t *cur_cu = LCU_GET_CU_AT_PX(lcu, SUB_SCU(x), SUB_SCU(y));
  const cu_info_t saved = *cur_cu;
  double cost = MAX_INT;
  uint32_t bitcost = MAX_INT;
  CU_SET_MV_CAND(cur_cu, 0, 0);
  CU_SET_MV_CAND(cur_cu, 1, 0);
  /* the tile: the CU array on the device is the current tile's (tile-relative, like lcu->cu), descriptors carry picture
   * coordinates, the search and the candidate derivation are told where the tile lies */
  const int tile_x = state->tile->offset_x, tile_y = state->tile->offset_y;
  g_gpu.ip.pic_width = state->tile->frame->width; g_gpu.ip.pic_height = state->tile->frame->height;
  g_gpu.ip.tile_x = tile_x; g_gpu.ip.tile_y = tile_y;
  kvz_hip_me_params mp = g_gpu.mp;
  mp.tile_x = tile_x; mp.tile_y = tile_y; mp.tile_w = state->tile->frame->width; mp.tile_h = state->tile->frame->height;
  mp.lambda_cost = (int32_t)(state->lambda_sqrt + 0.5);
  const int longer = width > height ? width : height;
  mp.size_classes = longer <= 16 ? 1 : (longer <= 32 ? 2 : 4);
  const int8_t lx_max = MAX(fr->ref_LX_size[0], fr->ref_LX_size[1]);
  for (int ref_idx = 0; ref_idx < nref && !bad; ++ref_idx) {
    int8_t ref_list = -1, LX_idx;                      /* :1143-1166 */
    for (LX_idx = 0; LX_idx < lx_max; LX_idx++) {
      if (LX_idx < fr->ref_LX_size[0] && fr->ref_LX[0][LX_idx] == ref_idx) { ref_list = 0; break; }
      if (LX_idx < fr->ref_LX_size[1] && fr->ref_LX[1][LX_idx] == ref_idx) { ref_list = 1; break; }
    }
    if (ref_list < 0) { bad = 1; break; }
    kvz_hip_me_pu pu;
    memset(&pu, 0, sizeof(pu));
    pu.x = tile_x + x; pu.y = tile_y + y; pu.width = width; pu.height = height;
    pu.pad = (int16_t)((merge_a1 ? 0 : 1) | (merge_b1 ? 0 : 2));
    const uint32_t beat = (uint32_t)cost;               /* *inter_cost as search_pu_inter_ref finds it (:1239) */
    kvz_hip_me_result res;
    memset(&res, 0, sizeof(res));
    g_gpu.ip.ref_idx = ref_idx;
    const int c = fr->ref_LX_size[0] > 0 ? fr->ref_LX[0][0] : 0;
    bad |= g_gpu.h2d(g_gpu.d_pu, &pu, sizeof(pu), NULL);
    bad |= g_gpu.h2d(g_gpu.d_beat, &beat, sizeof(beat), NULL);
    bad |= g_gpu.cand(g_gpu.d_cus, g_gpu.d_refcus[c], g_gpu.d_refcus[ref_idx], &g_gpu.ip, g_gpu.d_pu, 1, g_gpu.d_merge, NULL);
    bad |= g_gpu.search(g_gpu.d_pic, (uint32_t)w, w, h, g_gpu.d_ref[ref_idx], (uint32_t)w, w, h, g_gpu.d_pu, 1, &mp, g_gpu.d_res, NULL);
    bad |= g_gpu.d2h(&res, g_gpu.d_res, sizeof(res), NULL);
    ++g_gpu.launches;
    if (bad || res.reserved == -1) { bad = 1; break; }
    if (res.cost != 0xffffffffu && res.cost < cost) {  /* :1275-1290 */
      cur_cu->inter.mv_dir = ref_list + 1;
      cur_cu->merged = (uint8_t)res.merged;
      cur_cu->merge_idx = (uint8_t)res.merge_idx;
      cur_cu->inter.mv_ref[ref_list] = LX_idx;
      cur_cu->inter.mv[ref_list][0] = (int16_t)res.mv[0];
      cur_cu->inter.mv[ref_list][1] = (int16_t)res.mv[1];
      CU_SET_MV_CAND(cur_cu, ref_list, res.mv_cand);
      cost = res.cost;
      bitcost = res.bitcost + cur_cu->inter.mv_dir - 1 + LX_idx;
    }
  }
  /* search_pu_inter_bipred (search_inter.c:1304-1440): pairs of an L0 and an L1 merge candidate, each scored by
   * kvz_hip_bipred_cost_batch (the blended prediction's SATD) + the reference's own MV bit costs (:1509-1516) */
  if (!bad && fr->slicetype == KVZ_SLICE_B && state->encoder_control->cfg.bipred && width + height >= 16) {
    kvz_hip_merge_cand mc[5];
    kvz_hip_me_pu last;
    bad |= g_gpu.d2h(mc, g_gpu.d_merge, sizeof(mc), NULL);
    bad |= g_gpu.d2h(&last, g_gpu.d_pu, sizeof(last), NULL);        /* info->mv_cand as the last picture's search left it */
    int16_t mv_cand[2][2];
    memcpy(mv_cand, last.mv_cand, sizeof(mv_cand));
    static const uint8_t first[12] = { 0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3 }, second[12] = { 1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2 };
    const int n_merge = last.num_merge_cand;
    const unsigned pairs = MIN(n_merge * (n_merge - 1), 12);
    /* which pairs the loop of :1326-1350 scores, and their pictures */
    int valid[12], pic0[12], pic1[12];
    uint32_t satd[12];
    kvz_hip_bipred_cand bc[12];
    unsigned stop = pairs;
    for (unsigned idx = 0; idx < pairs; ++idx) {
      const int i = first[idx], j = second[idx];
      valid[idx] = 0;
      if (i >= n_merge || j >= n_merge) { stop = idx; break; }
      if (!(mc[i].dir & 1) || !(mc[j].dir & 2)) continue;
      if (fr->ref_LX[0][mc[i].ref[0]] == fr->ref_LX[1][mc[j].ref[1]] && mc[i].mv[0][0] == mc[j].mv[1][0] && mc[i].mv[0][1] == mc[j].mv[1][1]) continue;
      if (!refme_fracmv_within_tile(state, x, y, width, height, mc[i].mv[0][0], mc[i].mv[0][1]) ||
          !refme_fracmv_within_tile(state, x, y, width, height, mc[j].mv[1][0], mc[j].mv[1][1])) continue;
      valid[idx] = 1;
      pic0[idx] = fr->ref_LX[0][mc[i].ref[0]]; pic1[idx] = fr->ref_LX[1][mc[j].ref[1]];
      memset(&bc[idx], 0, sizeof(bc[idx]));
      bc[idx].x = tile_x + x; bc[idx].y = tile_y + y; bc[idx].width = width; bc[idx].height = height;
      bc[idx].mv0[0] = mc[i].mv[0][0]; bc[idx].mv0[1] = mc[i].mv[0][1]; bc[idx].mv1[0] = mc[j].mv[1][0]; bc[idx].mv1[1] = mc[j].mv[1][1];
    }
    /* the blended predictions' SATDs do not depend on one another: every pair that shares its two pictures goes into ONE
     * kvz_hip_bipred_cost_batch call */
    int done[12] = { 0 };
    for (unsigned a = 0; a < stop && !bad; ++a) {
      if (!valid[a] || done[a]) continue;
      kvz_hip_bipred_cand group[12];
      unsigned member[12], n = 0;
      for (unsigned b = a; b < stop; ++b)
        if (valid[b] && !done[b] && pic0[b] == pic0[a] && pic1[b] == pic1[a]) { group[n] = bc[b]; member[n++] = b; done[b] = 1; }
      uint32_t out[12];
      bad |= g_gpu.h2d(g_gpu.d_bcand, group, n * sizeof(group[0]), NULL);
      bad |= g_gpu.bipred(g_gpu.d_pic, (uint32_t)w, w, h, g_gpu.d_ref[pic0[a]], (uint32_t)w, g_gpu.d_ref[pic1[a]], (uint32_t)w, w, h,
                          g_gpu.d_bcand, n, g_gpu.d_bcost, NULL);
      bad |= g_gpu.d2h(out, g_gpu.d_bcost, n * sizeof(out[0]), NULL);
      ++g_gpu.bipred_launches;
      g_gpu.bipred_pairs += n;
      for (unsigned k = 0; k < n; ++k) { satd[member[k]] = out[k]; if (out[k] == 0xffffffffu) bad = 1; }
    }
    /* the decisions in the reference's order: a winning pair changes info->mv_cand and with it the bit costs of the pairs after it */
    for (unsigned idx = 0; idx < stop && !bad; ++idx) {
      if (!valid[idx]) continue;
      const int i = first[idx], j = second[idx];
      int16_t mv[2][2] = { { mc[i].mv[0][0], mc[i].mv[0][1] }, { mc[j].mv[1][0], mc[j].mv[1][1] } };
      uint32_t pair_cost = satd[idx];
      uint32_t bits[2] = { 0, 0 };
      pair_cost += refme_calc_mvd_cost(state, mc[i].mv[0][0], mc[i].mv[0][1], 0, mv_cand, &bits[0]);
      pair_cost += refme_calc_mvd_cost(state, mc[i].mv[1][0], mc[i].mv[1][1], 0, mv_cand, &bits[1]);     /* [i], as :1379-1386 has it */
      const int extra_bits = mc[i].ref[0] + mc[j].ref[1] + 2;
      pair_cost += state->lambda_sqrt * extra_bits + 0.5;
      if (pair_cost < cost) {
        cur_cu->inter.mv_dir = 3;
        cur_cu->inter.mv_ref[0] = mc[i].ref[0];
        cur_cu->inter.mv_ref[1] = mc[j].ref[1];
        memcpy(cur_cu->inter.mv, mv, sizeof(mv));
        cur_cu->merged = 0;
        for (int m = 0; m < n_merge; ++m)
          if (mc[m].mv[0][0] == mv[0][0] && mc[m].mv[0][1] == mv[0][1] && mc[m].mv[1][0] == mv[1][0] && mc[m].mv[1][1] == mv[1][1] &&
              mc[m].ref[0] == cur_cu->inter.mv_ref[0] && mc[m].ref[1] == cur_cu->inter.mv_ref[1]) {
            cur_cu->merged = 1;
            cur_cu->merge_idx = m;
            break;
          }
        for (int reflist = 0; reflist < 2; reflist++) {   /* each vector has its own candidate; mv_cand stays the last list's (:1424-1433) */
          kvz_inter_get_mv_cand(state, x, y, width, height, mv_cand, cur_cu, lcu, reflist);
          CU_SET_MV_CAND(cur_cu, reflist, refme_select_mv_cand(state, mv_cand, cur_cu->inter.mv[reflist][0], cur_cu->inter.mv[reflist][1]));
        }
        cost = pair_cost;
        bitcost = bits[0] + bits[1] + extra_bits;
      }
    }
  }
  if (bad) {
    if (g_gpu.failed++ == 0) fprintf(stderr, "gpu_serve_pu: %s\n", g_gpu.last_error());
    *cur_cu = saved;
    return 0;
  }
  *inter_cost = cost;
  *inter_bitcost = bitcost;
  return 1;
}

/* kvz_search_cu_inter (search_inter.c:1587-1608): the 2Nx2N PU */
static int gpu_search_serve(encoder_state_t *state, int x, int y, int depth, lcu_t *lcu, double *inter_cost, uint32_t *inter_bitcost)
{
  if (!g_gpu.on) return 0;
  gpu_flush_deblock(state);
  if (!gpu_can_serve_inter(state)) { ++g_gpu.passed_on; return 0; }
  const int width = LCU_WIDTH >> depth;
  if (!gpu_serve_pu(state, x, y, width, width, 1, 1, lcu, inter_cost, inter_bitcost)) return 0;   /* the reference's own search takes over */
  ++g_gpu.served;
  if (state->encoder_control->cfg.rdo >= 2) kvz_cu_cost_inter_rd2(state, x, y, depth, lcu, inter_cost, inter_bitcost);   /* :1600-1607 */
  return 1;
}

/* kvz_search_cu_smp (search_inter.c:1626-1718): the PUs of a SMP / AMP partition one after the other, each
 * seeing its predecessor's decision in lcu->cu (-Wl,--wrap=kvz_search_cu_smp) */
void __real_kvz_search_cu_smp(encoder_state_t * const state, int x, int y, int depth, part_mode_t part_mode, lcu_t *lcu,
                              double *inter_cost, uint32_t *inter_bitcost);

void __wrap_kvz_search_cu_smp(encoder_state_t * const state, int x, int y, int depth, part_mode_t part_mode, lcu_t *lcu,
                              double *inter_cost, uint32_t *inter_bitcost)
{
  if (g_gpu.on) gpu_flush_deblock(state);
  if (!g_gpu.on || !gpu_can_serve_inter(state)) {
    if (g_gpu.on) ++g_gpu.passed_on;
    __real_kvz_search_cu_smp(state, x, y, depth, part_mode, lcu, inter_cost, inter_bitcost);
    return;
  }
  const int num_pu = kvz_part_mode_num_parts[part_mode];
  const int width = LCU_WIDTH >> depth;
  const int x_local = SUB_SCU(x), y_local = SUB_SCU(y);
  /* a failed GPU call hands the whole CU back to the reference, from the state it started with */
  cu_info_t keep[LCU_T_CU_WIDTH * LCU_T_CU_WIDTH + 1];
  memcpy(keep, lcu->cu, sizeof(keep));
  *inter_cost = 0;
  *inter_bitcost = 0;
  for (int i = 0; i < num_pu; ++i) {
    const int x_pu = PU_GET_X(part_mode, width, x_local, i), y_pu = PU_GET_Y(part_mode, width, y_local, i);
    const int width_pu = PU_GET_W(part_mode, width, i), height_pu = PU_GET_H(part_mode, width, i);
    cu_info_t *cur_pu = LCU_GET_CU_AT_PX(lcu, x_pu, y_pu);
    cur_pu->type = CU_INTER;
    cur_pu->part_size = part_mode;
    cur_pu->depth = depth;
    cur_pu->qp = state->qp;
    double cost = MAX_INT;
    uint32_t bitcost = MAX_INT;
    /* search_pu_inter :1463-1475: the PU in picture coordinates, the merge neighbour barred for a second PU */
    const int xp = PU_GET_X(part_mode, width, x, i), yp = PU_GET_Y(part_mode, width, y, i);
    const int merge_a1 = i == 0 || width_pu >= height_pu, merge_b1 = i == 0 || width_pu <= height_pu;
    if (!gpu_serve_pu(state, xp, yp, width_pu, height_pu, merge_a1, merge_b1, lcu, &cost, &bitcost)) {
      memcpy(lcu->cu, keep, sizeof(keep));
      __real_kvz_search_cu_smp(state, x, y, depth, part_mode, lcu, inter_cost, inter_bitcost);
      return;
    }
    ++g_gpu.served;
    if (cost >= MAX_INT) {                               /* no vector found */
      *inter_cost = MAX_INT;
      *inter_bitcost = MAX_INT;
      return;
    }
    *inter_cost += cost;
    *inter_bitcost += bitcost;
    for (int yy = y_pu; yy < y_pu + height_pu; yy += SCU_WIDTH)
      for (int xx = x_pu; xx < x_pu + width_pu; xx += SCU_WIDTH) {
        cu_info_t *scu = LCU_GET_CU_AT_PX(lcu, xx, yy);
        scu->type = CU_INTER;
        scu->inter = cur_pu->inter;
      }
  }
  if (state->encoder_control->cfg.rdo >= 2) kvz_cu_cost_inter_rd2(state, x, y, depth, lcu, inter_cost, inter_bitcost);   /* :1693-1700 */
  /* the partition mode's own bits (:1702-1716) */
  int smp_extra_bits = 1;
  if (state->encoder_control->cfg.amp_enable) {
    smp_extra_bits += 1;
    if (part_mode != SIZE_2NxN && part_mode != SIZE_Nx2N) smp_extra_bits += 1;
  }
  smp_extra_bits += 6;
  *inter_cost += (state->encoder_control->cfg.rdo >= 2 ? state->lambda : state->lambda_sqrt) * smp_extra_bits;
  *inter_bitcost += smp_extra_bits;
}

/* ------------------------------------------------------------------------
 * The encoder's intra mode searches served the same way (tests only; -Wl,--wrap=kvz_search_cu_intra).  For the searches
 * that start with the rough search (rd < 3, depth > 0: search_intra.c:840-883) the wrapper
 *   1. lays what kvz_intra_build_reference would read -- lcu->rec.y with lcu->top_ref.y / left_ref.y around it -- out as
 *      a small picture on the device (the LCU at (64, 64), or at 0 where the real picture ends, so that every
 *      availability test sees the distances to the picture edges it would see in the real one),
 *   2. runs kvz_hip_intra_build_reference_batch and kvz_hip_intra_rough_batch back to back: 35 SATD (and SAD) costs,
 *   3. walks that table in the order of search_intra_rough (search_intra.c:404-545: coarse grid, halving refinement
 *      around the best, the three predicted modes + DC + planar, lambda * mode bits) -- the host half the header of
 *      kvz_hip_intra_rough_batch describes -- and returns mode and cost like kvz_search_cu_intra.
 * ------------------------------------------------------------------------ */
void __real_kvz_search_cu_intra(encoder_state_t * const state, const int x_px, const int y_px, const int depth, lcu_t *lcu,
                                int8_t *mode_out, double *cost_out);
int8_t refintra_refine(encoder_state_t *state, int x_px, int y_px, int depth, kvz_pixel *orig, int32_t origstride, int8_t *intra_preds,
                       int modes_to_check, int8_t number_of_modes, int8_t modes[35], double costs[35], lcu_t *lcu);

#ifndef TRSKIP_RATIO
# define TRSKIP_RATIO 1.7                      /* search_intra.c:39-41 */
#endif

static int gpu_intra_serve(encoder_state_t *state, int x_px, int y_px, int depth, lcu_t *lcu, int8_t *mode_out, double *cost_out)
{
  if (!g_gpu.on) return 0;
  gpu_flush_deblock(state);
  const encoder_control_t *ctrl = state->encoder_control;
  const kvz_config *cfg = &ctrl->cfg;
  if (depth == 0 || cfg->rdo >= 3 || ctrl->in.width != g_gpu.w || ctrl->in.height != g_gpu.h) {
    ++g_gpu.intra_passed_on;                  /* no rough search in these (search_intra.c:840): the reference's */
    return 0;
  }
  const vector2d_t lcu_px = { SUB_SCU(x_px), SUB_SCU(y_px) };
  const int log2_width = LOG2_LCU_WIDTH - depth, width = 1 << log2_width;
  cu_info_t *cur_cu = LCU_GET_CU_AT_PX(lcu, lcu_px.x, lcu_px.y);
  cu_info_t *left_cu = NULL, *above_cu = NULL;                    /* search_intra.c:814-825 */
  if (x_px >= SCU_WIDTH) left_cu = LCU_GET_CU_AT_PX(lcu, lcu_px.x - 1, lcu_px.y);
  if (y_px >= SCU_WIDTH && lcu_px.y > 0) above_cu = LCU_GET_CU_AT_PX(lcu, lcu_px.x, lcu_px.y - 1);
  int8_t intra_preds[3];
  kvz_intra_get_dir_luma_predictor(x_px, y_px, intra_preds, cur_cu, left_cu, above_cu);

  /* 1. the LCU and its borders as a picture */
  const int lcu_x0 = x_px - lcu_px.x, lcu_y0 = y_px - lcu_px.y;
  const int ox = lcu_x0 > 0 ? 64 : 0, oy = lcu_y0 > 0 ? 64 : 0;
  const int vw = ox + MIN(state->tile->frame->width - lcu_x0, 128), vh = oy + MIN(state->tile->frame->height - lcu_y0, 64);   /* the tile's picture */
  uint8_t *v = g_gpu.h_vplane;
  for (int y = 0; y < 64; ++y) memcpy(v + (size_t)(oy + y) * VPLANE_W + ox, lcu->rec.y + y * LCU_WIDTH, 64);
  if (oy > 0) {
    memcpy(v + (size_t)(oy - 1) * VPLANE_W + ox, &lcu->top_ref.y[1], LCU_REF_PX_WIDTH);
    if (ox > 0) v[(size_t)(oy - 1) * VPLANE_W + ox - 1] = lcu->left_ref.y[0];
  }
  if (ox > 0) for (int y = 0; y < 64; ++y) v[(size_t)(oy + y) * VPLANE_W + ox - 1] = lcu->left_ref.y[1 + y];
  const int r0 = oy > 0 ? oy - 1 : 0;
  int bad = g_gpu.h2d(g_gpu.d_vplane + (size_t)r0 * VPLANE_W, v + (size_t)r0 * VPLANE_W, (size_t)(oy + 64 - r0) * VPLANE_W, NULL);
  uint8_t orig_block[32 * 32];
  for (int y = 0; y < width; ++y) memcpy(orig_block + y * width, &lcu->ref.y[lcu_px.x + (lcu_px.y + y) * LCU_WIDTH], (size_t)width);
  bad |= g_gpu.h2d(g_gpu.d_orig, orig_block, (size_t)width * width, NULL);
  const kvz_hip_intra_pos pos = { ox + lcu_px.x, oy + lcu_px.y };
  bad |= g_gpu.h2d(g_gpu.d_pos, &pos, sizeof(pos), NULL);
  /* 2. references and the 35-mode cost table */
  const bool filter_boundary = !(cfg->lossless && cfg->implicit_rdpcm);
  uint32_t table[70];
  bad |= g_gpu.build_ref(log2_width, 0, g_gpu.d_vplane, VPLANE_W, vw, vh, g_gpu.d_pos, 1, g_gpu.d_iref, NULL);
  bad |= g_gpu.rough(log2_width, KVZ_HIP_INTRA_LUMA | (filter_boundary ? KVZ_HIP_INTRA_FILTER_BOUNDARY : 0), g_gpu.d_iref, g_gpu.d_orig, 1,
                     g_gpu.d_icost, g_gpu.d_icost + 35, NULL);
  bad |= g_gpu.d2h(table, g_gpu.d_icost, sizeof(table), NULL);
  if (bad) {
    if (g_gpu.failed++ == 0) fprintf(stderr, "gpu_intra_serve: %s\n", g_gpu.last_error());
    return 0;
  }
  /* get_cost / get_cost_dual (search_intra.c:99-172) from the table */
  const bool trskip = TRSKIP_RATIO != 0 && width == 4 && cfg->trskip_enable;
  double trskip_bits = 0;
  if (trskip) {
    const cabac_ctx_t *ctx = &state->cabac.ctx.transform_skip_model_luma;
    trskip_bits = CTX_ENTROPY_FBITS(ctx, 1) - CTX_ENTROPY_FBITS(ctx, 0);
    if (ctrl->chroma_format != KVZ_CSP_400) {
      ctx = &state->cabac.ctx.transform_skip_model_chroma;
      trskip_bits += 2.0 * (CTX_ENTROPY_FBITS(ctx, 1) - CTX_ENTROPY_FBITS(ctx, 0));
    }
  }
#define MODE_COST(m, out) do {                                                                      \
    double c__ = (double)table[(m)];                                                                \
    if (trskip) {                                                                                   \
      const double s__ = TRSKIP_RATIO * (double)table[35 + (m)] + state->lambda_sqrt * trskip_bits; \
      if (s__ < c__) c__ = s__;                                                                     \
    }                                                                                               \
    (out) = c__;                                                                                    \
  } while (0)
  /* 3. search_intra_rough's walk (search_intra.c:430-540) */
  int8_t modes[35];
  double costs[35];
  int8_t n = 0;
  unsigned min_cost = UINT_MAX, max_cost = 0;
  int offset;
  if (cfg->full_intra_search) offset = 1;
  else { static const int8_t offsets[4] = { 2, 4, 8, 8 }; offset = offsets[log2_width - 2]; }
  for (int mode = 2; mode <= 34; mode += 2 * offset)
    for (int i = 0; i < 2; ++i)
      if (mode + i * offset <= 34) {
        MODE_COST(mode + i * offset, costs[n]);
        modes[n] = (int8_t)(mode + i * offset);
        min_cost = MIN(min_cost, costs[n]);
        max_cost = MAX(max_cost, costs[n]);
        ++n;
      }
  int best_i = 0;
  for (int i = 1; i < n; ++i) if (costs[i] < costs[best_i]) best_i = i;
  int8_t best_mode = modes[best_i];
  double best_cost = min_cost;
  if (min_cost != max_cost) {
    while (offset > 1) {
      offset >>= 1;
      const int8_t test_modes[2] = { (int8_t)(best_mode - offset), (int8_t)(best_mode + offset) };
      for (int i = 0; i < 2; ++i)
        if (test_modes[i] >= 2 && test_modes[i] <= 34) {
          MODE_COST(test_modes[i], costs[n]);
          modes[n] = test_modes[i];
          if (costs[n] < best_cost) { best_cost = costs[n]; best_mode = modes[n]; }
          ++n;
        }
    }
  }
  const int8_t add_modes[5] = { intra_preds[0], intra_preds[1], intra_preds[2], 0, 1 };
  for (int k = 0; k < 5; ++k) {
    bool has = false;
    for (int i = 0; i < n; ++i) if (modes[i] == add_modes[k]) { has = true; break; }
    if (!has) { MODE_COST(add_modes[k], costs[n]); modes[n] = add_modes[k]; ++n; }
  }
  const int lambda_cost = (int)(state->lambda_sqrt + 0.5);
  for (int i = 0; i < n; ++i) costs[i] += lambda_cost * kvz_luma_mode_bits(state, modes[i], intra_preds);
#undef MODE_COST
  if (getenv("KVZ_GPU_SERVE_CHECK")) {                           /* debugging aid: the table against the reference's own functions */
    static int shown = 0;
    kvz_intra_references refs;
    const vector2d_t luma_px = { x_px, y_px }, pic_px = { state->tile->frame->width, state->tile->frame->height };
    kvz_intra_build_reference(log2_width, COLOR_Y, &luma_px, &pic_px, lcu, &refs);
    kvz_hip_intra_ref got;
    g_gpu.d2h(&got, g_gpu.d_iref, sizeof(got), NULL);
    const int refs_equal = !memcmp(got.left, refs.ref.left, 2 * width + 1) && !memcmp(got.top, refs.ref.top, 2 * width + 1);
    kvz_pixel pred[32 * 32 + 64];
    kvz_pixel *pp = (kvz_pixel *)(((uintptr_t)pred + 31) & ~(uintptr_t)31);
    for (int m = 0; m < 35 && shown < 12; ++m) {
      kvz_intra_predict(&refs, log2_width, m, COLOR_Y, pp, filter_boundary);
      const unsigned want = kvz_pixels_get_satd_func(width)(pp, orig_block);
      if (want != table[m] || !refs_equal) {
        fprintf(stderr, "intra table mismatch at (%d, %d) log2 %d mode %d: table %u, reference satd %u, references %s, lambda_cost %d\n",
                x_px, y_px, log2_width, m, table[m], want, refs_equal ? "equal" : "DIFFER", lambda_cost);
        ++shown;
        break;
      }
    }
  }
  kvz_lcu_set_trdepth(lcu, x_px, y_px, depth, depth);            /* search_intra.c:856 */
  if (cfg->rdo >= 2) {
    /* rd 2: the best two (4x4: three) modes of the rough search re-scored by full reconstruction -- the reference's own
     * search_intra_rdo (search_intra.c:857-878), which goes through the strategy table */
    const int to_search = width == 4 ? 3 : 2;
    n = refintra_refine(state, x_px, y_px, depth, &lcu->ref.y[lcu_px.x + lcu_px.y * LCU_WIDTH], LCU_WIDTH, intra_preds, MIN(n, to_search), n,
                        modes, costs, lcu);
  }
  best_i = 0;
  for (int i = 1; i < n; ++i) if (costs[i] < costs[best_i]) best_i = i;
  *mode_out = modes[best_i];
  *cost_out = costs[best_i];
  ++g_gpu.intra_served;
  return 1;
}

void __wrap_kvz_search_cu_intra(encoder_state_t * const state, const int x_px, const int y_px, const int depth, lcu_t *lcu,
                                int8_t *mode_out, double *cost_out)
{
  if (gpu_intra_serve(state, x_px, y_px, depth, lcu, mode_out, cost_out)) {
    if (getenv("KVZ_GPU_SERVE_CHECK")) {                 /* debugging aid: the reference's answer next to the served one */
      int8_t m = -1; double c = -1;
      static int shown = 0;
      __real_kvz_search_cu_intra(state, x_px, y_px, depth, lcu, &m, &c);
      if ((m != *mode_out || c != *cost_out) && shown++ < 10)
        fprintf(stderr, "intra serve mismatch at (%d, %d) depth %d poc %d: served mode %d cost %.3f, reference mode %d cost %.3f\n",
                x_px, y_px, depth, state->frame->poc, *mode_out, *cost_out, m, c);
    }
    return;
  }
  __real_kvz_search_cu_intra(state, x_px, y_px, depth, lcu, mode_out, cost_out);
}


/* ------------------------------------------------------------------------
 * The deblocking filter of a whole picture in one call (tests only; -Wl,--wrap=kvz_filter_deblock_lcu).  With
 * ref_gpu_serve_deblock(1) the per-LCU calls of the encoder (encoderstate.c:635-637) do nothing but remember the
 * picture; when the encoder first touches the NEXT picture, the remembered one -- complete, unfiltered, with its final CU
 * array -- is filtered by kvz_hip_deblock_frame in place.  Nothing reads a picture's filtered pixels before that when SAO is
 * off (the intra borders are the unfiltered copies of hor_buf / ver_buf), so the encode must produce the untouched
 * encoder's bitstream: every later picture predicts from the pixels the GPU filtered.
 * ------------------------------------------------------------------------ */
void __real_kvz_filter_deblock_lcu(encoder_state_t * const state, int x_px, int y_px);

static struct {
  int on;
  int (*deblock)(kvz_hip_pixel *, uint32_t, kvz_hip_pixel *, kvz_hip_pixel *, uint32_t, int, int, const kvz_hip_cu_info *,
                 const kvz_hip_deblock_params *, kvz_hip_stream);
  kvz_picture *pic;                            /* the picture waiting to be filtered (a reference is held) */
  cu_array_t *cua;
  kvz_hip_deblock_params prm;
  long frames, lcus_skipped;
} g_dbk;

int ref_gpu_serve_deblock(int on)
{
  memset(&g_dbk, 0, sizeof(g_dbk));
  if (!on) return 0;
  if (!g_gpu.on) return -1;
  *(void **)&g_dbk.deblock = dlsym(g_gpu.lib, "kvz_hip_deblock_frame");
  if (!g_dbk.deblock) return -1;
  g_dbk.on = 1;
  return 0;
}

/* out[0..1] = pictures filtered by kvz_hip_deblock_frame, per-LCU filter calls of the encoder that were skipped for them */
void ref_gpu_serve_deblock_end(long *out)
{
  if (g_dbk.on) gpu_flush_deblock(NULL);
  if (out) { out[0] = g_dbk.frames; out[1] = g_dbk.lcus_skipped; }
  memset(&g_dbk, 0, sizeof(g_dbk));
}

static void gpu_flush_deblock(const encoder_state_t *state)
{
  if (!g_dbk.on || !g_dbk.pic) return;
  if (state && state->tile->frame->rec == g_dbk.pic) return;          /* still the picture being coded */
  kvz_picture *pic = g_dbk.pic;
  cu_array_t *cua = g_dbk.cua;
  const int w = pic->width, h = pic->height, sy = pic->stride, sc = pic->stride / 2;
  const int has_chroma = g_dbk.prm.chroma;
  const int cstride = (w + 3) / 4, crows = (h + 3) / 4;
  kvz_hip_cu_info *map = calloc((size_t)cstride * crows, sizeof(*map));
  for (int y = 0; y < crows; ++y)
    for (int x = 0; x < cstride; ++x) {
      const cu_info_t *cu = kvz_cu_array_at_const(cua, x * 4, y * 4);
      kvz_hip_cu_info *o = &map[y * cstride + x];
      cu_to_hip(cu, o);
      o->cbf_y = cbf_is_set(cu->cbf, cu->tr_depth, COLOR_Y);
    }
  uint8_t *d_y = g_gpu.dmalloc((size_t)sy * h), *d_u = NULL, *d_v = NULL;
  kvz_hip_cu_info *d_map = g_gpu.dmalloc((size_t)cstride * crows * sizeof(*map));
  int bad = !d_y || !d_map;
  if (has_chroma) { d_u = g_gpu.dmalloc((size_t)sc * (h / 2)); d_v = g_gpu.dmalloc((size_t)sc * (h / 2)); bad |= !d_u || !d_v; }
  if (!bad) {
    bad |= g_gpu.h2d(d_y, pic->y, (size_t)sy * h, NULL);
    if (has_chroma) { bad |= g_gpu.h2d(d_u, pic->u, (size_t)sc * (h / 2), NULL); bad |= g_gpu.h2d(d_v, pic->v, (size_t)sc * (h / 2), NULL); }
    bad |= g_gpu.h2d(d_map, map, (size_t)cstride * crows * sizeof(*map), NULL);
    bad |= g_dbk.deblock(d_y, (uint32_t)sy, d_u, d_v, (uint32_t)sc, w, h, d_map, &g_dbk.prm, NULL);
    bad |= g_gpu.d2h(pic->y, d_y, (size_t)sy * h, NULL);
    if (has_chroma) { bad |= g_gpu.d2h(pic->u, d_u, (size_t)sc * (h / 2), NULL); bad |= g_gpu.d2h(pic->v, d_v, (size_t)sc * (h / 2), NULL); }
  }
  if (bad) { fprintf(stderr, "gpu_flush_deblock: %s\n", g_gpu.last_error()); ++g_gpu.failed; }
  else ++g_dbk.frames;
  g_gpu.dfree(d_y); g_gpu.dfree(d_u); g_gpu.dfree(d_v); g_gpu.dfree(d_map);
  free(map);
  kvz_image_free(pic);
  kvz_cu_array_free(&cua);
  g_dbk.pic = NULL; g_dbk.cua = NULL;
}

void __wrap_kvz_filter_deblock_lcu(encoder_state_t * const state, int x_px, int y_px)
{
  const encoder_control_t *ctrl = state->encoder_control;
  if (!g_dbk.on || ctrl->cfg.sao_type || ctrl->cfg.lossless || ctrl->cfg.tiles_width_count * ctrl->cfg.tiles_height_count > 1) {
    __real_kvz_filter_deblock_lcu(state, x_px, y_px);
    return;
  }
  kvz_picture *rec = state->tile->frame->rec;
  if (g_dbk.pic != rec) {
    gpu_flush_deblock(state);                            /* an earlier picture nobody searched after */
    g_dbk.pic = kvz_image_copy_ref(rec);
    g_dbk.cua = kvz_cu_array_copy_ref(state->tile->frame->cu_array);
    kvz_hip_deblock_params *p = &g_dbk.prm;
    memset(p, 0, sizeof(*p));
    p->beta_offset_div2 = ctrl->cfg.deblock_beta; p->tc_offset_div2 = ctrl->cfg.deblock_tc;
    p->qp = state->qp; p->frame_qp = state->frame->QP;
    p->per_cu_qp = ctrl->max_qp_delta_depth >= 0;
    p->slice_is_b = state->frame->slicetype == KVZ_SLICE_B;
    p->chroma = ctrl->chroma_format != KVZ_CSP_400;
    memcpy(p->ref_LX, state->frame->ref_LX, sizeof(p->ref_LX));
  }
  ++g_dbk.lcus_skipped;
}

/* the decoded picture hash SEI (encoder_state-bitstream.c:905-930) is taken from the finished picture: a pending filter
 * pass has to run before it (with --hash=none the pass waits for the next picture's first search) */
void __real_kvz_image_checksum(const kvz_picture *im, unsigned char checksum_out[][SEI_HASH_MAX_LENGTH], const uint8_t bitdepth);
void __real_kvz_image_md5(const kvz_picture *im, unsigned char checksum_out[][SEI_HASH_MAX_LENGTH], const uint8_t bitdepth);
void __wrap_kvz_image_checksum(const kvz_picture *im, unsigned char checksum_out[][SEI_HASH_MAX_LENGTH], const uint8_t bitdepth)
{
  if (g_dbk.on && g_dbk.pic && g_dbk.pic->y == im->y) gpu_flush_deblock(NULL);      /* the tile's picture is a view of the frame's */
  __real_kvz_image_checksum(im, checksum_out, bitdepth);
}
void __wrap_kvz_image_md5(const kvz_picture *im, unsigned char checksum_out[][SEI_HASH_MAX_LENGTH], const uint8_t bitdepth)
{
  if (g_dbk.on && g_dbk.pic && g_dbk.pic->y == im->y) gpu_flush_deblock(NULL);
  __real_kvz_image_md5(im, checksum_out, bitdepth);
}

/* the harness's own copies of the record layouts against the header's */
_Static_assert(sizeof(flat_cu_t) == sizeof(kvz_hip_cu_info) && sizeof(flat_inter_params_t) == sizeof(kvz_hip_inter_params) &&
               sizeof(flat_merge_t) == sizeof(kvz_hip_merge_cand) && sizeof(rec_pu_t) == sizeof(kvz_hip_me_pu) &&
               sizeof(rec_params_t) == sizeof(kvz_hip_me_params) && sizeof(rec_result_t) == sizeof(kvz_hip_me_result),
               "record layouts of include/kvz_hip.h");
