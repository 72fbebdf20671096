/*
 * kvz_oracle.c -- CPU restatement of Kvazaar's `generic` block kernels.
 *
 * TEST INFRASTRUCTURE ONLY (see kvz_oracle.h).  Plain C99, scalar, one thread.
 * All paths cited are relative to /root/reference/src.
 */
#include "kvz_oracle.h"

#include <stdlib.h>
#include <string.h>

#define ORC_CLIP(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))

/* ------------------------------------------------------------------ */
/* picture group                                                      */
/* ------------------------------------------------------------------ */

/* picture-generic.c:30-48.  The argument is an int16: callers that pass a
 * wider value get it truncated first.  Any bit outside 0..255 set => the
 * result is the low byte of (-v >> 15): v < 0 gives 0, v > 255 gives -1 i.e.
 * 255.  v == -32768 negates to +32768 in int arithmetic, >> 15 == 1: byte 1. */
orc_pixel orc_fast_clip_16bit_to_pixel(int16_t value)
{
  if (value & ~255) {
    int16_t t = (int16_t)((-(int)value) >> 15);
    return (orc_pixel)t;
  }
  return (orc_pixel)value;
}

/* picture-generic.c:52-70.  -INT32_MIN is evaluated with wrap-around (the
 * reference relies on two's complement): result byte 0xFF. */
orc_pixel orc_fast_clip_32bit_to_pixel(int32_t value)
{
  if (value & ~255) {
    int32_t neg = (int32_t)(0u - (uint32_t)value);
    return (orc_pixel)(neg >> 31);
  }
  return (orc_pixel)value;
}

/* picture-generic.c:86-99 */
unsigned orc_reg_sad(const orc_pixel *d1, const orc_pixel *d2, int w, int h,
                     unsigned stride1, unsigned stride2)
{
  unsigned sad = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      sad += (unsigned)abs((int)d1[(size_t)y * stride1 + x] - (int)d2[(size_t)y * stride2 + x]);
  return sad;
}

/* picture-generic.c:460-486 (SAD_NXN); >> (bitdepth-8) == >> 0 */
unsigned orc_sad_nxn(int n, const orc_pixel *b1, const orc_pixel *b2)
{
  unsigned sum = 0;
  for (int i = 0; i < n * n; ++i) sum += (unsigned)abs((int)b1[i] - (int)b2[i]);
  return sum;
}

/* Sum of |H * D * H^T| for an order-N Hadamard matrix.  The reference's
 * butterflies (picture-generic.c:105-184, :240-328) compute the same set of
 * coefficients in a different (sequency) order and sign; the absolute sum is
 * invariant to both. */
static int32_t hadamard_abs_sum(const int32_t *d, int n)
{
  int32_t t[64], sum = 0;
  /* rows: t = D * H^T  (H[k][i] = (-1)^popcount(k & i)) */
  for (int y = 0; y < n; ++y)
    for (int k = 0; k < n; ++k) {
      int32_t acc = 0;
      for (int x = 0; x < n; ++x)
        acc += (__builtin_popcount(k & x) & 1) ? -d[y * n + x] : d[y * n + x];
      t[y * n + k] = acc;
    }
  /* columns */
  for (int k = 0; k < n; ++k)
    for (int x = 0; x < n; ++x) {
      int32_t acc = 0;
      for (int y = 0; y < n; ++y)
        acc += (__builtin_popcount(k & y) & 1) ? -t[y * n + x] : t[y * n + x];
      sum += abs(acc);
    }
  return sum;
}

/* picture-generic.c:201-213 + :105-184: (sum + 1) >> 1 */
unsigned orc_satd_4x4_subblock(const orc_pixel *b1, int s1, const orc_pixel *b2, int s2)
{
  int32_t d[16];
  for (int y = 0; y < 4; ++y)
    for (int x = 0; x < 4; ++x) d[y * 4 + x] = (int)b1[y * s1 + x] - (int)b2[y * s2 + x];
  return (unsigned)((hadamard_abs_sum(d, 4) + 1) >> 1);
}

/* picture-generic.c:189-196 */
unsigned orc_satd_4x4(const orc_pixel *b1, const orc_pixel *b2)
{
  return orc_satd_4x4_subblock(b1, 4, b2, 4);
}

/* picture-generic.c:240-328: (sum + 2) >> 2 per 8x8 */
unsigned orc_satd_8x8_subblock(const orc_pixel *b1, int s1, const orc_pixel *b2, int s2)
{
  int32_t d[64];
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) d[y * 8 + x] = (int)b1[y * s1 + x] - (int)b2[y * s2 + x];
  return (unsigned)((hadamard_abs_sum(d, 8) + 2) >> 2);
}

/* strategies-picture.h:40-56 (SATD_NxN) and picture-generic.c:189 for n==4 */
unsigned orc_satd_nxn(int n, const orc_pixel *b1, const orc_pixel *b2)
{
  if (n == 4) return orc_satd_4x4(b1, b2);
  unsigned sum = 0;
  for (int y = 0; y < n; y += 8)
    for (int x = 0; x < n; x += 8)
      sum += orc_satd_8x8_subblock(&b1[y * n + x], n, &b2[y * n + x], n);
  return sum;
}

/* strategies-picture.h:62-100 (SATD_ANY_SIZE) */
unsigned orc_satd_any_size(int w, int h, const orc_pixel *b1, int s1,
                           const orc_pixel *b2, int s2)
{
  unsigned sum = 0;
  if (w % 8 != 0) {               /* first 4-px column in 4x4s */
    for (int y = 0; y < h; y += 4)
      sum += orc_satd_4x4_subblock(&b1[y * s1], s1, &b2[y * s2], s2);
    b1 += 4; b2 += 4; w -= 4;
  }
  if (h % 8 != 0) {               /* first 4-px row of what is left */
    for (int x = 0; x < w; x += 4)
      sum += orc_satd_4x4_subblock(&b1[x], s1, &b2[x], s2);
    b1 += 4 * s1; b2 += 4 * s2; h -= 4;
  }
  for (int y = 0; y < h; y += 8)
    for (int x = 0; x < w; x += 8)
      sum += orc_satd_8x8_subblock(&b1[y * s1 + x], s1, &b2[y * s2 + x], s2);
  return sum;
}

/* picture-generic.c:497-519 */
void orc_sad_nxn_dual(int n, const orc_pixel *preds, size_t pred_stride,
                      const orc_pixel *orig, unsigned costs[2])
{
  costs[0] = orc_sad_nxn(n, preds, orig);
  costs[1] = orc_sad_nxn(n, preds + pred_stride, orig);
}

/* picture-generic.c:357-390 */
void orc_satd_nxn_dual(int n, const orc_pixel *preds, size_t pred_stride,
                       const orc_pixel *orig, unsigned costs[2])
{
  costs[0] = orc_satd_nxn(n, preds, orig);
  costs[1] = orc_satd_nxn(n, preds + pred_stride, orig);
}

/* picture-generic.c:392-456.  The 4x4 stages of the reference write a scratch
 * `sums[]` that is never accumulated, and the 8x8 loop re-bases its pointers to
 * the block origin, so their only effect is `width -= 4` / `height -= 4`.
 * What is summed is therefore the 8x8 grid over [0,w') x [0,h') from the
 * ORIGIN (w' = w-4 if w%8 else w), each 8x8 read in full. */
void orc_satd_any_size_quad(int w, int h, const orc_pixel *const preds[4], int stride,
                            const orc_pixel *orig, int orig_stride, unsigned costs[4])
{
  if (w % 8 != 0) w -= 4;
  if (h % 8 != 0) h -= 4;
  for (int k = 0; k < 4; ++k) {
    unsigned sum = 0;
    for (int y = 0; y < h; y += 8)
      for (int x = 0; x < w; x += 8)
        sum += orc_satd_8x8_subblock(&orig[y * orig_stride + x], orig_stride,
                                     &preds[k][y * stride + x], stride);
    costs[k] = sum;
  }
}

/* picture-generic.c:521-536 */
unsigned orc_pixels_calc_ssd(const orc_pixel *ref, const orc_pixel *rec,
                             int ref_stride, int rec_stride, int width)
{
  int ssd = 0;
  for (int y = 0; y < width; ++y)
    for (int x = 0; x < width; ++x) {
      int d = (int)ref[x + y * ref_stride] - (int)rec[x + y * rec_stride];
      ssd += d * d;
    }
  return (unsigned)ssd;
}

/* picture-generic.c:538-588 for one plane: samples are held in int16,
 * shift = 15 - 8 = 7, offset = 64. */
void orc_bipred_blend_plane(int w, int h,
                            int hi_prec0, const int16_t *hp0, const orc_pixel *px0, int stride0,
                            int hi_prec1, const int16_t *hp1, const orc_pixel *px1, int stride1,
                            orc_pixel *dst, int dst_stride)
{
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int16_t s0 = hi_prec0 ? hp0[y * stride0 + x] : (int16_t)(px0[y * stride0 + x] << 6);
      int16_t s1 = hi_prec1 ? hp1[y * stride1 + x] : (int16_t)(px1[y * stride1 + x] << 6);
      dst[y * dst_stride + x] = orc_fast_clip_32bit_to_pixel((s0 + s1 + 64) >> 7);
    }
}

/* image.c:455-486 with :320-444: the branchy cor/ver/hor_sad decomposition is
 * exactly a SAD against the edge-replicated reference (coordinates clamped to
 * the frame), which is what is restated here. */
unsigned orc_image_calc_sad(const orc_pixel *pic, int pic_stride,
                            const orc_pixel *ref, int ref_stride, int ref_w, int ref_h,
                            int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh)
{
  unsigned sad = 0;
  for (int y = 0; y < bh; ++y) {
    int ry = ORC_CLIP(0, ref_h - 1, ref_y + y);
    for (int x = 0; x < bw; ++x) {
      int rx = ORC_CLIP(0, ref_w - 1, ref_x + x);
      sad += (unsigned)abs((int)pic[(pic_y + y) * pic_stride + pic_x + x] - (int)ref[ry * ref_stride + rx]);
    }
  }
  return sad;
}

/* image.c:488-545: inside => satd_any_size on the frame; else on an
 * edge-replicated copy (kvz_get_extended_block with filter_size 0). */
unsigned orc_image_calc_satd(const orc_pixel *pic, int pic_stride,
                             const orc_pixel *ref, int ref_stride, int ref_w, int ref_h,
                             int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh)
{
  orc_pixel *tmp = (orc_pixel *)malloc((size_t)bw * bh);
  for (int y = 0; y < bh; ++y) {
    int ry = ORC_CLIP(0, ref_h - 1, ref_y + y);
    for (int x = 0; x < bw; ++x) {
      int rx = ORC_CLIP(0, ref_w - 1, ref_x + x);
      tmp[y * bw + x] = ref[ry * ref_stride + rx];
    }
  }
  unsigned r = orc_satd_any_size(bw, bh, &pic[pic_y * pic_stride + pic_x], pic_stride, tmp, bw);
  free(tmp);
  return r;
}

/* ------------------------------------------------------------------ */
/* dct group                                                          */
/* ------------------------------------------------------------------ */

/* HEVC core transform (dct-generic.c:34-108): first column of the 32-point
 * matrix; entry (k,n) = +-c[m] with m folded from k*(2n+1) mod 128 by the
 * cosine symmetries.  The 16/8/4-point matrices are the even rows of the next
 * larger one restricted to the first half of the columns. */
static const int16_t c32[32] = {
  64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
  64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };

static int16_t g_mat[4][32 * 32];   /* [log2n - 2][k*n + i] */
static int g_mat_ready = 0;

static void build_matrices(void)
{
  int16_t m32[32][32];
  for (int k = 0; k < 32; ++k)
    for (int n = 0; n < 32; ++n) {
      int m = (k * (2 * n + 1)) % 128, sign = 1;
      if (m > 64) m = 128 - m;            /* cos(2pi - a) = cos a  */
      if (m > 32) { m = 64 - m; sign = -1; }  /* cos(pi - a) = -cos a */
      m32[k][n] = (int16_t)((m == 32) ? 0 : sign * c32[m]);
    }
  for (int l = 0; l < 4; ++l) {
    int n = 4 << l, step = 32 / n;
    for (int k = 0; k < n; ++k)
      for (int i = 0; i < n; ++i) g_mat[l][k * n + i] = m32[k * step][i];
  }
  g_mat_ready = 1;
}

/* HEVC 4x4 DST-VII (dct-generic.c:26-32) */
static const int16_t dst4[16] = {
  29, 55, 74, 84,
  74, 74, 0, -74,
  84, -29, -74, 55,
  55, -84, 74, -29 };

const int16_t *orc_dct_matrix(int n)
{
  if (!g_mat_ready) build_matrices();
  switch (n) { case 4: return g_mat[0]; case 8: return g_mat[1];
               case 16: return g_mat[2]; case 32: return g_mat[3]; }
  return NULL;
}
const int16_t *orc_dst4_matrix(void) { return dst4; }

/* forward pass (partial_butterfly_N_generic, dct-generic.c:243-266 etc.):
 * dst[k][j] = (short)((sum_i M[k][i] * src[j][i] + add) >> shift) -- the cast
 * truncates (wraps), it does not clip. */
static void fwd_pass(const int16_t *M, int n, const int16_t *src, int16_t *dst, int shift)
{
  int32_t add = 1 << (shift - 1);
  for (int j = 0; j < n; ++j)
    for (int k = 0; k < n; ++k) {
      int32_t acc = 0;
      for (int i = 0; i < n; ++i) acc += (int32_t)M[k * n + i] * src[j * n + i];
      dst[k * n + j] = (int16_t)((acc + add) >> shift);
    }
}

/* inverse pass (partial_butterfly_inverse_N_generic, :269-293 etc.):
 * dst[j][i] = clip16((sum_k M[k][i] * src[k][j] + add) >> shift) */
static void inv_pass(const int16_t *M, int n, const int16_t *src, int16_t *dst, int shift)
{
  int32_t add = 1 << (shift - 1);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      int32_t acc = 0;
      for (int k = 0; k < n; ++k) acc += (int32_t)M[k * n + i] * src[k * n + j];
      int32_t v = (acc + add) >> shift;
      dst[j * n + i] = (int16_t)ORC_CLIP(-32768, 32767, v);
    }
}

/* dct-generic.c:567-617.  bitdepth 8: forward shifts log2(n)-1 and log2(n)+6,
 * inverse shifts 7 and 12. */
void orc_transform(int kind, int n, const int16_t *in, int16_t *out)
{
  int16_t tmp[32 * 32];
  int log2n = (n == 4) ? 2 : (n == 8) ? 3 : (n == 16) ? 4 : 5;
  const int16_t *M = (kind == ORC_DST || kind == ORC_IDST) ? dst4 : orc_dct_matrix(n);
  if (kind == ORC_DCT || kind == ORC_DST) {
    fwd_pass(M, n, in, tmp, log2n - 1);
    fwd_pass(M, n, tmp, out, log2n + 6);
  } else {
    inv_pass(M, n, in, tmp, 7);
    inv_pass(M, n, tmp, out, 12);
  }
}

/* ------------------------------------------------------------------ */
/* quant group                                                        */
/* ------------------------------------------------------------------ */

static const uint8_t chroma_scale[58] = {           /* transform.c:44-50 */
   0, 1, 2, 3, 4, 5, 6, 7, 8, 9,10,11,12,13,14,15,16,
  17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,
  33,33,34,34,35,35,36,36,37,37,38,39,40,41,42,43,44,
  45,46,47,48,49,50,51 };
static const int16_t quant_scales[6]     = { 26214, 23302, 20560, 18396, 16384, 14564 }; /* scalinglist.c:66 */
static const int16_t inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                    /* scalinglist.c:67 */

/* transform.c:129-143 */
int32_t orc_get_scaled_qp(int type, int qp, int qp_offset)
{
  if (type == 0) return qp + qp_offset;
  int32_t q = ORC_CLIP(-qp_offset, 57, qp);
  return (q < 0) ? q + qp_offset : chroma_scale[q] + qp_offset;
}

/* kvz_g_sig_last_scan[scan_idx][log2_size - 1] (tables.c, generated by
 * tools/generate_tables.c): scan_idx 0 = up-right diagonal, 1 = horizontal,
 * 2 = vertical; blocks >= 8x8 are scanned in 4x4 coefficient groups, groups
 * and positions inside a group both following the pattern. */
static uint32_t g_scan[3][6][32 * 32];
static int g_scan_ready = 0;

static int pattern_order(int scan_idx, int n, int *xs, int *ys)
{
  int c = 0;
  if (scan_idx == 1) { for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) { xs[c] = x; ys[c] = y; ++c; } }
  else if (scan_idx == 2) { for (int x = 0; x < n; ++x) for (int y = 0; y < n; ++y) { xs[c] = x; ys[c] = y; ++c; } }
  else {
    for (int d = 0; d < 2 * n - 1; ++d)       /* anti-diagonals, bottom-left to top-right */
      for (int y = (d < n ? d : n - 1); y >= 0 && d - y < n; --y) { xs[c] = d - y; ys[c] = y; ++c; }
  }
  return c;
}

static void build_scans(void)
{
  for (int s = 0; s < 3; ++s)
    for (int l = 1; l <= 5; ++l) {
      int n = 1 << l, c = 0;
      uint32_t *out = g_scan[s][l];
      int xs[64], ys[64];
      if (n <= 4) {
        int cnt = pattern_order(s, n, xs, ys);
        for (int i = 0; i < cnt; ++i) out[c++] = (uint32_t)(ys[i] * n + xs[i]);
      } else {
        int gx[64], gy[64], g = n / 4;
        int ng = pattern_order(s, g, gx, gy);
        int np = pattern_order(s, 4, xs, ys);
        for (int i = 0; i < ng; ++i)
          for (int j = 0; j < np; ++j)
            out[c++] = (uint32_t)((gy[i] * 4 + ys[j]) * n + gx[i] * 4 + xs[j]);
      }
    }
  g_scan_ready = 1;
}

const uint32_t *orc_scan_order(int scan_idx, int log2_size)
{
  if (!g_scan_ready) build_scans();
  return g_scan[scan_idx][log2_size];
}

static int log2_of(int w) { int l = 0; while ((1 << l) < w) ++l; return l; }

/* quant-generic.c:37-163 */
void orc_quant(const orc_quant_params *p, const orc_coeff *coef, orc_coeff *q_coef,
               int w, int h, int type, int scan_idx, int block_is_intra)
{
  (void)block_is_intra;   /* only selects the scaling list, which the caller resolved */
  const int log2_tr = log2_of(w);
  const uint32_t *scan = orc_scan_order(scan_idx, log2_tr);
  const int32_t qp_scaled = orc_get_scaled_qp(type, p->qp, 0);
  const int32_t transform_shift = 15 - 8 - log2_tr;
  const int32_t q_bits = 14 + qp_scaled / 6 + transform_shift;
  const int32_t add = (p->slice_is_intra ? 171 : 85) << (q_bits - 9);
  const int32_t q_bits8 = q_bits - 8;
  const int32_t flat = quant_scales[qp_scaled % 6];
  const int n_coef = w * h;
  uint32_t ac_sum = 0;

  for (int n = 0; n < n_coef; ++n) {
    int32_t qc = (p->scaling_list && p->quant_coeff) ? p->quant_coeff[n] : flat;
    int32_t level = coef[n];
    int32_t sign = level < 0 ? -1 : 1;
    level = (int32_t)(((int64_t)abs(level) * qc + add) >> q_bits);
    ac_sum += (uint32_t)level;
    level *= sign;
    q_coef[n] = (orc_coeff)ORC_CLIP(-32768, 32767, level);
  }
  if (!p->signhide || ac_sum < 2) return;

  /* sign bit hiding, :69-162 */
  int32_t delta_u[32 * 32];
  for (int n = 0; n < n_coef; ++n) {
    int32_t qc = (p->scaling_list && p->quant_coeff) ? p->quant_coeff[n] : flat;
    int64_t prod = (int64_t)abs((int)coef[n]) * qc;
    int32_t level = (int32_t)((prod + add) >> q_bits);
    delta_u[n] = (int32_t)((prod - ((int64_t)(int32_t)((uint32_t)level << q_bits))) >> q_bits8);
  }

  int32_t last_cg = -1;
  for (int subset = (n_coef - 1) >> 4; subset >= 0; --subset) {
    const int subpos = subset << 4;
    int first_nz = 16, last_nz = -1, abssum = 0, n;
    for (n = 15; n >= 0; --n) if (q_coef[scan[n + subpos]]) { last_nz = n; break; }
    for (n = 0; n < 16; ++n)  if (q_coef[scan[n + subpos]]) { first_nz = n; break; }
    for (n = first_nz; n <= last_nz; ++n) abssum += q_coef[scan[n + subpos]];
    if (last_nz >= 0 && last_cg == -1) last_cg = 1;

    if (last_nz - first_nz >= 4) {
      int32_t signbit = q_coef[scan[subpos + first_nz]] > 0 ? 0 : 1;
      if (signbit != (abssum & 1)) {
        int32_t min_cost_inc = 0x7fffffff, min_pos = -1, cur_cost = 0x7fffffff;
        int16_t final_change = 0, cur_change = 0;
        for (n = (last_cg == 1 ? last_nz : 15); n >= 0; --n) {
          uint32_t pos = scan[n + subpos];
          if (q_coef[pos] != 0) {
            if (delta_u[pos] > 0) { cur_cost = -delta_u[pos]; cur_change = 1; }
            else if (n == first_nz && abs((int)q_coef[pos]) == 1) { cur_cost = 0x7fffffff; }
            else { cur_cost = delta_u[pos]; cur_change = -1; }
          } else if (n < first_nz && ((coef[pos] >= 0) ? 0 : 1) != signbit) {
            cur_cost = 0x7fffffff;
          } else { cur_cost = -delta_u[pos]; cur_change = 1; }
          if (cur_cost < min_cost_inc) { min_cost_inc = cur_cost; final_change = cur_change; min_pos = (int32_t)pos; }
        }
        if (q_coef[min_pos] == 32767 || q_coef[min_pos] == -32768) final_change = -1;
        if (coef[min_pos] >= 0) q_coef[min_pos] = (orc_coeff)(q_coef[min_pos] + final_change);
        else                    q_coef[min_pos] = (orc_coeff)(q_coef[min_pos] - final_change);
      }
    }
    if (last_cg == 1) last_cg = 0;
  }
}

/* quant-generic.c:279-321 */
void orc_dequant(const orc_quant_params *p, const orc_coeff *q_coef, orc_coeff *coef,
                 int w, int h, int type, int block_is_intra)
{
  (void)block_is_intra;
  const int log2_tr = log2_of(w);
  const int32_t transform_shift = 15 - 8 - log2_tr;
  const int32_t qp_scaled = orc_get_scaled_qp(type, p->qp, 0);
  int32_t shift = 20 - 14 - transform_shift;
  const int n_coef = w * h;

  if (p->scaling_list && p->dequant_coeff) {
    shift += 4;
    if (shift > qp_scaled / 6) {
      int32_t add = 1 << (shift - qp_scaled / 6 - 1);
      for (int n = 0; n < n_coef; ++n) {
        int32_t v = (q_coef[n] * p->dequant_coeff[n] + add) >> (shift - qp_scaled / 6);
        coef[n] = (orc_coeff)ORC_CLIP(-32768, 32767, v);
      }
    } else {
      for (int n = 0; n < n_coef; ++n) {
        int32_t v = ORC_CLIP(-32768, 32767, q_coef[n] * p->dequant_coeff[n]);
        v = (int32_t)((uint32_t)v << (qp_scaled / 6 - shift));
        coef[n] = (orc_coeff)ORC_CLIP(-32768, 32767, v);
      }
    }
  } else {
    int32_t scale = inv_quant_scales[qp_scaled % 6] << (qp_scaled / 6);
    int32_t add = 1 << (shift - 1);
    for (int n = 0; n < n_coef; ++n) {
      int32_t v = (int32_t)((uint32_t)((int32_t)q_coef[n] * scale) + (uint32_t)add) >> shift;
      coef[n] = (orc_coeff)ORC_CLIP(-32768, 32767, v);
    }
  }
}

/* quant-generic.c:323-330 */
uint32_t orc_coeff_abs_sum(const orc_coeff *c, size_t len)
{
  uint32_t sum = 0;
  for (size_t i = 0; i < len; ++i) sum += (uint32_t)abs((int)c[i]);
  return sum;
}

/* quant-generic.c:180-273, rdoq disabled.  transform choice: strategies-dct.c:66-85
 * (4x4 intra luma => DST).  trskip: transform.c:150-180. */
int orc_quantize_residual(const orc_quant_params *p, int cu_is_intra, int width, int color,
                          int scan_order, int use_trskip, int in_stride, int out_stride,
                          const orc_pixel *ref_in, const orc_pixel *pred_in,
                          orc_pixel *rec_out, orc_coeff *coeff_out)
{
  int16_t residual[32 * 32];
  orc_coeff coeff[32 * 32];
  const int log2_tr = log2_of(width);
  const int ts_shift = 15 - 8 - log2_tr;
  const int use_dst = (width == 4 && color == 0 && cu_is_intra);
  int has_coeffs = 0;

  for (int y = 0; y < width; ++y)
    for (int x = 0; x < width; ++x)
      residual[x + y * width] = (int16_t)((int)ref_in[x + y * in_stride] - (int)pred_in[x + y * in_stride]);

  if (use_trskip) {
    for (int i = 0; i < width * width; ++i) coeff[i] = (orc_coeff)((int)residual[i] << ts_shift);
  } else {
    orc_transform(use_dst ? ORC_DST : ORC_DCT, width, residual, coeff);
  }

  orc_quant(p, coeff, coeff_out, width, width, color == 0 ? 0 : 2, scan_order, cu_is_intra);

  for (int i = 0; i < width * width; ++i) if (coeff_out[i] != 0) { has_coeffs = 1; break; }

  if (has_coeffs) {
    orc_dequant(p, coeff_out, coeff, width, width, color == 0 ? 0 : (color == 1 ? 2 : 3), cu_is_intra);
    if (use_trskip) {
      int32_t offset = 1 << (ts_shift - 1);
      for (int i = 0; i < width * width; ++i) residual[i] = (int16_t)((coeff[i] + offset) >> ts_shift);
    } else {
      orc_transform(use_dst ? ORC_IDST : ORC_IDCT, width, coeff, residual);
    }
    for (int y = 0; y < width; ++y)
      for (int x = 0; x < width; ++x) {
        int16_t val = (int16_t)(residual[x + y * width] + pred_in[x + y * in_stride]);
        rec_out[x + y * out_stride] = (orc_pixel)ORC_CLIP(0, 255, val);
      }
  } else if (rec_out != pred_in) {
    for (int y = 0; y < width; ++y)
      for (int x = 0; x < width; ++x) rec_out[x + y * out_stride] = pred_in[x + y * in_stride];
  }
  return has_coeffs;
}

/* ------------------------------------------------------------------ */
/* ipol group                                                         */
/* ------------------------------------------------------------------ */

const int8_t orc_luma_filter[4][8] = {          /* filter.c:54-60 */
  {  0, 0,   0, 64,  0,   0, 0,  0 },
  { -1, 4, -10, 58, 17,  -5, 1,  0 },
  { -1, 4, -11, 40, 40, -11, 4, -1 },
  {  0, 1,  -5, 17, 58, -10, 4, -1 } };
const int8_t orc_chroma_filter[8][4] = {        /* filter.c:62-72 */
  {  0, 64,  0,  0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
  { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

static int32_t fir_px(const int8_t *f, int taps, const orc_pixel *d, int step)
{
  int32_t t = 0;
  for (int i = 0; i < taps; ++i) t += f[i] * d[i * step];
  return t;
}
static int32_t fir_s16(const int8_t *f, int taps, const int16_t *d, int step)
{
  int32_t t = 0;
  for (int i = 0; i < taps; ++i) t += f[i] * d[i * step];
  return t;
}

/* Two-pass separable filter shared by ipol-generic.c:122-190 and :660-728:
 * horizontal pass over h+taps-1 rows into int16, vertical pass >> 6. */
static void sample_two_pass(const int8_t *hf, const int8_t *vf, int taps,
                            const orc_pixel *src, int src_stride, int w, int h,
                            orc_pixel *dst8, int16_t *dst16, int dst_stride)
{
  const int off = taps / 2 - 1;                /* 3 luma, 1 chroma */
  int16_t *hor = (int16_t *)malloc(sizeof(int16_t) * (size_t)(h + taps - 1) * (size_t)w);
  for (int y = 0; y < h + taps - 1; ++y)
    for (int x = 0; x < w; ++x)
      hor[y * w + x] = (int16_t)fir_px(hf, taps, &src[src_stride * (y - off) + (x - off)], 1);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int32_t v = fir_s16(vf, taps, &hor[y * w + x], w) >> 6;
      if (dst16) dst16[y * dst_stride + x] = (int16_t)v;
      else       dst8[y * dst_stride + x] = orc_fast_clip_32bit_to_pixel((v + 32) >> 6);
    }
  free(hor);
}

void orc_sample_quarterpel_luma(const orc_pixel *src, int src_stride, int w, int h,
                                orc_pixel *dst, int dst_stride, const int16_t mv[2])
{
  sample_two_pass(orc_luma_filter[mv[0] & 3], orc_luma_filter[mv[1] & 3], 8,
                  src, src_stride, w, h, dst, NULL, dst_stride);
}
void orc_sample_14bit_quarterpel_luma(const orc_pixel *src, int src_stride, int w, int h,
                                      int16_t *dst, int dst_stride, const int16_t mv[2])
{
  sample_two_pass(orc_luma_filter[mv[0] & 3], orc_luma_filter[mv[1] & 3], 8,
                  src, src_stride, w, h, NULL, dst, dst_stride);
}
void orc_sample_octpel_chroma(const orc_pixel *src, int src_stride, int w, int h,
                              orc_pixel *dst, int dst_stride, const int16_t mv[2])
{
  sample_two_pass(orc_chroma_filter[mv[0] & 7], orc_chroma_filter[mv[1] & 7], 4,
                  src, src_stride, w, h, dst, NULL, dst_stride);
}
void orc_sample_14bit_octpel_chroma(const orc_pixel *src, int src_stride, int w, int h,
                                    int16_t *dst, int dst_stride, const int16_t mv[2])
{
  sample_two_pass(orc_chroma_filter[mv[0] & 7], orc_chroma_filter[mv[1] & 7], 4,
                  src, src_stride, w, h, NULL, dst, dst_stride);
}

/* (int16 sample + 32) >> 6 through the int16-argument clip */
static orc_pixel round_clip16(int16_t sample)
{
  return orc_fast_clip_16bit_to_pixel((int16_t)((sample + 32) >> 6));
}

#define HS ORC_LCU_WIDTH   /* hor_stride == dst_stride == 64 */

/* horizontal 8-tap of `fir` on rows y0..rows-1 of the (h+8)-row window:
 * hor[y][x] = fir . src[y-3][x-2 .. x+5], col[y] = fir . src[y-3][-3 .. 4] */
static void hor_plane(const int8_t *fir, const orc_pixel *src, int ss, int w, int rows, int y0,
                      int16_t *hor, int16_t *col)
{
  for (int y = y0; y < rows; ++y) {
    for (int x = 0; x < w; ++x)
      hor[y * HS + x] = (int16_t)fir_px(fir, 8, &src[ss * (y - 3) + (x - 3 + 1)], 1);
    col[y] = (int16_t)fir_px(fir, 8, &src[ss * (y - 3) + (0 - 3)], 1);
  }
}

/* One output block of the qpel steps (ipol-generic.c:480-545 and :595-657):
 * vertical filter `vf` on plane `hor` from row y+yo; when !xo the first column
 * comes from the contiguous column array and the rest is shifted one to the
 * right. */
static void qpel_block(const int8_t *vf, const int16_t *hor, const int16_t *col, int xo, int yo,
                       int w, int h, orc_pixel *out)
{
  for (int y = 0; y < h; ++y) {
    if (!xo) out[y * HS] = round_clip16((int16_t)(fir_s16(vf, 8, &col[y + yo], 1) >> 6));
    for (int x = !xo; x < w; ++x)
      out[y * HS + x] = round_clip16((int16_t)(fir_s16(vf, 8, &hor[(y + yo) * HS + x - !xo], HS) >> 6));
  }
}

void orc_filter_frac_blocks(int step, const orc_pixel *src, int ss, int w, int h,
                            orc_pixel *filtered, orc_ipol_state *st,
                            int fme_level, int hpel_off_x, int hpel_off_y)
{
  orc_pixel *f0 = filtered, *f1 = filtered + 64 * 64, *f2 = filtered + 2 * 64 * 64, *f3 = filtered + 3 * 64 * 64;
  const int8_t *fir0 = orc_luma_filter[0], *fir1 = orc_luma_filter[1];
  const int8_t *fir2 = orc_luma_filter[2], *fir3 = orc_luma_filter[3];
  const int rows = h + 8;        /* height + KVZ_EXT_PADDING_LUMA + 1 */
  int x, y;

  if (step == 0) {               /* :192-305 hpel left/right/top/bottom */
    hor_plane(fir0, src, ss, w, rows, 0, st->hor[0], st->cols[0]);
    hor_plane(fir2, src, ss, w, rows, fme_level > 1 ? 0 : 1, st->hor[1], st->cols[2]);
    for (y = 0; y < h; ++y)      /* right: horizontal only */
      for (x = 0; x < w; ++x) f1[y * HS + x] = round_clip16(st->hor[1][(y + 4) * HS + x]);
    for (y = 0; y < h; ++y) {    /* left */
      f0[y * HS] = round_clip16(st->cols[2][y + 4]);
      for (x = 1; x < w; ++x) f0[y * HS + x] = f1[y * HS + x - 1];
    }
    for (y = 0; y < h; ++y)      /* top: vertical only, on pixels */
      for (x = 0; x < w; ++x)
        f2[y * HS + x] = round_clip16((int16_t)fir_px(fir2, 8, &src[ss * (y - 3) + x + 1], ss));
    for (y = 0; y < h - 1; ++y)  /* bottom */
      for (x = 0; x < w; ++x) f3[y * HS + x] = f2[(y + 1) * HS + x];
    for (x = 0; x < w; ++x)
      f3[y * HS + x] = round_clip16((int16_t)fir_px(fir2, 8, &src[ss * (y - 3 + 1) + x + 1], ss));
  } else if (step == 1) {        /* :307-386 hpel diagonals */
    const int16_t *h1 = st->hor[1], *c2 = st->cols[2];
    for (y = 0; y < h; ++y)      /* top-right */
      for (x = 0; x < w; ++x) f1[y * HS + x] = round_clip16((int16_t)(fir_s16(fir2, 8, &h1[y * HS + x], HS) >> 6));
    for (y = 0; y < h; ++y) {    /* top-left */
      f0[y * HS] = round_clip16((int16_t)(fir_s16(fir2, 8, &c2[y], 1) >> 6));
      for (x = 1; x < w; ++x) f0[y * HS + x] = f1[y * HS + x - 1];
    }
    for (y = 0; y < h - 1; ++y)  /* bottom-right */
      for (x = 0; x < w; ++x) f3[y * HS + x] = f1[(y + 1) * HS + x];
    for (x = 0; x < w; ++x) f3[y * HS + x] = round_clip16((int16_t)(fir_s16(fir2, 8, &h1[(y + 1) * HS + x], HS) >> 6));
    for (y = 0; y < h - 1; ++y)  /* bottom-left */
      for (x = 0; x < w; ++x) f2[y * HS + x] = f0[(y + 1) * HS + x];
    for (x = 1; x < w; ++x) f2[y * HS + x] = f3[y * HS + x - 1];
    f2[y * HS] = round_clip16((int16_t)(fir_s16(fir2, 8, &c2[y + 1], 1) >> 6));
  } else {
    const int off_x_fir_l = hpel_off_x < 1 ? 0 : 1, off_x_fir_r = hpel_off_x < 0 ? 0 : 1;
    const int off_y_fir_t = hpel_off_y < 1 ? 0 : 1, off_y_fir_b = hpel_off_y < 0 ? 0 : 1;
    const int8_t *ver_fir_t = hpel_off_y != 0 ? fir1 : fir3;
    const int8_t *ver_fir_b = hpel_off_y != 0 ? fir3 : fir1;
    if (step == 2) {             /* :388-546 qpel left/right/top/bottom */
      const int8_t *hor_fir_l = hpel_off_x != 0 ? fir1 : fir3;
      const int8_t *hor_fir_r = hpel_off_x != 0 ? fir3 : fir1;
      const int8_t *ver_fir_lr = hpel_off_y != 0 ? fir2 : fir0;
      const int16_t *hor_hpel = hpel_off_x != 0 ? st->hor[1] : st->hor[0];
      const int16_t *col_hor = hpel_off_x != 0 ? st->cols[2] : st->cols[0];
      const int sample_off_y = hpel_off_y < 0 ? 0 : 1;
      const int sample_off_x = hpel_off_x > -1 ? 1 : 0;
      hor_plane(hor_fir_l, src, ss, w, rows, 0, st->hor[3], st->cols[1]);
      hor_plane(hor_fir_r, src, ss, w, rows, 0, st->hor[4], st->cols[3]);
      qpel_block(ver_fir_lr, st->hor[3], st->cols[1], off_x_fir_l, sample_off_y, w, h, f0);
      qpel_block(ver_fir_lr, st->hor[4], st->cols[3], off_x_fir_r, sample_off_y, w, h, f1);
      qpel_block(ver_fir_t, hor_hpel, col_hor, sample_off_x, off_y_fir_t, w, h, f2);
      qpel_block(ver_fir_b, hor_hpel, col_hor, sample_off_x, off_y_fir_b, w, h, f3);
    } else {                     /* :548-658 qpel diagonals */
      qpel_block(ver_fir_t, st->hor[3], st->cols[1], off_x_fir_l, off_y_fir_t, w, h, f0);
      qpel_block(ver_fir_t, st->hor[4], st->cols[3], off_x_fir_r, off_y_fir_t, w, h, f1);
      qpel_block(ver_fir_b, st->hor[3], st->cols[1], off_x_fir_l, off_y_fir_b, w, h, f2);
      qpel_block(ver_fir_b, st->hor[4], st->cols[3], off_x_fir_r, off_y_fir_b, w, h, f3);
    }
  }
}

/* ipol-generic.c:731-784 */
int orc_get_extended_block(int xpos, int ypos, int mv_x, int mv_y, int off_x, int off_y,
                           const orc_pixel *ref, int ref_w, int ref_h, int filter_size,
                           int w, int h, orc_pixel *out, long *inside_off)
{
  const int half = filter_size >> 1;
  const int min_y = ypos - half + off_y + mv_y, max_y = min_y + h + filter_size;
  const int min_x = xpos - half + off_x + mv_x, max_x = min_x + w + filter_size;
  const int oob_y = (min_y < 0) || (max_y >= ref_h);
  const int oob_x = (min_x < 0) || (max_x >= ref_w);
  if (inside_off) *inside_off = (long)min_y * ref_w + min_x;
  if (!(oob_y || oob_x)) return 0;
  const int stride = w + filter_size;
  for (int dy = 0, y = ypos - half; y < ypos + h + half; ++dy, ++y) {
    int cy = ORC_CLIP(0, ref_h - 1, y + off_y + mv_y);
    for (int dx = 0, x = xpos - half; x < xpos + w + half; ++dx, ++x) {
      int cx = ORC_CLIP(0, ref_w - 1, x + off_x + mv_x);
      out[dy * stride + dx] = ref[cy * ref_w + cx];
    }
  }
  return 1;
}

/* ---- MV cost model: search_inter.c:87-176, :235-412 ---- */
typedef struct { const orc_me_pu *pu; const orc_me_params *prm; } me_ctx;

/* fracmv_within_tile (:87-176); x, y in quarter-pel.  The reference's info->origin is relative to the tile
 * (the search runs on the tile's sub-frame), ours is a picture position: subtract the tile offset first. */
static int me_within(const me_ctx *mc, int x, int y)
{
  if (!mc) return 1;
  const orc_me_pu *pu = mc->pu;
  const orc_me_params *prm = mc->prm;
  const int is_frac_luma = x % 4 != 0 || y % 4 != 0, is_frac_chroma = x % 8 != 0 || y % 8 != 0;
  const int org_x = pu->x - prm->tile_x, org_y = pu->y - prm->tile_y;
  if (prm->wpp_owf) {                                   /* :95-139: only final pixels of the reference may be read */
    int margin = is_frac_luma ? 4 : (is_frac_chroma ? 2 : 0);
    margin += prm->ref_delay_px;
    const int lcu_x = org_x / 64, lcu_y = org_y / 64;
    const int mv_lcu_x = ((org_x + pu->width + margin) * 4 + x) / (64 << 2) - lcu_x;
    const int mv_lcu_y = ((org_y + pu->height + margin) * 4 + y) / (64 << 2) - lcu_y;
    if (mv_lcu_y > prm->max_ref_lcu_down) return 0;
    if (mv_lcu_x + mv_lcu_y > prm->max_ref_lcu_down + prm->max_ref_lcu_right) return 0;
  }
  if (prm->mv_constraint == 0) return 1;                /* :142-144 */
  int margin_q = 0;                                     /* :146-154: quarter-pel margin, only for FRAME_AND_TILE_MARGIN */
  if (prm->mv_constraint == 4) margin_q = is_frac_luma ? 4 << 2 : (is_frac_chroma ? 2 << 2 : 0);
  const int abs_x = org_x * 4 + x, abs_y = org_y * 4 + y;
  const int from_right = (prm->tile_w << 2) - (abs_x + (pu->width << 2));
  const int from_bottom = (prm->tile_h << 2) - (abs_y + (pu->height << 2));
  return abs_x >= margin_q && abs_y >= margin_q && from_right >= margin_q && from_bottom >= margin_q;
}

/* get_ep_ex_golomb_bitcost (:235-254) */
static unsigned me_golomb(unsigned symbol)
{
  unsigned bins = 0;
  symbol += 2;
  if (symbol >= 1u << 8) { bins += 16; symbol >>= 8; }
  if (symbol >= 1u << 4) { bins += 8; symbol >>= 4; }
  if (symbol >= 1u << 2) { bins += 4; symbol >>= 2; }
  if (symbol >= 1u << 1) { bins += 2; }
  return bins;
}
/* get_mvd_coding_cost (:310-323): both terms are whole bits, so the fixed-point rounding is exact */
static unsigned me_mvd_bits(int dx, int dy) { return me_golomb((unsigned)abs(dx)) + me_golomb((unsigned)abs(dy)); }

/* select_mv_cand (:326-370): returns the chosen candidate, *cost_out = the smaller cost */
static int me_select_cand(const orc_me_pu *pu, int mvx, int mvy, unsigned *cost_out)
{
  const unsigned c1 = me_mvd_bits(mvx - pu->mv_cand[0][0], mvy - pu->mv_cand[0][1]);
  const unsigned c2 = me_mvd_bits(mvx - pu->mv_cand[1][0], mvy - pu->mv_cand[1][1]);
  if (cost_out) *cost_out = c1 < c2 ? c1 : c2;
  return c2 < c1 ? 1 : 0;
}

/* ---- --mv-rdo: kvz_calc_mvd_cost_cabac (rdo.c:908-1060) and kvz_get_mvd_coding_cost_cabac (:883-903) ----
 * The reference runs its CABAC encoder in counting mode on a copy of state->cabac.  What it counts,
 * (23 - bits_left) + 8 * num_buffered_bytes (cabac.c:95-140), is the number of renormalisation shifts, which depends
 * only on `range` and on the states of the contexts used -- `low` and the byte buffering never matter.
 * Tables: ITU-T H.265 (04/2013) Table 9-46 (rangeTabLps), Table 9-47 (transIdxLps); an MPS moves to min(state + 1, 62). */
static const unsigned char cabac_range_lps[64][4] = {
  {128,176,208,240},{128,167,197,227},{128,158,187,216},{123,150,178,205},{116,142,169,195},{111,135,160,185},{105,128,152,175},{100,122,144,166},
  { 95,116,137,158},{ 90,110,130,150},{ 85,104,123,142},{ 81, 99,117,135},{ 77, 94,111,128},{ 73, 89,105,122},{ 69, 85,100,116},{ 66, 80, 95,110},
  { 62, 76, 90,104},{ 59, 72, 86, 99},{ 56, 69, 81, 94},{ 53, 65, 77, 89},{ 51, 62, 73, 85},{ 48, 59, 69, 80},{ 46, 56, 66, 76},{ 43, 53, 63, 72},
  { 41, 50, 59, 69},{ 39, 48, 56, 65},{ 37, 45, 54, 62},{ 35, 43, 51, 59},{ 33, 41, 48, 56},{ 32, 39, 46, 53},{ 30, 37, 43, 50},{ 29, 35, 41, 48},
  { 27, 33, 39, 45},{ 26, 31, 37, 43},{ 24, 30, 35, 41},{ 23, 28, 33, 39},{ 22, 27, 32, 37},{ 21, 26, 30, 35},{ 20, 24, 29, 33},{ 19, 23, 27, 31},
  { 18, 22, 26, 30},{ 17, 21, 25, 28},{ 16, 20, 23, 27},{ 15, 19, 22, 25},{ 14, 18, 21, 24},{ 14, 17, 20, 23},{ 13, 16, 19, 22},{ 12, 15, 18, 21},
  { 12, 14, 17, 20},{ 11, 14, 16, 19},{ 11, 13, 15, 18},{ 10, 12, 15, 17},{ 10, 12, 14, 16},{  9, 11, 13, 15},{  9, 11, 12, 14},{  8, 10, 12, 14},
  {  8,  9, 11, 13},{  7,  9, 11, 12},{  7,  9, 10, 12},{  7,  8, 10, 11},{  6,  8,  9, 11},{  6,  7,  9, 10},{  6,  7,  8,  9},{  2,  2,  2,  2} };
static const unsigned char cabac_trans_lps[64] = {
   0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9,11,11,12,13,13,15,15,16,16,18,18,19,19,21,21,22,22,23,24,
  24,25,26,26,27,27,28,29,29,30,30,30,31,32,32,33,33,33,34,34,35,35,35,36,36,36,37,37,37,38,38,63 };
/* uc_state = (state << 1) | mps (cabac.h:119-122) */
static unsigned cabac_next_mps(unsigned uc) { const unsigned s = uc >> 1; return ((s < 62 ? s + 1 : s) << 1) | (uc & 1); }
static unsigned cabac_next_lps(unsigned uc) { const unsigned s = uc >> 1; return ((unsigned)cabac_trans_lps[s] << 1) | ((uc & 1) ^ (s == 0)); }
/* kvz_g_auc_renorm_table[lps >> 3]: the shift that brings the smallest lps of the octet to >= 256; 6 for the first octet
 * (rangeTabLps never goes below 6 for the states a context can reach) */
static unsigned cabac_renorm(unsigned lps)
{
  const unsigned i = lps >> 3;
  unsigned n = 0;
  if (i == 0) return 6;
  while (((i << 3) << n) < 256) ++n;
  return n;
}
int orc_cabac_table(int kind, int i)
{
  switch (kind) {
    case 0: return cabac_range_lps[i >> 2][i & 3];
    case 1: return (int)cabac_next_mps((unsigned)i);
    case 2: return (int)cabac_next_lps((unsigned)i);
    default: return (int)cabac_renorm((unsigned)i << 3);
  }
}
typedef struct { unsigned range; unsigned char ctx[8]; } cabac_model;
enum { CTX_MERGE_FLAG = 0, CTX_MERGE_IDX, CTX_REF0, CTX_REF1, CTX_MVD0, CTX_MVD1, CTX_MVP };
/* kvz_cabac_encode_bin (cabac.c:90-122), bits produced */
static unsigned cabac_bin(cabac_model *m, int ctx, int bin)
{
  const unsigned uc = m->ctx[ctx], lps = cabac_range_lps[uc >> 1][(m->range >> 6) & 3];
  m->range -= lps;
  if ((unsigned)(bin ? 1 : 0) != (uc & 1)) {
    const unsigned n = cabac_renorm(lps);
    m->range = lps << n;
    m->ctx[ctx] = (unsigned char)cabac_next_lps(uc);
    return n;
  }
  m->ctx[ctx] = (unsigned char)cabac_next_mps(uc);
  if (m->range >= 256) return 0;
  m->range <<= 1;
  return 1;
}
/* kvz_cabac_write_ep_ex_golomb (cabac.c:535-570): bypass bins of symbol with parameter count */
static unsigned cabac_ex_golomb_bins(unsigned symbol, unsigned count)
{
  unsigned n = 0;
  while (symbol >= (1u << count)) { ++n; symbol -= 1u << count; ++count; }
  return n + 1 + count;
}
/* kvz_encode_mvd (encode_coding_tree.c:1156-1202) */
static unsigned cabac_mvd_bits(cabac_model *m, int hor, int ver)
{
  const unsigned ah = (unsigned)abs(hor), av = (unsigned)abs(ver);
  unsigned bits = cabac_bin(m, CTX_MVD0, hor != 0);
  bits += cabac_bin(m, CTX_MVD0, ver != 0);
  if (hor) bits += cabac_bin(m, CTX_MVD1, ah > 1);
  if (ver) bits += cabac_bin(m, CTX_MVD1, av > 1);
  if (hor) bits += (ah > 1 ? cabac_ex_golomb_bins(ah - 2, 1) : 0) + 1;      /* + sign */
  if (ver) bits += (av > 1 ? cabac_ex_golomb_bins(av - 2, 1) : 0) + 1;
  return bits;
}
static cabac_model cabac_start(const me_ctx *mc)
{
  const orc_me_cabac *c = &mc->prm->cabac[mc->pu->reserved];
  cabac_model m;
  m.range = c->range;
  memcpy(m.ctx, c->ctx, 8);
  return m;
}
/* kvz_get_mvd_coding_cost_cabac (rdo.c:883-903) on a fresh copy of the state */
static unsigned me_mvd_bits_cabac(const me_ctx *mc, int dx, int dy)
{
  cabac_model m = cabac_start(mc);
  return cabac_mvd_bits(&m, dx, dy);
}
/* kvz_calc_mvd_cost_cabac (rdo.c:908-1060) */
static unsigned me_mv_cost_cabac(const me_ctx *mc, int x, int y, int mv_shift, unsigned *bitcost)
{
  const orc_me_pu *pu = mc->pu;
  const orc_me_params *prm = mc->prm;
  int merged = 0, merge_idx, cur_cand = 0, mvd[2] = { 0, 0 };
  x *= 1 << mv_shift;
  y *= 1 << mv_shift;
  for (merge_idx = 0; merge_idx < pu->num_merge_cand; ++merge_idx) {
    if (!pu->merge[merge_idx].usable) continue;
    if (pu->merge[merge_idx].mv[0] == x && pu->merge[merge_idx].mv[1] == y && pu->merge[merge_idx].same_ref) { merged = 1; break; }
  }
  if (!merged) {                                                  /* :952-972 */
    const int d1[2] = { x - pu->mv_cand[0][0], y - pu->mv_cand[0][1] }, d2[2] = { x - pu->mv_cand[1][0], y - pu->mv_cand[1][1] };
    const unsigned c1 = me_mvd_bits_cabac(mc, d1[0], d1[1]), c2 = me_mvd_bits_cabac(mc, d2[0], d2[1]);
    if (c2 < c1) { cur_cand = 1; mvd[0] = d2[0]; mvd[1] = d2[1]; } else { mvd[0] = d1[0]; mvd[1] = d1[1]; }
  }
  cabac_model m = cabac_start(mc);
  unsigned bits = cabac_bin(&m, CTX_MERGE_FLAG, merged);          /* :976 */
  if (merged) {                                                   /* :978-992: MRG_MAX_NUM_CANDS = 5 */
    for (int ui = 0; ui < 4; ++ui) {
      const int symbol = ui != merge_idx;
      bits += ui == 0 ? cabac_bin(&m, CTX_MERGE_IDX, symbol) : 1;
      if (!symbol) break;
    }
  } else {
    if (prm->refs_before > 1) {                                   /* :1004-1030 */
      int ref_frame = prm->ref_idx;
      bits += cabac_bin(&m, CTX_REF0, ref_frame != 0);
      if (ref_frame > 0) {
        const int ref_num = prm->refs_before - 2;
        --ref_frame;
        for (int i = 0; i < ref_num; ++i) {
          const int symbol = i == ref_frame ? 0 : 1;
          bits += i == 0 ? cabac_bin(&m, CTX_REF1, symbol) : 1;
          if (!symbol) break;
        }
      }
    }
    bits += cabac_mvd_bits(&m, mvd[0], mvd[1]);                   /* :1033-1037 */
    bits += cabac_bin(&m, CTX_MVP, cur_cand);                     /* kvz_cabac_write_unary_max_symbol(.., cur_mv_cand, 1, 1): one bin */
  }
  *bitcost = bits;
  return bits * (unsigned)prm->lambda_cost;
}

/* calc_mvd_cost (:373-412) */
static unsigned me_mv_cost(const me_ctx *mc, int x, int y, int mv_shift, unsigned *bitcost)
{
  if (!mc) { *bitcost = 0; return 0; }
  if (mc->prm->mv_rdo) return me_mv_cost_cabac(mc, x, y, mv_shift, bitcost);
  const orc_me_pu *pu = mc->pu;
  unsigned bits = 0;
  int merged = 0;
  x *= 1 << mv_shift;
  y *= 1 << mv_shift;
  for (int i = 0; i < pu->num_merge_cand; ++i) {
    if (!pu->merge[i].usable) continue;
    if (pu->merge[i].mv[0] == x && pu->merge[i].mv[1] == y && pu->merge[i].same_ref) { bits += (unsigned)i; merged = 1; break; }
  }
  if (!merged) { unsigned c; me_select_cand(pu, x, y, &c); bits += c; }
  *bitcost = bits;
  return bits * (unsigned)mc->prm->lambda_cost;
}

/* search_frac (:965-1128).  mc == NULL: no MV costs, no constraint (the 17 raw SATD costs).
 * mv_io: in = integer-pel MV, out = info->best_mv (quarter-pel). */
static void frac_search(const orc_pixel *pic, int pic_stride, const orc_pixel *ref, int ref_w, int ref_h,
                        int x, int y, int w, int h, int mv_io[2], const me_ctx *mc, int fme_level,
                        unsigned costs_out[17], int best_out[2], unsigned *best_cost_out, unsigned *best_bitcost_out)
{
  static const int sq[9][2] = { {0,0}, {-1,0}, {1,0}, {0,-1}, {0,1}, {-1,-1}, {1,-1}, {-1,1}, {1,1} };
  const int iw = ((w + 7) >> 3) << 3, ih = ((h + 7) >> 3) << 3;
  const int es = iw + 1 + 8;                         /* extended block stride */
  orc_pixel *ext = (orc_pixel *)malloc((size_t)es * (size_t)(ih + 1 + 8));
  orc_pixel *filtered = (orc_pixel *)malloc(4 * 64 * 64);
  orc_ipol_state *st = (orc_ipol_state *)calloc(1, sizeof(*st));
  const orc_pixel *src_tl; int src_stride; long off;
  int mx = mv_io[0], my = mv_io[1];

  if (orc_get_extended_block(x, y, mx - 1, my - 1, 0, 0, ref, ref_w, ref_h, 8, iw + 1, ih + 1, ext, &off)) {
    src_tl = ext + es * 4 + 4; src_stride = es;
  } else {
    src_tl = ref + off + (long)ref_w * 4 + 4; src_stride = ref_w;
  }
  const orc_pixel *cur = pic + (long)y * pic_stride + x;

  unsigned best_bitcost = 0;
  unsigned best_cost = orc_satd_any_size(w, h, cur, pic_stride, src_tl + src_stride + 1, src_stride);
  if (costs_out) costs_out[0] = best_cost;
  best_cost += me_mv_cost(mc, mx, my, 2, &best_bitcost);
  mx *= 2; my *= 2;                                   /* half-pel precision (:1031-1032) */
  int best_index = 0, i = 1, offx = 0, offy = 0;
  memset(filtered, 0, 4 * 64 * 64);
  for (int step = 0; step < fme_level; ++step) {
    const int mv_shift = step < 2 ? 1 : 0;
    unsigned c[4], bits[4] = { 0, 0, 0, 0 };
    int within[4];
    const orc_pixel *fp[4] = { filtered, filtered + 4096, filtered + 8192, filtered + 12288 };
    orc_filter_frac_blocks(step, src_tl, src_stride, iw, ih, filtered, st, fme_level, offx, offy);
    for (int j = 0; j < 4; ++j)
      within[j] = me_within(mc, (mx + sq[i + j][0]) * (1 << mv_shift), (my + sq[i + j][1]) * (1 << mv_shift));
    orc_satd_any_size_quad(w, h, fp, 64, cur, pic_stride, c);
    for (int j = 0; j < 4; ++j) {
      if (costs_out) costs_out[(step >= 2 ? 8 : 0) + i + j] = c[j];
      if (within[j]) c[j] += me_mv_cost(mc, mx + sq[i + j][0], my + sq[i + j][1], mv_shift, &bits[j]);
    }
    for (int j = 0; j < 4; ++j)
      if (within[j] && c[j] < best_cost) { best_cost = c[j]; best_bitcost = bits[j]; best_index = i + j; }
    i += 4;
    if (step == 1 || step == fme_level - 1) {          /* :1107-1122 */
      if (best_out && step == 3) best_out[1] = best_index;
      mx += sq[best_index][0]; my += sq[best_index][1];
      if (step == (fme_level - 1 < 1 ? fme_level - 1 : 1)) {
        if (best_out) best_out[0] = best_index;
        mx *= 2; my *= 2;                             /* quarter-pel precision */
        offx = sq[best_index][0]; offy = sq[best_index][1];
        best_index = 0; i = 1;
      }
    }
  }
  mv_io[0] = mx; mv_io[1] = my;
  if (best_cost_out) *best_cost_out = best_cost;
  if (best_bitcost_out) *best_bitcost_out = best_bitcost;
  free(st); free(filtered); free(ext);
}

/* search_inter.c:965-1128 without MV bit costs / tile constraints */
void orc_search_frac_costs(const orc_pixel *pic, int pic_stride,
                           const orc_pixel *ref, int ref_w, int ref_h,
                           int x, int y, int w, int h, int mvx, int mvy,
                           unsigned costs_out[17], int best_out[2])
{
  int mv[2] = { mvx, mvy };
  frac_search(pic, pic_stride, ref, ref_w, ref_h, x, y, w, h, mv, NULL, 4, costs_out, best_out, NULL, NULL);
}

/* ---- integer search state: inter_search_info_t's best_* fields (:40-76) ---- */
typedef struct {
  const orc_pixel *pic, *ref;
  int pic_stride, ref_w, ref_h;
  me_ctx mc;
  int best_mv[2];                  /* quarter-pel */
  unsigned best_cost, best_bitcost;
} me_info;

/* check_mv_cost (:195-232) */
static int me_check(me_info *in, int x, int y)
{
  const orc_me_pu *pu = in->mc.pu;
  if (!me_within(&in->mc, x * 4, y * 4)) return 0;
  unsigned bits = 0;
  unsigned cost = orc_image_calc_sad(in->pic, in->pic_stride, in->ref, in->ref_w, in->ref_w, in->ref_h,
                                     pu->x, pu->y, pu->x + x, pu->y + y, pu->width, pu->height);
  if (cost >= in->best_cost) return 0;
  cost += me_mv_cost(&in->mc, x, y, 2, &bits);
  if (cost >= in->best_cost) return 0;
  in->best_mv[0] = x * 4; in->best_mv[1] = y * 4;
  in->best_cost = cost; in->best_bitcost = bits;
  return 1;
}

/* mv_in_merge (:260-273) */
static int me_in_merge(const orc_me_pu *pu, int x, int y)
{
  for (int i = 0; i < pu->num_merge_cand; ++i) {
    if (!pu->merge[i].usable) continue;
    if (((pu->merge[i].mv[0] + 2) >> 2) == x && ((pu->merge[i].mv[1] + 2) >> 2) == y) return 1;
  }
  return 0;
}

/* the common opening of hexagon_search and diamond_search: select_starting_point (:282-307) and, if enabled,
 * early_terminate (:415-460).  Returns 1 when the search is finished. */
static int me_start(me_info *in)
{
  static const int et[7][2] = { {0,-1}, {-1,0}, {0,1}, {1,0}, {0,-1}, {-1,0}, {0,0} };
  const orc_me_pu *pu = in->mc.pu;
  const orc_me_params *prm = in->mc.prm;
  in->best_cost = 0xffffffffu;

  me_check(in, 0, 0);
  const int ex = pu->extra_mv[0] >> 2, ey = pu->extra_mv[1] >> 2;
  if ((ex != 0 || ey != 0) && !me_in_merge(pu, ex, ey)) me_check(in, ex, ey);
  for (int i = 0; i < pu->num_merge_cand; ++i) {
    if (!pu->merge[i].usable) continue;
    const int x = (pu->merge[i].mv[0] + 2) >> 2, y = (pu->merge[i].mv[1] + 2) >> 2;
    if (x == 0 && y == 0) continue;
    me_check(in, x, y);
  }

  if (prm->early_termination) {
    int mvx = in->best_mv[0] >> 2, mvy = in->best_mv[1] >> 2;
    int first = 0, last = 3;
    for (int k = 0; k < 2; ++k) {
      const double threshold = prm->early_termination == 2 ? in->best_cost * 0.95 : (double)in->best_cost;
      int best_index = 6;
      for (int i = first; i <= last; ++i)
        if (me_check(in, mvx + et[i][0], mvy + et[i][1])) best_index = i;
      mvx += et[best_index][0]; mvy += et[best_index][1];
      if (in->best_cost >= threshold) return 1;
      first = (best_index + 3) % 4;
      last = first + 2;
    }
  }
  return 0;
}

/* diamond_search (:796-883) */
static void me_diamond(me_info *in)
{
  static const int dia[5][2] = { {0,-1}, {1,0}, {0,1}, {-1,0}, {0,0} };
  if (me_start(in)) return;
  int mvx = in->best_mv[0] >> 2, mvy = in->best_mv[1] >> 2;
  int best_index = 4;
  unsigned steps = in->mc.prm->max_steps;
  for (int i = 0; i < 5; ++i)
    if (me_check(in, mvx + dia[i][0], mvy + dia[i][1])) best_index = i;
  if (best_index == 4) return;
  mvx += dia[best_index][0]; mvy += dia[best_index][1];
  int from_dir = 4, better;
  do {
    better = 0;
    if (steps > 0) steps -= 1;
    for (int i = 0; i < 4; ++i) {
      if (i == from_dir) continue;
      if (me_check(in, mvx + dia[i][0], mvy + dia[i][1])) { best_index = i; better = 1; }
    }
    if (better) {
      mvx += dia[best_index][0]; mvy += dia[best_index][1];
      from_dir = best_index ^ 3;
    }
  } while (better && steps != 0);
}

/* kvz_tz_pattern_search (:463-577) with pattern_type 0, the 8-point diamond (4 points at distance 1) */
static void me_tz_pattern(me_info *in, int dist, int sx, int sy, int *best_dist)
{
  const int pat[8][2] = { {0, dist}, {dist, 0}, {0, -dist}, {-dist, 0},
                          {dist / 2, dist / 2}, {dist / 2, -dist / 2}, {-dist / 2, -dist / 2}, {-dist / 2, dist / 2} };
  const int n = dist == 1 ? 4 : 8;
  int improved = 0;
  for (int i = 0; i < n; ++i)
    if (me_check(in, sx + pat[i][0], sy + pat[i][1])) improved = 1;
  if (improved) *best_dist = dist;
}

/* tz_search (:595-672) with its fixed parameters: search range 96, diamond patterns, no raster scan, star refinement */
static void me_tz(me_info *in)
{
  const int range = 96;
  int best_dist = 0;
  if (me_start(in)) return;
  int sx = in->best_mv[0] >> 2, sy = in->best_mv[1] >> 2;
  int rounds = 0;
  for (int dist = 1; dist <= range; dist *= 2) {
    me_tz_pattern(in, dist, sx, sy, &best_dist);
    if (best_dist != dist) rounds++;
    if (rounds >= 3) break;
  }
  if (sx != 0 || sy != 0) {
    rounds = 0;
    for (int dist = 1; dist <= range / 2; dist *= 2) {
      me_tz_pattern(in, dist, 0, 0, &best_dist);
      if (best_dist != dist) rounds++;
      if (rounds >= 3) break;
    }
  }
  while (best_dist > 0) {
    best_dist = 0;
    sx = in->best_mv[0] >> 2; sy = in->best_mv[1] >> 2;
    for (int dist = 1; dist <= range; dist *= 2) me_tz_pattern(in, dist, sx, sy, &best_dist);
  }
}

/* search_mv_full (:886-962): every position of a (2R+1)^2 window around the zero vector, around extra_mv (unless it
 * is a merge candidate) and around each merge candidate, skipping positions inside a window visited earlier */
static void me_full(me_info *in, int range)
{
  const orc_me_pu *pu = in->mc.pu;
  in->best_cost = 0xffffffffu;
  for (int y = -range; y <= range; ++y)
    for (int x = -range; x <= range; ++x) me_check(in, x, y);
  const int ex = pu->extra_mv[0] >> 2, ey = pu->extra_mv[1] >> 2;
  if (!me_in_merge(pu, ex, ey))
    for (int y = -range; y <= range; ++y)
      for (int x = -range; x <= range; ++x) me_check(in, ex + x, ey + y);
  for (int i = 0; i < pu->num_merge_cand; ++i) {
    if (!pu->merge[i].usable) continue;
    const int mx = pu->merge[i].mv[0] >> 2, my = pu->merge[i].mv[1] >> 2;     /* plain shift here (:917-920) */
    if (mx == 0 && my == 0) continue;
    for (int y = my - range; y <= my + range; ++y)
      for (int x = mx - range; x <= mx + range; ++x) {
        if (!me_within(&in->mc, x * 4, y * 4)) continue;
        int tested = 0;
        for (int j = -1; j < i; ++j) {
          int xx = 0, yy = 0;
          if (j >= 0) {
            if (!pu->merge[j].usable) continue;
            xx = pu->merge[j].mv[0] >> 2; yy = pu->merge[j].mv[1] >> 2;
          }
          if (x >= xx - range && x <= xx + range && y >= yy - range && y <= yy + range) {
            tested = 1;
            x = xx + range;                                                    /* jump past the earlier window (:948) */
            break;
          }
        }
        if (tested) continue;
        me_check(in, x, y);
      }
  }
}

/* hexagon_search (:690-778) */
static void me_hexagon(me_info *in)
{
  static const int large[9][2] = { {0,0}, {1,-2}, {2,0}, {1,2}, {-1,2}, {-2,0}, {-1,-2}, {1,-2}, {2,0} };
  static const int small[9][2] = { {0,0}, {0,-1}, {-1,0}, {1,0}, {0,1}, {-1,-1}, {1,-1}, {-1,1}, {1,1} };
  const orc_me_params *prm = in->mc.prm;
  if (me_start(in)) return;

  int mvx = in->best_mv[0] >> 2, mvy = in->best_mv[1] >> 2;
  int best_index = 0;
  unsigned steps = prm->max_steps;
  for (int i = 1; i < 7; ++i)
    if (me_check(in, mvx + large[i][0], mvy + large[i][1])) best_index = i;
  while (best_index != 0 && steps != 0) {
    if (steps > 0) steps -= 1;
    const int start = best_index == 1 ? 6 : (best_index == 8 ? 1 : best_index - 1);
    mvx += large[best_index][0]; mvy += large[best_index][1];
    best_index = 0;
    for (int i = 0; i < 3; ++i)
      if (me_check(in, mvx + large[start + i][0], mvy + large[start + i][1])) best_index = start + i;
  }
  for (int i = 1; i < 9; ++i) me_check(in, mvx + small[i][0], mvy + small[i][1]);
}

/* the hexbs path of search_pu_inter_ref (:1134-1300) for one reference picture; inter_cost = *inter_cost on entry */
static void search_pu_against(const orc_pixel *pic, int pic_stride, const orc_pixel *ref, int ref_w, int ref_h,
                              const orc_me_pu *pu, const orc_me_params *prm, unsigned inter_cost, orc_me_result *res);

void orc_search_pu(const orc_pixel *pic, int pic_stride, const orc_pixel *ref, int ref_w, int ref_h,
                   const orc_me_pu *pu, const orc_me_params *prm, orc_me_result *res)
{
  search_pu_against(pic, pic_stride, ref, ref_w, ref_h, pu, prm, 0xffffffffu, res);      /* the first picture searched: MAX_INT (:1456) */
}

void orc_search_pu_many(const orc_pixel *pic, int pic_stride, const orc_pixel *ref, int ref_w, int ref_h,
                        const orc_me_pu *pus, size_t count, const orc_me_params *prm, orc_me_result *res)
{
  for (size_t i = 0; i < count; ++i)
    search_pu_against(pic, pic_stride, ref, ref_w, ref_h, &pus[i], prm, prm->cost_to_beat ? prm->cost_to_beat[i] : 0xffffffffu, &res[i]);
}

static void search_pu_against(const orc_pixel *pic, int pic_stride, const orc_pixel *ref, int ref_w, int ref_h,
                              const orc_me_pu *pu, const orc_me_params *prm, unsigned inter_cost, orc_me_result *res)
{
  me_info in;
  memset(&in, 0, sizeof(in));
  orc_me_params whole;
  if (prm->tile_w == 0 && prm->tile_h == 0) {            /* no tiles: the picture (same size as its references) is the tile */
    whole = *prm; whole.tile_x = 0; whole.tile_y = 0; whole.tile_w = ref_w; whole.tile_h = ref_h;
    prm = &whole;
  }
  in.pic = pic; in.ref = ref; in.pic_stride = pic_stride; in.ref_w = ref_w; in.ref_h = ref_h;
  in.mc.pu = pu; in.mc.prm = prm;
  if (prm->algorithm == 1) me_diamond(&in);
  else if (prm->algorithm == 2) me_tz(&in);
  else if (prm->algorithm == 3) me_full(&in, prm->search_range);
  else me_hexagon(&in);
  if (prm->fme_level > 0 && in.best_cost < inter_cost) {    /* :1239 */
    int mv[2] = { in.best_mv[0] >> 2, in.best_mv[1] >> 2 };
    frac_search(pic, pic_stride, ref, ref_w, ref_h, pu->x, pu->y, pu->width, pu->height, mv, &in.mc, prm->fme_level,
                NULL, NULL, &in.best_cost, &in.best_bitcost);
    in.best_mv[0] = mv[0]; in.best_mv[1] = mv[1];
  } else if (in.best_cost < 0xffffffffu) {                  /* :1236-1248 */
    in.best_cost = orc_image_calc_satd(pic, pic_stride, ref, ref_w, ref_w, ref_h, pu->x, pu->y,
                                       pu->x + (in.best_mv[0] >> 2), pu->y + (in.best_mv[1] >> 2), pu->width, pu->height);
    in.best_cost += in.best_bitcost * (unsigned)prm->lambda_cost;
  }
  memset(res, 0, sizeof(*res));
  res->mv[0] = in.best_mv[0]; res->mv[1] = in.best_mv[1];
  res->cost = in.best_cost; res->bitcost = in.best_bitcost;
  int idx = 0;
  for (idx = 0; idx < pu->num_merge_cand; ++idx)
    if (pu->merge[idx].usable && pu->merge[idx].mv[0] == in.best_mv[0] && pu->merge[idx].mv[1] == in.best_mv[1] &&
        pu->merge[idx].same_ref) { res->merged = 1; break; }
  res->merge_idx = idx;
  if (!res->merged) {
    /* select_mv_cand with cost_out == NULL returns 0 for identical candidates (:332-338): same answer */
    if (prm->mv_rdo) {   /* select_mv_cand with kvz_get_mvd_coding_cost_cabac (:343-344) */
      const unsigned c1 = me_mvd_bits_cabac(&in.mc, in.best_mv[0] - pu->mv_cand[0][0], in.best_mv[1] - pu->mv_cand[0][1]);
      const unsigned c2 = me_mvd_bits_cabac(&in.mc, in.best_mv[0] - pu->mv_cand[1][0], in.best_mv[1] - pu->mv_cand[1][1]);
      res->mv_cand = c2 < c1 ? 1 : 0;
    } else
    res->mv_cand = me_select_cand(pu, in.best_mv[0], in.best_mv[1], NULL);
  }
}

/* Batched ME costs of one CTU: the values check_mv_cost (search_inter.c:195-232) obtains
 * one by one from kvz_image_calc_sad (image.c:455-486), for every candidate offset and
 * each of the 85 square PUs (64x64, 4 x 32x32, 16 x 16x16, 64 x 8x8; raster order per
 * size).  PUs not entirely inside the picture get 0xFFFFFFFF. */
void orc_ctu_sad_grid(const orc_pixel *pic, int pic_stride, int pic_w, int pic_h,
                      const orc_pixel *ref, int ref_stride, int ref_w, int ref_h,
                      int ctu_x, int ctu_y, int mvx, int mvy,
                      const int16_t *mv_offsets, int n_mv, uint32_t *costs)
{
  static const int sizes[4] = { 64, 32, 16, 8 }, first[4] = { 0, 1, 5, 21 };
  for (int m = 0; m < n_mv; ++m) {
    const int dx = mv_offsets[2 * m], dy = mv_offsets[2 * m + 1];
    uint32_t *o = costs + (size_t)m * 85;
    for (int l = 0; l < 4; ++l) {
      const int n = sizes[l], per = 64 / n;
      for (int j = 0; j < per; ++j)
        for (int i = 0; i < per; ++i) {
          const int bx = ctu_x + i * n, by = ctu_y + j * n;
          uint32_t v = 0xffffffffu;
          if (dx >= -64 && dx <= 64 && dy >= -64 && dy <= 64 && bx + n <= pic_w && by + n <= pic_h)
            v = orc_image_calc_sad(pic, pic_stride, ref, ref_stride, ref_w, ref_h, bx, by, bx + mvx + dx, by + mvy + dy, n, n);
          o[first[l] + j * per + i] = v;
        }
    }
  }
}

/* =====================================================================
 * intra group (intra-generic.c, intra.c)
 * ===================================================================== */

/* intra-generic.c:46-47: displacement per row in 1/32 sample for |mode - 26| (or |10 - mode|),
 * and 256*32/disp for projecting the side reference */
static const int orc_ang_disp[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
static const int orc_ang_inv[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };

/* intra-generic.c:37-145.  Restated per pixel: the prediction is built in the "vertical"
 * orientation from an extended main reference m[-N .. 2N] and transposed for modes < 18. */
void orc_angular_pred(int log2_width, int mode, const orc_pixel *ref_above, const orc_pixel *ref_left, orc_pixel *dst)
{
  const int n = 1 << log2_width;
  const int vertical = mode >= 18;
  const int md = vertical ? mode - 26 : 10 - mode;
  const int amd = md < 0 ? -md : md;
  const int disp = md < 0 ? -orc_ang_disp[amd] : orc_ang_disp[amd];
  const orc_pixel *mainr = vertical ? ref_above : ref_left;   /* index 0 = corner */
  const orc_pixel *side = vertical ? ref_left : ref_above;
  int ext_store[3 * 32 + 2];
  int *m = ext_store + 32;                                    /* m[i], i = -n .. 2n */
  for (int i = -n; i <= 2 * n; ++i) m[i] = 0;
  for (int i = -1; i < 2 * n; ++i) m[i] = mainr[i + 1];
  if (disp < 0) {
    /* :78-93: indices below -1 are projected from the side reference */
    const int most_negative = (n * disp) >> 5;
    for (int i = -2; i >= most_negative; --i) m[i] = side[(128 + (-i - 1) * orc_ang_inv[amd]) >> 8];
  }
  for (int y = 0; y < n; ++y) {
    const int pos = (y + 1) * disp;
    const int di = pos >> 5, f = pos & 31;                    /* :107-108 (arithmetic shift) */
    for (int x = 0; x < n; ++x) {
      /* :110-123; with f == 0 the second tap has weight 0 and is not read by the reference */
      const int v = f ? ((32 - f) * m[x + di] + f * m[x + di + 1] + 16) >> 5 : m[x + di];
      if (vertical) dst[y * n + x] = (orc_pixel)v; else dst[x * n + y] = (orc_pixel)v;   /* :137-144 flip */
    }
  }
}

/* intra-generic.c:155-189 in its closed form (the #if 0 branch :167-175) */
void orc_intra_pred_planar(int log2_width, const orc_pixel *ref_top, const orc_pixel *ref_left, orc_pixel *dst)
{
  const int n = 1 << log2_width;
  const int tr = ref_top[n + 1], bl = ref_left[n + 1];
  for (int y = 0; y < n; ++y)
    for (int x = 0; x < n; ++x) {
      const int hor = (n - 1 - x) * ref_left[y + 1] + (x + 1) * tr;
      const int ver = (n - 1 - y) * ref_top[x + 1] + (y + 1) * bl;
      dst[y * n + x] = (orc_pixel)((hor + ver + n) >> (log2_width + 1));
    }
}

/* intra.c:164-192 */
void orc_intra_filter_reference(int log2_width, const orc_intra_ref *ref, orc_intra_ref *fil)
{
  const int last = 2 * (1 << log2_width);                     /* ref_width - 1 */
  fil->left[0] = fil->top[0] = (orc_pixel)((ref->left[1] + 2 * ref->left[0] + ref->top[1] + 2) / 4);
  for (int i = 1; i < last; ++i) {
    fil->left[i] = (orc_pixel)((ref->left[i - 1] + 2 * ref->left[i] + ref->left[i + 1] + 2) / 4);
    fil->top[i] = (orc_pixel)((ref->top[i - 1] + 2 * ref->top[i] + ref->top[i + 1] + 2) / 4);
  }
  fil->left[last] = ref->left[last];
  fil->top[last] = ref->top[last];
}

/* intra.c:281-331 with intra_pred_dc :217-237, intra_pred_filtered_dc :247-278,
 * intra_post_process_angular :195-208 */
void orc_intra_predict(const orc_intra_ref *ref, int log2_width, int mode, int is_luma, int filter_boundary, orc_pixel *dst)
{
  const int n = 1 << log2_width;
  orc_intra_ref fil;
  const orc_intra_ref *used = ref;
  if (!is_luma || mode == 1 || n == 4) {
    /* :291-292 unfiltered */
  } else if (mode == 0) {
    used = &fil;
  } else {
    static const int thres[5] = { 0, 7, 1, 0, 0 };              /* :298 */
    const int dv = mode > 26 ? mode - 26 : 26 - mode, dh = mode > 10 ? mode - 10 : 10 - mode;
    if ((dv < dh ? dv : dh) > thres[log2_width - 2]) used = &fil;
  }
  if (used == &fil) orc_intra_filter_reference(log2_width, ref, &fil);

  if (mode == 0) {
    orc_intra_pred_planar(log2_width, used->top, used->left, dst);
  } else if (mode == 1) {
    int sum = 0;
    for (int i = 1; i <= n; ++i) sum += used->top[i] + used->left[i];
    const int dc = (orc_pixel)((sum + n) >> (log2_width + 1));
    for (int i = 0; i < n * n; ++i) dst[i] = (orc_pixel)dc;
    if (is_luma && n < 32) {
      dst[0] = (orc_pixel)((used->left[1] + 2 * dc + used->top[1] + 2) / 4);
      for (int x = 1; x < n; ++x) dst[x] = (orc_pixel)((used->top[x + 1] + 3 * dc + 2) / 4);
      for (int y = 1; y < n; ++y) dst[y * n] = (orc_pixel)((used->left[y + 1] + 3 * dc + 2) / 4);
    }
  } else {
    orc_angular_pred(log2_width, mode, used->top, used->left, dst);
    if (is_luma && n < 32 && filter_boundary && (mode == 10 || mode == 26)) {
      /* mode 10: first row corrected with the top gradient; mode 26: first column with the left gradient */
      const orc_pixel *g = mode == 10 ? used->top : used->left;
      const int stride = mode == 10 ? 1 : n;
      for (int i = 0; i < n; ++i) {
        int v = dst[i * stride] + ((g[i + 1] - g[0]) >> 1);
        dst[i * stride] = (orc_pixel)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
    }
  }
}

void orc_intra_rough_costs(const orc_intra_ref *ref, int log2_width, int filter_boundary, const orc_pixel *orig,
                           unsigned satd_out[35], unsigned sad_out[35])
{
  const int n = 1 << log2_width;
  orc_pixel pred[32 * 32];
  for (int mode = 0; mode < 35; ++mode) {
    orc_intra_predict(ref, log2_width, mode, 1, filter_boundary, pred);
    /* get_cost / get_cost_dual (search_intra.c:99-172): satd_func(pred, orig) and sad_func(pred, orig) */
    if (satd_out) satd_out[mode] = n == 4 ? orc_satd_4x4(pred, orig) : orc_satd_nxn(n, pred, orig);
    if (sad_out) sad_out[mode] = orc_sad_nxn(n, pred, orig);
  }
}

/* ---- kvz_intra_build_reference (intra.c:334-588), read from a whole reconstruction plane ----
 * The reference reads lcu->rec inside the LCU and lcu->top_ref / left_ref on its border; both are views of the
 * reconstruction before deblocking (init_lcu_t, search.c:761-835, fills the borders from the hor_buf / ver_buf
 * kept before the loop filters), so on a plane the neighbour above
 * is rec[(y - 1) * stride + x + i] and the neighbour to the left rec[(y + i) * stride + x - 1] wherever they exist. */

/* position of a 4x4 unit in the coding order of its LCU: the bits of (ux, uy) interleaved */
static unsigned intra_unit_order(unsigned ux, unsigned uy)
{
  unsigned z = 0;
  for (int b = 0; b < 4; ++b) z |= ((ux >> b) & 1u) << (2 * b) | ((uy >> b) & 1u) << (2 * b + 1);
  return z;
}

/* num_ref_pixels_top / num_ref_pixels_left (intra.c:35-70) from first principles: luma pixels of the row above /
 * the column to the left that were coded before the unit at (ux, uy); the whole LCU row above and the LCU to the
 * left are complete, the LCU below-left never is, and at most 64 pixels are ever asked for. */
static int intra_coded_above(int ux, int uy)
{
  if (uy == 0) return 64;
  int n = 0;
  while (ux + n < 16 && intra_unit_order(ux + n, uy - 1) < intra_unit_order(ux, uy)) ++n;
  return 4 * n;
}

static int intra_coded_left(int ux, int uy)
{
  int n = 0;
  while (uy + n < 16 && (ux == 0 || intra_unit_order(ux - 1, uy + n) < intra_unit_order(ux, uy))) ++n;
  return 4 * n;
}

void orc_intra_build_reference(int log2_width, int color, const orc_pixel *rec, int stride, int pic_w, int pic_h,
                               int luma_x, int luma_y, orc_intra_ref *out)
{
  const int c = color != 0, n2 = 2 << log2_width;
  const int x = luma_x >> c, y = luma_y >> c;                 /* position in the plane of `color` */
  const int ux = (luma_x & 63) >> 2, uy = (luma_y & 63) >> 2;
  const orc_pixel dc = 1 << 7;                                /* intra.c:348, KVZ_BIT_DEPTH 8 */
  const int has_left = luma_x > 0, has_top = luma_y > 0;

  /* left column, intra.c:386-412 / :519-543: the coded pixels, then the last of them repeated; at the left picture
   * edge the first pixel above (or mid grey at the origin) */
  if (has_left) {
    int avail = intra_coded_left(ux, uy) >> c;
    if (avail > n2) avail = n2;
    if (avail > ((pic_h - luma_y) >> c)) avail = (pic_h - luma_y) >> c;
    for (int i = 0; i < n2; ++i) out->left[1 + i] = rec[(y + (i < avail ? i : avail - 1)) * stride + x - 1];
  } else {
    const orc_pixel v = has_top ? rec[(y - 1) * stride + x] : dc;
    for (int i = 0; i < n2; ++i) out->left[1 + i] = v;
  }

  /* corner, intra.c:414-428 / :504-517 */
  out->left[0] = (has_left && has_top) ? rec[(y - 1) * stride + x - 1] : out->left[1];
  out->top[0] = out->left[0];

  /* row above, intra.c:430-455 / :545-571 */
  if (has_top) {
    int avail = intra_coded_above(ux, uy) >> c;
    if (avail > n2) avail = n2;
    if (avail > ((pic_w - luma_x) >> c)) avail = (pic_w - luma_x) >> c;
    for (int i = 0; i < n2; ++i) out->top[1 + i] = rec[(y - 1) * stride + x + (i < avail ? i : avail - 1)];
  } else {
    const orc_pixel v = has_left ? rec[y * stride + x - 1] : dc;
    for (int i = 0; i < n2; ++i) out->top[1 + i] = v;
  }
}

void orc_intra_build_reference_many(int log2_width, int color, const orc_pixel *rec, int stride, int pic_w, int pic_h,
                                    const int32_t *xy, size_t count, orc_intra_ref *out)
{
  for (size_t i = 0; i < count; ++i) {
    memset(&out[i], 0, sizeof(out[i]));
    orc_intra_build_reference(log2_width, color, rec, stride, pic_w, pic_h, xy[2 * i], xy[2 * i + 1], &out[i]);
  }
}

/* =====================================================================
 * SAO group (sao-generic.c, sao.c)
 * ===================================================================== */

/* g_sao_edge_offsets (sao.h:58-63): neighbours a and b of c per edge class */
static const int orc_sao_ofs[4][2][2] = { { { -1, 0 }, { 1, 0 } }, { { 0, -1 }, { 0, 1 } }, { { -1, -1 }, { 1, 1 } }, { { 1, -1 }, { -1, 1 } } };

/* sao_calc_eo_cat (sao-generic.c:34-43): 2 + sign(c-a) + sign(c-b) mapped through {1,2,0,3,4} */
static int orc_sao_cat(int a, int b, int c)
{
  static const int map[5] = { 1, 2, 0, 3, 4 };
  const int sa = (c > a) - (c < a), sb = (c > b) - (c < b);
  return map[2 + sa + sb];
}

static int orc_sao_cat_at(const orc_pixel *rec, int stride, int x, int y, int eo_class)
{
  const int a = rec[(y + orc_sao_ofs[eo_class][0][1]) * stride + x + orc_sao_ofs[eo_class][0][0]];
  const int b = rec[(y + orc_sao_ofs[eo_class][1][1]) * stride + x + orc_sao_ofs[eo_class][1][0]];
  return orc_sao_cat(a, b, rec[y * stride + x]);
}

int orc_sao_edge_ddistortion(const orc_pixel *orig, const orc_pixel *rec, int bw, int bh, int eo_class, const int offsets[5])
{
  int sum = 0;
  for (int y = 1; y < bh - 1; ++y)
    for (int x = 1; x < bw - 1; ++x) {
      const int offset = offsets[orc_sao_cat_at(rec, bw, x, y, eo_class)];
      if (offset != 0) {
        const int diff = orig[y * bw + x] - rec[y * bw + x];
        sum += (diff - offset) * (diff - offset) - diff * diff;
      }
    }
  return sum;
}

void orc_calc_sao_edge_dir(const orc_pixel *orig, const orc_pixel *rec, int eo_class, int bw, int bh, int cat_sum_cnt[2][5])
{
  for (int y = 1; y < bh - 1; ++y)
    for (int x = 1; x < bw - 1; ++x) {
      const int cat = orc_sao_cat_at(rec, bw, x, y, eo_class);
      cat_sum_cnt[0][cat] += orig[y * bw + x] - rec[y * bw + x];
      cat_sum_cnt[1][cat] += 1;
    }
}

void orc_sao_reconstruct_color(const orc_pixel *rec, orc_pixel *new_rec, const orc_sao_info *sao, int stride, int new_stride,
                               int bw, int bh, int color)
{
  const int is_v = color == 2;
  if (sao->type == 1) {
    /* kvz_calc_sao_offset_array (sao.c:164-180) applied per pixel; bitdepth 8: band = value >> 3 */
    const int bp = sao->band_position[is_v];
    for (int y = 0; y < bh; ++y)
      for (int x = 0; x < bw; ++x) {
        const int val = rec[y * stride + x], band = val >> 3;
        int v = val;
        if (band >= bp && band < bp + 4) {
          v = val + sao->offsets[band - bp + 1 + 5 * is_v];
          v = v < 0 ? 0 : v > 255 ? 255 : v;
        }
        new_rec[y * new_stride + x] = (orc_pixel)v;
      }
  } else {
    for (int y = 0; y < bh; ++y)
      for (int x = 0; x < bw; ++x) {
        int v = rec[y * stride + x] + sao->offsets[orc_sao_cat_at(rec, stride, x, y, sao->eo_class) + 5 * is_v];
        new_rec[y * new_stride + x] = (orc_pixel)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
  }
}

int orc_sao_band_ddistortion(const orc_pixel *orig, const orc_pixel *rec, int bw, int bh, int band_pos, const int sao_bands[4])
{
  int sum = 0;
  for (int i = 0; i < bw * bh; ++i) {
    const int band = (rec[i] >> 3) - band_pos;
    const int offset = (band >= 0 && band < 4) ? sao_bands[band] : 0;
    if (offset != 0) {
      const int diff = orig[i] - rec[i];
      sum += (diff - offset) * (diff - offset) - diff * diff;
    }
  }
  return sum;
}

void orc_calc_sao_bands(const orc_pixel *orig, const orc_pixel *rec, int bw, int bh, int sao_bands[2][32])
{
  for (int i = 0; i < bw * bh; ++i) {
    sao_bands[0][rec[i] >> 3] += orig[i] - rec[i];
    sao_bands[1][rec[i] >> 3] += 1;
  }
}

/* =====================================================================
 * bi-prediction candidate cost (search_inter.c:1304-1440, inter.c:300-477)
 * ===================================================================== */

/* one reference's luma predictor as the blend sees it: *hi = 1 and hp[] (stride w) for a fractional vector
 * (inter_recon_14bit_frac_luma, inter.c:86-122), else *hi = 0 and px[] (stride w) (inter.c:355-371) */
static void bipred_luma_part(const orc_pixel *ref, int ref_w, int ref_h, int x, int y, int w, int h, const int16_t mv[2],
                             int *hi, int16_t *hp, orc_pixel *px)
{
  *hi = (mv[0] & 3) || (mv[1] & 3);
  const int ix = x + (mv[0] >> 2), iy = y + (mv[1] >> 2);
  if (*hi) {
    const int es = w + 8;
    orc_pixel *ext = (orc_pixel *)malloc((size_t)es * (size_t)(h + 8));
    const orc_pixel *src; int stride; long off;
    if (orc_get_extended_block(x, y, mv[0] >> 2, mv[1] >> 2, 0, 0, ref, ref_w, ref_h, 8, w, h, ext, &off)) {
      src = ext + es * 4 + 4; stride = es;                /* orig_topleft: half the filter size inside the copy */
    } else {
      src = ref + off + (long)ref_w * 4 + 4; stride = ref_w;
    }
    orc_sample_14bit_quarterpel_luma(src, stride, w, h, hp, w, mv);
    free(ext);
  } else {
    for (int r = 0; r < h; ++r)
      for (int c = 0; c < w; ++c)
        px[r * w + c] = ref[ORC_CLIP(0, ref_h - 1, iy + r) * ref_w + ORC_CLIP(0, ref_w - 1, ix + c)];
  }
}

unsigned orc_bipred_luma_satd(const orc_pixel *pic, int pic_stride, const orc_pixel *ref0, const orc_pixel *ref1, int ref_w, int ref_h,
                              int x, int y, int w, int h, const int16_t mv0[2], const int16_t mv1[2], orc_pixel *out)
{
  int16_t *hp0 = (int16_t *)malloc((size_t)w * h * 2), *hp1 = (int16_t *)malloc((size_t)w * h * 2);
  orc_pixel *px0 = (orc_pixel *)malloc((size_t)w * h), *px1 = (orc_pixel *)malloc((size_t)w * h), *pred = (orc_pixel *)malloc((size_t)w * h);
  int hi0, hi1;
  bipred_luma_part(ref0, ref_w, ref_h, x, y, w, h, mv0, &hi0, hp0, px0);
  bipred_luma_part(ref1, ref_w, ref_h, x, y, w, h, mv1, &hi1, hp1, px1);
  orc_bipred_blend_plane(w, h, hi0, hp0, px0, w, hi1, hp1, px1, w, pred, w);
  /* search_inter.c:1359-1362: kvz_satd_any_size(width, height, rec, LCU_WIDTH, src, stride) */
  const unsigned cost = orc_satd_any_size(w, h, pred, w, pic + (long)y * pic_stride + x, pic_stride);
  if (out) memcpy(out, pred, (size_t)w * h);
  free(hp0); free(hp1); free(px0); free(px1); free(pred);
  return cost;
}


/* ======================================================================== */
/* deblocking -- src/filter.c                                               */
/* ======================================================================== */
static const uint8_t db_tc_table[54] = {       /* kvz_g_tc_table_8x8, filter.c:34-42 */
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5,
  6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
static const uint8_t db_beta_table[52] = {     /* kvz_g_beta_table_8x8, filter.c:44-52 */
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32,
  34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
static const uint8_t db_chroma_scale[58] = {   /* kvz_g_chroma_scale, transform.c:44-50 */
  0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
  33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
/* kvz_part_mode_num_parts / kvz_part_mode_offsets, cu.c:33-60 */
static const uint8_t db_num_parts[8] = { 1, 2, 2, 4, 2, 2, 2, 2 };
static const uint8_t db_part_off[8][4][2] = {
  { { 0, 0 } }, { { 0, 0 }, { 0, 2 } }, { { 0, 0 }, { 2, 0 } }, { { 0, 0 }, { 2, 0 }, { 0, 2 }, { 2, 2 } },
  { { 0, 0 }, { 0, 1 } }, { { 0, 0 }, { 0, 3 } }, { { 0, 0 }, { 1, 0 } }, { { 0, 0 }, { 3, 0 } } };

typedef struct { const orc_cu_info *cus; int w4, width, height; const orc_deblock_params *prm; } db_ctx;
static const orc_cu_info *db_cu(const db_ctx *c, int x, int y) { return &c->cus[(y >> 2) * c->w4 + (x >> 2)]; }
static int db_clip(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

/* is_tu_boundary (filter.c:190-206) || is_pu_boundary (:216-243); dir 0 = vertical edge, 1 = horizontal edge */
static int db_boundary(const db_ctx *c, int x, int y, int dir, int *tu_boundary)
{
  const orc_cu_info *scu = db_cu(c, x, y);
  const int tu_width = 64 >> scu->tr_depth;
  *tu_boundary = ((dir ? y : x) & (tu_width - 1)) == 0;
  if (*tu_boundary) return 1;
  const int cu_width = 64 >> scu->depth, x_cu = x & ~(cu_width - 1), y_cu = y & ~(cu_width - 1);
  const orc_cu_info *cu = db_cu(c, x_cu, y_cu);
  for (int i = 0; i < db_num_parts[cu->part_size & 7]; ++i) {
    if (dir) { if (y_cu + db_part_off[cu->part_size & 7][i][1] * cu_width / 4 == y) return 1; }
    else if (x_cu + db_part_off[cu->part_size & 7][i][0] * cu_width / 4 == x) return 1;
  }
  return 0;
}
/* get_qp_y_pred (:263-282) */
static int db_qp(const db_ctx *c, int x, int y, int dir)
{
  if (!c->prm->per_cu_qp) return c->prm->qp;
  int qp_p;
  if (dir && y > 0) qp_p = db_cu(c, x, y - 1)->qp;
  else if (!dir && x > 0) qp_p = db_cu(c, x - 1, y)->qp;
  else qp_p = c->prm->frame_qp;
  return (qp_p + db_cu(c, x, y)->qp + 1) >> 1;
}
/* the boundary strength of filter_deblock_edge_luma (:379-460) */
static int db_strength(const db_ctx *c, const orc_cu_info *p, const orc_cu_info *q, int tu_boundary)
{
  if (q->type == 1 || p->type == 1) return 2;
  if (tu_boundary && (q->cbf_y || p->cbf_y)) return 1;
  if (p->mv_dir != 3 && q->mv_dir != 3) {
    const int lp = (p->mv_dir - 1) & 1, lq = (q->mv_dir - 1) & 1;
    if (abs(q->mv[lq][0] - p->mv[lp][0]) >= 4 || abs(q->mv[lq][1] - p->mv[lp][1]) >= 4) return 1;
    if (q->mv_ref[lq] != p->mv_ref[lp]) return 1;
  }
  if (!c->prm->slice_is_b) return 0;
  /* B slices: undefined vectors count as zero (:400-417) */
  int mvp[2][2], mvq[2][2];
  for (int l = 0; l < 2; ++l)
    for (int k = 0; k < 2; ++k) {
      mvp[l][k] = (p->mv_dir & (1 << l)) ? p->mv[l][k] : 0;
      mvq[l][k] = (q->mv_dir & (1 << l)) ? q->mv[l][k] : 0;
    }
  const int refP0 = (p->mv_dir & 1) ? c->prm->ref_LX[0][p->mv_ref[0] & 15] : -1, refP1 = (p->mv_dir & 2) ? c->prm->ref_LX[1][p->mv_ref[1] & 15] : -1;
  const int refQ0 = (q->mv_dir & 1) ? c->prm->ref_LX[0][q->mv_ref[0] & 15] : -1, refQ1 = (q->mv_dir & 2) ? c->prm->ref_LX[1][q->mv_ref[1] & 15] : -1;
#define DB_FAR(a, b) (abs((a)[0] - (b)[0]) >= 4 || abs((a)[1] - (b)[1]) >= 4)
  if ((refP0 == refQ0 && refP1 == refQ1) || (refP0 == refQ1 && refP1 == refQ0)) {
    if (refP0 != refP1) {
      if (refP0 == refQ0) return (DB_FAR(mvq[0], mvp[0]) || DB_FAR(mvq[1], mvp[1])) ? 1 : 0;
      return (DB_FAR(mvq[1], mvp[0]) || DB_FAR(mvq[0], mvp[1])) ? 1 : 0;
    }
    return ((DB_FAR(mvq[0], mvp[0]) || DB_FAR(mvq[1], mvp[1])) && (DB_FAR(mvq[1], mvp[0]) || DB_FAR(mvq[0], mvp[1]))) ? 1 : 0;
  }
#undef DB_FAR
  return 1;
}
/* one 4-line luma segment: src = q0 of line 0, xs = step across the edge, ys = step along it (:462-520, :83-153) */
static void db_luma_segment(orc_pixel *src, int xs, int ys, int beta, int tc)
{
  int b[4][8];
  for (int i = 0; i < 4; ++i)
    for (int k = -4; k < 4; ++k) b[i][k + 4] = src[k * xs + i * ys];
  const int dp0 = abs(b[0][1] - 2 * b[0][2] + b[0][3]), dq0 = abs(b[0][4] - 2 * b[0][5] + b[0][6]);
  const int dp3 = abs(b[3][1] - 2 * b[3][2] + b[3][3]), dq3 = abs(b[3][4] - 2 * b[3][5] + b[3][6]);
  const int dp = dp0 + dp3, dq = dq0 + dq3;
  if (dp + dq >= beta) return;
  const int sw = 2 * (dp0 + dq0) < (beta >> 2) && 2 * (dp3 + dq3) < (beta >> 2) &&
                 abs(b[0][3] - b[0][4]) < ((5 * tc + 1) >> 1) && abs(b[3][3] - b[3][4]) < ((5 * tc + 1) >> 1) &&
                 abs(b[0][0] - b[0][3]) + abs(b[0][4] - b[0][7]) < (beta >> 3) && abs(b[3][0] - b[3][3]) + abs(b[3][4] - b[3][7]) < (beta >> 3);
  const int side = (beta + (beta >> 1)) >> 3;
  for (int i = 0; i < 4; ++i) {
    const int *m = b[i];
    int o[8];
    for (int k = 0; k < 8; ++k) o[k] = m[k];
    if (sw) {
      o[1] = db_clip(m[1] - 2 * tc, m[1] + 2 * tc, (2 * m[0] + 3 * m[1] + m[2] + m[3] + m[4] + 4) >> 3);
      o[2] = db_clip(m[2] - 2 * tc, m[2] + 2 * tc, (m[1] + m[2] + m[3] + m[4] + 2) >> 2);
      o[3] = db_clip(m[3] - 2 * tc, m[3] + 2 * tc, (m[1] + 2 * m[2] + 2 * m[3] + 2 * m[4] + m[5] + 4) >> 3);
      o[4] = db_clip(m[4] - 2 * tc, m[4] + 2 * tc, (m[2] + 2 * m[3] + 2 * m[4] + 2 * m[5] + m[6] + 4) >> 3);
      o[5] = db_clip(m[5] - 2 * tc, m[5] + 2 * tc, (m[3] + m[4] + m[5] + m[6] + 2) >> 2);
      o[6] = db_clip(m[6] - 2 * tc, m[6] + 2 * tc, (m[3] + m[4] + m[5] + 3 * m[6] + 2 * m[7] + 4) >> 3);
    } else {
      int delta = (9 * (m[4] - m[3]) - 3 * (m[5] - m[2]) + 8) >> 4;
      if (abs(delta) < tc * 10) {
        const int tc2 = tc >> 1;
        delta = db_clip(-tc, tc, delta);
        o[3] = db_clip(0, 255, m[3] + delta);
        o[4] = db_clip(0, 255, m[4] - delta);
        if (dp < side) o[2] = db_clip(0, 255, m[2] + db_clip(-tc2, tc2, (((m[1] + m[3] + 1) >> 1) - m[2] + delta) >> 1));
        if (dq < side) o[5] = db_clip(0, 255, m[5] + db_clip(-tc2, tc2, (((m[6] + m[4] + 1) >> 1) - m[5] - delta) >> 1));
      }
    }
    /* the strong filter's clip bounds can leave 0..255 only on paper: m +- 2 tc brackets a mean of pixels */
    for (int k = 1; k < 7; ++k) src[(k - 4) * xs + i * ys] = (orc_pixel)o[k];
  }
}
/* kvz_filter_deblock_chroma (:158-180) on the 4 lines of a segment */
static void db_chroma_segment(orc_pixel *src, int xs, int ys, int tc)
{
  for (int i = 0; i < 4; ++i) {
    orc_pixel *s = src + i * ys;
    const int m2 = s[-2 * xs], m3 = s[-xs], m4 = s[0], m5 = s[xs];
    const int delta = db_clip(-tc, tc, (((m4 - m3) * 4) + m2 - m5 + 4) >> 3);
    s[-xs] = (orc_pixel)db_clip(0, 255, m3 + delta);
    s[0] = (orc_pixel)db_clip(0, 255, m4 - delta);
  }
}

void orc_deblock_frame(orc_pixel *y, int stride_y, orc_pixel *u, orc_pixel *v, int stride_c, int width, int height,
                       const orc_cu_info *cus, const orc_deblock_params *prm)
{
  const db_ctx c = { cus, (width + 3) >> 2, width, height, prm };
  for (int dir = 0; dir < 2; ++dir)
    for (int uy = 0; uy < height; uy += 8)
      for (int ux = 0; ux < width; ux += 8) {
        if ((dir == 0 && ux == 0) || (dir == 1 && uy == 0)) continue;            /* filter_deblock_unit, :635-636 */
        /* the second half of a horizontal edge at the right border of an LCU is filtered with the next LCU
         * (filter_deblock_lcu_rightmost, :711-731), its boundary flags and QP taken at that half's own SCU */
        const int deferred = dir == 1 && (ux + 8) % 64 == 0 && ux + 8 != width;
        for (int s = 0; s < 2; ++s) {
          const int fx = (deferred && s == 1) ? ux + 4 : ux;
          int tu_b;
          if (!db_boundary(&c, fx, uy, dir, &tu_b)) continue;
          const int qp = db_qp(&c, fx, uy, dir);
          const int sx = dir ? ux + 4 * s : ux, sy = dir ? uy : uy + 4 * s;
          const orc_cu_info *cp = dir ? db_cu(&c, sx, sy - 1) : db_cu(&c, sx - 1, sy), *cq = db_cu(&c, sx, sy);
          const int bs = db_strength(&c, cp, cq, tu_b);
          if (!bs) continue;
          const int beta = db_beta_table[db_clip(0, 51, qp + (prm->beta_offset_div2 << 1))];
          const int tc = db_tc_table[db_clip(0, 53, qp + 2 * (bs - 1) + (prm->tc_offset_div2 << 1))];
          db_luma_segment(y + (size_t)sy * stride_y + sx, dir ? stride_y : 1, dir ? 1 : stride_y, beta, tc);
        }
        /* chroma: edges on the 8x8 chroma grid, only next to intra CUs (:554-615); one 4-pixel segment per unit */
        if (prm->chroma && ((dir ? uy : ux) & 15) == 0) {
          int tu_b;
          if (!db_boundary(&c, ux, uy, dir, &tu_b)) continue;
          const orc_cu_info *cp = dir ? db_cu(&c, ux, uy - 2) : db_cu(&c, ux - 2, uy), *cq = db_cu(&c, ux, uy);
          if (cq->type != 1 && cp->type != 1) continue;
          const int qpc = db_chroma_scale[db_clip(0, 57, db_qp(&c, ux, uy, dir))];
          const int tc = db_tc_table[db_clip(0, 53, qpc + 2 + (prm->tc_offset_div2 << 1))];
          const int xc = ux >> 1, yc = uy >> 1;
          db_chroma_segment(u + (size_t)yc * stride_c + xc, dir ? stride_c : 1, dir ? 1 : stride_c, tc);
          db_chroma_segment(v + (size_t)yc * stride_c + xc, dir ? stride_c : 1, dir ? 1 : stride_c, tc);
        }
      }
}

/* ---- whole-launch checks (tests/test_gpu_fullsize.py): the per-block functions above over `count` contiguous blocks.
 * Pure loops, so that a test can hand disjoint ranges to several host threads. ---- */
void orc_cost_nxn_many(int satd, int n, const orc_pixel *b1, const orc_pixel *b2, size_t count, unsigned *costs)
{
  const size_t bs = (size_t)n * n;
  for (size_t i = 0; i < count; ++i)
    costs[i] = satd ? orc_satd_nxn(n, b1 + i * bs, b2 + i * bs) : orc_sad_nxn(n, b1 + i * bs, b2 + i * bs);
}
void orc_transform_many(int kind, int n, const int16_t *in, int16_t *out, size_t count)
{
  const size_t bs = (size_t)n * n;
  for (size_t i = 0; i < count; ++i) orc_transform(kind, n, in + i * bs, out + i * bs);
}
void orc_quantize_residual_many(const orc_quant_params *p, int cu_is_intra, int width, int color, int scan_order, int use_trskip,
                                const orc_pixel *ref_in, const orc_pixel *pred_in, orc_pixel *rec_out, orc_coeff *coeff_out,
                                int32_t *has_coeffs, size_t count)
{
  const size_t bs = (size_t)width * width;
  for (size_t i = 0; i < count; ++i)
    has_coeffs[i] = orc_quantize_residual(p, cu_is_intra, width, color, scan_order, use_trskip, width, width,
                                          ref_in + i * bs, pred_in + i * bs, rec_out + i * bs, coeff_out + i * bs);
}

/* =====================================================================
 * AMVP / merge candidate derivation (inter.c:546-1446), flattened state
 * ===================================================================== */

/* a neighbour counts when its CU is an inter CU (inter.c:822-870: type == CU_INTER); its unused list reads mv 0, ref 255
 * (inter_clear_cu_unused, inter.c:546-555) */
typedef struct { int ok; int dir; int mv[2][2]; int ref[2]; } cand_view;

static cand_view cand_at(const orc_cu_info *map, int stride, int x, int y)
{
  cand_view v;
  memset(&v, 0, sizeof(v));
  const orc_cu_info *c = &map[(y >> 2) * stride + (x >> 2)];
  if (c->type != 2) return v;
  v.ok = 1; v.dir = c->mv_dir;
  for (int l = 0; l < 2; ++l) {
    const int used = (c->mv_dir >> l) & 1;
    v.mv[l][0] = used ? c->mv[l][0] : 0;
    v.mv[l][1] = used ? c->mv[l][1] : 0;
    v.ref[l] = used ? c->mv_ref[l] : 255;
  }
  return v;
}

/* is_a0_cand_coded / is_b0_cand_coded (inter.c:566-705) from first principles: the neighbour's 4x4 unit precedes, in the
 * LCU's coding order, the aligned square at the PU's lower-left (A0) / upper-right (B0) corner whose side is the largest
 * power of two dividing both PU dimensions; of the other LCUs those that precede this one in raster order are coded. */
static int corner_unit_coded(int nx, int ny, int sx, int sy)
{
  /* another LCU: LCUs are coded in raster order (the LCU row above, or the same row further left, came first) */
  if ((nx >> 6) != (sx >> 6) || (ny >> 6) != (sy >> 6)) return (ny >> 6) < (sy >> 6) || ((ny >> 6) == (sy >> 6) && (nx >> 6) < (sx >> 6));
  return intra_unit_order((unsigned)(nx & 63) >> 2, (unsigned)(ny & 63) >> 2) < intra_unit_order((unsigned)(sx & 63) >> 2, (unsigned)(sy & 63) >> 2);
}

typedef struct { cand_view a[2], b[3], tmp; } cand_set;    /* a0 a1 / b0 b1 b2 / the temporal one (H, else C3) */

/* Where get_spatial_merge_candidates (inter.c:799-875) looks: pos[0..4] = A0 A1 B0 B1 B2 as picture coordinates, present[i] = the
 * geometry (picture and LCU borders, coding order) lets that neighbour be used at all. */
static void spatial_positions(int x, int y, int w, int h, int pic_w, int pic_h, int pos[5][2], int present[5])
{
  const int side = ((w & -w) < (h & -h)) ? (w & -w) : (h & -h);
  const int xl = x & 63, yl = y & 63;
  memset(present, 0, 5 * sizeof(int));
  pos[0][0] = x - 1; pos[0][1] = y + h;          /* A0 */
  pos[1][0] = x - 1; pos[1][1] = y + h - 1;      /* A1 */
  pos[2][0] = x + w; pos[2][1] = y - 1;          /* B0 */
  pos[3][0] = x + w - 1; pos[3][1] = y - 1;      /* B1 */
  pos[4][0] = x - 1; pos[4][1] = y - 1;          /* B2 */
  (void)pic_w;
  if (x != 0) {
    present[1] = 1;
    present[0] = yl + h < 64 && y + h < pic_h && corner_unit_coded(x - 1, y + h, x, y + h - side);
  }
  if (y != 0) {
    present[2] = x + w < pic_w && (xl + w < 64 || yl == 0) && corner_unit_coded(x + w, y - 1, x + w - side, y);
    present[3] = 1;
    present[4] = x != 0;
  }
}

/* get_spatial_merge_candidates (inter.c:799-875) */
static void spatial_cands(const orc_cu_info *cus, const orc_inter_params *p, int x, int y, int w, int h, cand_set *s)
{
  int pos[5][2], present[5];
  spatial_positions(x, y, w, h, p->pic_width, p->pic_height, pos, present);
  memset(s, 0, sizeof(*s));
  if (present[0]) s->a[0] = cand_at(cus, p->cus_stride, pos[0][0], pos[0][1]);
  if (present[1]) s->a[1] = cand_at(cus, p->cus_stride, pos[1][0], pos[1][1]);
  if (present[2]) s->b[0] = cand_at(cus, p->cus_stride, pos[2][0], pos[2][1]);
  if (present[3]) s->b[1] = cand_at(cus, p->cus_stride, pos[3][0], pos[3][1]);
  if (present[4]) s->b[2] = cand_at(cus, p->cus_stride, pos[4][0], pos[4][1]);
}

/* The reference's unit test of these helpers (tests/mv_cand_tests.c:26-260) in the oracle's terms: the coding-order
 * tests alone (is_a0_cand_coded / is_b0_cand_coded, inter.c:566-705), and -- for an LCU whose CUs are all inter -- the
 * entries of lcu_t.cu (cu.h:324-344: 17 per row, (0, 0) at index 18, the top-right neighbour at 289) that
 * get_spatial_merge_candidates picks: out[0..4] = b0 b1 b2 a0 a1, -1 = none. */
int orc_is_a0_cand_coded(int x, int y, int width, int height)
{
  const int side = ((width & -width) < (height & -height)) ? (width & -width) : (height & -height);
  return corner_unit_coded(x - 1, y + height, x, y + height - side);
}
int orc_is_b0_cand_coded(int x, int y, int width, int height)
{
  const int side = ((width & -width) < (height & -height)) ? (width & -width) : (height & -height);
  return corner_unit_coded(x + width, y - 1, x + width - side, y);
}
void orc_spatial_merge_candidate_indices(int x, int y, int width, int height, int pic_w, int pic_h, int *out)
{
  int pos[5][2], present[5];
  spatial_positions(x, y, width, height, pic_w, pic_h, pos, present);
  static const int order[5] = { 2, 3, 4, 0, 1 };
  const int lx0 = x & ~63, ly0 = y & ~63;
  for (int i = 0; i < 5; ++i) {
    const int k = order[i];
    if (!present[k]) { out[i] = -1; continue; }
    const int sx = (pos[k][0] - lx0) >> 2, sy = (pos[k][1] - ly0) >> 2;      /* -1 .. 16; arithmetic shift of -1 .. -4 gives -1 */
    out[i] = (sx == 16 && sy == -1) ? 289 : 18 + sx + sy * 17;
  }
}

/* get_temporal_merge_candidates (inter.c:713-780) with ref_list 1, ref_idx 0 as every caller passes: H below-right of the
 * PU unless that starts a new LCU row, else / otherwise C3 at the centre, both on the 16x16 grid of the collocated picture */
static void temporal_cand(const orc_cu_info *col, const orc_inter_params *p, int x, int y, int w, int h, cand_set *s)
{
  memset(&s->tmp, 0, sizeof(s->tmp));
  if (!p->num_refs || p->ref_LX_size[0] == 0 || !col) return;
  const int bx = x + w, by = y + h, cx = x + w / 2, cy = y + h / 2;
  cand_view hh, c3;
  memset(&hh, 0, sizeof(hh)); memset(&c3, 0, sizeof(c3));
  if (bx < p->in_width && by < p->in_height && (by & 63) != 0) hh = cand_at(col, p->col_stride, bx & ~15, by & ~15);
  if (cx < p->in_width && cy < p->in_height) c3 = cand_at(col, p->col_stride, cx & ~15, cy & ~15);
  s->tmp = hh.ok ? hh : c3;
}

/* apply_mv_scaling_pocs + get_scaled_mv (inter.c:955-980) */
static void scale_mv(int cur_poc, int cur_ref_poc, int nb_poc, int nb_ref_poc, int mv[2])
{
  int dc = cur_poc - cur_ref_poc, dn = nb_poc - nb_ref_poc;
  if (dc == dn || dn == 0) return;                      /* dn == 0: the reference divides by zero (a picture referencing itself) */
  dc = dc < -128 ? -128 : dc > 127 ? 127 : dc;
  dn = dn < -128 ? -128 : dn > 127 ? 127 : dn;
  int scale = (dc * ((0x4000 + (abs(dn) >> 1)) / dn) + 32) >> 6;
  scale = scale < -4096 ? -4096 : scale > 4095 ? 4095 : scale;
  for (int k = 0; k < 2; ++k) {
    const int prod = scale * (int16_t)mv[k];
    const int v = (prod + 127 + (prod < 0)) >> 8;
    mv[k] = v < -32768 ? -32768 : v > 32767 ? 32767 : v;
  }
}

/* add_temporal_candidate (inter.c:1011-1061): the collocated PU's vector of the list that points away from the current
 * picture's future (L1 as soon as any reference lies in the future), else its other list, scaled by the POC distances */
static int temporal_mv(const orc_inter_params *p, const cand_view *c, int cur_ref, int reflist, int out[2])
{
  if (!c->ok || p->ref_LX_size[0] == 0) return 0;
  const int col_pic = p->ref_LX[0][0];
  int l = reflist;
  for (int i = 0; i < p->num_refs; ++i) if (p->ref_pocs[i] > p->poc) { l = 1; break; }
  if (!(c->dir & (l + 1))) l = 1 - l;
  out[0] = c->mv[l][0]; out[1] = c->mv[l][1];
  scale_mv(p->poc, p->ref_pocs[cur_ref & 15], p->ref_pocs[col_pic & 15], p->col_ref_pocs[p->col_ref_LX[l][c->ref[l] & 15] & 15], out);
  return 1;
}

/* add_mvp_candidate (inter.c:1063-1098): list `reflist` of the neighbour first, then the other one */
static int mvp_from(const orc_inter_params *p, const cand_view *c, int reflist, int cur_pic, int scaling, int out[2])
{
  if (!c->ok) return 0;
  for (int i = 0; i < 2; ++i) {
    const int l = i == 0 ? reflist : !reflist;
    if (!(c->dir & (1 << l))) continue;
    const int nb_pic = p->ref_LX[l][c->ref[l] & 15];
    if (scaling) {
      out[0] = c->mv[l][0]; out[1] = c->mv[l][1];
      scale_mv(p->poc, p->ref_pocs[cur_pic & 15], p->poc, p->ref_pocs[nb_pic & 15], out);
      return 1;
    }
    if (nb_pic == cur_pic) { out[0] = c->mv[l][0]; out[1] = c->mv[l][1]; return 1; }
  }
  return 0;
}

void orc_inter_get_mv_cand(const orc_cu_info *cus, const orc_cu_info *col_cus, const orc_inter_params *p,
                           int x, int y, int width, int height, int reflist, int lx_idx, int16_t mv_cand[2][2])
{
  cand_set s;
  spatial_cands(cus, p, x, y, width, height, &s);
  temporal_cand(col_cus, p, x, y, width, height, &s);
  const int cur_pic = p->ref_LX[reflist][lx_idx];
  int mv[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } }, n = 0;
  /* get_mv_cand_from_candidates (inter.c:1102-1195).  Left: the first of A0, A1 that points at the same picture, else
   * the first that has a vector at all, scaled */
  for (int sc = 0; sc < 2 && n == 0; ++sc)
    for (int i = 0; i < 2; ++i) if (mvp_from(p, &s.a[i], reflist, cur_pic, sc, mv[n])) { ++n; break; }
  /* above: the first of B0, B1, B2 pointing at the same picture ... */
  int above = 0;
  for (int i = 0; i < 3; ++i) if (mvp_from(p, &s.b[i], reflist, cur_pic, 0, mv[n])) { above = 1; break; }
  n += above;
  /* ... and a scaled one only when there is no left neighbour at all and the list is still short */
  if (s.a[0].ok || s.a[1].ok) above = 1; else if (n != 2) above = 0;
  if (!above)
    for (int i = 0; i < 3; ++i) if (mvp_from(p, &s.b[i], reflist, cur_pic, 1, mv[n])) { ++n; break; }
  if (n == 2 && mv[0][0] == mv[1][0] && mv[0][1] == mv[1][1]) n = 1;
  if (p->tmvp_enable && p->poc > 1 && p->num_refs && n < 2 && s.tmp.ok && temporal_mv(p, &s.tmp, cur_pic, reflist, mv[n])) ++n;
  for (; n < 2; ++n) mv[n][0] = mv[n][1] = 0;
  for (int i = 0; i < 2; ++i) { mv_cand[i][0] = (int16_t)mv[i][0]; mv_cand[i][1] = (int16_t)mv[i][1]; }
}

/* is_duplicate_candidate (inter.c:1262-1278) */
static int same_motion(const cand_view *a, const cand_view *b)
{
  if (!b->ok || a->dir != b->dir) return 0;
  for (int l = 0; l < 2; ++l)
    if ((a->dir >> l) & 1)
      if (a->mv[l][0] != b->mv[l][0] || a->mv[l][1] != b->mv[l][1] || a->ref[l] != b->ref[l]) return 0;
  return 1;
}

static int merge_push(const cand_view *c, const cand_view *d1, const cand_view *d2, orc_merge_cand *o)
{
  if (!c->ok || (d1 && same_motion(c, d1)) || (d2 && same_motion(c, d2))) return 0;
  for (int l = 0; l < 2; ++l) { o->mv[l][0] = (int16_t)c->mv[l][0]; o->mv[l][1] = (int16_t)c->mv[l][1]; o->ref[l] = (uint8_t)c->ref[l]; }
  o->dir = (uint8_t)c->dir;
  return 1;
}

int orc_inter_get_merge_cand(const orc_cu_info *cus, const orc_cu_info *col_cus, const orc_inter_params *p,
                             int x, int y, int width, int height, int use_a1, int use_b1, orc_merge_cand out[5])
{
  cand_set s;
  int n = 0;
  memset(out, 0, 5 * sizeof(out[0]));
  spatial_cands(cus, p, x, y, width, height, &s);
  if (!use_a1) s.a[1].ok = 0;
  if (!use_b1) s.b[1].ok = 0;
  /* inter.c:1338-1343: A1, B1, B0, A0, B2 with the pairwise duplicate checks of the standard */
  n += merge_push(&s.a[1], NULL, NULL, &out[n]);
  n += merge_push(&s.b[1], &s.a[1], NULL, &out[n]);
  n += merge_push(&s.b[0], &s.b[1], NULL, &out[n]);
  n += merge_push(&s.a[0], &s.a[1], NULL, &out[n]);
  if (n < 4) n += merge_push(&s.b[2], &s.a[1], &s.b[1], &out[n]);
  /* the temporal candidate always points at index 0 of a list (inter.c:1345-1375) */
  if (p->tmvp_enable && n < 5 && p->num_refs) {
    temporal_cand(col_cus, p, x, y, width, height, &s);
    out[n].dir = 0;
    for (int l = 0; l <= (p->slice_is_b ? 1 : 0); ++l) {
      int mv[2];
      if (temporal_mv(p, &s.tmp, p->ref_LX[l][0], l, mv)) {
        out[n].mv[l][0] = (int16_t)mv[0]; out[n].mv[l][1] = (int16_t)mv[1];
        out[n].ref[l] = 0;
        out[n].dir |= (uint8_t)(1 << l);
      }
    }
    if (out[n].dir) ++n;
  }
  /* B slices: pairs of an L0 and an L1 motion of the candidates so far (inter.c:1377-1413) */
  if (n < 5 && p->slice_is_b) {
    static const uint8_t first[12] = { 0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3 }, second[12] = { 1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2 };
    const int cutoff = n;
    for (int k = 0; k < cutoff * (cutoff - 1) && n != 5; ++k) {
      const int i = first[k], j = second[k];
      if (i >= n || j >= n) break;
      if (!(out[i].dir & 1) || !(out[j].dir & 2)) continue;
      out[n].dir = 3;
      out[n].mv[0][0] = out[i].mv[0][0]; out[n].mv[0][1] = out[i].mv[0][1];
      out[n].mv[1][0] = out[j].mv[1][0]; out[n].mv[1][1] = out[j].mv[1][1];
      out[n].ref[0] = out[i].ref[0]; out[n].ref[1] = out[j].ref[1];
      const int same = p->ref_LX[0][out[i].ref[0] & 15] == p->ref_LX[1][out[j].ref[1] & 15] &&
                       out[i].mv[0][0] == out[j].mv[1][0] && out[i].mv[0][1] == out[j].mv[1][1];
      if (!same) ++n;
    }
  }
  /* zero vectors over the reference indices (inter.c:1415-1443) */
  int num_ref = p->num_refs;
  if (n < 5 && p->slice_is_b) {
    int before = 0, after = 0;
    for (int j = 0; j < p->num_refs; ++j) { if (p->ref_pocs[j] < p->poc) ++before; else ++after; }
    num_ref = before < after ? before : after;
  }
  for (int zero_idx = 0; n != 5; ++zero_idx, ++n) {
    out[n].mv[0][0] = out[n].mv[0][1] = 0;
    out[n].ref[0] = (uint8_t)(zero_idx >= num_ref - 1 ? 0 : zero_idx);
    out[n].ref[1] = out[n].ref[0];
    out[n].dir = 1;
    if (p->slice_is_b) { out[n].mv[1][0] = out[n].mv[1][1] = 0; out[n].dir = 3; }
  }
  return n;
}

void orc_inter_candidates(const orc_cu_info *cus, const orc_cu_info *col_cus, const orc_cu_info *ref_cus, const orc_inter_params *p,
                          orc_me_pu *pus, size_t count, orc_merge_cand *merge_out)
{
  /* which list holds picture ref_idx, and where (search_pu_inter_ref, search_inter.c:1143-1166) */
  int reflist = -1, lx = 0;
  const int lx_max = p->ref_LX_size[0] > p->ref_LX_size[1] ? p->ref_LX_size[0] : p->ref_LX_size[1];
  for (lx = 0; lx < lx_max; ++lx) {
    if (lx < p->ref_LX_size[0] && p->ref_LX[0][lx] == p->ref_idx) { reflist = 0; break; }
    if (lx < p->ref_LX_size[1] && p->ref_LX[1][lx] == p->ref_idx) { reflist = 1; break; }
  }
  for (size_t i = 0; i < count; ++i) {
    orc_me_pu *u = &pus[i];
    orc_merge_cand mc[5];
    const int tx = u->x - p->tile_x, ty = u->y - p->tile_y;     /* descriptors carry picture coordinates, the derivation works in the tile's */
    const int n = orc_inter_get_merge_cand(cus, col_cus, p, tx, ty, u->width, u->height, !(u->pad & 1), !(u->pad & 2), mc);
    u->num_merge_cand = (int16_t)n;
    for (int k = 0; k < 5; ++k) {
      memset(&u->merge[k], 0, sizeof(u->merge[k]));
      if (k >= n) continue;
      u->merge[k].usable = mc[k].dir != 3;
      if (mc[k].dir != 3) {
        const int l = mc[k].dir - 1;
        u->merge[k].mv[0] = mc[k].mv[l][0]; u->merge[k].mv[1] = mc[k].mv[l][1];
        u->merge[k].same_ref = p->ref_LX[l][mc[k].ref[l] & 15] == p->ref_idx;
      }
    }
    if (merge_out) memcpy(merge_out + 5 * i, mc, sizeof(mc));
    memset(u->mv_cand, 0, sizeof(u->mv_cand));
    if (reflist >= 0) orc_inter_get_mv_cand(cus, col_cus, p, tx, ty, u->width, u->height, reflist, lx, u->mv_cand);
    /* the collocated CU's vector as one more start point (search_inter.c:1190-1206) */
    u->extra_mv[0] = u->extra_mv[1] = 0;
    if (ref_cus) {
      const cand_view c = cand_at(ref_cus, p->col_stride, u->x + (u->width >> 1), u->y + (u->height >> 1));
      if (c.ok) { const int l = (c.dir & 1) ? 0 : 1; u->extra_mv[0] = (int16_t)c.mv[l][0]; u->extra_mv[1] = (int16_t)c.mv[l][1]; }
    }
  }
}
