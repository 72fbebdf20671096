// frac_core.h -- the fused fractional motion search shared by ipol.hip (kvz_hip_search_frac_batch) and
// me.hip (kvz_hip_search_pu_batch).  Device code only.
#pragma once
#include "kvz_hip_internal.h"
#include "satd_regs.h"

namespace kvzhip {

static __constant__ signed char c_luma_filter[4][8] = {       // filter.c:54-60
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };

struct refplane_t { const u8 *p; u32 stride; int w, h; };

__device__ __forceinline__ u8 ref_px(const refplane_t &r, int x, int y)
{
  return r.p[(size_t)clampi(y, 0, r.h - 1) * r.stride + clampi(x, 0, r.w - 1)];
}

// (int16 sample + 32) >> 6 through the int16-argument clip (ipol-generic.c:285-287 etc.)
__device__ __forceinline__ u8 round_clip16(i16 sample) { return fast_clip16((i16)(((int)sample + 32) >> 6)); }

// ---------------------------------------------------------------------------
// Fused fractional motion search (search_inter.c:965-1128 without MV bit
// costs).  For the integer position P(0,0) = ref(x2, y2):
//   H_f(r, c) = sum_i f[i] * P[r][c-3+i]                 (int16, never overflows)
//   S(fx, fy; r, c) = round_clip16((int16)(sum_j f_fy[j] * H_fx(r-3+j, c) >> 6))
// Every block the four reference filter steps produce (ipol-generic.c:192-658)
// is S at a quarter-pel offset (qx, qy): fx = qx & 3, fy = qy & 3, r = y + (qy >> 2),
// c = x + (qx >> 2) -- including the int16 truncation of the vertical sum and the
// cases where the reference skips a pass (a pass with taps {0,0,0,64,0,0,0,0} is
// exact).  Candidates are filtered into LDS and scored with the 8x8 Hadamard
// SATD by a quad of lanes per (candidate, 8x8 sub-block).
// ---------------------------------------------------------------------------
#define FR_HS 65                 /* H plane row stride of the per-call filter step kernel */

struct frac_cand { int fx, fy, ry, cx; };
// square[] of search_inter.c:972-976
static __constant__ signed char c_sq_x[9] = { 0, -1, 1, 0, 0, -1, 1, -1, 1 };
static __constant__ signed char c_sq_y[9] = { 0, 0, 0, -1, 1, -1, -1, 1, 1 };

// LDS geometry of one block's working set.  BIG: blocks up to 64x64, the whole 256-thread workgroup
// cooperates (barriers).  SMALL: blocks up to 16x16, ONE WAVE per block, four blocks per workgroup,
// wave-private LDS slices and no barrier (DS operations of a wave execute in order).
template <int MAXW>
struct frac_geom {
  static constexpr int PS = MAXW + 8;                 // P window stride: cols -4 .. w+3
  static constexpr int PR = MAXW + 8;                 // P rows -4 .. h+3
  static constexpr int HS = MAXW + 1;                 // H plane stride: cols -1 .. w-1
  static constexpr int CS = MAXW;                     // cur / candidate stride
  static constexpr int P_BYTES = PS * PR, CUR_BYTES = CS * MAXW, H_ELEMS = PR * HS, CAND_BYTES = CS * MAXW;
  static constexpr int TOTAL = ((P_BYTES + CUR_BYTES + 4 * CAND_BYTES + 15) & ~15) + 3 * H_ELEMS * 2 + 32;
};

// MV cost policy of the search: cost() = calc_mvd_cost (search_inter.c:373-412) of the vector (x, y) << shift,
// within() = fracmv_within_tile (:87-176) of a quarter-pel vector.  frac_no_cost gives the bare SATD search.
struct frac_no_cost {
  __device__ __forceinline__ u32 cost(int, int, int, u32 &bits) const { bits = 0; return 0; }
  __device__ __forceinline__ bool within(int, int) const { return true; }
};
struct frac_result { int mvx, mvy; u32 cost, bitcost; };   // info->best_mv (quarter-pel), best_cost, best_bitcost

// d: (x1, y1) block in pic, (x2, y2) its integer-pel position in ref.  fme_level = cfg.fme_level (number of filter
// steps, 0..4).  out (17 raw SATD costs) and best (hpel / qpel indices) may be null.
// FW, FH: the block size when it is known at compile time (0 = read it from d): the many index divisions of the
// staging and filter loops then reduce to shifts, which is what bounds the latency of the one-wave-per-block kernels.
template <int MAXW, int T, bool WAVE, class MVC, int FW = 0, int FH = 0>
__device__ __forceinline__ frac_result search_frac_core(int tid, u8 *lds, const u8 *__restrict__ pic, u32 pic_stride, const refplane_t &ref,
                                                        const kvz_hip_block_pair &d, int fme_level, const MVC &mvc,
                                                        u32 *__restrict__ out, i32 *__restrict__ best)
{
  typedef frac_geom<MAXW> G;
  u8 *s_p = lds, *s_cur = s_p + G::P_BYTES, *s_cand = s_cur + G::CUR_BYTES;
  i16 *s_h = (i16 *)(lds + ((G::P_BYTES + G::CUR_BYTES + 4 * G::CAND_BYTES + 15) & ~15));
  u32 *s_cost = (u32 *)(s_h + 3 * G::H_ELEMS);
  int *s_sel = (int *)(s_cost + 4);
  auto sync = [&]() { if (WAVE) wave_lds_fence(); else __syncthreads(); };

  const int w = FW ? FW : d.width, h = FH ? FH : d.height;
  const int pw = w + 8, ph = h + 8;
  for (int i = tid; i < pw * ph; i += T) {
    const int y = i / pw, x = i - y * pw;
    s_p[y * G::PS + x] = ref_px(ref, d.x2 - 4 + x, d.y2 - 4 + y);
  }
  for (int i = tid; i < w * h; i += T) {
    const int y = i / w, x = i - y * w;
    s_cur[y * G::CS + x] = pic[(size_t)(d.y1 + y) * pic_stride + d.x1 + x];
  }
  sync();

  // H plane for filter f: rows r = -4 .. h+3 (index r+4), cols c = -1 .. w-1 (index c+1)
  auto hor_plane = [&](int f, i16 *dst) {
    const signed char *fl = c_luma_filter[f];
    for (int i = tid; i < ph * (w + 1); i += T) {
      const int y = i / (w + 1), x = i - y * (w + 1);     // c = x - 1 -> P cols c-3 .. c+4 -> window index x .. x+7
      int acc = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += fl[t] * (int)s_p[y * G::PS + x + t];
      dst[y * G::HS + x] = (i16)acc;
    }
  };
  // candidate: S(fx, fy; y + ry, x + cx) for the whole block
  auto filter_cand = [&](const frac_cand &c, int plane, u8 *dst) {
    const signed char *vf = c_luma_filter[c.fy];
    for (int i = tid; i < w * h; i += T) {
      const int y = i / w, x = i - y * w;
      const int r = y + c.ry, cc = x + c.cx;              // H row index of (r-3+j) is r+1+j, col index cc+1
      int acc = 0;
      if (plane < 0) {                                    // fx == 0: H_0 = 64 * P
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += vf[j] * 64 * (int)s_p[(r + 1 + j) * G::PS + cc + 4];
      } else {
        const i16 *pl = s_h + plane * G::H_ELEMS;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += vf[j] * (int)pl[(r + 1 + j) * G::HS + cc + 1];
      }
      dst[y * G::CS + x] = round_clip16((i16)(acc >> 6));
    }
  };
  // SATD of candidates 0..ncand-1 against s_cur -> s_cost.  Four lanes per (candidate, 8x8 sub-block): lane p takes rows
  // 2p, 2p+1 and the last two vertical Hadamard stages cross the quad with DPP (satd8_quad_part), a quarter of the
  // instructions of one lane per sub-block -- the scoring rounds are the largest part of a small block's search.
  auto score = [&](int ncand, const u8 *cand0, int cand_stride, int cand_pitch) {
    if (tid < 4) s_cost[tid] = 0;
    sync();
    const int w8 = w >> 3, n8 = w8 * (h >> 3);
    const int p = tid & 3;
    const short sg1 = (p & 1) ? (short)-1 : (short)1, sg2 = (p & 2) ? (short)-1 : (short)1;
    const v2s m1 = { sg1, sg1 }, m2 = { sg2, sg2 };
    for (int i = tid; i < ncand * n8 * 4; i += T) {        // whole quads are active or idle together: counts are multiples of 4
      const int q = i >> 2, k = q / n8, sb = q - k * n8, by = sb / w8, bx = sb - by * w8;
      const u8 *a = s_cur + (by * 8 + 2 * p) * G::CS + bx * 8;
      const u8 *b = cand0 + (size_t)k * cand_pitch + (by * 8 + 2 * p) * cand_stride + bx * 8;
      uint4 x, y;
      __builtin_memcpy(&x.x, a, 4); __builtin_memcpy(&x.y, a + 4, 4); __builtin_memcpy(&x.z, a + G::CS, 4); __builtin_memcpy(&x.w, a + G::CS + 4, 4);
      __builtin_memcpy(&y.x, b, 4); __builtin_memcpy(&y.y, b + 4, 4); __builtin_memcpy(&y.z, b + cand_stride, 4); __builtin_memcpy(&y.w, b + cand_stride + 4, 4);
      u32 m = satd8_quad_part(x, y, m1, m2);
      m = group_sum<4>(m);
      if (p == 0) atomicAdd(&s_cost[k], (m + 2) >> 2);
    }
    sync();
  };

  // integer position: candidate = P[y][x]
  score(1, s_p + 4 * G::PS + 4, G::PS, 0);
  int mx = d.x2 - d.x1, my = d.y2 - d.y1;              // pixel precision
  u32 best_bitcost = 0;
  u32 best_cost = s_cost[0];
  if (out && tid == 0) out[0] = best_cost;
  best_cost += mvc.cost(mx, my, 2, best_bitcost);
  mx *= 2; my *= 2;                                    // half-pel precision (search_inter.c:1031-1032)

  hor_plane(2, s_h);
  sync();

  int best_index = 0, pat = 1;                         // pat: first index of the step's 4 positions in square[]
  for (int step = 0; step < fme_level; ++step) {
    frac_cand c[4];
    int plane[4];
    if (step < 2) {
      if (step == 0) {
        c[0] = { 2, 0, 0, -1 }; c[1] = { 2, 0, 0, 0 }; c[2] = { 0, 2, -1, 0 }; c[3] = { 0, 2, 0, 0 };
        plane[0] = 0; plane[1] = 0; plane[2] = -1; plane[3] = -1;
      } else {
        c[0] = { 2, 2, -1, -1 }; c[1] = { 2, 2, -1, 0 }; c[2] = { 2, 2, 0, -1 }; c[3] = { 2, 2, 0, 0 };
        plane[0] = plane[1] = plane[2] = plane[3] = 0;
      }
    } else {
      const int hx = s_sel[0], hy = s_sel[1];              // best half-pel offset in {-1,0,1}^2
      const int bx = 2 * hx, by = 2 * hy;
      const int hp = (bx & 3) ? 0 : -1;                    // plane of the half-pel column itself: fx 2 -> plane 0, fx 0 -> P
      if (step == 2) {
        c[0] = { (bx - 1) & 3, by & 3, by >> 2, (bx - 1) >> 2 };
        c[1] = { (bx + 1) & 3, by & 3, by >> 2, (bx + 1) >> 2 };
        c[2] = { bx & 3, (by - 1) & 3, (by - 1) >> 2, bx >> 2 };
        c[3] = { bx & 3, (by + 1) & 3, (by + 1) >> 2, bx >> 2 };
        plane[0] = 1; plane[1] = 2; plane[2] = hp; plane[3] = hp;
        hor_plane((bx - 1) & 3, s_h + G::H_ELEMS);
        hor_plane((bx + 1) & 3, s_h + 2 * G::H_ELEMS);
        sync();
      } else {
        c[0] = { (bx - 1) & 3, (by - 1) & 3, (by - 1) >> 2, (bx - 1) >> 2 };
        c[1] = { (bx + 1) & 3, (by - 1) & 3, (by - 1) >> 2, (bx + 1) >> 2 };
        c[2] = { (bx - 1) & 3, (by + 1) & 3, (by + 1) >> 2, (bx - 1) >> 2 };
        c[3] = { (bx + 1) & 3, (by + 1) & 3, (by + 1) >> 2, (bx + 1) >> 2 };
        plane[0] = 1; plane[1] = 2; plane[2] = 1; plane[3] = 2;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) filter_cand(c[k], plane[k], s_cand + k * G::CAND_BYTES);
    sync();
    score(4, s_cand, G::CS, G::CAND_BYTES);
    // decision: same order and strict '<' as search_inter.c:1069-1102
    const int mv_shift = step < 2 ? 1 : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32 cj = s_cost[j], bj = 0;
      if (out && tid == 0) out[(step >= 2 ? 8 : 0) + pat + j] = cj;
      const int px = mx + c_sq_x[pat + j], py = my + c_sq_y[pat + j];
      if (!mvc.within(px * (1 << mv_shift), py * (1 << mv_shift))) continue;
      cj += mvc.cost(px, py, mv_shift, bj);
      if (cj < best_cost) { best_cost = cj; best_bitcost = bj; best_index = pat + j; }
    }
    pat += 4;
    if (step == 1 || step == fme_level - 1) {            // search_inter.c:1107-1122
      if (best && tid == 0 && step == 3) best[1] = best_index;
      mx += c_sq_x[best_index]; my += c_sq_y[best_index];
      if (step == (fme_level - 1 < 1 ? fme_level - 1 : 1)) {
        if (best && tid == 0) best[0] = best_index;
        mx *= 2; my *= 2;                                // quarter-pel precision
        sync();
        if (tid == 0) { s_sel[0] = c_sq_x[best_index]; s_sel[1] = c_sq_y[best_index]; }
        best_index = 0; pat = 1;
      }
    }
    sync();
  }
  frac_result res = { mx, my, best_cost, best_bitcost };
  return res;
}

__device__ __forceinline__ bool frac_shape_ok(int w, int h) { return !(w < 8 || h < 8 || w > 64 || h > 64 || ((w | h) & 7)); }

}  // namespace kvzhip
