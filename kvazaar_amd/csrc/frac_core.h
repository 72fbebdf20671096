// frac_core.h -- the fused fractional motion search shared by ipol.hip (kvz_hip_search_frac_batch) and
// me.hip (kvz_hip_search_pu_batch).  Device code only.
#pragma once
#include "kvz_hip_internal.h"
#include "satd_regs.h"

namespace kvzhip {

static __constant__ __attribute__((aligned(8))) signed char c_luma_filter[4][8] = {       // filter.c:54-60
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

// square[] of search_inter.c:972-976
static __constant__ signed char c_sq_x[9] = { 0, -1, 1, 0, 0, -1, 1, -1, 1 };
static __constant__ signed char c_sq_y[9] = { 0, 0, 0, -1, 1, -1, -1, 1, 1 };

// LDS geometry of one block's working set.  BIG: blocks up to 64x64, the whole 256-thread workgroup
// cooperates (barriers).  SMALL: blocks up to 16x16, ONE WAVE per block, four blocks per workgroup,
// wave-private LDS slices and no barrier (DS operations of a wave execute in order).
//
// The horizontal planes are kept TRANSPOSED (column-major): the vertical 8-tap filter then finds the samples it
// multiplies next to each other, two int16 per dword = one operand of v_dot2_i32_i16, and a lane that produces
// two rows x eight columns of a candidate reads each column's ten samples as five dwords instead of sixteen
// 2-byte reads.  Four planes: fx = 0 (64 P), fx = 2 and the two odd filters of the quarter-pel steps.
template <int MAXW>
struct frac_geom {
  static constexpr int PS = MAXW + 8;                 // P window stride: cols -4 .. w+3
  static constexpr int PR = MAXW + 8;                 // P rows -4 .. h+3
  static constexpr int CS = MAXW;                     // cur stride
  static constexpr int HT = MAXW + 10;                // int16 per H column: rows -4 .. h+3 at index r+4; HT/2 is odd, so
                                                      // neighbouring columns start in different LDS banks
  static constexpr int HC = MAXW + 4;                 // H columns: c = -1 .. w-1 at index c+1, filled in groups of 4
  static constexpr int P_BYTES = PS * PR + 16;        // + the dwords the last column group reads past the last row
  static constexpr int CUR_BYTES = CS * MAXW, H_ELEMS = HC * HT;
  static constexpr int H_OFF = (P_BYTES + CUR_BYTES + 15) & ~15;
  static constexpr int TOTAL = H_OFF + 4 * H_ELEMS * 2 + 32;
};

// Coefficient pairs of the vertical filter for v_dot2: a lane owns output rows (y0, y0+1), y0 even, and reads the five
// row pairs that start at the even row b - parity, b = first tap row of y0.  [fy][parity][output row][pair]
struct frac_vcoef { u32 c[4][2][2][5]; };
__host__ __device__ constexpr u32 frac_pack16(int lo, int hi) { return ((u32)lo & 0xffffu) | ((u32)hi << 16); }
__host__ __device__ constexpr frac_vcoef frac_make_vcoef()
{
  // the taps times 4: the vertical sums then carry the reference's (int16)(sum >> 6) in bytes 1..2 of the accumulator,
  // where one v_perm_b32 picks it up for two neighbouring samples at once (4 sum >> 8 == sum >> 6, exactly)
  constexpr int f[4][8] = { { 0, 0, 0, 256, 0, 0, 0, 0 }, { -4, 16, -40, 232, 68, -20, 4, 0 }, { -4, 16, -44, 160, 160, -44, 16, -4 },
                            { 0, 4, -20, 68, 232, -40, 16, -4 } };
  frac_vcoef t = {};
  for (int fy = 0; fy < 4; ++fy)
    for (int i = 0; i < 5; ++i) {
      const u32 aligned = i < 4 ? frac_pack16(f[fy][2 * i], f[fy][2 * i + 1]) : 0u;                        // taps (2i, 2i+1) on pair i
      const u32 shifted = frac_pack16(i > 0 ? f[fy][2 * i - 1] : 0, i < 4 ? f[fy][2 * i] : 0);             // taps (2i-1, 2i) on pair i
      const u32 late = i > 0 ? frac_pack16(f[fy][2 * i - 2], f[fy][2 * i - 1]) : 0u;                       // taps (2i-2, 2i-1) on pair i
      t.c[fy][0][0][i] = aligned; t.c[fy][0][1][i] = shifted;      // b even: row y0 starts on pair 0, row y0+1 half a pair later
      t.c[fy][1][0][i] = shifted; t.c[fy][1][1][i] = late;         // b odd: pairs start one row early
    }
  return t;
}
static __constant__ const frac_vcoef c_frac_vcoef = frac_make_vcoef();

// Two neighbouring samples from their vertical sums (taps x 4, see frac_make_vcoef): the int16 truncation of sum >> 6 is
// the byte pick, then (s + 32) >> 6 and the 0..255 clamp in packed int16 -- the addition saturates, which is where the
// reference's int arithmetic and a wrapping 16-bit add would part (s >= 32736 clips to 255 either way).
__device__ __forceinline__ v2s frac_round_clip_pair(int acc_even, int acc_odd)
{
  const v2s sft = as_v2s(__builtin_amdgcn_perm((u32)acc_odd, (u32)acc_even, 0x06050201u));
  const v2s r32 = { 32, 32 }, six = { 6, 6 }, zero = { 0, 0 }, top = { 255, 255 };
  v2s t = __builtin_elementwise_add_sat(sft, r32) >> six;
  t = __builtin_elementwise_max(t, zero);
  return __builtin_elementwise_min(t, top);
}

// MV cost policy of the search: cost() = calc_mvd_cost (search_inter.c:373-412) of the vector (x, y) << shift,
// within() = fracmv_within_tile (:87-176) of a quarter-pel vector.  frac_no_cost gives the bare SATD search.
struct frac_no_cost {
  __device__ __forceinline__ u32 cost(int, int, int, u32 &bits) const { bits = 0; return 0; }
  __device__ __forceinline__ bool within(int, int) const { return true; }
};
struct frac_result { int mvx, mvy; u32 cost, bitcost; u32 cost0; };   // info->best_mv (quarter-pel), best_cost, best_bitcost; cost0: the integer position's
                                                                   // SATD + MV cost (search_inter.c:1019-1029) = the re-score of :1242-1252

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
  u8 *s_p = lds, *s_cur = s_p + G::P_BYTES;
  u32 *s_h = (u32 *)(lds + G::H_OFF);                  // plane k at dword k * H_ELEMS / 2
  u32 *s_cost = s_h + 4 * (G::H_ELEMS / 2);
  int *s_sel = (int *)(s_cost + 4);
  auto sync = [&]() { if (WAVE) wave_lds_fence(); else __syncthreads(); };

  const int w = FW ? FW : d.width, h = FH ? FH : d.height;
  const int pw = w + 8, ph = h + 8;
  // P window: dword loads when it lies inside the frame, clamped bytes otherwise (ipol-generic.c:731-784)
  if (d.x2 - 4 >= 0 && d.x2 + w + 4 <= ref.w && d.y2 - 4 >= 0 && d.y2 + h + 4 <= ref.h) {
    const int pw4 = pw >> 2;
    const u8 *src = ref.p + (size_t)(d.y2 - 4) * ref.stride + (d.x2 - 4);
    for (int i = tid; i < pw4 * ph; i += T) {
      const int y = i / pw4, x = (i - y * pw4) * 4;
      u32 v;
      __builtin_memcpy(&v, src + (size_t)y * ref.stride + x, 4);
      *(u32 *)(s_p + y * G::PS + x) = v;
    }
  } else {
    for (int i = tid; i < pw * ph; i += T) {
      const int y = i / pw, x = i - y * pw;
      s_p[y * G::PS + x] = ref_px(ref, d.x2 - 4 + x, d.y2 - 4 + y);
    }
  }
  if (w & 4) {                                          // 4, 12: dword segments
    const int w4 = w >> 2;
    for (int i = tid; i < w4 * h; i += T) {
      const int y = i / w4, x = (i - y * w4) * 4;
      u32 v;
      __builtin_memcpy(&v, pic + (size_t)(d.y1 + y) * pic_stride + d.x1 + x, 4);
      *(u32 *)(s_cur + y * G::CS + x) = v;
    }
  } else {
    const int w8 = w >> 3;
    for (int i = tid; i < w8 * h; i += T) {
      const int y = i / w8, x = (i - y * w8) * 8;
      uint2 v;
      __builtin_memcpy(&v, pic + (size_t)(d.y1 + y) * pic_stride + d.x1 + x, 8);
      *(uint2 *)(s_cur + y * G::CS + x) = v;
    }
  }
  sync();

  // Two horizontal planes in one pass over the window.  Work item = 2 rows x 4 columns: three aligned dwords of each row,
  // the byte windows of the four columns cut out with v_alignbyte, each sample two v_dot4_i32_i8 on pixels - 128
  // (sum of the taps = 64, so + 128 * 64 restores the offset; exact in int32, the result fits int16 like the reference's).
  // H(r, c): plane rows r = -4 .. h+3 (index r+4), columns c = -1 .. w-1 (index c+1), stored column-major.
  auto hor_planes = [&](int fa, int slot_a, int fb, int slot_b) {
    const u32 *fl = (const u32 *)&c_luma_filter[0][0];
    const u32 fa0 = fl[2 * fa], fa1 = fl[2 * fa + 1], fb0 = fl[2 * fb], fb1 = fl[2 * fb + 1];
    u32 *pa = s_h + slot_a * (G::H_ELEMS / 2), *pb = s_h + slot_b * (G::H_ELEMS / 2);
    const int npair = ph >> 1, ngrp = (w >> 2) + 1;
    for (int i = tid; i < npair * ngrp; i += T) {
      const int g = i / npair, yp = i - g * npair, x0 = 4 * g;       // consecutive lanes: consecutive dwords of one column
      int ha[4][2], hb[4][2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const u32 *q = (const u32 *)(s_p + (2 * yp + rr) * G::PS + x0);
        const u32 d0 = q[0] ^ 0x80808080u, d1 = q[1] ^ 0x80808080u, d2 = q[2] ^ 0x80808080u;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const u32 lo = k ? __builtin_amdgcn_alignbyte(d1, d0, (u32)k) : d0, hi = k ? __builtin_amdgcn_alignbyte(d2, d1, (u32)k) : d1;
          ha[k][rr] = __builtin_amdgcn_sdot4((int)fa0, (int)lo, __builtin_amdgcn_sdot4((int)fa1, (int)hi, 8192, false), false);
          hb[k][rr] = __builtin_amdgcn_sdot4((int)fb0, (int)lo, __builtin_amdgcn_sdot4((int)fb1, (int)hi, 8192, false), false);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pa[((x0 + k) * G::HT >> 1) + yp] = ((u32)ha[k][0] & 0xffffu) | ((u32)ha[k][1] << 16);
        pb[((x0 + k) * G::HT >> 1) + yp] = ((u32)hb[k][0] & 0xffffu) | ((u32)hb[k][1] << 16);
      }
    }
  };

  // SATD of one candidate given as bytes (the integer position: P itself) against s_cur -> s_cost[0].  Four lanes per
  // 8x8 sub-block: lane p takes rows 2p, 2p+1 and the last two vertical Hadamard stages cross the quad with DPP.
  const int p = tid & 3;
  const short sg1 = (p & 1) ? (short)-1 : (short)1, sg2 = (p & 2) ? (short)-1 : (short)1;
  const v2s m1 = { sg1, sg1 }, m2 = { sg2, sg2 };
  // The 8x8 grid of the scores.  For a dimension that is 4 mod 8 the reference's two SATD helpers differ: satd_any_size
  // (the integer position, strategies-picture.h:62-100) scores the first 4-pixel column / row in 4x4 blocks and puts the
  // 8x8 grid behind them; satd_any_size_quad (the fractional candidates, picture-generic.c:392-456) only shrinks the
  // size by 4 -- its 4x4 stages add nothing -- and keeps the grid at the block origin.  Both are reproduced: ox / oy
  // below is the grid origin of the integer position, the candidates use (0, 0).
  const int ox = w & 4, oy = h & 4;
  const int w8 = (w - ox) >> 3, n8 = w8 * ((h - oy) >> 3);
  // lanes that share a candidate form aligned runs of min(4 n8, 64) lanes when n8 is a power of two: their SATDs are
  // added in registers first (many lanes on one LDS address with an atomic serialise)
  const int run = 4 * n8 < 64 ? 4 * n8 : 64;
  const bool run_pow2 = (n8 & (n8 - 1)) == 0;
  auto add_cost = [&](int k, u32 m) {
    m = group_sum<4>(m);
    u32 v = p == 0 ? (m + 2) >> 2 : 0u;
    if (run_pow2) {
      v = run == 4 ? v : run == 8 ? group_sum<8>(v) : run == 16 ? group_sum<16>(v) : run == 32 ? group_sum<32>(v) : group_sum<64>(v);
      if (((tid & 63) & (run - 1)) == 0) atomicAdd(&s_cost[k], v);
    } else if (p == 0) {
      atomicAdd(&s_cost[k], v);
    }
  };
  auto score_integer = [&]() {
    if (tid < 4) s_cost[tid] = 0;
    sync();
    if (ox | oy) {
      // 4x4 blocks: the first 4-pixel column over the full height, or the first 4-pixel row (one lane per block)
      const int nb = ox ? h >> 2 : w >> 2;
      for (int i = tid; i < nb; i += T) {
        const int bx4 = ox ? 0 : 4 * i, by4 = ox ? 4 * i : 0;
        u32 ra[4], rb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __builtin_memcpy(&ra[r], s_cur + (by4 + r) * G::CS + bx4, 4);
          __builtin_memcpy(&rb[r], s_p + (4 + by4 + r) * G::PS + 4 + bx4, 4);
        }
        atomicAdd(&s_cost[0], satd4x4_regs(ra, rb));
      }
    }
    for (int i = tid; i < n8 * 4; i += T) {
      const int sb = i >> 2, by = sb / w8, bx = sb - by * w8;
      const u8 *a = s_cur + (oy + by * 8 + 2 * p) * G::CS + ox + bx * 8;
      const u8 *b = s_p + (4 + oy + by * 8 + 2 * p) * G::PS + 4 + ox + bx * 8;
      uint4 x, y;
      __builtin_memcpy(&x.x, a, 4); __builtin_memcpy(&x.y, a + 4, 4); __builtin_memcpy(&x.z, a + G::CS, 4); __builtin_memcpy(&x.w, a + G::CS + 4, 4);
      __builtin_memcpy(&y.x, b, 4); __builtin_memcpy(&y.y, b + 4, 4); __builtin_memcpy(&y.z, b + G::PS, 4); __builtin_memcpy(&y.w, b + G::PS + 4, 4);
      add_cost(0, satd8_quad_part(x, y, m1, m2));
    }
    sync();
  };

  // The four candidates of a step, filtered and scored without leaving registers.  Candidate k of the step sits at the
  // quarter-pel offset (qx, qy) = (ox, oy) + scale * square[pat + k]: S(qx & 3, qy & 3; y + (qy >> 2), x + (qx >> 2)).
  // Work item = (candidate, 8x8 sub-block, quad lane p): the lane filters rows 2p, 2p+1 x 8 columns -- per column five
  // dwords of the transposed plane and ten v_dot2_i32_i16 -- and feeds the differences straight into the quad SATD.
  auto score_step = [&](int pat, int ox, int oy, int scale) {
    if (tid < 4) s_cost[tid] = 0;
    sync();
    for (int i = tid; i < 4 * n8 * 4; i += T) {            // whole quads are active or idle together
      const int q = i >> 2, k = q / n8, sb = q - k * n8, by = sb / w8, bx = sb - by * w8;
      // square[] (search_inter.c:972-976) entries 1..8 as 2-bit fields of (value + 1)
      const int sx = (int)((0x8858u >> (2 * (pat + k - 1))) & 3u) - 1;       // -1, 1, 0, 0, -1, 1, -1, 1
      const int sy = (int)((0xa085u >> (2 * (pat + k - 1))) & 3u) - 1;       //  0, 0,-1, 1, -1,-1,  1, 1
      const int qx = ox + scale * sx, qy = oy + scale * sy;
      const int fx = qx & 3, fy = qy & 3, cx = qx >> 2, ry = qy >> 2;
      const int slot = fx == 0 ? 0 : fx == 2 ? 1 : (sx > 0 ? 3 : 2);
      const int y0 = by * 8 + 2 * p, b = y0 + ry + 1, par = b & 1;          // first tap row of output row y0 is plane row b
      u32 c0[5], c1[5];
#pragma unroll
      for (int t = 0; t < 5; ++t) { c0[t] = c_frac_vcoef.c[fy][par][0][t]; c1[t] = c_frac_vcoef.c[fy][par][1][t]; }
      const u8 *a = s_cur + y0 * G::CS + bx * 8;
      u32 cur[4];
      __builtin_memcpy(&cur[0], a, 4); __builtin_memcpy(&cur[1], a + 4, 4); __builtin_memcpy(&cur[2], a + G::CS, 4); __builtin_memcpy(&cur[3], a + G::CS + 4, 4);
      const u32 *col = s_h + slot * (G::H_ELEMS / 2) + ((bx * 8 + cx + 1) * G::HT >> 1) + (b >> 1);
      int a0[8], a1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const u32 *cj = col + j * (G::HT >> 1);
        a0[j] = 0; a1[j] = 0;
#pragma unroll
        for (int t = 0; t < 5; ++t) {
          const v2s pr = as_v2s(cj[t]);
          a0[j] = __builtin_amdgcn_sdot2(pr, as_v2s(c0[t]), a0[j], false);
          a1[j] = __builtin_amdgcn_sdot2(pr, as_v2s(c1[t]), a1[j], false);
        }
      }
      v2s dd[2][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const u32 cw0 = cur[jj >> 1], cw1 = cur[2 + (jj >> 1)];
        dd[0][jj] = ((jj & 1) ? unpack_hi(cw0) : unpack_lo(cw0)) - frac_round_clip_pair(a0[2 * jj], a0[2 * jj + 1]);
        dd[1][jj] = ((jj & 1) ? unpack_hi(cw1) : unpack_lo(cw1)) - frac_round_clip_pair(a1[2 * jj], a1[2 * jj + 1]);
      }
      add_cost(k, satd8_quad_part_diff(dd, m1, m2));
    }
    sync();
  };

  // The same step for an 8x8 block owned by one wave: with one sub-block the scheme above keeps 16 lanes busy, so the
  // work is cut finer -- lane = (candidate k, row pair p, column pair g) filters 2 x 2 samples, and the Hadamard crosses
  // the 16 lanes of a candidate with DPP (quad_perm for g, row_shl/shr:4 and row_ror:8 for p).
  auto score_step8 = [&](int pat, int ox, int oy, int scale) {
    const int g = tid & 3, pp = (tid >> 2) & 3, k = tid >> 4;
    const int sx = (int)((0x8858u >> (2 * (pat + k - 1))) & 3u) - 1, sy = (int)((0xa085u >> (2 * (pat + k - 1))) & 3u) - 1;
    const int qx = ox + scale * sx, qy = oy + scale * sy;
    const int fx = qx & 3, fy = qy & 3, cx = qx >> 2, ry = qy >> 2;
    const int slot = fx == 0 ? 0 : fx == 2 ? 1 : (sx > 0 ? 3 : 2);
    const int y0 = 2 * pp, b = y0 + ry + 1, par = b & 1;
    u32 c0[5], c1[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) { c0[t] = c_frac_vcoef.c[fy][par][0][t]; c1[t] = c_frac_vcoef.c[fy][par][1][t]; }
    const u32 *col = s_h + slot * (G::H_ELEMS / 2) + ((2 * g + cx + 1) * G::HT >> 1) + (b >> 1);
    int a0[2], a1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32 *cj = col + j * (G::HT >> 1);
      a0[j] = 0; a1[j] = 0;
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const v2s pr = as_v2s(cj[t]);
        a0[j] = __builtin_amdgcn_sdot2(pr, as_v2s(c0[t]), a0[j], false);
        a1[j] = __builtin_amdgcn_sdot2(pr, as_v2s(c1[t]), a1[j], false);
      }
    }
    const u8 *a = s_cur + y0 * G::CS + 2 * g;
    const u32 cw0 = *(const unsigned short *)a, cw1 = *(const unsigned short *)(a + G::CS);
    const v2s r0 = unpack_lo(cw0) - frac_round_clip_pair(a0[0], a0[1]), r1 = unpack_lo(cw1) - frac_round_clip_pair(a1[0], a1[1]);
    const short s4 = (tid & 4) ? (short)-1 : (short)1, s8 = (tid & 8) ? (short)-1 : (short)1;
    const v2s m4 = { s4, s4 }, m8 = { s8, s8 };
    v2s v[2] = { r0 + r1, r0 - r1 };                    // row bit 0
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v2s t = dpp_v2s<0xB1>(v[i]);                      // column bit 1
      v2s u = v[i] * m1 + t;
      t = dpp_v2s<0x4E>(u);                             // column bit 2
      u = u * m2 + t;
      t = dpp_xor4_v2s(u);                              // row bit 1
      u = u * m4 + t;
      t = dpp_v2s<0x128>(u);                            // row bit 2 (row_ror:8)
      u = u * m8 + t;
      acc = abs_last_stage(u, acc);                     // column bit 0 and the absolute sum
    }
    acc = group_sum<16>(acc);
    if ((tid & 15) == 0) s_cost[k] = (acc + 2) >> 2;
    sync();
  };
  constexpr bool SPLIT8 = WAVE && T == 64 && FW == 8 && FH == 8;
  constexpr bool SPLIT16 = WAVE && T == 64 && FW == 16 && FH == 16;
  auto score_any = [&](int pat, int ox, int oy, int scale) {
    if constexpr (SPLIT8) score_step8(pat, ox, oy, scale); else score_step(pat, ox, oy, scale);
  };

  // integer position: candidate = P[y][x]
  if constexpr (SPLIT8 || SPLIT16) {
    // blocks owned by one wave, cut into 2 x 2 samples per lane: 16 lanes per 8x8 sub-block (an 8x8 block fills lanes
    // 0..15, a 16x16 block the wave), the Hadamard crosses them with DPP -- a quarter of the instructions of the quad
    // scheme, which keeps 4 lanes per sub-block busy here
    const int g = tid & 3, pp = (tid >> 2) & 3, sb = SPLIT16 ? tid >> 4 : 0, y0 = (sb >> 1) * 8 + 2 * pp, x0 = (sb & 1) * 8 + 2 * g;
    const u8 *a = s_cur + y0 * G::CS + x0, *bq = s_p + (4 + y0) * G::PS + 4 + x0;
    const u32 cw0 = *(const unsigned short *)a, cw1 = *(const unsigned short *)(a + G::CS);
    const u32 pw0 = *(const unsigned short *)bq, pw1 = *(const unsigned short *)(bq + G::PS);
    const v2s r0 = unpack_lo(cw0) - unpack_lo(pw0), r1 = unpack_lo(cw1) - unpack_lo(pw1);
    const short s4 = (tid & 4) ? (short)-1 : (short)1, s8 = (tid & 8) ? (short)-1 : (short)1;
    const v2s m4 = { s4, s4 }, m8 = { s8, s8 };
    v2s v[2] = { r0 + r1, r0 - r1 };
    u32 acc = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      v2s t = dpp_v2s<0xB1>(v[i]);
      v2s u = v[i] * m1 + t;
      t = dpp_v2s<0x4E>(u);
      u = u * m2 + t;
      t = dpp_xor4_v2s(u);
      u = u * m4 + t;
      t = dpp_v2s<0x128>(u);
      u = u * m8 + t;
      acc = abs_last_stage(u, acc);
    }
    acc = (group_sum<16>(acc) + 2) >> 2;                 // the sub-block's SATD, in each of its 16 lanes
    if (SPLIT16) acc = group_sum<64>((tid & 15) == 0 ? acc : 0u);
    if (tid == 0) s_cost[0] = acc;
    sync();
  } else {
    score_integer();
  }
  int mx = d.x2 - d.x1, my = d.y2 - d.y1;              // pixel precision
  u32 best_bitcost = 0;
  u32 best_cost = s_cost[0];
  if (out && tid == 0) out[0] = best_cost;
  best_cost += mvc.cost(mx, my, 2, best_bitcost);
  const u32 cost0 = best_cost;
  mx *= 2; my *= 2;                                    // half-pel precision (search_inter.c:1031-1032)

  if (fme_level > 0) hor_planes(0, 0, 2, 1);
  sync();

  int best_index = 0, pat = 1;                         // pat: first index of the step's 4 positions in square[]
  for (int step = 0; step < fme_level; ++step) {
    if (step < 2) {
      score_any(pat, 0, 0, 2);                         // half-pel ring: offsets +-2 quarter-pels
    } else {
      const int hx = s_sel[0], hy = s_sel[1];          // best half-pel offset in {-1,0,1}^2
      if (step == 2) {
        hor_planes((2 * hx - 1) & 3, 2, (2 * hx + 1) & 3, 3);
        sync();
      }
      score_any(pat, 2 * hx, 2 * hy, 1);
    }
    // decision: same order and strict '<' as search_inter.c:1069-1102.  calc_mvd_cost is ~100 instructions: lane j of
    // every wave prices candidate j and the four results are read back with v_readlane (all lanes pricing all four
    // candidates was a third of a small block's search).
    const int mv_shift = step < 2 ? 1 : 0;
    const int lj = tid & 3;
    const int lpx = mx + (int)((0x8858u >> (2 * (pat + lj - 1))) & 3u) - 1, lpy = my + (int)((0xa085u >> (2 * (pat + lj - 1))) & 3u) - 1;
    u32 lbits = 0;
    const u32 lraw = s_cost[lj];
    const int lok = mvc.within(lpx * (1 << mv_shift), lpy * (1 << mv_shift)) ? 1 : 0;
    const u32 lcost = lraw + mvc.cost(lpx, lpy, mv_shift, lbits);
    if (out && tid < 4) out[(step >= 2 ? 8 : 0) + pat + tid] = lraw;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 cj = (u32)__builtin_amdgcn_readlane((int)lcost, j), bj = (u32)__builtin_amdgcn_readlane((int)lbits, j);
      if (!__builtin_amdgcn_readlane(lok, j)) continue;
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
  frac_result res = { mx, my, best_cost, best_bitcost, cost0 };
  return res;
}

// every PU shape of the inter search: multiples of 4 up to 64 (the AMP / SMP shapes 8x4, 4x8, 16x4, 4x16, 16x12, 12x16
// have one dimension that is 4 mod 8; never both)
__device__ __forceinline__ bool frac_shape_ok(int w, int h)
{
  return !(w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || ((w & 4) && (h & 4)));
}

}  // namespace kvzhip
