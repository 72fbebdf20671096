// me_search.hip -- one motion search per PU, entirely on the device: hexagon search with MV bit
// costs, then the fused fractional search.
//
// Reference: the --me hexbs (and --me dia, diamond_search :796-883; --me tz, tz_search :595-672) path of search_pu_inter_ref (src/search_inter.c:1134-1300):
// hexagon_search (:690-778) = select_starting_point (:282-307) + early_terminate (:415-460) +
// the 6/3/8-point patterns, every candidate through check_mv_cost (:195-232) = kvz_image_calc_sad
// (image.c:455-486) + calc_mvd_cost (:373-412); then search_frac (:965-1128).  SURVEY.md 8(f) row 1.
//
// The reference walks each pattern one candidate at a time and keeps the running best with a
// strict '<', i.e. it takes the first minimum of the group in visiting order.  Here the SADs of a
// whole group (up to 8 candidates) are computed at once -- work item = (candidate, 8-pixel row
// segment), spread over the wave / workgroup that owns the PU, partial sums through LDS atomics --
// and the same first-minimum rule is applied to the group, so every decision (and therefore the
// path the search takes) is the reference's.  The current block lives in LDS for the whole search;
// reference pixels come from L2/HBM with the clamp addressing of image_interpolated_sad
// (image.c:320-444).  The fractional stage is frac_core.h with the MV cost model plugged in.
#include "kvz_hip_internal.h"
#include "serve_seq.h"
#include <type_traits>
#include "frac_core.h"

using namespace kvzhip;

namespace {

// calc_mvd_cost / fracmv_within_tile on the flattened encoder state (include/kvz_hip.h).  The descriptor is copied
// into (scalar) registers once per PU: the cost model runs for every one of the ~60 candidates of a search, and
// reading the merge list from memory each time made scalar loads the longest chain of the kernel.
// --mv-rdo: the CABAC probability tables of ITU-T H.265 (Tables 9-46 rangeTabLps, 9-47 transIdxLps) -- the reference's
// kvz_g_auc_lpst_table / kvz_g_auc_next_state_lps (cabac.c:28-75); an MPS moves to min(state + 1, 62); the
// renormalisation shift kvz_g_auc_renorm_table[lps >> 3] is clz(lps >> 3) - 26
__constant__ unsigned char c_range_lps[64 * 4] = {
  128,176,208,240, 128,167,197,227, 128,158,187,216, 123,150,178,205, 116,142,169,195, 111,135,160,185, 105,128,152,175, 100,122,144,166,
   95,116,137,158,  90,110,130,150,  85,104,123,142,  81, 99,117,135,  77, 94,111,128,  73, 89,105,122,  69, 85,100,116,  66, 80, 95,110,
   62, 76, 90,104,  59, 72, 86, 99,  56, 69, 81, 94,  53, 65, 77, 89,  51, 62, 73, 85,  48, 59, 69, 80,  46, 56, 66, 76,  43, 53, 63, 72,
   41, 50, 59, 69,  39, 48, 56, 65,  37, 45, 54, 62,  35, 43, 51, 59,  33, 41, 48, 56,  32, 39, 46, 53,  30, 37, 43, 50,  29, 35, 41, 48,
   27, 33, 39, 45,  26, 31, 37, 43,  24, 30, 35, 41,  23, 28, 33, 39,  22, 27, 32, 37,  21, 26, 30, 35,  20, 24, 29, 33,  19, 23, 27, 31,
   18, 22, 26, 30,  17, 21, 25, 28,  16, 20, 23, 27,  15, 19, 22, 25,  14, 18, 21, 24,  14, 17, 20, 23,  13, 16, 19, 22,  12, 15, 18, 21,
   12, 14, 17, 20,  11, 14, 16, 19,  11, 13, 15, 18,  10, 12, 15, 17,  10, 12, 14, 16,   9, 11, 13, 15,   9, 11, 12, 14,   8, 10, 12, 14,
    8,  9, 11, 13,   7,  9, 11, 12,   7,  9, 10, 12,   7,  8, 10, 11,   6,  8,  9, 11,   6,  7,  9, 10,   6,  7,  8,  9,   2,  2,  2,  2 };
__constant__ unsigned char c_trans_lps[64] = {
   0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9,11,11,12,13,13,15,15,16,16,18,18,19,19,21,21,22,22,23,24,
  24,25,26,26,27,27,28,29,29,30,30,30,31,32,32,33,33,33,34,34,35,35,35,36,36,36,37,37,37,38,38,63 };

// what kvz_calc_mvd_cost_cabac reads of state->cabac, and the bits its counting-mode encoder produces: the count
// (23 - bits_left) + 8 * num_buffered_bytes (cabac.c:95-140) is the number of renormalisation shifts, a function of `range`
// and the context states alone
struct cabac_model {
  u32 range;
  u32 ctx[7];        // uc_state of merge_flag, merge_idx, ref_pic[0], ref_pic[1], mvd[0], mvd[1], mvp_idx[0]
  template <int C>
  __device__ __forceinline__ u32 bin(bool b)           // kvz_cabac_encode_bin, cabac.c:90-122
  {
    const u32 uc = ctx[C], st = uc >> 1, lps = c_range_lps[st * 4 + ((range >> 6) & 3)];
    range -= lps;
    if ((b ? 1u : 0u) != (uc & 1u)) {
      const u32 n = (u32)__clz((int)(lps >> 3)) - 26u;
      range = lps << n;
      ctx[C] = ((u32)c_trans_lps[st] << 1) | ((uc & 1u) ^ (st == 0 ? 1u : 0u));
      return n;
    }
    ctx[C] = ((st < 62 ? st + 1 : st) << 1) | (uc & 1u);
    if (range >= 256) return 0;
    range <<= 1;
    return 1;
  }
  // kvz_cabac_write_ep_ex_golomb(symbol, 1), cabac.c:535-570: number of bypass bins
  static __device__ __forceinline__ u32 ex_golomb1(u32 symbol)
  {
    u32 n = 0, count = 1;
    while (symbol >= (1u << count)) { ++n; symbol -= 1u << count; ++count; }
    return n + 1 + count;
  }
  // kvz_encode_mvd, encode_coding_tree.c:1156-1202
  __device__ __forceinline__ u32 mvd(int hor, int ver)
  {
    const u32 ah = (u32)(hor < 0 ? -hor : hor), av = (u32)(ver < 0 ? -ver : ver);
    u32 bits = bin<4>(hor != 0);
    bits += bin<4>(ver != 0);
    if (hor) bits += bin<5>(ah > 1);
    if (ver) bits += bin<5>(av > 1);
    if (hor) bits += (ah > 1 ? ex_golomb1(ah - 2) : 0u) + 1u;
    if (ver) bits += (av > 1 ? ex_golomb1(av - 2) : 0u) + 1u;
    return bits;
  }
};

// RDO: --mv-rdo cost model.  CONSTR: some fracmv_within_tile rule is active (WPP / OWF availability or an mv_constraint); the
// common unconstrained search is compiled without the rule and its scalar state (the kernels sit at the edge of their SGPR budget).
template <bool RDO, bool CONSTR = true>
struct me_cost_model_t {
  int px, py, pw, ph;
  int cand[2][2];
  int n_merge;
  int mx[5], my[5];
  u32 usable, same_ref;                                // bit i = merge[i].usable / .same_ref
  u32 mkey[5];                                         // merge vector i as (x & 0xffff) | y << 16, for merge_match
  int lambda_cost, wpp_owf, ref_delay_px, max_down, max_right;
  int constraint, ox, oy, tw, th;                      // cfg.mv_constraint, tile-relative origin of the PU, tile size
  cabac_model cab;                                     // RDO only
  int rdo_ref_idx, rdo_refs_before;

  __device__ __forceinline__ me_cost_model_t(const kvz_hip_me_pu &pu, const kvz_hip_me_params &prm)
  {
    if (RDO) {
      const kvz_hip_me_cabac &c = prm.cabac[pu.reserved];
      cab.range = c.range;
#pragma unroll
      for (int i = 0; i < 7; ++i) cab.ctx[i] = c.ctx[i];
      rdo_ref_idx = prm.ref_idx; rdo_refs_before = prm.refs_before;
    }
    px = pu.x; py = pu.y; pw = pu.width; ph = pu.height;
    cand[0][0] = pu.mv_cand[0][0]; cand[0][1] = pu.mv_cand[0][1]; cand[1][0] = pu.mv_cand[1][0]; cand[1][1] = pu.mv_cand[1][1];
    n_merge = pu.num_merge_cand;
    usable = 0; same_ref = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      mx[i] = pu.merge[i].mv[0]; my[i] = pu.merge[i].mv[1];
      mkey[i] = ((u32)mx[i] & 0xffffu) | ((u32)my[i] << 16);
      if (i < n_merge && pu.merge[i].usable) usable |= 1u << i;
      if (pu.merge[i].same_ref) same_ref |= 1u << i;
    }
    lambda_cost = prm.lambda_cost; wpp_owf = prm.wpp_owf; ref_delay_px = prm.ref_delay_px;
    max_down = prm.max_ref_lcu_down; max_right = prm.max_ref_lcu_right;
    constraint = prm.mv_constraint;
    ox = pu.x - prm.tile_x; oy = pu.y - prm.tile_y; tw = prm.tile_w; th = prm.tile_h;
  }

  // fracmv_within_tile (search_inter.c:87-176), all mv_constraint branches; quarter-pel vector.  info->origin is
  // relative to the tile, and so are the LCU indices of the availability rule (C division: truncation toward zero).
  __device__ __forceinline__ bool within(int x, int y) const
  {
    if (!CONSTR) return true;
    const bool frac_luma = x % 4 != 0 || y % 4 != 0, frac_chroma = x % 8 != 0 || y % 8 != 0;
    if (wpp_owf) {
      int margin = frac_luma ? 4 : (frac_chroma ? 2 : 0);
      margin += ref_delay_px;
      const int lcu_x = ox / 64, lcu_y = oy / 64;
      const int mv_lcu_x = ((ox + pw + margin) * 4 + x) / (64 << 2) - lcu_x;
      const int mv_lcu_y = ((oy + ph + margin) * 4 + y) / (64 << 2) - lcu_y;
      if (mv_lcu_y > max_down) return false;
      if (mv_lcu_x + mv_lcu_y > max_down + max_right) return false;
    }
    if (constraint == 0) return true;
    const int margin = constraint == 4 ? (frac_luma ? 4 << 2 : (frac_chroma ? 2 << 2 : 0)) : 0;
    const int ax = ox * 4 + x, ay = oy * 4 + y;
    const int from_right = (tw << 2) - (ax + (pw << 2)), from_bottom = (th << 2) - (ay + (ph << 2));
    return ax >= margin && ay >= margin && from_right >= margin && from_bottom >= margin;
  }
  // get_ep_ex_golomb_bitcost (:235-254)
  static __device__ __forceinline__ u32 golomb(u32 symbol)
  {
    symbol += 2;
    // the reference's four range tests add up to 2 * floor(log2(symbol)) while symbol < 2^16 (they test bits 8, 4, 2, 1
    // of the exponent once each); vectors are int16, so only a difference of two extreme vectors gets past that
    if (__builtin_expect(symbol < (1u << 16), 1)) return 2u * (31u - (u32)__builtin_clz(symbol));
    u32 bins = 0;
    if (symbol >= 1u << 8) { bins += 16; symbol >>= 8; }
    if (symbol >= 1u << 4) { bins += 8; symbol >>= 4; }
    if (symbol >= 1u << 2) { bins += 4; symbol >>= 2; }
    if (symbol >= 1u << 1) { bins += 2; }
    return bins;
  }
  // get_mvd_coding_cost (:310-323): whole bits, the fixed-point rounding is exact
  static __device__ __forceinline__ u32 mvd_bits(int dx, int dy)
  {
    return golomb((u32)(dx < 0 ? -dx : dx)) + golomb((u32)(dy < 0 ? -dy : dy));
  }
  // select_mv_cand (:326-370).  |d| + 2 of each vector component is one v_sad_u32 on operands moved into the unsigned
  // range, the exp-Golomb length of a component 2 * (31 - clz(|d| + 2)) (see golomb), so a candidate costs
  // 124 - 2 * (clz + clz) bits and the cheaper of the two is the one with the larger clz sum.
  __device__ __forceinline__ int select_cand(int mvx, int mvy, u32 &cost) const
  {
    constexpr u32 BIAS = 1u << 20;                       // |mv|, |candidate| < 2^18: sums stay positive
    const u32 xb = (u32)mvx + BIAS, yb = (u32)mvy + BIAS;
    u32 s[4];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const u32 cx = (u32)cand[c][0] + BIAS, cy = (u32)cand[c][1] + BIAS;
      s[2 * c] = (xb > cx ? xb - cx : cx - xb) + 2u;     // v_sad_u32
      s[2 * c + 1] = (yb > cy ? yb - cy : cy - yb) + 2u;
    }
    if (__builtin_expect(((s[0] | s[1] | s[2] | s[3]) >> 16) != 0, 0)) {        // a difference of two extreme vectors
      const u32 c1 = mvd_bits(mvx - cand[0][0], mvy - cand[0][1]), c2 = mvd_bits(mvx - cand[1][0], mvy - cand[1][1]);
      cost = c1 < c2 ? c1 : c2;
      return c2 < c1 ? 1 : 0;
    }
    const u32 z1 = (u32)__builtin_clz(s[0]) + (u32)__builtin_clz(s[1]), z2 = (u32)__builtin_clz(s[2]) + (u32)__builtin_clz(s[3]);
    cost = 124u - 2u * (z1 > z2 ? z1 : z2);
    return z2 > z1 ? 1 : 0;
  }
  // index of the first merge candidate that codes (x, y) (quarter-pel) for this reference, or -1
  __device__ __forceinline__ int merge_match(int x, int y) const
  {
    // one compare per candidate on the packed vector; a vector outside int16 matches nothing (the candidates are int16)
    const u32 key = ((u32)x & 0xffffu) | ((u32)y << 16), live = usable & same_ref;
    int m = -1;
#pragma unroll
    for (int i = 4; i >= 0; --i)
      if ((live >> i & 1u) && mkey[i] == key) m = i;
    return ((u32)(x + 32768) < 65536u && (u32)(y + 32768) < 65536u) ? m : -1;
  }
  // kvz_get_mvd_coding_cost_cabac (rdo.c:883-903): a fresh copy of the state per call
  __device__ __forceinline__ u32 mvd_bits_cabac(int dx, int dy) const
  {
    cabac_model m = cab;
    return m.mvd(dx, dy);
  }
  // select_mv_cand (:326-370) with --mv-rdo, cost_out == NULL
  __device__ __forceinline__ int select_cand_cabac(int mvx, int mvy) const
  {
    const u32 c1 = mvd_bits_cabac(mvx - cand[0][0], mvy - cand[0][1]), c2 = mvd_bits_cabac(mvx - cand[1][0], mvy - cand[1][1]);
    return c2 < c1 ? 1 : 0;
  }
  // kvz_calc_mvd_cost_cabac (rdo.c:908-1060)
  __device__ __forceinline__ u32 cost_cabac(int x, int y, u32 &bits) const
  {
    const int mi = merge_match(x, y);
    int cur_cand = 0, dx = 0, dy = 0;
    if (mi < 0) {
      const int d1x = x - cand[0][0], d1y = y - cand[0][1], d2x = x - cand[1][0], d2y = y - cand[1][1];
      const u32 c1 = mvd_bits_cabac(d1x, d1y), c2 = mvd_bits_cabac(d2x, d2y);
      if (c2 < c1) { cur_cand = 1; dx = d2x; dy = d2y; } else { dx = d1x; dy = d1y; }
    }
    cabac_model m = cab;
    u32 b = m.template bin<0>(mi >= 0);
    if (mi >= 0) {
      for (int ui = 0; ui < 4; ++ui) {                   // MRG_MAX_NUM_CANDS - 1
        const bool symbol = ui != mi;
        b += ui == 0 ? m.template bin<1>(symbol) : 1u;
        if (!symbol) break;
      }
    } else {
      if (rdo_refs_before > 1) {
        int ref_frame = rdo_ref_idx;
        b += m.template bin<2>(ref_frame != 0);
        if (ref_frame > 0) {
          const int ref_num = rdo_refs_before - 2;
          --ref_frame;
          for (int i = 0; i < ref_num; ++i) {
            const bool symbol = i != ref_frame;
            b += i == 0 ? m.template bin<3>(symbol) : 1u;
            if (!symbol) break;
          }
        }
      }
      b += m.mvd(dx, dy);
      b += m.template bin<6>(cur_cand != 0);
    }
    bits = b;
    return __umul24(b, (u32)lambda_cost);
  }
  // calc_mvd_cost (:373-412)
  __device__ __forceinline__ u32 cost(int x, int y, int mv_shift, u32 &bits) const
  {
    x *= 1 << mv_shift;
    y *= 1 << mv_shift;
    if (RDO) return cost_cabac(x, y, bits);
    const int m = merge_match(x, y);
    if (m >= 0) bits = (u32)m;
    else select_cand(x, y, bits);
    return __umul24(bits, (u32)lambda_cost);            // bits < 2^7, lambda_cost <= 2^20 (checked by the entry): a full-rate multiply
  }
  // mv_in_merge (:260-273), integer-pel vector
  __device__ __forceinline__ bool in_merge(int x, int y) const
  {
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if ((usable >> i & 1u) && ((mx[i] + 2) >> 2) == x && ((my[i] + 2) >> 2) == y) hit = true;
    return hit;
  }
};
typedef me_cost_model_t<false, true> me_cost_model;

__constant__ signed char c_large_hex[9][2] = { { 0, 0 }, { 1, -2 }, { 2, 0 }, { 1, 2 }, { -1, 2 }, { -2, 0 }, { -1, -2 }, { 1, -2 }, { 2, 0 } };
__constant__ signed char c_small_hex[9][2] = { { 0, 0 }, { 0, -1 }, { -1, 0 }, { 1, 0 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
__constant__ signed char c_diamond[5][2] = { { 0, -1 }, { 1, 0 }, { 0, 1 }, { -1, 0 }, { 0, 0 } };
__constant__ signed char c_et_hex[7][2] = { { 0, -1 }, { -1, 0 }, { 0, 1 }, { 1, 0 }, { 0, -1 }, { -1, 0 }, { 0, 0 } };

constexpr int ME_GROUP = 64;                          // candidates evaluated per round (the patterns use at most 8)
struct me_shared { u32 sad[ME_GROUP]; int cx[ME_GROUP], cy[ME_GROUP]; };

constexpr int FULL_OUT = 32;                          // me_shared slot where full_search_wg leaves (x, y, cost, bits)
constexpr int FULL_MAX_WINDOWS = 7;                   // zero vector, extra_mv, five merge candidates

// search_mv_full (search_inter.c:886-962) for the search service: ALL threads of the workgroup on the positions of one PU
// and one reference picture, whatever the PU's size -- the latency form of the exhaustive search (the one-wave-per-PU form
// in search_pu_core is the throughput form: 64 positions per round).
//   * the (w + 2R) x (h + 2R) reference pixels of as many windows as fit are staged in LDS at once (edge replicated,
//     image.c:320-444), the current block beside them;
//   * a work item is FOUR neighbouring positions of one window row: v_qsad_pk_u16_u8 prices the four alignments of a
//     reference dword pair against one dword of the block in one instruction (16-bit packed sums, emptied into 32-bit
//     ones before 64 of them can overflow: 64 x 4 x 255 < 2^16); QSAD = false keeps v_alignbyte + v_sad_u8;
//   * the reference walks the windows in order and replaces its best on a strictly smaller cost, so the winner is the
//     smallest (cost, visiting order) pair: every thread keeps its own 64-bit minimum and one reduction ends the search.
// A position inside an earlier window is skipped as :936-952 does; one that fails fracmv_within_tile costs 2^32 - 1 and
// never wins.  Result (all threads must call; ends with the values in sh, NOT yet visible: the caller synchronises).
template <bool CONSTR, bool QSAD, int T>
__device__ __forceinline__ void full_search_wg(int tid, u8 *lds, int lds_bytes, me_shared *sh, const u8 *__restrict__ pic, u32 pic_stride,
                                               const refplane_t &ref, const kvz_hip_me_pu &pu, const kvz_hip_me_params &prm)
{
  const me_cost_model_t<false, CONSTR> mvc(pu, prm);
  const int w = pu.width, h = pu.height, R = prm.search_range, side = 2 * R + 1;
  const int wq = w >> 2;                               // every PU width is a multiple of 4
  // ---- the windows, in the reference's order: sh->cx / cy = centre, sh->sad = index of the merge candidate (or -1) ----
  int n_win = 1;
  auto add_window = [&](int cx, int cy, int merge_index) {
    if (tid == 0) { sh->cx[n_win] = cx; sh->cy[n_win] = cy; sh->sad[n_win] = (u32)merge_index; }
    ++n_win;
  };
  if (tid == 0) { sh->cx[0] = 0; sh->cy[0] = 0; sh->sad[0] = ~0u; }
  {
    const int ex = pu.extra_mv[0] >> 2, ey = pu.extra_mv[1] >> 2;
    // (an extra window on the zero vector repeats window 0 and can improve nothing: costs must be strictly smaller)
    if (!mvc.in_merge(ex, ey) && (ex != 0 || ey != 0)) add_window(ex, ey, -1);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (!(mvc.usable >> i & 1u)) continue;
      const int cx0 = mvc.mx[i] >> 2, cy0 = mvc.my[i] >> 2;        // plain shift here (:917-920)
      if (cx0 == 0 && cy0 == 0) continue;
      add_window(cx0, cy0, i);
    }
  }
  // ---- the current block never enters LDS: its address is the same in every lane, so it is read with scalar loads (constant address
  // space: s_load_dwordx2..x16 per row) and feeds the SAD instructions as SGPR operands.  The scalar cache may hold what a previous
  // unit of a resident worker read from a slot that has been overwritten since: dropped here.
  __builtin_amdgcn_s_dcache_inv();
  typedef const u32 __attribute__((address_space(4))) cu32;
  const cu32 *const cur_c = (const cu32 *)(unsigned long long)(pic + (size_t)pu.y * pic_stride + pu.x);     // 4-byte aligned: x and the stride are multiples of 4
  const int cstride = (int)(pic_stride >> 2);
  u8 *const s_win = lds;
  // an item is NQ quads of neighbouring positions of one window row: 4 positions when the windows give the 512 threads one round of
  // items or less, 8 otherwise (one more reference dword per row serves four more positions: half the LDS traffic per position)
  const int groups4 = (side + 3) >> 2, groups8 = (side + 7) >> 3;
  // (blocks up to 16 pixels wide are bound by the pricing of the positions, not by LDS: there 8 positions per item only pay when they
  // save rounds outright -- an item of two quads costs about 1.6 items of one)
  const int rounds4 = (n_win * side * groups4 + T - 1) / T, rounds8 = (n_win * side * groups8 + T - 1) / T;
  const bool wide = wq >= 8 ? rounds4 > 1 : 16 * rounds8 < 10 * rounds4;
  const int groups = wide ? groups8 : groups4;
  const int wstride = (w + 8 * groups8 + 4 + 3) & ~3, wrows = h + 2 * R, win_bytes = wstride * wrows, wsq = wstride >> 2;
  int per_chunk = lds_bytes / win_bytes;                // >= 1 for every legal PU and range (64x64, R = 64: 204 x 192 bytes)
  if (per_chunk > FULL_MAX_WINDOWS) per_chunk = FULL_MAX_WINDOWS;
  const int items_per_win = side * groups;
  unsigned long long best = ~0ull;
  __syncthreads();                                      // the window list

  // prices the positions of one quad of an item and keeps the thread's smallest (cost, visiting order)
  auto price_quad = [&](int kk, int r, int col0, const u32 (&tot)[4]) {
    const int cx = sh->cx[kk], cy = sh->cy[kk], mine = (int)sh->sad[kk];
    const int y = cy + r - R;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int col = col0 + p, x = cx + col - R;
      if (col >= side) continue;
      bool skip = false;
      if (mine >= 0) {                                   // a merge candidate's window: :936-952
        if (x >= -R && x <= R && y >= -R && y <= R) skip = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (j >= mine || !(mvc.usable >> j & 1u)) continue;
          const int xx = mvc.mx[j] >> 2, yy = mvc.my[j] >> 2;
          if (x >= xx - R && x <= xx + R && y >= yy - R && y <= yy + R) skip = true;
        }
      }
      if (skip || !mvc.within(x * 4, y * 4)) continue;
      u32 bits;
      const u32 cost = tot[p] + mvc.cost(x, y, 2, bits);                  // < 2^32: lambda_cost is bounded by the entry
      const unsigned long long key = ((unsigned long long)cost << 32) | (u32)(kk * side * side + r * side + col);
      best = key < best ? key : best;
    }
  };
  // the SADs of an item: WQ = dwords per block row (0: any width, one scalar load per dword), NQ = quads
  auto run_items = [&](auto wq_tag, auto nq_tag, int k0, int nk) {
    constexpr int WQ = decltype(wq_tag)::value, NQ = decltype(nq_tag)::value;
    const int nwq = WQ ? WQ : wq;
    const int flush_rows = nwq >= 64 ? 1 : 64 / nwq;    // rows of 16-bit sums that cannot overflow: 64 x 4 x 255 < 2^16
    for (int it = tid; it < nk * items_per_win; it += T) {
      const int k = it / items_per_win, rem = it - k * items_per_win, r = rem / groups, g = rem - r * groups;
      const u32 *q = (const u32 *)(s_win + k * win_bytes + r * wstride) + NQ * g;
      const cu32 *c = cur_c;
      u32 tot[NQ][4] = {};
      for (int yb = 0; yb < h; yb += flush_rows) {
        const int ye = yb + flush_rows < h ? yb + flush_rows : h;
        unsigned long long acc[NQ] = {};
        for (int y = yb; y < ye; ++y) {
          u32 d[NQ + 1];
#pragma unroll
          for (int i = 0; i < NQ; ++i) d[i] = q[i];
#pragma unroll 16
          for (int xq = 0; xq < nwq; ++xq) {
            d[NQ] = q[xq + NQ];
            const u32 cv = c[xq];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
              if (QSAD) {
                acc[i] = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)d[i + 1] << 32) | d[i], cv, acc[i]);
              } else {
                tot[i][0] = __builtin_amdgcn_sad_u8(cv, d[i], tot[i][0]);
                tot[i][1] = __builtin_amdgcn_sad_u8(cv, __builtin_amdgcn_alignbyte(d[i + 1], d[i], 1u), tot[i][1]);
                tot[i][2] = __builtin_amdgcn_sad_u8(cv, __builtin_amdgcn_alignbyte(d[i + 1], d[i], 2u), tot[i][2]);
                tot[i][3] = __builtin_amdgcn_sad_u8(cv, __builtin_amdgcn_alignbyte(d[i + 1], d[i], 3u), tot[i][3]);
              }
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) d[i] = d[i + 1];
          }
          q += wsq; c += cstride;
        }
        if (QSAD) {
#pragma unroll
          for (int i = 0; i < NQ; ++i) {
            tot[i][0] += (u32)acc[i] & 0xffffu; tot[i][1] += (u32)(acc[i] >> 16) & 0xffffu;
            tot[i][2] += (u32)(acc[i] >> 32) & 0xffffu; tot[i][3] += (u32)(acc[i] >> 48);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NQ; ++i) price_quad(k0 + k, r, 4 * (NQ * g + i), tot[i]);
    }
  };
  auto run_width = [&](auto nq_tag, int k0, int nk) {
    switch (wq) {
      case 2: run_items(std::integral_constant<int, 2>(), nq_tag, k0, nk); break;
      case 4: run_items(std::integral_constant<int, 4>(), nq_tag, k0, nk); break;
      case 8: run_items(std::integral_constant<int, 8>(), nq_tag, k0, nk); break;
      case 16: run_items(std::integral_constant<int, 16>(), nq_tag, k0, nk); break;
      default: run_items(std::integral_constant<int, 0>(), nq_tag, k0, nk); break;
    }
  };

  for (int k0 = 0; k0 < n_win; k0 += per_chunk) {
    const int nk = n_win - k0 < per_chunk ? n_win - k0 : per_chunk;
    if (k0) __syncthreads();                            // the previous chunk has been read
    for (int k = 0; k < nk; ++k) {
      const int x0 = pu.x + sh->cx[k0 + k] - R, y0 = pu.y + sh->cy[k0 + k] - R;
      u8 *const dst = s_win + k * win_bytes;
      for (int i = tid; i < wsq * wrows; i += T) {
        const int y = i / wsq, q = (i - y * wsq) * 4;
        u32 v;
        if (x0 + q >= 0 && x0 + q + 4 <= ref.w && y0 + y >= 0 && y0 + y < ref.h) {
          __builtin_memcpy(&v, ref.p + (size_t)(y0 + y) * ref.stride + x0 + q, 4);
        } else {
          v = (u32)ref_px(ref, x0 + q, y0 + y) | ((u32)ref_px(ref, x0 + q + 1, y0 + y) << 8) |
              ((u32)ref_px(ref, x0 + q + 2, y0 + y) << 16) | ((u32)ref_px(ref, x0 + q + 3, y0 + y) << 24);
        }
        *(u32 *)(dst + y * wstride + q) = v;
      }
    }
    __syncthreads();
    if (wide) run_width(std::integral_constant<int, 2>(), k0, nk);
    else run_width(std::integral_constant<int, 1>(), k0, nk);
  }
  // ---- the smallest (cost, order) of the workgroup ----
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const u32 lo = (u32)__shfl_xor((int)(u32)best, off, 64), hi = (u32)__shfl_xor((int)(u32)(best >> 32), off, 64);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    best = o < best ? o : best;
  }
  __syncthreads();                                      // every reader of the window list is done; sh->sad is reused
  if ((tid & 63) == 0) { sh->sad[48 + 2 * (tid >> 6)] = (u32)best; sh->sad[49 + 2 * (tid >> 6)] = (u32)(best >> 32); }
  __syncthreads();
  best = ~0ull;
#pragma unroll
  for (int v = 0; v < T / 64; ++v) {
    const unsigned long long o = ((unsigned long long)sh->sad[49 + 2 * v] << 32) | sh->sad[48 + 2 * v];
    best = o < best ? o : best;
  }
  if (tid == 0) {
    int bx = 0, by = 0;
    u32 bcost = 0xffffffffu, bbits = 0;
    if ((u32)(best >> 32) != 0xffffffffu) {
      const int seq = (int)(u32)best, kk = seq / (side * side), rem = seq - kk * side * side, r = rem / side, col = rem - r * side;
      bx = sh->cx[kk] + col - R; by = sh->cy[kk] + r - R;
      bcost = (u32)(best >> 32);
      mvc.cost(bx, by, 2, bbits);
    }
    sh->cx[FULL_OUT] = bx; sh->cy[FULL_OUT] = by; sh->sad[FULL_OUT] = bcost; sh->sad[FULL_OUT + 1] = bbits;
  }
}

// One PU.  T threads (a wave with wave-private LDS, or the whole workgroup) share the work; every thread
// carries the same search state, so all decisions are uniform across them.
// BOTH (the search service, serve.hip): the fractional stage always runs and `out` is a serve_result that also receives the
// outcome search_pu_inter_ref reaches when the integer search does not beat *inter_cost (:1239-1252), so that the pictures
// of a PU can be searched in parallel and the sequential rule replayed afterwards.
template <int MAXW, int T, bool WAVE, int FW = 0, int FH = 0, bool RDO = false, bool CONSTR = true, bool BOTH = false>
__device__ __forceinline__ void search_pu_core(int tid, u8 *lds, me_shared *sh, const u8 *__restrict__ pic, u32 pic_stride,
                                               const refplane_t &ref, const kvz_hip_me_pu &pu, const kvz_hip_me_params &prm,
                                               kvz_hip_me_result *__restrict__ out, size_t pu_index)
{
  typedef frac_geom<MAXW> G;
  u8 *s_cur = lds + G::P_BYTES;                        // same place search_frac_core keeps the current block
  auto sync = [&]() { if (WAVE) wave_lds_fence(); else __syncthreads(); };
  const me_cost_model_t<RDO, CONSTR> mvc(pu, prm);
  const int w = FW ? FW : pu.width, h = FH ? FH : pu.height;                       // FW, FH: compile-time size (0 = runtime)
  // a row is cut into 8-pixel segments, or 4-pixel ones when the width is 4 or 12 (AMP / SMP shapes)
  const bool seg4 = !FW && (w & 4);
  const int segw = seg4 ? 4 : 8, w8 = seg4 ? w >> 2 : w >> 3, segs = w8 * h;       // w8: segments per row

  for (int i = tid; i < segs; i += T) {
    const int y = i / w8, x = (i - y * w8) * segw;
    if (seg4) {
      u32 v;
      __builtin_memcpy(&v, pic + (size_t)(pu.y + y) * pic_stride + pu.x + x, 4);
      *(u32 *)(s_cur + y * G::CS + x) = v;
    } else {
      uint2 v;
      __builtin_memcpy(&v, pic + (size_t)(pu.y + y) * pic_stride + pu.x + x, 8);
      *(uint2 *)(s_cur + y * G::CS + x) = v;
    }
  }

  int best_x = 0, best_y = 0;                          // info->best_mv, integer-pel here
  u32 best_cost = 0xffffffffu, best_bits = 0;

  // Exhaustive search only: the reference pixels of one (2R+1)^2 window, staged in LDS behind the current block
  // (the fractional stage's buffers are idle until the integer search is over) when they fit.
  u8 *const s_win = lds + G::P_BYTES + G::CUR_BYTES;
  constexpr int WIN_BYTES = G::TOTAL - (G::P_BYTES + G::CUR_BYTES);
  int win_cx = 0, win_cy = 0, win_R = 0, win_stride = 0;
  bool win_on = false;

  const bool seg_pow2 = (segs & (segs - 1)) == 0 && segs >= 8;
  const int run = segs < 64 ? segs : 64;

  // SADs of candidates 0 .. n-1 (offsets in sh->cx / cy, written by the caller) -> sh->sad
  auto group_sads = [&](int n) {
    if (tid < ME_GROUP) sh->sad[tid] = 0;
    sync();
    if (win_on) {
      // Exhaustive search, window in LDS: ONE LANE PER POSITION walks the block's row segments (the current block's
      // segment is the same address for every lane: an LDS broadcast), so a position's SAD never leaves its lane -- no
      // per-item index arithmetic, no cross-lane reduction, no atomics for a one-wave PU.  Wider workgroups split the
      // segments between their waves.
      const int k = tid & (ME_GROUP - 1), part = tid / ME_GROUP, nparts = T / ME_GROUP;
      if (k < n) {
        const int col0 = sh->cx[k] - win_cx + win_R, row0 = sh->cy[k] - win_cy + win_R;
        u32 acc = 0;
        for (int s = part; s < segs; s += nparts) {
          const int y = s / w8, x = (s - y * w8) * segw;
          const int col = col0 + x;
          const u32 *q = (const u32 *)(s_win + (row0 + y) * win_stride + (col & ~3));
          const u32 sh8 = (u32)col & 3u;
          if (seg4) {
            acc = __builtin_amdgcn_sad_u8(*(const u32 *)(s_cur + y * G::CS + x), __builtin_amdgcn_alignbyte(q[1], q[0], sh8), acc);
          } else {
            const uint2 c = *(const uint2 *)(s_cur + y * G::CS + x);
            const u32 d0 = q[0], d1 = q[1], d2 = q[2];
            acc = __builtin_amdgcn_sad_u8(c.x, __builtin_amdgcn_alignbyte(d1, d0, sh8), acc);
            acc = __builtin_amdgcn_sad_u8(c.y, __builtin_amdgcn_alignbyte(d2, d1, sh8), acc);
          }
        }
        if (nparts == 1) sh->sad[k] = acc; else atomicAdd(&sh->sad[k], acc);
      }
      sync();
      return;
    }
    for (int it = tid; it < n * segs; it += T) {
      const int k = it / segs, s = it - k * segs, y = s / w8, x = (s - y * w8) * segw;
      uint2 c, r;
      if (seg4) {
        c.x = *(const u32 *)(s_cur + y * G::CS + x); c.y = 0u; r.y = 0u;
        const int rx = pu.x + sh->cx[k] + x, ry = pu.y + sh->cy[k] + y;
        if (rx >= 0 && rx + 4 <= ref.w && ry >= 0 && ry < ref.h) {
          __builtin_memcpy(&r.x, ref.p + (size_t)ry * ref.stride + rx, 4);
        } else {
          r.x = (u32)ref_px(ref, rx, ry) | ((u32)ref_px(ref, rx + 1, ry) << 8) | ((u32)ref_px(ref, rx + 2, ry) << 16) |
                ((u32)ref_px(ref, rx + 3, ry) << 24);
        }
      } else {
        c = *(const uint2 *)(s_cur + y * G::CS + x);
        const int rx = pu.x + sh->cx[k] + x, ry = pu.y + sh->cy[k] + y;
        if (rx >= 0 && rx + 8 <= ref.w && ry >= 0 && ry < ref.h) {
          __builtin_memcpy(&r, ref.p + (size_t)ry * ref.stride + rx, 8);
        } else {
          u32 b[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) b[i] = ref_px(ref, rx + i, ry);
          r.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
          r.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        }
      }
      u32 v = __builtin_amdgcn_sad_u8(c.y, r.y, __builtin_amdgcn_sad_u8(c.x, r.x, 0u));
      if (seg_pow2) {
        // the lanes that share a candidate are an aligned run of min(segs, 64): add them up in registers first --
        // 32 lanes hitting one LDS address with an atomic serialise
        v = run == 8 ? group_sum<8>(v) : run == 16 ? group_sum<16>(v) : run == 32 ? group_sum<32>(v) : group_sum<64>(v);   // DPP up to 16 lanes
        if (((tid & 63) & (run - 1)) == 0) atomicAdd(&sh->sad[k], v);
      } else {
        atomicAdd(&sh->sad[k], v);
      }
    }
    sync();
  };
  // check_mv_cost (:195-232) over the evaluated group at once.  The reference walks the candidates in order and
  // replaces the best on a strictly smaller cost, i.e. it ends on the FIRST candidate that attains the group's minimum,
  // provided that minimum beats the incoming best -- and that candidate is also the last one that "improved", which is
  // what the patterns record as best_index.  Lane k prices candidate k; a wave-wide minimum and a ballot pick the
  // same winner.  Returns its index in the group, or -1 when nothing improved.  (Every wave of a workgroup computes
  // this redundantly from the same LDS values, so the result is uniform without another barrier.)
  auto decide = [&](int n) -> int {
    const int k = tid & 63;
    u32 cost = 0xffffffffu, bits = 0;
    int x = 0, y = 0;
    if (k < n) {
      x = sh->cx[k]; y = sh->cy[k];
      if (mvc.within(x * 4, y * 4)) cost = sh->sad[k] + mvc.cost(x, y, 2, bits);   // < 2^32: lambda_cost is bounded by the entry
    }
    u32 m = cost;
    if (n <= 16) {
      // every pattern but the exhaustive search: the candidates sit in lanes 0..15, one DPP row -- four v_min with
      // DPP operands instead of six LDS-crossbar exchanges
      u32 o;
      o = dpp_mov<0xB1>(m); m = o < m ? o : m;             // quad_perm [1,0,3,2]
      o = dpp_mov<0x4E>(m); m = o < m ? o : m;             // quad_perm [2,3,0,1]
      o = dpp_mov<0x141>(m); m = o < m ? o : m;            // row_half_mirror
      o = dpp_mov<0x140>(m); m = o < m ? o : m;            // row_mirror
      m = (u32)__builtin_amdgcn_readfirstlane((int)m);
    } else {
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const u32 o = (u32)__shfl_xor((int)m, off, 64);
        m = o < m ? o : m;
      }
    }
    if (m >= best_cost) return -1;
    const int win = __builtin_ctzll(__ballot(cost == m));
    best_x = __shfl(x, win, 64); best_y = __shfl(y, win, 64);
    best_cost = m; best_bits = (u32)__shfl((int)bits, win, 64);
    return win;
  };
  auto set_cand = [&](int k, int x, int y) { if (tid == 0) { sh->cx[k] = x; sh->cy[k] = y; } };

  bool done = false;
  if (BOTH && prm.algorithm == 3) {
    // the search service: the whole workgroup has already walked the windows (full_search_wg, below); its outcome waits in sh
    best_x = sh->cx[FULL_OUT]; best_y = sh->cy[FULL_OUT];
    best_cost = sh->sad[FULL_OUT]; best_bits = sh->sad[FULL_OUT + 1];
    done = true;
  } else if (prm.algorithm == 3) {
    // ---- search_mv_full (:886-962): the windows around the zero vector, extra_mv and the merge candidates, in the
    // reference's visiting order, ME_GROUP positions per round ----
    const int R = prm.search_range;
    int n = 0;
    sync();
    auto flush = [&]() {
      if (n > 0) {
        group_sads(n);
        decide(n);
        n = 0;
        sync();
      }
    };
    auto push = [&](int x, int y) {
      set_cand(n++, x, y);
      if (n == ME_GROUP) flush();
    };
    // a window's (w + 2R) x (h + 2R) reference pixels (edge replicated, image.c:320-444) go to LDS when they fit
    auto begin_window = [&](int cx, int cy) {
      flush();
      const int stride = ((w + 2 * R + 3) & ~3) + 4, rows = h + 2 * R;
      win_on = stride * rows <= WIN_BYTES;
      if (win_on) {
        win_cx = cx; win_cy = cy; win_R = R; win_stride = stride;
        const int x0 = pu.x + cx - R, y0 = pu.y + cy - R, wq = stride >> 2;
        for (int i = tid; i < wq * rows; i += T) {
          const int y = i / wq, q = (i - y * wq) * 4;
          u32 v;
          if (x0 + q >= 0 && x0 + q + 4 <= ref.w && y0 + y >= 0 && y0 + y < ref.h) {
            __builtin_memcpy(&v, ref.p + (size_t)(y0 + y) * ref.stride + x0 + q, 4);
          } else {
            v = (u32)ref_px(ref, x0 + q, y0 + y) | ((u32)ref_px(ref, x0 + q + 1, y0 + y) << 8) |
                ((u32)ref_px(ref, x0 + q + 2, y0 + y) << 16) | ((u32)ref_px(ref, x0 + q + 3, y0 + y) << 24);
          }
          *(u32 *)(s_win + y * stride + q) = v;
        }
        sync();
      }
    };
    begin_window(0, 0);
    for (int y = -R; y <= R; ++y)
      for (int x = -R; x <= R; ++x) push(x, y);
    const int ex = pu.extra_mv[0] >> 2, ey = pu.extra_mv[1] >> 2;
    if (!mvc.in_merge(ex, ey)) {
      begin_window(ex, ey);
      for (int y = -R; y <= R; ++y)
        for (int x = -R; x <= R; ++x) push(ex + x, ey + y);
    }
#pragma unroll                                              // i and j are unrolled so that mx[] / my[] stay in registers
    for (int i = 0; i < 5; ++i) {
      if (!(mvc.usable >> i & 1u)) continue;
      const int cx0 = mvc.mx[i] >> 2, cy0 = mvc.my[i] >> 2;        // plain shift here (:917-920)
      if (cx0 == 0 && cy0 == 0) continue;
      begin_window(cx0, cy0);
      for (int y = cy0 - R; y <= cy0 + R; ++y)
        for (int x = cx0 - R; x <= cx0 + R; ++x) {
          if (!mvc.within(x * 4, y * 4)) continue;
          bool tested = false;
#pragma unroll
          for (int j = -1; j < 4; ++j) {
            if (j >= i || tested) continue;
            int xx = 0, yy = 0;
            if (j >= 0) {
              if (!(mvc.usable >> j & 1u)) continue;
              xx = mvc.mx[j >= 0 ? j : 0] >> 2; yy = mvc.my[j >= 0 ? j : 0] >> 2;
            }
            if (x >= xx - R && x <= xx + R && y >= yy - R && y <= yy + R) {
              tested = true;
              x = xx + R;                                          // jump past the earlier window (:948)
            }
          }
          if (!tested) push(x, y);
        }
    }
    flush();
    win_on = false;
    done = true;
  }

  // ---- select_starting_point (:282-307) ----
  int n = 0;
  sync();                                              // s_cur complete; previous readers of sh are done
  if (!done) {
  set_cand(n++, 0, 0);
  {
    const int ex = pu.extra_mv[0] >> 2, ey = pu.extra_mv[1] >> 2;
    if ((ex != 0 || ey != 0) && !mvc.in_merge(ex, ey)) set_cand(n++, ex, ey);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (!(mvc.usable >> i & 1u)) continue;
      const int x = (mvc.mx[i] + 2) >> 2, y = (mvc.my[i] + 2) >> 2;
      if (x == 0 && y == 0) continue;
      set_cand(n++, x, y);
    }
  }
  group_sads(n);
  decide(n);
  }

  // ---- early_terminate (:415-460) ----
  if (!done && prm.early_termination) {
    int mvx = best_x, mvy = best_y, first = 0, last = 3;
    for (int round = 0; round < 2 && !done; ++round) {
      const double threshold = prm.early_termination == 2 ? (double)best_cost * 0.95 : (double)best_cost;
      sync();
      for (int i = first; i <= last; ++i) set_cand(i - first, mvx + c_et_hex[i][0], mvy + c_et_hex[i][1]);
      group_sads(last - first + 1);
      const int hit = decide(last - first + 1);
      const int best_index = hit >= 0 ? first + hit : 6;
      mvx += c_et_hex[best_index][0]; mvy += c_et_hex[best_index][1];
      if ((double)best_cost >= threshold) done = true;
      first = (best_index + 3) % 4;
      last = first + 2;
    }
  }

  if (!done && prm.algorithm == 2) {
    // ---- tz_search (:595-672): search range 96, 8-point diamond patterns (4 points at distance 1), no raster
    // scan, star refinement; kvz_tz_pattern_search (:463-577) is one group ----
    int best_dist = 0;
    auto pattern = [&](int dist, int sx, int sy) {
      const int hd = dist / 2, n = dist == 1 ? 4 : 8;
      sync();
      set_cand(0, sx, sy + dist); set_cand(1, sx + dist, sy); set_cand(2, sx, sy - dist); set_cand(3, sx - dist, sy);
      if (n == 8) {
        set_cand(4, sx + hd, sy + hd); set_cand(5, sx + hd, sy - hd); set_cand(6, sx - hd, sy - hd); set_cand(7, sx - hd, sy + hd);
      }
      group_sads(n);
      if (decide(n) >= 0) best_dist = dist;
    };
    int sx = best_x, sy = best_y, rounds = 0;
    for (int dist = 1; dist <= 96; dist *= 2) {
      pattern(dist, sx, sy);
      if (best_dist != dist) rounds++;
      if (rounds >= 3) break;
    }
    if (sx != 0 || sy != 0) {
      rounds = 0;
      for (int dist = 1; dist <= 48; dist *= 2) {
        pattern(dist, 0, 0);
        if (best_dist != dist) rounds++;
        if (rounds >= 3) break;
      }
    }
    while (best_dist > 0) {
      best_dist = 0;
      sx = best_x; sy = best_y;
      for (int dist = 1; dist <= 96; dist *= 2) pattern(dist, sx, sy);
    }
  } else if (!done && prm.algorithm == 1) {
    // ---- diamond_search (:826-882) ----
    int mvx = best_x, mvy = best_y, best_index = 4;
    u32 steps = prm.max_steps;
    sync();
    for (int i = 0; i < 5; ++i) set_cand(i, mvx + c_diamond[i][0], mvy + c_diamond[i][1]);
    group_sads(5);
    {
      const int hit = decide(5);
      if (hit >= 0) best_index = hit;
    }
    if (best_index != 4) {
      mvx += c_diamond[best_index][0]; mvy += c_diamond[best_index][1];
      int from_dir = 4;
      bool better;
      do {
        better = false;
        if (steps > 0) steps -= 1;
        sync();
        int n = 0, idx[4];
        for (int i = 0; i < 4; ++i) {
          if (i == from_dir) continue;                  // where we came from is checked already
          idx[n] = i;
          set_cand(n++, mvx + c_diamond[i][0], mvy + c_diamond[i][1]);
        }
        group_sads(n);
        const int hit = decide(n);
        if (hit >= 0) { best_index = hit == 0 ? idx[0] : (hit == 1 ? idx[1] : (hit == 2 ? idx[2] : idx[3])); better = true; }
        if (better) {
          mvx += c_diamond[best_index][0]; mvy += c_diamond[best_index][1];
          from_dir = best_index ^ 3;
        }
      } while (better && steps != 0);
    }
  } else if (!done) {
    // ---- the hexagon (:723-777) ----
    int mvx = best_x, mvy = best_y, best_index = 0;
    u32 steps = prm.max_steps;
    sync();
    for (int i = 1; i < 7; ++i) set_cand(i - 1, mvx + c_large_hex[i][0], mvy + c_large_hex[i][1]);
    group_sads(6);
    {
      const int hit = decide(6);
      if (hit >= 0) best_index = hit + 1;
    }
    while (best_index != 0 && steps != 0) {
      steps -= 1;
      const int start = best_index == 1 ? 6 : (best_index == 8 ? 1 : best_index - 1);
      mvx += c_large_hex[best_index][0]; mvy += c_large_hex[best_index][1];
      best_index = 0;
      sync();
      for (int i = 0; i < 3; ++i) set_cand(i, mvx + c_large_hex[start + i][0], mvy + c_large_hex[start + i][1]);
      group_sads(3);
      const int hit = decide(3);
      if (hit >= 0) best_index = start + hit;
    }
    sync();
    for (int i = 1; i < 9; ++i) set_cand(i - 1, mvx + c_small_hex[i][0], mvy + c_small_hex[i][1]);
    group_sads(8);
    decide(8);
  }

  // ---- search_frac, or the SATD re-cost of :1236-1248 when cfg.fme_level == 0 ----
  int mv_x = best_x * 4, mv_y = best_y * 4;
  const u32 int_cost = best_cost, int_bits = best_bits;
  u32 cost0 = 0xffffffffu;
  if (best_cost != 0xffffffffu) {
    sync();
    const kvz_hip_block_pair d = { pu.x, pu.y, pu.x + best_x, pu.y + best_y, w, h };
    // :1239: the fractional search only if the integer result beats what the pictures searched before reached
    const u32 beat = (!BOTH && prm.cost_to_beat) ? prm.cost_to_beat[pu_index] : 0xffffffffu;
    const int level = __builtin_amdgcn_readfirstlane(best_cost < beat ? prm.fme_level : 0);   // the same in every lane: keep the level's branches scalar
    const frac_result fr = search_frac_core<MAXW, T, WAVE, me_cost_model_t<RDO, CONSTR>, FW, FH>(tid, lds, pic, pic_stride, ref, d, level, mvc, (u32 *)nullptr, (i32 *)nullptr);
    best_cost = fr.cost;                               // level 0: satd + bits(int mv) * lambda, the same bits as best_bits
    cost0 = fr.cost0;
    if (level > 0) { mv_x = fr.mvx; mv_y = fr.mvy; best_bits = fr.bitcost; }
  }

  if (BOTH) {
    if (tid == 0) {
      serve_result *so = reinterpret_cast<serve_result *>(out);
      kvz_hip_me_result r;
      u32 unused;
      r.mv[0] = mv_x; r.mv[1] = mv_y;
      r.cost = best_cost; r.bitcost = best_bits;
      int m = mvc.merge_match(mv_x, mv_y);
      r.merged = m >= 0;
      r.merge_idx = m >= 0 ? m : mvc.n_merge;
      r.mv_cand = m >= 0 ? 0 : mvc.select_cand(mv_x, mv_y, unused);
      r.reserved = 0;
      so->frac = r;
      r.mv[0] = best_x * 4; r.mv[1] = best_y * 4;          // :1242-1252: the integer vector, SATD + its bits
      r.cost = cost0; r.bitcost = int_bits;
      m = mvc.merge_match(best_x * 4, best_y * 4);
      r.merged = m >= 0;
      r.merge_idx = m >= 0 ? m : mvc.n_merge;
      r.mv_cand = m >= 0 ? 0 : mvc.select_cand(best_x * 4, best_y * 4, unused);
      so->integer = r;
      so->integer_search_cost = int_cost;
      // the caller polls `done` in page-locked host memory: results first, system-wide, then the flag
      __threadfence_system();
      __hip_atomic_store(&so->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }

  if (tid == 0) {
    kvz_hip_me_result r;
    r.mv[0] = mv_x; r.mv[1] = mv_y;
    r.cost = best_cost; r.bitcost = best_bits;
    const int m = mvc.merge_match(mv_x, mv_y);          // :1253-1266
    r.merged = m >= 0;
    r.merge_idx = m >= 0 ? m : mvc.n_merge;
    u32 unused;
    r.mv_cand = m >= 0 ? 0 : (RDO ? mvc.select_cand_cabac(mv_x, mv_y) : mvc.select_cand(mv_x, mv_y, unused));   // :1268-1273
    r.reserved = 0;
    *out = r;
  }
}

__device__ __forceinline__ bool pu_ok(const kvz_hip_me_pu &pu, int pic_w, int pic_h)
{
  return frac_shape_ok(pu.width, pu.height) && pu.x >= 0 && pu.y >= 0 && pu.x + pu.width <= pic_w && pu.y + pu.height <= pic_h &&
         pu.num_merge_cand >= 0 && pu.num_merge_cand <= 5;
}

__device__ __forceinline__ void flag_bad(kvz_hip_me_result *out)
{
  kvz_hip_me_result r = { { 0, 0 }, 0xffffffffu, 0, 0, 0, 0, -1 };
  *out = r;
}

// size class of a PU as kvz_hip_me_params.size_classes names them: 1 = up to 16x16, 2 = up to 32x32, 4 = larger
__device__ __forceinline__ int pu_class(const kvz_hip_me_pu &pu)
{
  return (pu.width > 32 || pu.height > 32) ? 4 : ((pu.width > 16 || pu.height > 16) ? 2 : 1);
}
// A launch with a size-class hint starts only the kernels of the classes named, so a PU of another class is searched by
// no kernel: the kernel of the lowest class named flags it (cost 0xFFFFFFFF, reserved -1) on its way past.
__device__ __forceinline__ bool pu_orphan(int cls, int mine, int hinted) { return !(hinted & cls) && mine == (hinted & -hinted); }

// kvz_hip_search_pu_multi_batch: several pictures of one size in a launch (frames in flight, tiles' or instances' pictures).
// The `pic` / `ref.p` arguments are then DEVICE TABLES of plane pointers, a PU names its pair in pad >> 2, the table length
// travels in prm.n_cabac (unused without mv_rdo).  A template flag, so that the one-picture kernels compile exactly as before.
template <bool MULTI>
__device__ __forceinline__ bool pick_planes(const u8 *__restrict__ &pic, refplane_t &ref, const kvz_hip_me_pu &pu, int n_planes)
{
  if (!MULTI) return true;
  const int k = pu.pad >> 2;
  if (k < 0 || k >= n_planes) return false;
  pic = reinterpret_cast<const u8 *const *>(pic)[k];
  ref.p = reinterpret_cast<const u8 *const *>(ref.p)[k];
  return true;
}

// PUs larger than 32x32 in either direction (and malformed descriptors, which are flagged): one workgroup (T threads) per PU
template <int T, bool CONSTR, bool MULTI = false>
__global__ __launch_bounds__(T) void search_pu_big_kernel(const u8 *__restrict__ pic, u32 pic_stride, int pic_w, int pic_h, refplane_t ref,
                                                            const kvz_hip_me_pu *__restrict__ pus, kvz_hip_me_params prm,
                                                            kvz_hip_me_result *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) u8 lds[frac_geom<64>::TOTAL];
  __shared__ me_shared sh;
  const kvz_hip_me_pu &pu = pus[blockIdx.x];            // uniform address: the compiler reads it with scalar loads
  if (!pu_ok(pu, pic_w, pic_h) || !pick_planes<MULTI>(pic, ref, pu, prm.n_cabac)) { if (threadIdx.x == 0) flag_bad(out + blockIdx.x); return; }
  const int cls = pu_class(pu);
  if (cls != 4) {                                       // the one-wave-per-PU kernels'
    if (pu_orphan(cls, 4, prm.size_classes) && threadIdx.x == 0) flag_bad(out + blockIdx.x);
    return;
  }
  search_pu_core<64, T, false, 0, 0, false, CONSTR>(threadIdx.x, lds, &sh, pic, pic_stride, ref, pu, prm, out + blockIdx.x, blockIdx.x);
}

// --mv-rdo (cfg.mv_rdo, off in every preset): MV bits from the CABAC model.  One workgroup per PU for every size -- a
// correctness path; the model walks probability tables per candidate and is kept out of the kernels above so that their
// register budget is untouched.
__global__ __launch_bounds__(256) void search_pu_rdo_kernel(const u8 *__restrict__ pic, u32 pic_stride, int pic_w, int pic_h, refplane_t ref,
                                                            const kvz_hip_me_pu *__restrict__ pus, kvz_hip_me_params prm,
                                                            kvz_hip_me_result *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) u8 lds[frac_geom<64>::TOTAL];
  __shared__ me_shared sh;
  const kvz_hip_me_pu &pu = pus[blockIdx.x];
  if (!pu_ok(pu, pic_w, pic_h) || pu.reserved < 0 || pu.reserved >= prm.n_cabac) { if (threadIdx.x == 0) flag_bad(out + blockIdx.x); return; }   // a stale snapshot index is flagged, never dereferenced
  search_pu_core<64, 256, false, 0, 0, true>(threadIdx.x, lds, &sh, pic, pic_stride, ref, pu, prm, out + blockIdx.x, blockIdx.x);
}

// PUs up to 16x16: one wave per PU, four PUs per workgroup, wave-private LDS, no barrier.  The register budget is held at
// 6 waves per SIMD (80 VGPRs): at 82 the kernel dropped to 5 and lost 8 % (measured A/B on one box).
template <bool CONSTR, bool MULTI = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void search_pu_small_kernel(const u8 *__restrict__ pic, u32 pic_stride, int pic_w, int pic_h, refplane_t ref,
                                                              const kvz_hip_me_pu *__restrict__ pus, size_t count, kvz_hip_me_params prm,
                                                              kvz_hip_me_result *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) u8 lds[4][(frac_geom<16>::TOTAL + 15) & ~15];
  __shared__ me_shared sh[4];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t i = (size_t)blockIdx.x * 4 + wv;
  if (i >= count) return;
  const kvz_hip_me_pu &pu = pus[i];
  const int lane = threadIdx.x & 63;
  if (!pu_ok(pu, pic_w, pic_h) || !pick_planes<MULTI>(pic, ref, pu, prm.n_cabac)) { if (lane == 0) flag_bad(out + i); return; }
  const int cls = pu_class(pu);
  if (cls != 1) {
    if (pu_orphan(cls, 1, prm.size_classes) && lane == 0) flag_bad(out + i);
    return;
  }
  if (pu.width == 8 && pu.height == 8) search_pu_core<16, 64, true, 8, 8, false, CONSTR>(lane, lds[wv], &sh[wv], pic, pic_stride, ref, pu, prm, out + i, i);
  else if (pu.width == 16 && pu.height == 16) search_pu_core<16, 64, true, 16, 16, false, CONSTR>(lane, lds[wv], &sh[wv], pic, pic_stride, ref, pu, prm, out + i, i);
  else search_pu_core<16, 64, true, 0, 0, false, CONSTR>(lane, lds[wv], &sh[wv], pic, pic_stride, ref, pu, prm, out + i, i);
}

// PUs up to 32x32 that are not the small kernel's: one wave per PU as well (two per workgroup: 15 KiB of LDS each).
// With a workgroup per PU a 32x32 search spent its time in barriers around little work per thread (11 M PUs/s).
template <bool CONSTR>
__global__ __launch_bounds__(128) void search_pu_medium_kernel(const u8 *__restrict__ pic, u32 pic_stride, int pic_w, int pic_h, refplane_t ref,
                                                               const kvz_hip_me_pu *__restrict__ pus, size_t count, kvz_hip_me_params prm,
                                                               kvz_hip_me_result *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) u8 lds[2][(frac_geom<32>::TOTAL + 15) & ~15];
  __shared__ me_shared sh[2];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t i = (size_t)blockIdx.x * 2 + wv;
  if (i >= count) return;
  const kvz_hip_me_pu &pu = pus[i];
  const int lane = threadIdx.x & 63;
  if (!pu_ok(pu, pic_w, pic_h)) { if (lane == 0) flag_bad(out + i); return; }
  const int cls = pu_class(pu);
  if (cls != 2) {
    if (pu_orphan(cls, 2, prm.size_classes) && lane == 0) flag_bad(out + i);
    return;
  }
  if (pu.width == 32 && pu.height == 32) search_pu_core<32, 64, true, 32, 32, false, CONSTR>(lane, lds[wv], &sh[wv], pic, pic_stride, ref, pu, prm, out + i, i);
  else search_pu_core<32, 64, true, 0, 0, false, CONSTR>(lane, lds[wv], &sh[wv], pic, pic_stride, ref, pu, prm, out + i, i);
}

// The same class with one workgroup of T threads per PU: lower latency per search (more lanes on each step, barriers
// instead of wave-local fences), lower throughput -- for batches too small to fill the chip with one wave per PU.
template <int T, bool CONSTR, bool MULTI = false>
__global__ __launch_bounds__(T) void search_pu_medium_wg_kernel(const u8 *__restrict__ pic, u32 pic_stride, int pic_w, int pic_h, refplane_t ref,
                                                                const kvz_hip_me_pu *__restrict__ pus, kvz_hip_me_params prm,
                                                                kvz_hip_me_result *__restrict__ out)
{
  __shared__ __attribute__((aligned(16))) u8 lds[(frac_geom<32>::TOTAL + 15) & ~15];
  __shared__ me_shared sh;
  const kvz_hip_me_pu &pu = pus[blockIdx.x];
  if (!pu_ok(pu, pic_w, pic_h) || !pick_planes<MULTI>(pic, ref, pu, prm.n_cabac)) { if (threadIdx.x == 0) flag_bad(out + blockIdx.x); return; }
  const int cls = pu_class(pu);
  if (cls != 2) {
    if (pu_orphan(cls, 2, prm.size_classes) && threadIdx.x == 0) flag_bad(out + blockIdx.x);
    return;
  }
  if (pu.width == 32 && pu.height == 32) search_pu_core<32, T, false, 32, 32, false, CONSTR>(threadIdx.x, lds, &sh, pic, pic_stride, ref, pu, prm, out + blockIdx.x, blockIdx.x);
  else search_pu_core<32, T, false, 0, 0, false, CONSTR>(threadIdx.x, lds, &sh, pic, pic_stride, ref, pu, prm, out + blockIdx.x, blockIdx.x);
}

// ---- the search service's kernels (serve.hip): one (PU, reference picture) unit each, descriptor + parameters read from
// page-locked host memory ONCE into registers (every later access would be another trip over PCIe), planes resident in
// slots of one device allocation, results written straight back to the caller's page-locked area ----
__device__ __forceinline__ void serve_flag_bad(serve_result *so)
{
  kvz_hip_me_result r = { { 0, 0 }, 0xffffffffu, 0, 0, 0, 0, -1 };
  so->frac = r; so->integer = r; so->integer_search_cost = 0xffffffffu;
  __threadfence_system();
  __hip_atomic_store(&so->done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool serve_unit_ok(const serve_unit &u, int pic_w, int pic_h, int n_slots)
{
  return pu_ok(u.pu, pic_w, pic_h) && u.pic_slot >= 0 && u.pic_slot < n_slots && u.ref_slot >= 0 && u.ref_slot < n_slots;
}

// One launch per batch whatever the PU sizes in it (the launch path is what the callers queue for: one command instead of one
// per size class).  A workgroup of 512 threads takes one unit; the waves its size class does not need leave at once --
// s_barrier counts only the waves of a workgroup that have not terminated -- so a PU up to 16x16 is searched by one wave
// with wave-local fences, one up to 32x32 by 128 threads, a larger one by all 512: the thread counts of the batched kernels above.
template <bool CONSTR, bool QSAD>
__global__ __launch_bounds__(512) void serve_kernel(const u8 *__restrict__ planes, size_t plane_bytes, int n_slots, u32 stride, int pic_w, int pic_h,
                                                    const serve_unit *__restrict__ units, int count)
{
  __shared__ __attribute__((aligned(16))) u8 lds[frac_geom<64>::TOTAL];
  __shared__ me_shared sh;
  if ((int)blockIdx.x >= count) return;
  const serve_unit u = units[blockIdx.x];
  serve_result *so = reinterpret_cast<serve_result *>(u.result);
  if (!serve_unit_ok(u, pic_w, pic_h, n_slots)) { if (threadIdx.x == 0) serve_flag_bad(so); return; }
  const int cls = pu_class(u.pu);
  const int tid = threadIdx.x;
  const u8 *pic = planes + (size_t)u.pic_slot * plane_bytes;
  const refplane_t ref = { planes + (size_t)u.ref_slot * plane_bytes, stride, pic_w, pic_h };
  if (u.prm.algorithm == 3) {                            // the exhaustive search: every wave on the positions, then the class's waves go on
    full_search_wg<CONSTR, QSAD, 512>(tid, lds, (int)sizeof(lds), &sh, pic, stride, ref, u.pu, u.prm);
    __syncthreads();
  }
  if (tid >= (cls == 1 ? 64 : (cls == 2 ? 128 : 512))) return;
  kvz_hip_me_result *out = reinterpret_cast<kvz_hip_me_result *>(so);
  if (cls == 1) {
    if (u.pu.width == 8 && u.pu.height == 8) search_pu_core<16, 64, true, 8, 8, false, CONSTR, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
    else if (u.pu.width == 16 && u.pu.height == 16) search_pu_core<16, 64, true, 16, 16, false, CONSTR, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
    else search_pu_core<16, 64, true, 0, 0, false, CONSTR, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
  } else if (cls == 2) {
    if (u.pu.width == 32 && u.pu.height == 32) search_pu_core<32, 128, false, 32, 32, false, CONSTR, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
    else search_pu_core<32, 128, false, 0, 0, false, CONSTR, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
  } else {
    search_pu_core<64, 512, false, 0, 0, false, CONSTR, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
  }
}

// ---- resident workers ----
// A workgroup that stays on the device and takes units from the ring by ticket (protocol: serve.hip).  Every wait in here ends on a
// wall-clock limit, so the grid always drains: no work for linger_ticks, the first idle moment after life_ticks, ctl->quit, or -- a
// slot that is not written within a second of its ticket being published (never observed; the host publishes after writing) -- failure.
__device__ __forceinline__ unsigned long long sys_load64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ u32 sys_load32(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void sys_store32(u32 *p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

constexpr unsigned long long SERVE_NO_TICKET = ~0ull;
constexpr unsigned long long SERVE_POLL_TICKS = 50;          // ctl->tail is read across PCIe at most every 0.5 us, by one worker at a time
constexpr unsigned long long SERVE_SLOT_LIMIT_TICKS = 100000000ull;     // 1 s

// thread 0 of a worker: the next ticket, or SERVE_NO_TICKET when it is time to leave (alive[me] is 0 by then).
// ctl->tail lives in host memory; a worker reads it across PCIe every poll_period ticks (the period is the workers' number x 0.5 us
// and their phases are spread, so SOMEBODY reads it every 0.5 us) and mirrors it in device memory, where everybody looks.
__device__ __forceinline__ unsigned long long serve_take_ticket(serve_ring_ctl *ctl, const serve_push *push, serve_ring_dev *dev, u32 me, unsigned long long born,
                                                                unsigned long long linger_ticks, unsigned long long life_ticks,
                                                                unsigned long long poll_period, unsigned long long &next_poll, bool &retired)
{
  const unsigned long long idle0 = wall_clock64();
  bool quit = false;
  for (;;) {
    unsigned long long now = wall_clock64();
    // Device memory is cached in the L2 of the XCD that reads it, and the eight L2s are only made coherent at kernel boundaries: a
    // plain (even agent-scope) load in this loop may return the same stale line for as long as the kernel runs -- measured: with
    // loads only, a handful of the workers ever saw a unit.  Read-modify-write atomics are performed at the memory, so the tail is
    // read with the atomicMax that also publishes what the host said; a stale head only costs a compare-and-swap that fails and
    // returns the fresh one.
    if (!retired && now - born > life_ticks) {             // end of life: said at once, busy or not (see the leaving protocol below)
      sys_store32(&ctl->alive[me], 0u);
      __threadfence_system();
      retired = true;
    }
    unsigned long long from_host = 0;
    if (retired) {
      from_host = sys_load64(&ctl->tail);                  // behind the store of alive[me] = 0 on the way to the host: see below
    } else if (now >= next_poll) {
      if (push) {                                          // the host's copy in device memory: no PCIe read
        from_host = sys_load64(&push->tail);
        quit = sys_load32(&push->quit) != 0u;
      } else {
        from_host = sys_load64(&ctl->tail);
        quit = sys_load32(&ctl->quit) != 0u;
      }
      next_poll = now + poll_period;
    }
    unsigned long long tail = atomicMax(&dev->tail, from_host);
    if (from_host > tail) tail = from_host;
    const unsigned long long head = __hip_atomic_load(&dev->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // a compare-and-swap that loses returns the head it lost to: the next try needs no reload (many idle workers see the same unit)
    for (unsigned long long h = head; h < tail;) {
      const unsigned long long seen = atomicCAS(&dev->head, h, h + 1);
      if (seen == h) {
        now = wall_clock64();
        atomicMax(&dev->last_claim, now);
        atomicAdd(&dev->backlog, tail - h - 1); atomicAdd(&dev->idle_ticks, now - idle0);
        return h;
      }
      h = seen;
    }
    if (retired) return SERVE_NO_TICKET;
    // Idle means nobody has taken a ticket for linger_ticks, not "not me": the workers of a launch leave together (a kernel ends when
    // its last workgroup does, and the next launch on its stream waits for that), and so they do at the end of their life.
    bool idle = false;
    if (now - idle0 > linger_ticks) {
      const unsigned long long last_claim = atomicMax(&dev->last_claim, 0ull);
      idle = now < last_claim || now - last_claim > linger_ticks;
    }
    if (idle || quit) {
      // Leaving.  The host publishes a unit FIRST and looks at alive[] AFTERWARDS; this side says "gone" first and looks for work
      // afterwards, with a read that cannot overtake the store on its way to host memory.  So either the host sees the 0 and starts
      // a worker, or the read above sees the unit -- and then it is taken here (being uncounted while still working is harmless).
      sys_store32(&ctl->alive[me], 0u);
      __threadfence_system();
      retired = true;
      continue;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

template <bool QSAD>
__global__ __launch_bounds__(512) void serve_worker_kernel(const u8 *__restrict__ planes, size_t plane_bytes, int n_slots, u32 stride, int pic_w, int pic_h,
                                                           serve_slot *ring, serve_slot *host_ring, const serve_push *push, u32 ring_mask, serve_ring_ctl *ctl, serve_ring_dev *dev,
                                                           serve_worker_ids ids, unsigned long long linger_ticks, unsigned long long life_ticks, unsigned long long poll_period)
{
  __shared__ __attribute__((aligned(16))) u8 lds[frac_geom<64>::TOTAL];
  __shared__ me_shared sh;
  __shared__ u32 s_unit[sizeof(serve_unit) / 4];
  __shared__ unsigned long long s_ticket;
  const int tid = threadIdx.x;
  const u32 me = ids.id[blockIdx.x];
  const unsigned long long born = wall_clock64();
  __shared__ u32 s_fail;
  __shared__ unsigned long long s_claimed;
  bool retired = false;                                     // thread 0's: alive[me] is 0, the host no longer counts this worker
  unsigned long long next_poll = born + (unsigned long long)me * SERVE_POLL_TICKS;      // thread 0's
  int served_retired = 0;
  for (;;) {
    if (tid == 0) {
      // A worker that has said it is gone (end of life, or idle) looks once more and serves what it finds, twice at most, then goes
      // without looking: the callers that wait keep starting the workers that are missing (serve.hip), and this one is not among
      // the counted.  Its kernel must END -- the next launch on the same hardware queue starts only then (measured: a few workers
      // that never found an idle moment kept a whole new crowd waiting behind them).
      unsigned long long t = SERVE_NO_TICKET;
      if (!(retired && served_retired >= 2)) {
        t = serve_take_ticket(ctl, push, dev, me, born, linger_ticks, life_ticks, poll_period, next_poll, retired);
        if (t != SERVE_NO_TICKET && retired) ++served_retired;
      }
      s_ticket = t;
      s_fail = 0u;
      s_claimed = wall_clock64();
    }
    __syncthreads();
    const unsigned long long ticket = s_ticket;
    if (ticket == SERVE_NO_TICKET) return;
    serve_slot *slot = ring + (ticket & ring_mask);
    // the slot was written before its ticket was published: unit and sequence word come in one pass; the retry is a guard
    if (tid < 64) {
      const unsigned long long t0 = wall_clock64();
      constexpr int UNIT_DWORDS = (int)(sizeof(serve_unit) / 4);
      for (;;) {
        u32 v = 0;
        if (tid <= UNIT_DWORDS) v = sys_load32(reinterpret_cast<const u32 *>(slot) + tid);      // dword UNIT_DWORDS is slot->seq
        const u32 seq = (u32)__shfl((int)v, UNIT_DWORDS, 64);
        if (seq == serve_seq(ticket)) {
          if (tid < UNIT_DWORDS) s_unit[tid] = v;
          break;
        }
        if (wall_clock64() - t0 > SERVE_SLOT_LIMIT_TICKS) {
          if (tid == 0) { sys_store32(&ctl->failed, 1u); sys_store32(&ctl->alive[me], 0u); __threadfence_system(); s_fail = 1u; }
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      // What the previous units' pictures left in this CU's vector cache may be older than an upload that was finished before this
      // unit was posted: one wave drops it (what a kernel boundary would have done) before the others are let through the barrier.
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (s_fail) return;
    unsigned long long t_fetched = 0;
    if (tid == 0) {
      sys_store32(&host_ring[ticket & ring_mask].seq, 0u);   // the host may write the slot again (its handshake word is the host ring's)
      t_fetched = wall_clock64();
    }
    serve_unit u;
    {
      u32 *d = reinterpret_cast<u32 *>(&u);
#pragma unroll
      for (int i = 0; i < (int)(sizeof(serve_unit) / 4); ++i) d[i] = (u32)__builtin_amdgcn_readfirstlane((int)s_unit[i]);   // uniform: keep it in SGPRs
    }
    serve_result *so = reinterpret_cast<serve_result *>(u.result);
    if (tid == 0) { const unsigned long long c = s_claimed; so->pad[0] = (u32)c; so->pad[1] = (u32)(c >> 32); }   // when the ticket was taken (statistics)
    if (!serve_unit_ok(u, pic_w, pic_h, n_slots)) {
      if (tid == 0) serve_flag_bad(so);
    } else {
      const int cls = pu_class(u.pu);
      const u8 *pic = planes + (size_t)u.pic_slot * plane_bytes;
      const refplane_t ref = { planes + (size_t)u.ref_slot * plane_bytes, stride, pic_w, pic_h };
      kvz_hip_me_result *out = reinterpret_cast<kvz_hip_me_result *>(so);
      if (u.prm.algorithm == 3) {
        full_search_wg<true, QSAD, 512>(tid, lds, (int)sizeof(lds), &sh, pic, stride, ref, u.pu, u.prm);
        __syncthreads();
      }
      if (cls == 1) {                                        // one wave, wave-local fences; the others wait at the barrier below
        if (tid < 64) {
          if (u.pu.width == 8 && u.pu.height == 8) search_pu_core<16, 64, true, 8, 8, false, true, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
          else if (u.pu.width == 16 && u.pu.height == 16) search_pu_core<16, 64, true, 16, 16, false, true, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
          else search_pu_core<16, 64, true, 0, 0, false, true, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
        }
      } else {
        search_pu_core<64, 512, false, 0, 0, false, true, true>(tid, lds, &sh, pic, stride, ref, u.pu, u.prm, out, 0);
      }
    }
    __syncthreads();                                         // lds, sh and s_unit are free again
    if (tid == 0) {
      const unsigned long long t_claimed = s_claimed, now = wall_clock64();
      atomicAdd(&dev->fetch_ticks, t_fetched - t_claimed); atomicAdd(&dev->busy_ticks, now - t_claimed); atomicAdd(&dev->units_served, 1ull);
    }
  }
}

}  // namespace

int kvzhip::serve_workers_launch_push(const u8 *planes, size_t plane_bytes, int n_slots, u32 stride, int w, int h, serve_slot *ring, serve_slot *host_ring,
                                      const serve_push *push, u32 ring_mask, serve_ring_ctl *ctl, serve_ring_dev *dev, const serve_worker_ids &ids, int count,
                                 unsigned long long linger_ticks, unsigned long long life_ticks, unsigned long long poll_period, hipStream_t st)
{
  if (count <= 0) return KVZ_HIP_OK;
  if (kvzhip::tuning("full_qsad", 1))
    hipLaunchKernelGGL((serve_worker_kernel<true>), dim3((unsigned)count), dim3(512), 0, st, planes, plane_bytes, n_slots, stride, w, h, ring, host_ring, push, ring_mask, ctl, dev, ids, linger_ticks, life_ticks, poll_period);
  else
    hipLaunchKernelGGL((serve_worker_kernel<false>), dim3((unsigned)count), dim3(512), 0, st, planes, plane_bytes, n_slots, stride, w, h, ring, host_ring, push, ring_mask, ctl, dev, ids, linger_ticks, life_ticks, poll_period);
  KVZ_CHECK_LAUNCH("search service workers");
  return KVZ_HIP_OK;
}

int kvzhip::serve_workers_launch(const u8 *planes, size_t plane_bytes, int n_slots, u32 stride, int w, int h, serve_slot *ring, u32 ring_mask,
                                 serve_ring_ctl *ctl, serve_ring_dev *dev, const serve_worker_ids &ids, int count,
                                 unsigned long long linger_ticks, unsigned long long life_ticks, unsigned long long poll_period, hipStream_t st)
{
  return serve_workers_launch_push(planes, plane_bytes, n_slots, stride, w, h, ring, ring, nullptr, ring_mask, ctl, dev, ids, count, linger_ticks, life_ticks, poll_period, st);
}

// serve.hip's launch of one batch; `units` is device-visible host memory
int kvzhip::serve_launch(bool constrained, const u8 *planes, size_t plane_bytes, int n_slots, u32 stride, int w, int h,
                         const serve_unit *units, int count, hipStream_t st)
{
  if (count <= 0) return KVZ_HIP_OK;
  const bool qsad = kvzhip::tuning("full_qsad", 1) != 0;
  if (constrained && qsad) hipLaunchKernelGGL((serve_kernel<true, true>), dim3((unsigned)count), dim3(512), 0, st, planes, plane_bytes, n_slots, stride, w, h, units, count);
  else if (constrained) hipLaunchKernelGGL((serve_kernel<true, false>), dim3((unsigned)count), dim3(512), 0, st, planes, plane_bytes, n_slots, stride, w, h, units, count);
  else if (qsad) hipLaunchKernelGGL((serve_kernel<false, true>), dim3((unsigned)count), dim3(512), 0, st, planes, plane_bytes, n_slots, stride, w, h, units, count);
  else hipLaunchKernelGGL((serve_kernel<false, false>), dim3((unsigned)count), dim3(512), 0, st, planes, plane_bytes, n_slots, stride, w, h, units, count);
  KVZ_CHECK_LAUNCH("search service kernel");
  return KVZ_HIP_OK;
}

template <bool CONSTR, bool MULTI>
static void launch_classes(int classes, const u8 *pic, u32 pic_stride, int pic_w, int pic_h, const refplane_t &r, const kvz_hip_me_pu *pus,
                           size_t count, const kvz_hip_me_params &prm, kvz_hip_me_result *results, hipStream_t st)
{
  // one launch per size class over the same descriptor list; each kernel takes its class and skips the rest
  if (classes == 7 || (classes & 4))
    hipLaunchKernelGGL((search_pu_big_kernel<512, CONSTR, MULTI>), dim3((unsigned)count), dim3(512), 0, st, pic, pic_stride, pic_w, pic_h, r, pus, prm, results);
  if (classes & 1)
    hipLaunchKernelGGL((search_pu_small_kernel<CONSTR, MULTI>), dim3((unsigned)((count + 3) / 4)), dim3(256), 0, st, pic, pic_stride, pic_w, pic_h, r, pus, count, prm, results);
  if (classes & 2)
    hipLaunchKernelGGL((search_pu_medium_wg_kernel<128, CONSTR, MULTI>), dim3((unsigned)count), dim3(128), 0, st, pic, pic_stride, pic_w, pic_h, r, pus, prm, results);
}

// n_planes == 0: pic / ref are the planes; > 0: device tables of that many plane pointers (kvz_hip_search_pu_multi_batch)
static int search_pu_launch(const kvz_hip_pixel *pic, uint32_t pic_stride, int pic_w, int pic_h,
                            const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h, int n_planes,
                            const kvz_hip_me_pu *pus, size_t count, const kvz_hip_me_params *params,
                            kvz_hip_me_result *results, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!pic || !ref || !pus || !params || !results || pic_w <= 0 || pic_h <= 0 || ref_w <= 0 || ref_h <= 0) {
    set_error_msg("kvz_hip_search_pu_batch: null buffer or empty plane");
    return KVZ_HIP_ERR_INVALID;
  }
  if (params->lambda_cost < 0 || params->lambda_cost > (1 << 20)) {
    set_error_msg("kvz_hip_search_pu_batch: lambda_cost must be within 0 .. 2^20 (costs are 32-bit like the reference's)");
    return KVZ_HIP_ERR_INVALID;
  }
  if (params->fme_level < 0 || params->fme_level > 4 || params->early_termination < 0 || params->early_termination > 2 ||
      params->algorithm < 0 || params->algorithm > 3 ||
      (params->algorithm == 3 && (params->search_range < 1 || params->search_range > 64))) {
    set_error_msg("kvz_hip_search_pu_batch: fme_level must be 0..4, early_termination 0..2, algorithm 0 (hexbs), 1 (dia), 2 (tz) or 3 (full, search_range 1..64)");
    return KVZ_HIP_ERR_INVALID;
  }
  kvz_hip_me_params prm_v = *params;
  if (prm_v.tile_w == 0 && prm_v.tile_h == 0) { prm_v.tile_x = 0; prm_v.tile_y = 0; prm_v.tile_w = pic_w; prm_v.tile_h = pic_h; }
  if (prm_v.mv_constraint < 0 || prm_v.mv_constraint > 4 || prm_v.tile_x < 0 || prm_v.tile_y < 0 || prm_v.tile_w <= 0 || prm_v.tile_h <= 0 ||
      prm_v.tile_x + prm_v.tile_w > pic_w || prm_v.tile_y + prm_v.tile_h > pic_h || (prm_v.wpp_owf && ((prm_v.tile_x & 63) || (prm_v.tile_y & 63)))) {
    set_error_msg("kvz_hip_search_pu_batch: mv_constraint must be 0..4 and the tile must lie inside the picture (origin a multiple of 64 when wpp_owf is set: its rule counts LCUs from there)");
    return KVZ_HIP_ERR_INVALID;
  }
  params = &prm_v;
  if (count == 0) return KVZ_HIP_OK;
  if (count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  hipStream_t st = ctx_stream(s);
  const refplane_t r = { ref, ref_stride, ref_w, ref_h };
  // one launch per size class over the same descriptor list; each kernel takes its class and skips the rest.  The big
  // kernel also flags malformed descriptors, so it only goes when the caller vouches for the classes it names.
  if (params->mv_rdo) {
    if (n_planes) { set_error_msg("kvz_hip_search_pu_multi_batch: mv_rdo is a one-picture path"); return KVZ_HIP_ERR_INVALID; }
    if (!params->cabac || params->n_cabac < 1 || params->refs_before < 1 || params->refs_before > 16 || params->ref_idx < 0 || params->ref_idx >= 16) {
      set_error_msg("kvz_hip_search_pu_batch: mv_rdo needs the cabac snapshots (device array of n_cabac >= 1), refs_before 1..16 and ref_idx 0..15");
      return KVZ_HIP_ERR_INVALID;
    }
    hipLaunchKernelGGL(search_pu_rdo_kernel, dim3((unsigned)count), dim3(256), 0, st, pic, pic_stride, pic_w, pic_h, r, pus, *params, results);
    KVZ_CHECK_LAUNCH("search_pu_rdo_kernel");
    return KVZ_HIP_OK;
  }
  // with a hint, PUs of a class it does not name are searched by no kernel: they read cost 0xFFFFFFFF, reserved -1
  // (pu_orphan: written by the kernels themselves -- a separate fill would be one more command per dependency front)
  const int classes = (params->size_classes & 7) ? (params->size_classes & 7) : 7;
  prm_v.size_classes = classes;
  prm_v.n_cabac = n_planes;                             // the multi-picture kernels find the table length here (mv_rdo is a one-picture path)
  // thread counts are measured choices: > 32x32: 512 threads per PU 10.5 M/s (256: 9.5, 1024: 6.5); <= 32x32: 128 per PU 48.9 M/s
  // (256: 46.6, one wave: 42.6); <= 16x16: one wave per PU, four per workgroup
  const bool constrained = params->wpp_owf != 0 || params->mv_constraint != 0;
  if (n_planes) {
    if (constrained) launch_classes<true, true>(classes, pic, pic_stride, pic_w, pic_h, r, pus, count, *params, results, st);
    else launch_classes<false, true>(classes, pic, pic_stride, pic_w, pic_h, r, pus, count, *params, results, st);
  } else {
    if (constrained) launch_classes<true, false>(classes, pic, pic_stride, pic_w, pic_h, r, pus, count, *params, results, st);
    else launch_classes<false, false>(classes, pic, pic_stride, pic_w, pic_h, r, pus, count, *params, results, st);
  }
  KVZ_CHECK_LAUNCH("search_pu kernels");
  return KVZ_HIP_OK;
}


extern "C" int kvz_hip_search_pu_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, int pic_w, int pic_h,
                                       const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                       const kvz_hip_me_pu *pus, size_t count, const kvz_hip_me_params *params,
                                       kvz_hip_me_result *results, kvz_hip_stream s)
{
  return search_pu_launch(pic, pic_stride, pic_w, pic_h, ref, ref_stride, ref_w, ref_h, 0, pus, count, params, results, s);
}

extern "C" int kvz_hip_search_pu_multi_batch(const kvz_hip_pixel *const *pics, uint32_t pic_stride, int pic_w, int pic_h,
                                             const kvz_hip_pixel *const *refs, uint32_t ref_stride, int ref_w, int ref_h, int n_planes,
                                             const kvz_hip_me_pu *pus, size_t count, const kvz_hip_me_params *params,
                                             kvz_hip_me_result *results, kvz_hip_stream s)
{
  if (n_planes < 1 || n_planes > 8192) { set_error_msg("kvz_hip_search_pu_multi_batch: 1 .. 8192 plane pairs"); return KVZ_HIP_ERR_INVALID; }
  return search_pu_launch(reinterpret_cast<const kvz_hip_pixel *>(pics), pic_stride, pic_w, pic_h, reinterpret_cast<const kvz_hip_pixel *>(refs),
                          ref_stride, ref_w, ref_h, n_planes, pus, count, params, results, s);
}
