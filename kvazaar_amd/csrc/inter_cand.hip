// inter_cand.hip -- AMVP / merge candidate derivation next to the motion search.
//
// Reference: src/inter.c:546-1446 (kvz_inter_get_mv_cand :1209-1240, kvz_inter_get_merge_cand :1314-1446 and what they
// are made of: the spatial neighbours A0 A1 B0 B1 B2 with their coding-order tests :566-875, the temporal neighbour on
// the collocated picture's 16x16 grid :713-780, POC-distance scaling :955-1061) and the start vector that
// search_pu_inter_ref takes from the searched picture's CU array (search_inter.c:1190-1206).
// SURVEY.md section 8(f) row 1, the "driver" half: this is what a host has to derive between two dependency fronts
// of a frame -- it reads nothing but the neighbours' decided motion, so with the CU arrays resident in HBM
// (kvz_hip_cu_info per 4x4 SCU, the layout the deblocking entry already uses) the descriptors of a front's PUs are
// completed on the device and handed to kvz_hip_search_pu_batch on the same stream.
//
// One LANE per PU: the derivation is a short chain of dependent table lookups (five spatial records, one temporal),
// a frame has tens of thousands of PUs, and a front that has only a dozen is bound by the launch, not by this kernel.
#include "kvz_hip_internal.h"

using namespace kvzhip;

static_assert(sizeof(kvz_hip_inter_params) == 252 && sizeof(kvz_hip_merge_cand) == 12 && sizeof(kvz_hip_me_pu) == 64 && sizeof(kvz_hip_inter_picture) == 280,
              "layouts of include/kvz_hip.h");

namespace {

// the motion of one neighbour; a list the CU does not use reads vector 0, reference 255 (inter_clear_cu_unused, inter.c:546-555)
struct cand_t { bool ok; int dir; int mv[2][2]; int ref[2]; };

__device__ __forceinline__ cand_t no_cand()
{
  cand_t c;
  c.ok = false; c.dir = 0;
  c.mv[0][0] = c.mv[0][1] = c.mv[1][0] = c.mv[1][1] = 0;
  c.ref[0] = c.ref[1] = 255;
  return c;
}

// list l of a candidate without indexing its arrays by a run-time value (which would move every candidate to scratch memory)
__device__ __forceinline__ int cand_mvx(const cand_t &c, int l) { return l ? c.mv[1][0] : c.mv[0][0]; }
__device__ __forceinline__ int cand_mvy(const cand_t &c, int l) { return l ? c.mv[1][1] : c.mv[0][1]; }
__device__ __forceinline__ int cand_ref(const cand_t &c, int l) { return l ? c.ref[1] : c.ref[0]; }

// a fetched record as a candidate; `want` = the position passed its availability tests
__device__ __forceinline__ cand_t cand_of(const kvz_hip_cu_info &c, bool want)
{
  cand_t v = no_cand();
  if (!want || c.type != 2) return v;                   // inter.c:822-870: only inter CUs are candidates
  v.ok = true; v.dir = c.mv_dir;
#pragma unroll
  for (int l = 0; l < 2; ++l)
    if ((c.mv_dir >> l) & 1) { v.mv[l][0] = c.mv[l][0]; v.mv[l][1] = c.mv[l][1]; v.ref[l] = c.mv_ref[l]; }
  return v;
}

// one record of a CU array as five dwords (records are 20 bytes, 4-byte aligned)
__device__ __forceinline__ kvz_hip_cu_info fetch_cu(const kvz_hip_cu_info *__restrict__ map, int stride, int x, int y, bool want)
{
  const size_t at = want ? (size_t)(y >> 2) * stride + (x >> 2) : 0;    // a position that failed its tests reads record 0 and is dropped
  const u32 *q = reinterpret_cast<const u32 *>(map + at);
  u32 w[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) w[k] = q[k];
  kvz_hip_cu_info r;
  __builtin_memcpy(&r, w, sizeof(r));
  return r;
}

// position of a 4x4 unit in the coding order of its LCU (bits of its coordinates interleaved)
__device__ __forceinline__ unsigned unit_order(unsigned ux, unsigned uy)
{
  unsigned z = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) z |= ((ux >> b) & 1u) << (2 * b) | ((uy >> b) & 1u) << (2 * b + 1);
  return z;
}

// is_a0_cand_coded / is_b0_cand_coded (inter.c:566-705): the neighbour's unit precedes the aligned square at the PU's
// lower-left / upper-right corner; other LCUs are coded when they precede this one in raster order
__device__ __forceinline__ bool corner_unit_coded(int nx, int ny, int sx, int sy)
{
  if ((nx >> 6) != (sx >> 6) || (ny >> 6) != (sy >> 6)) return (ny >> 6) < (sy >> 6) || ((ny >> 6) == (sy >> 6) && (nx >> 6) < (sx >> 6));
  return unit_order((unsigned)(nx & 63) >> 2, (unsigned)(ny & 63) >> 2) < unit_order((unsigned)(sx & 63) >> 2, (unsigned)(sy & 63) >> 2);
}

struct cand_set { cand_t a[2], b[3]; };

// get_spatial_merge_candidates (inter.c:799-875), get_temporal_merge_candidates (inter.c:713-780; every caller passes
// list 1, index 0: H below-right of the PU unless that opens a new LCU row, else the centre C3, both on the 16x16 grid of
// the collocated picture) and the CU of the searched picture under the PU's centre (search_inter.c:1190-1206).
// All eight records are requested before any is looked at: the availability tests need only the geometry, so the
// fetches are independent and one memory round trip serves the PU (as a chain of conditional lookups the kernel took
// 12 us for 480 PUs).
__device__ __forceinline__ void gather_cands(const kvz_hip_cu_info *__restrict__ cus, const kvz_hip_cu_info *__restrict__ col,
                                             const kvz_hip_cu_info *__restrict__ ref_cus, const kvz_hip_inter_params &p,
                                             int x, int y, int w, int h, cand_set &s, cand_t &tmp, cand_t &centre)
{
  const int lw = w & -w, lh = h & -h, side = lw < lh ? lw : lh;
  const int xl = x & 63, yl = y & 63;
  const bool left = x != 0, top = y != 0;
  const bool want_a0 = left && yl + h < 64 && y + h < p.pic_height && corner_unit_coded(x - 1, y + h, x, y + h - side);
  const bool want_b0 = top && x + w < p.pic_width && (xl + w < 64 || yl == 0) && corner_unit_coded(x + w, y - 1, x + w - side, y);
  const bool have_col = p.num_refs && p.ref_LX_size[0] != 0 && col != nullptr;
  const int bx = x + w, by = y + h, cx = x + w / 2, cy = y + h / 2;
  const bool want_h = have_col && bx < p.in_width && by < p.in_height && (by & 63) != 0;
  const bool want_c3 = have_col && cx < p.in_width && cy < p.in_height;
  const bool want_ctr = ref_cus != nullptr;
  const kvz_hip_cu_info r_a0 = fetch_cu(cus, p.cus_stride, x - 1, y + h, want_a0);
  const kvz_hip_cu_info r_a1 = fetch_cu(cus, p.cus_stride, x - 1, y + h - 1, left);
  const kvz_hip_cu_info r_b0 = fetch_cu(cus, p.cus_stride, x + w, y - 1, want_b0);
  const kvz_hip_cu_info r_b1 = fetch_cu(cus, p.cus_stride, x + w - 1, y - 1, top);
  const kvz_hip_cu_info r_b2 = fetch_cu(cus, p.cus_stride, x - 1, y - 1, left && top);
  const kvz_hip_cu_info r_h = fetch_cu(have_col ? col : cus, p.col_stride, bx & ~15, by & ~15, want_h);
  const kvz_hip_cu_info r_c3 = fetch_cu(have_col ? col : cus, p.col_stride, cx & ~15, cy & ~15, want_c3);
  const kvz_hip_cu_info r_ctr = fetch_cu(want_ctr ? ref_cus : cus, p.col_stride, p.tile_x + x + (w >> 1), p.tile_y + y + (h >> 1), want_ctr);
  s.a[0] = cand_of(r_a0, want_a0); s.a[1] = cand_of(r_a1, left);
  s.b[0] = cand_of(r_b0, want_b0); s.b[1] = cand_of(r_b1, top); s.b[2] = cand_of(r_b2, left && top);
  const cand_t hh = cand_of(r_h, want_h);
  tmp = hh.ok ? hh : cand_of(r_c3, want_c3);
  centre = cand_of(r_ctr, want_ctr);
}

// apply_mv_scaling_pocs + get_scaled_mv (inter.c:955-980); C integer division (towards zero) and arithmetic shifts
__device__ __forceinline__ void scale_mv(int cur_poc, int cur_ref_poc, int nb_poc, int nb_ref_poc, int mv[2])
{
  int dc = cur_poc - cur_ref_poc, dn = nb_poc - nb_ref_poc;
  if (dc == dn || dn == 0) return;
  dc = min(max(dc, -128), 127);
  dn = min(max(dn, -128), 127);
  int scale = (dc * ((0x4000 + (abs(dn) >> 1)) / dn) + 32) >> 6;
  scale = min(max(scale, -4096), 4095);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int prod = scale * (int)(short)mv[k];
    mv[k] = min(max((prod + 127 + (prod < 0)) >> 8, -32768), 32767);
  }
}

// add_temporal_candidate (inter.c:1011-1061)
__device__ __forceinline__ bool temporal_mv(const kvz_hip_inter_params &p, const cand_t &c, int cur_pic, int reflist, int out[2])
{
  if (!c.ok || p.ref_LX_size[0] == 0) return false;
  const int col_pic = p.ref_LX[0][0] & 15;
  int l = reflist;
  for (int i = 0; i < p.num_refs; ++i) if (p.ref_pocs[i] > p.poc) { l = 1; break; }
  if (!(c.dir & (l + 1))) l = 1 - l;
  out[0] = cand_mvx(c, l); out[1] = cand_mvy(c, l);
  scale_mv(p.poc, p.ref_pocs[cur_pic & 15], p.ref_pocs[col_pic], p.col_ref_pocs[p.col_ref_LX[l][cand_ref(c, l) & 15] & 15], out);
  return true;
}

// add_mvp_candidate (inter.c:1063-1098)
__device__ __forceinline__ bool mvp_from(const kvz_hip_inter_params &p, const cand_t &c, int reflist, int cur_pic, bool scaling, int out[2])
{
  if (!c.ok) return false;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int l = i == 0 ? reflist : 1 - reflist;
    if (!(c.dir & (1 << l))) continue;
    const int nb_pic = p.ref_LX[l][cand_ref(c, l) & 15];
    if (scaling) {
      out[0] = cand_mvx(c, l); out[1] = cand_mvy(c, l);
      scale_mv(p.poc, p.ref_pocs[cur_pic & 15], p.poc, p.ref_pocs[nb_pic & 15], out);
      return true;
    }
    if (nb_pic == cur_pic) { out[0] = cand_mvx(c, l); out[1] = cand_mvy(c, l); return true; }
  }
  return false;
}

// get_mv_cand_from_candidates (inter.c:1102-1195).  The list position n is data dependent: entries are written through
// selects over the three slots so that everything stays in registers.
__device__ __forceinline__ void amvp(const kvz_hip_inter_params &p, const cand_set &s, const cand_t &tmp, int reflist, int lx_idx, int16_t mv_cand[2][2])
{
  const int cur_pic = p.ref_LX[reflist][lx_idx & 15];
  int mvx[3] = { 0, 0, 0 }, mvy[3] = { 0, 0, 0 }, n = 0, v[2];
  auto put = [&](int at) {
#pragma unroll
    for (int k = 0; k < 3; ++k) if (k == at) { mvx[k] = v[0]; mvy[k] = v[1]; }
  };
  // left: the first of A0, A1 that points at the same picture, else the first that has a vector at all, scaled
  bool got = false;
#pragma unroll
  for (int sc = 0; sc < 2; ++sc)
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (!got && mvp_from(p, s.a[i], reflist, cur_pic, sc != 0, v)) { put(n); ++n; got = true; }
  // above: the first of B0, B1, B2 pointing at the same picture ...
  int above = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (!above && mvp_from(p, s.b[i], reflist, cur_pic, false, v)) { put(n); above = 1; }
  n += above;
  // ... and a scaled one only when there is no left neighbour at all and the list is still short
  if (s.a[0].ok || s.a[1].ok) above = 1; else if (n != 2) above = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (!above && mvp_from(p, s.b[i], reflist, cur_pic, true, v)) { put(n); ++n; above = 1; }
  if (n == 2 && mvx[0] == mvx[1] && mvy[0] == mvy[1]) n = 1;
  if (p.tmvp_enable && p.poc > 1 && p.num_refs && n < 2 && tmp.ok && temporal_mv(p, tmp, cur_pic, reflist, v)) { put(n); ++n; }
  mv_cand[0][0] = (int16_t)(n > 0 ? mvx[0] : 0); mv_cand[0][1] = (int16_t)(n > 0 ? mvy[0] : 0);
  mv_cand[1][0] = (int16_t)(n > 1 ? mvx[1] : 0); mv_cand[1][1] = (int16_t)(n > 1 ? mvy[1] : 0);
}

// is_duplicate_candidate (inter.c:1262-1278)
__device__ __forceinline__ bool same_motion(const cand_t &a, const cand_t &b)
{
  if (!b.ok || a.dir != b.dir) return false;
#pragma unroll
  for (int l = 0; l < 2; ++l)
    if ((a.dir >> l) & 1)
      if (a.mv[l][0] != b.mv[l][0] || a.mv[l][1] != b.mv[l][1] || a.ref[l] != b.ref[l]) return false;
  return true;
}

__device__ __forceinline__ int merge_push(const cand_t &c, const cand_t *d1, const cand_t *d2, kvz_hip_merge_cand &o)
{
  if (!c.ok || (d1 && same_motion(c, *d1)) || (d2 && same_motion(c, *d2))) return 0;
#pragma unroll
  for (int l = 0; l < 2; ++l) { o.mv[l][0] = (int16_t)c.mv[l][0]; o.mv[l][1] = (int16_t)c.mv[l][1]; o.ref[l] = (uint8_t)c.ref[l]; }
  o.dir = (uint8_t)c.dir;
  return 1;
}

__constant__ u8 c_pair_first[12] = { 0, 1, 0, 2, 1, 2, 0, 3, 1, 3, 2, 3 };     // priorityList0 / 1, inter.c:1379-1380
__constant__ u8 c_pair_second[12] = { 1, 0, 2, 0, 2, 1, 3, 0, 3, 1, 3, 2 };

// kvz_inter_get_merge_cand (inter.c:1314-1446); entries the reference leaves unwritten stay zero
__device__ __forceinline__ int merge_list(const kvz_hip_inter_params &p, cand_set s, const cand_t &tmp, bool use_a1, bool use_b1, kvz_hip_merge_cand out[5])
{
  int n = 0;
  for (int i = 0; i < 5; ++i) { out[i].dir = 0; out[i].ref[0] = out[i].ref[1] = 0; out[i].pad = 0; out[i].mv[0][0] = out[i].mv[0][1] = out[i].mv[1][0] = out[i].mv[1][1] = 0; }
  if (!use_a1) s.a[1].ok = false;
  if (!use_b1) s.b[1].ok = false;
  n += merge_push(s.a[1], nullptr, nullptr, out[n]);
  n += merge_push(s.b[1], &s.a[1], nullptr, out[n]);
  n += merge_push(s.b[0], &s.b[1], nullptr, out[n]);
  n += merge_push(s.a[0], &s.a[1], nullptr, out[n]);
  if (n < 4) n += merge_push(s.b[2], &s.a[1], &s.b[1], out[n]);
  if (p.tmvp_enable && n < 5 && p.num_refs) {
    out[n].dir = 0;
    for (int l = 0; l <= (p.slice_is_b ? 1 : 0); ++l) {
      int mv[2];
      if (temporal_mv(p, tmp, p.ref_LX[l][0], l, mv)) {
        out[n].mv[l][0] = (int16_t)mv[0]; out[n].mv[l][1] = (int16_t)mv[1];
        out[n].ref[l] = 0;
        out[n].dir |= (uint8_t)(1 << l);
      }
    }
    if (out[n].dir) ++n;
  }
  if (n < 5 && p.slice_is_b) {
    const int cutoff = n;
    for (int k = 0; k < cutoff * (cutoff - 1) && n != 5; ++k) {
      const int i = c_pair_first[k], j = c_pair_second[k];
      if (i >= n || j >= n) break;
      if (!(out[i].dir & 1) || !(out[j].dir & 2)) continue;
      out[n].dir = 3;
      out[n].mv[0][0] = out[i].mv[0][0]; out[n].mv[0][1] = out[i].mv[0][1];
      out[n].mv[1][0] = out[j].mv[1][0]; out[n].mv[1][1] = out[j].mv[1][1];
      out[n].ref[0] = out[i].ref[0]; out[n].ref[1] = out[j].ref[1];
      const bool same = p.ref_LX[0][out[i].ref[0] & 15] == p.ref_LX[1][out[j].ref[1] & 15] &&
                        out[i].mv[0][0] == out[j].mv[1][0] && out[i].mv[0][1] == out[j].mv[1][1];
      if (!same) ++n;
    }
  }
  int num_ref = p.num_refs;
  if (n < 5 && p.slice_is_b) {
    int before = 0, after = 0;
    for (int j = 0; j < p.num_refs; ++j) { if (p.ref_pocs[j] < p.poc) ++before; else ++after; }
    num_ref = before < after ? before : after;
  }
  for (int zero_idx = 0; n != 5; ++zero_idx, ++n) {
    out[n].mv[0][0] = out[n].mv[0][1] = 0;
    out[n].ref[0] = (uint8_t)(zero_idx >= num_ref - 1 ? 0 : zero_idx);
    out[n].ref[1] = out[n].ref[0];
    out[n].dir = 1;
    if (p.slice_is_b) { out[n].mv[1][0] = out[n].mv[1][1] = 0; out[n].dir = 3; }
  }
  return n;
}

// everything of one PU, given its picture's arrays and parameters (p: LDS in the one-picture kernel, global memory in the
// several-pictures one)
__device__ __forceinline__ void derive_pu(const kvz_hip_cu_info *__restrict__ cus, const kvz_hip_cu_info *__restrict__ col_cus,
                                          const kvz_hip_cu_info *__restrict__ ref_cus, const kvz_hip_inter_params &p, bool params_ok,
                                          int reflist, int lx_idx, kvz_hip_me_pu *__restrict__ pus, size_t i, kvz_hip_merge_cand *mc,
                                          kvz_hip_merge_cand *__restrict__ merge_out)
{
  kvz_hip_me_pu u = pus[i];
  // descriptors carry PICTURE coordinates, like kvz_hip_search_pu_batch's; the reference's functions work in the tile's
  const int x = u.x - p.tile_x, y = u.y - p.tile_y, w = u.width, h = u.height;
  // a descriptor outside the picture or off the 4-pixel grid: num_merge_cand -1, nothing read
  const bool ok = params_ok && x >= 0 && y >= 0 && w >= 4 && h >= 4 && w <= 64 && h <= 64 && ((x | y | w | h) & 3) == 0 &&
                  x + w <= p.pic_width && y + h <= p.pic_height;
  u.extra_mv[0] = u.extra_mv[1] = 0;
  u.mv_cand[0][0] = u.mv_cand[0][1] = u.mv_cand[1][0] = u.mv_cand[1][1] = 0;
#pragma unroll
  for (int k = 0; k < 5; ++k) { u.merge[k].mv[0] = u.merge[k].mv[1] = 0; u.merge[k].usable = 0; u.merge[k].same_ref = 0; }
  if (!ok) {
    u.num_merge_cand = -1;
    pus[i] = u;
    if (merge_out) for (int k = 0; k < 5; ++k) { kvz_hip_merge_cand z = { 0, { 0, 0 }, 0, { { 0, 0 }, { 0, 0 } } }; merge_out[5 * i + k] = z; }
    return;
  }
  cand_set s;
  cand_t tmp, centre;
  gather_cands(cus, col_cus, ref_cus, p, x, y, w, h, s, tmp, centre);
  // search_pu_inter (search_inter.c:1470-1500): the merge list, as calc_mvd_cost / the merge match read it
  const int n = merge_list(p, s, tmp, !(u.pad & 1), !(u.pad & 2), mc);
  u.num_merge_cand = (int16_t)n;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    if (k >= n) continue;
    u.merge[k].usable = mc[k].dir != 3;
    if (mc[k].dir != 3) {
      const int l = mc[k].dir - 1;
      u.merge[k].mv[0] = mc[k].mv[l][0]; u.merge[k].mv[1] = mc[k].mv[l][1];
      u.merge[k].same_ref = p.ref_LX[l][mc[k].ref[l] & 15] == p.ref_idx;
    }
  }
  if (merge_out) for (int k = 0; k < 5; ++k) merge_out[5 * i + k] = mc[k];
  // search_pu_inter_ref (search_inter.c:1168-1206): the AMVP pair of the picture searched, the collocated CU's vector
  if (reflist >= 0) amvp(p, s, tmp, reflist, lx_idx, u.mv_cand);
  if (centre.ok) { const int l = (centre.dir & 1) ? 0 : 1; u.extra_mv[0] = (int16_t)cand_mvx(centre, l); u.extra_mv[1] = (int16_t)cand_mvy(centre, l); }
  pus[i] = u;
}

__global__ __launch_bounds__(64) void inter_candidates_kernel(const kvz_hip_cu_info *__restrict__ cus, const kvz_hip_cu_info *__restrict__ col_cus,
                                                              const kvz_hip_cu_info *__restrict__ ref_cus, kvz_hip_inter_params p_arg, int reflist, int lx_idx,
                                                              kvz_hip_me_pu *__restrict__ pus, size_t count, kvz_hip_merge_cand *__restrict__ merge_out)
{
  // The POC and list tables are indexed with per-lane values and a merge list is filled at a per-lane position: both
  // live in LDS (as kernel arguments / registers the compiler had moved them to 288 bytes of scratch memory per lane,
  // several dependent memory round trips on a kernel that is nothing but latency).
  __shared__ kvz_hip_inter_params p;
  __shared__ kvz_hip_merge_cand s_mc[64][5];
  if (threadIdx.x == 0) {
    const u32 *src = reinterpret_cast<const u32 *>(&p_arg);
    u32 *dst = reinterpret_cast<u32 *>(&p);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(p) / 4); ++k) dst[k] = src[k];
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= count) return;
  derive_pu(cus, col_cus, ref_cus, p, true, reflist, lx_idx, pus, i, s_mc[threadIdx.x], merge_out);
}

// which list holds picture ref_idx, and where (search_pu_inter_ref, search_inter.c:1143-1166); in neither: no AMVP pair
__device__ __forceinline__ void find_list(const kvz_hip_inter_params &p, int &reflist, int &lx)
{
  reflist = -1;
  const int lx_max = p.ref_LX_size[0] > p.ref_LX_size[1] ? p.ref_LX_size[0] : p.ref_LX_size[1];
  for (lx = 0; lx < lx_max; ++lx) {
    if (lx < p.ref_LX_size[0] && p.ref_LX[0][lx] == p.ref_idx) { reflist = 0; return; }
    if (lx < p.ref_LX_size[1] && p.ref_LX[1][lx] == p.ref_idx) { reflist = 1; return; }
  }
}

// kvz_hip_inter_candidates_multi_batch: every PU names its picture (pad >> 2); arrays and parameters come from that picture's
// record in device memory.  The records cannot be checked on the host, so what the one-picture entry refuses is refused
// per PU here (num_merge_cand -1).
__global__ __launch_bounds__(64) void inter_candidates_multi_kernel(const kvz_hip_inter_picture *__restrict__ pictures, int n_pictures,
                                                                    kvz_hip_me_pu *__restrict__ pus, size_t count,
                                                                    kvz_hip_merge_cand *__restrict__ merge_out)
{
  __shared__ kvz_hip_merge_cand s_mc[64][5];
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= count) return;
  int k = pus[i].pad >> 2;
  const bool known = k >= 0 && k < n_pictures;
  if (!known) k = 0;
  const kvz_hip_inter_picture &pc = pictures[k];
  const kvz_hip_inter_params &p = pc.params;
  bool ok = known && pc.cus != nullptr && p.num_refs >= 0 && p.num_refs <= 16 && p.ref_LX_size[0] <= 16 && p.ref_LX_size[1] <= 16 &&
            p.pic_width > 0 && p.pic_height > 0 && p.tile_x >= 0 && p.tile_y >= 0 && p.in_width >= p.tile_x + p.pic_width &&
            p.in_height >= p.tile_y + p.pic_height && p.cus_stride * 4 >= p.pic_width && p.col_stride * 4 >= p.in_width &&
            p.ref_idx >= 0 && p.ref_idx < 16 && !(p.num_refs > 0 && p.tmvp_enable && pc.col_cus == nullptr);
  // list entries 0..15, like the one-picture entry's host check (a bad entry flags the picture's PUs; it is never masked into range)
  if (ok) {
    for (int l = 0; l < 2; ++l)
      for (int e = 0; e < 16; ++e) if (e < p.ref_LX_size[l] && p.ref_LX[l][e] > 15) ok = false;
  }
  int reflist = -1, lx = 0;
  if (ok) find_list(p, reflist, lx);
  derive_pu(pc.cus, pc.col_cus, pc.ref_cus, p, ok, reflist, lx, pus, i, s_mc[threadIdx.x], merge_out);
}

}  // namespace

extern "C" int kvz_hip_inter_candidates_batch(const kvz_hip_cu_info *cus, const kvz_hip_cu_info *col_cus, const kvz_hip_cu_info *ref_cus,
                                              const kvz_hip_inter_params *params, kvz_hip_me_pu *pus, size_t count,
                                              kvz_hip_merge_cand *merge_out, kvz_hip_stream stream)
{
  KVZ_CHECK_CTX();
  if (!params) { set_error_msg("kvz_hip_inter_candidates_batch: null params"); return KVZ_HIP_ERR_INVALID; }
  const kvz_hip_inter_params &p = *params;
  bool ok = p.num_refs >= 0 && p.num_refs <= 16 && p.ref_LX_size[0] <= 16 && p.ref_LX_size[1] <= 16 && p.pic_width > 0 && p.pic_height > 0 &&
            p.tile_x >= 0 && p.tile_y >= 0 && p.in_width >= p.tile_x + p.pic_width && p.in_height >= p.tile_y + p.pic_height &&
            p.cus_stride * 4 >= p.pic_width && p.col_stride * 4 >= p.in_width && p.ref_idx >= 0 && p.ref_idx < 16;
  // entries past a list's length are whatever the encoder left there (the reference never reads them; the kernel masks its indices)
  for (int l = 0; l < 2 && ok; ++l)
    for (int i = 0; i < p.ref_LX_size[l]; ++i) if (p.ref_LX[l][i] > 15) ok = false;
  if (!ok) {
    set_error_msg("kvz_hip_inter_candidates_batch: at most 16 references, list entries 0..15, the tile inside the input picture, strides covering the pictures");
    return KVZ_HIP_ERR_INVALID;
  }
  if (count == 0) return KVZ_HIP_OK;
  if (!cus || !pus || (p.num_refs > 0 && p.tmvp_enable && !col_cus)) {
    set_error_msg("kvz_hip_inter_candidates_batch: null buffer (col_cus is needed whenever tmvp_enable is set and there are references)");
    return KVZ_HIP_ERR_INVALID;
  }
  if (count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  // which list holds picture ref_idx, and where (search_pu_inter_ref, search_inter.c:1143-1166); in neither: no AMVP pair
  int reflist = -1, lx = 0;
  const int lx_max = p.ref_LX_size[0] > p.ref_LX_size[1] ? p.ref_LX_size[0] : p.ref_LX_size[1];
  for (lx = 0; lx < lx_max; ++lx) {
    if (lx < p.ref_LX_size[0] && p.ref_LX[0][lx] == p.ref_idx) { reflist = 0; break; }
    if (lx < p.ref_LX_size[1] && p.ref_LX[1][lx] == p.ref_idx) { reflist = 1; break; }
  }
  hipStream_t st = ctx_stream(stream);
  hipLaunchKernelGGL(inter_candidates_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, cus, col_cus, ref_cus, p, reflist, lx, pus, count, merge_out);
  KVZ_CHECK_LAUNCH("inter_candidates_kernel");
  return KVZ_HIP_OK;
}

extern "C" int kvz_hip_inter_candidates_multi_batch(const kvz_hip_inter_picture *pictures, int n_pictures, kvz_hip_me_pu *pus, size_t count,
                                                    kvz_hip_merge_cand *merge_out, kvz_hip_stream stream)
{
  KVZ_CHECK_CTX();
  if (n_pictures < 1 || n_pictures > 8192) { set_error_msg("kvz_hip_inter_candidates_multi_batch: 1 .. 8192 pictures"); return KVZ_HIP_ERR_INVALID; }
  if (count == 0) return KVZ_HIP_OK;
  if (!pictures || !pus) { set_error_msg("kvz_hip_inter_candidates_multi_batch: null buffer"); return KVZ_HIP_ERR_INVALID; }
  if (count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  hipStream_t st = ctx_stream(stream);
  hipLaunchKernelGGL(inter_candidates_multi_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, pictures, n_pictures, pus, count, merge_out);
  KVZ_CHECK_LAUNCH("inter_candidates_multi_kernel");
  return KVZ_HIP_OK;
}
