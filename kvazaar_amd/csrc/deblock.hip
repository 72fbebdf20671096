// deblock.hip -- the deblocking filter over one reconstructed frame.
//
// Reference: src/filter.c:83-779 (kvz_filter_deblock_lcu and everything below it).  SURVEY.md section 8(f) row 4.
//
// The reference walks the LCUs in coding order: vertical edges of the LCU, the deferred rightmost 4 pixels of the
// horizontal edges of the LCU to the left, then its own horizontal edges (filter.c:770-779) -- an order built so that
// every horizontal edge sees vertically filtered pixels, i.e. the HEVC process "all vertical edges of the picture, then
// all horizontal edges".  That is what runs here: TWO launches over the frame, one per direction, every 4-pixel edge
// segment its own thread.  Edges are 8 apart and the filter reaches 3 pixels (4 read), so the segments of one pass
// touch disjoint pixels and need no ordering among themselves.
//
// A thread = one luma segment (unit = 8x8 block of the edge grid, 2 segments per unit); the two threads of a unit also
// take the unit's chroma segment, one plane each.  Everything the reference derives from cu_array on the way --
// TU / PU boundary tests, boundary strength incl. the B-slice vector rules, per-CU QP prediction, tc / beta -- is
// derived on the device from a flat copy of the cu_info_t fields (kvz_hip_cu_info, one per 4x4 SCU).
// Lanes walk along x: a wave's loads of one picture row are contiguous (8 B per lane, vertical edges; 4 B, horizontal).
#include "kvz_hip_internal.h"

using namespace kvzhip;

static_assert(sizeof(kvz_hip_cu_info) == 20 && sizeof(kvz_hip_deblock_params) == 64, "layouts of include/kvz_hip.h");

namespace {

__constant__ u8 c_tc_table[54] = {       // kvz_g_tc_table_8x8, filter.c:34-42
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 5, 5,
  6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__constant__ u8 c_beta_table[52] = {     // kvz_g_beta_table_8x8, filter.c:44-52
  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32,
  34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
__constant__ u8 c_chroma_scale[58] = {   // kvz_g_chroma_scale, transform.c:44-50
  0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
  33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
// kvz_part_mode_num_parts / kvz_part_mode_offsets (cu.c:33-60), [part][pu] -> (x, y) in quarters of the CU, 2 bits each
__constant__ u8 c_num_parts[8] = { 1, 2, 2, 4, 2, 2, 2, 2 };
__constant__ u8 c_part_off_x[8][4] = { { 0 }, { 0, 0 }, { 0, 2 }, { 0, 2, 0, 2 }, { 0, 0 }, { 0, 0 }, { 0, 1 }, { 0, 3 } };
__constant__ u8 c_part_off_y[8][4] = { { 0 }, { 0, 2 }, { 0, 0 }, { 0, 0, 2, 2 }, { 0, 1 }, { 0, 3 }, { 0, 0 }, { 0, 0 } };

struct cu_t { int type, depth, part, trd, cbf_y, mv_dir, qp; int mv[2][2]; int ref[2]; };

struct frame_t {
  const u32 *cus;      // kvz_hip_cu_info records as dwords (5 each)
  int w4, width, height;
};

__device__ __forceinline__ u32 cu_head(const frame_t &f, int x, int y) { return f.cus[((size_t)(y >> 2) * f.w4 + (x >> 2)) * 5]; }
__device__ __forceinline__ int cu_qp(const frame_t &f, int x, int y) { return (int)((f.cus[((size_t)(y >> 2) * f.w4 + (x >> 2)) * 5 + 1] >> 16) & 255u); }
__device__ __forceinline__ cu_t cu_load(const frame_t &f, int x, int y)
{
  const u32 *p = f.cus + ((size_t)(y >> 2) * f.w4 + (x >> 2)) * 5;
  const u32 a = p[0], b = p[1], c = p[2], d = p[3], e = p[4];
  cu_t r;
  r.type = a & 255; r.depth = (a >> 8) & 255; r.part = (a >> 16) & 7; r.trd = a >> 24;
  r.cbf_y = b & 255; r.mv_dir = (b >> 8) & 255; r.qp = (b >> 16) & 255;
  r.mv[0][0] = (int)(short)(c & 0xffffu); r.mv[0][1] = (int)(short)(c >> 16);
  r.mv[1][0] = (int)(short)(d & 0xffffu); r.mv[1][1] = (int)(short)(d >> 16);
  r.ref[0] = e & 255; r.ref[1] = (e >> 8) & 255;
  return r;
}

// is_tu_boundary (filter.c:190-206) || is_pu_boundary (:216-243)
template <int DIR>
__device__ __forceinline__ bool edge_is_boundary(const frame_t &f, int x, int y, bool &tu_boundary)
{
  const u32 scu = cu_head(f, x, y);
  const int tu_width = 64 >> (scu >> 24), pos = DIR ? y : x;
  tu_boundary = (pos & (tu_width - 1)) == 0;
  if (tu_boundary) return true;
  const int cu_width = 64 >> ((scu >> 8) & 7), x_cu = x & ~(cu_width - 1), y_cu = y & ~(cu_width - 1);
  const int part = (cu_head(f, x_cu, y_cu) >> 16) & 7;
  bool hit = false;
  for (int i = 0; i < c_num_parts[part]; ++i) {
    const int at = DIR ? y_cu + c_part_off_y[part][i] * cu_width / 4 : x_cu + c_part_off_x[part][i] * cu_width / 4;
    hit |= at == pos;
  }
  return hit;
}
// get_qp_y_pred (:263-282)
template <int DIR>
__device__ __forceinline__ int edge_qp(const frame_t &f, const kvz_hip_deblock_params &prm, int x, int y)
{
  if (!prm.per_cu_qp) return prm.qp;
  int qp_p;
  if (DIR && y > 0) qp_p = cu_qp(f, x, y - 1);
  else if (!DIR && x > 0) qp_p = cu_qp(f, x - 1, y);
  else qp_p = prm.frame_qp;
  const int qp_q = cu_qp(f, x, y);
  return (qp_p + qp_q + 1) >> 1;
}
__device__ __forceinline__ bool mv_far(const int *a, const int *b) { return abs(a[0] - b[0]) >= 4 || abs(a[1] - b[1]) >= 4; }
// the boundary strength of filter_deblock_edge_luma (:379-460)
__device__ __forceinline__ int edge_strength(const kvz_hip_deblock_params &prm, const cu_t &p, const cu_t &q, bool tu_boundary)
{
  if (q.type == 1 || p.type == 1) return 2;
  if (tu_boundary && (q.cbf_y || p.cbf_y)) return 1;
  if (p.mv_dir != 3 && q.mv_dir != 3) {
    const int lp = (p.mv_dir - 1) & 1, lq = (q.mv_dir - 1) & 1;
    if (mv_far(lq ? q.mv[1] : q.mv[0], lp ? p.mv[1] : p.mv[0])) return 1;
    if ((lq ? q.ref[1] : q.ref[0]) != (lp ? p.ref[1] : p.ref[0])) return 1;
  }
  if (!prm.slice_is_b) return 0;
  int mvp[2][2], mvq[2][2];                                  // undefined vectors count as zero (:400-417)
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      mvp[l][k] = (p.mv_dir & (1 << l)) ? p.mv[l][k] : 0;
      mvq[l][k] = (q.mv_dir & (1 << l)) ? q.mv[l][k] : 0;
    }
  const int refP0 = (p.mv_dir & 1) ? prm.ref_LX[0][p.ref[0] & 15] : -1, refP1 = (p.mv_dir & 2) ? prm.ref_LX[1][p.ref[1] & 15] : -1;
  const int refQ0 = (q.mv_dir & 1) ? prm.ref_LX[0][q.ref[0] & 15] : -1, refQ1 = (q.mv_dir & 2) ? prm.ref_LX[1][q.ref[1] & 15] : -1;
  if ((refP0 == refQ0 && refP1 == refQ1) || (refP0 == refQ1 && refP1 == refQ0)) {
    if (refP0 != refP1) {
      if (refP0 == refQ0) return (mv_far(mvq[0], mvp[0]) || mv_far(mvq[1], mvp[1])) ? 1 : 0;
      return (mv_far(mvq[1], mvp[0]) || mv_far(mvq[0], mvp[1])) ? 1 : 0;
    }
    return ((mv_far(mvq[0], mvp[0]) || mv_far(mvq[1], mvp[1])) && (mv_far(mvq[1], mvp[0]) || mv_far(mvq[0], mvp[1]))) ? 1 : 0;
  }
  return 1;
}

// one line of 8 pixels across the edge: b[0..3] = p3..p0, b[4..7] = q0..q3 (:83-153)
__device__ __forceinline__ void filter_line(int (&m)[8], bool strong, int tc, bool p_2nd, bool q_2nd)
{
  if (strong) {
    const int m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7];
    m[1] = clampi((2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3, m1 - 2 * tc, m1 + 2 * tc);
    m[2] = clampi((m1 + m2 + m3 + m4 + 2) >> 2, m2 - 2 * tc, m2 + 2 * tc);
    m[3] = clampi((m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3, m3 - 2 * tc, m3 + 2 * tc);
    m[4] = clampi((m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3, m4 - 2 * tc, m4 + 2 * tc);
    m[5] = clampi((m3 + m4 + m5 + m6 + 2) >> 2, m5 - 2 * tc, m5 + 2 * tc);
    m[6] = clampi((m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3, m6 - 2 * tc, m6 + 2 * tc);
  } else {
    const int m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6];
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (abs(delta) < tc * 10) {
      const int tc2 = tc >> 1;
      delta = clampi(delta, -tc, tc);
      m[3] = clampi(m3 + delta, 0, 255);
      m[4] = clampi(m4 - delta, 0, 255);
      if (p_2nd) m[2] = clampi(m2 + clampi((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1, -tc2, tc2), 0, 255);
      if (q_2nd) m[5] = clampi(m5 + clampi((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1, -tc2, tc2), 0, 255);
    }
  }
}

__device__ __forceinline__ void unpack8(u32 lo, u32 hi, int (&m)[8])
{
#pragma unroll
  for (int k = 0; k < 4; ++k) { m[k] = (int)((lo >> (8 * k)) & 255u); m[4 + k] = (int)((hi >> (8 * k)) & 255u); }
}
__device__ __forceinline__ u32 pack4(const int *m) { return (u32)m[0] | ((u32)m[1] << 8) | ((u32)m[2] << 16) | ((u32)m[3] << 24); }

template <int DIR>
__global__ __launch_bounds__(256) void deblock_pass_kernel(u8 *__restrict__ rec_y, u32 stride_y, u8 *__restrict__ rec_u, u8 *__restrict__ rec_v,
                                                          u32 stride_c, frame_t f, kvz_hip_deblock_params prm)
{
  const int uw = f.width >> 3, rows = (f.height >> 3) * 2;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)uw * rows) return;
  const int row = (int)(t / uw), ux = (int)(t - (size_t)row * uw) * 8, uy = (row >> 1) * 8, s = row & 1;
  if ((DIR == 0 && ux == 0) || (DIR == 1 && uy == 0)) return;            // filter_deblock_unit, :635-636

  // ---- luma segment s of the unit ----
  {
    // the second half of a horizontal edge at the right border of an LCU is the reference's deferred segment
    // (filter_deblock_lcu_rightmost, :711-731): its boundary flags and QP are taken at that half's own SCU
    const bool deferred = DIR == 1 && ((ux + 8) & 63) == 0 && ux + 8 != f.width;
    const int fx = (deferred && s == 1) ? ux + 4 : ux;
    bool tu_b;
    if (edge_is_boundary<DIR>(f, fx, uy, tu_b)) {
      const int sx = DIR ? ux + 4 * s : ux, sy = DIR ? uy : uy + 4 * s;
      const cu_t cp = DIR ? cu_load(f, sx, sy - 1) : cu_load(f, sx - 1, sy), cq = cu_load(f, sx, sy);
      const int bs = edge_strength(prm, cp, cq, tu_b);
      if (bs) {
        const int qp = edge_qp<DIR>(f, prm, fx, uy);
        const int beta = c_beta_table[clampi(qp + (prm.beta_offset_div2 << 1), 0, 51)];
        const int tc = c_tc_table[clampi(qp + 2 * (bs - 1) + (prm.tc_offset_div2 << 1), 0, 53)];
        // b[i][k]: line i (along the edge), sample k across it (p3 .. q3)
        int b[4][8];
        if (DIR == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const u32 *r = (const u32 *)(rec_y + (size_t)(sy + i) * stride_y + sx - 4);
            unpack8(r[0], r[1], b[i]);
          }
        } else {
          u32 rw[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) rw[k] = *(const u32 *)(rec_y + (size_t)(sy - 4 + k) * stride_y + sx);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 8; ++k) b[i][k] = (int)((rw[k] >> (8 * i)) & 255u);
        }
        const int dp0 = abs(b[0][1] - 2 * b[0][2] + b[0][3]), dq0 = abs(b[0][4] - 2 * b[0][5] + b[0][6]);
        const int dp3 = abs(b[3][1] - 2 * b[3][2] + b[3][3]), dq3 = abs(b[3][4] - 2 * b[3][5] + b[3][6]);
        const int dp = dp0 + dp3, dq = dq0 + dq3;
        if (dp + dq < beta) {
          const bool sw = 2 * (dp0 + dq0) < (beta >> 2) && 2 * (dp3 + dq3) < (beta >> 2) &&
                          abs(b[0][3] - b[0][4]) < ((5 * tc + 1) >> 1) && abs(b[3][3] - b[3][4]) < ((5 * tc + 1) >> 1) &&
                          abs(b[0][0] - b[0][3]) + abs(b[0][4] - b[0][7]) < (beta >> 3) &&
                          abs(b[3][0] - b[3][3]) + abs(b[3][4] - b[3][7]) < (beta >> 3);
          const int side = (beta + (beta >> 1)) >> 3;
#pragma unroll
          for (int i = 0; i < 4; ++i) filter_line(b[i], sw, tc, dp < side, dq < side);
          // the segment's 8 x 4 pixels belong to this thread alone in this pass: whole dwords go back
          if (DIR == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              u32 *r = (u32 *)(rec_y + (size_t)(sy + i) * stride_y + sx - 4);
              r[0] = pack4(&b[i][0]); r[1] = pack4(&b[i][4]);
            }
          } else {
#pragma unroll
            for (int k = 1; k < 7; ++k) {
              const int col[4] = { b[0][k], b[1][k], b[2][k], b[3][k] };
              *(u32 *)(rec_y + (size_t)(sy - 4 + k) * stride_y + sx) = pack4(col);
            }
          }
        }
      }
    }
  }

  // ---- chroma: edges on the 8x8 chroma grid next to an intra CU (:554-615); thread s = 0 takes U, s = 1 takes V ----
  if (prm.chroma && ((DIR ? uy : ux) & 15) == 0) {
    bool tu_b;
    if (!edge_is_boundary<DIR>(f, ux, uy, tu_b)) return;
    const u32 hp = DIR ? cu_head(f, ux, uy - 2) : cu_head(f, ux - 2, uy), hq = cu_head(f, ux, uy);
    if ((hp & 255u) != 1u && (hq & 255u) != 1u) return;
    const int qpc = c_chroma_scale[clampi(edge_qp<DIR>(f, prm, ux, uy), 0, 57)];
    const int tc = c_tc_table[clampi(qpc + 2 + (prm.tc_offset_div2 << 1), 0, 53)];
    u8 *plane = s ? rec_v : rec_u;
    const int xc = ux >> 1, yc = uy >> 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u8 *p = DIR ? plane + (size_t)yc * stride_c + xc + i : plane + (size_t)(yc + i) * stride_c + xc;
      const ptrdiff_t xs = DIR ? (ptrdiff_t)stride_c : 1;
      const int m2 = p[-2 * xs], m3 = p[-xs], m4 = p[0], m5 = p[xs];
      const int delta = clampi((((m4 - m3) * 4) + m2 - m5 + 4) >> 3, -tc, tc);      // kvz_filter_deblock_chroma, :158-180
      p[-xs] = (u8)clampi(m3 + delta, 0, 255);
      p[0] = (u8)clampi(m4 - delta, 0, 255);
    }
  }
}

}  // namespace

extern "C" int kvz_hip_deblock_frame(kvz_hip_pixel *rec_y, uint32_t stride_y, kvz_hip_pixel *rec_u, kvz_hip_pixel *rec_v, uint32_t stride_c,
                                     int width, int height, const kvz_hip_cu_info *cus, const kvz_hip_deblock_params *params,
                                     kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!rec_y || !cus || !params || width < 8 || height < 8 || ((width | height) & 7) || (stride_y & 3) || ((uintptr_t)rec_y & 3) ||
      ((uintptr_t)cus & 3) || stride_y < (uint32_t)width) {
    set_error_msg("kvz_hip_deblock_frame: planes and the SCU map must be 4-byte aligned, width / height multiples of 8 (the minimum CU)");
    return KVZ_HIP_ERR_INVALID;
  }
  if (params->chroma && (!rec_u || !rec_v || stride_c < (uint32_t)(width >> 1) || (stride_c & 3) || (((uintptr_t)rec_u | (uintptr_t)rec_v) & 3)))
    return kvzhip::invalid_arg(__func__);
  const frame_t f = { (const u32 *)cus, (width + 3) >> 2, width, height };
  const size_t threads = (size_t)(width >> 3) * (size_t)(height >> 3) * 2;
  const unsigned grid = (unsigned)((threads + 255) / 256);
  hipStream_t st = ctx_stream(s);
  hipLaunchKernelGGL(deblock_pass_kernel<0>, dim3(grid), dim3(256), 0, st, rec_y, stride_y, rec_u, rec_v, stride_c, f, *params);
  KVZ_CHECK_LAUNCH("deblock_pass_kernel<vertical edges>");
  hipLaunchKernelGGL(deblock_pass_kernel<1>, dim3(grid), dim3(256), 0, st, rec_y, stride_y, rec_u, rec_v, stride_c, f, *params);
  KVZ_CHECK_LAUNCH("deblock_pass_kernel<horizontal edges>");
  return KVZ_HIP_OK;
}
