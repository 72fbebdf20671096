// ipol.hip -- sub-pel interpolation for gfx950: the 8-tap luma / 4-tap chroma
// sample filters and the fused fractional motion search.
//
// Reference: src/strategies/generic/ipol-generic.c (cited per kernel), taps from
// src/filter.c:54-72, caller src/search_inter.c:965-1128.
//
// One workgroup owns one block: the (w+taps-1) x (h+taps-1) source window is
// fetched from HBM once with edge replication (kvz_get_extended_block semantics,
// ipol-generic.c:731-784), staged in LDS, and every intermediate (horizontal
// pass planes, candidate blocks) stays on chip.
#include "kvz_hip_internal.h"

using namespace kvzhip;

__constant__ signed char c_luma_filter[4][8] = {       // filter.c:54-60
  { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ signed char c_chroma_filter[8][4] = {     // filter.c:62-72
  { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
  { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

struct refplane_t { const u8 *p; u32 stride; int w, h; };

__device__ __forceinline__ u8 ref_px(const refplane_t &r, int x, int y)
{
  return r.p[(size_t)clampi(y, 0, r.h - 1) * r.stride + clampi(x, 0, r.w - 1)];
}

// (int16 sample + 32) >> 6 through the int16-argument clip (ipol-generic.c:285-287 etc.)
__device__ __forceinline__ u8 round_clip16(i16 sample) { return fast_clip16((i16)(((int)sample + 32) >> 6)); }

// ---------------------------------------------------------------------------
// kvz_sample_quarterpel_luma / kvz_sample_octpel_chroma (+14-bit variants),
// ipol-generic.c:122-190, :660-728: horizontal pass over h+TAPS-1 rows into
// int16, vertical pass >> 6, then (+32) >> 6 and the 32-bit clip (or the raw
// 14-bit sample).  Always both passes, like the reference.
// ---------------------------------------------------------------------------
template <int TAPS, bool OUT14, int MAXW, int T, bool WAVE>
__device__ __forceinline__ void sample_core(int tid, u8 *s_win, i16 *s_hor, const refplane_t &ref, const kvz_hip_ipol_block &b,
                                            size_t o, void *__restrict__ dst)
{
  constexpr int OFF = TAPS / 2 - 1;                 // 3 luma, 1 chroma
  constexpr int WS = MAXW + TAPS;
  const int w = b.width, h = b.height;
  const int ww = w + TAPS - 1, wh = h + TAPS - 1;
  const signed char *hf = TAPS == 8 ? c_luma_filter[b.mv_frac_x & 3] : c_chroma_filter[b.mv_frac_x & 7];
  const signed char *vf = TAPS == 8 ? c_luma_filter[b.mv_frac_y & 3] : c_chroma_filter[b.mv_frac_y & 7];
  {
    // window rows as (unaligned) dwords when the dword-rounded window lies inside the frame,
    // else byte by byte with edge replication (kvz_get_extended_block, ipol-generic.c:731-784)
    const int x0 = b.x - OFF, y0 = b.y - OFF, wq = (ww + 3) >> 2;
    if (x0 >= 0 && y0 >= 0 && x0 + 4 * wq <= ref.w && y0 + wh <= ref.h) {
      for (int i = tid; i < wq * wh; i += T) {
        const int y = i / wq, q = i - y * wq;
        u32 v;
        __builtin_memcpy(&v, ref.p + (size_t)(y0 + y) * ref.stride + x0 + 4 * q, 4);
        *(u32 *)(s_win + y * WS + 4 * q) = v;
      }
    } else {
      for (int i = tid; i < ww * wh; i += T) {
        const int y = i / ww, x = i - y * ww;
        s_win[y * WS + x] = ref_px(ref, x0 + x, y0 + y);
      }
    }
  }
  if (WAVE) wave_lds_fence(); else __syncthreads();
  for (int i = tid; i < w * wh; i += T) {
    const int y = i / w, x = i - y * w;
    int acc = 0;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc += hf[t] * (int)s_win[y * WS + x + t];
    s_hor[y * MAXW + x] = (i16)acc;
  }
  if (WAVE) wave_lds_fence(); else __syncthreads();
  if ((w & 3) == 0) {                               // four outputs per lane, one 4- or 8-byte store
    const int w4 = w >> 2;
    for (int i = tid; i < w4 * h; i += T) {
      const int y = i / w4, x = (i - y * w4) << 2;
      int v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        int acc = 0;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc += vf[t] * (int)s_hor[(y + t) * MAXW + x + k];
        v[k] = acc >> 6;
      }
      if (OUT14) {
        const uint2 pk = make_uint2((u32)(v[0] & 0xffff) | ((u32)v[1] << 16), (u32)(v[2] & 0xffff) | ((u32)v[3] << 16));
        __builtin_memcpy((i16 *)dst + o + (size_t)y * w + x, &pk, 8);
      } else {
        const u32 pk = (u32)fast_clip32((v[0] + 32) >> 6) | ((u32)fast_clip32((v[1] + 32) >> 6) << 8) |
                       ((u32)fast_clip32((v[2] + 32) >> 6) << 16) | ((u32)fast_clip32((v[3] + 32) >> 6) << 24);
        __builtin_memcpy((u8 *)dst + o + (size_t)y * w + x, &pk, 4);
      }
    }
    return;
  }
  for (int i = tid; i < w * h; i += T) {
    const int y = i / w, x = i - y * w;
    int acc = 0;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) acc += vf[t] * (int)s_hor[(y + t) * MAXW + x];
    acc >>= 6;
    if (OUT14) ((i16 *)dst)[o + i] = (i16)acc;
    else ((u8 *)dst)[o + i] = fast_clip32((acc + 32) >> 6);
  }
}

// blocks wider or taller than 16: one workgroup per block
template <int TAPS, bool OUT14>
__global__ __launch_bounds__(256) void sample_big_kernel(refplane_t ref, const kvz_hip_ipol_block *__restrict__ blocks,
                                                         const unsigned long long *__restrict__ out_offsets, void *__restrict__ dst)
{
  constexpr int MAXW = TAPS == 8 ? 64 : 32;
  __shared__ u8 s_win[(MAXW + TAPS - 1) * (MAXW + TAPS)];
  __shared__ i16 s_hor[(MAXW + TAPS - 1) * MAXW];
  const kvz_hip_ipol_block b = blocks[blockIdx.x];
  if (b.width < 1 || b.height < 1 || b.width > MAXW || b.height > MAXW) return;     // unsupported shape: nothing written
  if (b.width <= 16 && b.height <= 16) return;                                      // sample_small_kernel's
  sample_core<TAPS, OUT14, MAXW, 256, false>(threadIdx.x, s_win, s_hor, ref, b, (size_t)out_offsets[blockIdx.x], dst);
}

// blocks up to 16x16: one wave per block, four blocks per workgroup, wave-private LDS, no barrier
template <int TAPS, bool OUT14>
__global__ __launch_bounds__(256) void sample_small_kernel(refplane_t ref, const kvz_hip_ipol_block *__restrict__ blocks, size_t count,
                                                           const unsigned long long *__restrict__ out_offsets, void *__restrict__ dst)
{
  __shared__ u8 s_win[4][(16 + TAPS - 1) * (16 + TAPS)];
  __shared__ i16 s_hor[4][(16 + TAPS - 1) * 16];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: descriptor loads and block geometry go scalar
  const size_t i = (size_t)blockIdx.x * 4 + wv;
  if (i >= count) return;
  const kvz_hip_ipol_block b = blocks[i];
  if (b.width < 1 || b.height < 1 || b.width > 16 || b.height > 16) return;
  sample_core<TAPS, OUT14, 16, 64, true>(threadIdx.x & 63, s_win[wv], s_hor[wv], ref, b, (size_t)out_offsets[i], dst);
}

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
// SATD by one lane per (candidate, 8x8 sub-block).
// ---------------------------------------------------------------------------
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s f_unpack_lo(u32 d) { return __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, d, 0x0c010c00u)); }
__device__ __forceinline__ v2s f_unpack_hi(u32 d) { return __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, d, 0x0c030c02u)); }
__device__ __forceinline__ u32 f_absmax(v2s x)
{
  v2s ax = __builtin_elementwise_max(x, -x);
  u32 w = __builtin_bit_cast(u32, ax);
  u32 lo = w & 0xffffu, hi = w >> 16;
  return lo > hi ? lo : hi;
}
// 8x8 Hadamard SATD of two LDS blocks (picture-generic.c:240-328), rows of 8 bytes
__device__ __forceinline__ u32 satd8x8_lds(const u8 *a, int sa, const u8 *b, int sb)
{
  v2s x[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    u32 a0, a1, b0, b1;
    __builtin_memcpy(&a0, a + r * sa, 4); __builtin_memcpy(&a1, a + r * sa + 4, 4);
    __builtin_memcpy(&b0, b + r * sb, 4); __builtin_memcpy(&b1, b + r * sb + 4, 4);
    x[r][0] = f_unpack_lo(a0) - f_unpack_lo(b0); x[r][1] = f_unpack_hi(a0) - f_unpack_hi(b0);
    x[r][2] = f_unpack_lo(a1) - f_unpack_lo(b1); x[r][3] = f_unpack_hi(a1) - f_unpack_hi(b1);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    v2s s0 = x[r][0] + x[r][2], s1 = x[r][1] + x[r][3], d0 = x[r][0] - x[r][2], d1 = x[r][1] - x[r][3];
    x[r][0] = s0 + s1; x[r][1] = s0 - s1; x[r][2] = d0 + d1; x[r][3] = d0 - d1;
  }
  u32 m = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v2s t[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { t[r] = x[r][q] + x[r + 4][q]; t[r + 4] = x[r][q] - x[r + 4][q]; }
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      v2s u0 = t[h] + t[h + 2], u1 = t[h + 1] + t[h + 3], u2 = t[h] - t[h + 2], u3 = t[h + 1] - t[h + 3];
      m += f_absmax(u0 + u1) + f_absmax(u0 - u1) + f_absmax(u2 + u3) + f_absmax(u2 - u3);
    }
  }
  return (m + 1) >> 1;
}

#define FR_HS 65                 /* H plane row stride of the per-call filter step kernel */

struct frac_cand { int fx, fy, ry, cx; };
// square[] of search_inter.c:972-976
__constant__ signed char c_sq_x[9] = { 0, -1, 1, 0, 0, -1, 1, -1, 1 };
__constant__ signed char c_sq_y[9] = { 0, 0, 0, -1, 1, -1, -1, 1, 1 };

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

template <int MAXW, int T, bool WAVE>
__device__ __forceinline__ void search_frac_core(int tid, u8 *lds, const u8 *__restrict__ pic, u32 pic_stride, const refplane_t &ref,
                                                 const kvz_hip_block_pair &d, u32 *__restrict__ out, i32 *__restrict__ best)
{
  typedef frac_geom<MAXW> G;
  u8 *s_p = lds, *s_cur = s_p + G::P_BYTES, *s_cand = s_cur + G::CUR_BYTES;
  i16 *s_h = (i16 *)(lds + ((G::P_BYTES + G::CUR_BYTES + 4 * G::CAND_BYTES + 15) & ~15));
  u32 *s_cost = (u32 *)(s_h + 3 * G::H_ELEMS);
  int *s_sel = (int *)(s_cost + 4);
  auto sync = [&]() { if (WAVE) wave_lds_fence(); else __syncthreads(); };

  const int w = d.width, h = d.height;
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
  // SATD of candidates 0..ncand-1 against s_cur -> s_cost
  auto score = [&](int ncand, const u8 *cand0, int cand_stride, int cand_pitch) {
    if (tid < 4) s_cost[tid] = 0;
    sync();
    const int w8 = w >> 3, n8 = w8 * (h >> 3);
    for (int i = tid; i < ncand * n8; i += T) {
      const int k = i / n8, sb = i - k * n8, by = sb / w8, bx = sb - by * w8;
      const u32 v = satd8x8_lds(s_cur + by * 8 * G::CS + bx * 8, G::CS, cand0 + (size_t)k * cand_pitch + by * 8 * cand_stride + bx * 8, cand_stride);
      atomicAdd(&s_cost[k], v);
    }
    sync();
  };

  // integer position: candidate = P[y][x]
  score(1, s_p + 4 * G::PS + 4, G::PS, 0);
  u32 best_cost = s_cost[0];
  if (tid == 0) out[0] = best_cost;

  hor_plane(2, s_h);
  sync();

  int best_index = 0;
  for (int step = 0; step < 4; ++step) {
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
    // decision: same order and strict '<' as search_inter.c:1096-1102
    const int i0 = (step & 1) ? 5 : 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 cj = s_cost[j];
      if (tid == 0) out[(step >= 2 ? 8 : 0) + i0 + j] = cj;
      if (cj < best_cost) { best_cost = cj; best_index = i0 + j; }
    }
    if (step == 1 || step == 3) {
      if (tid == 0) best[step == 3] = best_index;
      if (step == 1) {
        sync();
        if (tid == 0) { s_sel[0] = c_sq_x[best_index]; s_sel[1] = c_sq_y[best_index]; }
        best_index = 0;
      }
    }
    sync();
  }
}

__device__ __forceinline__ bool frac_shape_ok(int w, int h) { return !(w < 8 || h < 8 || w > 64 || h > 64 || ((w | h) & 7)); }

// blocks larger than 16x16 (and malformed descriptors, which are flagged): one workgroup per descriptor
__global__ __launch_bounds__(256) void search_frac_big_kernel(const u8 *__restrict__ pic, u32 pic_stride, refplane_t ref,
                                                              const kvz_hip_block_pair *__restrict__ pairs,
                                                              u32 *__restrict__ costs, i32 *__restrict__ best)
{
  __shared__ __attribute__((aligned(16))) u8 lds[frac_geom<64>::TOTAL];
  const kvz_hip_block_pair d = pairs[blockIdx.x];
  const int tid = threadIdx.x;
  if (!frac_shape_ok(d.width, d.height)) {            // unsupported shape: flag it, touch nothing else
    if (tid < 17) costs[(size_t)blockIdx.x * 17 + tid] = 0xffffffffu;
    if (tid < 2) best[(size_t)blockIdx.x * 2 + tid] = -1;
    return;
  }
  if (d.width <= 16 && d.height <= 16) return;        // handled by search_frac_small_kernel
  search_frac_core<64, 256, false>(tid, lds, pic, pic_stride, ref, d, costs + (size_t)blockIdx.x * 17, best + (size_t)blockIdx.x * 2);
}

// blocks up to 16x16: one wave per descriptor, four descriptors per workgroup, no barrier
__global__ __launch_bounds__(256) void search_frac_small_kernel(const u8 *__restrict__ pic, u32 pic_stride, refplane_t ref,
                                                                const kvz_hip_block_pair *__restrict__ pairs, size_t count,
                                                                u32 *__restrict__ costs, i32 *__restrict__ best)
{
  __shared__ __attribute__((aligned(16))) u8 lds[4][(frac_geom<16>::TOTAL + 15) & ~15];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: descriptor loads and block geometry go scalar
  const size_t i = (size_t)blockIdx.x * 4 + wv;
  if (i >= count) return;
  const kvz_hip_block_pair d = pairs[i];
  if (!frac_shape_ok(d.width, d.height) || d.width > 16 || d.height > 16) return;
  search_frac_core<16, 64, true>(threadIdx.x & 63, lds[wv], pic, pic_stride, ref, d, costs + i * 17, best + i * 2);
}

// ---------------------------------------------------------------------------
// One reference filter step (ipol_blocks_func, strategies-ipol.h:36-38) for the
// per-call strategy shim: produces exactly what the generic step writes -- the
// four filtered blocks and the horizontal planes / first-column arrays the
// following steps read from the caller's scratch (ipol-generic.c:192-658).
// `win` is the window the reference reads around `src`: rows -3 .. h+4,
// cols -3 .. w+5 (stride w + 9), i.e. P rows -4 .. h+3, P cols -4 .. w+4.
// Planes are emitted compactly: hor_out[p][(h+8) * w], cols_out[p][h+8].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void frac_step_kernel(const u8 *__restrict__ win, int w, int h, int step, int fme_level,
                                                        int hx, int hy, u8 *__restrict__ filtered /*[4][h*w]*/,
                                                        i16 *__restrict__ hor_out /*[2][(h+8)*w]*/, i16 *__restrict__ cols_out /*[2][h+8]*/)
{
  __shared__ u8 s_p[72 * 76];
  __shared__ i16 s_h[2][72 * FR_HS];
  const int tid = threadIdx.x, ph = h + 8, pw = w + 9, PS = 76;
  for (int i = tid; i < ph * pw; i += 256) { const int y = i / pw, x = i - y * pw; s_p[y * PS + x] = win[i]; }
  __syncthreads();
  // H plane of filter f over rows -4 .. h+3 (index y), cols -1 .. w-1 (index x = c + 1)
  auto hor_plane = [&](int f, i16 *dst) {
    const signed char *fl = c_luma_filter[f];
    for (int i = tid; i < ph * (w + 1); i += 256) {
      const int y = i / (w + 1), x = i - y * (w + 1);
      int acc = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += fl[t] * (int)s_p[y * PS + x + t];
      dst[y * FR_HS + x] = (i16)acc;
    }
  };
  auto emit_plane = [&](const i16 *pl, int slot, int first_y) {        // hor_intermediate / hor_first_cols contents
    for (int i = tid; i < ph * (w + 1); i += 256) {
      const int y = i / (w + 1), x = i - y * (w + 1);
      if (y < first_y) continue;
      if (x == 0) cols_out[slot * ph + y] = pl[y * FR_HS];
      else hor_out[(size_t)slot * ph * w + y * w + (x - 1)] = pl[y * FR_HS + x];
    }
  };
  auto filter_cand = [&](int fy, int ry, int cx, const i16 *pl, u8 *dst) {
    const signed char *vf = c_luma_filter[fy];
    for (int i = tid; i < w * h; i += 256) {
      const int y = i / w, x = i - y * w, r = y + ry, cc = x + cx;
      int acc = 0;
      if (!pl) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += vf[j] * 64 * (int)s_p[(r + 1 + j) * PS + cc + 4];
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += vf[j] * (int)pl[(r + 1 + j) * FR_HS + cc + 1];
      }
      dst[y * w + x] = round_clip16((i16)(acc >> 6));
    }
  };
  u8 *f0 = filtered, *f1 = filtered + w * h, *f2 = filtered + 2 * w * h, *f3 = filtered + 3 * w * h;
  if (step < 2) {
    hor_plane(2, s_h[0]);
    __syncthreads();
    if (step == 0) {
      // hor_intermediate[0] / hor_first_cols[0] hold the fir0 plane = 64 * P (ipol-generic.c:226-241)
      for (int i = tid; i < ph * (w + 1); i += 256) {
        const int y = i / (w + 1), x = i - y * (w + 1);
        const i16 v = (i16)(64 * (int)s_p[y * PS + x + 3]);
        if (x == 0) cols_out[y] = v; else hor_out[y * w + (x - 1)] = v;
      }
      emit_plane(s_h[0], 1, fme_level > 1 ? 0 : 1);                    // hor_intermediate[1] / hor_first_cols[2]
      filter_cand(0, 0, -1, s_h[0], f0); filter_cand(0, 0, 0, s_h[0], f1);
      filter_cand(2, -1, 0, nullptr, f2); filter_cand(2, 0, 0, nullptr, f3);
    } else {
      filter_cand(2, -1, -1, s_h[0], f0); filter_cand(2, -1, 0, s_h[0], f1);
      filter_cand(2, 0, -1, s_h[0], f2); filter_cand(2, 0, 0, s_h[0], f3);
    }
  } else {
    const int bx = 2 * hx, by = 2 * hy;
    hor_plane((bx - 1) & 3, s_h[0]);
    hor_plane((bx + 1) & 3, s_h[1]);
    __syncthreads();
    if (step == 2) {
      emit_plane(s_h[0], 0, 0);                                        // hor_intermediate[3] / hor_first_cols[1]
      emit_plane(s_h[1], 1, 0);                                        // hor_intermediate[4] / hor_first_cols[3]
      filter_cand(by & 3, by >> 2, (bx - 1) >> 2, s_h[0], f0);
      filter_cand(by & 3, by >> 2, (bx + 1) >> 2, s_h[1], f1);
      __syncthreads();
      // top / bottom use the half-pel column plane: fx = bx & 3 (2 -> recompute H_2, 0 -> pixels)
      if (bx & 3) { hor_plane(2, s_h[0]); __syncthreads(); }
      const i16 *hp = (bx & 3) ? s_h[0] : nullptr;
      filter_cand((by - 1) & 3, (by - 1) >> 2, bx >> 2, hp, f2);
      filter_cand((by + 1) & 3, (by + 1) >> 2, bx >> 2, hp, f3);
    } else {
      filter_cand((by - 1) & 3, (by - 1) >> 2, (bx - 1) >> 2, s_h[0], f0);
      filter_cand((by - 1) & 3, (by - 1) >> 2, (bx + 1) >> 2, s_h[1], f1);
      filter_cand((by + 1) & 3, (by + 1) >> 2, (bx - 1) >> 2, s_h[0], f2);
      filter_cand((by + 1) & 3, (by + 1) >> 2, (bx + 1) >> 2, s_h[1], f3);
    }
  }
}

namespace kvzhip {
int launch_frac_step(const u8 *win, int w, int h, int step, int fme_level, int hx, int hy,
                     u8 *filtered, i16 *hor_out, i16 *cols_out, hipStream_t st)
{
  if (w < 8 || h < 8 || w > 64 || h > 64 || ((w | h) & 7) || step < 0 || step > 3) return KVZ_HIP_ERR_INVALID;
  hipLaunchKernelGGL(frac_step_kernel, dim3(1), dim3(256), 0, st, win, w, h, step, fme_level, hx, hy, filtered, hor_out, cols_out);
  KVZ_CHECK_LAUNCH("frac_step_kernel");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip

extern "C" {

static int sample_launch(bool luma, const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                         const kvz_hip_ipol_block *blocks, const uint64_t *out_offsets, size_t count,
                         int out_14bit, void *dst, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!ref || !blocks || !out_offsets || !dst || ref_w <= 0 || ref_h <= 0) return KVZ_HIP_ERR_INVALID;
  if (count == 0) return KVZ_HIP_OK;
  if (count > 0x7fffffffu) return KVZ_HIP_ERR_INVALID;
  refplane_t r = { ref, ref_stride, ref_w, ref_h };
  hipStream_t st = ctx_stream(s);
  const unsigned long long *oo = (const unsigned long long *)out_offsets;
  const unsigned gs = (unsigned)((count + 3) / 4), gb = (unsigned)count;
#define KVZ_SAMPLE(TAPS, O14)                                                                                        \
  do {                                                                                                               \
    hipLaunchKernelGGL((sample_small_kernel<TAPS, O14>), dim3(gs), dim3(256), 0, st, r, blocks, count, oo, dst);     \
    hipLaunchKernelGGL((sample_big_kernel<TAPS, O14>), dim3(gb), dim3(256), 0, st, r, blocks, oo, dst);              \
  } while (0)
  if (luma) { if (out_14bit) KVZ_SAMPLE(8, true); else KVZ_SAMPLE(8, false); }
  else { if (out_14bit) KVZ_SAMPLE(4, true); else KVZ_SAMPLE(4, false); }
#undef KVZ_SAMPLE
  KVZ_CHECK_LAUNCH("sample_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_sample_luma_batch(const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                              const kvz_hip_ipol_block *blocks, const uint64_t *out_offsets, size_t count,
                              int out_14bit, void *dst, kvz_hip_stream s)
{
  return sample_launch(true, ref, ref_stride, ref_w, ref_h, blocks, out_offsets, count, out_14bit, dst, s);
}

int kvz_hip_sample_chroma_batch(const kvz_hip_pixel *ref, uint32_t ref_stride, int ref_w, int ref_h,
                                const kvz_hip_ipol_block *blocks, const uint64_t *out_offsets, size_t count,
                                int out_14bit, void *dst, kvz_hip_stream s)
{
  return sample_launch(false, ref, ref_stride, ref_w, ref_h, blocks, out_offsets, count, out_14bit, dst, s);
}

int kvz_hip_search_frac_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, const kvz_hip_pixel *ref, uint32_t ref_stride,
                              int ref_w, int ref_h, const kvz_hip_block_pair *pairs, size_t count,
                              uint32_t *costs, int32_t *best, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!pic || !ref || !pairs || !costs || !best || ref_w <= 0 || ref_h <= 0) return KVZ_HIP_ERR_INVALID;
  if (count == 0) return KVZ_HIP_OK;
  if (count > 0x7fffffffu) return KVZ_HIP_ERR_INVALID;
  refplane_t r = { ref, ref_stride, ref_w, ref_h };
  // two passes over the same descriptor list: each kernel takes the size class it is built for and skips the rest
  hipLaunchKernelGGL(search_frac_small_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, ctx_stream(s), pic, pic_stride, r, pairs, count, costs, best);
  KVZ_CHECK_LAUNCH("search_frac_small_kernel");
  hipLaunchKernelGGL(search_frac_big_kernel, dim3((unsigned)count), dim3(256), 0, ctx_stream(s), pic, pic_stride, r, pairs, costs, best);
  KVZ_CHECK_LAUNCH("search_frac_big_kernel");
  return KVZ_HIP_OK;
}

}  // extern "C"
