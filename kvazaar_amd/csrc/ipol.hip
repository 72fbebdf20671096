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
#include "frac_core.h"

using namespace kvzhip;

__constant__ signed char c_chroma_filter[8][4] = {     // filter.c:62-72
  { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
  { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };


// The same two passes for the shapes motion compensation really asks for (width a multiple of 4, even height), with the
// arithmetic of frac_core.h: the horizontal pass cuts the byte windows of 4 neighbouring samples out of aligned dwords
// (v_alignbyte) and takes each sample as one (chroma) or two (luma) v_dot4_i32_i8 on pixels - 128 (the taps sum to 64, so
// + 128 * 64 restores the offset; exact); its int16 plane is stored TRANSPOSED, two vertically adjacent samples per dword,
// so the vertical pass is v_dot2_i32_i16 on consecutive dwords -- a lane produces 2 rows x 4 columns from 3 (chroma) or
// 5 (luma) dwords per column instead of TAPS 16-bit reads per sample.
template <int TAPS, int MAXW> struct sample_geom {
  static constexpr int WS = MAXW + TAPS;                       // window row stride (a multiple of 4)
  static constexpr int WIN_BYTES = (MAXW + TAPS) * WS;         // one row more than the window: the last row pair may be half
  static constexpr int HP = (MAXW + TAPS) / 2 + 1;             // dwords per transposed column (odd: columns start in different banks)
  static constexpr int HOR_DWORDS = MAXW * HP > (MAXW + TAPS - 1) * MAXW / 2 ? MAXW * HP : (MAXW + TAPS - 1) * MAXW / 2 + 1;
};
template <int TAPS, bool OUT14, int MAXW, int T, bool WAVE>
__device__ __forceinline__ void sample_core_fast(int tid, u8 *s_win, u32 *s_hor, const kvz_hip_ipol_block &b, size_t o, void *__restrict__ dst)
{
  typedef sample_geom<TAPS, MAXW> G;
  constexpr int HALF = TAPS / 2;
  const int w = b.width, h = b.height, wh = h + TAPS - 1;
  const signed char *hf = TAPS == 8 ? c_luma_filter[b.mv_frac_x & 3] : c_chroma_filter[b.mv_frac_x & 7];
  const signed char *vf = TAPS == 8 ? c_luma_filter[b.mv_frac_y & 3] : c_chroma_filter[b.mv_frac_y & 7];
  u32 h0 = 0, h1 = 0;                                          // horizontal taps as bytes
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h0 |= (u32)(u8)hf[t] << (8 * t);
    if (TAPS == 8) h1 |= (u32)(u8)hf[4 + t] << (8 * t);
  }
  // vertical taps as int16 pairs: va[t] = (f[2t], f[2t+1]) prices the row pair t for the even output row, vb[t] =
  // (f[2t-1], f[2t]) for the odd one, which starts half a pair later
  u32 va[HALF + 1], vb[HALF + 1];
#pragma unroll
  for (int t = 0; t <= HALF; ++t) {
    va[t] = t < HALF ? frac_pack16(vf[2 * t], vf[2 * t + 1]) : 0u;
    vb[t] = frac_pack16(t > 0 ? vf[2 * t - 1] : 0, t < HALF ? vf[2 * t] : 0);
  }
  const int npair = (wh + 1) >> 1, w4 = w >> 2;
  for (int i = tid; i < npair * w4; i += T) {
    const int g = i / npair, yp = i - g * npair, x0 = 4 * g;
    int hv[4][2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const u32 *q = (const u32 *)(s_win + (2 * yp + rr) * G::WS + x0);
      const u32 d0 = q[0] ^ 0x80808080u, d1 = q[1] ^ 0x80808080u, d2 = TAPS == 8 ? q[2] ^ 0x80808080u : 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32 lo = k ? __builtin_amdgcn_alignbyte(d1, d0, (u32)k) : d0;
        if (TAPS == 8) {
          const u32 hi = k ? __builtin_amdgcn_alignbyte(d2, d1, (u32)k) : d1;
          hv[k][rr] = __builtin_amdgcn_sdot4((int)h0, (int)lo, __builtin_amdgcn_sdot4((int)h1, (int)hi, 8192, false), false);
        } else {
          hv[k][rr] = __builtin_amdgcn_sdot4((int)h0, (int)lo, 8192, false);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_hor[(x0 + k) * G::HP + yp] = ((u32)hv[k][0] & 0xffffu) | ((u32)hv[k][1] << 16);
  }
  if (WAVE) wave_lds_fence(); else __syncthreads();
  const int h2 = h >> 1;
  for (int i = tid; i < h2 * w4; i += T) {
    const int g = i / h2, yo = i - g * h2, x0 = 4 * g;
    int v0[4], v1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32 *col = s_hor + (x0 + k) * G::HP + yo;
      int a0 = 0, a1 = 0;
#pragma unroll
      for (int t = 0; t <= HALF; ++t) {
        const v2s pr = as_v2s(col[t]);
        if (t < HALF) a0 = __builtin_amdgcn_sdot2(pr, as_v2s(va[t]), a0, false);
        a1 = __builtin_amdgcn_sdot2(pr, as_v2s(vb[t]), a1, false);
      }
      v0[k] = a0 >> 6; v1[k] = a1 >> 6;
    }
    const size_t at = o + (size_t)(2 * yo) * w + x0;
    if (OUT14) {
      const uint2 p0 = make_uint2((u32)(v0[0] & 0xffff) | ((u32)v0[1] << 16), (u32)(v0[2] & 0xffff) | ((u32)v0[3] << 16));
      const uint2 p1 = make_uint2((u32)(v1[0] & 0xffff) | ((u32)v1[1] << 16), (u32)(v1[2] & 0xffff) | ((u32)v1[3] << 16));
      __builtin_memcpy((i16 *)dst + at, &p0, 8);
      __builtin_memcpy((i16 *)dst + at + w, &p1, 8);
    } else {
      const u32 p0 = (u32)fast_clip32((v0[0] + 32) >> 6) | ((u32)fast_clip32((v0[1] + 32) >> 6) << 8) |
                     ((u32)fast_clip32((v0[2] + 32) >> 6) << 16) | ((u32)fast_clip32((v0[3] + 32) >> 6) << 24);
      const u32 p1 = (u32)fast_clip32((v1[0] + 32) >> 6) | ((u32)fast_clip32((v1[1] + 32) >> 6) << 8) |
                     ((u32)fast_clip32((v1[2] + 32) >> 6) << 16) | ((u32)fast_clip32((v1[3] + 32) >> 6) << 24);
      __builtin_memcpy((u8 *)dst + at, &p0, 4);
      __builtin_memcpy((u8 *)dst + at + w, &p1, 4);
    }
  }
}

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
  if ((w & 3) == 0 && (h & 1) == 0) {                  // every PU shape of motion compensation
    sample_core_fast<TAPS, OUT14, MAXW, T, WAVE>(tid, s_win, (u32 *)s_hor, b, o, dst);
    return;
  }
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

// 8x8 luma blocks owned by one wave -- the commonest motion-compensation block, and the worst fit of the 2 x 4-sample work items
// above: its 15 x 8 horizontal samples are 16 items and its 64 outputs 8 items, so a quarter and an eighth of the wave worked
// (246 vector instructions per block, instruction-issue bound: profiles/r02_z_frame_kernels_pmc.txt).  Here every lane works in
// both passes: horizontal = (window row l >> 2, two neighbouring columns 2 (l & 3), + 1), the byte windows cut from three aligned
// dwords with a per-lane v_alignbyte shift; vertical = one output sample per lane (x = l & 7, y = l >> 3) as five v_dot2_i32_i16
// down its transposed column, the coefficient pairs picked per lane by the parity of y.  Same arithmetic, same results.
template <bool OUT14>
__device__ __forceinline__ void sample8x8_luma_wave(int lane, u8 *s_win, u32 *s_hor, const refplane_t &ref, const kvz_hip_ipol_block &b,
                                                    size_t o, void *__restrict__ dst)
{
  typedef sample_geom<8, 16> G;
  constexpr int WS = 16 + 8;
  const signed char *hf = c_luma_filter[b.mv_frac_x & 3], *vf = c_luma_filter[b.mv_frac_y & 3];
  {
    // the 15 x 15 window: rows of 4 dwords when the dword-rounded window lies inside the frame, else bytes with edge replication
    const int x0 = b.x - 3, y0 = b.y - 3;
    if (x0 >= 0 && y0 >= 0 && x0 + 16 <= ref.w && y0 + 15 <= ref.h) {
      if (lane < 60) {
        const int y = lane >> 2, q = lane & 3;
        u32 v;
        __builtin_memcpy(&v, ref.p + (size_t)(y0 + y) * ref.stride + x0 + 4 * q, 4);
        *(u32 *)(s_win + y * WS + 4 * q) = v;
      }
    } else {
      for (int i = lane; i < 15 * 15; i += 64) {
        const int y = i / 15, x = i - y * 15;
        s_win[y * WS + x] = ref_px(ref, x0 + x, y0 + y);
      }
    }
  }
  wave_lds_fence();
  u32 h0 = 0, h1 = 0;
#pragma unroll
  for (int t = 0; t < 4; ++t) { h0 |= (u32)(u8)hf[t] << (8 * t); h1 |= (u32)(u8)hf[4 + t] << (8 * t); }
  {
    const int row = lane >> 2, c = 2 * (lane & 3);           // window row 0..15 (15: past the window), columns c, c + 1
    if (row < 15) {
      const u32 *q = (const u32 *)(s_win + row * WS + (c & ~3));
      const u32 d0 = q[0] ^ 0x80808080u, d1 = q[1] ^ 0x80808080u, d2 = q[2] ^ 0x80808080u, k = (u32)c & 3u;
      const u32 lo0 = __builtin_amdgcn_alignbyte(d1, d0, k), hi0 = __builtin_amdgcn_alignbyte(d2, d1, k);
      const u32 lo1 = __builtin_amdgcn_alignbyte(d1, d0, k + 1), hi1 = __builtin_amdgcn_alignbyte(d2, d1, k + 1);
      const int s0 = __builtin_amdgcn_sdot4((int)h0, (int)lo0, __builtin_amdgcn_sdot4((int)h1, (int)hi0, 8192, false), false);
      const int s1 = __builtin_amdgcn_sdot4((int)h0, (int)lo1, __builtin_amdgcn_sdot4((int)h1, (int)hi1, 8192, false), false);
      // transposed plane: column x at dword x * HP, rows (2t, 2t + 1) in dword t
      unsigned short *col0 = (unsigned short *)(s_hor + c * G::HP) + row, *col1 = (unsigned short *)(s_hor + (c + 1) * G::HP) + row;
      *col0 = (unsigned short)s0;
      *col1 = (unsigned short)s1;
    }
  }
  wave_lds_fence();
  {
    const int x = lane & 7, y = lane >> 3, odd = y & 1;
    const u32 *col = s_hor + x * G::HP + (y >> 1);
    int acc = 0;
#pragma unroll
    for (int t = 0; t <= 4; ++t) {
      // even y: taps (2t, 2t + 1) on row pair t; odd y: the window starts half a pair later, taps (2t - 1, 2t)
      const u32 ce = t < 4 ? frac_pack16(vf[2 * t], vf[2 * t + 1]) : 0u;
      const u32 co = frac_pack16(t > 0 ? vf[2 * t - 1] : 0, t < 4 ? vf[2 * t] : 0);
      acc = __builtin_amdgcn_sdot2(as_v2s(col[t]), as_v2s(odd ? co : ce), acc, false);
    }
    acc >>= 6;
    if (OUT14) ((i16 *)dst)[o + (size_t)lane] = (i16)acc;
    else ((u8 *)dst)[o + (size_t)lane] = fast_clip32((acc + 32) >> 6);
  }
}

// The workgroup-per-descriptor kernels share their descriptor list with the wave-per-descriptor kernels of the smaller
// size classes, and on a frame of small blocks they own nothing: a launch of `count` workgroups that each fetch one
// descriptor and leave took 23 us of the 83 us of 129 600 8x8 samples.  So a workgroup owns `chunk` (<= 64) consecutive
// descriptors: lane t of wave 0 judges descriptor t, one ballot names the ones of this kernel's class, and the
// workgroup works through those.  chunk stays 1 until the list is long enough for more than "wg_chunk_min_wgs" workgroups
// (a workgroup takes its descriptors one after the other, and the ballot costs a second memory round trip before the
// first descriptor is worked on: 1 us on a 5 us workgroup).
static inline unsigned wg_chunk(size_t count)
{
  const int min_wgs = kvzhip::tuning("wg_chunk_min_wgs", 4096);      // measured: 0: 1.37 G 8x8 samples/s, 2048: 2.30, 4096: 2.27, 16384: 2.10
  if (min_wgs <= 0) return 1;
  const size_t c = count / (size_t)min_wgs;
  return (unsigned)(c < 1 ? 1 : (c > 64 ? 64 : c));
}

template <typename Mine>
__device__ __forceinline__ unsigned long long wg_chunk_mask(unsigned long long *s_mask, size_t first, unsigned chunk, size_t count, Mine mine)
{
  if (threadIdx.x < 64) {
    const size_t i = first + threadIdx.x;
    const unsigned long long m = __ballot(threadIdx.x < chunk && i < count && mine(i));
    if (threadIdx.x == 0) *s_mask = m;
  }
  __syncthreads();
  // the same value in every lane, and known to the compiler as such: the descriptors of the loop that follows are then
  // fetched with scalar loads and the block geometry stays in SGPRs (as a plain LDS read it cost sample_big 20 %)
  const unsigned long long m = *s_mask;
  return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}

// blocks wider or taller than 16: one workgroup per block
template <int TAPS, bool OUT14>
__global__ __launch_bounds__(256) void sample_big_kernel(refplane_t ref, const kvz_hip_ipol_block *__restrict__ blocks, size_t count, unsigned chunk,
                                                         const unsigned long long *__restrict__ out_offsets, void *__restrict__ dst)
{
  constexpr int MAXW = TAPS == 8 ? 64 : 32;
  __shared__ __attribute__((aligned(16))) u8 s_win[sample_geom<TAPS, MAXW>::WIN_BYTES];
  __shared__ __attribute__((aligned(16))) i16 s_hor[2 * sample_geom<TAPS, MAXW>::HOR_DWORDS];
  __shared__ unsigned long long s_mask;
  // unsupported shapes: nothing written; up to 16x16: sample_small_kernel's
  auto mine = [&](size_t i) {
    const int w = blocks[i].width, h = blocks[i].height;
    return w >= 1 && h >= 1 && w <= MAXW && h <= MAXW && (w > 16 || h > 16);
  };
  if (chunk == 1) {                                     // short lists: straight to the descriptor (one memory round trip less)
    if (mine(blockIdx.x)) sample_core<TAPS, OUT14, MAXW, 256, false>(threadIdx.x, s_win, s_hor, ref, blocks[blockIdx.x], (size_t)out_offsets[blockIdx.x], dst);
    return;
  }
  const size_t first = (size_t)blockIdx.x * chunk;
  unsigned long long mask = wg_chunk_mask(&s_mask, first, chunk, count, mine);
  while (mask) {
    const size_t i = first + (unsigned)__builtin_ctzll(mask);
    mask &= mask - 1;
    const kvz_hip_ipol_block b = blocks[i];
    sample_core<TAPS, OUT14, MAXW, 256, false>(threadIdx.x, s_win, s_hor, ref, b, (size_t)out_offsets[i], dst);
    __syncthreads();
  }
}

// blocks up to 16x16: one wave per block, four blocks per workgroup, wave-private LDS, no barrier
template <int TAPS, bool OUT14>
__global__ __launch_bounds__(256) void sample_small_kernel(refplane_t ref, const kvz_hip_ipol_block *__restrict__ blocks, size_t count,
                                                           const unsigned long long *__restrict__ out_offsets, void *__restrict__ dst, int kvz_sample8_wave)
{
  __shared__ __attribute__((aligned(16))) u8 s_win[4][sample_geom<TAPS, 16>::WIN_BYTES];
  __shared__ __attribute__((aligned(16))) i16 s_hor[4][2 * sample_geom<TAPS, 16>::HOR_DWORDS];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: descriptor loads and block geometry go scalar
  const size_t i = (size_t)blockIdx.x * 4 + wv;
  if (i >= count) return;
  const kvz_hip_ipol_block b = blocks[i];
  if (b.width < 1 || b.height < 1 || b.width > 16 || b.height > 16) return;
  if (TAPS == 8 && b.width == 8 && b.height == 8 && kvz_sample8_wave) {
    sample8x8_luma_wave<OUT14>(threadIdx.x & 63, s_win[wv], (u32 *)s_hor[wv], ref, b, (size_t)out_offsets[i], dst);
    return;
  }
  sample_core<TAPS, OUT14, 16, 64, true>(threadIdx.x & 63, s_win[wv], s_hor[wv], ref, b, (size_t)out_offsets[i], dst);
}


// blocks larger than 32x32 (and malformed descriptors, which are flagged): one workgroup per descriptor
__global__ __launch_bounds__(256) void search_frac_big_kernel(const u8 *__restrict__ pic, u32 pic_stride, refplane_t ref,
                                                              const kvz_hip_block_pair *__restrict__ pairs, size_t count, unsigned chunk,
                                                              u32 *__restrict__ costs, i32 *__restrict__ best)
{
  __shared__ __attribute__((aligned(16))) u8 lds[frac_geom<64>::TOTAL];
  __shared__ unsigned long long s_mask;
  const int tid = threadIdx.x;
  const size_t first = (size_t)blockIdx.x * chunk;
  // up to 32x32: handled by the one-wave-per-block kernels
  auto mine = [&](size_t i) {
    const int w = pairs[i].width, h = pairs[i].height;
    return !frac_shape_ok(w, h) || w > 32 || h > 32;
  };
  // short lists (chunk 1): straight to the descriptor, one memory round trip less
  unsigned long long mask = chunk == 1 ? (unsigned long long)mine(first) : wg_chunk_mask(&s_mask, first, chunk, count, mine);
  while (mask) {
    const size_t i = first + (unsigned)__builtin_ctzll(mask);
    mask &= mask - 1;
    const kvz_hip_block_pair d = pairs[i];
    if (!frac_shape_ok(d.width, d.height)) {            // unsupported shape: flag it, touch nothing else
      if (tid < 17) costs[i * 17 + tid] = 0xffffffffu;
      if (tid < 2) best[i * 2 + tid] = -1;
      continue;
    }
    search_frac_core<64, 256, false>(tid, lds, pic, pic_stride, ref, d, 4, frac_no_cost(), costs + i * 17, best + i * 2);
    __syncthreads();
  }
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
  const int lane = threadIdx.x & 63;
  // the four shapes a wave can own, with the size as a compile-time constant
  if (d.width == 8 && d.height == 8)
    search_frac_core<16, 64, true, frac_no_cost, 8, 8>(lane, lds[wv], pic, pic_stride, ref, d, 4, frac_no_cost(), costs + i * 17, best + i * 2);
  else if (d.width == 16 && d.height == 16)
    search_frac_core<16, 64, true, frac_no_cost, 16, 16>(lane, lds[wv], pic, pic_stride, ref, d, 4, frac_no_cost(), costs + i * 17, best + i * 2);
  else
    search_frac_core<16, 64, true, frac_no_cost>(lane, lds[wv], pic, pic_stride, ref, d, 4, frac_no_cost(), costs + i * 17, best + i * 2);
}

// blocks up to 32x32 that are not the small kernel's: one wave per descriptor too (two per workgroup)
__global__ __launch_bounds__(128) void search_frac_medium_kernel(const u8 *__restrict__ pic, u32 pic_stride, refplane_t ref,
                                                                 const kvz_hip_block_pair *__restrict__ pairs, size_t count,
                                                                 u32 *__restrict__ costs, i32 *__restrict__ best)
{
  __shared__ __attribute__((aligned(16))) u8 lds[2][(frac_geom<32>::TOTAL + 15) & ~15];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t i = (size_t)blockIdx.x * 2 + wv;
  if (i >= count) return;
  const kvz_hip_block_pair d = pairs[i];
  if (!frac_shape_ok(d.width, d.height) || d.width > 32 || d.height > 32 || (d.width <= 16 && d.height <= 16)) return;
  const int lane = threadIdx.x & 63;
  if (d.width == 32 && d.height == 32)
    search_frac_core<32, 64, true, frac_no_cost, 32, 32>(lane, lds[wv], pic, pic_stride, ref, d, 4, frac_no_cost(), costs + i * 17, best + i * 2);
  else
    search_frac_core<32, 64, true, frac_no_cost>(lane, lds[wv], pic, pic_stride, ref, d, 4, frac_no_cost(), costs + i * 17, best + i * 2);
}

// ---------------------------------------------------------------------------
// Bi-prediction candidate cost (search_pu_inter_bipred, search_inter.c:1304-1440): the luma of
// kvz_inter_recon_bipred (inter.c:430-477) -- per reference the 14-bit quarter-pel sample when its vector is
// fractional (inter.c:86-122), else the edge-clamped pixels << 6 (inter.c:277-298, :355-371) -- blended like
// inter_recon_bipred_generic (picture-generic.c:538-588) and scored with satd_any_size against the source block.
// One workgroup per candidate; both predictors, the blend and the source block stay in LDS.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bipred_cost_kernel(const u8 *__restrict__ pic, u32 pic_stride, int pic_w, int pic_h,
                                                         refplane_t ref0, refplane_t ref1, const kvz_hip_bipred_cand *__restrict__ cands,
                                                         u32 *__restrict__ costs)
{
  __shared__ __attribute__((aligned(16))) u8 s_win[sample_geom<8, 64>::WIN_BYTES];
  __shared__ __attribute__((aligned(16))) i16 s_hor[2 * sample_geom<8, 64>::HOR_DWORDS];
  __shared__ __attribute__((aligned(16))) i16 s_s[2][64 * 64];
  __shared__ __attribute__((aligned(16))) u8 s_pred[64 * 64], s_cur[64 * 64];
  __shared__ u32 s_cost;
  const kvz_hip_bipred_cand &c = cands[blockIdx.x];
  const int tid = threadIdx.x, w = c.width, h = c.height;
  if (!frac_shape_ok(w, h) || c.x < 0 || c.y < 0 || c.x + w > pic_w || c.y + h > pic_h) {
    if (tid == 0) costs[blockIdx.x] = 0xffffffffu;
    return;
  }
  if (tid == 0) s_cost = 0;
  for (int k = 0; k < 2; ++k) {
    const refplane_t &ref = k ? ref1 : ref0;
    const int mvx = k ? c.mv1[0] : c.mv0[0], mvy = k ? c.mv1[1] : c.mv0[1];
    const int ix = c.x + (mvx >> 2), iy = c.y + (mvy >> 2);
    if ((mvx & 3) || (mvy & 3)) {
      const kvz_hip_ipol_block b = { ix, iy, mvx & 3, mvy & 3, w, h };
      sample_core<8, true, 64, 256, false>(tid, s_win, s_hor, ref, b, 0, s_s[k]);
    } else {
      for (int i = tid; i < w * h; i += 256) {
        const int y = i / w, x = i - y * w;
        s_s[k][i] = (i16)((int)ref_px(ref, ix + x, iy + y) << 6);
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < w * h; i += 256) {
    const int y = i / w, x = i - y * w;
    s_pred[i] = fast_clip32(((int)s_s[0][i] + (int)s_s[1][i] + 64) >> 7);
    s_cur[i] = pic[(size_t)(c.y + y) * pic_stride + c.x + x];
  }
  __syncthreads();
  {
    // satd_any_size (strategies-picture.h:62-100): 4x4 blocks on the first 4-pixel column / row of an SMP / AMP shape,
    // the 8x8 grid behind them
    const int ox = w & 4, oy = h & 4;
    const int w8 = (w - ox) >> 3, n8 = w8 * ((h - oy) >> 3), p = tid & 3;
    const short sg1 = (p & 1) ? (short)-1 : (short)1, sg2 = (p & 2) ? (short)-1 : (short)1;
    const v2s m1 = { sg1, sg1 }, m2 = { sg2, sg2 };
    u32 acc = 0;
    if (ox | oy) {
      const int nb = ox ? h >> 2 : w >> 2;
      for (int i = tid; i < nb; i += 256) {
        const int bx4 = ox ? 0 : 4 * i, by4 = ox ? 4 * i : 0;
        u32 ra[4], rb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __builtin_memcpy(&ra[r], s_pred + (by4 + r) * w + bx4, 4);
          __builtin_memcpy(&rb[r], s_cur + (by4 + r) * w + bx4, 4);
        }
        acc += satd4x4_regs(ra, rb);
      }
    }
    for (int i = tid; i < n8 * 4; i += 256) {
      const int sb = i >> 2, by = sb / w8, bx = sb - by * w8;
      const u8 *a = s_pred + (oy + by * 8 + 2 * p) * w + ox + bx * 8, *b = s_cur + (oy + by * 8 + 2 * p) * w + ox + bx * 8;
      uint4 x, y;                                        // rows are 4-byte aligned (w is a multiple of 4)
      __builtin_memcpy(&x.x, a, 4); __builtin_memcpy(&x.y, a + 4, 4); __builtin_memcpy(&x.z, a + w, 4); __builtin_memcpy(&x.w, a + w + 4, 4);
      __builtin_memcpy(&y.x, b, 4); __builtin_memcpy(&y.y, b + 4, 4); __builtin_memcpy(&y.z, b + w, 4); __builtin_memcpy(&y.w, b + w + 4, 4);
      u32 m = satd8_quad_part(x, y, m1, m2);
      m = group_sum<4>(m);
      if (p == 0) acc += (m + 2) >> 2;
    }
    acc = group_sum<64>(acc);
    if ((tid & 63) == 0 && acc) atomicAdd(&s_cost, acc);
  }
  __syncthreads();
  if (tid == 0) costs[blockIdx.x] = s_cost;
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

// kvz_get_extended_block's copy (ipol-generic.c:759-783): out[dy][dx] = rect[clip(dy - oy)][clip(dx - ox)], where rect is
// the part of the reference plane the window overlaps (rw x rh, contiguous) and (ox, oy) its position inside the window
__global__ __launch_bounds__(256) void extend_block_kernel(const u8 *__restrict__ rect, int rw, int rh, int ox, int oy,
                                                           u8 *__restrict__ out, int ow, int oh)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ow * oh; i += gridDim.x * blockDim.x) {
    const int dy = i / ow, dx = i - dy * ow;
    out[i] = rect[clampi(dy - oy, 0, rh - 1) * rw + clampi(dx - ox, 0, rw - 1)];
  }
}

namespace kvzhip {
int launch_extend_block(const u8 *rect, int rw, int rh, int ox, int oy, u8 *out, int ow, int oh, hipStream_t st)
{
  hipLaunchKernelGGL(extend_block_kernel, dim3((unsigned)((ow * oh + 255) / 256)), dim3(256), 0, st, rect, rw, rh, ox, oy, out, ow, oh);
  KVZ_CHECK_LAUNCH("extend_block_kernel");
  return KVZ_HIP_OK;
}
int launch_frac_step(const u8 *win, int w, int h, int step, int fme_level, int hx, int hy,
                     u8 *filtered, i16 *hor_out, i16 *cols_out, hipStream_t st)
{
  if (w < 4 || h < 4 || w > 64 || h > 64 || ((w | h) & 3) || step < 0 || step > 3) return kvzhip::invalid_arg(__func__);   // any PU shape incl. SMP / AMP
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
  if (!ref || !blocks || !out_offsets || !dst || ref_w <= 0 || ref_h <= 0) return kvzhip::invalid_arg("kvz_hip_sample_luma_batch / kvz_hip_sample_chroma_batch");
  if (count == 0) return KVZ_HIP_OK;
  if (count > 0x7fffffffu) return kvzhip::invalid_arg("kvz_hip_sample_luma_batch / kvz_hip_sample_chroma_batch");
  refplane_t r = { ref, ref_stride, ref_w, ref_h };
  hipStream_t st = ctx_stream(s);
  const unsigned long long *oo = (const unsigned long long *)out_offsets;
  const unsigned chunk = wg_chunk(count);
  const int s8w = kvzhip::tuning("sample8_wave", 1);          // 0: 8x8 luma blocks on the general 2 x 4-sample path (A/B)
  const unsigned gs = (unsigned)((count + 3) / 4), gb = (unsigned)((count + chunk - 1) / chunk);
#define KVZ_SAMPLE(TAPS, O14)                                                                                        \
  do {                                                                                                               \
    hipLaunchKernelGGL((sample_small_kernel<TAPS, O14>), dim3(gs), dim3(256), 0, st, r, blocks, count, oo, dst, s8w); \
    hipLaunchKernelGGL((sample_big_kernel<TAPS, O14>), dim3(gb), dim3(256), 0, st, r, blocks, count, chunk, oo, dst); \
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
  if (!pic || !ref || !pairs || !costs || !best || ref_w <= 0 || ref_h <= 0) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  if (count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  refplane_t r = { ref, ref_stride, ref_w, ref_h };
  // three passes over the same descriptor list: each kernel takes the size class it is built for and skips the rest
  hipLaunchKernelGGL(search_frac_small_kernel, dim3((unsigned)((count + 3) / 4)), dim3(256), 0, ctx_stream(s), pic, pic_stride, r, pairs, count, costs, best);
  KVZ_CHECK_LAUNCH("search_frac_small_kernel");
  const unsigned chunk = wg_chunk(count);
  hipLaunchKernelGGL(search_frac_big_kernel, dim3((unsigned)((count + chunk - 1) / chunk)), dim3(256), 0, ctx_stream(s), pic, pic_stride, r, pairs, count, chunk, costs, best);
  KVZ_CHECK_LAUNCH("search_frac_big_kernel");
  hipLaunchKernelGGL(search_frac_medium_kernel, dim3((unsigned)((count + 1) / 2)), dim3(128), 0, ctx_stream(s), pic, pic_stride, r, pairs, count, costs, best);
  KVZ_CHECK_LAUNCH("search_frac_medium_kernel");
  return KVZ_HIP_OK;
}

int kvz_hip_bipred_cost_batch(const kvz_hip_pixel *pic, uint32_t pic_stride, int pic_w, int pic_h,
                              const kvz_hip_pixel *ref0, uint32_t ref0_stride, const kvz_hip_pixel *ref1, uint32_t ref1_stride,
                              int ref_w, int ref_h, const kvz_hip_bipred_cand *cands, size_t count, uint32_t *costs, kvz_hip_stream s)
{
  KVZ_CHECK_CTX();
  if (!pic || !ref0 || !ref1 || !cands || !costs || pic_w <= 0 || pic_h <= 0 || ref_w <= 0 || ref_h <= 0) return kvzhip::invalid_arg(__func__);
  if (count == 0) return KVZ_HIP_OK;
  if (count > 0x7fffffffu) return kvzhip::invalid_arg(__func__);
  const refplane_t r0 = { ref0, ref0_stride, ref_w, ref_h }, r1 = { ref1, ref1_stride, ref_w, ref_h };
  hipLaunchKernelGGL(bipred_cost_kernel, dim3((unsigned)count), dim3(256), 0, ctx_stream(s), pic, pic_stride, pic_w, pic_h, r0, r1, cands, costs);
  KVZ_CHECK_LAUNCH("bipred_cost_kernel");
  return KVZ_HIP_OK;
}

}  // extern "C"
