// dct16_mfma_core.h -- 16x16 forward / inverse integer DCT of TWO blocks per v_mfma_i32_32x32x32_i8,
// shared by dct16_mfma.hip (transform_batch) and quant16_mfma.hip (fused quantize_residual).
//
// Reference: src/strategies/generic/dct-generic.c:368-455, :567-597 (N = 16).  Method as in
// dct32_mfma_core.h: byte planes X = 256*Xh + Xl' + 128, exact int32 partial sums, the accumulator
// tile re-used as the next operand.  Two blocks a, b are stacked into the 32 rows of the tile:
//   pass 1   D1 = [S_a; S_b] * M16^T             (K: 16 live of 32, dead K operands are 0)
//   pass 2   D2[k][x] = sum_j T'[j][k] * B2[j][x], B2 block diagonal in (block of j, block of x)
// "Natural" layout of a pair: lane (r, h), r = lane & 31, h = lane >> 5, holds columns 8h .. 8h+7 of
// row r & 15 of block r >> 4 -- one 16-byte chunk (index 2r + h) of the pair's 1 KiB of int16.
// "Accumulator" layout of a result: the same lane holds columns 4h .. 4h+3 (o[0..3]) and
// 8+4h .. 8+4h+3 (o[4..7]) of that row; acc16_to_chunk() converts with v_permlane32_swap.
#pragma once
#include "dct32_mfma_core.h"

namespace kvzhip {

typedef unsigned int u32x2w __attribute__((ext_vector_type(2)));

struct m16_table {
  signed char v[16 * 16];
  constexpr m16_table() : v()
  {
    for (int k = 0; k < 16; ++k)
      for (int n = 0; n < 16; ++n) v[k * 16 + n] = (signed char)dct_coef(16, k, n);
  }
};
static __constant__ m16_table c_m16 = m16_table();

// column of accumulator register g on lane half h
__device__ __forceinline__ int acc16_col(int h, int g) { return g < 4 ? 4 * h + g : 8 + 4 * h + (g - 4); }

// byte planes of the lane's 8 live int16 (4 dwords); elements 8..15 are dead K (zero in both planes)
__device__ __forceinline__ void planes8(const u32x4v &c, op16 &hi, op16 &lo)
{
  lo.w[0] = __builtin_amdgcn_perm(c.y, c.x, 0x06040200u) ^ 0x80808080u;
  lo.w[1] = __builtin_amdgcn_perm(c.w, c.z, 0x06040200u) ^ 0x80808080u;
  hi.w[0] = __builtin_amdgcn_perm(c.y, c.x, 0x07050301u);
  hi.w[1] = __builtin_amdgcn_perm(c.w, c.z, 0x07050301u);
  lo.w[2] = lo.w[3] = hi.w[2] = hi.w[3] = 0u;
}

// per-lane constant operands of one direction
struct dct16_lane {
  op16 tA, tB, tC;   // forward: tA = pass-1 B (M16[k][8h+e], k < 16), tB = pass-2 B (block diagonal M16[x&15][j&15])
                     // inverse: tA = identity (natural K), tB = block diagonal M16[k2&15][j'&15] (pass 1),
                     //          tC = pass-2 A: M16[k][i'] for k, i' < 16 (K = kappa order, k >= 16 dead)
  int sum;           // forward: row sum of M16 row (r & 15); inverse: column sum of column (r & 15)
};

// The records above for all 64 lanes (r = lane & 31, h = lane >> 5) and both directions, built at compile time: a wave
// fetches its operands with vector loads instead of ~50 byte loads and a 16-step sum per lane.
struct dct16_lane_consts { u32 tA[4], tB[4], tC[4]; int sum, pad[3]; };
template <bool INVERSE>
struct dct16_lane_table {
  dct16_lane_consts l[64];
  constexpr dct16_lane_table() : l()
  {
    for (int lane = 0; lane < 64; ++lane) {
      const int r = lane & 31, h = lane >> 5;
      int sum = 0;
      for (int n = 0; n < 16; ++n) sum += INVERSE ? dct_coef(16, n, r & 15) : dct_coef(16, r & 15, n);
      l[lane].sum = sum; l[lane].pad[0] = l[lane].pad[1] = l[lane].pad[2] = 0;
      for (int q = 0; q < 4; ++q) {
        u32 a = 0, b = 0, c = 0;
        for (int j = 0; j < 4; ++j) {
          const int e = 4 * q + j, kk = (e & 3) + 8 * (e >> 2) + 4 * h;
          int va = 0, vb = 0, vc = 0;
          if (!INVERSE) {
            va = (e < 8 && r < 16) ? dct_coef(16, r, 8 * h + e) : 0;
            vb = ((kk >> 4) == (r >> 4)) ? dct_coef(16, r & 15, kk & 15) : 0;
          } else {
            va = (e < 8 && r < 16 && 8 * h + e == r) ? 1 : 0;
            vb = ((kk >> 4) == (r >> 4)) ? dct_coef(16, kk & 15, r & 15) : 0;
            vc = (kk < 16 && r < 16) ? dct_coef(16, kk, r) : 0;
          }
          a |= ((u32)va & 255u) << (8 * j); b |= ((u32)vb & 255u) << (8 * j); c |= ((u32)vc & 255u) << (8 * j);
        }
        l[lane].tA[q] = a; l[lane].tB[q] = b; l[lane].tC[q] = c;
      }
    }
  }
};
static __constant__ dct16_lane_table<false> c_dct16_fwd = dct16_lane_table<false>();
static __constant__ dct16_lane_table<true> c_dct16_inv = dct16_lane_table<true>();

template <bool INVERSE>
__device__ __forceinline__ void dct16_setup(int r, int h, dct16_lane &k)
{
  const dct16_lane_consts &lc = INVERSE ? c_dct16_inv.l[r + 32 * h] : c_dct16_fwd.l[r + 32 * h];
#pragma unroll
  for (int q = 0; q < 4; ++q) { k.tA.w[q] = lc.tA[q]; k.tB.w[q] = lc.tB[q]; k.tC.w[q] = lc.tC[q]; }
  k.sum = lc.sum;
}

// inverse pass 2 constants: 128 * (column sum of M16)[kappa(h,g)] + 2048 for the 8 live registers of each lane half.
// Call from every thread of the workgroup, then __syncthreads().
struct dct16_c2_table {
  int v[16];
  constexpr dct16_c2_table() : v()
  {
    for (int i = 0; i < 16; ++i) {
      const int hh = i >> 3, g = i & 7, row = (g & 3) + 8 * (g >> 2) + 4 * hh;
      int cs = 0;
      for (int n = 0; n < 16; ++n) cs += dct_coef(16, n, row);
      v[i] = 128 * cs + (1 << 11);
    }
  }
};
static __constant__ dct16_c2_table c_dct16_c2 = dct16_c2_table();
__device__ __forceinline__ void dct16_fill_c2(int (*s_c2)[8])
{
  if (threadIdx.x < 16) s_c2[threadIdx.x >> 3][threadIdx.x & 7] = c_dct16_c2.v[threadIdx.x];
}

// forward DCT of the pair: cur = natural chunk of residuals, o = coefficients in accumulator layout
__device__ __forceinline__ void dct16_fwd_pair(const u32x4v &cur, const dct16_lane &k, int (&o)[8])
{
  const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  op16 hi, lo;
  planes8(cur, hi, lo);
  // pass 1: D1[j][k] = sum_n S[j][n] M16[k][n]; rows j (both blocks) in registers, column k = lane (k < 16 live)
  const i32x16 ah = mfma_i8(hi, k.tA, zero), al = mfma_i8(lo, k.tA, zero);
  const int c1 = 128 * k.sum + (1 << 2);
  int tt[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) tt[g] = ((ah[g] << 8) + al[g] + c1) >> 3;
  // pass 2: D2[k][x] = sum_j T'[j][k] * (same block ? M16[x&15][j&15] : 0) = out_{x>>4}[x&15][k]
  op16 h2, l2;
  planes_from_regs(tt, h2, l2, 0x80808080u);
  const i32x16 bh = mfma_i8(h2, k.tB, zero), bl = mfma_i8(l2, k.tB, zero);
  const int c2 = 128 * k.sum + (1 << 9);
#pragma unroll
  for (int g = 0; g < 8; ++g) o[g] = ((bh[g] << 8) + bl[g] + c2) >> 10;
}

// inverse DCT of the pair: cur = natural chunk of coefficients, c2 = s_c2[h], o = residuals (clipped) in accumulator layout
__device__ __forceinline__ void dct16_inv_pair(const u32x4v &cur, const dct16_lane &k, const int *c2, int (&o)[8])
{
  const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  op16 hi, lo;
  planes8(cur, hi, lo);
  // transpose both blocks through the matrix core: column c (< 16) of block a / b on lane c, rows kappa < 16 / >= 16
  const i32x16 xh = mfma_i8(hi, k.tA, zero), xl = mfma_i8(lo, k.tA, zero);
  int th[16], tl[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) { th[g] = xh[g]; tl[g] = xl[g]; }
  op16 ph, pl, dummy;
  planes_from_regs(th, dummy, ph, 0u);
  planes_from_regs(tl, dummy, pl, 0u);
  // pass 1: D[k][j'] = sum_k2 in[k2][k] * (same block ? M16[k2&15][j'&15] : 0) = tmp_{j'>>4}[k][j'&15]
  const i32x16 ah = mfma_i8(ph, k.tB, zero), al = mfma_i8(pl, k.tB, zero);
  const int c1 = 128 * k.sum + (1 << 6);
  int uu[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) uu[g] = clip16(((ah[g] << 8) + al[g] + c1) >> 7);
  // pass 2: D2[i'][j'] = sum_{k<16} M16[k][i'] * U[j'][k]; registers with kappa >= 16 are dead K (tC is 0 there)
  op16 h2, l2;
  planes_from_regs(uu, h2, l2, 0x80808080u);
  const i32x16 bh = mfma_i8(k.tC, h2, zero), bl = mfma_i8(k.tC, l2, zero);
#pragma unroll
  for (int g = 0; g < 8; ++g) o[g] = clip16(((bh[g] << 8) + bl[g] + c2[g]) >> 12);
}

// accumulator layout -> natural chunk (8 consecutive int16, low 16 bits of each o[g])
__device__ __forceinline__ u32x4v acc16_to_chunk(const int (&o)[8])
{
  const u32 p0x = __builtin_amdgcn_perm((u32)o[1], (u32)o[0], 0x05040100u), p0y = __builtin_amdgcn_perm((u32)o[3], (u32)o[2], 0x05040100u);
  const u32 p1x = __builtin_amdgcn_perm((u32)o[5], (u32)o[4], 0x05040100u), p1y = __builtin_amdgcn_perm((u32)o[7], (u32)o[6], 0x05040100u);
  const u32x2w sx = __builtin_amdgcn_permlane32_swap(p0x, p1x, false, false);
  const u32x2w sy = __builtin_amdgcn_permlane32_swap(p0y, p1y, false, false);
  const u32x4v ov = { sx.x, sy.x, sx.y, sy.y };
  return ov;
}

}  // namespace kvzhip
