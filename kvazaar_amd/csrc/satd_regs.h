// satd_regs.h -- lane-private Hadamard cost of one 8x8 / 4x4 block held in registers.
// Packed int16 arithmetic (|coef| <= 64*255 fits); the last butterfly stage is folded
// into the absolute sum: |a+b| + |a-b| = 2*max(|a|,|b|).
// picture-generic.c:240-328 (8x8: (sum+2)>>2), :105-196 (4x4: (sum+1)>>1).
#pragma once
#include "kvz_hip_internal.h"

namespace kvzhip {

typedef short v2s __attribute__((ext_vector_type(2)));

__device__ __forceinline__ v2s as_v2s(u32 x) { return __builtin_bit_cast(v2s, x); }
__device__ __forceinline__ u32 as_u32(v2s x) { return __builtin_bit_cast(u32, x); }
__device__ __forceinline__ v2s unpack_lo(u32 d) { return as_v2s(__builtin_amdgcn_perm(0u, d, 0x0c010c00u)); }
__device__ __forceinline__ v2s unpack_hi(u32 d) { return as_v2s(__builtin_amdgcn_perm(0u, d, 0x0c030c02u)); }

// sum over the register's two halves of max(|lo|,|hi|)  (the last stage folded: |a+b| + |a-b| = 2 max(|a|,|b|))
__device__ __forceinline__ u32 absmax_halves(v2s x)
{
  v2s n = -x;
  v2s ax = __builtin_elementwise_max(x, n);
  u32 w = as_u32(ax);
  u32 lo = w & 0xffffu, hi = w >> 16;
  return lo > hi ? lo : hi;
}

// Last butterfly stage + absolute sum of one register in 4 instructions: v_pk_mad_u16 with op_sel builds
// (lo + hi, lo - hi), v_pk_sub / v_pk_max take the absolute values, v_sad_u16 against 0 adds both halves
// to the accumulator.  acc += |lo + hi| + |lo - hi|.
__device__ __forceinline__ u32 abs_last_stage(v2s x, u32 acc)
{
  const v2s hh = { x.y, x.y }, ll = { x.x, x.x }, pm = { 1, -1 };
  const v2s t = hh * pm + ll;
  const v2s n = -t;
  return __builtin_amdgcn_sad_u16(as_u32(__builtin_elementwise_max(t, n)), 0u, acc);
}

// x[r][q]: difference row r, columns 2q (low half) and 2q+1 (high half).  Destroys x.
// Returns the reference's satd_8x8_subblock value.
__device__ __forceinline__ u32 satd8x8_diff(v2s (&x)[8][4])
{
  // horizontal, column bit 2 (distance 4) and bit 1 (distance 2)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    v2s s0 = x[r][0] + x[r][2], s1 = x[r][1] + x[r][3];
    v2s d0 = x[r][0] - x[r][2], d1 = x[r][1] - x[r][3];
    x[r][0] = s0 + s1; x[r][1] = s0 - s1;
    x[r][2] = d0 + d1; x[r][3] = d0 - d1;
  }
  // vertical, three stages
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    v2s t[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { t[r] = x[r][q] + x[r + 4][q]; t[r + 4] = x[r][q] - x[r + 4][q]; }
#pragma unroll
    for (int h = 0; h < 8; h += 4) {
      v2s u0 = t[h] + t[h + 2], u1 = t[h + 1] + t[h + 3], u2 = t[h] - t[h + 2], u3 = t[h + 1] - t[h + 3];
      x[h][q] = u0 + u1; x[h + 1][q] = u0 - u1; x[h + 2][q] = u2 + u3; x[h + 3][q] = u2 - u3;
    }
  }
  // column bit 0 (inside the register) and the absolute sum
  u32 sum = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) sum = abs_last_stage(x[r][q], sum);
  return (sum + 2) >> 2;
}

// a[16], b[16]: row r of the 8x8 = dwords 2r (cols 0..3) and 2r+1 (cols 4..7) of packed bytes.
__device__ __forceinline__ u32 satd8x8_regs(const u32 *a, const u32 *b)
{
  v2s x[8][4];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    x[r][0] = unpack_lo(a[2 * r]) - unpack_lo(b[2 * r]);
    x[r][1] = unpack_hi(a[2 * r]) - unpack_hi(b[2 * r]);
    x[r][2] = unpack_lo(a[2 * r + 1]) - unpack_lo(b[2 * r + 1]);
    x[r][3] = unpack_hi(a[2 * r + 1]) - unpack_hi(b[2 * r + 1]);
  }
  return satd8x8_diff(x);
}

// x[r][q]: difference row r of a 4x4, columns 2q, 2q+1.  Returns satd_4x4.
__device__ __forceinline__ u32 satd4x4_diff(v2s (&x)[4][2])
{
#pragma unroll
  for (int r = 0; r < 4; ++r) { v2s s = x[r][0] + x[r][1], d = x[r][0] - x[r][1]; x[r][0] = s; x[r][1] = d; }
  u32 sum = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    v2s s0 = x[0][q] + x[2][q], s1 = x[1][q] + x[3][q], d0 = x[0][q] - x[2][q], d1 = x[1][q] - x[3][q];
    sum = abs_last_stage(s0 + s1, sum);
    sum = abs_last_stage(s0 - s1, sum);
    sum = abs_last_stage(d0 + d1, sum);
    sum = abs_last_stage(d0 - d1, sum);
  }
  return (sum + 1) >> 1;
}

// a[4], b[4]: row r of the 4x4 = dword r of packed bytes.
__device__ __forceinline__ u32 satd4x4_regs(const u32 *a, const u32 *b)
{
  v2s x[4][2];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    x[r][0] = unpack_lo(a[r]) - unpack_lo(b[r]);
    x[r][1] = unpack_hi(a[r]) - unpack_hi(b[r]);
  }
  return satd4x4_diff(x);
}

// ---- an 8x8 block spread over a quad of lanes (lane p = rows 2p, 2p+1) ----
template <int CTRL>
__device__ __forceinline__ v2s dpp_v2s(v2s v)
{
  return as_v2s((u32)__builtin_amdgcn_update_dpp(0, (int)as_u32(v), CTRL, 0xF, 0xF, true));
}

// lane ^ 4 within a row of 16 lanes: a bank-masked row_shl:4 / row_shr:4 pair
__device__ __forceinline__ v2s dpp_xor4_v2s(v2s v)
{
  int t = __builtin_amdgcn_update_dpp(0, (int)as_u32(v), 0x104, 0xF, 0x5, false);      // row_shl:4 into banks 0, 2
  t = __builtin_amdgcn_update_dpp(t, (int)as_u32(v), 0x114, 0xF, 0xA, false);          // row_shr:4 into banks 1, 3
  return as_v2s((u32)t);
}
__device__ __forceinline__ u32 dpp_xor4_u32(u32 v)
{
  int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);
  t = __builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
  return (u32)t;
}

// d[j][q]: the lane's differences, row 2p+j, columns (2q, 2q+1) packed.  Returns the lane's share of the
// block's absolute Hadamard sum; the block's SATD is (sum over the quad + 2) >> 2.
__device__ __forceinline__ u32 satd8_quad_part_diff(v2s (&d)[2][4], v2s m1, v2s m2)
{
#pragma unroll
  for (int j = 0; j < 2; ++j) {          // column bits 2 and 1
    v2s s0 = d[j][0] + d[j][2], s1 = d[j][1] + d[j][3], e0 = d[j][0] - d[j][2], e1 = d[j][1] - d[j][3];
    d[j][0] = s0 + s1; d[j][1] = s0 - s1; d[j][2] = e0 + e1; d[j][3] = e0 - e1;
  }
  v2s w[8];
#pragma unroll
  for (int q = 0; q < 4; ++q) { w[q] = d[0][q] + d[1][q]; w[q + 4] = d[0][q] - d[1][q]; }   // row bit 0
  u32 m = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v2s t = dpp_v2s<0xB1>(w[i]);         // quad_perm [1,0,3,2]: row bit 1
    v2s u = w[i] * m1 + t;
    t = dpp_v2s<0x4E>(u);                // quad_perm [2,3,0,1]: row bit 2
    u = u * m2 + t;
    m = abs_last_stage(u, m);            // column bit 0 and the absolute sum
  }
  return m;
}

// x, y: the lane's 16-byte chunk of each block (rows 2p and 2p+1).
__device__ __forceinline__ u32 satd8_quad_part(uint4 x, uint4 y, v2s m1, v2s m2)
{
  v2s d[2][4];
  d[0][0] = unpack_lo(x.x) - unpack_lo(y.x); d[0][1] = unpack_hi(x.x) - unpack_hi(y.x);
  d[0][2] = unpack_lo(x.y) - unpack_lo(y.y); d[0][3] = unpack_hi(x.y) - unpack_hi(y.y);
  d[1][0] = unpack_lo(x.z) - unpack_lo(y.z); d[1][1] = unpack_hi(x.z) - unpack_hi(y.z);
  d[1][2] = unpack_lo(x.w) - unpack_lo(y.w); d[1][3] = unpack_hi(x.w) - unpack_hi(y.w);
  return satd8_quad_part_diff(d, m1, m2);
}

}  // namespace kvzhip
