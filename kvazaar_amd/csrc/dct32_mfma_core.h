// dct32_mfma_core.h -- building blocks of the 32x32 integer DCT on the CDNA4 matrix
// cores, shared by dct32_mfma.hip (standalone transforms) and quant_tile_mfma.hip (fused
// quantize_residual).  See dct32_mfma.hip for the method.
#pragma once

#include "kvz_hip_internal.h"
#include "transform_core.h"

namespace kvzhip {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

struct m32_table {
  signed char v[32 * 32];
  constexpr m32_table() : v()
  {
    for (int k = 0; k < 32; ++k)
      for (int n = 0; n < 32; ++n) v[k * 32 + n] = (signed char)dct_coef(32, k, n);
  }
};
static __constant__ m32_table c_m32 = m32_table();

// accumulator row of register g in lane half h: rows (g&3) + 8*(g>>2) + 4h
__device__ __forceinline__ int kappa(int h, int e) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

union op16 { i32x4 v; signed char b[16]; u32 w[4]; };

// The constant MFMA operands of a lane (r = lane & 31, h = lane >> 5) and the plane-offset sums, built at compile time:
// a wave fetches them with five vector loads instead of ~100 byte loads and two 32-step sums -- the kernels are launched
// as many short-lived workgroups, so the per-wave set-up is paid once per two or three blocks.
struct dct32_lane_consts {
  u32 t_nat[4];      // M[r][16h + e]
  u32 t_kap[4];      // M[r][kappa(h, e)]
  u32 t_col[4];      // M[kappa(h, e)][r]
  u32 t_id[4];       // identity, natural K order
  u32 t_idk[4];      // identity in kappa K order
  int rowsum, colsum, pad0, pad1;
};
struct dct32_lane_table {
  dct32_lane_consts l[64];
  constexpr dct32_lane_table() : l()
  {
    for (int lane = 0; lane < 64; ++lane) {
      const int r = lane & 31, h = lane >> 5;
      int rs = 0, cs = 0;
      for (int n = 0; n < 32; ++n) { rs += dct_coef(32, r, n); cs += dct_coef(32, n, r); }
      l[lane].rowsum = rs; l[lane].colsum = cs; l[lane].pad0 = 0; l[lane].pad1 = 0;
      for (int q = 0; q < 4; ++q) {
        u32 a = 0, b = 0, c = 0, d = 0, dk = 0;
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * q + k, kap = (e & 3) + 8 * (e >> 2) + 4 * h;
          a |= ((u32)dct_coef(32, r, 16 * h + e) & 255u) << (8 * k);
          b |= ((u32)dct_coef(32, r, kap) & 255u) << (8 * k);
          c |= ((u32)dct_coef(32, kap, r) & 255u) << (8 * k);
          d |= (u32)(16 * h + e == r ? 1 : 0) << (8 * k);
          dk |= (u32)(kap == r ? 1 : 0) << (8 * k);
        }
        l[lane].t_nat[q] = a; l[lane].t_kap[q] = b; l[lane].t_col[q] = c; l[lane].t_id[q] = d; l[lane].t_idk[q] = dk;
      }
    }
  }
};
static __constant__ dct32_lane_table c_dct32_lanes = dct32_lane_table();

// The same records for a 32x32 tile whose coefficient matrix is block diagonal, diag(M_N, ..., M_N): (32 / N)^2 N x N blocks
// arranged as a grid transform independently with the instruction stream of one 32x32 block, every lane and accumulator
// register live.  N = 16 (four blocks), N = 4 (sixty-four; DST: the 4x4 DST-VII matrix instead of the DCT's).
template <int N, bool DST>
__host__ __device__ constexpr int tile_mx(int a, int b)
{
  return (a / N == b / N) ? (DST ? dst_coef(a % N, b % N) : dct_coef(N, a % N, b % N)) : 0;
}
__host__ __device__ constexpr int tile_mx16(int a, int b) { return tile_mx<16, false>(a, b); }
template <int N, bool DST>
struct tile_lane_table {
  dct32_lane_consts l[64];
  constexpr tile_lane_table() : l()
  {
    for (int lane = 0; lane < 64; ++lane) {
      const int r = lane & 31, h = lane >> 5;
      int rs = 0, cs = 0;
      for (int n = 0; n < 32; ++n) { rs += tile_mx<N, DST>(r, n); cs += tile_mx<N, DST>(n, r); }
      l[lane].rowsum = rs; l[lane].colsum = cs; l[lane].pad0 = 0; l[lane].pad1 = 0;
      for (int q = 0; q < 4; ++q) {
        u32 a = 0, b = 0, c = 0, d = 0, dk = 0;
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * q + k, kap = (e & 3) + 8 * (e >> 2) + 4 * h;
          a |= ((u32)tile_mx<N, DST>(r, 16 * h + e) & 255u) << (8 * k);
          b |= ((u32)tile_mx<N, DST>(r, kap) & 255u) << (8 * k);
          c |= ((u32)tile_mx<N, DST>(kap, r) & 255u) << (8 * k);
          d |= (u32)(16 * h + e == r ? 1 : 0) << (8 * k);
          dk |= (u32)(kap == r ? 1 : 0) << (8 * k);
        }
        l[lane].t_nat[q] = a; l[lane].t_kap[q] = b; l[lane].t_col[q] = c; l[lane].t_id[q] = d; l[lane].t_idk[q] = dk;
      }
    }
  }
};
static __constant__ tile_lane_table<16, false> c_dct16x4_lanes = tile_lane_table<16, false>();
static __constant__ tile_lane_table<4, false> c_dct4x64_lanes = tile_lane_table<4, false>();
static __constant__ tile_lane_table<4, true> c_dst4x64_lanes = tile_lane_table<4, true>();

// byte planes of 16 int16 held as 8 dwords (element pairs): hi = X >> 8, lo' = (X & 255) - 128
__device__ __forceinline__ void planes_from_rows(const u32 (&d)[8], op16 &hi, op16 &lo)
{
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    lo.w[q] = __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x06040200u) ^ 0x80808080u;
    hi.w[q] = __builtin_amdgcn_perm(d[2 * q + 1], d[2 * q], 0x07050301u);
  }
}
// byte planes of 16 values held one per register (low 16 bits significant)
__device__ __forceinline__ void planes_from_regs(const int (&t)[16], op16 &hi, op16 &lo, u32 lo_xor)
{
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32 p01 = __builtin_amdgcn_perm((u32)t[4 * q + 1], (u32)t[4 * q], 0x05010400u);       // l0 l1 h0 h1
    const u32 p23 = __builtin_amdgcn_perm((u32)t[4 * q + 3], (u32)t[4 * q + 2], 0x05010400u);   // l2 l3 h2 h3
    lo.w[q] = __builtin_amdgcn_perm(p23, p01, 0x05040100u) ^ lo_xor;
    hi.w[q] = __builtin_amdgcn_perm(p23, p01, 0x07060302u);
  }
}

__device__ __forceinline__ i32x16 mfma_i8(const op16 &a, const op16 &b, i32x16 c)
{
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(a.v, b.v, c, 0, 0, 0);
}

// 16-byte slot of logical chunk c (row j = c >> 2, quarter c & 3) inside the wave's 2 KiB LDS tile
__device__ __forceinline__ int slot_of(int c) { const int j = c >> 2; return (c & ~3) | ((c & 3) ^ ((j >> 2) & 3)); }

__device__ __forceinline__ void load_chunks(const i16 *blk, int lane, u32x4v (&c)[2])
{
  c[0] = __builtin_nontemporal_load((const u32x4v *)blk + lane);
  c[1] = __builtin_nontemporal_load((const u32x4v *)blk + 64 + lane);
}

// linear chunks (lane l holds chunks l and 64 + l) -> lane (r, h) holds row r, columns 16h .. 16h+15
__device__ __forceinline__ void chunks_to_rows(u8 *tile, int lane, int r, int h, const u32x4v (&c)[2], u32 (&d)[8])
{
  *(u32x4v *)(tile + slot_of(lane) * 16) = c[0];
  *(u32x4v *)(tile + slot_of(64 + lane) * 16) = c[1];
  wave_lds_fence();
  const u32x4v a = *(const u32x4v *)(tile + slot_of(4 * r + 2 * h) * 16);
  const u32x4v b = *(const u32x4v *)(tile + slot_of(4 * r + 2 * h + 1) * 16);
  wave_lds_fence();
  d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
}

// lane (r, h) holds out[r][kappa(h, g)], g = 0..15 (four runs of four consecutive int16)
// -> linear chunks, stored with two coalesced 16-byte stores per lane
__device__ __forceinline__ void rows_to_chunks_store(u8 *tile, int lane, int r, int h, const int (&o)[16], i16 *blk)
{
#pragma unroll
  for (int gg = 0; gg < 4; ++gg) {
    uint2 v;
    v.x = __builtin_amdgcn_perm((u32)o[4 * gg + 1], (u32)o[4 * gg], 0x05040100u);
    v.y = __builtin_amdgcn_perm((u32)o[4 * gg + 3], (u32)o[4 * gg + 2], 0x05040100u);
    *(uint2 *)(tile + slot_of(4 * r + gg) * 16 + 8 * h) = v;     // columns 8gg + 4h .. +3 of row r
  }
  wave_lds_fence();
  const u32x4v a = *(const u32x4v *)(tile + slot_of(lane) * 16);
  const u32x4v b = *(const u32x4v *)(tile + slot_of(64 + lane) * 16);
  wave_lds_fence();               // the tile is rewritten by the next block only after these reads
  __builtin_nontemporal_store(a, (u32x4v *)blk + lane);
  __builtin_nontemporal_store(b, (u32x4v *)blk + 64 + lane);
}


// forward 2-D core.  (hi, lo') = byte planes of the A operand (lane = row, 16 K elements in the order
// table `tb1` uses: tb1 element e of lane (k, h) = M[k][column of element e]).  o[g] = out[r][kappa(h, g)]
// before the (short) wrap.  dct-generic.c:458-511, :567-576: shifts 4 and 11.
template <int LOG2N = 5>
__device__ __forceinline__ void fwd32_core(const op16 &hi, const op16 &lo, const op16 &tb1, const op16 &t_kap, int rowsum, int (&o)[16])
{
  constexpr int S1 = LOG2N - 1, S2 = LOG2N + 6;          // dct-generic.c:567-576
  const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  // pass 1: T' = S * M^T  (A = S rows, B[n][k] = M[k][n]); D[j][k]: row j = kappa(h,g), col k = r
  const i32x16 ah = mfma_i8(hi, tb1, zero), al = mfma_i8(lo, tb1, zero);
  const int c1 = 128 * rowsum + (1 << (S1 - 1));
  int tt[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) tt[g] = ((ah[g] << 8) + al[g] + c1) >> S1;      // low 16 bits = (short) wrap
  // pass 2: D[k][x] = sum_j T'[j][k] * M[x][j] = out[x][k]  (A = T'^T from the accumulator, B[j][x] = M[x][j])
  op16 h2, l2;
  planes_from_regs(tt, h2, l2, 0x80808080u);
  const i32x16 bh = mfma_i8(h2, t_kap, zero), bl = mfma_i8(l2, t_kap, zero);
  const int c2 = 128 * rowsum + (1 << (S2 - 1));
#pragma unroll
  for (int g = 0; g < 16; ++g) o[g] = ((bh[g] << 8) + bl[g] + c2) >> S2;
}

// inverse 2-D core.  (hi, lo') = byte planes of the input rows with K order matching the identity table
// `t_ident`; c2 = per-(half, register) constants 128 * colsum(M)[kappa(h,g)] + 2048 (LDS, this lane half's 16).
// o[g] = out[r][kappa(h, g)], clipped to int16.  dct-generic.c:514-565, :578-587: shifts 7 and 12.
__device__ __forceinline__ void inv32_core(const op16 &hi, const op16 &lo, const op16 &t_ident, const op16 &t_col, int colsum,
                                           const int *c2, int (&o)[16])
{
  const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  // transpose through the matrix core: D = in * I puts column r of `in` on lane r (rows kappa(h,g) in registers)
  const i32x16 xh = mfma_i8(hi, t_ident, zero), xl = mfma_i8(lo, t_ident, zero);
  int th[16], tl[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) { th[g] = xh[g]; tl[g] = xl[g]; }
  op16 ph, pl, dummy;
  planes_from_regs(th, dummy, ph, 0u);           // the planes are already split: pack the low bytes of each
  planes_from_regs(tl, dummy, pl, 0u);
  // pass 1: U^T = in^T * M  (A = in^T, B[k2][j'] = M[k2][j']); D[k][j']: row k = kappa(h,g), col j' = r
  const i32x16 ah = mfma_i8(ph, t_col, zero), al = mfma_i8(pl, t_col, zero);
  const int c1 = 128 * colsum + (1 << 6);
  int uu[16];
#pragma unroll
  for (int g = 0; g < 16; ++g) uu[g] = clip16(((ah[g] << 8) + al[g] + c1) >> 7);
  // pass 2: D[i'][j'] = sum_k M[k][i'] * U[j'][k] = out[j'][i']  (A[i'][k] = M[k][i'], B = U^T from the accumulator)
  op16 h2, l2;
  planes_from_regs(uu, h2, l2, 0x80808080u);
  const i32x16 bh = mfma_i8(t_col, h2, zero), bl = mfma_i8(t_col, l2, zero);
#pragma unroll
  for (int g = 0; g < 16; ++g) o[g] = clip16(((bh[g] << 8) + bl[g] + c2[g]) >> 12);
}

// fills the 2 x 16 table of inverse pass-2 constants (call with all threads, then __syncthreads())
struct dct32_c2_table {
  int v[32];         // [half][register]: 128 * colsum(M)[kappa(h, g)] + 2048
  constexpr dct32_c2_table() : v()
  {
    for (int i = 0; i < 32; ++i) {
      const int h = i >> 4, g = i & 15, row = (g & 3) + 8 * (g >> 2) + 4 * h;
      int cs = 0;
      for (int n = 0; n < 32; ++n) cs += dct_coef(32, n, row);
      v[i] = 128 * cs + (1 << 11);
    }
  }
};
static __constant__ dct32_c2_table c_dct32_c2 = dct32_c2_table();
template <int N, bool DST>
struct tile_c2_table {
  int v[32];
  constexpr tile_c2_table() : v()
  {
    for (int i = 0; i < 32; ++i) {
      const int h = i >> 4, g = i & 15, row = (g & 3) + 8 * (g >> 2) + 4 * h;
      int cs = 0;
      for (int n = 0; n < 32; ++n) cs += tile_mx<N, DST>(n, row);
      v[i] = 128 * cs + (1 << 11);
    }
  }
};
static __constant__ tile_c2_table<16, false> c_dct16x4_c2 = tile_c2_table<16, false>();
static __constant__ tile_c2_table<4, false> c_dct4x64_c2 = tile_c2_table<4, false>();
static __constant__ tile_c2_table<4, true> c_dst4x64_c2 = tile_c2_table<4, true>();
__device__ __forceinline__ void fill_inv_c2(int (*s_c2)[16])
{
  if (threadIdx.x < 32) s_c2[threadIdx.x >> 4][threadIdx.x & 15] = c_dct32_c2.v[threadIdx.x];
}
template <int N, bool DST>
__device__ __forceinline__ void fill_inv_c2_tile(int (*s_c2)[16])
{
  if (threadIdx.x < 32)
    s_c2[threadIdx.x >> 4][threadIdx.x & 15] = N == 16 ? c_dct16x4_c2.v[threadIdx.x] : (DST ? c_dst4x64_c2.v[threadIdx.x] : c_dct4x64_c2.v[threadIdx.x]);
}
template <int N, bool DST>
__device__ __forceinline__ const dct32_lane_consts &tile_lane_consts(int lane)
{
  if (N == 32) return c_dct32_lanes.l[lane];
  if (N == 16) return c_dct16x4_lanes.l[lane];
  return DST ? c_dst4x64_lanes.l[lane] : c_dct4x64_lanes.l[lane];
}

// 16x16 blocks in a tile: the linear 16-byte chunk c of FOUR consecutive blocks (2 KiB; block c >> 5, row (c >> 1) & 15, half c & 1)
// <-> the chunk of the 2 x 2 tile it occupies (block b at row half b & 1, column half b >> 1)
__device__ __forceinline__ int tile_chunk16(int c)
{
  const int b = c >> 5, row = (c >> 1) & 15, half = c & 1;
  return ((b & 1) * 16 + row) * 4 + (b >> 1) * 2 + half;
}
// 4x4 blocks in a tile: the linear chunk c of SIXTY-FOUR consecutive blocks (2 KiB) is rows 2p, 2p + 1 (p = c & 1) of block
// c >> 1, two 8-byte pieces of the tile: block b sits at block row b >> 3, block column b & 7.  Byte address of the FIRST piece
// (tile row 4 (b >> 3) + 2p, columns 4 (b & 7) ..) in the swizzled LDS tile; the second piece is the next tile row.
__device__ __forceinline__ int tile_piece4(int c, int second)
{
  const int b = c >> 1, row = 4 * (b >> 3) + 2 * (c & 1) + second, bx = b & 7;
  return slot_of(4 * row + (bx >> 1)) * 16 + (bx & 1) * 8;
}

}  // namespace kvzhip
