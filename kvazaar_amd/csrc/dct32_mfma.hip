// dct32_mfma.hip -- 32x32 forward / inverse integer DCT on the CDNA4 matrix cores.
//
// Reference: src/strategies/generic/dct-generic.c:458-597 (partial_butterfly_32 /
// partial_butterfly_inverse_32, DCT_NXN_GENERIC / IDCT_NXN_GENERIC).  Each 1-D pass
// is an exact integer matrix product with the 32x32 coefficient matrix M
// (|M| <= 90, fits int8).  The int16 operand X is split into byte planes,
//     X = 256 * Xh + Xl' + 128,   Xh = X >> 8,  Xl' = (X & 255) - 128  (both int8),
// so that   M*X = 256 * (M*Xh) + (M*Xl') + 128 * (M*1)
// is two v_mfma_i32_32x32x32_i8 per pass plus a per-row constant; every partial
// sum is exact in int32, so the results are bit-identical to the butterflies.
// rocprof showed the VALU/LDS butterfly kernel (dct.hip) at 3.3 TB/s, VALU- and
// LDS-issue bound; the MFMA form leaves both nearly idle and runs at HBM speed.
//
// One wave owns one 32x32 block at a time and needs no barrier:
//   * HBM is touched only with fully coalesced 16-byte-per-lane accesses (1 KiB per
//     wave instruction); a 2 KiB wave-private LDS tile (XOR-swizzled 16-byte slots,
//     conflict-free for ds_write_b128 / ds_read_b128 / ds_write_b64) turns the
//     linear chunks into the A-operand layout (lane (r, h) = row r, columns
//     16h..16h+15) and the row-per-lane result back into linear chunks.  Direct
//     operand-shaped global accesses (16 B at a 64 B stride) measured 4.0 TB/s,
//     the staged form reaches the streaming rate;
//   * the 32x32 i32 accumulator tile of pass 1 (column on the lane, 16 rows in
//     registers) is re-used in place as the operand of pass 2, whose contraction
//     index is exactly the accumulator's row index -- no data movement;
//   * the operand orientation of pass 2 is chosen so that each lane ends up with
//     one output ROW in 4 runs of 4 consecutive int16, stored as 4 x 8 bytes.
// K-index permutation: an MFMA computes sum_k A[i][k]*B[k][j] for whatever
// bijection maps (lane half h, element e) to k, as long as A and B use the same
// one; kappa(h, e) below is the accumulator's row map, natural order 16h+e is
// used where the operand comes from memory.  (Lane maps verified with exact
// integer data on MI355X: tools/mfma_probe.hip.)
#include "kvz_hip_internal.h"
#include "transform_core.h"

using namespace kvzhip;

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
__constant__ m32_table c_m32 = m32_table();

// accumulator row of register g in lane half h: rows (g&3) + 8*(g>>2) + 4h
__device__ __forceinline__ int kappa(int h, int e) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

union op16 { i32x4 v; signed char b[16]; u32 w[4]; };

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
  const u32x4v a = *(const u32x4v *)(tile + slot_of(4 * r + 2 * h) * 16);
  const u32x4v b = *(const u32x4v *)(tile + slot_of(4 * r + 2 * h + 1) * 16);
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
  const u32x4v a = *(const u32x4v *)(tile + slot_of(lane) * 16);
  const u32x4v b = *(const u32x4v *)(tile + slot_of(64 + lane) * 16);
  *((u32x4v *)blk + lane) = a;
  *((u32x4v *)blk + 64 + lane) = b;
}

template <bool INVERSE>
__global__ __launch_bounds__(256, 4) void dct32_mfma_kernel(const i16 *__restrict__ in, i16 *__restrict__ out, size_t count)
{
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const signed char *M = c_m32.v;
  const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };

  // constant operands (built once per wave)
  op16 t_nat, t_kap, t_col, t_id;
  int rowsum = 0, colsum = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    t_nat.b[e] = M[r * 32 + 16 * h + e];            // M[r][16h+e]
    t_kap.b[e] = M[r * 32 + kappa(h, e)];           // M[r][kappa(h,e)]
    t_col.b[e] = M[kappa(h, e) * 32 + r];           // M[kappa(h,e)][r]
    t_id.b[e] = (16 * h + e == r) ? 1 : 0;          // identity, natural K order
  }
  for (int n = 0; n < 32; ++n) { rowsum += M[r * 32 + n]; colsum += M[n * 32 + r]; }
  __shared__ __attribute__((aligned(16))) u8 s_tile[4][2048];
  u8 *tile = s_tile[threadIdx.x >> 6];               // wave-private: DS ops of one wave execute in order, no barrier

  // inverse pass 2: the plane offset 128 * (column sum of M) + rounding depends on the output ROW, i.e. on
  // (lane half, register): a 2 x 16 table in LDS, read back as broadcasts, instead of 16 live registers
  __shared__ __attribute__((aligned(16))) int s_c2[2][16];
  if (INVERSE) {
    if (threadIdx.x < 32) {
      const int row = kappa(threadIdx.x >> 4, threadIdx.x & 15);
      int cs = 0;
      for (int n = 0; n < 32; ++n) cs += M[n * 32 + row];
      s_c2[threadIdx.x >> 4][threadIdx.x & 15] = 128 * cs + (1 << 11);
    }
    __syncthreads();
  }

  size_t t = wave;
  u32x4v cur[2], nx1[2], nx2[2];
  if (t < count) load_chunks(in + t * 1024, lane, cur);
  if (t + nwaves < count) load_chunks(in + (t + nwaves) * 1024, lane, nx1);
  for (; t < count; t += nwaves) {
    const size_t tn = t + 2 * nwaves;
    if (tn < count) load_chunks(in + tn * 1024, lane, nx2);      // keep two of the wave's next blocks in flight
    u32 d[8];
    chunks_to_rows(tile, lane, r, h, cur, d);
    op16 hi, lo;
    planes_from_rows(d, hi, lo);
    int o[16];
    if (!INVERSE) {
      // pass 1: T' = S * M^T  (A = S rows, B[n][k] = M[k][n]); D[j][k]: row j = kappa(h,g), col k = r
      const i32x16 ah = mfma_i8(hi, t_nat, zero), al = mfma_i8(lo, t_nat, zero);
      const int c1 = 128 * rowsum + (1 << 3);
      int tt[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) tt[g] = ((ah[g] << 8) + al[g] + c1) >> 4;       // low 16 bits = (short) wrap
      // pass 2: D[k][x] = sum_j T'[j][k] * M[x][j] = out[x][k]  (A = T'^T from the accumulator, B[j][x] = M[x][j])
      op16 h2, l2;
      planes_from_regs(tt, h2, l2, 0x80808080u);
      const i32x16 bh = mfma_i8(h2, t_kap, zero), bl = mfma_i8(l2, t_kap, zero);
      const int c2 = 128 * rowsum + (1 << 10);
#pragma unroll
      for (int g = 0; g < 16; ++g) o[g] = ((bh[g] << 8) + bl[g] + c2) >> 11;
    } else {
      // transpose through the matrix core: D = in * I puts column r of `in` on lane r (rows kappa(h,g) in registers)
      const i32x16 xh = mfma_i8(hi, t_id, zero), xl = mfma_i8(lo, t_id, zero);
      int th[16], tl[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) { th[g] = xh[g]; tl[g] = xl[g]; }
      op16 ph, pl, dummy;
      // the planes are already split: pack the low bytes of each
      planes_from_regs(th, dummy, ph, 0u);
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
      for (int g = 0; g < 16; ++g) o[g] = clip16(((bh[g] << 8) + bl[g] + s_c2[h][g]) >> 12);
    }
    rows_to_chunks_store(tile, lane, r, h, o, out + t * 1024);
    cur[0] = nx1[0]; cur[1] = nx1[1]; nx1[0] = nx2[0]; nx1[1] = nx2[1];
  }
}

namespace kvzhip {
int launch_dct32_mfma(bool inverse, const i16 *in, i16 *out, size_t count, hipStream_t st)
{
  // 4 waves per workgroup, each wave strides over blocks; enough workgroups for ~4 waves per SIMD
  size_t wgs = (count + 3) / 4;
  const size_t cap = (size_t)num_cus() * (size_t)(inverse ? tuning("idct32_wgs_per_cu", 2) : tuning("dct32_wgs_per_cu", 2));       // measured optimum: 2 workgroups (8 waves) per CU, deeper queues only add HBM contention
  if (wgs > cap) wgs = cap;
  if (inverse) hipLaunchKernelGGL((dct32_mfma_kernel<true>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  else hipLaunchKernelGGL((dct32_mfma_kernel<false>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  KVZ_CHECK_LAUNCH("dct32_mfma_kernel");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip
