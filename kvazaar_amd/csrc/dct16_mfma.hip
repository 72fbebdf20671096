// dct16_mfma.hip -- 16x16 forward / inverse integer DCT on the matrix cores, two
// blocks per v_mfma_i32_32x32x32_i8.
//
// Reference: src/strategies/generic/dct-generic.c:368-455, :567-597 (N = 16).
// Same method as dct32_mfma.hip (byte planes X = 256*Xh + Xl' + 128, exact int32
// partial sums, accumulator tile re-used as the next operand).  Two 16x16 blocks
// a, b are stacked into the 32 rows of the MFMA tile:
//   pass 1   D1 = [S_a; S_b] * M16^T             (K: 16 live of 32, dead K operands are 0)
//   pass 2   D2[k][x] = sum_j T'[j][k] * B2[j][x], B2 block diagonal in (block of j, block of x)
// so that lane x of the result holds row (x & 15) of block (x >> 4).  The blocks are
// independent in HBM; a lane loads / stores exactly one 16-byte chunk of the pair's
// 1 KiB (chunk 2r + h: row r, columns 8h .. 8h+7): every global access instruction
// covers a dense 1 KiB, no LDS staging is needed; v_permlane32_swap turns the
// accumulator's column order into contiguous columns before the store.
#include "dct32_mfma_core.h"

using namespace kvzhip;

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

// byte planes of the lane's 8 live int16 (4 dwords); elements 8..15 are dead K (zero in both planes)
__device__ __forceinline__ void planes8(const u32x4v &c, op16 &hi, op16 &lo)
{
  lo.w[0] = __builtin_amdgcn_perm(c.y, c.x, 0x06040200u) ^ 0x80808080u;
  lo.w[1] = __builtin_amdgcn_perm(c.w, c.z, 0x06040200u) ^ 0x80808080u;
  hi.w[0] = __builtin_amdgcn_perm(c.y, c.x, 0x07050301u);
  hi.w[1] = __builtin_amdgcn_perm(c.w, c.z, 0x07050301u);
  lo.w[2] = lo.w[3] = hi.w[2] = hi.w[3] = 0u;
}

template <bool INVERSE>
__global__ __launch_bounds__(256, 4) void dct16_mfma_kernel(const i16 *__restrict__ in, i16 *__restrict__ out, size_t count)
{
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const signed char *M = c_m16.v;
  const i32x16 zero = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
  const size_t npairs = (count + 1) >> 1;

  // constant operands.  Natural K order of the loaded operand: element e < 8 <-> column 8h + e, e >= 8 dead.
  op16 tA, tB;       // forward: tA = pass-1 B (M16[k][8h+e], k < 16), tB = pass-2 B (block diagonal M16[x&15][j&15])
                     // inverse: tA = identity (natural K), tB = block diagonal M16[k2&15][j'&15] (pass 1) ...
  op16 tC;           // inverse pass 2 A: M16[k][i'] for k, i' < 16 (K = kappa order, k >= 16 dead)
  int sum = 0;       // forward: row sum of M16 row (r & 15); inverse: column sum of column (r & 15)
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int kk = kappa(h, e);
    if (!INVERSE) {
      tA.b[e] = (e < 8 && r < 16) ? M[r * 16 + 8 * h + e] : 0;
      tB.b[e] = ((kk >> 4) == (r >> 4)) ? M[(r & 15) * 16 + (kk & 15)] : 0;
      tC.b[e] = 0;
    } else {
      tA.b[e] = (e < 8 && r < 16 && 8 * h + e == r) ? 1 : 0;
      tB.b[e] = ((kk >> 4) == (r >> 4)) ? M[(kk & 15) * 16 + (r & 15)] : 0;
      tC.b[e] = (kk < 16 && r < 16) ? M[kk * 16 + r] : 0;
    }
  }
  for (int n = 0; n < 16; ++n) sum += INVERSE ? M[n * 16 + (r & 15)] : M[(r & 15) * 16 + n];

  // inverse pass 2: 128 * (column sum of M16)[kappa(h,g)] + 2048 for the 8 live registers of each lane half
  __shared__ int s_c2[2][8];
  if (INVERSE) {
    if (threadIdx.x < 16) {
      const int hh = threadIdx.x >> 3, g = threadIdx.x & 7, row = kappa(hh, g);
      int cs = 0;
      for (int n = 0; n < 16; ++n) cs += M[n * 16 + row];
      s_c2[hh][g] = 128 * cs + (1 << 11);
    }
    __syncthreads();
  }

  const int chunk = 2 * r + h;
  // a single trailing block (odd count): the second block's 32 chunks are clamped onto the first
  auto load = [&](size_t p, u32x4v &c) {
    const bool tail = (2 * p + 1 >= count);
    const int ch = (tail && chunk >= 32) ? chunk - 32 : chunk;
    c = __builtin_nontemporal_load((const u32x4v *)(in + p * 512) + ch);
  };

  size_t p = wave;
  constexpr int DEPTH = 5;                             // pairs in flight per wave (1 KiB each)
  u32x4v q[DEPTH];
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i)
    if (p + i * nwaves < npairs) load(p + i * nwaves, q[i]);
  for (; p < npairs; p += nwaves) {
    if (p + (DEPTH - 1) * nwaves < npairs) load(p + (DEPTH - 1) * nwaves, q[DEPTH - 1]);
    const u32x4v cur = q[0];
    op16 hi, lo;
    planes8(cur, hi, lo);
    int o[8];
    if (!INVERSE) {
      // pass 1: D1[j][k] = sum_n S[j][n] M16[k][n]; rows j (both blocks) in registers, column k = lane (k < 16 live)
      const i32x16 ah = mfma_i8(hi, tA, zero), al = mfma_i8(lo, tA, zero);
      const int c1 = 128 * sum + (1 << 2);
      int tt[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) tt[g] = ((ah[g] << 8) + al[g] + c1) >> 3;
      // pass 2: D2[k][x] = sum_j T'[j][k] * (same block ? M16[x&15][j&15] : 0) = out_{x>>4}[x&15][k]
      op16 h2, l2;
      planes_from_regs(tt, h2, l2, 0x80808080u);
      const i32x16 bh = mfma_i8(h2, tB, zero), bl = mfma_i8(l2, tB, zero);
      const int c2 = 128 * sum + (1 << 9);
#pragma unroll
      for (int g = 0; g < 8; ++g) o[g] = ((bh[g] << 8) + bl[g] + c2) >> 10;
    } else {
      // transpose both blocks through the matrix core: column c (< 16) of block a / b on lane c, rows kappa < 16 / >= 16
      const i32x16 xh = mfma_i8(hi, tA, zero), xl = mfma_i8(lo, tA, zero);
      int th[16], tl[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) { th[g] = xh[g]; tl[g] = xl[g]; }
      op16 ph, pl, dummy;
      planes_from_regs(th, dummy, ph, 0u);
      planes_from_regs(tl, dummy, pl, 0u);
      // pass 1: D[k][j'] = sum_k2 in[k2][k] * (same block ? M16[k2&15][j'&15] : 0) = tmp_{j'>>4}[k][j'&15]
      const i32x16 ah = mfma_i8(ph, tB, zero), al = mfma_i8(pl, tB, zero);
      const int c1 = 128 * sum + (1 << 6);
      int uu[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) uu[g] = clip16(((ah[g] << 8) + al[g] + c1) >> 7);
      // pass 2: D2[i'][j'] = sum_{k<16} M16[k][i'] * U[j'][k]; registers with kappa >= 16 are dead K (tC is 0 there)
      op16 h2, l2;
      planes_from_regs(uu, h2, l2, 0x80808080u);
      const i32x16 bh = mfma_i8(tC, h2, zero), bl = mfma_i8(tC, l2, zero);
#pragma unroll
      for (int g = 0; g < 8; ++g) o[g] = clip16(((bh[g] << 8) + bl[g] + s_c2[h][g]) >> 12);
    }
    // lane (x, h): o[0..3] = columns 4h..4h+3, o[4..7] = columns 8+4h..; swap halves -> columns 8h .. 8h+7
    u32 p0x = __builtin_amdgcn_perm((u32)o[1], (u32)o[0], 0x05040100u), p0y = __builtin_amdgcn_perm((u32)o[3], (u32)o[2], 0x05040100u);
    u32 p1x = __builtin_amdgcn_perm((u32)o[5], (u32)o[4], 0x05040100u), p1y = __builtin_amdgcn_perm((u32)o[7], (u32)o[6], 0x05040100u);
    const u32x2w sx = __builtin_amdgcn_permlane32_swap(p0x, p1x, false, false);
    const u32x2w sy = __builtin_amdgcn_permlane32_swap(p0y, p1y, false, false);
    const u32x4v ov = { sx.x, sy.x, sx.y, sy.y };
    if (2 * p + 1 < count || chunk < 32) __builtin_nontemporal_store(ov, (u32x4v *)(out + p * 512) + chunk);
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) q[i] = q[i + 1];
  }
}

namespace kvzhip {
int launch_dct16_mfma(bool inverse, const i16 *in, i16 *out, size_t count, hipStream_t st)
{
  const size_t npairs = (count + 1) / 2;
  size_t wgs = (npairs + 3) / 4;
  const size_t cap = (size_t)num_cus() * (size_t)tuning(inverse ? "idct16_wgs_per_cu" : "dct16_wgs_per_cu", inverse ? 6 : 3);
  if (wgs > cap) wgs = cap;
  if (inverse) hipLaunchKernelGGL((dct16_mfma_kernel<true>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  else hipLaunchKernelGGL((dct16_mfma_kernel<false>), dim3((unsigned)wgs), dim3(256), 0, st, in, out, count);
  KVZ_CHECK_LAUNCH("dct16_mfma_kernel");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip
