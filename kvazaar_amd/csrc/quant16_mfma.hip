// quant16_mfma.hip -- fused kvz_quantize_residual for 16x16 TUs on the matrix cores, two TUs per wave step.
//
// Reference: src/strategies/generic/quant-generic.c:180-273 (rdoq off, no transform skip, sign hiding off --
// the other variants stay on quantize_residual_kernel in quant.hip).  Built on dct16_mfma_core.h:
//   ref / pred: 8 pixels per lane (the lane's natural chunk: row r & 15, columns 8h .. 8h+7 of TU r >> 4)
//   -> residual (packed int16) -> forward DCT of the pair (4 MFMA) -> quant in registers
//   -> v_permlane32_swap to the natural chunk -> coeff_out, 16 B per lane
//   -> dequant -> inverse DCT of the pair (6 MFMA) -> + pred, clip -> rec_out, 8 B per lane
// No LDS staging and no barrier in the loop; has_coeffs of each TU is a ballot over its 32 lanes.
// HBM traffic per TU: 256 B ref + 256 B pred + 256 B rec + 512 B coeff = 5*N*N.
#include "dct16_mfma_core.h"

using namespace kvzhip;

typedef short v2s __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));

struct q16_consts {
  int q_bits, add, flat_qc;
  const int32_t *qtable;
  int dq_mode, dq_shift, dq_add, dq_scale;
  const int32_t *dqtable;
};

__device__ __forceinline__ int q16_quant(int c, int n, const q16_consts &k)
{
  const int a = c < 0 ? -c : c;
  int level;
  if (k.qtable) level = (int)(((long long)a * k.qtable[n] + k.add) >> k.q_bits);
  else level = (int)((__umul24((u32)a, (u32)k.flat_qc) + (u32)k.add) >> k.q_bits);      // < 2^31: |c| <= 2^15, qc < 2^15, add < 2^24
  level = c < 0 ? -level : level;
  return clip16(level);
}
__device__ __forceinline__ int q16_dequant(int q, int n, const q16_consts &k)
{
  if (k.dq_mode == 0) return clip16((int)((u32)__mul24(q, k.dq_scale) + (u32)k.dq_add) >> k.dq_shift);
  const int d = k.dqtable[n];
  if (k.dq_mode == 1) return clip16((q * d + k.dq_add) >> k.dq_shift);
  return clip16((int)((u32)clip16(q * d) << k.dq_shift));
}

__global__ __launch_bounds__(256, 2) void quantize_residual16_mfma_kernel(const u8 *__restrict__ ref_in, const u8 *pred_in, u8 *rec_out,
                                                                           i16 *__restrict__ coeff_out, i32 *__restrict__ has_coeffs,
                                                                           size_t count, q16_consts k,
                                                                           u32 *__restrict__ ssd_out, u32 *__restrict__ abs_sum_out)
{
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (size_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const size_t npairs = (count + 1) >> 1;

  dct16_lane kf, ki;
  dct16_setup<false>(r, h, kf);
  dct16_setup<true>(r, h, ki);
  __shared__ int s_c2[2][8];
  dct16_fill_c2(s_c2);
  __syncthreads();

  const int chunk = 2 * r + h;                          // 8-pixel (16-byte coefficient) chunk of the pair owned by this lane
  const int tu = r >> 4, row = r & 15;
  const unsigned long long tu_mask = tu ? 0xffff0000ffff0000ull : 0x0000ffff0000ffffull;
  auto load = [&](size_t p, u32x2v &rv, u32x2v &pv) {
    const bool tail = (2 * p + 1 >= count);
    const int ch = (tail && chunk >= 32) ? chunk - 32 : chunk;   // a single trailing TU: the second TU's lanes mirror the first
    rv = __builtin_nontemporal_load((const u32x2v *)(ref_in + p * 512) + ch);
    pv = *((const u32x2v *)(pred_in + p * 512) + ch);
  };

  size_t p = wave;
  u32x2v rv, pv, rn, pn;
  if (p < npairs) load(p, rv, pv);
  for (; p < npairs; p += nwaves) {
    const size_t pnx = p + nwaves;
    if (pnx < npairs) load(pnx, rn, pn);               // prefetch the wave's next pair (never the one being written)
    // residual, natural order: 8 packed int16
    const v2s a0 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rv.x, 0x0c010c00u)), a1 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rv.x, 0x0c030c02u));
    const v2s a2 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rv.y, 0x0c010c00u)), a3 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, rv.y, 0x0c030c02u));
    const v2s b0 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, pv.x, 0x0c010c00u)), b1 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, pv.x, 0x0c030c02u));
    const v2s b2 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, pv.y, 0x0c010c00u)), b3 = __builtin_bit_cast(v2s, __builtin_amdgcn_perm(0u, pv.y, 0x0c030c02u));
    const u32x4v resid = { __builtin_bit_cast(u32, a0 - b0), __builtin_bit_cast(u32, a1 - b1), __builtin_bit_cast(u32, a2 - b2), __builtin_bit_cast(u32, a3 - b3) };

    int c[8];
    dct16_fwd_pair(resid, kf, c);
    int qv[8], any = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      qv[g] = q16_quant((int)(short)c[g], row * 16 + acc16_col(h, g), k);
      any |= qv[g];
    }
    const unsigned long long bal = __ballot(any != 0);
    const bool has = (bal & tu_mask) != 0ull;            // this lane's TU
    const bool live = (2 * p + 1 < count) || chunk < 32;
    const u32x4v qchunk = acc16_to_chunk(qv);
    if (live) __builtin_nontemporal_store(qchunk, (u32x4v *)(coeff_out + p * 512) + chunk);

    u32x2v out = pv;
    if (bal != 0ull) {                                   // wave-uniform: at least one of the two TUs has coefficients
      int dq[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) dq[g] = q16_dequant(qv[g], row * 16 + acc16_col(h, g), k);
      const u32x4v dchunk = acc16_to_chunk(dq);
      int res[8];
      dct16_inv_pair(dchunk, ki, s_c2[h], res);
      const u32x4v rchunk = acc16_to_chunk(res);         // 8 residuals, natural order
      const u32 rw[4] = { rchunk.x, rchunk.y, rchunk.z, rchunk.w };
      const u32 pw[2] = { pv.x, pv.y };
      u32 ow[2] = { 0u, 0u };
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int px = (int)((pw[i >> 2] >> (8 * (i & 3))) & 255u);
        const int rs = (int)(short)((rw[i >> 1] >> (16 * (i & 1))) & 0xffffu);
        const int val = (int)(short)(rs + px);           // int16_t val = residual + pred (quant-generic.c:255)
        ow[i >> 2] |= (u32)(val < 0 ? 0 : (val > 255 ? 255 : val)) << (8 * (i & 3));
      }
      if (has) { out.x = ow[0]; out.y = ow[1]; }         // a TU without coefficients keeps its prediction (:262-271)
    }
    if (ssd_out) {
      // rd=0 TU cost inputs (search.c:291, rdo.c:219): per-TU sums over the TU's 32 lanes {r & 15, h}
      u32 sq2 = 0, sab = 0;
      const u32 rr[2] = { rv.x, rv.y }, oo[2] = { out.x, out.y };
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int dd = (int)((rr[i >> 2] >> (8 * (i & 3))) & 255u) - (int)((oo[i >> 2] >> (8 * (i & 3))) & 255u);
        sq2 += (u32)(dd * dd);
        sab += (u32)(qv[i] < 0 ? -qv[i] : qv[i]);
      }
      sq2 = group_sum<16>(sq2); sab = group_sum<16>(sab);
      sq2 += (u32)__shfl_xor((int)sq2, 32, 64); sab += (u32)__shfl_xor((int)sab, 32, 64);
      if (live && row == 0 && h == 0) { ssd_out[2 * p + tu] = sq2; abs_sum_out[2 * p + tu] = sab; }
    }
    if (live) {
      *((u32x2v *)(rec_out + p * 512) + chunk) = out;
      if (row == 0 && h == 0) has_coeffs[2 * p + tu] = has ? 1 : 0;
    }
    rv = rn; pv = pn;
  }
}

namespace kvzhip {
// consts are produced by quant.hip's make_consts (same field meaning)
int launch_quantize_residual16_mfma(const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, i32 *has_coeffs, size_t count,
                                    int q_bits, int add, int flat_qc, const int32_t *qtable, int dq_mode, int dq_shift, int dq_add,
                                    int dq_scale, const int32_t *dqtable, u32 *ssd_out, u32 *abs_sum_out, hipStream_t st)
{
  q16_consts k = { q_bits, add, flat_qc, qtable, dq_mode, dq_shift, dq_add, dq_scale, dqtable };
  const size_t npairs = (count + 1) / 2;
  size_t wgs = (npairs + 3) / 4;
  const size_t cap = (size_t)num_cus() * (size_t)tuning("qr16_wgs_per_cu", 16)       /* 4: 2.71, 8: 2.88, 16: 2.97, 32: 2.95, 64: 2.87 TB/s */;
  if (wgs > cap) wgs = cap;
  hipLaunchKernelGGL(quantize_residual16_mfma_kernel, dim3((unsigned)wgs), dim3(256), 0, st, ref_in, pred_in, rec_out, coeff_out, has_coeffs,
                     count, k, ssd_out, abs_sum_out);
  KVZ_CHECK_LAUNCH("quantize_residual16_mfma_kernel");
  return KVZ_HIP_OK;
}
}  // namespace kvzhip
